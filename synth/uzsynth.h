/* uzsynth.h -- benchmark-scale synthetic 30x pile-ups, one source for CPU and GPU.
 *
 * TEST / BENCH INFRASTRUCTURE (not part of the product).  Generates, for every DNM
 * of a batch, the coordinate-sorted alignment records of the kid within
 * +-(search_dist + pad) of the DNM, directly as the column arrays of
 * include/uz_types.h (uz_reads_view).  Everything is a pure function of
 * (seed, DNM index, pair index, base index) through a counter-based hash, so
 *   * hipcc builds a kernel that fills HBM in place (no host copy of the 100k-DNM table),
 *   * gcc builds the same functions so that any subset of DNMs can be regenerated on the
 *     host for the CPU oracle (bench.py's cpu_baseline sample, parity tests).
 *
 * Model: 151-bp pairs, alternating paternal / maternal
 * haplotype, fragment starts stratified over the window, insert ~450 (sd ~58) clipped to
 * [302, 900]; bases follow the kid's haplotype alleles at the sites of the sites table and
 * the DNM allele on the origin haplotype; 0.4 % substitution errors, base quality 37
 * (3 %: 12), MAPQ 60 (3 %: 0), 1 % of left reads soft-clipped, 0.5 % with a 1-3 bp indel;
 * small insertion / deletion DNMs put an I / D operation behind the anchor base.
 */
#ifndef UZSYNTH_H
#define UZSYNTH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define UZS_HD __host__ __device__ static inline
#else
#define UZS_HD static inline
#endif

#define UZS_READLEN 151
#define UZS_ROW 160 /* bytes per seq / qual row (16-byte aligned) */
#define UZS_MAXOPS 3

typedef struct uzs_cfg {
    uint64_t seed;
    int32_t n_pairs;    /* pairs per DNM (both haplotypes together) */
    int32_t half_width; /* reads are laid out within +-half_width of the DNM */
    int32_t n_dnms;
    int32_t reserved;
} uzs_cfg;

typedef struct uzs_sites { /* the columns of the sites table the generator needs */
    const int64_t *contig_off;
    const int32_t *pos;
    const uint8_t *ref_base, *alt_base;
    const uint8_t *khap; /* bit0 allele on the kid's paternal haplotype, bit1 maternal */
    int32_t n_contigs;
} uzs_sites;

typedef struct uzs_dnms {
    const int32_t *contig, *pos;
    const int32_t *site_idx; /* the DNM's own record in the sites table */
    const uint8_t *kind;     /* 0 SNV, 1 insertion, 2 deletion */
    const uint8_t *len;      /* inserted / deleted bases */
    const uint8_t *origin;   /* 0 paternal haplotype carries the DNM, 1 maternal */
} uzs_dnms;

typedef struct uzs_out { /* uz_reads_view columns; block of DNM d starts at (d - d0) * 2 * n_pairs */
    int32_t *start, *end;
    uint16_t *flag;
    uint8_t *mapq, *aux;
    int32_t *tlen;
    uint32_t *qname;
    int32_t *mate;
    uint32_t *cigar_off;
    uint16_t *n_cigar;
    uint32_t *cigar; /* UZS_MAXOPS per record */
    uint16_t *l_seq;
    uint32_t *sq_off16;
    uint8_t *seq, *qual; /* UZS_ROW per record */
} uzs_out;

UZS_HD uint64_t uzs_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
UZS_HD uint64_t uzs_h(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    return uzs_mix(uzs_mix(uzs_mix(seed ^ (a * 0xD6E8FEB86659FD93ULL)) ^ (b * 0xA24BAED4963EE407ULL)) ^ c);
}
UZS_HD uint8_t uzs_acgt(unsigned k) { return (uint8_t)("ACGT"[k & 3]); }
UZS_HD uint8_t uzs_refbase(int32_t contig, int64_t pos) {
    return uzs_acgt((unsigned)(uzs_mix((uint64_t)pos * 0x9E3779B1ULL + (uint64_t)contig) >> 20));
}
UZS_HD uint8_t uzs_other(uint8_t b, unsigned k) {
    const unsigned i = b == 'A' ? 0 : (b == 'C' ? 1 : (b == 'G' ? 2 : 3));
    return uzs_acgt(i + 1 + (k % 3));
}

typedef struct uzs_seg {
    int32_t start, end, tlen;
    uint16_t flag;
    uint8_t mapq, n_ops, hap;
    uint32_t ops[UZS_MAXOPS];
    int32_t carries_dnm;
    uint64_t bseed; /* per-base hash stream of this record */
} uzs_seg;

/* geometry + CIGAR of segment `which` (0 left / 1 right) of pair k of DNM d */
UZS_HD void uzs_segment(const uzs_cfg *c, const uzs_dnms *D, int32_t d, int32_t k, int which, uzs_seg *s) {
    const int64_t p = D->pos[d];
    const int W = c->half_width, L = UZS_READLEN;
    const uint64_t h = uzs_h(c->seed, (uint64_t)d, (uint64_t)k, 1);
    const int hap = k & 1;
    const int per_hap = c->n_pairs / 2;
    const int kk = k >> 1;
    /* fragment starts stratified over [p - W, p + W - 900) per haplotype */
    const int64_t span = 2LL * W - 900;
    const int64_t cell = span / per_hap > 0 ? span / per_hap : 1;
    const int64_t fs = p - W + (int64_t)kk * span / per_hap + (int64_t)((h >> 8) % (uint64_t)cell);
    int ins = 450;
    for (int j = 0; j < 4; j++) ins += (int)((h >> (20 + 8 * j)) & 0xFF) * 100 / 256 - 50;
    if (ins < 2 * L) ins = 2 * L;
    if (ins > 900) ins = 900;
    const int64_t a = which == 0 ? fs : fs + ins - L;
    const uint64_t hv = uzs_h(c->seed, (uint64_t)d, (uint64_t)k, 2 + (uint64_t)which);
    s->hap = (uint8_t)hap;
    s->bseed = uzs_h(c->seed ^ 0x5bd1e995ULL, (uint64_t)d, (uint64_t)(2 * k + which), 7);
    s->flag = (uint16_t)(which == 0 ? (1 | 2 | 32 | 64) : (1 | 2 | 16 | 128));
    s->mapq = (uint8_t)(((hv >> 40) % 100) < 3 ? 0 : 60);
    s->tlen = which == 0 ? ins : -ins;
    s->carries_dnm = hap == (int)D->origin[d];
    const int kind = D->kind[d], dl = D->len[d];
    int64_t start = a;
    int n = 0;
    const int covers_anchor = (a <= p) && (p < a + L - 12) && (p - a >= 1);
    if (s->carries_dnm && kind != 0 && covers_anchor) {
        const int x = (int)(p - a) + 1;
        if (kind == 1) { s->ops[0] = ((uint32_t)x << 4) | 0; s->ops[1] = ((uint32_t)dl << 4) | 1; s->ops[2] = ((uint32_t)(L - x - dl) << 4) | 0; }
        else { s->ops[0] = ((uint32_t)x << 4) | 0; s->ops[1] = ((uint32_t)dl << 4) | 2; s->ops[2] = ((uint32_t)(L - x) << 4) | 0; }
        n = 3;
    } else {
        const unsigned u = (unsigned)(hv % 1000);
        if (which == 0 && u < 10) { /* soft clip at the left end */
            const int cl = 3 + (int)((hv >> 12) % 27);
            s->ops[0] = ((uint32_t)cl << 4) | 4; s->ops[1] = ((uint32_t)(L - cl) << 4) | 0;
            n = 2;
            start = a + cl;
        } else if (which == 0 && u < 15) { /* 1-3 bp indel */
            const int kl = 1 + (int)((hv >> 12) % 3);
            const int x = 20 + (int)((hv >> 16) % (uint64_t)(L - 50));
            if ((hv >> 30) & 1) { s->ops[0] = ((uint32_t)x << 4) | 0; s->ops[1] = ((uint32_t)kl << 4) | 1; s->ops[2] = ((uint32_t)(L - x - kl) << 4) | 0; }
            else { s->ops[0] = ((uint32_t)x << 4) | 0; s->ops[1] = ((uint32_t)kl << 4) | 2; s->ops[2] = ((uint32_t)(L - x) << 4) | 0; }
            n = 3;
        } else { s->ops[0] = ((uint32_t)L << 4) | 0; n = 1; }
    }
    for (int j = n; j < UZS_MAXOPS; j++) s->ops[j] = 0;
    s->n_ops = (uint8_t)n;
    int64_t r = start;
    for (int j = 0; j < n; j++) {
        const int op = s->ops[j] & 15, l = (int)(s->ops[j] >> 4);
        if (op == 0 || op == 2) r += l;
    }
    s->start = (int32_t)start;
    s->end = (int32_t)r;
}

/* first index in [lo, hi) with a[idx] >= v */
UZS_HD int64_t uzs_lower_bound(const int32_t *a, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* site index range of DNM d's read window (all sites a read of this DNM can touch) */
UZS_HD void uzs_site_window(const uzs_cfg *c, const uzs_sites *S, const uzs_dnms *D, int32_t d, int64_t *s_lo, int64_t *s_hi) {
    const int32_t contig = D->contig[d];
    const int64_t lo = S->contig_off[contig], hi = S->contig_off[contig + 1];
    *s_lo = uzs_lower_bound(S->pos, lo, hi, (int64_t)D->pos[d] - c->half_width - 64);
    *s_hi = uzs_lower_bound(S->pos, lo, hi, (int64_t)D->pos[d] + c->half_width + 1200);
}

/* bases and qualities of query indices [i0, i1) of a segment (out[0] is index i0) */
UZS_HD void uzs_fill(const uzs_cfg *c, const uzs_sites *S, const uzs_dnms *D, int32_t d, const uzs_seg *s,
                     int64_t s_lo, int64_t s_hi, int i0, int i1, uint8_t *seq, uint8_t *qual) {
    const int32_t contig = D->contig[d];
    /* walk the CIGAR to the op holding query index i0 */
    int64_t r = s->start;
    int q = 0, j = 0, op = 0, left = 0;
    for (; j < s->n_ops; j++) {
        op = s->ops[j] & 15;
        const int l = (int)(s->ops[j] >> 4);
        if (op == 2) { r += l; continue; }
        if (i0 < q + l) { left = q + l - i0; if (op == 0) r += i0 - q; break; }
        q += l;
        if (op == 0) r += l;
    }
    int64_t si = uzs_lower_bound(S->pos, s_lo, s_hi, r);
    for (int i = i0; i < i1; i++) {
        while (left == 0 && j + 1 < s->n_ops) { /* next op */
            j++;
            op = s->ops[j] & 15;
            const int l = (int)(s->ops[j] >> 4);
            if (op == 2) { r += l; continue; }
            left = l;
        }
        const uint64_t hb = uzs_mix(s->bseed + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
        uint8_t b;
        if (op == 0) {
            while (si < s_hi && (int64_t)S->pos[si] < r) si++;
            b = uzs_refbase(contig, r);
            if (si < s_hi && (int64_t)S->pos[si] == r) {
                if (si == D->site_idx[d]) {
                    b = S->ref_base[si];
                    if (s->carries_dnm && D->kind[d] == 0) b = S->alt_base[si];
                } else if (S->ref_base[si]) {
                    b = ((S->khap[si] >> s->hap) & 1) ? S->alt_base[si] : S->ref_base[si];
                }
            }
            r++;
        } else b = uzs_acgt((unsigned)(hb >> 7)); /* inserted / clipped bases */
        if ((hb % 1000) < 4) b = uzs_other(b, (unsigned)(hb >> 13)); /* 0.4 % substitution errors */
        seq[i - i0] = b;
        qual[i - i0] = (uint8_t)(((hb >> 24) % 100) < 3 ? 12 : 37);
        left--;
    }
}

/* sort key of a segment inside its DNM block: (start - window start) then slot = 2k + which */
UZS_HD uint64_t uzs_key(const uzs_cfg *c, const uzs_dnms *D, int32_t d, int32_t slot, const uzs_seg *s) {
    const int64_t rel = (int64_t)s->start - ((int64_t)D->pos[d] - c->half_width - 64);
    return ((uint64_t)rel << 16) | (uint64_t)slot;
}

/* writes every column of the record that lands at position `pos_in_block` of DNM d's block.
 * inv[slot] = position of slot in the block (for the mate link). */
UZS_HD void uzs_write_record(const uzs_cfg *c, const uzs_dnms *D, int32_t d, int32_t d0, int32_t slot,
                             int32_t pos_in_block, int32_t mate_pos_in_block, const uzs_seg *s, const uzs_out *o) {
    const int64_t base = (int64_t)(d - d0) * 2 * c->n_pairs;
    const int64_t i = base + pos_in_block;
    o->start[i] = s->start;
    o->end[i] = s->end;
    o->flag[i] = s->flag;
    o->mapq[i] = s->mapq;
    o->aux[i] = 1; /* mate on the same contig */
    o->tlen[i] = s->tlen;
    o->qname[i] = (uint32_t)((int64_t)(d - d0) * c->n_pairs + (slot >> 1));
    o->mate[i] = (int32_t)(base + mate_pos_in_block);
    o->cigar_off[i] = (uint32_t)(i * UZS_MAXOPS);
    o->n_cigar[i] = s->n_ops;
    for (int j = 0; j < UZS_MAXOPS; j++) o->cigar[i * UZS_MAXOPS + j] = s->ops[j];
    int ql = 0;
    for (int j = 0; j < s->n_ops; j++) {
        const int op = s->ops[j] & 15;
        if (op == 0 || op == 1 || op == 4) ql += (int)(s->ops[j] >> 4);
    }
    o->l_seq[i] = (uint16_t)ql;
    o->sq_off16[i] = (uint32_t)(i * (UZS_ROW / 16));
}

#endif /* UZSYNTH_H */
