/* uzsynth.h -- benchmark-scale synthetic 30x pile-ups, one source for CPU and GPU.
 *
 * TEST / BENCH INFRASTRUCTURE (not part of the product).  DNMs are placed uniformly over the genome
 * (SURVEY.md 8(d) config 3), so the +-(search_dist + pad) windows of neighbouring DNMs overlap; overlapping
 * windows are merged into CLUSTERS and every cluster gets ONE 30x pile-up, so that neighbouring DNMs share
 * their alignment records as they do in a real BAM.  The table is the concatenation of the clusters'
 * coordinate-sorted record blocks = one coordinate-sorted table.
 * Everything is a pure function of (seed, cluster, pair, base index) through a counter-based hash, so
 *   * hipcc builds kernels that fill HBM in place, directly in the staged format (uz_reads_packed_view:
 *     4-bit bases, quality-below-threshold plane, CIGAR words back to back),
 *   * gcc builds the same functions so that any cluster range can be regenerated on the host in the ASCII
 *     form (uz_reads_view) for the CPU oracle (bench.py's cpu_baseline sample, parity tests).
 *
 * Model: 151-bp pairs, alternating paternal / maternal haplotype, fragment starts stratified over the
 * cluster, insert ~450 (sd ~58) clipped to [302, 900]; bases follow the kid's haplotype alleles at the sites
 * of the sites table and the DNM alleles on their origin haplotypes; 0.4 % substitution errors, base quality
 * 37 (3 %: 12), MAPQ 60 (3 %: 0), 1 % of left reads soft-clipped, 0.5 % with a 1-3 bp indel; small insertion /
 * deletion DNMs put an I / D operation behind the anchor base (the first such DNM a read covers).
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
#define UZS_ROW 160 /* bytes per seq / qual row of the ASCII form (16-byte aligned) */
#define UZS_UNITS 5 /* row units (32 bases) of a record in the packed form */
#define UZS_MAXOPS 3
#define UZS_MAXSEG 16384 /* records per cluster (sort key: 14 bits of slot, 17 bits of relative start) */

typedef struct uzs_cfg {
    uint64_t seed;
    int32_t n_clusters;
    int32_t min_base_qual; /* threshold of the quality plane of the packed form */
} uzs_cfg;

typedef struct uzs_sites { /* the columns of the sites table the generator needs */
    const int64_t *contig_off;
    const int32_t *pos;
    const uint8_t *ref_base, *alt_base;
    const uint8_t *khap; /* bit0 allele on the kid's paternal haplotype, bit1 maternal */
    int32_t n_contigs;
} uzs_sites;

typedef struct uzs_dnms { /* sorted by (contig, pos) */
    const int32_t *pos;
    const int32_t *site_idx; /* the DNM's own record in the sites table */
    const uint8_t *kind;     /* 0 SNV, 1 insertion, 2 deletion */
    const uint8_t *len;      /* inserted / deleted bases */
    const uint8_t *origin;   /* 0 paternal haplotype carries the DNM, 1 maternal */
} uzs_dnms;

typedef struct uzs_clusters {
    const int32_t *contig;
    const int32_t *lo, *hi;    /* reads are laid out within [lo, hi) */
    const int32_t *d0, *nd;    /* DNMs [d0, d0 + nd) lie in the cluster */
    const int64_t *pair_off;   /* [n+1] first pair of the cluster (records: 2 * pair) */
    const int64_t *cigar_off;  /* [n+1] first CIGAR word of the cluster (filled after the counting pass) */
} uzs_clusters;

typedef struct uzs_out_packed { /* uz_reads_packed_view columns */
    int32_t *start, *end, *tlen, *mate;
    uint32_t *qname;
    uint16_t *flag, *l_seq, *n_cigar;
    uint8_t *mapq, *aux;
    uint32_t *cigar; /* back to back */
    uint8_t *seq2;   /* UZS_UNITS * 8 bytes per record: two-bit base rows (the generator writes A/C/G/T only: no listed bases) */
    uint8_t *qlow;   /* UZS_UNITS * 4 bytes per record */
} uzs_out_packed;

typedef struct uzs_out_ascii { /* uz_reads_view columns; record / pair / CIGAR numbering relative to the first cluster generated */
    int32_t *start, *end;
    uint16_t *flag;
    uint8_t *mapq, *aux;
    int32_t *tlen;
    uint32_t *qname;
    int32_t *mate;
    uint32_t *cigar_off;
    uint16_t *n_cigar;
    uint32_t *cigar; /* back to back */
    uint16_t *l_seq;
    uint32_t *sq_off16;
    uint8_t *seq, *qual; /* UZS_ROW per record */
} uzs_out_ascii;

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
    uint64_t bseed; /* per-base hash stream of this record */
} uzs_seg;

/* geometry + CIGAR of segment `which` (0 left / 1 right) of pair k of cluster c */
UZS_HD void uzs_segment(const uzs_cfg *cf, const uzs_clusters *C, const uzs_dnms *D, int32_t c, int32_t k, int which, uzs_seg *s) {
    const int L = UZS_READLEN;
    const int64_t lo = C->lo[c], hi = C->hi[c];
    const int64_t np = C->pair_off[c + 1] - C->pair_off[c];
    const uint64_t h = uzs_h(cf->seed, (uint64_t)c, (uint64_t)k, 1);
    const int hap = k & 1;
    const int64_t per_hap = np / 2 > 0 ? np / 2 : 1;
    const int64_t kk = k >> 1;
    /* fragment starts stratified over [lo, hi - 900) per haplotype */
    const int64_t span = (hi - lo) - 900;
    const int64_t cell = span / per_hap > 0 ? span / per_hap : 1;
    const int64_t fs = lo + kk * span / per_hap + (int64_t)((h >> 8) % (uint64_t)cell);
    int ins = 450;
    for (int j = 0; j < 4; j++) ins += (int)((h >> (20 + 8 * j)) & 0xFF) * 100 / 256 - 50;
    if (ins < 2 * L) ins = 2 * L;
    if (ins > 900) ins = 900;
    const int64_t a = which == 0 ? fs : fs + ins - L;
    const uint64_t hv = uzs_h(cf->seed, (uint64_t)c, (uint64_t)k, 2 + (uint64_t)which);
    s->hap = (uint8_t)hap;
    s->bseed = uzs_h(cf->seed ^ 0x5bd1e995ULL, (uint64_t)c, (uint64_t)(2 * k + which), 7);
    s->flag = (uint16_t)(which == 0 ? (1 | 2 | 32 | 64) : (1 | 2 | 16 | 128));
    s->mapq = (uint8_t)(((hv >> 40) % 100) < 3 ? 0 : 60);
    s->tlen = which == 0 ? ins : -ins;
    /* the first insertion / deletion DNM of this haplotype whose anchor base the read covers */
    int dn = -1;
    for (int j = 0; j < C->nd[c]; j++) {
        const int d = C->d0[c] + j;
        const int64_t p = D->pos[d];
        if (D->kind[d] != 0 && (int)D->origin[d] == hap && a <= p && p < a + L - 12 && p - a >= 1) { dn = d; break; }
    }
    int64_t start = a;
    int n = 0;
    if (dn >= 0) {
        const int x = (int)(D->pos[dn] - a) + 1, dl = D->len[dn];
        if (D->kind[dn] == 1) { s->ops[0] = ((uint32_t)x << 4) | 0; s->ops[1] = ((uint32_t)dl << 4) | 1; s->ops[2] = ((uint32_t)(L - x - dl) << 4) | 0; }
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

/* site index range a read of cluster c can touch */
UZS_HD void uzs_site_window(const uzs_sites *S, const uzs_clusters *C, int32_t c, int64_t *s_lo, int64_t *s_hi) {
    const int32_t contig = C->contig[c];
    const int64_t lo = S->contig_off[contig], hi = S->contig_off[contig + 1];
    *s_lo = uzs_lower_bound(S->pos, lo, hi, (int64_t)C->lo[c] - 64);
    *s_hi = uzs_lower_bound(S->pos, lo, hi, (int64_t)C->hi[c] + 1200);
}

/* bases and qualities of query indices [i0, i1) of a segment (out[0] is index i0) */
UZS_HD void uzs_fill(const uzs_sites *S, const uzs_clusters *C, const uzs_dnms *D, int32_t c, const uzs_seg *s, int64_t s_lo, int64_t s_hi,
                     int i0, int i1, uint8_t *seq, uint8_t *qual) {
    const int32_t contig = C->contig[c];
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
                int dn = -1; /* is this site a DNM of the cluster? */
                for (int t = 0; t < C->nd[c]; t++)
                    if ((int64_t)D->site_idx[C->d0[c] + t] == si) { dn = C->d0[c] + t; break; }
                if (dn >= 0) {
                    b = S->ref_base[si];
                    if ((int)D->origin[dn] == (int)s->hap && D->kind[dn] == 0) b = S->alt_base[si];
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

/* sort key of a segment inside its cluster block: (start - block origin) then slot = 2k + which */
UZS_HD uint32_t uzs_key(const uzs_clusters *C, int32_t c, int32_t slot, const uzs_seg *s) {
    const int64_t rel = (int64_t)s->start - ((int64_t)C->lo[c] - 64);
    return ((uint32_t)rel << 14) | (uint32_t)slot;
}

UZS_HD int uzs_query_len(const uzs_seg *s) {
    int ql = 0;
    for (int j = 0; j < s->n_ops; j++) {
        const int op = s->ops[j] & 15;
        if (op == 0 || op == 1 || op == 4) ql += (int)(s->ops[j] >> 4);
    }
    return ql;
}

#endif /* UZSYNTH_H */
