/* uz_oracle.c -- CPU restatement of the reference's per-DNM phasing path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file.  The product
 * (unfazed_amd + libunfazed_hip.so) never does.
 *
 * It restates, sequentially and in the reference's own iteration order, the
 * logic of /root/reference/unfazed/{informative_site_finder,read_collector,
 * site_searcher,snv_phaser,sv_phaser,unfazed}.py over the decoded column
 * arrays of include/uz_types.h.  Every function cites the lines it follows.
 * The third-party pieces the reference leans on (cyvcf2 0.31.0, pysam 0.22.1:
 * tabix/BAI overlap queries, mate(), get_reference_positions) are not in
 * /root/reference; their published behaviour is restated in SURVEY.md
 * Appendix B and in unfazed_amd/model.py.
 *
 * Parity pinning: checked against outputs of the reference itself, imported in
 * the authoring container through tests/refshim stand-ins for cyvcf2/pysam
 * (tests/golden/make_golden.py generates the committed vectors; tests/
 * test_oracle_golden.py replays them).  The reference's own shell tests need
 * two input files that are missing from the snapshot (.MISSING_LARGE_BLOBS), so
 * their expected lines cannot be replayed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "uz_types.h"

#define OP_M 0
#define OP_I 1
#define OP_D 2
#define OP_N 3
#define OP_S 4
#define OP_H 5
#define OP_P 6
#define OP_EQ 7
#define OP_X 8

/* ---------------------------------------------------------------- sites */

static inline int dec16(uint16_t v) { return v == UZ_U16_MISSING ? -1 : (int)v; }

/* is_high_quality_site: informative_site_finder.py:46-73 */
static int hq_site(const uz_params *P, int gt, int ref, int alt, int gq) {
    double lo, hi;
    if (gt == UZ_HOM_REF) { lo = P->ab_homref[0]; hi = P->ab_homref[1]; }
    else if (gt == UZ_HOM_ALT) { lo = P->ab_homalt[0]; hi = P->ab_homalt[1]; }
    else if (gt == UZ_HET) { lo = P->ab_het[0]; hi = P->ab_het[1]; }
    else return 0;                               /* :62-63 */
    if (gq < P->min_gt_qual) return 0;           /* :64 */
    if (ref + alt < P->min_depth) return 0;      /* :66 */
    double ab = (double)alt / (double)(ref + alt); /* :69 numpy int32 / float -> float64 */
    return (lo <= ab) && (ab <= hi);             /* :71 */
}

/* get_kid_allele: informative_site_finder.py:76-134; vt = UZ_VT_DEL / UZ_VT_DUP */
static int kid_allele(const uz_params *P, int vt, const int gt[3], const int rd[3], const int ad[3]) {
    if (vt == UZ_VT_DEL && (rd[0] + ad[0]) > 4) {          /* :80 */
        if (gt[0] == UZ_HOM_ALT) return UZ_KA_REF_PARENT;  /* :82-83 */
        if (gt[0] == UZ_HOM_REF) return UZ_KA_ALT_PARENT;  /* :84-85 */
        return UZ_KA_NONE;                                 /* :86-88 */
    } else if (vt == UZ_VT_DUP && rd[0] > 2 && ad[0] > 2 && (rd[0] + ad[0]) > P->min_depth) { /* :89-94 */
        if (gt[0] != UZ_HET) return UZ_KA_NONE;            /* :128-130 */
        double k = (double)ad[0] / (double)(rd[0] + ad[0]);
        double d = (double)ad[1] / (double)(rd[1] + ad[1]);
        double m = (double)ad[2] / (double)(rd[2] + ad[2]);
        if ((((d + m) < 1) && (k > 0.5)) || (((d + m) > 1) && (k < 0.5))) return UZ_KA_NONE; /* :110-116 */
        if (k >= 0.67) return UZ_KA_ALT_PARENT;            /* :119-121 */
        if (k <= 0.33) return UZ_KA_REF_PARENT;            /* :122-124 */
        return UZ_KA_NONE;
    }
    return UZ_KA_NONE;                                     /* :131-133 */
}

/* The DNM-independent part of find()'s per-variant body (:239-339, duplicated at
 * :442-543) for one site of one family -> class byte (uz_types.h UZ_CL_*). */
uint8_t uzo_classify_one(const uz_params *P, uint8_t sflags, uint8_t gtp, const int rd[3], const int ad[3],
                         const int gq[3]) {
    if (sflags & UZ_SF_COMPLEX) return 0; /* :239-244 */
    int gt[3] = {gtp & 3, (gtp >> 2) & 3, (gtp >> 4) & 3};
    int hqk = hq_site(P, gt[0], rd[0], ad[0], gq[0]);
    int hqd = hq_site(P, gt[1], rd[1], ad[1], gq[1]);
    int hqm = hq_site(P, gt[2], rd[2], ad[2], gq[2]);
    uint8_t c = 0;
    if (gt[0] == UZ_HET && hqd && hqm) c |= UZ_CL_HET; /* :268-284 */
    /* parental pattern :307-320 */
    int pattern = 0, alt_dad = 0;
    int dad = gt[1], mom = gt[2];
    if ((dad == UZ_HET || dad == UZ_HOM_ALT) && mom == UZ_HOM_REF) { pattern = 1; alt_dad = 1; }
    else if ((mom == UZ_HET || mom == UZ_HOM_ALT) && dad == UZ_HOM_REF) { pattern = 1; alt_dad = 0; }
    else if (mom == UZ_HET && dad == UZ_HOM_ALT) { pattern = 1; alt_dad = 1; }
    else if (dad == UZ_HET && mom == UZ_HOM_ALT) { pattern = 1; alt_dad = 0; }
    if (pattern && alt_dad) c |= UZ_CL_ALT_DAD;
    if (!(hqd && hqm && pattern)) return c; /* :297-305, :319-320 */
    /* SNV / breakpoint mode :292-295 */
    if (gt[0] == UZ_HET && hqk) c |= UZ_CL_CAND;
    /* CNV mode :286-291 then the hemizygous unique-allele check :324-337 */
    int unique = 1;
    if (gt[0] == UZ_HOM_ALT || gt[0] == UZ_HOM_REF) {
        int het_in = (dad == UZ_HET) || (mom == UZ_HET);
        int hom_in = (dad == UZ_HOM_ALT) || (mom == UZ_HOM_ALT) || (dad == UZ_HOM_REF) || (mom == UZ_HOM_REF);
        if (het_in && hom_in) {
            if ((dad == UZ_HOM_ALT || dad == UZ_HOM_REF) && gt[0] == dad) unique = 0;
            if ((mom == UZ_HOM_ALT || mom == UZ_HOM_REF) && gt[0] == mom) unique = 0;
        }
    }
    if (unique) {
        int kdel = kid_allele(P, UZ_VT_DEL, gt, rd, ad);
        int kdup = kid_allele(P, UZ_VT_DUP, gt, rd, ad);
        c |= (uint8_t)(kdel << UZ_CL_DEL_SHIFT);
        c |= (uint8_t)(kdup << UZ_CL_DUP_SHIFT);
    }
    return c;
}

/* depths and GQ of site i: from the 16-bit columns, or -- a site listed as too deep for them (uz_family_view.wide_*) -- the
 * 32-bit depths of the list */
static void fam_values(const uz_family_view *F, int64_t i, int rd[3], int ad[3], int gq[3]) {
    for (int m = 0; m < 3; m++) {
        rd[m] = dec16(F->ref_depth[m][i]);
        ad[m] = dec16(F->alt_depth[m][i]);
        gq[m] = dec16(F->gq[m][i]);
    }
    if (F->n_wide > 0) {
        int64_t lo = 0, hi = F->n_wide;
        while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if (F->wide_site[mid] < i) lo = mid + 1; else hi = mid; }
        if (lo < F->n_wide && F->wide_site[lo] == i)
            for (int m = 0; m < 3; m++) { rd[m] = F->wide_ref_depth[m][lo]; ad[m] = F->wide_alt_depth[m][lo]; }
    }
}

void uzo_classify(const uz_params *P, const uz_sites_view *S, const uz_family_view *F, int64_t lo, int64_t hi,
                  uint8_t *cls) {
    for (int64_t i = lo; i < hi; i++) {
        int rd[3], ad[3], gq[3];
        fam_values(F, i, rd, ad, gq);
        cls[i] = uzo_classify_one(P, S->sflags[i], F->gt[i], rd, ad, gq);
    }
}

static int64_t lower_bound_i32(const int32_t *a, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* Per-DNM window scan: get_position (:10-43) + the per-variant body of find (:237-343).
 * mode bit0 = whole_region, bit1 = second window allowed (find; find_many has none).
 * Two-call sizing: pass cand_idx == NULL to only fill the offsets. */
int uzo_find(const uz_params *P, const uz_sites_view *S, const uz_family_view *F, const uz_dnms_view *D, int mode,
             int64_t *cand_off, int32_t *cand_idx, uint8_t *cand_flags, int64_t *het_off, int32_t *het_idx) {
    int whole = mode & 1, second = (mode & 2) != 0;
    int64_t nc = 0, nh = 0;
    int64_t sd = P->search_dist;
    for (int32_t d = 0; d < D->n; d++) {
        cand_off[d] = nc;
        het_off[d] = nh;
        int32_t c = D->contig[d];
        if (c < 0 || c >= S->n_contigs) continue;
        int64_t clo = S->contig_off[c], chi = S->contig_off[c + 1];
        int64_t st = D->start[d], en = D->end[d];
        /* 1-based inclusive POS windows (:14-40); region start clamped to 1 (unpinned) */
        int64_t w[2][2];
        int nw = 1;
        if (whole) { w[0][0] = st - sd; w[0][1] = en + sd; }
        else {
            w[0][0] = st - sd; w[0][1] = st + sd;
            if (second && (en - st) > sd) { w[1][0] = en - sd; w[1][1] = en + sd; nw = 2; }
        }
        int vt = D->vartype[d];
        int mult = D->mult ? D->mult[d] : 1;
        /* Python's sorted() is stable: the concatenation of the windows sorted by pos equals, for
         * windows given in ascending order, emitting by position with window-1 copies first. */
        int64_t lo0 = lower_bound_i32(S->pos, clo, chi, (w[0][0] < 1 ? 1 : w[0][0]) - 1);
        int64_t hi_all = lower_bound_i32(S->pos, clo, chi, w[nw - 1][1]); /* pos0 <= b-1 */
        int64_t i = lo0;
        while (i < hi_all) {
            int64_t j = i; /* run of equal positions */
            while (j < hi_all && S->pos[j] == S->pos[i]) j++;
            int64_t pos1 = (int64_t)S->pos[i] + 1;
            for (int k = 0; k < nw; k++) {
                int64_t a = w[k][0] < 1 ? 1 : w[k][0], b = w[k][1];
                if (pos1 < a || pos1 > b) continue;
                for (int64_t s = i; s < j; s++) {
                    uint8_t cl;
                    {
                        int rd[3], ad[3], gq[3];
                        fam_values(F, s, rd, ad, gq);
                        cl = uzo_classify_one(P, S->sflags[s], F->gt[s], rd, ad, gq);
                    }
                    if (!cl) continue;
                    /* small-event exclusion :253-256 */
                    if ((en - st) < 20 && S->pos[s] >= st && S->pos[s] < en) continue;
                    for (int r = 0; r < mult; r++) {
                        if (cl & UZ_CL_HET) {
                            if (het_idx) het_idx[nh] = (int32_t)s;
                            nh++;
                        }
                        int ka = 0, is_c = 0;
                        if (whole) { /* :286-291: vartype is always present in the DNM dict */
                            if (vt == UZ_VT_DEL) ka = (cl >> UZ_CL_DEL_SHIFT) & 3;
                            else if (vt == UZ_VT_DUP) ka = (cl >> UZ_CL_DUP_SHIFT) & 3;
                            is_c = ka != 0;
                        } else is_c = (cl & UZ_CL_CAND) != 0;
                        if (is_c) {
                            if (cand_idx) {
                                cand_idx[nc] = (int32_t)s;
                                cand_flags[nc] = (uint8_t)(((cl & UZ_CL_ALT_DAD) ? UZ_CF_ALT_DAD : 0) | (ka << UZ_CF_KA_SHIFT));
                            }
                            nc++;
                        }
                    }
                }
            }
            i = j;
        }
    }
    cand_off[D->n] = nc;
    het_off[D->n] = nh;
    return 0;
}

/* ---------------------------------------------------------------- reads */

typedef struct {
    int32_t *v;
    int64_t n, cap;
} ivec;

static void iv_push(ivec *a, int32_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (int32_t *)realloc(a->v, (size_t)a->cap * sizeof(int32_t));
    }
    a->v[a->n++] = x;
}

typedef struct {
    const uz_params *P;
    const uz_reads_view *R;
    uint8_t *good_cache; /* per segment: 0 unknown, 1 good, 2 bad (non-discordant goodread) */
} rctx;

/* goodread(read, discordant=False): read_collector.py:28-53 */
static int goodread(rctx *X, int32_t i) {
    uint8_t g = X->good_cache[i];
    if (g) return g == 1;
    const uz_reads_view *R = X->R;
    int ok = 1;
    uint16_t f = R->flag[i];
    if (R->aux[i] & UZ_AUX_DECODE_BAD) ok = 0; /* no CIGAR/SEQ/QUAL: the reference raises TypeError; unpinned, treated as not good */
    else if ((f & 512) || (f & 4) || (f & 1024) || (int)R->mapq[i] < X->P->min_map_qual || (f & 256) || (f & 2048) ||
             (f & 8) || !(R->aux[i] & UZ_AUX_MATE_SAME_TID))
        ok = 0; /* :31-41 */
    else {
        const uint8_t *q = R->qual + ((int64_t)R->sq_off16[i] << 4);
        int low = 0;
        for (int k = 0; k < R->l_seq[i]; k++)
            if ((int)q[k] < X->P->min_gt_qual) low++; /* :43-46 */
        int mism = R->n_cigar[i]; /* :47-50 CIGAR_MAP[True] is truthy for every op (quirk Q9) */
        if (low > 10 || mism > 10) ok = 0;
    }
    X->good_cache[i] = ok ? 1 : 2;
    return ok;
}

/* goodread(read, discordant=True): read_collector.py:28-42 (flag / mapq / mate checks only) */
static int goodread_disc(rctx *X, int32_t i) {
    const uz_reads_view *R = X->R;
    uint16_t f = R->flag[i];
    if (R->aux[i] & UZ_AUX_DECODE_BAD) return 0;
    if ((f & 512) || (f & 4) || (f & 1024) || (int)R->mapq[i] < X->P->min_map_qual || (f & 256) || (f & 2048) ||
        (f & 8) || !(R->aux[i] & UZ_AUX_MATE_SAME_TID))
        return 0;
    return 1;
}

/* index of `pos` in get_reference_positions(full_length=True), -1 if absent */
static int qidx(const uz_reads_view *R, int32_t i, int64_t pos) {
    const uint32_t *c = R->cigar + R->cigar_off[i];
    int64_t r = R->start[i];
    int q = 0;
    for (int k = 0; k < R->n_cigar[i]; k++) {
        int op = c[k] & 15, l = (int)(c[k] >> 4);
        if (op == OP_M || op == OP_EQ || op == OP_X) {
            if (pos >= r && pos < r + l) return q + (int)(pos - r);
            q += l; r += l;
        } else if (op == OP_I || op == OP_S) q += l;
        else if (op == OP_D || op == OP_N) r += l;
    }
    return -1;
}
static int refpos_len(const uz_reads_view *R, int32_t i) {
    const uint32_t *c = R->cigar + R->cigar_off[i];
    int q = 0;
    for (int k = 0; k < R->n_cigar[i]; k++) {
        int op = c[k] & 15, l = (int)(c[k] >> 4);
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_S) q += l;
    }
    return q;
}
static int none_count(const uz_reads_view *R, int32_t i) {
    const uint32_t *c = R->cigar + R->cigar_off[i];
    int q = 0;
    for (int k = 0; k < R->n_cigar[i]; k++) {
        int op = c[k] & 15, l = (int)(c[k] >> 4);
        if (op == OP_I || op == OP_S) q += l;
    }
    return q;
}
static int nonmatch_ops(const uz_reads_view *R, int32_t i) {
    const uint32_t *c = R->cigar + R->cigar_off[i];
    int q = 0;
    for (int k = 0; k < R->n_cigar[i]; k++) {
        int op = c[k] & 15;
        if (op != OP_M && op != OP_EQ) q++;
    }
    return q;
}
static inline const uint8_t *seqp(const uz_reads_view *R, int32_t i) { return R->seq + ((int64_t)R->sq_off16[i] << 4); }
static inline const uint8_t *qualp(const uz_reads_view *R, int32_t i) { return R->qual + ((int64_t)R->sq_off16[i] << 4); }

/* get_allele_at: read_collector.py:56-73.  Returns pointer to n bases or NULL (False). */
static const uint8_t *allele_at(rctx *X, int32_t read, int32_t mate, int64_t pos, int n) {
    const uz_reads_view *R = X->R;
    int i = qidx(R, read, pos);
    if (i >= 0) {
        if (i < 4 || i > X->P->readlen - 4) return NULL; /* :63-64 */
        if ((int)R->l_seq[read] > i + n) return seqp(R, read) + i; /* :65-66 */
        return NULL; /* falls through to :73, the mate is not consulted (Q10) */
    } else if (mate >= 0) {
        int j = qidx(R, mate, pos);
        if (j >= 0) {
            if (j < 4 || j > X->P->readlen - 4) return NULL;
            if ((int)R->l_seq[mate] > j + n) return seqp(R, mate) + j;
        }
    }
    return NULL;
}

/* pysam fetch(contig, lo, hi): records with start < hi and endpos > lo, file order.
 * Returns the index range to scan; the caller tests `end > lo`. */
static void fetch_range(const uz_reads_view *R, int32_t tid, int64_t lo, int64_t hi, int64_t *a, int64_t *b) {
    if (tid < 0 || tid >= R->n_contigs) { *a = *b = 0; return; }
    int64_t clo = R->contig_off[tid], chi = R->contig_off[tid + 1];
    *a = lower_bound_i32(R->start, clo, chi, lo - R->max_span[tid]);
    *b = lower_bound_i32(R->start, clo, chi, hi);
}

/* binary_search: site_searcher.py:6-47 over a position list.  Returns count, fills out[]. */
static int bsearch_sites(int64_t start, int64_t end, const int32_t *pos, int n, int32_t *out, int outcap) {
    int nm = 0;
    int qs = 0, qe = n - 1, qsp = -1, qep = -1;
    while (nm <= 0 && qe > -1) {
        if (qs > qe) break;
        if (qs == qsp && qe == qep) break;
        qsp = qs; qep = qe;
        int qp = (qe + qs) / 2;
        if (start <= pos[qp] && pos[qp] < end) {
            if (nm < outcap) out[nm] = qp;
            nm++;
            for (int k = qp + 1; k < n; k++) {
                if (start <= pos[k] && pos[k] <= end) { if (nm < outcap) out[nm] = k; nm++; }
                else break;
            }
            for (int k = qp - 1; k >= 0; k--) {
                if (start <= pos[k] && pos[k] <= end) { if (nm < outcap) out[nm] = k; nm++; }
                else break;
            }
            break;
        } else if (pos[qp] > start) qe = qp - 1;
        else if (pos[qp] < start) qs = qp + 1;
    }
    return nm;
}

/* collect_reads_sv: read_collector.py:476-596, up to the hand-off to group_reads_by_haplotype.
 * Appends the supporting segments (in order) to `alt`. */
static void collect_sv(rctx *X, const uz_params *P, int32_t tid, int64_t start, int64_t end, double cutoff, ivec *alt) {
    const uz_reads_view *R = X->R;
    ivec supporting = {0}, banned = {0};
    const double var_len = fabs((double)end - (double)start); /* :477 */
    const int64_t icut = (int64_t)cutoff;
    const int64_t bp[2] = {start, end};
    for (int w = 0; w < 2; w++) { /* :478 */
        const int64_t position = bp[w];
        int64_t lo = position - icut, hi = position + icut; /* :481-485 */
        if (lo < 0) lo = 0;
        banned.n = 0; /* :498 a fresh list per breakpoint */
        int64_t a, b;
        fetch_range(R, tid, lo, hi, &a, &b);
        for (int64_t i = a; i < b; i++) {
            if (!(R->end[i] > lo)) continue;
            const int32_t read = (int32_t)i;
            const int32_t q = (int32_t)R->qname[read];
            int is_banned = 0;
            for (int64_t k = 0; k < banned.n; k++) if (banned.v[k] == q) { is_banned = 1; break; }
            if (is_banned) continue;                 /* :501-502 */
            if (!goodread_disc(X, read)) continue;   /* :503-504 */
            const int32_t mate = R->mate[read];      /* :507-510 */
            if (mate < 0) continue;
            const int64_t insert = llabs((int64_t)R->tlen[read] - 2 * (int64_t)P->readlen); /* :511 */
            if (!goodread_disc(X, mate)) continue;   /* :512-513 */
            /* :515-522 M/= among the first / last 10 entries of the per-base expansion of ALL ops */
            const uint32_t *c = R->cigar + R->cigar_off[read];
            const int nc = R->n_cigar[read];
            int64_t total = 0;
            for (int k = 0; k < nc; k++) total += (int64_t)(c[k] >> 4);
            int start_m = 0, end_m = 0;
            {
                int64_t o = 0;
                for (int k = 0; k < nc; k++) {
                    const int op = c[k] & 15;
                    const int64_t l = (int64_t)(c[k] >> 4);
                    const int64_t s0 = o, s1 = o + l; /* entries [s0, s1) */
                    if (op == OP_M || op == OP_EQ) {
                        int64_t x1 = s1 < 10 ? s1 : 10;
                        if (x1 > s0) start_m += (int)(x1 - s0);
                        int64_t t0 = total - 10 > 0 ? total - 10 : 0;
                        int64_t y0 = s0 > t0 ? s0 : t0;
                        if (s1 > y0) end_m += (int)(s1 - y0);
                    }
                    o = s1;
                }
            }
            if (end_m < 7 && start_m < 7) { iv_push(&banned, q); continue; } /* :520-522 */
            const int64_t rs = R->start[read], re = R->end[read];
            if (R->aux[read] & UZ_AUX_HAS_SA) { /* :524-533 split read clipped near the breakpoint */
                const int64_t m = P->split_error_margin;
                if ((position - m <= rs && rs <= position + m) || (position - m <= re && re <= position + m)) {
                    iv_push(&supporting, read); iv_push(&supporting, mate);
                }
            } else if ((double)insert > cutoff && 0.7 < fabs(var_len / (double)insert) && fabs(var_len / (double)insert) < 1.3) { /* :534-536 */
                const int32_t mate2 = R->mate[read]; /* :539-542 */
                if (mate2 < 0) continue;
                const int64_t ms = R->start[mate2];
                const int64_t left0 = ms < rs ? ms : rs, right0 = ms > rs ? ms : rs; /* :543-550 */
                const int64_t wig = (int64_t)cutoff;                                 /* :551 */
                if (!((start - wig) < left0 && left0 < (start + wig) && (end - wig) < right0 && right0 < (end + wig))) continue;
                iv_push(&supporting, mate2); iv_push(&supporting, read);             /* :562-563 */
            } else { /* :564-586 clipped reads that are not split alignments */
                int rp = qidx(R, read, position);
                if (rp < 0) rp = qidx(R, read, position - 1);
                if (rp < 0) rp = qidx(R, read, position + 1);
                if (rp < 0) continue;
                const int len = refpos_len(R, read);
                if (rp < 2 || rp > len - 4) continue; /* :574-575 */
                /* before = set(refpos[:rp-1]) == {None}: the first rp-1 query bases have no reference position */
                int lead = 0, trail = 0;
                for (int k = 0; k < nc; k++) {
                    const int op = c[k] & 15;
                    if (op == OP_S || op == OP_I) lead += (int)(c[k] >> 4);
                    else if (op == OP_M || op == OP_EQ || op == OP_X) break;
                }
                for (int k = nc - 1; k >= 0; k--) {
                    const int op = c[k] & 15;
                    if (op == OP_S || op == OP_I) trail += (int)(c[k] >> 4);
                    else if (op == OP_M || op == OP_EQ || op == OP_X) break;
                }
                if (lead >= rp - 1 || trail >= len - (rp + 1)) { iv_push(&supporting, mate); iv_push(&supporting, read); } /* :579-586 */
            }
        }
    }
    /* :588-596 -- `banned_reads` is the list of the LAST breakpoint only */
    ivec filtered = {0};
    for (int64_t k = 0; k < supporting.n; k++) {
        const int32_t q = (int32_t)R->qname[supporting.v[k]];
        int is_banned = 0;
        for (int64_t j = 0; j < banned.n; j++) if (banned.v[j] == q) { is_banned = 1; break; }
        if (!is_banned) iv_push(&filtered, supporting.v[k]);
    }
    if (filtered.n >= 2) /* :594-595 */
        for (int64_t k = 0; k < filtered.n; k++) iv_push(alt, filtered.v[k]);
    free(supporting.v); free(banned.v); free(filtered.v);
}

/* exported for the golden tests of binary_search */
int uzo_bsearch(int64_t start, int64_t end, const int32_t *pos, int n, int32_t *out) {
    return bsearch_sites(start, end, pos, n, out, n);
}

typedef struct uzo_result {
    int32_t n;
    int32_t *status;   /* [n] UZ_ST_* */
    int64_t *init_off; /* [2n+1]: alt list, ref list of collect_reads_snv before extension (segment ids, in order) */
    int32_t *init_seg;
    int64_t *grp_off;  /* [2n+1]: ref set, alt set after connect_reads (qname ids, ascending); empty when no_extended */
    int32_t *grp_q;
    int64_t *vote_off; /* [4n+1]: dad_reads (qname ids asc), mom_reads, dad_sites (positions asc), mom_sites */
    int32_t *vote_val;
    int32_t *counts;   /* [4n] dad_reads, mom_reads, dad_sites, mom_sites */
    int32_t *origin;   /* [n] UZ_OR_* */
    int32_t *evidence; /* [n] evidence_count of summarize_record's read-backed branch */
} uzo_result;

void uzo_result_free(uzo_result *r) {
    if (!r) return;
    free(r->status); free(r->init_off); free(r->init_seg); free(r->grp_off); free(r->grp_q);
    free(r->vote_off); free(r->vote_val); free(r->counts); free(r->origin); free(r->evidence);
    free(r);
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}
static int64_t sort_unique(int32_t *v, int64_t n) {
    if (n <= 1) return n;
    qsort(v, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++)
        if (v[i] != v[m - 1]) v[m++] = v[i];
    return m;
}

/* per-qname scratch (dict emulation), reset through the touched list */
typedef struct {
    ivec *read_sites;   /* per qname: het-list indices in append order (read_sites[qname]) */
    uint8_t *has_rs;    /* qname in read_sites */
    int32_t *fet0, *fet1; /* fetched_reads[qname] = [read, mate]; fet0 = -1: absent */
    uint8_t *grp;       /* bit0 in grouped["ref"], bit1 in grouped["alt"] */
    ivec touched;
    uint8_t *is_touched;
} qscratch;

static void touch(qscratch *Q, uint32_t q) {
    if (!Q->is_touched[q]) {
        Q->is_touched[q] = 1;
        Q->fet0[q] = Q->fet1[q] = -1; /* lazily initialised: a call only pays for the names it touches */
        iv_push(&Q->touched, (int32_t)q);
    }
}

typedef struct { int32_t q; int64_t found_pos; } nr_item; /* [readname, found_pos] */
typedef struct { nr_item *v; int64_t n, cap; } nrvec;
static void nr_push(nrvec *a, int32_t q, int64_t fp) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 64; a->v = (nr_item *)realloc(a->v, (size_t)a->cap * sizeof(nr_item)); }
    a->v[a->n].q = q; a->v[a->n].found_pos = fp; a->n++;
}

/* The per-DNM read stage: multithread_read_phasing (snv_phaser.py:87-203) from
 * collect_reads_snv on; candidate / het lists come from uzo_find. */
int uzo_phase(const uz_params *P, const uz_sites_view *S, const uz_reads_view *R, const uz_dnms_view *D,
              const int64_t *cand_off, const int32_t *cand_idx, const uint8_t *cand_flags, const int64_t *het_off,
              const int32_t *het_idx, int32_t d_lo, int32_t d_hi, int keep_lists, uzo_result **out) {
    int32_t n = D->n;
    uzo_result *res = (uzo_result *)calloc(1, sizeof(uzo_result));
    res->n = n;
    res->status = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    res->init_off = (int64_t *)calloc((size_t)2 * n + 1, sizeof(int64_t));
    res->grp_off = (int64_t *)calloc((size_t)2 * n + 1, sizeof(int64_t));
    res->vote_off = (int64_t *)calloc((size_t)4 * n + 1, sizeof(int64_t));
    res->counts = (int32_t *)calloc((size_t)4 * n, sizeof(int32_t));
    res->origin = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    res->evidence = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    ivec init_all = {0}, grp_all = {0}, vote_all = {0};

    rctx X;
    X.P = P; X.R = R;
    X.good_cache = (uint8_t *)calloc((size_t)(R->n_segs > 0 ? R->n_segs : 1), 1);
    uint32_t nq = R->n_qnames ? R->n_qnames : 1;
    qscratch Q;
    memset(&Q, 0, sizeof(Q));
    Q.read_sites = (ivec *)calloc(nq, sizeof(ivec));
    Q.has_rs = (uint8_t *)calloc(nq, 1);
    Q.fet0 = (int32_t *)malloc(nq * sizeof(int32_t));
    Q.fet1 = (int32_t *)malloc(nq * sizeof(int32_t));
    Q.grp = (uint8_t *)calloc(nq, 1);
    Q.is_touched = (uint8_t *)calloc(nq, 1);
    ivec all_touched = {0};
    uint8_t *ever = (uint8_t *)calloc(nq, 1);

    for (int32_t d = 0; d < n; d++) {
        res->init_off[2 * d] = res->init_off[2 * d + 1] = init_all.n;
        res->grp_off[2 * d] = res->grp_off[2 * d + 1] = grp_all.n;
        for (int k = 0; k < 4; k++) res->vote_off[4 * d + k] = vote_all.n;
        if (d < d_lo || d >= d_hi) { res->status[d] = UZ_ST_SKIPPED; continue; }
        int64_t nc = cand_off[d + 1] - cand_off[d], nh = het_off[d + 1] - het_off[d];
        if (nc <= 0) { res->status[d] = UZ_ST_NO_CAND; continue; } /* snv_phaser.py:254-262 */
        const int32_t *cidx = cand_idx + cand_off[d];
        const uint8_t *cfl = cand_flags + cand_off[d];
        const int32_t *hidx = het_idx + het_off[d];
        int32_t *hpos = (int32_t *)malloc((size_t)(nh + 1) * sizeof(int32_t));
        int32_t *hcanon = (int32_t *)malloc((size_t)(nh + 1) * sizeof(int32_t)); /* first het index with the same pos */
        for (int64_t k = 0; k < nh; k++) {
            hpos[k] = S->pos[hidx[k]];
            hcanon[k] = (k > 0 && hpos[k] == hpos[k - 1]) ? hcanon[k - 1] : (int32_t)k;
        }
        int32_t *cpos = (int32_t *)malloc((size_t)(nc + 1) * sizeof(int32_t));
        for (int64_t k = 0; k < nc; k++) cpos[k] = S->pos[cidx[k]];
        ivec *site_reads = (ivec *)calloc((size_t)(nh + 1), sizeof(ivec)); /* by canonical het index */
        uint8_t *sr_exists = (uint8_t *)calloc((size_t)(nh + 1), 1);

        int32_t tid = D->rcontig[d];
        int64_t position = D->start[d];
        const uint8_t *ref = D->alleles + D->allele_off[2 * d];
        int ref_len = (int)(D->allele_off[2 * d + 1] - D->allele_off[2 * d]);
        const uint8_t *alt = D->alleles + D->allele_off[2 * d + 1];
        int alt_len = (int)(D->allele_off[2 * d + 2] - D->allele_off[2 * d + 1]);
        double cutoff = D->cutoff;
        ivec lists[2];
        memset(lists, 0, sizeof(lists)); /* [0] = "alt", [1] = "ref" (informative_reads, read_collector.py:393) */

        const int is_sv = D->vartype[d] != UZ_VT_POINT; /* phase_svs -> collect_reads_sv (sv_phaser.py:111) */
        if (is_sv) collect_sv(&X, P, tid, D->start[d], D->end[d], cutoff, &lists[0]);
        /* ---- collect_reads_snv :382-425 ---- */
        int64_t fa, fb;
        int64_t flo = (D->dflags[d] & UZ_DF_FETCH_FALLBACK) ? position : position - 1; /* :385 / :392 */
        fetch_range(R, tid, flo, position + 1, &fa, &fb);
        if (is_sv) fb = fa;
        for (int64_t i = fa; i < fb; i++) {
            if (!(R->end[i] > flo)) continue;
            int32_t read = (int32_t)i;
            int64_t insert = llabs((int64_t)R->tlen[read] - 2 * (int64_t)P->readlen); /* :395 */
            if (!goodread(&X, read) || (double)insert > cutoff) continue;          /* :396 */
            int32_t mate = R->mate[read];                                          /* :400 */
            if (mate < 0) continue;                                                /* :403 ValueError */
            if (!goodread(&X, mate)) continue;                                     /* :401 */
            if (none_count(R, read) > 5 || none_count(R, mate) > 5) continue;      /* :405-408 */
            int64_t rs = R->start[read], re = R->end[read], ms = R->start[mate], me = R->end[mate];
            if ((ms <= rs && rs <= me) || (ms <= re && re <= me)) continue;        /* :411-418 */
            if (ref_len == alt_len) {
                /* snv_match_alleles :296-336 with equal lengths */
                const uint8_t *a = allele_at(&X, read, mate, position, ref_len > alt_len ? ref_len : alt_len);
                if (!a) continue;
                if (memcmp(a, ref, (size_t)ref_len) == 0) { iv_push(&lists[1], read); iv_push(&lists[1], mate); }
                else if (memcmp(a, alt, (size_t)alt_len) == 0) { iv_push(&lists[0], read); iv_push(&lists[0], mate); }
            } else {
                /* indel_match_alleles :266-293 */
                int var_len = ref_len > alt_len ? ref_len : alt_len;
                int rp = qidx(R, read, position);
                if (rp < 0) continue;
                /* operations: every CIGAR op repeated by its length (incl. D/N/H/P), indexed by the QUERY index (Q16) */
                const uint32_t *c = R->cigar + R->cigar_off[read];
                int has_id = 0, oi = 0;
                for (int k = 0; k < R->n_cigar[read] && oi < rp + var_len; k++) {
                    int op = c[k] & 15, l = (int)(c[k] >> 4);
                    int a0 = oi > rp ? oi : rp, a1 = (oi + l) < (rp + var_len) ? (oi + l) : (rp + var_len);
                    if (a0 < a1 && (op == OP_I || op == OP_D)) has_id = 1;
                    oi += l;
                }
                const uint8_t *ql = qualp(R, read);
                int lowq = 0;
                for (int k = rp; k < rp + var_len && k < (int)R->l_seq[read]; k++)
                    if ((int)ql[k] < P->min_gt_qual) lowq = 1; /* :281-284 */
                if (lowq) continue;
                if (has_id) { iv_push(&lists[0], read); iv_push(&lists[0], mate); }                       /* :286-289 */
                else if (7 < rp && rp < refpos_len(R, read) - 7) { iv_push(&lists[1], read); iv_push(&lists[1], mate); } /* :290-293 */
            }
        }
        if (keep_lists) {
            for (int64_t k = 0; k < lists[0].n; k++) iv_push(&init_all, lists[0].v[k]);
            res->init_off[2 * d + 1] = init_all.n;
            for (int64_t k = 0; k < lists[1].n; k++) iv_push(&init_all, lists[1].v[k]);
        }

        /* result lists handed to match_informative_sites: segments per haplotype */
        ivec hapsegs[2]; /* [0] = "ref", [1] = "alt" */
        memset(hapsegs, 0, sizeof(hapsegs));
        int exception = 0;

        if (P->no_extended) {
            for (int64_t k = 0; k < lists[1].n; k++) iv_push(&hapsegs[0], lists[1].v[k]);
            for (int64_t k = 0; k < lists[0].n; k++) iv_push(&hapsegs[1], lists[0].v[k]);
        } else {
            /* ---- group_reads_by_haplotype :155-263 ---- */
            for (int64_t h = 0; h < nh; h++) { /* :165 */
                int64_t a, b;
                fetch_range(R, tid, hpos[h], (int64_t)hpos[h] + 1, &a, &b);
                int64_t it = 0;
                for (int64_t i = a; i < b; i++) {
                    if (!(R->end[i] > hpos[h])) continue;
                    int64_t ei = it++;
                    if (ei > P->read_goal) continue; /* :179 */
                    int32_t read = (int32_t)i;
                    int64_t insert = llabs((int64_t)R->tlen[read] - 2 * (int64_t)P->readlen);
                    if (!(goodread(&X, read) && (double)insert <= cutoff)) continue; /* :183 */
                    int32_t mate = R->mate[read];
                    if (mate < 0) continue;              /* :186 */
                    if (!goodread(&X, mate)) continue;   /* :188 */
                    if (nonmatch_ops(R, read) > 5) continue; /* :190-196 */
                    if (none_count(R, read) > 5 || none_count(R, mate) > 5) continue; /* :198-203 */
                    int64_t rs = R->start[read], re = R->end[read], ms = R->start[mate], me = R->end[mate];
                    if ((ms <= rs && rs <= me) || (ms <= re && re <= me)) continue; /* :207-214 */
                    uint32_t q = R->qname[read];
                    touch(&Q, q);
                    Q.has_rs[q] = 1;                          /* :215-216 */
                    sr_exists[hcanon[h]] = 1;                 /* :217-218 */
                    iv_push(&Q.read_sites[q], (int32_t)h);    /* :220 */
                    iv_push(&site_reads[hcanon[h]], (int32_t)q); /* :221 */
                    Q.fet0[q] = read; Q.fet1[q] = mate;       /* :222 */
                }
            }
            nrvec cur[2]; /* [0] = "alt", [1] = "ref": new_reads = {"alt": [], "ref": []} :224 */
            memset(cur, 0, sizeof(cur));
            int32_t *msites = (int32_t *)malloc((size_t)(nh + 1) * sizeof(int32_t));
            for (int ra = 0; ra < 2; ra++) { /* :226 for refalt in ["ref", "alt"] */
                ivec *L = ra == 0 ? &lists[1] : &lists[0];
                int bit = ra == 0 ? 1 : 2;
                nrvec *NR = ra == 0 ? &cur[1] : &cur[0];
                for (int64_t k = 0; k < L->n; k++) {
                    int32_t read = L->v[k];
                    uint32_t q = R->qname[read];
                    touch(&Q, q);
                    Q.grp[q] |= (uint8_t)bit;  /* :230 */
                    nr_push(NR, (int32_t)q, -1); /* :231 */
                    int32_t mate = R->mate[read]; /* :233 */
                    if (mate < 0) continue;       /* :248 */
                    Q.fet0[q] = read; Q.fet1[q] = mate; /* :234 */
                    int nm = bsearch_sites(R->start[read], R->end[read], hpos, (int)nh, msites, (int)nh); /* :235 */
                    if (nm <= 0) continue;
                    Q.has_rs[q] = 1;                    /* :240-241 */
                    sr_exists[hcanon[nh - 1]] = 1;      /* :242-243 stale loop variable: last het site (Q13) */
                    for (int m = 0; m < nm; m++) {
                        iv_push(&Q.read_sites[q], msites[m]);              /* :246 */
                        iv_push(&site_reads[hcanon[nh - 1]], (int32_t)q);  /* :247 */
                    }
                }
            }
            free(msites);
            /* ---- connect_reads :76-152 (recursion unrolled) ---- */
            int level = 0;
            while (!exception) {
                nrvec add[2]; /* [0] = "ref", [1] = "alt": reads_to_add = {"ref": [], "alt": []} :78 */
                memset(add, 0, sizeof(add));
                for (int oi = 0; oi < 2 && !exception; oi++) { /* :79 dict order of new_reads */
                    /* level 0: "alt" then "ref"; deeper: "ref" then "alt" */
                    int hap_is_alt = level == 0 ? (oi == 0) : (oi == 1);
                    nrvec *NR = level == 0 ? &cur[oi] : &cur[oi]; /* cur laid out in iteration order */
                    for (int64_t e = 0; e < NR->n && !exception; e++) {
                        uint32_t q = (uint32_t)NR->v[e].q;
                        int64_t found_pos = NR->v[e].found_pos;
                        if (!Q.has_rs[q]) continue; /* :82-85 */
                        ivec *RS = &Q.read_sites[q];
                        for (int64_t si = 0; si < RS->n && !exception; si++) {
                            int32_t h = RS->v[si];
                            if ((int64_t)hpos[h] == found_pos) continue; /* :89-90 */
                            const uint8_t *fa_ = allele_at(&X, Q.fet0[q], Q.fet1[q], hpos[h], 1); /* :91-96 */
                            if (!fa_) continue;
                            uint8_t finder = *fa_, nonfinder;
                            uint8_t rb = S->ref_base[hidx[h]], ab = S->alt_base[hidx[h]];
                            if (finder == rb) nonfinder = ab;       /* :99-100 */
                            else if (finder == ab) nonfinder = rb;  /* :101-102 */
                            else continue;                          /* :104-105 */
                            if (!sr_exists[hcanon[h]]) { exception = 1; break; } /* :106 KeyError */
                            ivec *SR = &site_reads[hcanon[h]];
                            for (int64_t ri = 0; ri < SR->n; ri++) {
                                uint32_t q2 = (uint32_t)SR->v[ri];
                                if (Q.grp[q2]) continue; /* :108-110 */
                                int32_t read = Q.fet0[q2], mate = Q.fet1[q2];
                                const uint8_t *na = allele_at(&X, read, mate, hpos[h], 1); /* :114 */
                                if (!na) continue;
                                int rp = qidx(R, read, hpos[h]);
                                if (rp < 0) continue;                                      /* :120-121 */
                                if ((int)qualp(R, read)[rp] < P->min_gt_qual) continue;    /* :123-124 */
                                if (*na == finder) {                                       /* :134-136 */
                                    nr_push(&add[hap_is_alt ? 1 : 0], (int32_t)q2, hpos[h]);
                                    Q.grp[q2] |= (uint8_t)(hap_is_alt ? 2 : 1);
                                } else if (*na == nonfinder) {                             /* :137-141 */
                                    nr_push(&add[hap_is_alt ? 0 : 1], (int32_t)q2, hpos[h]);
                                    Q.grp[q2] |= (uint8_t)(hap_is_alt ? 1 : 2);
                                }
                            }
                        }
                    }
                }
                free(cur[0].v); free(cur[1].v);
                cur[0] = add[0]; cur[1] = add[1]; /* next level iterates "ref" then "alt" */
                level++;
                if (add[0].n + add[1].n <= 0) break; /* :143 */
            }
            free(cur[0].v); free(cur[1].v);
            /* :254-263 every grouped pair contributes both fetched segments */
            if (!exception) {
                for (int64_t t = 0; t < Q.touched.n; t++) {
                    uint32_t q = (uint32_t)Q.touched.v[t];
                    for (int hb = 0; hb < 2; hb++) {
                        if (!(Q.grp[q] & (1 << hb))) continue;
                        if (Q.fet0[q] < 0) continue; /* :258-259 */
                        iv_push(&hapsegs[hb], Q.fet0[q]);
                        iv_push(&hapsegs[hb], Q.fet1[q]);
                    }
                }
                if (keep_lists) {
                    for (int hb = 0; hb < 2; hb++) {
                        int64_t s0 = grp_all.n;
                        for (int64_t t = 0; t < Q.touched.n; t++) {
                            uint32_t q = (uint32_t)Q.touched.v[t];
                            if (Q.grp[q] & (1 << hb)) iv_push(&grp_all, (int32_t)q);
                        }
                        if (grp_all.n > s0) qsort(grp_all.v + s0, (size_t)(grp_all.n - s0), sizeof(int32_t), cmp_i32); /* (an empty list may have no storage yet) */
                        if (hb == 0) res->grp_off[2 * d + 1] = grp_all.n;
                    }
                }
            }
        }

        if (exception) res->status[d] = UZ_ST_REF_EXCEPTION;
        else {
            /* ---- match_informative_sites (site_searcher.py:50-78) + phase_by_reads (snv_phaser.py:16-70) ---- */
            ivec votes[4]; /* dad_reads, mom_reads, dad_sites, mom_sites */
            memset(votes, 0, sizeof(votes));
            int64_t n_match = 0;
            int32_t *msites = (int32_t *)malloc((size_t)(nc + 1) * sizeof(int32_t));
            for (int hb = 0; hb < 2; hb++) { /* hb 0 = "ref", 1 = "alt" */
                for (int64_t k = 0; k < hapsegs[hb].n; k++) {
                    int32_t seg = hapsegs[hb].v[k];
                    int nm = bsearch_sites(R->start[seg], R->end[seg], cpos, (int)nc, msites, (int)nc);
                    if (nm <= 0) continue;
                    int dad_alt = 0, mom_alt = 0; /* distinct alt_parent values <=> distinct ref_parent values */
                    for (int m = 0; m < nm; m++) {
                        if (cfl[msites[m]] & UZ_CF_ALT_DAD) dad_alt = 1; else mom_alt = 1;
                    }
                    if (dad_alt && mom_alt) continue; /* site_searcher.py:74-75 */
                    n_match++;
                    for (int m = 0; m < nm; m++) {
                        int32_t ci = msites[m];
                        int rp = qidx(R, seg, cpos[ci]); /* snv_phaser.py:28-33 */
                        if (rp < 0) continue;
                        if (rp >= (int)R->l_seq[seg]) continue; /* IndexError in the reference: unpinned */
                        uint8_t b = seqp(R, seg)[rp];
                        int from_ref;
                        if (b == S->ref_base[cidx[ci]]) from_ref = 1;       /* :41-42 */
                        else if (b == S->alt_base[cidx[ci]]) from_ref = 0;  /* :43-44 */
                        else continue;
                        int alt_is_dad = (cfl[ci] & UZ_CF_ALT_DAD) != 0;
                        /* :52-69: ref-parent read on the ref haplotype => DNM on alt_parent, ... */
                        int to_alt_parent = (from_ref && hb == 0) || (!from_ref && hb == 1);
                        int to_dad = to_alt_parent ? alt_is_dad : !alt_is_dad;
                        iv_push(&votes[to_dad ? 0 : 1], (int32_t)R->qname[seg]);
                        iv_push(&votes[to_dad ? 2 : 3], cpos[ci]);
                    }
                }
            }
            free(msites);
            if (n_match <= 0) res->status[d] = UZ_ST_NO_OVERLAP; /* snv_phaser.py:158-166 */
            else {
                res->status[d] = UZ_ST_OK;
                for (int k = 0; k < 4; k++) {
                    int64_t m = sort_unique(votes[k].v, votes[k].n); /* :169-185 sets */
                    res->counts[4 * d + k] = (int32_t)m;
                    res->vote_off[4 * d + k] = vote_all.n;
                    if (keep_lists)
                        for (int64_t t = 0; t < m; t++) iv_push(&vote_all, votes[k].v[t]);
                }
                /* summarize_record read-backed branch: unfazed.py:193-234 */
                int64_t dr = res->counts[4 * d], mr = res->counts[4 * d + 1], r = P->evidence_min_ratio;
                if (dr > 0 && dr >= r * mr) { res->origin[d] = UZ_OR_DAD; res->evidence[d] = res->counts[4 * d + 2]; }
                else if (mr > 0 && mr >= r * dr) { res->origin[d] = UZ_OR_MOM; res->evidence[d] = res->counts[4 * d + 3]; }
                else if (dr > 0 && mr > 0) { res->origin[d] = UZ_OR_AMBIGUOUS; res->evidence[d] = (int32_t)(dr + mr); }
            }
            for (int k = 0; k < 4; k++) free(votes[k].v);
        }
        if (res->status[d] != UZ_ST_OK)
            for (int k = 0; k < 4; k++) res->vote_off[4 * d + k] = vote_all.n;

        /* reset per-DNM scratch */
        for (int64_t t = 0; t < Q.touched.n; t++) {
            uint32_t q = (uint32_t)Q.touched.v[t];
            if (!ever[q]) { ever[q] = 1; iv_push(&all_touched, (int32_t)q); }
            Q.read_sites[q].n = 0; Q.has_rs[q] = 0; Q.fet0[q] = Q.fet1[q] = -1; Q.grp[q] = 0; Q.is_touched[q] = 0;
        }
        Q.touched.n = 0;
        for (int64_t k = 0; k <= nh; k++) free(site_reads[k].v);
        free(site_reads); free(sr_exists); free(hpos); free(hcanon); free(cpos);
        free(lists[0].v); free(lists[1].v); free(hapsegs[0].v); free(hapsegs[1].v);
    }
    res->init_off[2 * n] = init_all.n;
    res->grp_off[2 * n] = grp_all.n;
    res->vote_off[4 * n] = vote_all.n;
    res->init_seg = init_all.v; res->grp_q = grp_all.v; res->vote_val = vote_all.v;
    /* read_sites vectors are only ever allocated for touched names; they were kept for reuse */
    for (int64_t t = 0; t < all_touched.n; t++) free(Q.read_sites[all_touched.v[t]].v);
    free(all_touched.v);
    free(Q.read_sites); free(ever); free(Q.has_rs); free(Q.fet0); free(Q.fet1); free(Q.grp); free(Q.is_touched); free(Q.touched.v);
    free(X.good_cache);
    *out = res;
    return 0;
}

/* estimate_concordant_insert_len: read_collector.py:11-25.  tlen of the first
 * (insert_size_max_sample + 1) records of the file in file order.
 * np.percentile(x, 99.5) with linear interpolation, then int() -- the stdev of
 * the resulting scalar is 0 so --stdevs has no effect (quirk Q8). */
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}
double uzo_concordant_cutoff(const int32_t *tlen, int64_t n, int32_t readlen) {
    if (n <= 0) return NAN;
    int64_t *x = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) x[i] = llabs((int64_t)tlen[i] - 2 * (int64_t)readlen);
    qsort(x, (size_t)n, sizeof(int64_t), cmp_i64);
    /* numpy percentile, method="linear" (alpha = beta = 1):
     * virtual_indexes = n*q + (alpha + q*(1 - alpha - beta)) - 1, evaluated in this order */
    double q = 99.5 / 100.0;
    double vi = (double)n * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
    int64_t lo = (int64_t)floor(vi);
    int64_t hi = lo + 1 < n ? lo + 1 : lo;
    double g = vi - (double)lo;
    /* numpy _lerp: a + (b-a)*t, with the t>=0.5 branch b - (b-a)*(1-t) */
    double a = (double)x[lo], b = (double)x[hi];
    double v = g >= 0.5 ? b - (b - a) * (1.0 - g) : a + (b - a) * g;
    free(x);
    return (double)(int64_t)v; /* int(np.mean(scalar)) + 0*stdevs */
}

/* ------------------------------------------------------------------ allele-balance (CNV) stage
 * phase_by_snvs (sv_phaser.py:71-85) over the whole-region candidate list of uzo_find(UZ_FIND_WHOLE_REGION, search_dist 0):
 * every candidate votes for the parent its kid_allele names (site[site["kid_allele"]]); only DEL / DUP are phased (:401).
 * Then summarize_record (unfazed.py:193-298) on the read-backed counts rb[4] (NULL: none) and the CNV counts.
 * cnv_pos: [cand_off[n]] per DNM slice: dad's positions, then mom's, in list order. */
static void summarize_counts(long long dr, long long mr, long long ds, long long ms, long long cd, long long cm, long long r,
                             int32_t *origin, int32_t *evidence, int32_t *etype) {
    int org = UZ_OR_NONE, et = 0, ambig = 0;
    long long ev = 0;
    if (dr > 0 && dr >= r * mr) { org = UZ_OR_DAD; ev = ds; et = UZ_ET_READBACKED; }            /* :206-215 */
    else if (mr > 0 && mr >= r * dr) { org = UZ_OR_MOM; ev = ms; et = UZ_ET_READBACKED; }       /* :216-226 */
    else if (dr > 0 && mr > 0) { org = UZ_OR_AMBIGUOUS; ev = dr + mr; et = UZ_ET_AMBIGUOUS_READBACKED; ambig = 1; } /* :227-234 */
    if (cd > 0 && cd >= r * cm) {                                                                /* :239-262 */
        if (org == UZ_OR_MOM && !(et & UZ_ET_READBACKED)) { org = UZ_OR_NONE; ev += cd + cm; et = UZ_ET_AMBIGUOUS_BOTH; ambig = 1; }
        else {
            org = UZ_OR_DAD; ev = cd;
            if (et & UZ_ET_AMBIGUOUS_READBACKED) { et &= ~UZ_ET_AMBIGUOUS_READBACKED; ambig = 0; }
            et |= UZ_ET_ALLELE_BALANCE;
        }
    } else if (cm > 0 && cm >= r * cd) {                                                         /* :264-289 */
        if (org == UZ_OR_DAD && !(et & UZ_ET_READBACKED)) { org = UZ_OR_NONE; ev += cd + cm; et = UZ_ET_AMBIGUOUS_BOTH; ambig = 1; }
        else {
            org = UZ_OR_MOM; ev = cm;
            if (et & UZ_ET_AMBIGUOUS_READBACKED) et &= ~UZ_ET_AMBIGUOUS_READBACKED; /* ambig is NOT cleared here (:286-288) */
            et |= UZ_ET_ALLELE_BALANCE;
        }
    } else if (cd + cm > 0 && !(et & UZ_ET_READBACKED)) {                                        /* :290-298 */
        org = UZ_OR_NONE; ev += cd + cm; et |= UZ_ET_AMBIGUOUS_ALLELE_BALANCE; ambig = 1;
    }
    if (ambig) et |= UZ_ET_AMBIG_FLAG;
    *origin = org; *evidence = (int32_t)ev; *etype = et;
}

void uzo_summarize_counts(int32_t n, const int32_t *rb_counts, const int32_t *cnv_counts, int32_t ratio, int32_t *origin, int32_t *evidence,
                          int32_t *etype) {
    for (int32_t d = 0; d < n; d++) {
        const int32_t *rb = rb_counts ? rb_counts + 4 * (int64_t)d : NULL;
        summarize_counts(rb ? rb[0] : 0, rb ? rb[1] : 0, rb ? rb[2] : 0, rb ? rb[3] : 0, cnv_counts[2 * d], cnv_counts[2 * d + 1], ratio,
                         origin + d, evidence + d, etype + d);
    }
}

void uzo_phase_cnv(const uz_params *P, const uz_sites_view *S, const uz_dnms_view *D, const int64_t *cand_off, const int32_t *cand_idx,
                   const uint8_t *cand_flags, const int32_t *rb_counts, int32_t *cnv_counts, int32_t *cnv_pos, int32_t *origin,
                   int32_t *evidence, int32_t *etype) {
    for (int32_t d = 0; d < D->n; d++) {
        int64_t nd = 0, nm = 0;
        const int vt = D->vartype[d];
        if (vt == UZ_VT_DEL || vt == UZ_VT_DUP) {
            for (int pass = 0; pass < 2; pass++) { /* count, then fill dad's positions followed by mom's */
                int64_t od = cand_off[d], om = cand_off[d] + nd;
                for (int64_t j = cand_off[d]; j < cand_off[d + 1]; j++) {
                    const int ka = (cand_flags[j] >> UZ_CF_KA_SHIFT) & 3;
                    if (!ka) continue;
                    const int alt_dad = (cand_flags[j] & UZ_CF_ALT_DAD) != 0;
                    const int is_dad = (ka == UZ_KA_ALT_PARENT) == alt_dad;
                    if (!pass) { if (is_dad) nd++; else nm++; }
                    else if (cnv_pos) { if (is_dad) cnv_pos[od++] = S->pos[cand_idx[j]]; else cnv_pos[om++] = S->pos[cand_idx[j]]; }
                }
            }
        }
        cnv_counts[2 * d] = (int32_t)nd; cnv_counts[2 * d + 1] = (int32_t)nm;
        const int32_t *rb = rb_counts ? rb_counts + 4 * (int64_t)d : NULL;
        summarize_counts(rb ? rb[0] : 0, rb ? rb[1] : 0, rb ? rb[2] : 0, rb ? rb[3] : 0, nd, nm, P->evidence_min_ratio, origin + d, evidence + d,
                         etype + d);
    }
}
