/* uzsynth_cpu.c -- host build of the benchmark-scale generator (uzsynth.h).
 * TEST / BENCH INFRASTRUCTURE: regenerates the record blocks of any cluster range on the CPU, in the ASCII
 * form (uz_reads_view), so the oracle can be run and timed on exactly the data the GPU generated. */
#include <stdlib.h>
#include <string.h>

#include "uzsynth.h"

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* CIGAR words of clusters [c0, c1) (sizes the cigar column) */
int64_t uzs_count_ops_cpu(const uzs_cfg *cf, const uzs_clusters *C, const uzs_dnms *D, int32_t c0, int32_t c1) {
    int64_t total = 0;
    for (int32_t c = c0; c < c1; c++) {
        const int nseg = (int)(2 * (C->pair_off[c + 1] - C->pair_off[c]));
        for (int slot = 0; slot < nseg; slot++) {
            uzs_seg s;
            uzs_segment(cf, C, D, c, slot >> 1, slot & 1, &s);
            total += s.n_ops;
        }
    }
    return total;
}

/* fills the blocks of clusters [c0, c1) into `o`; record, pair and CIGAR numbering start at 0 for cluster c0 */
int uzs_gen_reads_cpu(const uzs_cfg *cf, const uzs_sites *S, const uzs_dnms *D, const uzs_clusters *C, int32_t c0, int32_t c1,
                      const uzs_out_ascii *o) {
    uint32_t *keys = (uint32_t *)malloc((size_t)UZS_MAXSEG * sizeof(uint32_t));
    int32_t *inv = (int32_t *)malloc((size_t)UZS_MAXSEG * sizeof(int32_t));
    int32_t *fid = (int32_t *)malloc((size_t)UZS_MAXSEG * sizeof(int32_t));
    const int64_t pair0 = C->pair_off[c0];
    int64_t cig = 0;
    for (int32_t c = c0; c < c1; c++) {
        const int nseg = (int)(2 * (C->pair_off[c + 1] - C->pair_off[c]));
        if (nseg > UZS_MAXSEG) { free(keys); free(inv); free(fid); return -2; }
        for (int slot = 0; slot < nseg; slot++) {
            uzs_seg s;
            uzs_segment(cf, C, D, c, slot >> 1, slot & 1, &s);
            keys[slot] = uzs_key(C, c, slot, &s);
        }
        qsort(keys, (size_t)nseg, sizeof(uint32_t), cmp_u32);
        for (int p = 0; p < nseg; p++) inv[keys[p] & 0x3FFF] = p;
        /* name ids as a decoder hands them out: in order of first appearance (a pair's id = pairs that start before it in the block) */
        for (int p = 0, run = 0; p < nseg; p++) { fid[p] = run; run += inv[(keys[p] & 0x3FFF) ^ 1] > p; }
        int64_t s_lo, s_hi;
        uzs_site_window(S, C, c, &s_lo, &s_hi);
        const int64_t rec0 = 2 * (C->pair_off[c] - pair0);
        for (int p = 0; p < nseg; p++) {
            const int slot = (int)(keys[p] & 0x3FFF);
            uzs_seg s;
            uzs_segment(cf, C, D, c, slot >> 1, slot & 1, &s);
            const int64_t i = rec0 + p;
            o->start[i] = s.start; o->end[i] = s.end; o->flag[i] = s.flag; o->mapq[i] = s.mapq; o->aux[i] = 1;
            o->tlen[i] = s.tlen;
            o->qname[i] = (uint32_t)(C->pair_off[c] - pair0 + fid[inv[slot ^ 1] > p ? p : inv[slot ^ 1]]);
            o->mate[i] = (int32_t)(rec0 + inv[slot ^ 1]);
            o->cigar_off[i] = (uint32_t)cig;
            o->n_cigar[i] = s.n_ops;
            for (int j = 0; j < s.n_ops; j++) o->cigar[cig++] = s.ops[j];
            o->l_seq[i] = (uint16_t)uzs_query_len(&s);
            o->sq_off16[i] = (uint32_t)(i * (UZS_ROW / 16));
            uint8_t *sq = o->seq + i * UZS_ROW, *ql = o->qual + i * UZS_ROW;
            uzs_fill(S, C, D, c, &s, s_lo, s_hi, 0, UZS_READLEN, sq, ql);
            memset(sq + UZS_READLEN, 0, UZS_ROW - UZS_READLEN);
            memset(ql + UZS_READLEN, 0, UZS_ROW - UZS_READLEN);
        }
    }
    free(keys);
    free(inv);
    free(fid);
    return 0;
}
