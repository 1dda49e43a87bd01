/* uzsynth_cpu.c -- host build of the benchmark-scale generator (uzsynth.h).
 * TEST / BENCH INFRASTRUCTURE: regenerates the read blocks of any DNM range on the CPU so
 * the oracle can be run and timed on exactly the data the GPU generated. */
#include <stdlib.h>
#include <string.h>

#include "uzsynth.h"

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* fills the blocks of DNMs [d0, d1) into `o` (arrays sized (d1-d0) * 2 * n_pairs records) */
int uzs_gen_reads_cpu(const uzs_cfg *c, const uzs_sites *S, const uzs_dnms *D, int32_t d0, int32_t d1, const uzs_out *o) {
    const int nseg = 2 * c->n_pairs;
    uint64_t *keys = (uint64_t *)malloc((size_t)nseg * sizeof(uint64_t));
    int32_t *inv = (int32_t *)malloc((size_t)nseg * sizeof(int32_t));
    for (int32_t d = d0; d < d1; d++) {
        for (int slot = 0; slot < nseg; slot++) {
            uzs_seg s;
            uzs_segment(c, D, d, slot >> 1, slot & 1, &s);
            keys[slot] = uzs_key(c, D, d, slot, &s);
        }
        qsort(keys, (size_t)nseg, sizeof(uint64_t), cmp_u64);
        for (int p = 0; p < nseg; p++) inv[keys[p] & 0xFFFF] = p;
        int64_t s_lo, s_hi;
        uzs_site_window(c, S, D, d, &s_lo, &s_hi);
        for (int p = 0; p < nseg; p++) {
            const int slot = (int)(keys[p] & 0xFFFF);
            uzs_seg s;
            uzs_segment(c, D, d, slot >> 1, slot & 1, &s);
            uzs_write_record(c, D, d, d0, slot, p, inv[slot ^ 1], &s, o);
            const int64_t i = (int64_t)(d - d0) * nseg + p;
            uint8_t *sq = o->seq + i * UZS_ROW, *ql = o->qual + i * UZS_ROW;
            uzs_fill(c, S, D, d, &s, s_lo, s_hi, 0, UZS_READLEN, sq, ql);
            memset(sq + UZS_READLEN, 0, UZS_ROW - UZS_READLEN);
            memset(ql + UZS_READLEN, 0, UZS_ROW - UZS_READLEN);
        }
    }
    free(keys);
    free(inv);
    return 0;
}
