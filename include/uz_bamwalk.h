/* uz_bamwalk.h -- the record walk of a staged batch ON THE DEVICE (what `bamfile.fetch(chrom, lo, hi)` walks for the
 * reference per DNM and het site, read_collector.py:385, :167).
 *
 * The BGZF blocks of a batch are inflated in HBM (csrc/k_inflate.hip) and STAY there: a wavefront per walk task follows the chain of
 * block_size fields, tests every record against the task's reach intervals and fetches (csrc/k_bamwalk.hip: k_bam_walk) and hands the
 * host one 64-byte descriptor per record inside a reach interval -- where the host's walk (csrc/io_stage.cpp: walk_task) copied the
 * record's bytes out of the inflated block.  The host runs the joins that need the whole batch (mates, names numbered by first appearance:
 * uz_bam_stage_finish_desc) on the descriptors and answers with the list of kept records (uz_stage_kept); a second kernel (k_bam_extract)
 * turns those records, still in HBM, into the columns of the device's record table.  Inflated bytes never cross the link.
 *
 * Shared by libunfazed_io.so (unfazed_io.h) and libunfazed_hip.so (unfazed_hip.h): plain C, no dependencies. */
#ifndef UZ_BAMWALK_H
#define UZ_BAMWALK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one walked record inside a reach interval of its task (csrc/io_stage.cpp: what walk_task pushes to `all`) */
typedef struct uz_walk_desc {
    uint64_t voff;   /* virtual offset of the record (of its block_size field) */
    uint64_t src;    /* byte offset of its fixed part (behind block_size) in the inflated buffer; bit 63: in the batch's aux bytes instead */
    uint64_t h1;     /* FNV-1a + final mix of the read name (io_stage.cpp: hash_name) */
    int32_t pos, end, tlen, mpos, mtid;
    uint32_t h2;     /* a second hash of the name (uz_name_hash2): two names are held equal when h1, h2 and the lengths agree */
    uint32_t task;
    uint16_t flag, l_seq, n_cigar;
    uint8_t mapq, l_name /* without the terminator */, direct /* 1: a fetch returns it */, pad8;
    uint16_t pad16;
} uz_walk_desc;

#define UZ_WALK_SRC_AUX (1ULL << 63)
/* uz_walk_desc.task of a record the HOST walked for the device's joins (uz_stage_walk_flagged, uz_stage_lookup): bit 31 + the join task it belongs
 * to -- a task of the stage, or n_tasks + k for the k-th look-up through the index; without the bit: the device's own walk task */
#define UZ_WALK_TASK_JOIN 0x80000000u

/* a member of the joins whose mate the walked tasks cannot answer (uz_bam_join -> uz_stage_lookup): its name, and where its mate is said to lie */
typedef struct uz_need_rec {
    uint64_t h1;
    uint32_t h2, l_name;
    int32_t mtid, mpos;
    uint32_t who; /* the device's index of the asking record */
    uint32_t pad;
} uz_need_rec;

/* the walk plan as flat arrays (uz_stage_walk_plan fills them, uz_bam_walk reads them) */
#define UZ_WALK_TASK_COLS 10 /* int32 per task: tid, b (first position behind its reach), span0, span1, reach0, reach1, fetch0, fetch1, fetch_max_len, host */
/* host (column 9): the task of the stage this walk task is a part of (uz_stage_walk_plan cuts a stage task into sub-tasks: groups of its reach
 * intervals, each a chain of its own for the device).  The sub-tasks of one stage task stand next to each other, in order, and the column never
 * decreases: the mate-candidate sets and the batch-wide joins are per stage task (uz_bam_walk refuses a plan that breaks this). */
#define UZ_WALK_SPAN_COLS 6  /* int64 per span: beg voff, end voff, buf_beg, buf_end, blk0, blk1 (its gathered blocks, indices into the block table) */
/* reach: int32 [2 n_reach] (a, b); fetch: int32 [3 n_fetch] (lo, hi, extra); blk_coff: int64 [n_blocks] file offset of every gathered block
 * (the block table of uz_stage_gather_blocks: out_off[k] .. out_off[k + 1] are its bytes in the inflated buffer) */

/* per task, out of the walk: bit 0 = the task's gathered bytes ended before its walk did (the host walks that task itself);
 * bit 1 = a malformed record (the host's walk reports it) */
#define UZ_WALK_TASK_INCOMPLETE 1
#define UZ_WALK_TASK_BAD 2

/* one kept record, host -> device (uz_stage_kept): where it lies and what the batch-wide joins found */
typedef struct uz_kept_rec {
    uint64_t src;      /* as in uz_walk_desc */
    uint32_t qname;    /* name id: order of first appearance in the batch */
    int32_t mate;      /* index of the record mate() returns, -1 none */
    uint32_t cig_off;  /* first CIGAR word in the table's store */
    uint32_t unit_off; /* first row unit of its quality plane (every record has one) */
    uint32_t seq_off;  /* first row unit of its bases, UZ_KEPT_NO_SEQ: the record travels without bases */
    uint32_t name_off; /* first byte of its read name in the batch's name store (names back to back, no terminators, record order) */
} uz_kept_rec;
#define UZ_KEPT_NO_SEQ 0xFFFFFFFFu

/* the first name hash: FNV-1a with a final mix (what csrc/io_stage.cpp has numbered names by since round 2) */
/* the second name hash (32 bits): murmur-style mix over the bytes, independent of FNV-1a */
#if defined(__HIPCC__)
#define UZ_BW_HD __host__ __device__ static inline
#else
#define UZ_BW_HD static inline
#endif
UZ_BW_HD uint32_t uz_name_hash2(const uint8_t *s, uint32_t n) {
    uint32_t h = 0x9747B28Cu ^ n;
    for (uint32_t i = 0; i < n; i++) { h ^= s[i]; h *= 0x5BD1E995u; h ^= h >> 15; }
    h ^= h >> 13; h *= 0x5BD1E995u; h ^= h >> 15;
    return h;
}

#ifdef __cplusplus
}
#endif
#endif /* UZ_BAMWALK_H */
