/* unfazed_hip.h -- C ABI of the MI355X (gfx950) per-DNM phasing path.
 *
 * The reference (unfazed v1.0.3) has no FFI seam; the drop-in boundary is the
 * pair of Python calls its driver makes (reference unfazed/unfazed.py:601-646):
 *     phase_snvs(dnms, kids, pedigrees, sites, threads, build, ...)  snv_phaser.py:356-399
 *     phase_svs (same 20 positional parameters)                      sv_phaser.py:427-493
 * unfazed_amd/snv_phaser.py and sv_phaser.py keep those two signatures and call
 * the entry points below through ctypes (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).  Each entry point names the reference
 * code it replaces.
 *
 * Conventions: every function returns 0 on success or a negative UZ_E_* code;
 * uz_last_error() gives the message.  No C++ exception, exit() or stdio crosses
 * this boundary.  All pointers are caller-owned HOST memory unless the function
 * name says `_device`.  One host thread drives one context; contexts are
 * independent (multi-GPU = one context per process/GPU, DNMs sharded by the
 * caller, no collective).
 */
#ifndef UNFAZED_HIP_H
#define UNFAZED_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "uz_types.h"
#include "uz_bamwalk.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UZ_E_ARG (-1)      /* bad argument / unknown handle */
#define UZ_E_HIP (-2)      /* a HIP runtime call failed */
#define UZ_E_NODEVICE (-3) /* no gfx950 device visible */
#define UZ_E_STATE (-4)    /* call order (e.g. fetch before find) */
#define UZ_E_RANGE (-5)    /* input exceeds an index range of the device layout */

#define UZ_FIND_WHOLE_REGION 1  /* find(..., whole_region=True): CNV interior, sv_phaser.py:375-389 */
#define UZ_FIND_SECOND_WINDOW 2 /* find()'s second window around `end` (informative_site_finder.py:32-40); find_many has none */

/* kernels whose launches are timed with HIP events on the context's stream */
#define UZ_K_SITE_SCAN 0   /* K1 site classify (the roofline kernel) */
#define UZ_K_WINDOW_COUNT 1
#define UZ_K_WINDOW_FILL 2
#define UZ_K_PHASE 3       /* the per-DNM read stage (both builds of k_phase) */
#define UZ_K_SIZING 4      /* fetch-range sizing pass */
#define UZ_K_CNV 5         /* K6 allele-balance count + decision */
#define UZ_K_COUNT 6

typedef struct uz_ctx uz_ctx;

/* ---- context ---------------------------------------------------------- */
int uz_create(int device, uz_ctx **out);
void uz_destroy(uz_ctx *ctx);
const char *uz_last_error(const uz_ctx *ctx);
int uz_sync(uz_ctx *ctx);
/* thresholds: replaces the module globals the reference sets per call
 * (informative_site_finder.py:187-204, read_collector.py:361-370) */
int uz_set_params(uz_ctx *ctx, const uz_params *p);

/* ---- staging ---------------------------------------------------------- */
/* Sites file columns -> HBM.  Replaces cyvcf2.VCF(...) + per-DNM tabix queries
 * (informative_site_finder.py:213, :42; find_many :558-568). */
int uz_sites_upload(uz_ctx *ctx, const uz_sites_view *sites, int *sites_id);
/* Genotype columns of one trio -> HBM (gt_types / gt_ref_depths / gt_alt_depths /
 * gt_quals of informative_site_finder.py:257-260). */
int uz_family_upload(uz_ctx *ctx, int sites_id, const uz_family_view *fam, int *fam_id);
/* Both at once, asynchronously: the copies are queued on the library's copy stream (behind whatever uz_reads_upload_packed
 * queued before) and the call returns; the first call that uses the family or the sites makes the compute stream wait for
 * them.  The host arrays (pinned memory for link speed, uz_pinned_alloc) must stay untouched until such a call has returned.
 * This is how a batch is streamed in sub-batches: the site stage of sub-batch k + 1 rides the link between the record
 * tables of sub-batches k - 1 and k instead of stopping it. */
int uz_sites_family_upload_async(uz_ctx *ctx, const uz_sites_view *sites, const uz_family_view *fam, int *sites_id, int *fam_id);
/* Alignment records of one BAM -> HBM.  Replaces pysam.AlignmentFile + fetch +
 * mate (read_collector.py:372-385, :400, :167, :185).  Whatever form a table arrives in, HBM holds the packed
 * one (uz_reads_packed_view in uz_types.h).
 * uz_reads_upload: the ASCII form (bases as characters, one quality byte per base); the rows are packed on the
 * device and the qualities are kept, so the table serves any --min-gt-qual.  Returns when the copy is done. */
int uz_reads_upload(uz_ctx *ctx, const uz_reads_view *reads, int *reads_id);
/* uz_reads_upload_packed: the staged form a decoder emits directly (4-bit bases, the quality-below-threshold
 * plane, no offset columns) -- 2.8x fewer bytes over the host link.  ASYNCHRONOUS: the copies are queued on the
 * context's copy stream, the header build behind them on a stream of its own (neither the next table's copies
 * nor the kernels of the table before wait for it), and the call returns; the host buffers (pinned memory for
 * full link speed, uz_pinned_alloc) must stay untouched until uz_reads_wait or a uz_phase on the table has
 * returned.  uz_phase on the table waits for the build on the device, so the upload and the build of the next
 * table overlap the kernels of the current one. */
int uz_reads_upload_packed(uz_ctx *ctx, const uz_reads_packed_view *reads, int *reads_id);
int uz_reads_wait(uz_ctx *ctx, int reads_id);
/* what the device made of a table's fixed-width columns, whatever form they travelled in (plain, 16- / 8-bit differences, the pair
 * form; `end` derived from the CIGAR when it was left out): [n_segs] each, any pointer may be NULL.  For parity tests of the
 * upload forms; waits for the table like uz_reads_wait. */
int uz_reads_headers(uz_ctx *ctx, int reads_id, int32_t *start, int32_t *end, int32_t *tlen, int32_t *mate, uint32_t *qname);
/* BGZF blocks inflated on the device (csrc/k_inflate.hip: one wavefront per block; fixed, dynamic and stored DEFLATE blocks).  This
 * entry is the measured form (repeat / kernel_ms) behind the parity tests; the session's path calls uz_bgzf_inflate_to_host below for
 * every staged batch (HipEngine.upload_reads_staged).  comp: the compressed bytes (host); in_off[k]: where the DEFLATE stream
 * of block k starts in them (behind its gzip header and BC field); out_off[k] .. out_off[k+1]: where its ISIZE bytes go in `out`
 * (host, [out_off[n_blocks]]).  repeat > 0: the kernel is run that many more times and *kernel_ms is their mean duration (HIP events).
 * A stream that does not decode to exactly its declared size fails the call (UZ_E_RANGE, the block named). */
int uz_bgzf_inflate(uz_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off,
                    uint8_t *out, int repeat, double *kernel_ms);
/* The working form: the same kernel on buffers the context keeps from call to call and on a stream of its own; comp and out in pinned
 * host memory for full link speed (uz_pinned_alloc).  This is what io_native.BamSource.select(inflate=...) calls between
 * uz_bam_stage_begin and uz_bam_stage_finish (unfazed_io.h): the blocks a batch's walk will read are gathered, inflated here, and handed
 * back; the host then copies records out of them instead of inflating (and still holds every block against its CRC-32). */
int uz_bgzf_inflate_to_host(uz_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off,
                            uint8_t *out);
/* ---- The record walk on the device (uz_bamwalk.h; the host half is uz_stage_walk_plan / uz_bam_stage_finish_desc / uz_stage_kept in unfazed_io.h).
 * What it replaces: `bamfile.fetch(chrom, lo, hi)` walking the records of a window (read_collector.py:385, :167) -- on the host until round 4, with
 * the inflated blocks crossing the link twice.
 * uz_bam_walk: the gathered BGZF blocks of a batch (uz_stage_gather_blocks: comp / in_off / out_off, blk_coff from uz_stage_walk_plan) go up, are
 * inflated in HBM and stay there; one wavefront per task of the plan walks them (every record inside a reach interval becomes a descriptor in HBM), and the
 * descriptors the batch-wide joins can need -- the records a fetch returns and those that share a name with one of them -- are counted.
 * -> *n_desc descriptors wait, *walk_id names the batch (four may be in flight).  Callable from a decoder's worker thread.
 * uz_bam_walk_fetch: those descriptors, task by task in file order: desc [n_desc], d_first [n_tasks + 1], d_flags / d_walked
 * [n_tasks] (UZ_WALK_TASK_*: a flagged task owns no descriptors -- the host walks it).
 * uz_reads_from_bam: the table of the kept records (uz_stage_kept), unpacked on the device from the bytes the walk left in HBM (and the aux bytes
 * of records only the host has seen) and built like an adopted table (uz_reads_adopt_device); releases the batch.  Owner's thread only.
 * uz_bam_walk_release: gives a walked batch up without building its table. */
int uz_bam_walk(uz_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off, const int64_t *blk_coff,
                const uint32_t *blk_crc /* [n_blocks] the CRC-32 of every block's footer: held against the inflated bytes (k_bgzf_crc32); NULL: not checked */,
                int32_t n_tasks, const int32_t *task, int64_t n_spans, const int64_t *span, int64_t n_reach, const int32_t *reach, int64_t n_fetch,
                const int32_t *fetch, int *walk_id, int64_t *n_desc);
/* k_bgzf_crc32 alone (the parity tests hold it against zlib): blocks data[off[k] .. off[k + 1]) of at most 64 KiB (host memory), want[k] their CRC-32;
 * *first_bad = -1, or a block whose checksum differs */
int uz_crc32_blocks(uz_ctx *ctx, const uint8_t *data, int64_t n_blocks, const int64_t *off, const uint32_t *want, int64_t *first_bad);
int uz_bam_walk_fetch(uz_ctx *ctx, int walk_id, uz_walk_desc *desc, int64_t *d_first, int32_t *d_flags, int64_t *d_walked);
int uz_bam_walk_release(uz_ctx *ctx, int walk_id);
/* ---- The batch-wide joins of a walked batch ON THE DEVICE (csrc/k_bamjoin.hip): mate() for every fetched record and every mate of a mate
 * (read_collector.py:400, :185), the names numbered by first appearance (the name-keyed tables of read_collector.py:226-234), file order, the
 * offsets of the record table -- through round 5 the host's uz_bam_stage_finish_desc over descriptors that crossed the link.  The descriptors stay
 * in the batch's slot; the host contributes only the records it has to walk itself (unfazed_io.h: uz_stage_walk_flagged, uz_stage_lookup), as
 * descriptors of its own (uz_walk_desc.task = UZ_WALK_TASK_JOIN | join task, .src = UZ_WALK_SRC_AUX | offset in the aux bytes).
 *   uz_bam_walk_flags   what the walk said of every walk task (d_flags, d_walked [n_tasks]) -> uz_stage_walk_flagged
 *   uz_bam_join         first call: n_host tasks of the stage, h_flags [n_host] (the tasks the host walked: their device descriptors are void),
 *                       n_ref references, all_bases; every call: xdesc [n_x] / xaux [xaux_bytes] = what the host walked SINCE the last call,
 *                       look_tid [n_look] = the reference of every look-up task so far, need_jtask = the answers to the last call's needs
 *                       (NULL on the first call).  -> *n_need > 0: the closure needs the index (uz_bam_join_needs -> uz_stage_lookup -> call again);
 *                       *n_need == 0: finished, totals = records, name ids, CIGAR words, row units, base-row units, name bytes
 *   uz_bam_join_fetch   parity / debug: the kept records in output order (any pointer may be NULL); contig_off [n_ref + 1], max_span [n_ref]
 *   uz_reads_from_walk  the record table, unpacked from the bytes in HBM through the kept list in HBM (releases the batch).  want_names: the read
 *                       names stay on the device with the table and uz_reads_names answers name ids (off [n + 1], the bytes in the context's page-locked memory)
 *   uz_walk_slot_stats  [0] device allocations the slots have made, [1] outgrown blocks parked, [2] their bytes, [4..7] the slots' inflated-bytes room */
int uz_bam_walk_flags(uz_ctx *ctx, int walk_id, int32_t *d_flags, int64_t *d_walked);
int uz_bam_join(uz_ctx *ctx, int walk_id, int32_t n_host, const int32_t *h_flags, int32_t n_ref, int all_bases, const uz_walk_desc *xdesc, int64_t n_x, const uint8_t *xaux,
                int64_t xaux_bytes, const int32_t *look_tid, int64_t n_look, const int32_t *need_jtask, int64_t *n_need, int64_t totals[8]);
int uz_bam_join_needs(uz_ctx *ctx, int walk_id, uz_need_rec *need);
int uz_bam_join_fetch(uz_ctx *ctx, int walk_id, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases, uz_kept_rec *kept, int64_t *contig_off, int32_t *max_span);
int uz_reads_from_walk(uz_ctx *ctx, int walk_id, int32_t min_base_qual, int want_names, int *reads_id, int64_t totals[8]);
int uz_reads_names(uz_ctx *ctx, int reads_id, const uint32_t *ids, int64_t n, int64_t *off /* [n + 1] */,
                   const uint8_t **bytes /* out: the names back to back in the context's page-locked memory, valid until the next call */);
int uz_walk_slot_stats(uz_ctx *ctx, int64_t out[8]);
/* every free slot among the first n_slots grown, now, to the largest sizes any batch of this context has asked for (a pipeline that will keep
 * n_slots batches in flight calls it once a first batch is through: no later batch's walk then pays for a slot's first gigabytes) */
int uz_walk_reserve(uz_ctx *ctx, int n_slots);
int uz_reads_from_bam(uz_ctx *ctx, int walk_id, const uz_kept_rec *kept, int64_t n, const uint8_t *aux, int64_t aux_bytes, const int64_t *contig_off,
                      const int32_t *max_span, int32_t n_contigs, int64_t n_cigar_total, int64_t n_row_units, int64_t n_seq_units, uint32_t n_qnames,
                      int32_t min_base_qual, uint8_t *names_out /* NULL, or [names_bytes]: the kept records' read names back to back (uz_kept_rec.name_off) */,
                      int64_t names_bytes, int *reads_id);
/* page-locked host memory for the staged columns (plain hipHostMalloc; no context needed) */
int uz_pinned_alloc(size_t bytes, void **out);
void uz_pinned_free(void *p);
/* The same for columns that already live in HBM (device pointers in the views; the library reads them in
 * place and never frees them). */
int uz_sites_adopt_device(uz_ctx *ctx, const uz_sites_view *sites, int *sites_id);
int uz_family_adopt_device(uz_ctx *ctx, int sites_id, const uz_family_view *fam, int *fam_id);
int uz_reads_adopt_device(uz_ctx *ctx, const uz_reads_packed_view *reads, int *reads_id);
/* Forget the derived columns (site classes of every family, per-record QC bits, the window lists of the last finds) so that
 * the next uz_find / uz_phase recomputes them: a timed "whole job" pass starts from the staged inputs only. */
int uz_drop_derived(uz_ctx *ctx);
int uz_sites_free(uz_ctx *ctx, int sites_id); /* also frees its families */
int uz_reads_free(uz_ctx *ctx, int reads_id);

/* ---- site stage ------------------------------------------------------- */
/* K1: one streaming pass over the family's columns -> one class byte per site
 * (UZ_CL_*).  Replaces is_high_quality_site (:46-73), get_kid_allele (:76-134) and
 * the DNM-independent part of find()'s per-variant body (:239-339 = :442-543). */
/* uz_site_scan / uz_site_classes compute every class bit; uz_find / uz_phase in SNV / breakpoint
 * mode run the variant without the DEL / DUP codes, which only whole_region=True reads (:286-291). */
int uz_site_scan(uz_ctx *ctx, int fam_id);
/* Cohort form (many kids / families in one sites VCF, README.md:208): classify n_fam
 * families of the SAME sites table in one launch.  Same result as n_fam calls of uz_site_scan. */
int uz_site_scan_many(uz_ctx *ctx, const int32_t *fam_ids, int32_t n_fam);
int uz_site_classes(uz_ctx *ctx, int fam_id, uint8_t *cls_out /* [n_sites] */);

/* K2: per-DNM window emit.  Replaces get_position (:10-43) / get_close_vars
 * (:399-420) and the list building of find (:262-343).  Runs uz_site_scan first
 * if the family's classes are stale.  Writes the CSR offsets; the lists stay in
 * HBM for uz_phase and can be copied out with uz_find_fetch. */
int uz_find(uz_ctx *ctx, int fam_id, const uz_dnms_view *dnms, int mode,
            int64_t *cand_off /* [n+1] */, int64_t *het_off /* [n+1] */);
int uz_find_fetch(uz_ctx *ctx, int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx);
/* (The lists of the last THREE finds stay in HBM: a uz_phase / uz_phase_begin over a batch -- same family, DNMs, mode and parameters --
 * that one of them covered takes its lists instead of running the window emit again.  A staged pass calls uz_find for chunk k + 1,
 * whose het lists tell the decoder what to stage, before it queues the read stage of chunk k.) */

/* ---- read stage ------------------------------------------------------- */
/* Everything multithread_read_phasing does after get_refalt (snv_phaser.py:131-203):
 * collect_reads_snv (+ group_reads_by_haplotype / connect_reads), match_informative_sites,
 * phase_by_reads, the unique-name / unique-site tally and summarize_record's integer
 * read-backed decision (unfazed.py:193-234).  Runs the window emit for the batch first
 * (find_mode: UZ_FIND_SECOND_WINDOW for find, 0 for find_many; never WHOLE_REGION),
 * whose lists can be read back with uz_find_fetch afterwards.
 * counts = dad_reads, mom_reads, dad_sites, mom_sites per DNM. */
int uz_phase(uz_ctx *ctx, int fam_id, int reads_id, const uz_dnms_view *dnms, int find_mode,
             int32_t *status /* [n] UZ_ST_* */, int32_t *counts /* [4n] */,
             int32_t *origin /* [n] UZ_OR_* */, int32_t *evidence /* [n] */);
/* uz_phase in two halves, for a caller that has the next batch's work ready: uz_phase_begin queues the window emit and the read
 * stage and returns without waiting (a first batch, whose sizes nothing predicts yet, runs to its end instead); uz_phase_end -- same
 * batch, same arguments -- waits and hands out the results of uz_phase.  Between the two the caller may upload tables and run
 * uz_site_scan / uz_find / uz_phase_cnv for OTHER batches (they queue up behind the read stage: the device does not idle through the
 * host's round trips); not another uz_phase / uz_phase_begin / uz_phase_cohort, and not uz_find_fetch / uz_phase_votes of this batch
 * before uz_phase_end has returned.  After an
 * intervening uz_find the window lists of the context are that batch's: uz_find_fetch then returns those. */
int uz_phase_begin(uz_ctx *ctx, int fam_id, int reads_id, const uz_dnms_view *dnms, int find_mode);
int uz_phase_end(uz_ctx *ctx, int fam_id, int reads_id, const uz_dnms_view *dnms, int find_mode,
                 int32_t *status, int32_t *counts, int32_t *origin, int32_t *evidence);
/* Cohort form (603 samples in one run, README.md:208; one alignment file per kid, unfazed.py:574-575): the
 * DNMs of many kids -- each with its own trio columns, its own alignment records and its own insert cutoff -- in ONE
 * launch sequence instead of one per kid.  `dnms` holds all groups' DNMs back to back; `rcontig` refers to the group's
 * own reads table and dnms->cutoff is ignored.  All families must belong to one sites table.  The kids' tables are laid
 * end to end in HBM as one table (virtual contigs = kid x contig; rebuilt only when the list of reads_ids changes), the
 * window emit reads every DNM's own family column, the read stage every DNM's own cutoff.  Results as uz_phase, in DNM
 * order; uz_phase_votes / uz_phase_groups afterwards give query-name ids of the group's own table. */
int uz_phase_cohort(uz_ctx *ctx, const uz_cohort_group *groups, int32_t n_groups, const uz_dnms_view *dnms, int find_mode,
                    int32_t *status, int32_t *counts, int32_t *origin, int32_t *evidence);
/* Vote lists of the last uz_phase (for --verbose and the records dict):
 * vote_off[4n+1] then vote_val: dad_reads (qname ids, ascending), mom_reads,
 * dad_sites (positions, ascending), mom_sites.  Call with vote_val == NULL to get
 * the offsets / total first. */
int uz_phase_votes(uz_ctx *ctx, int64_t *vote_off /* [4n+1] */, int32_t *vote_val);
/* Haplotype groups after connect_reads of the last uz_phase (diagnostics / tests):
 * grp_off[2n+1] then grp_q: "ref" set then "alt" set (qname ids, ascending). */
int uz_phase_groups(uz_ctx *ctx, int64_t *grp_off /* [2n+1] */, int32_t *grp_q);

/* ---- allele-balance (CNV) stage ---------------------------------------- */
/* K6.  Everything run_cnv_phasing does after its find (sv_phaser.py:357-423: phase_by_snvs :71-85,
 * multithread_cnv_phasing :269-301) and the decision of summarize_record (unfazed.py:193-298): runs the window
 * emit for the batch in UZ_FIND_WHOLE_REGION mode (search_dist 0, whole_region=True, :375-389), lets every candidate
 * of a DEL / DUP vote for the parent its kid_allele names, and decides.
 * rb_counts: optional [4n] read-backed counts of the same DNMs (dad_reads, mom_reads, dad_sites, mom_sites, e.g. the
 * `counts` of uz_phase): merged as summarize_record merges them (READBACKED + ALLELE-BALANCE, AMBIGUOUS_BOTH, ...);
 * NULL = allele-balance evidence only.
 * cnv_counts [2n] = cnv_dad_sites, cnv_mom_sites; origin UZ_OR_* (AMBIGUOUS = "dad|mom"); evidence = evidence_count;
 * etype = UZ_ET_* mask. */
int uz_phase_cnv(uz_ctx *ctx, int fam_id, const uz_dnms_view *dnms, const int32_t *rb_counts, int32_t *cnv_counts /* [2n] */,
                 int32_t *origin /* [n] */, int32_t *evidence /* [n] */, int32_t *etype /* [n] */);
/* Site lists of the last uz_phase_cnv: off[2n+1], then pos: cnv_dad_sites, cnv_mom_sites per DNM (positions in
 * candidate-list order).  pos == NULL: offsets only. */
int uz_phase_cnv_sites(uz_ctx *ctx, int64_t *off /* [2n+1] */, int32_t *pos);

/* ---- measurement ------------------------------------------------------ */
/* HIP-event timing of the kernels launched on the context's stream since the
 * last reset: total milliseconds and launch count per UZ_K_* id.
 * on: 0 none, 1 every id, otherwise a set of ids -- (1 << (id + 1)) for each id
 * wanted (two event records per timed launch are host work in front of the
 * launch: a caller that times a whole pass keeps the set small). */
int uz_prof_enable(uz_ctx *ctx, int on);
int uz_prof_reset(uz_ctx *ctx);
int uz_prof_get(uz_ctx *ctx, int kernel, double *total_ms, int64_t *launches);
/* Units the last launch of a kernel processed (K3a ids: alignment records examined; 0 when unknown). */
int uz_prof_units(uz_ctx *ctx, int kernel, int64_t *units);

#ifdef __cplusplus
}
#endif
#endif /* UNFAZED_HIP_H */
