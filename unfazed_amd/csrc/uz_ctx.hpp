// uz_ctx.hpp -- internal state of the C ABI context (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <memory>
#include <vector>

#include "unfazed_hip.h"
#include "uz_bamwalk.h"

struct UzError {
    int code;
    std::string msg;
};

#define UZ_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            throw UzError{UZ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)};               \
    } while (0)

#define UZ_REQUIRE(cond, code, text)          \
    do {                                      \
        if (!(cond)) throw UzError{code, text}; \
    } while (0)

// grow-only device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        size_t want = n + n / 4 + 64;
        UZ_HIP(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // The same for buffers that grow while OTHER work is running on the device (the slots of walked batches: uz_bam_walk on a decoder's thread,
    // beside the read stage of another chunk): hipFree waits for the whole device, so the block that was outgrown is parked, not freed (the
    // context frees the parked blocks when it is destroyed, or at a moment no walked batch is in flight), and the new one is sized for the largest
    // request any buffer of this kind has seen (*hi: a slot that grows once then fits every batch the process has staged).
    // used > 0: the first `used` elements are carried over (on stream st).
    void ensure_parked(size_t n, size_t *hi, std::vector<std::pair<void *, size_t>> &park, size_t used = 0, hipStream_t st = nullptr) {
        if (hi && n > *hi) *hi = n;
        if (n <= cap) return;
        const size_t base = hi ? *hi : n, want = base + base / 4 + 64;
        T *q = nullptr;
        UZ_HIP(hipMalloc((void **)&q, want * sizeof(T)));
        if (p && used) UZ_HIP(hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (p) park.push_back({(void *)p, cap * sizeof(T)});
        p = q;
        cap = want;
    }
};

// what a set of window lists was computed for
struct FindKey {
    bool valid = false;
    int fam = -1, mode = 0;
    int32_t n = -1;
    bool cohort = false;
    uint64_t hash = 0; // of the DNM columns the window emit reads
    uz_params P;
};
#define UZ_FIND_ALTS 2
struct FindSlot { // the window lists of one uz_find, parked
    FindKey key;
    unsigned long long stamp = 0;
    DevBuf<int32_t> cnt_c, cnt_h;
    DevBuf<int64_t> win_range;
    DevBuf<int64_t> cand_off, het_off;
    DevBuf<int32_t> cand_idx, het_idx;
    DevBuf<uint8_t> cand_flags;
    int64_t n_cand = 0, n_het = 0;
    std::vector<int64_t> cand_off_h, het_off_h;
};

// one device allocation out of the context's pool (tables come and go every staged pass: hipFree
// synchronises the device, so freed blocks are parked and handed out again)
struct DevBlock {
    uint8_t *p = nullptr;
    size_t cap = 0;
};

struct SitesDev {
    bool live = false, owned = false;
    DevBlock block; // owned tables: every column in one pooled block
    DevBlock mirror; // asynchronous upload of a slab of host columns: its image (sites + genotype columns; abi.hip: SlabPlan)
    int64_t n = 0;
    int32_t n_contigs = 0;
    std::vector<int64_t> contig_off_h;
    int64_t *contig_off = nullptr;
    int32_t *pos = nullptr;
    uint8_t *sflags = nullptr, *ref_base = nullptr, *alt_base = nullptr;
    hipEvent_t ready = nullptr; // asynchronous upload: end of the copies on the copy stream; the first use waits for it
    bool pending = false;
};

struct FamilyDev {
    bool live = false, owned = false;
    DevBlock block; // the class column and, for owned tables, the genotype columns
    int sites_id = -1;
    uint8_t *gt = nullptr; // bits 0-5 genotypes; bit 6 = the site's complex flag, folded in at upload
    uint16_t *rd[3] = {nullptr, nullptr, nullptr};
    uint16_t *ad[3] = {nullptr, nullptr, nullptr};
    uint16_t *gq[3] = {nullptr, nullptr, nullptr};
    uint8_t *cls = nullptr;
    // sites whose depths the 16-bit columns cannot hold (uz_family_view.wide_*): index + six 32-bit depths each, in a small block of
    // their own; their class bytes are rewritten after every site scan (k_site_scan_wide)
    int64_t n_wide = 0;
    int64_t *wide_site = nullptr;
    int32_t *wide_depth = nullptr; // [6][n_wide]: rd kid, dad, mom, ad kid, dad, mom
    DevBlock wide_block;
    std::shared_ptr<std::vector<int32_t>> wide_host; // the host copy the (asynchronous) upload of wide_depth reads: kept with the family
    // asynchronous upload (uz_sites_family_upload_async): the copies are queued on the copy stream; `ready` marks their end and
    // the first use of the family makes the compute stream wait for it and folds the complex flag (family_make_ready)
    hipEvent_t ready = nullptr;
    bool pending = false;
    // eight-bit link form (uz_family_view.ref_depth8 ...): the staged bytes, widened into rd / ad / gq at first use; gq was clamped at 254
    const uint8_t *stage8[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool widen_pending = false, gq_clamped = false;
    bool cls_valid = false;
    bool cls_has_cnv = false; // DEL / DUP codes (bits 3-6) computed
    uz_params cls_params;
};

struct ReadsDev {
    bool live = false;
    int64_t n = 0;
    int32_t n_contigs = 0;
    uint32_t n_qnames = 0;
    int64_t n_cigar_total = 0, n_row_units = 0, n_seq_units = 0;
    DevBlock block;            // everything the library owns for this table lives in this one block
    DevBlock mirror;           // ... except, when the host columns came over the link as one slab, the slab's image (abi.hip: SlabPlan)
    int64_t *contig_off = nullptr;
    int32_t *max_span = nullptr;
    void *rec_a = nullptr, *rec_b = nullptr; // RecA / RecB headers (phase_body.hpp)
    uint32_t *fm = nullptr;    // flag | mapq << 16 | aux << 24
    uint32_t *qoff = nullptr;  // quality-plane row of every record (row units); UZ_NO_QLOW_OFF: none (its bits can never be asked for)
    uint16_t *umask = nullptr; // staged 32-base units of every record's rows (UZ_UMASK_ALL: every unit)
    uint8_t *nlow = nullptr;   // low-quality bases of every record, saturated at 255 (what K3a needs of the qualities)
    int64_t n_qlow_pos = 0;    // list form of the staged plane: entries (checked against the columns by the header build)
    int64_t n_plane_units = 0; // units the quality-plane store holds: n_row_units (plane / ASCII form: every record has a row), n_seq_units (list form)
    const uint32_t *cigar = nullptr;
    const uint8_t *seq4 = nullptr;
    uint8_t *qlow = nullptr;   // caller's memory for adopted tables (then never written)
    int32_t qlow_thr = 0;      // threshold the qlow plane holds
    bool qlow_valid = false;
    // ASCII uploads (uz_reads_upload) keep the full qualities so that the plane can be rebuilt for another threshold
    uint8_t *qual8 = nullptr;
    uint32_t *qual_off16 = nullptr;
    uint16_t *qs = nullptr;    // per-record QC word (phase_body.hpp: uz_qs_word), written by the header build
    int32_t *coarse = nullptr; // start of every 4096th record
    int32_t *mid = nullptr;    // start of every 64th record (phase_body.hpp: uz_lower_bounds_c)
    int32_t *mid8 = nullptr;   // start of every 8th record (uz_mid8_refine)
    hipEvent_t ready = nullptr; // asynchronous uploads: recorded behind the last copy of the upload
    bool pending = false;       // the copies may still be in flight and the headers not built yet: the first use waits (uz_reads_make_ready)
    // The header build of an asynchronous upload is queued with the upload on a stream of its own (uz_ctx::build_stream), behind the
    // copies' event: on the copy stream it would hold up the next table's copies, on the compute stream (where it ran at first use) it
    // sat in the chain of every chunk's read stage.  `built` marks its end; null: no build queued yet, the first use runs it
    // (UZ_BUILD_LAZY=1).
    hipEvent_t built = nullptr;
    const void *col_ptrs[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *build_scratch = nullptr;
    const void *col_t[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // tup, tup_flag, tup_l_seq, tup_n_cigar, tup_mapq, tup_aux, tup_n_low, tup_umask
    const void *col_t8[4] = {nullptr, nullptr, nullptr, nullptr}; // tup8, tup_hot, tup_esc, tup_esc_off (the one-byte index: col_t[0] is then where the 16-bit column is rebuilt)
    int64_t col_ntesc = 0;
    int32_t col_lists = 0;
    int64_t col_ntup = 0;
    const void *col_d[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // start_d, tlen_s, mate_d, qname_d, esc16_key, esc16_val, start_d8, mate_d8, qname_d8, pair_d8
    int64_t col_nesc = 0;
    const void *col_q[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // plane_in / n_low / qlow_pos / cigar_in / umask / cigar_staged / cigar_out of RecColumns, for the deferred header build
    int64_t n_cigar_staged = 0; // cigar_compact: words that travelled (checked by the header build)
    int32_t col_qwide = 0;
    // a table that arrived with two-bit base rows (uz_reads_packed_view.seq2): the staged rows and the listed bases, expanded
    // into seq4 by the header build (uz_build_records); null afterwards / for four-bit tables
    const uint8_t *seq2_staged = nullptr;
    const uint32_t *exc_rec = nullptr;
    const uint16_t *exc_pos = nullptr;
    const uint8_t *exc_code = nullptr;
    int64_t n_exc = 0;
    // bases as lists (uz_types.h bl_*): the listed bases and the row units of the records that carry them (inside n_seq_units: their units
    // lie behind the n_seq_units - n_bl_units that travelled as rows); bl_n / tup_n_bl / bl_pos / bl_code for the deferred header build
    int64_t n_bl = 0, n_bl_units = 0;
    const void *col_pk = nullptr; // pk_sums of RecColumns, for the deferred header build
    const void *col_b[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t col_bwide = 0;
    // a table built from a batch joined on the device (uz_reads_from_walk with names): the kept list, the record that brought every name id first and
    // the read names of the records, all in `mirror` -- uz_reads_names answers ids from them
    const uz_kept_rec *kept_list = nullptr;
    const uint32_t *name_rec = nullptr;
    const uint8_t *names = nullptr;
    int64_t names_bytes = 0;
};

// DNM batch staged on the device
template <typename T>
struct DevPtr { T *p = nullptr; };
struct DnmsDev { // the columns of a DNM batch: one block in HBM, laid out like its pinned staging copy (ONE copy brings a batch over)
    int32_t n = 0;
    DevBuf<uint8_t> block;
    DevPtr<int32_t> contig, rcontig, start, end;
    DevPtr<uint8_t> vartype, dflags, mult;
    DevPtr<uint32_t> allele_off;
    DevPtr<uint8_t> alleles;
    double cutoff = 0;
};

struct ProfSlot {
    double total_ms = 0;
    int64_t launches = 0;
    int64_t last_units = 0;
};
struct ProfPending {
    int kernel;
    hipEvent_t a, b;
};

struct uz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // compute (and the synchronous uploads)
    hipStream_t copy_stream = nullptr; // asynchronous uploads of packed tables: the copies
    hipStream_t build_stream = nullptr; // ... and their header builds (created at the first asynchronous upload)
    int32_t *hflags = nullptr;         // pinned, device-visible: [0] upload consistency error (totals / alphabet), [1] bases of a row-less
                                       // record requested; [4..7] two int64 mailboxes (list totals of the window emit)
    std::vector<DevBlock> block_pool;
    uz_params P;
    std::string err;
    std::mutex err_mu; // (err is written by whichever thread's call failed: abi.hip set_error)
    std::vector<SitesDev> sites;
    std::vector<FamilyDev> fams;
    std::vector<ReadsDev> reads;

    // allele-balance threshold table of K1 (k_sites.hip), rebuilt when the thresholds change
    DevBuf<int32_t> ab_lut;
    DevBuf<uint8_t> fam_batch; // cohort scan: per-family column pointers
    bool ab_lut_valid = false;
    bool ab_lut_t0_special = false;
    uz_params ab_lut_params;

    // last find
    bool find_valid = false;
    int find_fam = -1, find_mode = 0;
    DnmsDev dn;
    uint8_t *dn_stage = nullptr; // pinned staging of a DNM batch
    uint8_t *find_pin = nullptr; // pinned landing place of a find's offsets (copied there by a kernel: see uz_kcopy)
    size_t find_pin_cap = 0;
    size_t dn_stage_cap = 0;
    hipEvent_t dn_stage_done = nullptr; // behind the copies out of dn_stage: a batch queued by uz_phase_begin may still be reading it
    DevBuf<int64_t> scan_part;   // tile sums of the find's two count arrays (k_scan2_sums -> k_scan2)
    DevBuf<int32_t> cnt_c, cnt_h;
    DevBuf<int64_t> win_range;
    DevBuf<int64_t> cand_off, het_off;
    DevBuf<int32_t> cand_idx, het_idx;
    DevBuf<uint8_t> cand_flags;
    int64_t n_cand = 0, n_het = 0;
    std::vector<int64_t> cand_off_h, het_off_h;
    // Whose window lists these are, and the lists of the two finds BEFORE the last one, kept aside (find_alt): a staged pass asks for
    // the lists of chunk k + 1 (uz_find: the decoder's input) before it queues the read stage of chunk k, whose own uz_find ran one
    // call earlier -- the read stage then takes those lists instead of running the window emit (and its host round trip) again; the
    // third set leaves room for the whole-region lists of an allele-balance stage in between.
    FindKey find_key;
    unsigned long long find_stamp = 0, find_counter = 0;
    FindSlot find_alt[UZ_FIND_ALTS];

    // cohort batch (uz_phase_cohort): per-DNM family / insert cutoff / query-name base, the class column of every family, and
    // the reads tables of the kids laid end to end as ONE table (virtual contigs = kid x contig)
    bool cohort_on = false;
    DevBuf<int32_t> dn_fam;
    DevBuf<double> dn_cutoff;
    DevBuf<uint8_t *> fam_cls;
    std::vector<uint32_t> phase_qbase; // per DNM: first query-name id of its kid in the merged table (empty: no offset)
    int cohort_reads = -1;             // slot in `reads` holding the merged table
    std::vector<int> cohort_ids;       // the tables it was built from

    // last CNV stage (K6)
    bool cnv_valid = false;
    int32_t cnv_n = 0;
    DevBuf<int32_t> cnv_counts, cnv_pos, cnv_origin, cnv_evidence, cnv_etype, cnv_rb;
    std::vector<int32_t> cnv_counts_h;

    // BGZF inflate (k_inflate.hip, uz_bgzf_inflate_to_host): buffers kept from call to call, a stream of its own (a batch's blocks can
    // be inflated while the read stage of the batch before is still queued on the compute stream)
    DevBuf<uint8_t> inf_comp, inf_out;
    DevBuf<int64_t> inf_in, inf_off;
    DevBuf<int32_t> inf_flags;
    hipStream_t inf_stream = nullptr, inf_stream2 = nullptr;
    hipEvent_t inf_ready = nullptr;

    // the record walk on the device (k_bamwalk.hip, uz_bam_walk): a batch's inflated blocks stay in HBM from the walk until the batch's table has
    // been packed from them (uz_reads_from_bam); four batches can be in flight (a feed pipeline walks up to three chunks ahead of the read stage)
    struct WalkSlot {
        bool busy = false;
        DevBuf<uint8_t> comp, out;
        DevBuf<int64_t> in_off, out_off, blk_coff, span, count, first, walked;
        DevBuf<int32_t> task, reach, fetch, flags, iflags;
        DevBuf<uint32_t> blk_crc;
        DevBuf<uz_walk_desc> desc, desc_kept;
        DevBuf<int64_t> n_direct, tab_first, kcount, kfirst;
        DevBuf<unsigned long long> tab;
        int64_t n_blocks = 0, out_bytes = 0, n_desc = 0, n_desc_all = 0, n_reach = 0;
        int32_t n_tasks = 0, max_host = -1; // (max_host: the last walk task's stage task, column 9 of the plan)
        hipStream_t s0 = nullptr, s1 = nullptr; // the slot's own streams (blocks up + inflate in slices on both, the walk on the first)
        hipEvent_t ev = nullptr;
        // the batch-wide joins on the device (k_bamjoin.hip, uz_bam_join): all descriptors of the batch -- the device's own (desc_kept[0 .. n_dev)) and
        // behind them what the host walked itself (tasks handed back, mates looked up through the index) -- with their join state
        struct Join {
            int64_t n_dev = 0, n_all = 0, aux_bytes = 0;
            int32_t n_host = 0, n_look = 0, n_ref = 0, round = 0;
            int64_t n_need = 0;
            bool filled = false, started = false, done = false, all_bases = false;
            DevBuf<int32_t> jtask, keep, mate, target, h_flags, jt_tid, reach_a, reach_host, cnt, cspan, look_tid;
            DevBuf<int64_t> reach_key, totals;
            DevBuf<unsigned long long> hkey_in, hkey, fkey_in, fkey, ccount;
            DevBuf<uint32_t> hval_in, hperm, inv, fval_in, fidx, front0, front1, need, first, runid, pos_of_k, fo, name_rec;
            DevBuf<int32_t> gidx;
            DevBuf<uint8_t> tmp, aux, s5_in, s5_out;
            DevBuf<uz_need_rec> need_rec;
            DevBuf<uz_kept_rec> kept;
            int64_t tot_h[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // the totals of the finished join (k_bamjoin.hip: JT_*)
            std::vector<int64_t> contig_off_h;
            std::vector<int32_t> max_span_h;
        } join;
    };
    static constexpr int WALK_SLOTS = 4;
    WalkSlot walk[WALK_SLOTS];
    // growth of the slots' buffers (DevBuf::ensure_parked): the largest request every kind of buffer has seen, and the blocks that were outgrown
    size_t walk_hi[96] = {0};
    std::vector<std::pair<void *, size_t>> walk_park;
    int64_t walk_allocs = 0; // device allocations the slots have made (uz_walk_slot_stats: a process whose batches stopped growing makes none)
    std::mutex walk_mu;      // guards the three above (walks and joins of different slots run on different decoder threads)

    // uz_reads_names: ids / lengths / offsets / bytes on the device and one page-locked staging block, kept from call to call
    DevBuf<uint32_t> nm_ids, nm_len, nm_off;
    DevBuf<uint8_t> nm_out;
    uint8_t *nm_pin = nullptr;
    size_t nm_pin_cap = 0;

    // last phase (k_reads.hip)
    bool phase_valid = false;
    bool phase_open = false; // uz_phase_begin without its uz_phase_end
    int32_t phase_n = 0;
    void *phase_state = nullptr;

    // profiling
    uint32_t prof_mask = 0; // bit k: kernel id k is timed (uz_prof_enable)
    ProfSlot prof[UZ_K_COUNT];
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> event_pool;
};

// a buffer of a walk slot grown without a hipFree (DevBuf::ensure_parked); kind: which of uz_ctx::walk_hi remembers the largest request
template <typename T>
inline void uz_walk_grow(uz_ctx *c, DevBuf<T> &b, size_t n, int kind, size_t used = 0, hipStream_t st = nullptr) {
    if (n <= b.cap) {
        if (n > c->walk_hi[kind]) { std::lock_guard<std::mutex> lk(c->walk_mu); if (n > c->walk_hi[kind]) c->walk_hi[kind] = n; }
        return;
    }
    std::lock_guard<std::mutex> lk(c->walk_mu);
    b.ensure_parked(n, &c->walk_hi[kind], c->walk_park, used, st);
    c->walk_allocs++;
}

// profiling helpers (abi.hip)
void uz_prof_begin(uz_ctx *c, int kernel, hipEvent_t *a, hipEvent_t *b);
void uz_prof_end(uz_ctx *c, int kernel, hipEvent_t a, hipEvent_t b);
void uz_prof_drain(uz_ctx *c);

struct ProfScope {
    uz_ctx *c;
    int k;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(uz_ctx *c_, int k_) : c(c_), k(k_) { uz_prof_begin(c, k, &a, &b); }
    ~ProfScope() { uz_prof_end(c, k, a, b); }
};

// device block pool (abi.hip)
DevBlock uz_block_get(uz_ctx *c, size_t bytes);
void uz_block_put(uz_ctx *c, DevBlock b);

// the fixed-width columns of a table as DEVICE pointers (staged by an upload, or the caller's for an adopted table)
struct RecColumns {
    const int32_t *start = nullptr, *end = nullptr, *tlen = nullptr, *mate = nullptr;
    const uint32_t *qname = nullptr;
    const uint16_t *flag = nullptr, *l_seq = nullptr, *n_cigar = nullptr;
    const uint8_t *mapq = nullptr, *aux = nullptr;
    // qualities of the staged form: the plane itself (plane_in: the header build counts its bits into nlow) or its list
    // form (n_low + qlow_pos: the header build copies the counts and writes the plane rows of the listed records); both
    // null for an ASCII upload, whose plane and counts are built from the quality bytes (uz_build_qlow)
    const uint32_t *cigar_staged = nullptr; // cigar_compact: the words that travelled (records with a simple code own none); the header build writes
    uint32_t *cigar_out = nullptr;          // ... every record's words here (the device's CIGAR store).  Both null: `cigar_in` is the store itself
    const uint32_t *cigar_in = nullptr; // the record's CIGAR words (at its cigar offset): `end` is derived from them when the column is left out (end == nullptr)
    const uint16_t *umask = nullptr;    // staged units per record (null: every unit)
    // dictionary form of the small columns (tup set: flag / l_seq / n_cigar / mapq / aux / n_low are read through the table)
    const uint16_t *tup = nullptr, *tup_flag = nullptr, *tup_l_seq = nullptr, *tup_n_cigar = nullptr;
    const uint8_t *tup_mapq = nullptr, *tup_aux = nullptr, *tup_n_low = nullptr;
    const uint16_t *tup_umask = nullptr;
    int64_t n_tup = 0; // entries of the dictionary (0: not known)
    // the index in one byte (uz_types.h tup8): the header build first rebuilds the 16-bit column into tup_out (k_tup_expand) and reads it as `tup`
    const uint8_t *tup8 = nullptr;
    const uint16_t *tup_hot = nullptr, *tup_esc = nullptr;
    const uint32_t *tup_esc_off = nullptr;
    int64_t n_tup_esc = 0;
    uint16_t *tup_out = nullptr;
    // bases as lists (uz_types.h: bl_*): per record the number of listed bases (plain column or through the dictionary), their query indices and
    // two-bit codes; seq4_out: the device's base rows (the header build writes the units of the listed records, behind the n_seq_link units that
    // travelled as rows)
    const uint8_t *bl_n = nullptr, *tup_n_bl = nullptr, *bl_pos = nullptr, *bl_code = nullptr;
    int32_t bl_wide = 0;
    int64_t n_seq_link = 0;
    uint32_t *seq4_out = nullptr;
    __host__ __device__ bool bl_form() const { return bl_n != nullptr || tup_n_bl != nullptr; }
    // 16-bit difference form of start / tlen / mate / qname (start_d set: the plain four are null)
    const int16_t *start_d = nullptr, *tlen_s = nullptr, *mate_d = nullptr, *qname_d = nullptr;
    const uint8_t *start_d8 = nullptr; // the start differences in eight bits (then start_d is null)
    const int8_t *mate_d8 = nullptr, *qname_d8 = nullptr; // mate / name-id differences in eight bits (then mate_d / qname_d are null)
    const uint8_t *pair_d8 = nullptr; // the pair form: tlen, mate and name id in one byte (then tlen_s and the four above are null)
    __host__ __device__ bool diff_form() const { return tlen_s != nullptr || pair_d8 != nullptr; } // start (and the rest) travel as differences
    const unsigned long long *esc16_key = nullptr;
    const int32_t *esc16_val = nullptr;
    int64_t n_esc16 = 0;
    int64_t esc_lo = 0, esc_hi = 0;   // set by the header build's workgroups: the escape entries of their own span of records
    int32_t pk_shift = 12;            // set by the header build: records per span of its passes = 1 << pk_shift (a small table gets shorter spans: more workgroups)
    int32_t lists = 0; // the qualities came as counts (+ positions): n_low, or tup_n_low through the table
    const uint32_t *plane_in = nullptr;
    const uint8_t *n_low = nullptr, *qlow_pos = nullptr;
    int32_t qpos_wide = 0;
    // the span sums from the packer (uz_types.h pk_sums: [(spans + 1) * UZ_PK_SUMS], exclusive): null -- the header build computes them itself
    const unsigned long long *pk_sums = nullptr;
};
// Small transfers on the COMPUTE path go through a copy kernel, one side in pinned host memory, never through
// hipMemcpyAsync: the DMA engine is in order, and a 2 KB result copy queued behind gigabytes of staged uploads would hold
// the kernels of the current table back until every later table has landed (measured: copies and kernels did not
// overlap at all).  dst / src: device memory or pinned host memory.
void uz_kcopy(uz_ctx *c, void *dst, const void *src, size_t bytes);
// waits for an asynchronous upload and builds its headers, on the compute stream (no-op for a table that is ready)
void uz_reads_make_ready(uz_ctx *c, ReadsDev &r);
// stage launchers
// offsets (prefix sums of n_cigar / row units), RecA / RecB / fm and the coarse index, on stream `st`;
// off_scratch: >= uz_rec_scratch_bytes(n) bytes of device memory
size_t uz_rec_scratch_bytes(int64_t n);
void uz_build_records(uz_ctx *c, hipStream_t st, ReadsDev &r, const RecColumns &col, void *off_scratch);
// ASCII upload: CIGAR words gathered back to back, bases packed to 4 bits (sets hflags[0] on a character outside the alphabet)
void uz_pack_ascii_rows(uz_ctx *c, hipStream_t st, ReadsDev &r, const uint32_t *cigar_in, const uint32_t *cigar_off_in,
                        const uint8_t *seq_in, const uint32_t *sq_off16_in, uint32_t *cigar_out, uint8_t *seq4_out);
void uz_build_qlow(uz_ctx *c, hipStream_t st, ReadsDev &r, int min_base_qual);
// appends table `src` to the merged table `dst` at the given bases (records, CIGAR words, row units, query names)
void uz_concat_table(uz_ctx *c, hipStream_t st, ReadsDev &dst, const ReadsDev &src, int64_t rec_base, int64_t cigar_base, int64_t unit_base,
                     int64_t seq_base, uint32_t qname_base);
void uz_finish_table(uz_ctx *c, hipStream_t st, ReadsDev &r); // coarse index of a table whose headers are in place
void uz_family_widen(uz_ctx *c, FamilyDev &f, int64_t n_sites); // eight-bit link columns -> the 16-bit ones, on the compute stream (once)
void uz_launch_site_scan(uz_ctx *c, FamilyDev &f, const SitesDev &s, bool with_cnv);
void uz_launch_site_scan_many(uz_ctx *c, FamilyDev *const *fams, int n_fam, const SitesDev &s, bool with_cnv);
void uz_launch_find(uz_ctx *c, FamilyDev &f, const SitesDev &s, int mode, bool host_offsets = true);
void uz_stage_dnms(uz_ctx *c, const uz_dnms_view *d);
void uz_launch_cnv(uz_ctx *c, const SitesDev &s, const int32_t *rb_counts_dev, int32_t *cnv_counts, int32_t *cnv_pos, int32_t *origin,
                   int32_t *evidence, int32_t *etype);
void uz_launch_phase(uz_ctx *c, FamilyDev &f, const SitesDev &s, ReadsDev &r, int32_t *status, int32_t *counts,
                     int32_t *origin, int32_t *evidence, bool defer = false);
bool uz_finish_phase(uz_ctx *c, int32_t *status, int32_t *counts, int32_t *origin, int32_t *evidence);
// BGZF blocks inflated on the device (k_inflate.hip): device pointers; comp padded by 1 KiB; cursor_and_err: two int32 of device memory
void uz_launch_inflate(uz_ctx *c, hipStream_t st, int64_t n_blocks, const uint8_t *comp, int64_t comp_bytes_padded, const int64_t *in_off,
                       const int64_t *out_off, uint8_t *out, int32_t *cursor_and_err);
// the record walk on the device (k_bamwalk.hip): every pointer device memory; fill = false: counts + offsets (count, first, walked, flags), fill = true: the descriptors
void uz_launch_bam_walk(uz_ctx *c, hipStream_t st, int n_tasks, const uint8_t *buf, const int64_t *blk_at, const int64_t *blk_coff, const int32_t *task,
                        const int64_t *span, const int32_t *reach, const int32_t *fetch, int64_t *count, const int64_t *first, int64_t *walked, int32_t *flags,
                        uz_walk_desc *out, int64_t *n_direct, int64_t *tab_first);
void uz_launch_desc_filter(uz_ctx *c, hipStream_t st, bool fill, int n_tasks, const uz_walk_desc *in, const int64_t *first, const int64_t *count,
                           const int32_t *task, const int64_t *tab_first, unsigned long long *tab, int64_t *kcount, int64_t *kfirst, uz_walk_desc *out);
size_t uz_bam_walk_pad(); // bytes the inflated buffer must be padded by (the walk's LDS windows read past the last record)
// the batch-wide joins on the device (k_bamjoin.hip).  Totals of a finished join (WalkSlot::Join::tot_h):
enum { JT_N = 0, JT_QNAMES = 1, JT_CIGAR = 2, JT_UNITS = 3, JT_SEQ_UNITS = 4, JT_NAME_BYTES = 5, JT_ERR = 6, JT_COUNT = 8 };
struct JoinPlanHost { // what the host knows when it starts the joins of a walked batch
    int32_t n_host = 0, n_ref = 0;
    const int32_t *h_flags = nullptr; // [n_host]: the tasks the host walked itself (their device descriptors are void)
    bool all_bases = false;
};
// first call: fills the filtered descriptors, sets the tables up.  Every call: appends what the host walked since the last one (x [n_x], aux bytes),
// sets the answers to the last call's needs (need_jtask [n_need of the last call]) and runs the closure until it needs the host again (-> n_need > 0:
// uz_join_needs) or is finished (n_need == 0: the kept list, its offsets and totals are in place)
void uz_join_run(uz_ctx *c, uz_ctx::WalkSlot &w, const JoinPlanHost &P, const uz_walk_desc *x, int64_t n_x, const uint8_t *xaux, int64_t xaux_bytes,
                 const int32_t *look_tid, int64_t n_look, const int32_t *need_jtask);
void uz_join_needs(uz_ctx *c, uz_ctx::WalkSlot &w, uz_need_rec *out);
// debug / parity: the kept records of a finished join in output order
void uz_join_fetch(uz_ctx *c, uz_ctx::WalkSlot &w, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases, uz_kept_rec *kept);
// the read names of name ids (a table built by uz_reads_from_walk with names): lengths, then bytes
void uz_launch_name_lens(uz_ctx *c, hipStream_t st, int64_t n_ids, const uint32_t *ids, const uint32_t *name_rec, const uz_kept_rec *kept, int64_t n_recs, int64_t names_bytes,
                         uint32_t *len);
void uz_launch_name_gather(uz_ctx *c, hipStream_t st, int64_t n_ids, const uint32_t *ids, const uint32_t *name_rec, const uz_kept_rec *kept, const uint8_t *names,
                           const uint32_t *len, const uint32_t *off, uint8_t *out);
void uz_scan_u32(uz_ctx *c, hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n, DevBuf<uint8_t> &tmp); // exclusive sum
void uz_launch_bam_extract(uz_ctx *c, hipStream_t st, int64_t n, const uint8_t *buf, int64_t buf_bytes, const uint8_t *aux, int64_t aux_bytes, const uz_kept_rec *kept,
                           int thr, int32_t *start, int32_t *tlen, int32_t *mate, uint32_t *qname, uint16_t *flag, uint16_t *l_seq, uint16_t *n_cigar, uint8_t *mapq,
                           uint8_t *aux_col, uint32_t *cigar, uint8_t *seq4, uint32_t *plane, int32_t *err, uint8_t *names, int64_t n_cigar_total, int64_t n_row_units,
                           int64_t n_seq_units, int64_t names_bytes);
// CRC-32 of inflated blocks against their BGZF footers (k_inflate.hip): device pointers; *err: 0, or 1 + a block whose checksum differs
void uz_launch_crc32(uz_ctx *c, hipStream_t st, int64_t n_blocks, const uint8_t *out, const int64_t *out_off, const uint32_t *want, int32_t *err);
int uz_phase_votes_impl(uz_ctx *c, int64_t *vote_off, int32_t *vote_val);
int uz_phase_groups_impl(uz_ctx *c, int64_t *grp_off, int32_t *grp_q);
void uz_phase_state_free(uz_ctx *c);
