// k_bamwalk.hip -- the record walk of a staged batch on the device (include/uz_bamwalk.h).
//
// What it replaces: the host's walk of the inflated BGZF blocks (csrc/io_stage.cpp: walk_task -- what `bamfile.fetch(chrom, lo, hi)` iterates for
// the reference per DNM and het site, read_collector.py:385, :167) and the copy of every record it keeps into the link form.  The blocks of a
// batch are inflated in HBM by k_bgzf_inflate and stay there.
//
//   k_bam_walk         one wavefront per walk task.  A BAM stream has no record index: every record starts where the one before ends, so the
//                      chain of block_size fields is inherently serial -- but only per task, and a batch has hundreds to thousands of tasks.
//                      The wave stages the stream through LDS in 16 KiB windows (64 lanes x 16-byte loads, coalesced); lane 0 follows the
//                      chain inside the window at LDS latency (~50 records per window) and lists the record starts; then all 64 lanes take
//                      one record each: virtual offset (block table), refID / pos against the task's stop rules, end from the CIGAR, the
//                      reach intervals, the fetches (binary search), the two name hashes.  The host's walk is sequential (`break` / `stop`
//                      at the first record that says so): the lanes vote, the first lane that ends the walk wins, later lanes are dropped.
//                      ONE pass: every task writes its 64-byte descriptors in file order into a slice of the output sized for the most records
//                      its bytes can hold (a record is at least 36 bytes) and counts them; every inflated byte is read once through the windows
//                      (the lanes' own reads hit the same lines in L2).  Then the mate-candidate filter: k_tab_insert, k_desc_filter.
//   k_bam_extract      one lane per KEPT record (the host's answer: uz_kept_rec): the plain columns of uz_reads_packed_view from the record's
//                      bytes -- start, tlen, flag, l_seq, n_cigar, mapq, the aux bits (mate on the same reference, an SA tag, no CIGAR / SEQ /
//                      QUAL), every CIGAR word, the bases as BAM packs them (four bits: the device's own row format), the one-bit plane
//                      "quality below the threshold" -- at the offsets the header build (k_reads.hip: k_pack_rec) will derive for itself.
//
// Parity: tests/test_bamwalk_gpu.py holds the descriptors against the host's twin (uz_stage_walk_host) and the table against the one the host
// route stages for the same fetches (uz_reads_headers, then the read stage's results).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "uz_bamwalk.h"
#include "uz_ctx.hpp"

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// io_stage.cpp: hash_name (FNV-1a with a final mix)
__device__ __forceinline__ uint64_t name_hash1(const uint8_t *s, uint32_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (uint32_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ULL; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ULL; h ^= h >> 29;
    return h;
}

// end of the alignment as htslib's bam_endpos gives it (io_stage.cpp: endpos_of)
__device__ __forceinline__ int32_t endpos_of(const uint8_t *p, int32_t pos, uint32_t fl, uint32_t ncig, uint32_t l_name) {
    if ((fl & 4u) || ncig == 0) return pos + 1;
    int64_t rl = 0;
    const uint8_t *q = p + 32 + l_name;
    for (uint32_t k = 0; k < ncig; k++) {
        const uint32_t v = ld32(q + 4 * k), op = v & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4;
    }
    return (int32_t)(pos + (rl > 0 ? rl : 1));
}

constexpr int WIN = 16384;     // bytes of the stream staged in LDS at a time
constexpr int WIN_RECS = 512;  // record starts listed per window (a record is at least 36 bytes: at most 456 fit)
enum { K_SKIP = 0, K_EMIT = 1, K_BREAK = 2, K_STOP = 3, K_BAD = 4 };

struct WalkArgs {
    const uint8_t *buf;       // the inflated blocks, back to back (padded by WIN + 64 bytes)
    const int64_t *blk_at;    // [n_blocks + 1]: where block k's bytes start
    const int64_t *blk_coff;  // [n_blocks]: its offset in the file
    const int32_t *task;      // [UZ_WALK_TASK_COLS n_tasks]
    const int64_t *span;      // [UZ_WALK_SPAN_COLS n_spans]
    const int32_t *reach;     // [2 n_reach]
    const int32_t *fetch;     // [3 n_fetch]
    int64_t *count;           // [n_tasks]  out: descriptors written
    const int64_t *first;     // [n_tasks + 1]  where every task's slice of `out` starts (worst-case sizes: abi.hip uz_bam_walk)
    int64_t *walked;          // [n_tasks]
    int32_t *flags;           // [n_tasks]
    uz_walk_desc *out;
    // the names of the direct records of a HOST task (all its sub-tasks), as a hash set (open addressing over a slice of `tab`): the walk
    // counts them, k_tab_insert inserts them; k_desc_filter then drops every descriptor that is neither direct nor shares a name hash with a
    // direct one -- exactly the records the host's walk drops first (io_stage.cpp finish_task: "mate candidates")
    int64_t *n_direct;        // [n_tasks]  out: direct records of the task
    const int64_t *tab_first; // [n_tasks + 1]  (the kernels behind the walk): the slices, a power of two of slots for the first sub-task of a host task
    unsigned long long *tab;
};
__device__ __forceinline__ unsigned long long tab_key(uint64_t h1) { return (unsigned long long)h1 | 1ULL; } // (never 0 = empty; merging two hashes keeps a record too many, never one too few)
__device__ __forceinline__ size_t tab_slot(unsigned long long key, int64_t size) { return (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & (size_t)(size - 1); }

__global__ __launch_bounds__(64) void k_bam_walk(WalkArgs a) {
    constexpr bool FILL = true;
    __shared__ uint4 win[WIN / 16];
    __shared__ uint16_t offs[WIN_RECS];
    __shared__ int s_n, s_state;
    __shared__ long long s_next;
    // the task's fetches and reach intervals, searched once per record: in LDS when they fit (they do unless a task holds thousands of fetches)
    constexpr int FCAP = 1024, RCAP = 256;
    __shared__ int32_t f_lo[FCAP], f_hi[FCAP], r_a[RCAP], r_b[RCAP];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int32_t *tc = a.task + UZ_WALK_TASK_COLS * (size_t)t;
    const int32_t tid = tc[0], tb = tc[1], sp0 = tc[2], sp1 = tc[3], r0 = tc[4], r1 = tc[5], f0 = tc[6], f1 = tc[7], max_len = tc[8];
    const uint8_t *win8 = reinterpret_cast<const uint8_t *>(win);
    const bool f_lds = f1 - f0 <= FCAP, r_lds = r1 - r0 <= RCAP;
    if (f_lds) for (int i = lane; i < f1 - f0; i += 64) { f_lo[i] = a.fetch[3 * (f0 + i)]; f_hi[i] = a.fetch[3 * (f0 + i) + 1]; }
    if (r_lds) for (int i = lane; i < r1 - r0; i += 64) { r_a[i] = a.reach[2 * (r0 + i)]; r_b[i] = a.reach[2 * (r0 + i) + 1]; }
    __syncthreads();
    int64_t n_out = 0, walked = 0, n_dir = 0;
    int flag = 0;
    bool stop = false;
    for (int sp = sp0; sp < sp1 && !stop && !flag; sp++) {
        const int64_t *sc = a.span + UZ_WALK_SPAN_COLS * (size_t)sp;
        const uint64_t span_end = (uint64_t)sc[1];
        int64_t cur = sc[2];
        const int64_t bend = sc[3], blk0 = sc[4], blk1 = sc[5];
        if (blk0 >= blk1) { stop = true; break; } // no block at the span's start: the end of the file (the host's walk stops there too)
        bool span_done = false;
        int64_t bi_cur = blk0; // the block that holds `cur` (the cursor only moves forward: a step or none per window)
        while (!span_done) {
            while (bi_cur + 1 < blk1 && a.blk_at[bi_cur + 1] <= cur) bi_cur++;
            const int64_t w0 = cur & ~(int64_t)15;
#pragma unroll
            for (int it = 0; it < WIN / 16 / 64; it++) {
                const int idx = it * 64 + lane;
                win[idx] = *reinterpret_cast<const uint4 *>(a.buf + w0 + 16 * (int64_t)idx);
            }
            __syncthreads();
            if (lane == 0) { // the chain of block_size fields inside the window
                int k = 0, state = 0; // 0: the window is used up; 1: the gathered bytes end here; 2: ... in the middle of a record; 3: not a record
                int64_t c = cur;
                const int64_t wend = w0 + WIN;
                while (k < WIN_RECS) {
                    if (c + 4 > bend) { state = c >= bend ? 1 : 2; break; }
                    if (c + 4 > wend) break;
                    const int32_t bs = (int32_t)ld32(win8 + (c - w0));
                    if (bs < 32) { state = 3; break; }
                    if (c + 4 + (int64_t)bs > bend) { state = 2; break; }
                    offs[k++] = (uint16_t)(c - w0);
                    c += 4 + (int64_t)bs;
                }
                s_n = k; s_state = state; s_next = c;
            }
            __syncthreads();
            const int n = s_n, state = s_state;
            const int64_t next = s_next;
            int ended = 0; // the kind of the record that ended the walk of this span, 0: none did
            for (int b0 = 0; b0 < n && !ended; b0 += 64) {
                const int j = b0 + lane;
                const bool valid = j < n;
                int kind = K_SKIP;
                bool counted = false;
                int64_t c = 0;
                uint64_t voff = 0;
                int32_t pos = 0, end = 0;
                uint32_t l_name = 0, ncig = 0, fl = 0, lseq = 0, bs = 0;
                const uint8_t *p = a.buf;
                if (valid) {
                    c = w0 + offs[j];
                    bs = ld32(win8 + offs[j]);
                    // the record's own bytes: from the window in LDS when it lies inside it whole (all but the last record or two of a window),
                    // else from where it lies -- every field below is read byte by byte (BAM fields are unaligned)
                    p = (uint32_t)offs[j] + 4u + bs <= (uint32_t)WIN ? win8 + offs[j] + 4 : a.buf + c + 4;
                    // the block that holds the record's first byte: the first whose end lies behind it (an empty block holds nothing)
                    int64_t lo = bi_cur; // (a window reaches into the next block or two at most)
                    while (lo + 1 < blk1 && a.blk_at[lo + 1] <= c) lo++;
                    voff = ((uint64_t)a.blk_coff[lo] << 16) | (uint64_t)(c - a.blk_at[lo]);
                    if (voff >= span_end) kind = K_BREAK;
                    else {
                        const int32_t rt = (int32_t)ld32(p);
                        pos = (int32_t)ld32(p + 4);
                        if (rt != tid) kind = (rt < 0 || rt > tid) ? K_STOP : K_SKIP;
                        else if (pos >= tb) kind = K_STOP;
                        else {
                            l_name = p[8]; ncig = ld16(p + 12); fl = ld16(p + 14);
                            const int32_t ls = (int32_t)ld32(p + 16);
                            lseq = (uint32_t)ls;
                            if (ls < 0 || ls > 0xFFFF || l_name < 1 || 32 + (uint64_t)l_name + 4 * (uint64_t)ncig > (uint64_t)bs) kind = K_BAD;
                            else {
                                end = endpos_of(p, pos, fl, ncig, l_name);
                                counted = true;
                                // between two reach intervals nothing can be fetched (and a mate position there goes through the index)
                                bool gap;
                                if (r_lds) {
                                    int ri = 0;
                                    while (ri < r1 - r0 && pos >= r_b[ri]) ri++;
                                    gap = ri < r1 - r0 && end <= r_a[ri];
                                } else {
                                    int ri = r0;
                                    while (ri < r1 && pos >= a.reach[2 * ri + 1]) ri++;
                                    gap = ri < r1 && end <= a.reach[2 * ri];
                                }
                                if (gap) kind = K_SKIP;
                                else if (32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs) kind = K_BAD; // (the host's extract)
                                else kind = K_EMIT;
                            }
                        }
                    }
                }
                const unsigned long long term = __ballot(valid && kind >= K_BREAK);
                const int first = term ? __ffsll((long long)term) - 1 : 64;
                const bool live = valid && lane < first;
                walked += __popcll(__ballot(live && counted));
                const bool emit = live && kind == K_EMIT;
                const unsigned long long m = __ballot(emit);
                // does a fetch return it?  (start < hi and end > lo: read_collector.py:385, :167)
                bool direct = false;
                if (emit) {
                    const int64_t key = (int64_t)pos - max_len;
                    if (f_lds) {
                        int lo = 0, hi = f1 - f0;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)f_lo[mid] < key) lo = mid + 1; else hi = mid; }
                        for (; lo < f1 - f0 && f_lo[lo] < end; lo++)
                            if (f_hi[lo] > pos) { direct = true; break; }
                    } else {
                        int lo = f0, hi = f1;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)a.fetch[3 * mid] < key) lo = mid + 1; else hi = mid; }
                        for (; lo < f1 && a.fetch[3 * lo] < end; lo++)
                            if (a.fetch[3 * lo + 1] > pos) { direct = true; break; }
                    }
                }
                n_dir += __popcll(__ballot(direct));
                if (FILL && emit) {
                    const int rank = __popcll(m & ((1ULL << lane) - 1ULL));
                    uz_walk_desc d;
                    d.voff = voff; d.src = (uint64_t)(c + 4);
                    d.h1 = name_hash1(p + 32, l_name - 1);

                    d.pos = pos; d.end = end; d.tlen = (int32_t)ld32(p + 28); d.mpos = (int32_t)ld32(p + 24); d.mtid = (int32_t)ld32(p + 20);
                    d.h2 = uz_name_hash2(p + 32, l_name - 1);
                    d.task = (uint32_t)t;
                    d.flag = (uint16_t)fl; d.l_seq = (uint16_t)lseq; d.n_cigar = (uint16_t)ncig;
                    d.mapq = p[9]; d.l_name = (uint8_t)(l_name - 1); d.direct = direct ? 1 : 0; d.pad8 = 0; d.pad16 = 0;
                    a.out[a.first[t] + n_out + rank] = d;
                }
                n_out += __popcll(m);
                if (first < 64) ended = __shfl(kind, first, 64);
            }
            if (ended == K_BREAK) span_done = true;
            else if (ended == K_STOP) { stop = true; span_done = true; }
            else if (ended == K_BAD || state == 3) { flag |= UZ_WALK_TASK_BAD; span_done = true; }
            else if (state == 1 || state == 2) { flag |= UZ_WALK_TASK_INCOMPLETE; span_done = true; }
            cur = next;
            __syncthreads(); // the window is read to the end before the next one lands
        }
    }
    if (lane == 0) { // (a flagged task goes back to the host: what it wrote into its slice is never looked at)
        a.count[t] = flag ? 0 : n_out; a.n_direct[t] = flag ? 0 : n_dir; a.walked[t] = walked; a.flags[t] = flag;
    }
}

// The hash set of mate candidates belongs to the HOST task (the stage's own task: finish_task in io_stage.cpp keeps the records that share a
// name with a direct record of the task), and a host task is walked as several sub-tasks here (column 9 of a walk task names its host task;
// the sub-tasks of one host task stand next to each other): all of them insert into and look up in ONE set, the slice of the first of
// them.  (Round 4 gave every sub-task a set of its own: a mate that lay in another sub-task's reach than its partner -- a long insert, a
// discordant pair -- was dropped here and never looked up through the index.)
__device__ __forceinline__ int walk_leader(const int32_t *__restrict__ task, int t) {
    const int32_t host = task[UZ_WALK_TASK_COLS * (size_t)t + 9];
    while (t > 0 && task[UZ_WALK_TASK_COLS * (size_t)(t - 1) + 9] == host) t--;
    return t;
}
// the names of every task's direct records into its host task's hash set (one wavefront per task over its descriptors)
__global__ __launch_bounds__(64) void k_tab_insert(const uz_walk_desc *__restrict__ in, const int64_t *__restrict__ first, const int64_t *__restrict__ count,
                                                   const int32_t *__restrict__ task, const int64_t *__restrict__ tab_first, unsigned long long *tab) {
    const int t = blockIdx.x, L = walk_leader(task, t);
    const int64_t a = first[t], b = a + count[t], size = tab_first[L + 1] - tab_first[L];
    unsigned long long *tb = tab + tab_first[L];
    for (int64_t i = a + threadIdx.x; i < b; i += 64) {
        if (!in[i].direct) continue;
        const unsigned long long key = tab_key(in[i].h1);
        for (size_t sl = tab_slot(key, size);; sl = (sl + 1) & (size_t)(size - 1)) { // (at most half full: the walk counted the direct records)
            const unsigned long long was = atomicCAS(&tb[sl], 0ULL, key);
            if (was == 0ULL || was == key) break;
        }
    }
}

// counts -> offsets (one workgroup); POW2: the hash sets -- the first sub-task of a host task stands for a set of the next power of two >= 2 c + 2
// slots, c the direct records of all its sub-tasks; the others for none (walk_leader)
template <bool POW2>
__global__ __launch_bounds__(256) void k_walk_scan(int n, const int64_t *count, int64_t *first, const int32_t *__restrict__ task) {
    __shared__ long long part[256];
    const int t = threadIdx.x;
    const int chunk = (n + 255) / 256;
    const int lo = min(t * chunk, n), hi = min(lo + chunk, n);
    auto val = [&](int i) -> long long {
        long long c = count[i];
        if (!POW2) return c;
        const int32_t host = task[UZ_WALK_TASK_COLS * (size_t)i + 9];
        if (i > 0 && task[UZ_WALK_TASK_COLS * (size_t)(i - 1) + 9] == host) return 0;
        for (int j = i + 1; j < n && task[UZ_WALK_TASK_COLS * (size_t)j + 9] == host; j++) c += count[j];
        long long sz = 4;
        while (sz < 2 * c + 2) sz <<= 1;
        return sz;
    };
    long long s = 0;
    for (int i = lo; i < hi; i++) s += val(i);
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int k = 0; k < 256; k++) { const long long x = part[k]; part[k] = run; run += x; }
        first[n] = run;
    }
    __syncthreads();
    long long run = part[t];
    for (int i = lo; i < hi; i++) { first[i] = run; run += val(i); }
}

// the descriptors of a task that are direct or share a name hash with a direct one, in order (one wavefront per task)
template <bool FILL>
__global__ __launch_bounds__(64) void k_desc_filter(const uz_walk_desc *__restrict__ in, const int64_t *__restrict__ first, const int64_t *__restrict__ count,
                                                    const int32_t *__restrict__ task, const int64_t *__restrict__ tab_first, const unsigned long long *__restrict__ tab,
                                                    int64_t *kcount, const int64_t *__restrict__ kfirst, uz_walk_desc *out) {
    const int t = blockIdx.x, lane = threadIdx.x, L = walk_leader(task, t);
    const int64_t a = first[t], b = a + count[t], size = tab_first[L + 1] - tab_first[L];
    const unsigned long long *tb = tab + tab_first[L];
    int64_t n_out = 0;
    for (int64_t i0 = a; i0 < b; i0 += 64) {
        const int64_t i = i0 + lane;
        bool keep = false;
        if (i < b) {
            keep = in[i].direct != 0;
            if (!keep) {
                const unsigned long long key = tab_key(in[i].h1);
                for (size_t sl = tab_slot(key, size);; sl = (sl + 1) & (size_t)(size - 1)) {
                    const unsigned long long x = tb[sl];
                    if (x == key) { keep = true; break; }
                    if (x == 0ULL) break;
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (FILL && keep) out[kfirst[t] + n_out + __popcll(m & ((1ULL << lane) - 1ULL))] = in[i];
        n_out += __popcll(m);
    }
    if (!FILL && lane == 0) kcount[t] = n_out;
}

// io_stage.cpp: has_sa_tag
__device__ bool has_sa_tag(const uint8_t *p, const uint8_t *end) {
    while (p + 3 <= end) {
        const uint8_t x = p[0], y = p[1], typ = p[2];
        p += 3;
        if (x == 'S' && y == 'A') return true;
        int sz = 0;
        switch (typ) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            while (p < end && *p) p++;
            if (p >= end) return false;
            p++;
            continue;
        }
        case 'B': {
            if (p + 5 > end) return false;
            int es = 0;
            switch (p[0]) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return false; }
            const int32_t cnt = (int32_t)ld32(p + 1);
            if (cnt < 0) return false;
            p += 5 + (size_t)cnt * (size_t)es;
            continue;
        }
        default: return false;
        }
        p += sz;
    }
    return false;
}

struct ExtractArgs {
    const uint8_t *buf, *aux;
    const uz_kept_rec *kept;
    int64_t n, buf_bytes, aux_bytes;
    int32_t thr;
    int32_t *start, *tlen, *mate;
    uint32_t *qname;
    uint16_t *flag, *l_seq, *n_cigar;
    uint8_t *mapq, *aux_col;
    uint32_t *cigar;
    uint8_t *seq4;
    uint32_t *plane;
    int32_t *err;
    uint8_t *names; // null: the names are not asked for
    // the sizes of the stores the kept list's offsets point into: a record that would be written beyond them is refused (the list is the caller's)
    int64_t n_cigar_total, n_row_units, n_seq_units, names_bytes;
};

// Eight lanes per record, 32 records per workgroup.  The record's bytes are staged in LDS first -- the eight lanes fetch consecutive 16-byte
// pieces, so a wave's load touches eight records' lines instead of 64 lanes' worth of scattered bytes -- and every byte-granular read below
// (BAM fields are unaligned) is an LDS read; a record longer than the stage is read from where it lies.
constexpr int EX_CAP = 768;
__global__ __launch_bounds__(256) void k_bam_extract(ExtractArgs a) {
    __shared__ uint4 stage[32][EX_CAP / 16];
    const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int64_t i = (int64_t)blockIdx.x * 32 + g;
    bool ok = i < a.n;
    uz_kept_rec k = {};
    const uint8_t *gp = a.buf;
    int64_t lim = 0;
    uint64_t off = 0;
    uint32_t bs = 0;
    if (ok) {
        k = a.kept[i];
        const bool in_aux = (k.src & UZ_WALK_SRC_AUX) != 0;
        off = k.src & ~UZ_WALK_SRC_AUX;
        lim = in_aux ? a.aux_bytes : a.buf_bytes;
        if (off < 4 || (int64_t)off + 32 > lim) { *a.err = 1; ok = false; }
        else {
            gp = (in_aux ? a.aux : a.buf) + off;
            bs = ld32(gp - 4);
            if (bs < 32 || (int64_t)off + (int64_t)bs > lim) { *a.err = 1; ok = false; }
        }
    }
    const uint8_t *src = gp - 4;
    const uint32_t d = ok ? (uint32_t)((uintptr_t)src & 15u) : 0u;
    const uint32_t total = d + 4 + bs;
    const bool in_lds = ok && total <= (uint32_t)EX_CAP;
    if (in_lds)
        for (uint32_t c = (uint32_t)l; 16 * c < total; c += 8) stage[g][c] = *reinterpret_cast<const uint4 *>(src - d + 16 * (size_t)c); // (the buffers are padded: the last piece may reach past the record)
    __syncthreads();
    if (!ok) return;
    const uint8_t *p = in_lds ? reinterpret_cast<const uint8_t *>(stage[g]) + d + 4 : gp;
    const uint32_t l_name = p[8], ncig = ld16(p + 12), fl = ld16(p + 14), L = ld32(p + 16);
    if (L > 0xFFFFu || l_name < 1 || 32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)L + 1) / 2 + (uint64_t)L > (uint64_t)bs) { *a.err = 1; return; }
    {
        const int64_t u = (int64_t)UZ_ROW_UNITS(L);
        if ((int64_t)k.cig_off + (int64_t)ncig > a.n_cigar_total || (int64_t)k.unit_off + u > a.n_row_units ||
            (k.seq_off != UZ_KEPT_NO_SEQ && (int64_t)k.seq_off + u > a.n_seq_units) || (a.names && (int64_t)k.name_off + (int64_t)l_name - 1 > a.names_bytes)) { *a.err = 2; return; }
    }
    const uint8_t *q = p + 32 + l_name;
    const uint8_t *sq = q + 4 * (size_t)ncig;
    const uint8_t *ql = sq + ((size_t)L + 1) / 2;
    const bool bases = k.seq_off != UZ_KEPT_NO_SEQ;
    const bool noqual = L > 0 && ql[0] == 0xFF;
    if (l == 0) {
        uint32_t ax = 0;
        if ((int32_t)ld32(p + 20) == (int32_t)ld32(p)) ax |= UZ_AUX_MATE_SAME_TID;
        if (has_sa_tag(ql + L, p + bs)) ax |= UZ_AUX_HAS_SA;
        if (ncig == 0 || L == 0 || noqual) ax |= UZ_AUX_DECODE_BAD;
        if (!bases) ax |= UZ_AUX_NO_SEQ;
        a.start[i] = (int32_t)ld32(p + 4);
        a.tlen[i] = (int32_t)ld32(p + 28);
        a.mate[i] = k.mate;
        a.qname[i] = k.qname;
        a.flag[i] = (uint16_t)fl; a.l_seq[i] = (uint16_t)L; a.n_cigar[i] = (uint16_t)ncig;
        a.mapq[i] = p[9]; a.aux_col[i] = (uint8_t)ax;
    }
    for (uint32_t w = (uint32_t)l; w < ncig; w += 8) a.cigar[(size_t)k.cig_off + w] = ld32(q + 4 * w);
    if (a.names)
        for (uint32_t b = (uint32_t)l; b + 1 < l_name; b += 8) a.names[(size_t)k.name_off + b] = p[32 + b];
    const uint32_t units = UZ_ROW_UNITS(L);
    if (bases) { // BAM's own nibbles; the pad nibble of an odd length and the rest of the last unit are zero
        const uint32_t nb = (L + 1) / 2;
        for (uint32_t u = (uint32_t)l; u < units; u += 8) {
            uint32_t w4[4];
#pragma unroll
            for (uint32_t qd = 0; qd < 4; qd++) {
                uint32_t w = 0;
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    const uint32_t b = 16 * u + 4 * qd + e;
                    uint32_t v = b < nb ? (uint32_t)sq[b] : 0u;
                    if (b + 1 == nb && (L & 1u)) v &= 0xF0u;
                    w |= v << (8 * e);
                }
                w4[qd] = w;
            }
            *reinterpret_cast<uint4 *>(a.seq4 + ((size_t)k.seq_off + u) * UZ_SEQ4_UNIT_BYTES) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    // the plane "quality below the threshold" (a record without qualities decodes to zeros: every base is below a positive threshold)
    for (uint32_t u = (uint32_t)l; u < units; u += 8) {
        uint32_t w = 0;
        const uint32_t k0 = 32 * u, k1 = min(L, k0 + 32);
        for (uint32_t b = k0; b < k1; b++) {
            const bool low = noqual ? a.thr > 0 : (int)ql[b] < a.thr;
            w |= (low ? 1u : 0u) << (b - k0);
        }
        a.plane[(size_t)k.unit_off + u] = w;
    }
}

} // namespace

// ---- launchers (abi.hip: uz_bam_walk / uz_bam_walk_fetch / uz_reads_from_bam)
// the walk: every record inside a reach interval as a descriptor into its task's slice of `out` (first[t]: where the slice starts -- sized by
// the caller for the most records the task's bytes can hold), count / n_direct / walked / flags per task; tab_first: the hash sets' slices
void uz_launch_bam_walk(uz_ctx *c, hipStream_t st, int n_tasks, const uint8_t *buf, const int64_t *blk_at, const int64_t *blk_coff, const int32_t *task,
                        const int64_t *span, const int32_t *reach, const int32_t *fetch, int64_t *count, const int64_t *first, int64_t *walked, int32_t *flags,
                        uz_walk_desc *out, int64_t *n_direct, int64_t *tab_first) {
    if (n_tasks <= 0) return;
    WalkArgs a{buf, blk_at, blk_coff, task, span, reach, fetch, count, first, walked, flags, out, n_direct, nullptr, nullptr};
    hipLaunchKernelGGL(k_bam_walk, dim3((unsigned)n_tasks), dim3(64), 0, st, a);
    hipLaunchKernelGGL((k_walk_scan<true>), dim3(1), dim3(256), 0, st, n_tasks, (const int64_t *)n_direct, tab_first, task);
    UZ_HIP(hipGetLastError());
}
// stage 0: the hash sets filled, the kept descriptors counted (kcount, kfirst); stage 1: the kept descriptors into `out`
void uz_launch_desc_filter(uz_ctx *c, hipStream_t st, bool fill, int n_tasks, const uz_walk_desc *in, const int64_t *first, const int64_t *count,
                           const int32_t *task, const int64_t *tab_first, unsigned long long *tab, int64_t *kcount, int64_t *kfirst, uz_walk_desc *out) {
    if (n_tasks <= 0) return;
    if (!fill) {
        hipLaunchKernelGGL(k_tab_insert, dim3((unsigned)n_tasks), dim3(64), 0, st, in, first, count, task, tab_first, tab);
        hipLaunchKernelGGL((k_desc_filter<false>), dim3((unsigned)n_tasks), dim3(64), 0, st, in, first, count, task, tab_first, (const unsigned long long *)tab, kcount,
                           (const int64_t *)kfirst, out);
        hipLaunchKernelGGL((k_walk_scan<false>), dim3(1), dim3(256), 0, st, n_tasks, (const int64_t *)kcount, kfirst, task);
    } else
        hipLaunchKernelGGL((k_desc_filter<true>), dim3((unsigned)n_tasks), dim3(64), 0, st, in, first, count, task, tab_first, (const unsigned long long *)tab, kcount,
                           (const int64_t *)kfirst, out);
    UZ_HIP(hipGetLastError());
}
size_t uz_bam_walk_pad() { return (size_t)WIN + 64; }

void uz_launch_bam_extract(uz_ctx *c, hipStream_t st, int64_t n, const uint8_t *buf, int64_t buf_bytes, const uint8_t *aux, int64_t aux_bytes, const uz_kept_rec *kept,
                           int thr, int32_t *start, int32_t *tlen, int32_t *mate, uint32_t *qname, uint16_t *flag, uint16_t *l_seq, uint16_t *n_cigar, uint8_t *mapq,
                           uint8_t *aux_col, uint32_t *cigar, uint8_t *seq4, uint32_t *plane, int32_t *err, uint8_t *names, int64_t n_cigar_total, int64_t n_row_units,
                           int64_t n_seq_units, int64_t names_bytes) {
    if (n <= 0) return;
    ExtractArgs a{buf, aux, kept, n, buf_bytes, aux_bytes, thr, start, tlen, mate, qname, flag, l_seq, n_cigar, mapq, aux_col, cigar, seq4, plane, err, names,
                  n_cigar_total, n_row_units, n_seq_units, names_bytes};
    hipLaunchKernelGGL(k_bam_extract, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, a);
    UZ_HIP(hipGetLastError());
}
