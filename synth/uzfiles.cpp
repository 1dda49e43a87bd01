// uzfiles.cpp -- the benchmark-scale generator (uzsynth.h) written out as REAL FILES: a coordinate-sorted BAM with its
// BAI index and a BGZF-compressed sites VCF with its tabix index, the way samtools / bgzip / tabix lay them out (SAM
// spec sections 4 and 5; the tabix paper's index format).  TEST / BENCH INFRASTRUCTURE (g++, zlib, threads): what the
// product's readers (libunfazed_io: uz_bam_decode_regions, uz_bam_stage_*, uz_vcf_decode_regions) are timed on at the
// sizes a Python writer cannot reach.  The records are those of uzs_gen_reads_cpu for the same (cfg, clusters): a file
// decoded back gives the table the oracle is run on.
//
// Layout: workers own contiguous ranges of clusters (BAM) / sites (VCF); each deflates its own BGZF blocks and keeps a
// partial index in offsets relative to its own first block; the parts are joined by adding the workers' file offsets.
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

extern "C" {
#include "uzsynth.h"
}

namespace {

const size_t BGZF_MAX_IN = 0xff00; // htslib's BGZF_BLOCK_SIZE: uncompressed bytes per block

struct Chunk { uint64_t beg, end; };

inline uint32_t reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// bins + linear index of one reference, offsets relative to the owner's first block
struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::map<int64_t, uint64_t> linear; // 16 kb window -> smallest offset of a record overlapping it
    void add(int64_t beg, int64_t end, uint64_t v0, uint64_t v1) {
        if (end <= beg) end = beg + 1;
        auto &ch = bins[reg2bin(beg, end)];
        if (!ch.empty() && ch.back().end == v0) ch.back().end = v1;
        else ch.push_back(Chunk{v0, v1});
        for (int64_t w = beg >> 14; w <= (end - 1) >> 14; w++)
            if (!linear.count(w)) linear[w] = v0;
    }
};

// libdeflate when the system has it (three functions of its stable ABI, no headers needed): several times faster than zlib at the
// same level; the blocks it writes are plain deflate either way
struct LibDeflateC {
    void *(*alloc)(int) = nullptr;
    size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflateC() {
        const char *e = getenv("UZ_INFLATE");
        if (e && strcmp(e, "zlib") == 0) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && compress && release && crc;
    }
};
const LibDeflateC &libdeflate_c() { static LibDeflateC L; return L; }

// one worker's output: BGZF blocks back to back + its partial index
struct Part {
    std::vector<uint8_t> cdata;
    std::vector<uint8_t> cur; // the block being filled
    std::map<int32_t, RefIndex> idx;
    int64_t n_records = 0, n_raw = 0, n_blocks = 0;
    // the record added last: when it ended exactly at the end of its block, its offset is rewritten to the start of the block
    // that follows (the offset a reader's tell gives there), here or -- for the part's last record -- when the parts are joined
    int32_t last_tid = -1;
    uint32_t last_bin = 0;
    uint64_t last_v1 = 0;
    bool tail_at_block_end = false;
    z_stream z;
    void *ld = nullptr;
    int level = 6;
    bool z_ok = false;
    std::string err;

    void init(int lvl) {
        level = lvl;
        if (libdeflate_c().ok) ld = libdeflate_c().alloc(lvl);
        memset(&z, 0, sizeof(z));
        z_ok = deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        if (!z_ok) err = "deflateInit2 failed";
        cur.reserve(BGZF_MAX_IN);
    }
    ~Part() {
        if (z_ok) deflateEnd(&z);
        if (ld) libdeflate_c().release(ld);
    }
    uint64_t tell() const { return ((uint64_t)cdata.size() << 16) | (uint64_t)cur.size(); }
    void flush() {
        if (cur.empty()) return;
        emit(cur.data(), cur.size());
        cur.clear();
    }
    // one BGZF block from `n` <= BGZF_MAX_IN raw bytes (n = 0: the end-of-file marker)
    void emit(const uint8_t *raw, size_t n) {
        const size_t at = cdata.size();
        cdata.resize(at + 18 + deflateBound(&z, (uLong)n) + 8);
        uint8_t *h = cdata.data() + at;
        static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(h, head, 16);
        size_t clen = 0;
        if (ld && n) clen = libdeflate_c().compress(ld, raw, n, h + 18, cdata.size() - at - 18 - 8);
        if (!clen) {
            deflateReset(&z);
            z.next_in = const_cast<Bytef *>(raw);
            z.avail_in = (uInt)n;
            z.next_out = h + 18;
            z.avail_out = (uInt)(cdata.size() - at - 18 - 8);
            if (deflate(&z, Z_FINISH) != Z_STREAM_END) { err = "deflate failed"; return; }
            clen = (size_t)z.total_out;
        }
        const size_t blen = 18 + clen + 8;
        if (blen > 0x10000) { err = "BGZF block does not fit 64 KiB"; return; }
        const uint16_t bsize = (uint16_t)(blen - 1);
        memcpy(h + 16, &bsize, 2);
        const uint32_t crc = ld ? libdeflate_c().crc(0, raw, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw, (uInt)n), isize = (uint32_t)n;
        memcpy(h + 18 + clen, &crc, 4);
        memcpy(h + 18 + clen + 4, &isize, 4);
        cdata.resize(at + blen);
        n_blocks++;
        n_raw += (int64_t)n;
    }
    // one record / line [beg, end) of reference tid: written (bgzf_flush_try + bgzf_write) and indexed
    void add_record(int32_t tid, int64_t beg, int64_t end, const uint8_t *p, size_t n) {
        if (end <= beg) end = beg + 1;
        const bool flushed = cur.size() + n > BGZF_MAX_IN && !cur.empty();
        const uint64_t before = tell();
        if (flushed) flush();
        const uint64_t v0 = tell();
        if (flushed && last_tid >= 0 && last_v1 == before) { // the record before ended with its block
            auto &ch = idx[last_tid].bins[last_bin];
            if (!ch.empty() && ch.back().end == last_v1) ch.back().end = v0;
        }
        put(p, n);
        const uint64_t v1 = tell();
        idx[tid].add(beg, end, v0, v1);
        last_tid = tid; last_bin = reg2bin(beg, end); last_v1 = v1;
        n_records++;
    }
    void finish() {
        tail_at_block_end = last_tid >= 0 && !cur.empty() && last_v1 == tell();
        flush();
    }
    // bgzf_flush_try + bgzf_write: a record / line that does not fit the rest of the block starts a new one
    void put(const uint8_t *p, size_t n) {
        if (cur.size() + n > BGZF_MAX_IN) flush();
        while (n > BGZF_MAX_IN) { emit(p, BGZF_MAX_IN); p += BGZF_MAX_IN; n -= BGZF_MAX_IN; } // (longer than a block: split)
        cur.insert(cur.end(), p, p + n);
    }
};

// threads <= 0: the machine's, held to twice the cgroup's CPU quota (a pod that shows 256 processors may be granted 16)
int default_threads() {
    int t = (int)std::thread::hardware_concurrency();
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        if (fscanf(f, "%63s %lld", a, &period) >= 1 && strcmp(a, "max") != 0) quota = atoll(a);
        fclose(f);
    }
    if (quota > 0 && period > 0) t = std::min<int>(t, (int)(2 * ((quota + period - 1) / period)));
    return std::max(1, t);
}

template <typename F>
void run_workers(int w, F fn) {
    std::vector<std::thread> th;
    for (int k = 0; k < w; k++) th.emplace_back([&, k] { fn(k); });
    for (auto &t : th) t.join();
}

inline void put32(std::vector<uint8_t> &v, int32_t x) { const size_t at = v.size(); v.resize(at + 4); memcpy(v.data() + at, &x, 4); }
inline void put64(std::vector<uint8_t> &v, uint64_t x) { const size_t at = v.size(); v.resize(at + 8); memcpy(v.data() + at, &x, 8); }

// joins the workers' partial indexes (part w starts at file offset base[w]) into the per-reference body of a BAI / TBI
void write_index_refs(std::vector<uint8_t> &out, int32_t n_ref, const std::vector<Part> &parts, const std::vector<uint64_t> &base) {
    for (int32_t t = 0; t < n_ref; t++) {
        std::map<uint32_t, std::vector<Chunk>> bins;
        std::map<int64_t, uint64_t> linear;
        for (size_t w = 0; w < parts.size(); w++) {
            auto it = parts[w].idx.find(t);
            if (it == parts[w].idx.end()) continue;
            const uint64_t add = base[w] << 16;
            for (const auto &b : it->second.bins) {
                auto &ch = bins[b.first];
                for (size_t k = 0; k < b.second.size(); k++) {
                    const Chunk &c = b.second[k];
                    Chunk g{c.beg + add, c.end + add};
                    if (parts[w].tail_at_block_end && parts[w].last_tid == t && parts[w].last_bin == b.first && k + 1 == b.second.size() &&
                        c.end == parts[w].last_v1 && w + 1 < base.size())
                        g.end = base[w + 1] << 16; // the part's last record ended with its block: the next part's first block
                    if (!ch.empty() && ch.back().end == g.beg) ch.back().end = g.end;
                    else ch.push_back(g);
                }
            }
            for (const auto &l : it->second.linear)
                if (!linear.count(l.first)) linear[l.first] = l.second + add; // workers are in file order: the first is the smallest
        }
        put32(out, (int32_t)bins.size());
        for (const auto &b : bins) {
            put32(out, (int32_t)b.first);
            put32(out, (int32_t)b.second.size());
            for (const Chunk &c : b.second) { put64(out, c.beg); put64(out, c.end); }
        }
        const int64_t n_intv = linear.empty() ? 0 : linear.rbegin()->first + 1;
        std::vector<uint64_t> lin((size_t)n_intv, 0);
        for (const auto &l : linear) lin[(size_t)l.first] = l.second;
        for (int64_t w = n_intv - 2; w >= 0; w--)
            if (lin[(size_t)w] == 0) lin[(size_t)w] = lin[(size_t)w + 1]; // an empty window takes the following window's offset
        put32(out, (int32_t)n_intv);
        for (uint64_t v : lin) put64(out, v);
    }
}

bool write_all(FILE *f, const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }

int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

} // namespace

extern "C" {

// query name of pair `pair` (global pair number): what the BAM carries; the decoders intern names in file order
int uzs_qname(int64_t pair, char *buf) { return snprintf(buf, 40, "UZSYN:30X:1:%03d:%07d", (int)(pair % 997), (int)(pair / 997)); }

} // extern "C"

namespace {
// FILLER: the file between the pile-ups.  A real 30 x file holds records everywhere; the index-driven readers meet them as the lead-in of a
// 16 kb bin in front of a window and as records to walk past between two reach intervals.  Read pairs at `cov`-fold coverage laid into [g0, g1) --
// a gap between two clusters -- whole inside it with a margin, so that no fetch of the batch can ever return one (the phasing results are those of
// the file without filler): hashed bases, the pile-ups' quality profile, names of their own ("UZFIL:...").
inline int filler_insert(uint64_t h) { // the pile-ups' own template-length distribution (uzsynth.h: uzs_segment)
    int ins = 450;
    for (int j = 0; j < 4; j++) ins += (int)((h >> (20 + 8 * j)) & 0xFF) * 100 / 256 - 50;
    return ins;
}
void emit_filler(Part &P, const uzs_cfg *cf, int32_t tid, int64_t g0, int64_t g1, double cov, int tags, uint64_t gap_id, std::vector<uint64_t> &keys,
                 std::vector<uint8_t> &rec) {
    const int L = UZS_READLEN;
    const int64_t margin = 200;
    g0 += margin; g1 -= margin;
    if (g1 - g0 < 1200) return;
    const int64_t np = (int64_t)((double)(g1 - g0) * cov / (2.0 * L));
    if (np <= 0) return;
    keys.clear();
    const int64_t span = (g1 - g0) - 700;
    for (int64_t k = 0; k < np; k++) {
        const uint64_t h = uzs_h(cf->seed ^ 0xF111E5ULL, gap_id, (uint64_t)k, 7);
        const int64_t cell = span / np > 0 ? span / np : 1;
        const int64_t fs = g0 + k * span / np + (int64_t)((h >> 8) % (uint64_t)cell);
        const int ins = filler_insert(h);
        keys.push_back(((uint64_t)fs << 24) | ((uint64_t)k << 1));
        keys.push_back(((uint64_t)(fs + ins - L) << 24) | ((uint64_t)k << 1) | 1ULL);
    }
    if (np >= (1 << 22)) return; // (23 bits of pair number in the key)
    std::sort(keys.begin(), keys.end());
    for (uint64_t key : keys) {
        const int64_t start = (int64_t)(key >> 24), k = (int64_t)((key >> 1) & 0x3FFFFF);
        const int which = (int)(key & 1);
        const uint64_t h = uzs_h(cf->seed ^ 0xF111E5ULL, gap_id, (uint64_t)k, 7);
        const int64_t cell = span / np > 0 ? span / np : 1;
        const int64_t fs = g0 + k * span / np + (int64_t)((h >> 8) % (uint64_t)cell);
        const int ins = filler_insert(h);
        const int64_t mate_start = which ? fs : fs + ins - L;
        const uint16_t flag = (uint16_t)(0x1 | 0x2 | (which ? 0x10 : 0x20) | (which ? 0x80 : 0x40));
        char name[48];
        const int ln = snprintf(name, sizeof(name), "UZFIL:%llx:%06lld", (unsigned long long)(gap_id & 0xFFFFFFFFFFULL), (long long)k) + 1;
        rec.clear();
        put32(rec, 0);
        put32(rec, tid);
        put32(rec, (int32_t)start);
        rec.push_back((uint8_t)ln);
        rec.push_back((uint8_t)60);
        const uint16_t bin = (uint16_t)reg2bin(start, start + L), nops = 1;
        rec.insert(rec.end(), (const uint8_t *)&bin, (const uint8_t *)&bin + 2);
        rec.insert(rec.end(), (const uint8_t *)&nops, (const uint8_t *)&nops + 2);
        rec.insert(rec.end(), (const uint8_t *)&flag, (const uint8_t *)&flag + 2);
        put32(rec, L);
        put32(rec, tid);
        put32(rec, (int32_t)mate_start);
        put32(rec, which ? -ins : ins);
        rec.insert(rec.end(), name, name + ln);
        put32(rec, (int32_t)((uint32_t)L << 4));
        const uint64_t bseed = uzs_mix(h ^ (uint64_t)(which + 1));
        for (int b = 0; b < L; b += 2) { // (the reference's bases, as the pile-ups carry them: the same compressibility)
            auto code = [](uint8_t ch) -> uint8_t { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 15; };
            const uint8_t b0 = uzs_refbase(tid, start + b), b1 = b + 1 < L ? uzs_refbase(tid, start + b + 1) : 0;
            rec.push_back((uint8_t)((code(b0) << 4) | (b + 1 < L ? code(b1) : 0)));
        }
        for (int b = 0; b < L; b++) {
            const uint64_t hb = uzs_mix(bseed + (uint64_t)b * 0x9E3779B97F4A7C15ULL);
            rec.push_back((uint8_t)(((hb >> 24) % 100) < 3 ? 12 : 37));
        }
        if (tags) {
            const uint8_t nm[] = {'N', 'M', 'C', (uint8_t)(h & 3)};
            rec.insert(rec.end(), nm, nm + 4);
            char md[16];
            const int lm = snprintf(md, sizeof(md), "MDZ%d", L);
            rec.insert(rec.end(), md, md + lm + 1);
            const uint8_t as[] = {'A', 'S', 'C', (uint8_t)(L - (int)(h & 3) * 5), 'X', 'S', 'C', (uint8_t)((h >> 8) % 40)};
            rec.insert(rec.end(), as, as + 8);
            const char rg[] = "RGZgrp1";
            rec.insert(rec.end(), rg, rg + sizeof(rg));
        }
        const int32_t bs = (int32_t)rec.size() - 4;
        memcpy(rec.data(), &bs, 4);
        P.add_record(tid, start, start + L, rec.data(), rec.size());
    }
}
} // namespace

extern "C" {

/* The records of clusters [c0, c1) as a coordinate-sorted BAM + BAI.  `tags`: 1 = every record carries the usual aligner tags
 * (NM, MD, AS, XS, RG: ~40 bytes the readers have to walk past).  stats: [0] records, [1] uncompressed bytes, [2] file bytes,
 * [3] BGZF blocks.  Returns 0, or a negative number with a message in err (cap bytes).
 * filler_cov > 0: the gaps between the clusters -- and filler_reach bases in front of a contig's first and behind its last cluster of the range --
 * are filled with read pairs at that coverage (emit_filler): a file that looks like a file, not like a list of pile-ups. */
int uzs_write_bam_filler(const char *bam_path, const char *bai_path, const uzs_cfg *cf, const uzs_sites *S, const uzs_dnms *D, const uzs_clusters *C,
                         int32_t c0, int32_t c1, const char *const *contig_names, const int32_t *contig_len, int level, int tags, int threads,
                         double filler_cov, int64_t filler_reach, int64_t *stats, char *err, int cap);
int uzs_write_bam(const char *bam_path, const char *bai_path, const uzs_cfg *cf, const uzs_sites *S, const uzs_dnms *D, const uzs_clusters *C,
                  int32_t c0, int32_t c1, const char *const *contig_names, const int32_t *contig_len, int level, int tags, int threads,
                  int64_t *stats, char *err, int cap) {
    return uzs_write_bam_filler(bam_path, bai_path, cf, S, D, C, c0, c1, contig_names, contig_len, level, tags, threads, 0.0, 0, stats, err, cap);
}
int uzs_write_bam_filler(const char *bam_path, const char *bai_path, const uzs_cfg *cf, const uzs_sites *S, const uzs_dnms *D, const uzs_clusters *C,
                         int32_t c0, int32_t c1, const char *const *contig_names, const int32_t *contig_len, int level, int tags, int threads,
                         double filler_cov, int64_t filler_reach, int64_t *stats, char *err, int cap) {
    auto fail = [&](const char *m) { if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m); return -1; };
    if (!bam_path || !cf || !S || !D || !C || c1 < c0) return fail("bad argument");
    if (threads <= 0) threads = default_threads();
    const int64_t pairs = C->pair_off[c1] - C->pair_off[c0];
    int W = (int)std::max<int64_t>(1, std::min<int64_t>(threads, pairs / 2048 + 1));
    W = std::min<int>(W, std::max(1, c1 - c0));
    // cluster ranges of about equal pair counts
    std::vector<int32_t> cut((size_t)W + 1, c0);
    for (int k = 1; k < W; k++) {
        const int64_t want = C->pair_off[c0] + pairs * k / W;
        cut[(size_t)k] = (int32_t)(std::lower_bound(C->pair_off + c0, C->pair_off + c1, want) - C->pair_off);
        if (cut[(size_t)k] < cut[(size_t)k - 1]) cut[(size_t)k] = cut[(size_t)k - 1];
    }
    cut[(size_t)W] = c1;
    const int32_t n_ref = S->n_contigs;
    std::vector<Part> parts((size_t)W + 2); // [0] header, [1 .. W] workers, [W + 1] end-of-file marker
    for (auto &p : parts) p.init(level);
    { // header: magic, text, references
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int32_t r = 0; r < n_ref; r++) text += std::string("@SQ\tSN:") + contig_names[r] + "\tLN:" + std::to_string(contig_len[r]) + "\n";
        text += "@RG\tID:grp1\tSM:kid\n@PG\tID:uzsynth\tPN:uzsynth\n";
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, (int32_t)text.size());
        h.insert(h.end(), text.begin(), text.end());
        put32(h, n_ref);
        for (int32_t r = 0; r < n_ref; r++) {
            const size_t ln = strlen(contig_names[r]) + 1;
            put32(h, (int32_t)ln);
            h.insert(h.end(), contig_names[r], contig_names[r] + ln);
            put32(h, contig_len[r]);
        }
        for (size_t at = 0; at < h.size(); at += BGZF_MAX_IN) parts[0].emit(h.data() + at, std::min(BGZF_MAX_IN, h.size() - at));
    }
    parts[(size_t)W + 1].emit(nullptr, 0);
    run_workers(W, [&](int k) {
        Part &P = parts[(size_t)k + 1];
        std::vector<uint32_t> keys(UZS_MAXSEG);
        std::vector<uint8_t> rec;
        uint8_t sq[UZS_ROW], ql[UZS_ROW];
        std::vector<uint64_t> fkeys;
        for (int32_t c = cut[(size_t)k]; c < cut[(size_t)k + 1] && P.err.empty(); c++) {
            if (filler_cov > 0) { // the gap in front of the cluster (its worker writes it: the parts stay in file order)
                const bool first_of_contig = c == c0 || C->contig[c - 1] != C->contig[c];
                const int64_t g0 = first_of_contig ? std::max<int64_t>(0, (int64_t)C->lo[c] - filler_reach) : (int64_t)C->hi[c - 1];
                emit_filler(P, cf, C->contig[c], g0, C->lo[c], filler_cov, tags, (uint64_t)c * 2, fkeys, rec);
            }
            const int nseg = (int)(2 * (C->pair_off[c + 1] - C->pair_off[c]));
            if (nseg > UZS_MAXSEG) { P.err = "a cluster holds more than UZS_MAXSEG records"; break; }
            for (int slot = 0; slot < nseg; slot++) {
                uzs_seg s;
                uzs_segment(cf, C, D, c, slot >> 1, slot & 1, &s);
                keys[(size_t)slot] = uzs_key(C, c, slot, &s);
            }
            qsort(keys.data(), (size_t)nseg, sizeof(uint32_t), cmp_u32);
            int64_t s_lo, s_hi;
            uzs_site_window(S, C, c, &s_lo, &s_hi);
            const int32_t tid = C->contig[c];
            for (int p = 0; p < nseg; p++) {
                const int slot = (int)(keys[(size_t)p] & 0x3FFF);
                uzs_seg s, m;
                uzs_segment(cf, C, D, c, slot >> 1, slot & 1, &s);
                uzs_segment(cf, C, D, c, slot >> 1, (slot & 1) ^ 1, &m);
                const int L = uzs_query_len(&s);
                uzs_fill(S, C, D, c, &s, s_lo, s_hi, 0, L, sq, ql);
                char name[48];
                const int ln = uzs_qname(C->pair_off[c] + (slot >> 1), name) + 1;
                rec.clear();
                put32(rec, 0); // block_size, patched below
                put32(rec, tid);
                put32(rec, s.start);
                rec.push_back((uint8_t)ln);
                rec.push_back(s.mapq);
                const uint16_t bin = (uint16_t)reg2bin(s.start, s.end), nops = s.n_ops;
                rec.insert(rec.end(), (const uint8_t *)&bin, (const uint8_t *)&bin + 2);
                rec.insert(rec.end(), (const uint8_t *)&nops, (const uint8_t *)&nops + 2);
                rec.insert(rec.end(), (const uint8_t *)&s.flag, (const uint8_t *)&s.flag + 2);
                put32(rec, L);
                put32(rec, tid);
                put32(rec, m.start);
                put32(rec, s.tlen);
                rec.insert(rec.end(), name, name + ln);
                for (int j = 0; j < s.n_ops; j++) put32(rec, (int32_t)s.ops[j]);
                for (int b = 0; b < L; b += 2) {
                    auto code = [](uint8_t ch) -> uint8_t { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 15; };
                    rec.push_back((uint8_t)((code(sq[b]) << 4) | (b + 1 < L ? code(sq[b + 1]) : 0)));
                }
                rec.insert(rec.end(), ql, ql + L);
                if (tags) {
                    const uint64_t h = uzs_mix(s.bseed ^ 0x7a67ULL);
                    const uint8_t nm[] = {'N', 'M', 'C', (uint8_t)(h & 3)};
                    rec.insert(rec.end(), nm, nm + 4);
                    char md[16];
                    const int lm = snprintf(md, sizeof(md), "MDZ%d", L);
                    rec.insert(rec.end(), md, md + lm + 1);
                    const uint8_t as[] = {'A', 'S', 'C', (uint8_t)(L - (int)(h & 3) * 5), 'X', 'S', 'C', (uint8_t)((h >> 8) % 40)};
                    rec.insert(rec.end(), as, as + 8);
                    const char rg[] = "RGZgrp1";
                    rec.insert(rec.end(), rg, rg + sizeof(rg));
                }
                const int32_t bs = (int32_t)rec.size() - 4;
                memcpy(rec.data(), &bs, 4);
                P.add_record(tid, s.start, s.end, rec.data(), rec.size());
            }
            if (filler_cov > 0 && (c + 1 == c1 || C->contig[c + 1] != tid)) // behind the contig's last cluster of the range
                emit_filler(P, cf, tid, C->hi[c], std::min<int64_t>((int64_t)C->hi[c] + filler_reach, contig_len[tid]), filler_cov, tags, (uint64_t)c * 2 + 1, fkeys, rec);
        }
        P.finish();
    });
    for (auto &p : parts) if (!p.err.empty()) return fail(p.err.c_str());
    std::vector<uint64_t> base(parts.size(), 0);
    for (size_t w = 1; w < parts.size(); w++) base[w] = base[w - 1] + parts[w - 1].cdata.size();
    FILE *f = fopen(bam_path, "wb");
    if (!f) return fail("cannot create the BAM file");
    bool ok = true;
    for (auto &p : parts) ok = ok && write_all(f, p.cdata.data(), p.cdata.size());
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail("short write on the BAM file");
    if (bai_path) {
        std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
        put32(bai, n_ref);
        write_index_refs(bai, n_ref, parts, base);
        FILE *g = fopen(bai_path, "wb");
        if (!g) return fail("cannot create the BAI file");
        ok = write_all(g, bai.data(), bai.size());
        ok = (fclose(g) == 0) && ok;
        if (!ok) return fail("short write on the BAI file");
    }
    if (stats) {
        stats[0] = stats[1] = stats[3] = 0;
        for (auto &p : parts) { stats[0] += p.n_records; stats[1] += p.n_raw; stats[3] += p.n_blocks; }
        stats[2] = (int64_t)(base.back() + parts.back().cdata.size());
    }
    return 0;
}

/* The sites table as a BGZF-compressed VCF (GT:AD:DP:GQ per sample; complex records alternate between a multi-allelic ALT and
 * a two-base REF) + its tabix index.  Depth / GQ columns: 0xFFFF = missing ("."); gt: kid | dad << 2 | mom << 4 in cyvcf2's
 * codes (0 "0/0", 1 "0/1", 2 "./.", 3 "1/1").  stats as uzs_write_bam ([0] = records). */
int uzs_write_vcf(const char *vcf_path, const char *tbi_path, int64_t n_sites, int32_t n_contigs, const int64_t *contig_off,
                  const char *const *contig_names, const int32_t *contig_len, const int32_t *pos, const uint8_t *sflags, const uint8_t *ref_base,
                  const uint8_t *alt_base, const uint8_t *gt, const uint16_t *const *rd, const uint16_t *const *ad, const uint16_t *const *gq,
                  const char *const *samples, int level, int threads, int64_t *stats, char *err, int cap) {
    auto fail = [&](const char *m) { if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m); return -1; };
    if (!vcf_path || !contig_off || !pos || !gt || !rd || !ad || !gq) return fail("bad argument");
    if (threads <= 0) threads = default_threads();
    const int W = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_sites / 4096 + 1));
    std::vector<Part> parts((size_t)W + 2);
    for (auto &p : parts) p.init(level);
    {
        std::string h = "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
        for (int32_t c = 0; c < n_contigs; c++) h += std::string("##contig=<ID=") + contig_names[c] + ",length=" + std::to_string(contig_len[c]) + ">\n";
        h += "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
             "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths\">\n"
             "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n"
             "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n"
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
        for (int m = 0; m < 3; m++) h += std::string("\t") + samples[m];
        h += "\n";
        for (size_t at = 0; at < h.size(); at += BGZF_MAX_IN) parts[0].emit((const uint8_t *)h.data() + at, std::min(BGZF_MAX_IN, h.size() - at));
    }
    parts[(size_t)W + 1].emit(nullptr, 0);
    run_workers(W, [&](int k) {
        Part &P = parts[(size_t)k + 1];
        const int64_t lo = n_sites * k / W, hi = n_sites * (k + 1) / W;
        int32_t c = (int32_t)(std::upper_bound(contig_off, contig_off + n_contigs + 1, lo) - contig_off) - 1;
        char line[512];
        static const char *GT_TEXT[4] = {"0/0", "0/1", "./.", "1/1"};
        for (int64_t i = lo; i < hi && P.err.empty(); i++) {
            while (c + 1 < n_contigs && i >= contig_off[c + 1]) c++;
            const bool cx = (sflags[i] & 1) != 0;
            const bool multi = cx && (i & 1);
            char ref[4] = {0, 0, 0, 0}, alt[8] = {0};
            if (!cx) { ref[0] = (char)ref_base[i]; alt[0] = (char)alt_base[i]; }
            else if (multi) { ref[0] = 'A'; strcpy(alt, "C,G"); }
            else { ref[0] = 'A'; ref[1] = 'T'; alt[0] = 'A'; }
            int n = snprintf(line, sizeof(line), "%s\t%d\t.\t%s\t%s\t50\tPASS\t.\tGT:AD:DP:GQ", contig_names[c], pos[i] + 1, ref, alt);
            for (int m = 0; m < 3; m++) {
                const unsigned g = (gt[i] >> (2 * m)) & 3u;
                const uint16_t r = rd[m][i], a = ad[m][i], q = gq[m][i];
                n += snprintf(line + n, sizeof(line) - (size_t)n, "\t%s:", GT_TEXT[g]);
                if (r == 0xFFFF || a == 0xFFFF) n += snprintf(line + n, sizeof(line) - (size_t)n, ".:.");
                else if (multi) n += snprintf(line + n, sizeof(line) - (size_t)n, "%u,%u,0:%u", r, a, r + a);
                else n += snprintf(line + n, sizeof(line) - (size_t)n, "%u,%u:%u", r, a, r + a);
                if (q == 0xFFFF) n += snprintf(line + n, sizeof(line) - (size_t)n, ":.");
                else n += snprintf(line + n, sizeof(line) - (size_t)n, ":%u", q);
            }
            line[n++] = '\n';
            P.add_record(c, pos[i], (int64_t)pos[i] + (cx && !multi ? 2 : 1), (const uint8_t *)line, (size_t)n);
        }
        P.finish();
    });
    for (auto &p : parts) if (!p.err.empty()) return fail(p.err.c_str());
    std::vector<uint64_t> base(parts.size(), 0);
    for (size_t w = 1; w < parts.size(); w++) base[w] = base[w - 1] + parts[w - 1].cdata.size();
    FILE *f = fopen(vcf_path, "wb");
    if (!f) return fail("cannot create the VCF file");
    bool ok = true;
    for (auto &p : parts) ok = ok && write_all(f, p.cdata.data(), p.cdata.size());
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail("short write on the VCF file");
    if (tbi_path) {
        // tabix lists only the sequences that have records, in file order
        std::vector<int32_t> have;
        for (int32_t c = 0; c < n_contigs; c++) if (contig_off[c + 1] > contig_off[c]) have.push_back(c);
        std::string names;
        for (int32_t c : have) { names += contig_names[c]; names.push_back('\0'); }
        std::vector<uint8_t> t = {'T', 'B', 'I', 1};
        put32(t, (int32_t)have.size());
        put32(t, 2); put32(t, 1); put32(t, 2); put32(t, 0); put32(t, '#'); put32(t, 0); // VCF preset: seq 1, beg 2, meta '#'
        put32(t, (int32_t)names.size());
        t.insert(t.end(), names.begin(), names.end());
        // the parts index by contig id: renumber to the listed sequences
        std::vector<Part> renum(parts.size());
        for (size_t w = 0; w < parts.size(); w++) {
            renum[w].tail_at_block_end = parts[w].tail_at_block_end; renum[w].last_bin = parts[w].last_bin; renum[w].last_v1 = parts[w].last_v1;
            for (size_t r = 0; r < have.size(); r++) {
                auto it = parts[w].idx.find(have[r]);
                if (it != parts[w].idx.end()) renum[w].idx[(int32_t)r] = it->second;
                if (parts[w].last_tid == have[r]) renum[w].last_tid = (int32_t)r;
            }
        }
        write_index_refs(t, (int32_t)have.size(), renum, base);
        Part z;
        z.init(level);
        for (size_t at = 0; at < t.size(); at += BGZF_MAX_IN) z.emit(t.data() + at, std::min(BGZF_MAX_IN, t.size() - at));
        z.emit(nullptr, 0);
        if (!z.err.empty()) return fail(z.err.c_str());
        FILE *g = fopen(tbi_path, "wb");
        if (!g) return fail("cannot create the TBI file");
        ok = write_all(g, z.cdata.data(), z.cdata.size());
        ok = (fclose(g) == 0) && ok;
        if (!ok) return fail("short write on the TBI file");
    }
    if (stats) {
        stats[0] = stats[1] = stats[3] = 0;
        for (auto &p : parts) { stats[0] += p.n_records; stats[1] += p.n_raw; stats[3] += p.n_blocks; }
        stats[2] = (int64_t)(base.back() + parts.back().cdata.size());
    }
    return 0;
}

} // extern "C"
