// io_bam.cpp -- native BAM decoder: BGZF file -> the alignment-record column table of
// include/uz_types.h (uz_reads_view).  Host only; built into libunfazed_io.so by unfazed_amd/build.py.
//
// What it stands in for: pysam.AlignmentFile as the reference uses it
// in unfazed/read_collector.py -- records in file order with flag, mapq, reference_start, bam_endpos,
// CIGAR, mate contig / position, template length, query name, sequence, qualities, "has an SA tag" --
// and AlignmentFile.mate(), which the reference calls once per read (:186, :402, :509) and which
// becomes one precomputed column here.  The column semantics are those of the Python decoder
// (unfazed_amd/io_bam.py + model.ReadsTable.from_segments), which stays in the tree as the readable
// statement of the format; tests/test_io_native.py holds the two against each other.
//
// Stages: read file -> inflate BGZF blocks (parallel) -> walk record boundaries (serial, a pointer
// chase) -> fixed-width columns + CIGAR / SEQ / QUAL payloads (parallel over records) -> query-name
// interning in order of first appearance (hash-sharded, parallel) -> mate links (parallel).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "io_common.hpp"
#include "io_index.hpp"

namespace uzio {

thread_local std::string last_error;

Bytes read_file(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) fail(UZ_IO_E_OPEN, "cannot open %s", path);
    Bytes data;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long sz = ftell(f);
        data.alloc(sz > 0 ? (size_t)sz : 0);
        fseek(f, 0, SEEK_SET);
    }
    size_t got = 0;
    while (got < data.size()) {
        const size_t k = fread(data.data() + got, 1, data.size() - got, f);
        if (k == 0) break;
        got += k;
    }
    fclose(f);
    if (got != data.size()) fail(UZ_IO_E_OPEN, "short read on %s", path);
    return data;
}

namespace {

struct Block {
    size_t cdata, clen; // compressed payload
    size_t out_off;
    uint32_t isize, crc;
};

// the BGZF framing: every gzip member carries its own compressed size in a "BC" extra subfield
bool scan_bgzf(const Bytes &f, std::vector<Block> &blocks, size_t &total) {
    size_t p = 0;
    total = 0;
    while (p < f.size()) {
        if (p + 18 > f.size()) return false;
        const uint8_t *h = f.data() + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const size_t xlen = rd16(h + 10);
        if (p + 12 + xlen > f.size()) return false;
        size_t q = 12, bsize = 0;
        bool found = false;
        while (q + 4 <= 12 + xlen) {
            const size_t slen = rd16(h + q + 2);
            if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) { bsize = rd16(h + q + 4); found = true; }
            q += 4 + slen;
        }
        if (!found) return false;
        const size_t blen = bsize + 1;
        if (blen < 12 + xlen + 8 || p + blen > f.size()) return false;
        Block b;
        b.cdata = p + 12 + xlen;
        b.clen = blen - 12 - xlen - 8;
        b.crc = rd32(h + blen - 8);
        b.isize = rd32(h + blen - 4);
        b.out_off = total;
        total += b.isize;
        blocks.push_back(b);
        p += blen;
    }
    return true;
}

Bytes inflate_stream(const Bytes &f) { // any gzip stream, member after member
    std::vector<uint8_t> out;
    size_t p = 0;
    std::vector<uint8_t> chunk(1 << 20);
    while (p < f.size()) {
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, 31) != Z_OK) fail(UZ_IO_E_FORMAT, "zlib init failed");
        z.next_in = const_cast<Bytef *>(f.data() + p);
        z.avail_in = (uInt)std::min<size_t>(f.size() - p, 1u << 30);
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            z.next_out = chunk.data();
            z.avail_out = (uInt)chunk.size();
            rc = inflate(&z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&z);
                fail(UZ_IO_E_FORMAT, "corrupt gzip stream at byte %zu", p);
            }
            out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - z.avail_out));
            if (rc == Z_OK && z.avail_in == 0 && z.avail_out != 0) {
                inflateEnd(&z);
                fail(UZ_IO_E_FORMAT, "truncated gzip stream");
            }
        }
        const size_t used = (size_t)z.total_in;
        inflateEnd(&z);
        if (used == 0) break;
        p += used;
    }
    Bytes b(out.size());
    if (!out.empty()) memcpy(b.p, out.data(), out.size());
    return b;
}

} // namespace

Bytes inflate_all(Bytes &f, int threads, bool *was_gzip) {
    if (f.size() < 2 || f[0] != 0x1f || f[1] != 0x8b) {
        if (was_gzip) *was_gzip = false;
        return std::move(f);
    }
    if (was_gzip) *was_gzip = true;
    std::vector<Block> blocks;
    size_t total = 0;
    if (!scan_bgzf(f, blocks, total)) return inflate_stream(f);
    Bytes out(total);
    parallel_slices((int64_t)blocks.size(), workers_for((int64_t)blocks.size(), threads, 8), [&](int64_t lo, int64_t hi, int) {
        Inflater inf;
        for (int64_t k = lo; k < hi; k++) {
            const Block &b = blocks[(size_t)k];
            if (b.isize == 0 && b.clen <= 2) continue; // EOF marker
            inf.block(f.data() + b.cdata, b.clen, out.data() + b.out_off, b.isize, b.crc, (int64_t)b.cdata);
        }
    });
    return out;
}

} // namespace uzio

using namespace uzio;

namespace {

template <typename T>
struct ZBuf { // calloc-backed column: zero pages come from the OS, nothing is touched twice
    T *p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        free(p);
        n = count;
        p = (T *)calloc(count ? count : 1, sizeof(T));
        if (!p) fail(UZ_IO_E_RANGE, "out of memory for %zu elements", count);
    }
    ~ZBuf() { free(p); }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

inline uint64_t hash_bytes(const uint8_t *s, size_t n) { // FNV-1a with a final mix
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ULL; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ULL; h ^= h >> 29;
    return h;
}

} // namespace

struct uz_bam {
    Bytes data; // inflated file (query names are referenced in place)
    std::vector<std::string> contigs;
    std::vector<int32_t> contig_len;
    int64_t n_file = 0, n = 0;
    std::vector<int64_t> contig_off;
    std::vector<int32_t> max_span;
    ZBuf<int32_t> start, end, tlen, mate;
    ZBuf<uint16_t> flag, n_cigar, l_seq;
    ZBuf<uint8_t> mapq, aux, seq, qual;
    ZBuf<uint32_t> qname, cigar_off, cigar, sq_off16;
    int64_t n_cigar_total = 0, n_sq_bytes = 0;
    std::vector<uint64_t> name_at; // per id: offset of the name in `data`
    std::vector<uint8_t> name_len;
    std::vector<int32_t> tlen_file;
    int64_t io_stats[4] = {0, 0, 0, 0}; // file bytes read, BGZF blocks inflated, records walked, records kept
    mutable std::string name_tmp;
    double timing[4] = {0, 0, 0, 0};
};

namespace {

const uint16_t FPAIRED = 1, FUNMAP = 4, FMUNMAP = 8, FREAD1 = 64, FREAD2 = 128;

bool has_tag(const uint8_t *p, const uint8_t *end, char t0, char t1) {
    while (p + 3 <= end) {
        const uint8_t a = p[0], b = p[1], typ = p[2];
        p += 3;
        if (a == (uint8_t)t0 && b == (uint8_t)t1) return true;
        size_t sz = 0;
        switch (typ) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (!z) return false;
            p = z + 1;
            continue;
        }
        case 'B': {
            if (p + 5 > end) return false;
            size_t es = 0;
            switch (p[0]) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return false; }
            const int32_t cnt = rdi32(p + 1);
            if (cnt < 0) return false;
            p += 5 + (size_t)cnt * es;
            continue;
        }
        default: return false;
        }
        p += sz;
    }
    return false;
}

// BAM header at the start of the inflated stream d[0, N): fills the contig tables, returns the offset of the first record
size_t parse_header(uz_bam &B, const uint8_t *d, size_t N, const char *path, int32_t &n_ref, bool allow_short = false) {
    if (N < 12 || memcmp(d, "BAM\1", 4) != 0) fail(UZ_IO_E_FORMAT, "%s is not a BAM file", path);
    if (rdi32(d + 4) < 0) fail(UZ_IO_E_FORMAT, "bad BAM header (negative l_text)");
    size_t off = 8 + (size_t)rdi32(d + 4);
    if (off + 4 > N) { if (allow_short) return 0; fail(UZ_IO_E_FORMAT, "truncated BAM header"); }
    n_ref = rdi32(d + off);
    if (n_ref < 0) fail(UZ_IO_E_FORMAT, "bad BAM header (negative n_ref)");
    off += 4;
    B.contigs.clear(); B.contig_len.clear();
    for (int32_t r = 0; r < n_ref; r++) {
        if (off + 4 > N) { if (allow_short) return 0; fail(UZ_IO_E_FORMAT, "truncated BAM header"); }
        const int32_t l_name = rdi32(d + off);
        if (l_name < 1) fail(UZ_IO_E_FORMAT, "bad BAM header");
        if (off + 4 + (size_t)l_name + 4 > N) { if (allow_short) return 0; fail(UZ_IO_E_FORMAT, "truncated BAM header"); }
        B.contigs.emplace_back((const char *)d + off + 4, (size_t)l_name - 1);
        B.contig_len.push_back(rdi32(d + off + 4 + l_name));
        off += 4 + (size_t)l_name + 4;
    }
    return off;
}

void build_table(uz_bam &B, std::vector<uint64_t> &rec, std::vector<uint32_t> &rec_end32, int32_t n_ref, int threads, double t2);

void decode_stream(uz_bam &B, const char *path, int threads, double t2);

void decode(uz_bam &B, const char *path, int threads) {
    double t0 = now_s();
    Bytes file = read_file(path);
    double t1 = now_s();
    B.timing[0] = t1 - t0;
    B.io_stats[0] = (int64_t)file.size();
    bool gz = false;
    B.data = inflate_all(file, threads, &gz);
    file.release();
    double t2 = now_s();
    B.timing[1] = t2 - t1;
    decode_stream(B, path, threads, t2);
}

// the table of an uncompressed BAM stream held in B.data
void decode_stream(uz_bam &B, const char *path, int threads, double t2) {
    const Bytes &d = B.data;
    const size_t N = d.size();
    int32_t n_ref = 0;
    size_t off = parse_header(B, d.data(), N, path, n_ref);
    // record boundaries (file order); kept records have a reference id
    std::vector<uint64_t> rec; // offsets of the kept records' fixed part (after block_size)
    std::vector<uint32_t> rec_end32; // block_size of the kept records
    while (off + 4 <= N) {
        const int32_t bs = rdi32(d.data() + off);
        if (bs < 32 || off + 4 + (size_t)bs > N) fail(UZ_IO_E_FORMAT, "truncated alignment record at byte %zu", off);
        const uint8_t *p = d.data() + off + 4;
        if (B.tlen_file.size() < ((size_t)1 << 24)) B.tlen_file.push_back(rdi32(p + 28)); // the head of the file only (insert-size estimate)
        if (rdi32(p) >= 0) { rec.push_back(off + 4); rec_end32.push_back((uint32_t)bs); }
        off += 4 + (size_t)bs;
        B.n_file++;
    }
    B.io_stats[2] = B.n_file;
    build_table(B, rec, rec_end32, n_ref, threads, t2);
}

// columns, names and mate links of the records rec[] (offsets into B.data, file order)
void build_table(uz_bam &B, std::vector<uint64_t> &rec, std::vector<uint32_t> &rec_end32, int32_t n_ref, int threads, double t2) {
    const Bytes &d = B.data;
    const int64_t n = (int64_t)rec.size();
    threads = workers_for(n, threads, 16384); // a thread per 16k records at most
    if (n >= ((int64_t)1 << 31)) fail(UZ_IO_E_RANGE, "more than 2^31 - 1 alignment records");
    B.n = n;
    const size_t un = (size_t)n;
    B.start.alloc(un); B.end.alloc(un); B.tlen.alloc(un); B.mate.alloc(un);
    B.flag.alloc(un); B.n_cigar.alloc(un); B.l_seq.alloc(un);
    B.mapq.alloc(un); B.aux.alloc(un);
    B.qname.alloc(un); B.cigar_off.alloc(un); B.sq_off16.alloc(un);
    std::vector<int32_t> tid(un), mtid(un), mpos(un);
    // pass 1: fixed-width columns
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; i++) {
            const uint8_t *p = d.data() + rec[(size_t)i];
            const uint8_t *pend = p + rec_end32[(size_t)i];
            const int32_t refid = rdi32(p), pos = rdi32(p + 4);
            const uint32_t l_name = p[8];
            const uint32_t ncig = rd16(p + 12);
            const uint16_t fl = rd16(p + 14);
            const int32_t lseq = rdi32(p + 16);
            if (lseq < 0 || lseq > 0xFFFF) fail(UZ_IO_E_RANGE, "record too long for the 16-bit length columns (l_seq %d)", lseq);
            const uint8_t *q = p + 32 + l_name;
            const uint8_t *sq = q + 4 * (size_t)ncig;
            const uint8_t *ql = sq + ((size_t)lseq + 1) / 2;
            const uint8_t *tags = ql + lseq;
            if (l_name < 1 || tags > pend) fail(UZ_IO_E_FORMAT, "alignment record %lld overruns its block", (long long)i);
            if (refid >= n_ref) fail(UZ_IO_E_FORMAT, "reference id %d out of range", refid);
            tid[(size_t)i] = refid;
            B.start[(size_t)i] = pos;
            B.flag[(size_t)i] = fl;
            B.mapq[(size_t)i] = p[9];
            mtid[(size_t)i] = rdi32(p + 20);
            mpos[(size_t)i] = rdi32(p + 24);
            B.tlen[(size_t)i] = rdi32(p + 28);
            B.n_cigar[(size_t)i] = (uint16_t)ncig;
            B.l_seq[(size_t)i] = (uint16_t)lseq;
            int64_t rl = 0; // bam_endpos
            for (uint32_t k = 0; k < ncig; k++) {
                const uint32_t v = rd32(q + 4 * k), op = v & 15;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4;
            }
            B.end[(size_t)i] = (fl & FUNMAP) || ncig == 0 ? pos + 1 : (int32_t)(pos + (rl > 0 ? rl : 1));
            uint8_t a = 0;
            if (mtid[(size_t)i] == refid) a |= UZ_AUX_MATE_SAME_TID;
            if (has_tag(tags, pend, 'S', 'A')) a |= UZ_AUX_HAS_SA;
            if (ncig == 0 || lseq == 0 || ql[0] == 0xFF) a |= UZ_AUX_DECODE_BAD;
            B.aux[(size_t)i] = a;
        }
    });
    // sortedness, contig ranges, payload offsets (serial prefix sums)
    B.contig_off.assign((size_t)n_ref + 1, 0);
    {
        int64_t cg = 0, sq16 = 0;
        for (int64_t i = 0; i < n; i++) {
            if (i > 0 && (tid[(size_t)i] < tid[(size_t)i - 1] ||
                          (tid[(size_t)i] == tid[(size_t)i - 1] && B.start[(size_t)i] < B.start[(size_t)i - 1])))
                fail(UZ_IO_E_UNSORTED, "alignment records are not coordinate sorted");
            B.contig_off[(size_t)tid[(size_t)i] + 1]++;
            B.cigar_off[(size_t)i] = (uint32_t)cg;
            cg += B.n_cigar[(size_t)i];
            if (cg > 0xFFFFFFFFLL) fail(UZ_IO_E_RANGE, "CIGAR operations exceed the 32-bit offset range");
            if (sq16 > 0xFFFFFFFFLL) fail(UZ_IO_E_RANGE, "sequence bytes exceed the 64 GiB row-offset range");
            B.sq_off16[(size_t)i] = (uint32_t)sq16;
            sq16 += ((int64_t)B.l_seq[(size_t)i] + 15) >> 4;
        }
        for (int32_t c = 0; c < n_ref; c++) B.contig_off[(size_t)c + 1] += B.contig_off[(size_t)c];
        B.n_cigar_total = cg;
        B.n_sq_bytes = sq16 << 4;
    }
    B.cigar.alloc((size_t)B.n_cigar_total);
    B.seq.alloc((size_t)B.n_sq_bytes);
    B.qual.alloc((size_t)B.n_sq_bytes);
    static const char CODE[] = "=ACMGRSVTWYHKDBN";
    uint16_t pair[256];
    for (int v = 0; v < 256; v++) { const uint8_t two[2] = {(uint8_t)CODE[v >> 4], (uint8_t)CODE[v & 15]}; memcpy(&pair[v], two, 2); }
    B.max_span.assign((size_t)n_ref, 0);
    std::vector<std::vector<int32_t>> span_w;
    const int W = (int)std::min<int64_t>(resolve_threads(threads), std::max<int64_t>(n, 1));
    span_w.assign((size_t)W, std::vector<int32_t>((size_t)n_ref, 0));
    // pass 2: payloads
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int w) {
        std::vector<int32_t> &sp = span_w[(size_t)w];
        for (int64_t i = lo; i < hi; i++) {
            const uint8_t *p = d.data() + rec[(size_t)i];
            const uint32_t l_name = p[8], ncig = B.n_cigar[(size_t)i], L = B.l_seq[(size_t)i];
            const uint8_t *q = p + 32 + l_name;
            if (ncig) memcpy(B.cigar.p + B.cigar_off[(size_t)i], q, 4 * (size_t)ncig);
            const int32_t s = B.end[(size_t)i] - B.start[(size_t)i];
            if (s > sp[(size_t)tid[(size_t)i]]) sp[(size_t)tid[(size_t)i]] = s;
            if (!L) continue;
            const uint8_t *sq = q + 4 * (size_t)ncig;
            const uint8_t *ql = sq + ((size_t)L + 1) / 2;
            uint8_t *so = B.seq.p + ((size_t)B.sq_off16[(size_t)i] << 4);
            uint32_t k = 0;
            for (; k + 1 < L; k += 2) memcpy(so + k, &pair[sq[k >> 1]], 2); // rows are padded to 16 bytes
            if (k < L) so[k] = (uint8_t)CODE[sq[k >> 1] >> 4];
            if (ql[0] != 0xFF) memcpy(B.qual.p + ((size_t)B.sq_off16[(size_t)i] << 4), ql, L);
        }
    });
    for (auto &sp : span_w)
        for (int32_t c = 0; c < n_ref; c++) if (sp[(size_t)c] > B.max_span[(size_t)c]) B.max_span[(size_t)c] = sp[(size_t)c];
    double t3 = now_s();
    B.timing[2] = t3 - t2;

    // query names -> ids in order of first appearance.  Records are bucketed by name hash; a bucket
    // keeps file order, so the first record seen for a name is its first in the file and the chain
    // of later records with that name is ascending.
    const int SH = 256;
    std::vector<uint64_t> hsh(un);
    const int Wn = (int)std::min<int64_t>(resolve_threads(threads), std::max<int64_t>(n, 1));
    std::vector<std::vector<int64_t>> hist((size_t)Wn, std::vector<int64_t>(SH, 0));
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int w) {
        for (int64_t i = lo; i < hi; i++) {
            const uint8_t *p = d.data() + rec[(size_t)i];
            const uint64_t h = hash_bytes(p + 32, (size_t)p[8] - 1);
            hsh[(size_t)i] = h;
            hist[(size_t)w][h >> 56]++;
        }
    });
    std::vector<int64_t> sh_off(SH + 1, 0);
    {
        int64_t run = 0;
        for (int s = 0; s < SH; s++) {
            sh_off[(size_t)s] = run;
            for (int w = 0; w < Wn; w++) { const int64_t c = hist[(size_t)w][(size_t)s]; hist[(size_t)w][(size_t)s] = run; run += c; }
        }
        sh_off[SH] = run;
    }
    std::vector<int32_t> order(un);
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int w) {
        std::vector<int64_t> &at = hist[(size_t)w];
        for (int64_t i = lo; i < hi; i++) order[(size_t)at[hsh[(size_t)i] >> 56]++] = (int32_t)i;
    });
    std::vector<int32_t> first_of(un), next_same(un, -1);
    parallel_slices(SH, threads, [&](int64_t slo, int64_t shi, int) {
        std::vector<int32_t> tab_first, tab_last;
        for (int64_t s = slo; s < shi; s++) {
            const int64_t a = sh_off[(size_t)s], b = sh_off[(size_t)s + 1];
            size_t cap = 16;
            while (cap < 2 * (size_t)(b - a)) cap <<= 1;
            tab_first.assign(cap, -1);
            tab_last.assign(cap, -1);
            for (int64_t k = a; k < b; k++) {
                const int32_t i = order[(size_t)k];
                const uint8_t *p = d.data() + rec[(size_t)i];
                const size_t len = (size_t)p[8] - 1;
                size_t slot = (size_t)(hsh[(size_t)i] * 0x9E3779B97F4A7C15ULL >> 20) & (cap - 1);
                for (;;) {
                    const int32_t f = tab_first[slot];
                    if (f < 0) { tab_first[slot] = i; tab_last[slot] = i; first_of[(size_t)i] = i; break; }
                    const uint8_t *pf = d.data() + rec[(size_t)f];
                    if (hsh[(size_t)f] == hsh[(size_t)i] && (size_t)pf[8] - 1 == len && memcmp(pf + 32, p + 32, len) == 0) {
                        first_of[(size_t)i] = f;
                        next_same[(size_t)tab_last[slot]] = i;
                        tab_last[slot] = i;
                        break;
                    }
                    slot = (slot + 1) & (cap - 1);
                }
            }
        }
    });
    {
        uint32_t ids = 0;
        std::vector<uint32_t> id_at(un);
        for (int64_t i = 0; i < n; i++) {
            if (first_of[(size_t)i] == (int32_t)i) {
                id_at[(size_t)i] = ids++;
                B.name_at.push_back(rec[(size_t)i] + 32);
                B.name_len.push_back((uint8_t)(d[rec[(size_t)i] + 8] - 1));
            }
        }
        parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; i++) B.qname[(size_t)i] = id_at[(size_t)first_of[(size_t)i]];
        });
    }
    // AlignmentFile.mate(): the first record in file order with this name on the mate's contig that
    // overlaps the mate position and carries the other read-of-pair flag
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; i++) {
            int32_t m = -1;
            const uint16_t fl = B.flag[(size_t)i];
            if ((fl & FPAIRED) && !(fl & FMUNMAP) && mtid[(size_t)i] >= 0) {
                const uint16_t want = (uint16_t)((fl ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2));
                const int64_t mp = mpos[(size_t)i];
                for (int32_t j = first_of[(size_t)i]; j >= 0; j = next_same[(size_t)j]) {
                    if (tid[(size_t)j] != mtid[(size_t)i]) continue;
                    if (!((int64_t)B.start[(size_t)j] < mp + 1 && (int64_t)B.end[(size_t)j] > mp)) continue;
                    if (B.flag[(size_t)j] & want) { m = j; break; }
                }
            }
            B.mate[(size_t)i] = m;
        }
    });
    B.timing[3] = now_s() - t3;
    B.io_stats[3] = n;
}

// ---------------------------------------------------------------------------------------------------------
// Region decode through the BAI index: what `bamfile.fetch(contig, lo, hi)` (read_collector.py:385, :167, :478-497)
// and `bamfile.mate(read)` (:400, :185) hand the reference, without inflating the rest of the file.
struct Kept { // one kept record: where it starts in the file, its bytes (block_size word included)
    uint64_t voff;
    std::vector<uint8_t> bytes;
    bool direct = false; // returned by one of the fetches (not only as a mate candidate)
};

// walks the records of one chunk [beg, end) and keeps those `want(tid, pos, end, record)` accepts
template <typename W>
void walk_chunk(const FileRd &f, Chunk ck, W &&want, std::vector<Kept> &out, int64_t *file_bytes, int64_t *blocks, int64_t *walked) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) fail(UZ_IO_E_FORMAT, "zlib init failed");
    std::vector<uint8_t> cbuf;
    Inflated inf;
    try {
        int64_t coff = (int64_t)(ck.beg >> 16);
        if (!inflate_one(f, coff, inf, z, cbuf, file_bytes, blocks)) { inflateEnd(&z); return; }
        size_t at = (size_t)(ck.beg & 0xFFFF); // offset in inf.bytes of the next record
        size_t blk = 0;                        // block the next record starts in
        for (;;) {
            // virtual offset of the record at `at`
            while (blk + 1 < inf.block_at.size() && inf.block_at[blk + 1].second <= at) blk++;
            // a record starting exactly at the end of the last inflated block starts in the NEXT block
            while (at >= inf.bytes.size()) {
                if (!inflate_one(f, inf.next_coff, inf, z, cbuf, file_bytes, blocks)) { inflateEnd(&z); return; }
                while (blk + 1 < inf.block_at.size() && inf.block_at[blk + 1].second <= at) blk++;
            }
            const uint64_t voff = ((uint64_t)inf.block_at[blk].first << 16) | (uint64_t)(at - inf.block_at[blk].second);
            if (voff >= ck.end) break;
            while (at + 4 > inf.bytes.size())
                if (!inflate_one(f, inf.next_coff, inf, z, cbuf, file_bytes, blocks)) fail(UZ_IO_E_FORMAT, "truncated alignment record");
            const int32_t bs = rdi32(inf.bytes.data() + at);
            if (bs < 32) fail(UZ_IO_E_FORMAT, "bad alignment record at virtual offset %llu", (unsigned long long)voff);
            while (at + 4 + (size_t)bs > inf.bytes.size())
                if (!inflate_one(f, inf.next_coff, inf, z, cbuf, file_bytes, blocks)) fail(UZ_IO_E_FORMAT, "truncated alignment record");
            const uint8_t *p = inf.bytes.data() + at + 4;
            if (walked) *walked += 1;
            const int32_t tid = rdi32(p), pos = rdi32(p + 4);
            const uint32_t l_name = p[8], ncig = rd16(p + 12);
            const uint16_t fl = rd16(p + 14);
            int64_t rl = 0;
            const uint8_t *q = p + 32 + l_name;
            if (32 + (size_t)l_name + 4 * (size_t)ncig > (size_t)bs) fail(UZ_IO_E_FORMAT, "alignment record overruns its block");
            for (uint32_t k = 0; k < ncig; k++) {
                const uint32_t v = rd32(q + 4 * k), op = v & 15;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4;
            }
            const int32_t end = (fl & FUNMAP) || ncig == 0 ? pos + 1 : (int32_t)(pos + (rl > 0 ? rl : 1));
            if (tid >= 0 && want(tid, pos, end, p, (uint32_t)bs)) {
                Kept k;
                k.voff = voff;
                k.bytes.assign(inf.bytes.data() + at, inf.bytes.data() + at + 4 + (size_t)bs);
                out.push_back(std::move(k));
            }
            at += 4 + (size_t)bs;
        }
    } catch (...) { inflateEnd(&z); throw; }
    inflateEnd(&z);
}

void decode_regions(uz_bam &B, const char *path, const char *bai_path, int64_t n_iv, const int32_t *iv_tid, const int32_t *iv_lo,
                    const int32_t *iv_hi, int64_t head_records, int threads) {
    double t0 = now_s();
    threads = resolve_threads(threads);
    FileRd f(path);
    std::string bai = bai_path ? std::string(bai_path) : std::string(path) + ".bai";
    if (!bai_path) { // NAME.bam.bai or NAME.bai
        FILE *t = fopen(bai.c_str(), "rb");
        if (t) fclose(t);
        else { std::string alt(path); if (alt.size() > 4) alt = alt.substr(0, alt.size() - 4) + ".bai"; bai = alt; }
    }
    const std::vector<BaiRef> refs = read_bai(bai.c_str());
    int64_t file_bytes = 0, blocks = 0, walked_n = 0;
    // header: inflate from the start of the file until it parses
    int32_t n_ref = 0;
    std::vector<uint8_t> header;
    {
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) fail(UZ_IO_E_FORMAT, "zlib init failed");
        std::vector<uint8_t> cbuf;
        Inflated inf;
        size_t off = 0;
        try {
            int64_t coff = 0;
            for (;;) {
                if (!inflate_one(f, coff, inf, z, cbuf, &file_bytes, &blocks)) fail(UZ_IO_E_FORMAT, "truncated BAM header");
                coff = inf.next_coff;
                off = parse_header(B, inf.bytes.data(), inf.bytes.size(), path, n_ref, true);
                if (off) break;
            }
            // the first records of the FILE: estimate_concordant_insert_len reads them (read_collector.py:11-25)
            size_t at = off;
            while ((int64_t)B.tlen_file.size() < head_records) {
                bool eof = false;
                while (at + 4 > inf.bytes.size() && !eof) eof = !inflate_one(f, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks);
                if (at + 4 > inf.bytes.size()) break;
                const int32_t bs = rdi32(inf.bytes.data() + at);
                if (bs < 32) fail(UZ_IO_E_FORMAT, "bad alignment record in the head of the file");
                while (at + 4 + (size_t)bs > inf.bytes.size() && !eof) eof = !inflate_one(f, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks);
                if (at + 4 + (size_t)bs > inf.bytes.size()) break;
                B.tlen_file.push_back(rdi32(inf.bytes.data() + at + 4 + 28));
                at += 4 + (size_t)bs;
            }
        } catch (...) { inflateEnd(&z); throw; }
        inflateEnd(&z);
        header.assign(inf.bytes.begin(), inf.bytes.begin() + (ptrdiff_t)off);
    }
    if ((size_t)n_ref != refs.size()) fail(UZ_IO_E_FORMAT, "the index %s holds %zu references, the BAM header %d", bai.c_str(), refs.size(), n_ref);

    // The pool: every record of every chunk walked so far (file order).  Chunks are walked at most once: `walked` holds
    // the merged virtual-offset ranges already read, and a later request only reads what lies outside them.
    struct TidIvs { std::vector<Iv> ivs; int32_t max_len = 0; };
    std::vector<Kept> pool;
    std::vector<Chunk> walked;
    auto ensure_walked = [&](std::vector<TidIvs> &per_tid) {
        std::vector<Chunk> want;
        for (int32_t t = 0; t < n_ref; t++) {
            auto &ivs = per_tid[(size_t)t].ivs;
            if (ivs.empty()) continue;
            std::sort(ivs.begin(), ivs.end(), [](const Iv &x, const Iv &y) { return x.lo < y.lo || (x.lo == y.lo && x.hi < y.hi); });
            for (const Iv &iv : ivs) per_tid[(size_t)t].max_len = std::max(per_tid[(size_t)t].max_len, iv.hi - iv.lo);
            chunks_for(refs[(size_t)t], ivs, want);
        }
        std::sort(want.begin(), want.end(), [](const Chunk &x, const Chunk &y) { return x.beg < y.beg; });
        std::vector<Chunk> merged;
        for (const Chunk &c : want) {
            if (!merged.empty() && c.beg <= merged.back().end) merged.back().end = std::max(merged.back().end, c.end);
            else merged.push_back(c);
        }
        // minus what has been walked
        std::vector<Chunk> work;
        size_t w = 0;
        for (Chunk c : merged) {
            while (w < walked.size() && walked[w].end <= c.beg) w++;
            size_t k = w;
            while (c.beg < c.end) {
                if (k >= walked.size() || walked[k].beg >= c.end) { work.push_back(c); break; }
                if (walked[k].beg > c.beg) work.push_back(Chunk{c.beg, walked[k].beg});
                c.beg = std::max(c.beg, walked[k].end);
                k++;
            }
        }
        if (work.empty()) return;
        std::vector<std::vector<Kept>> found(work.size());
        const int wk = workers_for((int64_t)work.size(), threads, 1);
        std::vector<int64_t> fb((size_t)wk, 0), bl((size_t)wk, 0), wa((size_t)wk, 0);
        parallel_slices((int64_t)work.size(), wk, [&](int64_t lo, int64_t hi, int wi) {
            for (int64_t k = lo; k < hi; k++)
                walk_chunk(f, work[(size_t)k], [](int32_t, int32_t, int32_t, const uint8_t *, uint32_t) { return true; }, found[(size_t)k],
                           &fb[(size_t)wi], &bl[(size_t)wi], &wa[(size_t)wi]);
        });
        for (int k = 0; k < wk; k++) { file_bytes += fb[(size_t)k]; blocks += bl[(size_t)k]; walked_n += wa[(size_t)k]; }
        for (auto &v : found) for (auto &k : v) pool.push_back(std::move(k));
        std::sort(pool.begin(), pool.end(), [](const Kept &x, const Kept &y) { return x.voff < y.voff; });
        for (const Chunk &c : work) walked.push_back(c);
        std::sort(walked.begin(), walked.end(), [](const Chunk &x, const Chunk &y) { return x.beg < y.beg; });
        std::vector<Chunk> m2;
        for (const Chunk &c : walked) {
            if (!m2.empty() && c.beg <= m2.back().end) m2.back().end = std::max(m2.back().end, c.end);
            else m2.push_back(c);
        }
        walked.swap(m2);
    };
    struct Info { int32_t tid, pos, end, mtid, mpos; uint16_t flag; uint64_t h; };
    auto info_of = [](const Kept &k) {
        const uint8_t *p = k.bytes.data() + 4;
        const uint32_t l_name = p[8], ncig = rd16(p + 12);
        Info x;
        x.tid = rdi32(p); x.pos = rdi32(p + 4); x.flag = rd16(p + 14); x.mtid = rdi32(p + 20); x.mpos = rdi32(p + 24);
        int64_t rl = 0;
        const uint8_t *q = p + 32 + l_name;
        for (uint32_t c = 0; c < ncig; c++) { const uint32_t v = rd32(q + 4 * c), op = v & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4; }
        x.end = (x.flag & FUNMAP) || ncig == 0 ? x.pos + 1 : (int32_t)(x.pos + (rl > 0 ? rl : 1));
        x.h = hash_bytes(p + 32, (size_t)l_name - 1);
        return x;
    };

    // the fetches themselves
    std::vector<TidIvs> per_tid((size_t)n_ref);
    for (int64_t k = 0; k < n_iv; k++)
        if (iv_tid[k] >= 0 && iv_tid[k] < n_ref && iv_hi[k] > iv_lo[k]) per_tid[(size_t)iv_tid[k]].ivs.push_back(Iv{iv_lo[k], iv_hi[k]});
    ensure_walked(per_tid);
    // members of the result, by virtual offset; frontier: members whose mate has not been looked up yet
    std::vector<uint64_t> member, frontier;
    for (const Kept &k : pool) {
        const uint8_t *p = k.bytes.data() + 4;
        const int32_t tid = rdi32(p);
        if (tid < 0 || tid >= n_ref) continue;
        const auto &ivs = per_tid[(size_t)tid].ivs;
        if (ivs.empty()) continue;
        const Info x = info_of(k);
        auto a = std::lower_bound(ivs.begin(), ivs.end(), (int64_t)x.pos - per_tid[(size_t)tid].max_len,
                                  [](const Iv &v, int64_t key) { return (int64_t)v.lo < key; });
        bool hit = false;
        for (; a != ivs.end() && a->lo < x.end && !hit; ++a) hit = a->hi > x.pos;
        if (hit) { member.push_back(k.voff); frontier.push_back(k.voff); }
    }
    // mate(): the FIRST record in file order with the name that overlaps the mate position on the mate's contig and carries
    // the other read-of-pair flag (it may be a secondary / supplementary record).  Transitively: the kernel follows
    // mate(mate(r)) for the reads at the DNM.  Generation by generation, so that the chunks a generation needs are
    // read together.
    for (int gen = 0; gen < 64 && !frontier.empty(); gen++) {
        auto at = [&](uint64_t voff) { return (size_t)(std::lower_bound(pool.begin(), pool.end(), voff, [](const Kept &k, uint64_t key) { return k.voff < key; }) - pool.begin()); };
        std::vector<TidIvs> mate_iv((size_t)n_ref);
        for (uint64_t v : frontier) {
            const Info x = info_of(pool[at(v)]);
            if ((x.flag & FPAIRED) && !(x.flag & FMUNMAP) && x.mtid >= 0 && x.mtid < n_ref) mate_iv[(size_t)x.mtid].ivs.push_back(Iv{x.mpos, x.mpos + 1});
        }
        ensure_walked(mate_iv); // the pool may grow (and re-sort): look positions up again below
        std::vector<std::pair<uint64_t, int32_t>> by_name(pool.size());
        std::vector<Info> info(pool.size());
        for (size_t i = 0; i < pool.size(); i++) { info[i] = info_of(pool[i]); by_name[i] = {info[i].h, (int32_t)i}; }
        std::sort(by_name.begin(), by_name.end());
        std::vector<uint64_t> next;
        for (uint64_t v : frontier) {
            const size_t i = at(v);
            const Info &x = info[i];
            if (!(x.flag & FPAIRED) || (x.flag & FMUNMAP) || x.mtid < 0 || x.mtid >= n_ref) continue;
            const uint16_t want = (uint16_t)((x.flag ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2));
            const uint8_t *pi = pool[i].bytes.data() + 4;
            auto it = std::lower_bound(by_name.begin(), by_name.end(), std::make_pair(x.h, (int32_t)-1));
            for (; it != by_name.end() && it->first == x.h; ++it) {
                const size_t j = (size_t)it->second;
                const Info &y = info[j];
                const uint8_t *pj = pool[j].bytes.data() + 4;
                if (pj[8] != pi[8] || memcmp(pj + 32, pi + 32, (size_t)pi[8] - 1) != 0) continue;
                if (y.tid != x.mtid) continue;
                if (!((int64_t)y.pos < (int64_t)x.mpos + 1 && (int64_t)y.end > (int64_t)x.mpos)) continue;
                if (y.flag & want) { next.push_back(pool[j].voff); break; }
            }
        }
        std::sort(member.begin(), member.end());
        std::sort(next.begin(), next.end());
        next.erase(std::unique(next.begin(), next.end()), next.end());
        frontier.clear();
        for (uint64_t v : next)
            if (!std::binary_search(member.begin(), member.end(), v)) frontier.push_back(v);
        for (uint64_t v : frontier) member.push_back(v);
    }
    std::sort(member.begin(), member.end());
    std::vector<Kept> kept;
    for (Kept &k : pool)
        if (std::binary_search(member.begin(), member.end(), k.voff)) kept.push_back(std::move(k));
    pool.clear();
    double t1 = now_s();
    B.timing[0] = 0;
    B.timing[1] = t1 - t0;
    // the table is built from a stream holding the header and the kept records only
    size_t total = header.size();
    for (const Kept &k : kept) total += k.bytes.size();
    B.data.alloc(total);
    memcpy(B.data.data(), header.data(), header.size());
    std::vector<uint64_t> rec;
    std::vector<uint32_t> rec_end32;
    size_t at = header.size();
    for (const Kept &k : kept) {
        memcpy(B.data.data() + at, k.bytes.data(), k.bytes.size());
        rec.push_back(at + 4);
        rec_end32.push_back((uint32_t)(k.bytes.size() - 4));
        at += k.bytes.size();
    }
    B.n_file = (int64_t)B.tlen_file.size();
    B.io_stats[0] = file_bytes; B.io_stats[1] = blocks; B.io_stats[2] = walked_n;
    kept.clear();
    build_table(B, rec, rec_end32, n_ref, threads, t1);
}

template <typename F>
int guarded(F fn) {
    try {
        fn();
        return UZ_IO_OK;
    } catch (const IoError &e) {
        last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        last_error = "out of memory";
        return UZ_IO_E_RANGE;
    } catch (const std::exception &e) {
        last_error = e.what();
        return UZ_IO_E_FORMAT;
    }
}

} // namespace

extern "C" {

const char *uz_io_last_error(void) { return last_error.c_str(); }

int uz_bam_decode(const char *path, int threads, uz_bam **out) {
    if (!path || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_bam *h = nullptr;
    const int rc = guarded([&] {
        h = new uz_bam();
        decode(*h, path, resolve_threads(threads));
    });
    if (rc != UZ_IO_OK) { delete h; return rc; }
    *out = h;
    return UZ_IO_OK;
}

int uz_bam_decode_memory(const uint8_t *stream, int64_t n, int threads, uz_bam **out) {
    if (!stream || n < 0 || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_bam *h = nullptr;
    const int rc = guarded([&] {
        h = new uz_bam();
        h->data.alloc((size_t)n);
        memcpy(h->data.data(), stream, (size_t)n);
        decode_stream(*h, "<memory>", resolve_threads(threads), now_s());
    });
    if (rc != UZ_IO_OK) { delete h; return rc; }
    *out = h;
    return UZ_IO_OK;
}

int uz_bam_decode_regions(const char *path, const char *bai_path, int64_t n_iv, const int32_t *tid, const int32_t *lo, const int32_t *hi,
                          int64_t head_records, int threads, uz_bam **out) {
    if (!path || !out || (n_iv > 0 && (!tid || !lo || !hi))) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_bam *b = new uz_bam();
    const int rc = guarded([&] { decode_regions(*b, path, bai_path, n_iv, tid, lo, hi, head_records, threads); });
    if (rc != UZ_IO_OK) { delete b; return rc; }
    *out = b;
    return UZ_IO_OK;
}

void uz_bam_io_stats(const uz_bam *h, int64_t out[4]) {
    for (int k = 0; k < 4; k++) out[k] = h ? h->io_stats[k] : 0;
}

void uz_bam_free(uz_bam *h) { delete h; }

int32_t uz_bam_n_contigs(const uz_bam *h) { return h ? (int32_t)h->contigs.size() : 0; }
const char *uz_bam_contig_name(const uz_bam *h, int32_t i) {
    return (h && i >= 0 && (size_t)i < h->contigs.size()) ? h->contigs[(size_t)i].c_str() : nullptr;
}
int32_t uz_bam_contig_length(const uz_bam *h, int32_t i) {
    return (h && i >= 0 && (size_t)i < h->contig_len.size()) ? h->contig_len[(size_t)i] : -1;
}
int64_t uz_bam_n_file_records(const uz_bam *h) { return h ? h->n_file : 0; }
int64_t uz_bam_n_records(const uz_bam *h) { return h ? h->n : 0; }

int uz_bam_view(const uz_bam *h, uz_reads_view *v) {
    if (!h || !v) { last_error = "null argument"; return UZ_IO_E_ARG; }
    memset(v, 0, sizeof(*v));
    v->n_segs = h->n;
    v->n_contigs = (int32_t)h->contigs.size();
    v->contig_off = h->contig_off.data();
    v->max_span = h->max_span.data();
    v->start = h->start.p; v->end = h->end.p; v->flag = h->flag.p; v->mapq = h->mapq.p; v->aux = h->aux.p;
    v->tlen = h->tlen.p; v->qname = h->qname.p; v->mate = h->mate.p;
    v->cigar_off = h->cigar_off.p; v->n_cigar = h->n_cigar.p; v->cigar = h->cigar.p;
    v->l_seq = h->l_seq.p; v->sq_off16 = h->sq_off16.p; v->seq = h->seq.p; v->qual = h->qual.p;
    v->n_cigar_total = h->n_cigar_total;
    v->n_sq_bytes = h->n_sq_bytes;
    v->n_qnames = (uint32_t)h->name_at.size();
    return UZ_IO_OK;
}

const char *uz_bam_qname(const uz_bam *h, uint32_t id, int32_t *len) {
    if (!h || id >= h->name_at.size()) return nullptr;
    if (len) *len = h->name_len[id];
    return (const char *)h->data.data() + h->name_at[id]; // NUL-terminated inside the record
}

int64_t uz_bam_tlen_head(const uz_bam *h, int32_t *out, int64_t cap) {
    if (!h || !out || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, (int64_t)h->tlen_file.size());
    memcpy(out, h->tlen_file.data(), (size_t)k * sizeof(int32_t));
    return k;
}

void uz_bam_timing(const uz_bam *h, double out[4]) {
    for (int k = 0; k < 4; k++) out[k] = h ? h->timing[k] : 0.0;
}

} // extern "C"
