// io_stage.cpp -- BAM file -> the staged (link) form of the alignment records in ONE pass (uz_bam_stage_*).
//
// What it replaces: the three passes the session used to make per batch -- uz_bam_decode_regions (BAI-driven decode
// into the ASCII column table: one byte per base and per quality), uz_reads_pack (ASCII -> packed table) and
// uz_reads_select_* (fetch reach + link-form columns) -- i.e. what `bamfile.fetch(chrom, lo, hi)` and
// `bamfile.mate(read)` hand the reference per DNM (read_collector.py:385, :167, :400, :185), delivered as the columns
// uz_reads_upload_packed takes.  Here a batch's fetch points are merged into reach intervals, each interval's BGZF
// blocks (BAI bins + linear index) are inflated once by one worker, and the worker turns the records it walks
// straight into what the link carries: 2-bit bases of the 32-base units a fetch point falls into, the low-quality
// positions inside those units, CIGAR words of the records that are not one M / = / X over the read, and the small
// fixed columns.  No intermediate table; what is not needed is never copied out of the inflate buffer.
//
// Equivalence: the output is byte for byte what ReadsSource(pack_reads(read_bam_regions(...))).select(...) builds for
// the same fetches (tests/test_io_stage.py) -- same records (fetched + closed under mate()), same order, same name
// ids (order of first appearance), same dictionary / escape / exception lists.
//
// Inflate: libdeflate when the system has it (dlopen, no headers needed: three functions of its stable ABI), else
// zlib.  UZ_INFLATE=zlib forces zlib.
#include <sys/mman.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "io_common.hpp"
#include "io_index.hpp"
#include "pack.hpp"
#include "uz_bamwalk.h"

using namespace uzio;

namespace {

inline uint64_t hash_name(const uint8_t *s, size_t n) { // FNV-1a with a final mix
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ULL; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ULL; h ^= h >> 29;
    return h;
}

const uint16_t FPAIRED = 1, FUNMAP = 4, FMUNMAP = 8, FREAD1 = 64, FREAD2 = 128;
const int32_t REACH_SLACK_DEFAULT = 1000; // a reach interval extends this far beyond its fetch points: the mates of a pile-up lie inside it

bool has_sa_tag(const uint8_t *p, const uint8_t *end) {
    while (p + 3 <= end) {
        const uint8_t a = p[0], b = p[1], typ = p[2];
        p += 3;
        if (a == 'S' && b == 'A') return true;
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

} // namespace

// ------------------------------------------------------------------------------------------------ the opened file
struct uz_bamsrc {
    std::string path;
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0;
    std::vector<BaiRef> refs;
    std::vector<std::string> contigs;
    std::vector<int32_t> contig_len;
    std::vector<int32_t> tlen_head;
    ~uz_bamsrc() {
        if (map && size) munmap(const_cast<uint8_t *>(map), size);
        if (fd >= 0) close(fd);
    }
};

namespace {

struct BlockHdr { size_t cdata, clen, blen; uint32_t crc, isize; };

// header of the BGZF block at compressed offset coff; false at the end of the file
bool block_at(const uz_bamsrc &S, int64_t coff, BlockHdr &b) {
    if (coff < 0 || (size_t)coff + 18 > S.size) return false;
    const uint8_t *h = S.map + coff;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) fail(UZ_IO_E_FORMAT, "not a BGZF block at byte %lld", (long long)coff);
    const size_t xlen = rd16(h + 10);
    if ((size_t)coff + 12 + xlen > S.size) fail(UZ_IO_E_FORMAT, "truncated BGZF block");
    size_t q = 12, bsize = 0;
    bool found = false;
    while (q + 4 <= 12 + xlen) {
        const size_t slen = rd16(h + q + 2);
        if (q + 4 + slen > 12 + xlen) fail(UZ_IO_E_FORMAT, "BGZF extra subfield overruns the extra field at byte %lld", (long long)coff);
        if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) { bsize = rd16(h + q + 4); found = true; }
        q += 4 + slen;
    }
    if (!found) fail(UZ_IO_E_FORMAT, "BGZF block without a BC field at byte %lld", (long long)coff);
    b.blen = bsize + 1;
    if (b.blen < 12 + xlen + 8 || (size_t)coff + b.blen > S.size) fail(UZ_IO_E_FORMAT, "truncated BGZF block at byte %lld", (long long)coff);
    b.cdata = (size_t)coff + 12 + xlen;
    b.clen = b.blen - 12 - xlen - 8;
    b.crc = rd32(h + b.blen - 8);
    b.isize = rd32(h + b.blen - 4);
    return true;
}

// A sequential reader of the inflated stream from a virtual offset on: inflates block after block into a rolling buffer and
// hands out whole records with their virtual offsets.
// blocks somebody else has inflated already (uz_stage_set_inflated: the device): where block `coff` lies in that buffer
struct PreBlk { int64_t coff; int64_t at; uint32_t isize, crc; uint32_t blen /* bytes of the block in the file */, hdr /* ... of its header: the DEFLATE stream starts behind them */; };
struct PreDir {
    const std::vector<PreBlk> *blks = nullptr; // ascending coff
    const uint8_t *base = nullptr;
    size_t hint = 0;
    const PreBlk *find(int64_t coff) {
        if (!blks || !base) return nullptr;
        const std::vector<PreBlk> &v = *blks;
        if (hint < v.size() && v[hint].coff == coff) return &v[hint++];
        auto it = std::lower_bound(v.begin(), v.end(), coff, [](const PreBlk &b, int64_t key) { return b.coff < key; });
        if (it == v.end() || it->coff != coff) return nullptr;
        hint = (size_t)(it - v.begin()) + 1;
        return &*it;
    }
};

struct Stream {
    const uz_bamsrc &S;
    Inflater &inf;
    PreDir pre;
    std::vector<uint8_t> own_buf;
    std::vector<uint8_t> &buf; // (a worker hands in its own buffer, kept from task to task: no fresh pages per task)
    struct Blk { int64_t coff; size_t at, isize; };
    std::vector<Blk> blks; // blocks held in buf (buffer offsets)
    size_t cur = 0;        // next unread byte of buf
    size_t blk = 0;        // block holding `cur`
    int64_t next_coff = 0;
    int64_t file_bytes = 0, n_blocks = 0, n_pre = 0;
    bool eof = false;
    Stream(const uz_bamsrc &s, Inflater &i) : S(s), inf(i), buf(own_buf) {}
    Stream(const uz_bamsrc &s, Inflater &i, std::vector<uint8_t> &b) : S(s), inf(i), buf(b) {}
    void seek(uint64_t voff) {
        buf.clear(); blks.clear(); cur = 0; blk = 0; eof = false;
        next_coff = (int64_t)(voff >> 16);
        if (!more()) { // exactly the end of the file: an empty stream (next() says so); anywhere else the index does not belong to this file
            if ((voff >> 16) != (uint64_t)S.size || (voff & 0xFFFF)) fail(UZ_IO_E_FORMAT, "virtual offset %llu points past the end of the file (a stale or truncated index?)", (unsigned long long)voff);
            return;
        }
        cur = (size_t)(voff & 0xFFFF);
        if (cur > blks[0].isize) fail(UZ_IO_E_FORMAT, "virtual offset beyond its block");
    }
    bool more() { // one more block; false at the end of the file
        if (eof) return false;
        if (cur > (1u << 20)) { // drop what has been consumed
            size_t b0 = blk;
            while (b0 > 0 && blks[b0].at > cur) b0--;
            const size_t cut = blks[b0].at;
            if (cut) {
                memmove(buf.data(), buf.data() + cut, buf.size() - cut);
                buf.resize(buf.size() - cut);
                cur -= cut;
                blks.erase(blks.begin(), blks.begin() + (ptrdiff_t)b0);
                for (auto &x : blks) x.at -= cut;
                blk -= b0;
            }
        }
        BlockHdr h;
        if (!block_at(S, next_coff, h)) { eof = true; return false; }
        if (h.isize > 65536u) fail(UZ_IO_E_FORMAT, "BGZF block at byte %lld declares %u inflated bytes (a block holds at most 65536)", (long long)next_coff, h.isize);
        const size_t at = buf.size();
        buf.resize(at + h.isize);
        if (const PreBlk *pb = pre.find(next_coff)) { // inflated elsewhere: its bytes are copied in and held against the block's checksum
            if (pb->isize != h.isize) fail(UZ_IO_E_FORMAT, "pre-inflated BGZF block at byte %lld has another size", (long long)next_coff);
            if (h.isize) {
                memcpy(buf.data() + at, pre.base + pb->at, h.isize);
                if (inf.crc_of(buf.data() + at, h.isize) != h.crc) fail(UZ_IO_E_FORMAT, "CRC mismatch in the pre-inflated BGZF block at byte %lld", (long long)next_coff);
            }
            n_pre++;
        } else
            inf.block(S.map + h.cdata, h.clen, buf.data() + at, h.isize, h.crc, next_coff);
        blks.push_back(Blk{next_coff, at, h.isize});
        next_coff += (int64_t)h.blen;
        file_bytes += (int64_t)h.blen;
        n_blocks++;
        return true;
    }
    // the record at the cursor: false at the end of the file.  *voff: where it starts; p: its fixed part; bs: its block_size
    bool next(uint64_t &voff, const uint8_t *&p, uint32_t &bs) {
        for (;;) { // the cursor at the end of a block is the start of the next one
            if (blks.empty()) { if (!more()) return false; continue; }
            while (blk + 1 < blks.size() && cur >= blks[blk].at + blks[blk].isize) blk++;
            if (cur < blks[blk].at + blks[blk].isize) break;
            if (!more()) return false;
        }
        voff = ((uint64_t)blks[blk].coff << 16) | (uint64_t)(cur - blks[blk].at);
        while (cur + 4 > buf.size()) if (!more()) fail(UZ_IO_E_FORMAT, "truncated alignment record");
        const int32_t n = rdi32(buf.data() + cur);
        if (n < 32) fail(UZ_IO_E_FORMAT, "bad alignment record at virtual offset %llu", (unsigned long long)voff);
        while (cur + 4 + (size_t)n > buf.size()) if (!more()) fail(UZ_IO_E_FORMAT, "truncated alignment record");
        p = buf.data() + cur + 4;
        bs = (uint32_t)n;
        return true;
    }
    void advance(uint32_t bs) { cur += 4 + (size_t)bs; }
    Stream(const Stream &) = delete;
};

void open_source(uz_bamsrc &S, const char *path, const char *bai_path, int64_t head_records) {
    S.path = path;
    S.fd = open(path, O_RDONLY);
    if (S.fd < 0) fail(UZ_IO_E_OPEN, "cannot open %s", path);
    struct stat st;
    if (fstat(S.fd, &st) != 0) fail(UZ_IO_E_OPEN, "cannot stat %s", path);
    S.size = (size_t)st.st_size;
    if (S.size) {
        void *m = mmap(nullptr, S.size, PROT_READ, MAP_SHARED, S.fd, 0);
        if (m == MAP_FAILED) fail(UZ_IO_E_OPEN, "cannot map %s", path);
        S.map = (const uint8_t *)m;
    }
    std::string bai = bai_path ? std::string(bai_path) : std::string(path) + ".bai";
    if (!bai_path) { // NAME.bam.bai or NAME.bai
        FILE *t = fopen(bai.c_str(), "rb");
        if (t) fclose(t);
        else { std::string alt(path); if (alt.size() > 4) alt = alt.substr(0, alt.size() - 4) + ".bai"; bai = alt; }
    }
    S.refs = read_bai(bai.c_str());
    // header + the head of the file (estimate_concordant_insert_len reads the first records, read_collector.py:11-25)
    Inflater inf;
    Stream s(S, inf);
    s.seek(0);
    for (;;) {
        if (s.buf.size() >= 12) {
            if (memcmp(s.buf.data(), "BAM\1", 4) != 0) fail(UZ_IO_E_FORMAT, "%s is not a BAM file", path);
            const int32_t l_text = rdi32(s.buf.data() + 4);
            if (l_text < 0) fail(UZ_IO_E_FORMAT, "bad BAM header");
            size_t off = 8 + (size_t)l_text;
            bool short_ = off + 4 > s.buf.size();
            int32_t n_ref = 0;
            if (!short_) {
                n_ref = rdi32(s.buf.data() + off);
                if (n_ref < 0) fail(UZ_IO_E_FORMAT, "bad BAM header");
                off += 4;
                S.contigs.clear(); S.contig_len.clear();
                for (int32_t r = 0; r < n_ref && !short_; r++) {
                    if (off + 4 > s.buf.size()) { short_ = true; break; }
                    const int32_t l_name = rdi32(s.buf.data() + off);
                    if (l_name < 1) fail(UZ_IO_E_FORMAT, "bad BAM header");
                    if (off + 4 + (size_t)l_name + 4 > s.buf.size()) { short_ = true; break; }
                    S.contigs.emplace_back((const char *)s.buf.data() + off + 4, (size_t)l_name - 1);
                    S.contig_len.push_back(rdi32(s.buf.data() + off + 4 + (size_t)l_name));
                    off += 4 + (size_t)l_name + 4;
                }
            }
            if (!short_) { s.cur = off; break; }
        }
        if (!s.more()) fail(UZ_IO_E_FORMAT, "truncated BAM header");
    }
    if (S.contigs.size() != S.refs.size()) fail(UZ_IO_E_FORMAT, "the index %s holds %zu references, the BAM header %zu", bai.c_str(), S.refs.size(), S.contigs.size());
    s.blk = 0;
    while ((int64_t)S.tlen_head.size() < head_records) {
        uint64_t voff;
        const uint8_t *p;
        uint32_t bs;
        if (!s.next(voff, p, bs)) break;
        S.tlen_head.push_back(rdi32(p + 28));
        s.advance(bs);
    }
}

// ------------------------------------------------------------------------------------------------ the plan
struct Fx { int32_t lo, hi; uint16_t extra; };

struct WRec { // one walked record that may be kept
    uint64_t voff, nhash;
    int64_t mate_ref;  // (task << 32 | index) of the record mate() returns, -1 none, -2 not looked up yet
    int32_t pos, end, tlen, mpos, mtid;
    uint32_t name_at, cigar_at, pay_at;
    uint32_t gidx, qid;
    uint16_t flag, l_seq, n_cigar, umask, n_exc, tup;
    uint8_t mapq, aux, n_low_full, l_name, n_qpos, n_low, simple, keep, n_units, has_pay;
    uint8_t n_bl, bl_units; // bases as a list (uz_types.h bl_*): the listed bases and the units of the mask they lie in (n_units is then 0: no rows travel)
    // the descriptor route (uz_bam_stage_finish_desc): where the record's fixed part lies (uz_walk_desc.src; a record the HOST walked there keeps its
    // bytes in Task::raw: pay_at = offset of its block_size field, cigar_at = 4 + block_size) and the second name hash
    uint64_t src;
    uint32_t nhash2;
};

struct Task { // reach intervals of one reference whose file spans meet (walked as one: no block is inflated twice), and what the walk kept
    int32_t tid = 0, a = 0, b = 0;   // first start / last end of the reach intervals
    std::vector<std::pair<int32_t, int32_t>> reach; // the intervals [a_i, b_i), ascending, disjoint
    size_t f0 = 0, f1 = 0;           // its fetches: [f0, f1) of the reference's sorted list
    std::vector<Chunk> spans;
    uint64_t est_end = 0;            // where the walk will probably stop (linear index of the window behind b): the spans of the wide bins reach much further
    std::vector<WRec> recs;          // file order
    std::vector<uint8_t> names;      // name bytes (no terminator)
    std::vector<uint32_t> cigars;    // words of the records that are not simple
    std::vector<uint8_t> pay;        // per record with has_pay: seq2 units | listed bases (pos u16, two-bit code) | exceptions (pos u16, code u8, pad) | positions (u16) | plane row
    int64_t n_walked = 0, file_bytes = 0, n_blocks = 0, n_pre = 0;
    std::vector<PreBlk> pre;         // blocks of this task inflated elsewhere (uz_stage_gather_blocks / uz_stage_set_inflated)
    std::vector<int64_t> span_stop;  // per span: the file offset of the last block gathered for it (uz_stage_gather_blocks), -1 none
    bool by_hash = false;            // descriptor route: no name bytes on the host, two names are equal when both hashes and the lengths agree
    std::vector<uint8_t> raw;        // descriptor route: the bytes (block_size field included) of the kept records of a task the host walked itself
    // sizes of the kept records (filled by the numbering pass)
    int64_t n_keep = 0, k0 = 0;
};

struct Opt { bool all_bases, masks, lists, wide_none, bl; int thr; };

inline uint8_t sat255(int v) { return (uint8_t)(v > 255 ? 255 : v); }

// everything the link can need of one record, extracted while its bytes are in the inflate buffer
void extract(Task &T, WRec &r, const uint8_t *p, uint32_t bs, const Opt &o, bool bases, const std::vector<uint16_t> *bl = nullptr) {
    const uint32_t l_name = p[8], ncig = r.n_cigar, L = r.l_seq;
    const uint8_t *q = p + 32 + l_name;
    const uint8_t *sq = q + 4 * (size_t)ncig;
    const uint8_t *ql = sq + ((size_t)L + 1) / 2;
    const uint8_t *tags = ql + L;
    if (l_name < 1 || tags > p + bs) fail(UZ_IO_E_FORMAT, "alignment record overruns its block");
    r.name_at = (uint32_t)T.names.size();
    r.l_name = (uint8_t)(l_name - 1);
    T.names.insert(T.names.end(), p + 32, p + 32 + l_name - 1);
    uint8_t a = 0;
    if (r.mtid == T.tid) a |= UZ_AUX_MATE_SAME_TID;
    if (has_sa_tag(tags, p + bs)) a |= UZ_AUX_HAS_SA;
    const bool noqual = L > 0 && ql[0] == 0xFF;
    if (ncig == 0 || L == 0 || noqual) a |= UZ_AUX_DECODE_BAD;
    r.aux = a;
    r.simple = (uint8_t)uz_cigar_simple_code(ncig, ncig ? rd32(q) : 0u, L);
    r.cigar_at = (uint32_t)T.cigars.size();
    if (!r.simple)
        for (uint32_t k = 0; k < ncig; k++) T.cigars.push_back(rd32(q + 4 * k));
    // low-quality bases: the count of every record (a record without qualities decodes to zeros: every base is below a positive threshold)
    int low = 0;
    if (noqual) low = o.thr > 0 ? (int)L : 0;
    else for (uint32_t k = 0; k < L; k++) low += (int)ql[k] < o.thr;
    r.n_low_full = sat255(low);
    r.n_low = r.n_low_full;
    r.has_pay = 0; r.n_units = 0; r.n_exc = 0; r.n_qpos = 0; r.n_bl = 0; r.bl_units = 0;
    r.pay_at = (uint32_t)T.pay.size();
    const uint32_t units = UZ_ROW_UNITS(L);
    if (bases) {
        r.has_pay = 1;
        // the staged units: two bits per base, first base of a byte in bits 7-6; a base that is not A/C/G/T is 0 here and listed
        static const struct Tab { uint8_t t[256]; Tab() { for (int v = 0; v < 256; v++) { auto c = [](int n) { return n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : n == 8 ? 3 : 0; }; t[v] = (uint8_t)((c(v >> 4) << 2) | c(v & 15)); } } } two;
        const uint16_t m16 = r.umask;
        if (bl) { // the listed bases instead of the units they lie in (a base that is not A/C/G/T: code 0 here, and in the exception list below)
            for (uint16_t k : *bl) {
                const uint32_t c = (k & 1) ? (uint32_t)(sq[k >> 1] & 15u) : (uint32_t)(sq[k >> 1] >> 4);
                const uint8_t e[3] = {(uint8_t)(k & 255), (uint8_t)(k >> 8), (uint8_t)(c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 0)};
                T.pay.insert(T.pay.end(), e, e + 3);
            }
            r.n_bl = (uint8_t)bl->size();
            r.bl_units = (uint8_t)__builtin_popcount(m16);
        }
        for (uint32_t u = 0; u < units && !bl; u++) {
            if (m16 != UZ_UMASK_ALL && !((m16 >> u) & 1u)) continue;
            uint8_t row[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t b0 = 32 * u, b1 = std::min<uint32_t>(L, b0 + 32);
            const uint32_t nb = (b1 - b0 + 1) / 2; // BAM bytes of the unit
            const uint8_t *s = sq + b0 / 2;
            for (uint32_t j = 0; j < nb; j++) {
                uint8_t v = s[j];
                if (b0 + 2 * j + 1 >= L) v &= 0xF0; // the pad nibble of an odd length
                row[j >> 1] |= (uint8_t)(two.t[v] << ((j & 1) ? 0 : 4));
            }
            T.pay.insert(T.pay.end(), row, row + 8);
            r.n_units++;
        }
        // every base of the record that is not A/C/G/T (entries in units that stay home travel along: the device skips them)
        for (uint32_t k = 0; k < L; k++) {
            const uint32_t c = (k & 1) ? (uint32_t)(sq[k >> 1] & 15u) : (uint32_t)(sq[k >> 1] >> 4);
            if (c == 1 || c == 2 || c == 4 || c == 8) continue;
            const uint16_t pos16 = (uint16_t)k;
            const uint8_t e[4] = {(uint8_t)(pos16 & 255), (uint8_t)(pos16 >> 8), (uint8_t)c, 0};
            T.pay.insert(T.pay.end(), e, e + 4);
            r.n_exc++;
        }
        if (o.lists && low <= UZ_QLOW_LIST_MAX) { // the listed positions, inside the staged units only
            int kept = 0;
            for (uint32_t k = 0; k < L; k++) {
                const bool lowq = noqual ? o.thr > 0 : (int)ql[k] < o.thr;
                if (!lowq) continue;
                if (m16 != UZ_UMASK_ALL && !((m16 >> (k >> 5)) & 1u)) continue;
                const uint8_t e[2] = {(uint8_t)(k & 255), (uint8_t)(k >> 8)};
                T.pay.insert(T.pay.end(), e, e + 2);
                kept++;
            }
            r.n_qpos = (uint8_t)kept;
            r.n_low = (uint8_t)kept;
        }
    }
    if (!o.lists) { // the plane row of every record
        r.has_pay = 1;
        const size_t at = T.pay.size();
        T.pay.resize(at + (size_t)units * UZ_QLOW_UNIT_BYTES, 0);
        for (uint32_t k = 0; k < L; k++)
            if (noqual ? o.thr > 0 : (int)ql[k] < o.thr) T.pay[at + (k >> 3)] |= (uint8_t)(1u << (k & 7));
    }
}

inline int32_t endpos_of(const uint8_t *p, int32_t pos, uint16_t fl, uint32_t ncig, uint32_t l_name) {
    if ((fl & FUNMAP) || ncig == 0) return pos + 1;
    int64_t rl = 0;
    const uint8_t *q = p + 32 + l_name;
    for (uint32_t k = 0; k < ncig; k++) {
        const uint32_t v = rd32(q + 4 * k), op = v & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += v >> 4;
    }
    return (int32_t)(pos + (rl > 0 ? rl : 1));
}

// the unit-mask bits one fetch contributes to a record it returns (uz_reads_select_plan states the rule)
inline uint16_t mask_bits(const Fx &f, int32_t pos, int32_t end, uint32_t ncig, uint32_t cw, uint32_t L, bool wide_none) {
    if (wide_none && f.hi - f.lo > 2) return 0;
    const uint32_t op = cw & 15u;
    const bool simple = ncig == 1 && (op == 0 || op == 7 || op == 8) && (cw >> 4) == L && L > 1 && (uint32_t)(end - pos) == L;
    if (!(simple && L <= 480 && f.hi - f.lo <= 2)) return (uint16_t)UZ_UMASK_ALL;
    const int64_t q0 = (int64_t)f.hi - 1 - pos;
    uint16_t bits = 0;
    if (q0 >= 0 && q0 < (int64_t)L) {
        const int64_t q1 = std::min<int64_t>(q0 + f.extra, (int64_t)L - 1);
        for (int64_t u = q0 >> 5; u <= (q1 >> 5); u++) bits |= (uint16_t)(1u << u);
    }
    return bits;
}

// the bases one fetch asks of a record it returns, as query indices appended to `out`; false: the fetch's rule for this record is
// "every unit" (mask_bits gives UZ_UMASK_ALL) -- the record cannot travel as a list
inline bool list_bits(const Fx &f, int32_t pos, int32_t end, uint32_t ncig, uint32_t cw, uint32_t L, bool wide_none, std::vector<uint16_t> &out) {
    if (wide_none && f.hi - f.lo > 2) return true;
    const uint32_t op = cw & 15u;
    const bool simple = ncig == 1 && (op == 0 || op == 7 || op == 8) && (cw >> 4) == L && L > 1 && (uint32_t)(end - pos) == L;
    if (!(simple && L <= 480 && f.hi - f.lo <= 2)) return false;
    const int64_t q0 = (int64_t)f.hi - 1 - pos;
    if (q0 >= 0 && q0 < (int64_t)L) {
        const int64_t q1 = std::min<int64_t>(q0 + f.extra, (int64_t)L - 1);
        for (int64_t q = q0; q <= q1; q++) out.push_back((uint16_t)q);
    }
    return true;
}

} // namespace

struct SliceBase { // running totals of the variable-length columns in front of a slice of the output order
    int64_t cig = 0, omitted = 0, units = 0, seq = 0, exc = 0, qpos = 0, esc = 0, bl = 0, blu = 0, names = 0;
    void add(const SliceBase &o) { names += o.names; cig += o.cig; omitted += o.omitted; units += o.units; seq += o.seq; exc += o.exc; qpos += o.qpos; esc += o.esc; bl += o.bl; blu += o.blu; }
};

struct uz_stage {
    const uz_bamsrc *src = nullptr;
    Opt opt{};
    std::vector<Task> tasks;
    std::vector<std::vector<Fx>> fx;       // per reference, sorted by lo
    std::vector<int32_t> fx_max_len;
    std::vector<int64_t> order;            // the kept records in file order: (task << 32 | record)
    std::unordered_map<int64_t, int64_t> folded; // a record kept by two tasks: duplicate -> survivor
    std::vector<int64_t> cut;              // slices of `order`
    std::vector<SliceBase> base;           // [slices + 1]: totals in front of every slice; the last entry holds the sizes
    int64_t n = 0, n_tup = 0, n_qnames = 0;
    int wide = 0;
    // the plan in two halves (uz_bam_stage_begin / uz_bam_stage_finish): between them the blocks the walk will read can be inflated
    // elsewhere -- on the device -- and handed back
    int threads = 0;
    bool begun = false, finished = false;
    std::vector<std::vector<uz_walk_desc>> twin; // uz_stage_walk_host: the host's twin of the device's walk, task by task
    // UZ_STAGE_SMALL_TASKS: the plan is made for the device's walk (one wavefront per walk task).  The host's tasks stay as they are -- their blocks are
    // gathered and inflated once -- but the walk plan cuts each into SUB-TASKS, groups of its reach intervals of at most ~32 kb: a sub-task starts
    // its walk where the linear index puts the first record of its first window (a record boundary the file's index vouches for) and stops at the
    // end of its last interval.  More, shorter chains of records for the device, not one more block to inflate.
    bool small_tasks = false;
    struct SubTask { int32_t host, r0, r1, s0, tb; uint64_t beg; };
    std::vector<SubTask> subs;       // built by the walk plan (build_subtasks)
    std::vector<uint8_t> sub_preflag; // per host task: a sub-task starts in a block the gather did not list -- the host walks the task itself
    bool desc = false; // the descriptor route (uz_bam_stage_finish_desc): the walk ran on the device, the host holds no record bytes
    // the joins on the device (uz_bam_join, csrc/k_bamjoin.hip): what the host still contributes are the records it walks itself -- the tasks the
    // device handed back (uz_stage_walk_flagged) and mates looked up through the index (uz_stage_lookup) -- as descriptors whose bytes lie in `xaux`
    std::vector<uz_walk_desc> xdesc;
    std::vector<uint8_t> xaux;
    std::vector<int32_t> look_tid; // the reference of every look-up task (join task n_tasks + k)
    bool joined = false;
    const uint8_t *inflated = nullptr;
    int64_t n_pre_blocks = 0, pre_bytes = 0;
    std::vector<uint64_t> tup_key;
    std::vector<uint32_t> tup_k2;
    std::vector<int64_t> name_of_id;       // id -> record
    std::vector<int64_t> contig_off;
    std::vector<int32_t> max_span;
    int64_t io_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // file bytes read, blocks inflated, records walked, records kept, reach intervals, mates looked up through the index
    double timing[6] = {0, 0, 0, 0, 0, 0};          // file spans, walk, mates, numbering, fill
    mutable std::string name_tmp;
};

namespace {

inline WRec &rec_of(uz_stage &P, int64_t ref) { return P.tasks[(size_t)(ref >> 32)].recs[(size_t)(ref & 0xFFFFFFFF)]; }
inline const WRec &rec_of(const uz_stage &P, int64_t ref) { return P.tasks[(size_t)(ref >> 32)].recs[(size_t)(ref & 0xFFFFFFFF)]; }

inline int64_t final_ref(const uz_stage &P, int64_t m) { // a mate that was folded away stands for its survivor
    for (int hop = 0; hop < 4 && m >= 0 && !P.folded.empty(); hop++) {
        auto it = P.folded.find(m);
        if (it == P.folded.end()) break;
        m = it->second;
    }
    return m;
}

// start / tlen / mate / qname of output record k in the link form (uz_types.h: the start difference in eight bits, the rest as the
// pair form's one byte): start8 / pair the column values, e[c] / has[c] the escape-list entries (columns 0 start, 1 tlen, 2 mate,
// 3 name id); returns their number
inline PairRec pair_rec(const uz_stage &P, int64_t k) {
    const int64_t ref = P.order[(size_t)k];
    const WRec &x = rec_of(P, ref);
    const int64_t m = final_ref(P, x.mate_ref);
    PairRec r;
    r.mate = m < 0 ? -1 : (int64_t)rec_of(P, m).gidx;
    r.start = x.pos; r.end = x.end; r.tlen = x.tlen; r.qid = x.qid;
    r.is_new = P.name_of_id[(size_t)x.qid] == ref;
    return r;
}
inline int link_of(const uz_stage &P, int64_t k, uint8_t &start8, uint8_t &pair, int32_t e[4], bool has[4]) {
    const WRec &x = rec_of(P, P.order[(size_t)k]);
    const WRec *px = k > 0 ? &rec_of(P, P.order[(size_t)k - 1]) : nullptr;
    const int64_t ds = (int64_t)x.pos - (px ? (int64_t)px->pos : 0);
    int n = 0;
    has[0] = !(ds >= 0 && ds <= 254);
    if (has[0]) { start8 = (uint8_t)UZ_D8_ESC; e[0] = (int32_t)ds; n++; }
    else start8 = (uint8_t)ds;
    pair = pair8_code(k, [&](int64_t j) { return pair_rec(P, j); });
    return n + pair8_escapes(pair, pair_rec(P, k), e, has);
}

// file spans holding every record that overlaps [a, b) of reference `ref` (bins + linear index); sorted, merged per block
void spans_for(const BaiRef &ref, int32_t a, int32_t b, std::vector<Chunk> &out, uint64_t *est_end = nullptr) {
    std::vector<uint32_t> bins;
    reg2bins(a, b, bins);
    uint64_t min_off = 0;
    const size_t w = (size_t)(std::max<int64_t>(a, 0) >> 14);
    if (!ref.linear.empty()) min_off = ref.linear[std::min(w, ref.linear.size() - 1)];
    std::vector<Chunk> cs;
    for (uint32_t bn : bins) {
        auto it = std::lower_bound(ref.bins.begin(), ref.bins.end(), bn, [](const auto &x, uint32_t key) { return x.first < key; });
        if (it == ref.bins.end() || it->first != bn) continue;
        for (const Chunk &c : it->second)
            if (c.end > min_off) cs.push_back(Chunk{std::max(c.beg, min_off), c.end});
    }
    std::sort(cs.begin(), cs.end(), [](const Chunk &x, const Chunk &y) { return x.beg < y.beg || (x.beg == y.beg && x.end < y.end); });
    out.clear();
    for (const Chunk &c : cs) {
        // spans that touch, overlap or meet in one BGZF block are walked as one: no block is inflated twice
        if (!out.empty() && (c.beg >> 16) <= (out.back().end >> 16)) out.back().end = std::max(out.back().end, c.end);
        else out.push_back(c);
    }
    if (est_end) {
        *est_end = out.empty() ? 0 : out.back().end;
        const size_t wb = (size_t)(std::max<int64_t>((int64_t)b - 1, 0) >> 14) + 1; // first record overlapping the window behind b
        if (!out.empty() && wb < ref.linear.size() && ref.linear[wb] > out.front().beg) *est_end = std::min(*est_end, ref.linear[wb]);
    }
}

// walks the spans of a task: direct records (a fetch returns them) and every other record, of which only those that share a
// name with a direct one are kept as mate candidates
inline bool same_name(const Task &A, const WRec &a, const Task &B, const WRec &b) {
    if (A.by_hash) return a.nhash == b.nhash && a.nhash2 == b.nhash2 && a.l_name == b.l_name; // (descriptor route: uz_bamwalk.h)
    return a.nhash == b.nhash && a.l_name == b.l_name && memcmp(A.names.data() + a.name_at, B.names.data() + b.name_at, a.l_name) == 0;
}

// mate(): the first record in file order with the name, on the mate's reference, overlapping the mate position, carrying the
// other read-of-pair flag (pysam's AlignmentFile.mate; it may be a secondary / supplementary record)
inline bool is_mate_of(const WRec &x, const WRec &y, int32_t y_tid) {
    if (y_tid != x.mtid) return false;
    if (!((int64_t)y.pos < (int64_t)x.mpos + 1 && (int64_t)y.end > (int64_t)x.mpos)) return false;
    const uint16_t want = (uint16_t)((x.flag ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2));
    return (y.flag & want) != 0;
}

inline bool wants_mate(const WRec &x, int32_t n_ref) { return (x.flag & FPAIRED) && !(x.flag & FMUNMAP) && x.mtid >= 0 && x.mtid < n_ref; }

struct Scratch { // a worker's buffers, kept from task to task
    Inflater inf;
    std::vector<uint8_t> buf;
    std::vector<WRec> all;
    Task tmp;
    std::vector<uint64_t> dn;
    std::vector<int32_t> tab, stack;
    std::vector<uint16_t> blv; // the bases the fetches ask of the record at hand
};

void finish_task(const uz_stage &P, Task &T, Scratch &W, size_t ti);
void mates_in_task(const uz_stage &P, Task &T, Scratch &W, size_t ti);

// descriptor route: a record the host walked itself keeps its bytes (the device packs it from there: uz_stage_kept hands them over as the
// batch's aux bytes) instead of the link-form extract
void keep_raw(Task &T, WRec &r, const uint8_t *p, uint32_t bs) {
    const uint32_t l_name = p[8];
    r.l_name = (uint8_t)(l_name - 1);
    r.nhash2 = uz_name_hash2(p + 32, l_name - 1);
    r.pay_at = (uint32_t)T.raw.size();
    r.cigar_at = 4 + bs;
    T.raw.insert(T.raw.end(), p - 4, p + bs);
    r.src = 0; // (set when the aux bytes are laid out: uz_stage_kept)
}

void walk_task(const uz_stage &P, Task &T, Scratch &W, size_t ti, bool finish = true) {
    T.n_walked = 0;
    const uz_bamsrc &S = *P.src;
    const Opt &o = P.opt;
    const std::vector<Fx> &fx = P.fx[(size_t)T.tid];
    const int32_t max_len = P.fx_max_len[(size_t)T.tid];
    Stream s(S, W.inf, W.buf);
    if (P.inflated && !T.pre.empty()) { s.pre.blks = &T.pre; s.pre.base = P.inflated; }
    std::vector<WRec> &all = W.all;
    Task &tmp = W.tmp; // pools of every walked record; the survivors are copied over
    all.clear(); tmp.names.clear(); tmp.cigars.clear(); tmp.pay.clear(); tmp.raw.clear();
    tmp.tid = T.tid;
    std::vector<uint16_t> &blv = W.blv;
    size_t ri = 0; // the reach interval the walk is in or in front of
    bool stop = false;
    for (size_t ci = 0; ci < T.spans.size() && !stop; ci++) {
        s.seek(T.spans[ci].beg);
        for (;;) {
            uint64_t voff;
            const uint8_t *p;
            uint32_t bs;
            if (!s.next(voff, p, bs)) { stop = true; break; } // end of the file
            if (voff >= T.spans[ci].end) break;
            const int32_t tid = rdi32(p), pos = rdi32(p + 4);
            if (tid != T.tid) {
                if (tid < 0 || tid > T.tid) { stop = true; break; }
                s.advance(bs);
                continue;
            }
            if (pos >= T.b) { stop = true; break; }
            const uint32_t l_name = p[8], ncig = rd16(p + 12);
            const uint16_t fl = rd16(p + 14);
            const int32_t lseq = rdi32(p + 16);
            if (lseq < 0 || lseq > 0xFFFF) fail(UZ_IO_E_RANGE, "record too long for the 16-bit length columns (l_seq %d)", lseq);
            if (l_name < 1 || 32 + (size_t)l_name + 4 * (size_t)ncig > (size_t)bs) fail(UZ_IO_E_FORMAT, "alignment record overruns its block (or has no read name)");
            const int32_t end = endpos_of(p, pos, fl, ncig, l_name);
            T.n_walked++;
            // between two reach intervals: nothing there can be fetched, and a mate position there is looked up through the index
            while (ri < T.reach.size() && pos >= T.reach[ri].second) ri++;
            if (ri < T.reach.size() && end <= T.reach[ri].first) { s.advance(bs); continue; }
            WRec r;
            memset(&r, 0, sizeof(r));
            r.voff = voff; r.pos = pos; r.end = end; r.flag = fl; r.mapq = p[9];
            r.n_cigar = (uint16_t)ncig; r.l_seq = (uint16_t)lseq;
            r.mtid = rdi32(p + 20); r.mpos = rdi32(p + 24); r.tlen = rdi32(p + 28);
            r.mate_ref = -2;
            r.nhash = hash_name(p + 32, (size_t)l_name - 1);
            // does a fetch return it?  (start < hi and end > lo: read_collector.py:385, :167)
            bool direct = false, listable = o.bl;
            uint16_t um = 0;
            blv.clear();
            {
                auto it = std::lower_bound(fx.begin() + (ptrdiff_t)T.f0, fx.begin() + (ptrdiff_t)T.f1, (int64_t)pos - max_len,
                                           [](const Fx &f, int64_t key) { return (int64_t)f.lo < key; });
                const uint32_t cw = ncig == 1 ? rd32(p + 32 + l_name) : 0u;
                for (; it != fx.begin() + (ptrdiff_t)T.f1 && it->lo < end; ++it)
                    if (it->hi > pos) {
                        direct = true;
                        if (o.masks) um |= mask_bits(*it, pos, end, ncig, cw, (uint32_t)lseq, o.wide_none);
                        if (listable) listable = list_bits(*it, pos, end, ncig, cw, (uint32_t)lseq, o.wide_none, blv);
                    }
            }
            r.keep = direct ? 2 : 0;
            const bool bases = direct || o.all_bases;
            bool use_list = false;
            r.umask = (uint16_t)UZ_UMASK_ALL;
            if (o.masks) {
                uint16_t m16 = bases ? um : (uint16_t)0;
                if (m16 != UZ_UMASK_ALL && (uint32_t)__builtin_popcount(m16) == UZ_ROW_UNITS(lseq)) m16 = (uint16_t)UZ_UMASK_ALL;
                r.umask = m16;
                // The bases as a list (uz_types.h bl_*): when every fetch names single positions, they are fewer bytes than the units they lie
                // in, and none of them is '=' (BAM code 0, which the device reads as "not listed").
                if (listable && bases && m16 != UZ_UMASK_ALL && m16 != 0) {
                    std::sort(blv.begin(), blv.end());
                    blv.erase(std::unique(blv.begin(), blv.end()), blv.end());
                    const uint8_t *sq0 = p + 32 + l_name + 4 * (size_t)ncig;
                    bool ok = !blv.empty() && blv.size() <= 255 && 5 * blv.size() < 32 * (size_t)__builtin_popcount(m16);
                    for (size_t j = 0; ok && j < blv.size(); j++) {
                        const uint32_t k = blv[j];
                        ok = ((k & 1) ? (uint32_t)(sq0[k >> 1] & 15u) : (uint32_t)(sq0[k >> 1] >> 4)) != 0u;
                    }
                    use_list = ok;
                }
            }
            if (P.desc) keep_raw(tmp, r, p, bs);
            else extract(tmp, r, p, bs, o, bases, use_list ? &blv : nullptr);
            all.push_back(r);
            s.advance(bs);
        }
    }
    T.file_bytes = s.file_bytes; T.n_blocks = s.n_blocks; T.n_pre = s.n_pre;
    if (finish) finish_task(P, T, W, ti);
}

// the second half of a task's walk, from the records in W.all: the mate candidates, and the mates inside the task
void finish_task(const uz_stage &P, Task &T, Scratch &W, size_t ti) {
    const Opt &o = P.opt;
    std::vector<WRec> &all = W.all;
    Task &tmp = W.tmp;
    // mate candidates: the records that share a name with a direct one
    std::vector<uint64_t> &dn = W.dn;
    dn.clear();
    for (const WRec &r : all) if (r.keep == 2) dn.push_back(r.nhash);
    std::sort(dn.begin(), dn.end());
    dn.erase(std::unique(dn.begin(), dn.end()), dn.end());
    for (const WRec &r0 : all) {
        if (r0.keep != 2 && !std::binary_search(dn.begin(), dn.end(), r0.nhash)) continue;
        WRec r = r0;
        if (P.desc) { // no pools: the record's bytes stay where they are (a host-walked task: in its raw bytes)
            if (!tmp.raw.empty()) {
                r.pay_at = (uint32_t)T.raw.size();
                T.raw.insert(T.raw.end(), tmp.raw.begin() + r0.pay_at, tmp.raw.begin() + r0.pay_at + r0.cigar_at);
            }
            T.recs.push_back(r);
            continue;
        }
        r.name_at = (uint32_t)T.names.size();
        T.names.insert(T.names.end(), tmp.names.begin() + r0.name_at, tmp.names.begin() + r0.name_at + r0.l_name);
        r.cigar_at = (uint32_t)T.cigars.size();
        if (!r0.simple) T.cigars.insert(T.cigars.end(), tmp.cigars.begin() + r0.cigar_at, tmp.cigars.begin() + r0.cigar_at + r0.n_cigar);
        r.pay_at = (uint32_t)T.pay.size();
        if (r0.has_pay) {
            const size_t len = (size_t)r0.n_units * 8 + (size_t)r0.n_bl * 3 + (size_t)r0.n_exc * 4 + (size_t)r0.n_qpos * 2 + (o.lists ? 0 : (size_t)UZ_ROW_UNITS(r0.l_seq) * UZ_QLOW_UNIT_BYTES);
            T.pay.insert(T.pay.end(), tmp.pay.begin() + r0.pay_at, tmp.pay.begin() + r0.pay_at + (ptrdiff_t)len);
        }
        T.recs.push_back(r);
    }
    mates_in_task(P, T, W, ti);
}

void mates_in_task(const uz_stage &P, Task &T, Scratch &W, size_t ti) {
    const uz_bamsrc &S = *P.src;
    // mates inside the task, while its records are hot: exact for a mate position inside one of the task's reach intervals (every
    // record overlapping such a position was walked, and the task holds ALL records of a name it holds at all).  Generation by
    // generation: a mate found becomes a member and has its own mate looked up.  What is left (mate_ref -2) goes through the
    // other tasks / the index afterwards.
    const int32_t n_ref = (int32_t)S.contigs.size();
    const size_t nr = T.recs.size();
    size_t cap = 16;
    while (cap < 2 * nr) cap <<= 1;
    std::vector<int32_t> &tab = W.tab, &stack = W.stack;
    tab.assign(cap, -1);
    for (size_t j = 0; j < nr; j++) { // linear probing, inserted in file order: equal names are met in file order along a probe chain
        size_t slot = (size_t)(T.recs[j].nhash * 0x9E3779B97F4A7C15ULL >> 20) & (cap - 1);
        while (tab[slot] >= 0) slot = (slot + 1) & (cap - 1);
        tab[slot] = (int32_t)j;
    }
    auto in_reach = [&](int32_t pos_q) {
        auto it = std::upper_bound(T.reach.begin(), T.reach.end(), pos_q, [](int32_t key, const std::pair<int32_t, int32_t> &x) { return key < x.second; });
        return it != T.reach.end() && it->first <= pos_q;
    };
    stack.clear();
    for (size_t j = nr; j-- > 0;) if (T.recs[j].keep == 2) stack.push_back((int32_t)j);
    while (!stack.empty()) {
        WRec &x = T.recs[(size_t)stack.back()];
        stack.pop_back();
        if (x.mate_ref != -2) continue;
        if (!wants_mate(x, n_ref)) { x.mate_ref = -1; continue; }
        if (!(x.mtid == T.tid && in_reach(x.mpos))) continue; // not this task's to answer
        x.mate_ref = -1;
        for (size_t slot = (size_t)(x.nhash * 0x9E3779B97F4A7C15ULL >> 20) & (cap - 1); tab[slot] >= 0; slot = (slot + 1) & (cap - 1)) {
            WRec &y = T.recs[(size_t)tab[slot]];
            if (y.nhash != x.nhash || !same_name(T, x, T, y) || !is_mate_of(x, y, T.tid)) continue;
            x.mate_ref = ((int64_t)ti << 32) | (int64_t)tab[slot];
            if (y.keep == 0) { y.keep = 1; stack.push_back(tab[slot]); }
            break;
        }
    }
}

// a mate looked up through the index: the records overlapping [mpos, mpos + 1) of the mate's reference, walked like a fetch
struct Lookup { int64_t who; int32_t mtid, mpos; };

// A mate looked up through the index: the records of reference `mtid` overlapping [mpos, mpos + 1), walked like a one-base fetch -- no fetch returns
// them (nothing is direct) -- of which those are kept whose name somebody asks for: asked(h1, h2, length, the name's bytes).
template <class Asked>
void lookup_walk(const uz_stage &P, Task &T, int32_t mtid, int32_t mpos, Inflater &inf, Asked asked) {
    const uz_bamsrc &S = *P.src;
    T.tid = mtid; T.a = mpos; T.b = mpos + 1; T.f0 = T.f1 = 0;
    T.by_hash = P.desc;
    spans_for(S.refs[(size_t)mtid], T.a, T.b, T.spans);
    Stream s(S, inf);
    bool stop = false;
    for (size_t ci = 0; ci < T.spans.size() && !stop; ci++) {
        s.seek(T.spans[ci].beg);
        for (;;) {
            uint64_t voff;
            const uint8_t *p;
            uint32_t bs;
            if (!s.next(voff, p, bs)) { stop = true; break; }
            if (voff >= T.spans[ci].end) break;
            const int32_t tid2 = rdi32(p), pos = rdi32(p + 4);
            if (tid2 != T.tid) { if (tid2 < 0 || tid2 > T.tid) { stop = true; break; } s.advance(bs); continue; }
            if (pos > mpos) { stop = true; break; }
            const uint32_t l_name = p[8], ncig = rd16(p + 12);
            const uint16_t fl = rd16(p + 14);
            const int32_t lseq = rdi32(p + 16);
            if (lseq < 0 || lseq > 0xFFFF) fail(UZ_IO_E_RANGE, "record too long for the 16-bit length columns (l_seq %d)", lseq);
            if (l_name < 1 || 32 + (size_t)l_name + 4 * (size_t)ncig > (size_t)bs) fail(UZ_IO_E_FORMAT, "alignment record overruns its block (or has no read name)");
            const int32_t end = endpos_of(p, pos, fl, ncig, l_name);
            T.n_walked++;
            if (end > mpos) {
                const uint64_t h = hash_name(p + 32, (size_t)l_name - 1);
                const uint32_t h2 = P.desc ? uz_name_hash2(p + 32, l_name - 1) : 0u;
                if (asked(h, h2, l_name - 1, p + 32)) {
                    WRec r;
                    memset(&r, 0, sizeof(r));
                    r.voff = voff; r.pos = pos; r.end = end; r.flag = fl; r.mapq = p[9];
                    r.n_cigar = (uint16_t)ncig; r.l_seq = (uint16_t)lseq;
                    r.mtid = rdi32(p + 20); r.mpos = rdi32(p + 24); r.tlen = rdi32(p + 28);
                    r.mate_ref = -2; r.nhash = h; r.keep = 0;
                    r.umask = P.opt.masks ? (uint16_t)0 : (uint16_t)UZ_UMASK_ALL;
                    if (P.desc) keep_raw(T, r, p, bs);
                    else extract(T, r, p, bs, P.opt, P.opt.all_bases);
                    T.recs.push_back(r);
                }
            }
            s.advance(bs);
        }
    }
    T.file_bytes = s.file_bytes; T.n_blocks = s.n_blocks;
}

// first half of the plan: fetches -> reach intervals -> tasks with their file spans
void plan_begin(uz_stage &P, int64_t n_fetch, const int32_t *tid, const int32_t *lo, const int32_t *hi, const uint16_t *extra, int threads) {
    const uz_bamsrc &S = *P.src;
    const int32_t n_ref = (int32_t)S.contigs.size();
    double t0 = now_s();
    threads = resolve_threads(threads);
    // (UZ_STAGE_SLACK: a test hook -- a small slack pushes the mates outside the reach intervals, onto the cross-task / index path)
    const int32_t REACH_SLACK = getenv("UZ_STAGE_SLACK") ? std::max(0, atoi(getenv("UZ_STAGE_SLACK"))) : REACH_SLACK_DEFAULT;
    // ---- the fetches per reference, sorted; reach intervals = fetches grown by the slack, merged
    P.fx.assign((size_t)n_ref, {});
    P.fx_max_len.assign((size_t)n_ref, 0);
    for (int64_t k = 0; k < n_fetch; k++)
        if (tid[k] >= 0 && tid[k] < n_ref && hi[k] > lo[k]) P.fx[(size_t)tid[k]].push_back(Fx{lo[k], hi[k], extra ? extra[k] : (uint16_t)0});
    for (int32_t t = 0; t < n_ref; t++) {
        auto &v = P.fx[(size_t)t];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end(), [](const Fx &x, const Fx &y) { return x.lo < y.lo || (x.lo == y.lo && x.hi < y.hi); });
        for (const Fx &f : v) P.fx_max_len[(size_t)t] = std::max(P.fx_max_len[(size_t)t], f.hi - f.lo);
        size_t f0 = 0;
        int64_t a = (int64_t)v[0].lo - REACH_SLACK, b = (int64_t)v[0].hi + REACH_SLACK;
        for (size_t k = 1; k <= v.size(); k++) {
            if (k < v.size() && (int64_t)v[k].lo - REACH_SLACK <= b) { b = std::max(b, (int64_t)v[k].hi + REACH_SLACK); continue; }
            Task T;
            T.tid = t; T.a = (int32_t)std::max<int64_t>(a, 0); T.b = (int32_t)std::min<int64_t>(b, INT32_MAX); T.f0 = f0; T.f1 = k;
            T.reach.push_back({T.a, T.b});
            P.tasks.push_back(std::move(T));
            if (k < v.size()) { f0 = k; a = (int64_t)v[k].lo - REACH_SLACK; b = (int64_t)v[k].hi + REACH_SLACK; }
        }
    }
    parallel_slices((int64_t)P.tasks.size(), workers_for((int64_t)P.tasks.size(), threads, 64), [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; i++) spans_for(S.refs[(size_t)P.tasks[(size_t)i].tid], P.tasks[(size_t)i].a, P.tasks[(size_t)i].b, P.tasks[(size_t)i].spans, &P.tasks[(size_t)i].est_end);
    });
    P.io_stats[4] = (int64_t)P.tasks.size();
    { // reach intervals whose file spans meet in a BGZF block (the same 16 kb bin, neighbouring windows) are walked as one task
        std::vector<Task> merged;
        for (Task &T : P.tasks) {
            if (T.spans.empty()) continue; // the index knows no record there
            Task *L = merged.empty() ? nullptr : &merged.back();
            // (up to a size: a file that holds nothing but the fetched windows would chain into one task per reference)
            // (compressed bytes.  Smaller tasks for the device's walk -- one wavefront per task -- were tried: 64 / 128 / 256 / 768 KB = 43 / 50 / 69 / 66 k DNMs/s
            // in the feed pass; every cut has the tasks on both sides gather the blocks around it, 22 % more blocks at 256 KB, 78 % at 128.  The walk plan
            // cuts the tasks into sub-tasks instead, without touching what is gathered: uz_stage::subs)
            static const int env_kb = getenv("UZ_STAGE_MAX_TASK_KB") ? std::max(16, atoi(getenv("UZ_STAGE_MAX_TASK_KB"))) : 0; // (development aid)
            const uint64_t MAX_TASK_BYTES = (uint64_t)(env_kb ? env_kb : 768) << 10;
            if (L && L->tid == T.tid && (T.spans.front().beg >> 16) <= (L->est_end >> 16) &&
                (std::max(T.est_end, L->est_end) >> 16) - (L->spans.front().beg >> 16) <= MAX_TASK_BYTES) {
                std::vector<Chunk> all(L->spans);
                all.insert(all.end(), T.spans.begin(), T.spans.end());
                std::sort(all.begin(), all.end(), [](const Chunk &x, const Chunk &y) { return x.beg < y.beg || (x.beg == y.beg && x.end < y.end); });
                L->spans.clear();
                for (const Chunk &c : all) {
                    if (!L->spans.empty() && (c.beg >> 16) <= (L->spans.back().end >> 16)) L->spans.back().end = std::max(L->spans.back().end, c.end);
                    else L->spans.push_back(c);
                }
                L->b = T.b; L->f1 = T.f1;
                L->est_end = std::max(L->est_end, T.est_end);
                L->reach.push_back({T.a, T.b});
            } else
                merged.push_back(std::move(T));
        }
        P.tasks.swap(merged);
    }
    P.timing[0] = now_s() - t0;
    P.threads = threads;
    P.begun = true;
}

// second half of the plan: walk, mates, numbering
// d (descriptor route): the device's walk of every task -- its records inside the reach intervals, file order, task by task
// (d_first[t] .. d_first[t + 1]); a task flagged there is walked here as well
// by_sub: d_first / d_walked count the walk plan's SUB-tasks (uz_stage::subs; d_flags stays per task of the stage): a task's records are its
// sub-tasks' in order, without those a later sub-task met again (uz_stage_merge_subtasks states the rule; here nothing is copied)
void plan_finish(uz_stage &P, const uz_walk_desc *d = nullptr, const int64_t *d_first = nullptr, const int32_t *d_flags = nullptr,
                 const int64_t *d_walked = nullptr, bool by_sub = false) {
    const uz_bamsrc &S = *P.src;
    const int32_t n_ref = (int32_t)S.contigs.size();
    const int threads = P.threads;
    double t1 = now_s();
    // ---- the walk
    {
        const int w = (int)std::min<int64_t>(threads, std::max<int64_t>(1, (int64_t)P.tasks.size()));
        std::vector<std::unique_ptr<Scratch>> scr((size_t)w);
        std::vector<size_t> sub_of_task; // by_sub: the first sub-task of every task of the stage (they lie in order)
        if (by_sub) {
            sub_of_task.assign(P.tasks.size() + 1, P.subs.size());
            for (size_t u = P.subs.size(); u-- > 0;) sub_of_task[(size_t)P.subs[u].host] = u;
            for (size_t i = P.tasks.size(); i-- > 0;) sub_of_task[i] = std::min(sub_of_task[i], sub_of_task[i + 1]);
        }
        parallel_dynamic((int64_t)P.tasks.size(), w, [&](int64_t i, int k) {
            if (!scr[(size_t)k]) scr[(size_t)k].reset(new Scratch());
            Task &T = P.tasks[(size_t)i];
            Scratch &W = *scr[(size_t)k];
            if (!d || (d_flags && d_flags[i])) { walk_task(P, T, W, (size_t)i); return; }
            // the device walked it: its records from the descriptors -- the direct ones and those that share a name hash with one of them
            // (finish_task's rule; the device has dropped most of the others already)
            std::vector<uint64_t> &dn = W.dn;
            dn.clear();
            // the ranges of descriptors that are this task's: one, or one per sub-task with the stop of the sub-task before it
            const size_t u0 = by_sub ? sub_of_task[(size_t)i] : (size_t)i, u1 = by_sub ? sub_of_task[(size_t)i + 1] : (size_t)i + 1;
            int64_t n_desc = 0, n_walk = 0;
            for (size_t u = u0; u < u1; u++) {
                const int32_t stop_before = (by_sub && u > u0) ? P.subs[u - 1].tb : INT32_MIN;
                for (int64_t j = d_first[u]; j < d_first[u + 1]; j++) if (d[j].direct && d[j].pos >= stop_before) dn.push_back(d[j].h1);
                n_desc += d_first[u + 1] - d_first[u];
                n_walk += d_walked ? d_walked[u] : d_first[u + 1] - d_first[u];
            }
            std::sort(dn.begin(), dn.end());
            dn.erase(std::unique(dn.begin(), dn.end()), dn.end());
            T.recs.reserve((size_t)n_desc);
            for (size_t u = u0; u < u1; u++) {
              const int32_t stop_before = (by_sub && u > u0) ? P.subs[u - 1].tb : INT32_MIN;
              for (int64_t j = d_first[u]; j < d_first[u + 1]; j++) {
                const uz_walk_desc &x = d[j];
                if (x.pos < stop_before) continue; // (the sub-task before met it too, and kept it)
                if (!x.direct && !std::binary_search(dn.begin(), dn.end(), x.h1)) continue;
                WRec r;
                memset(&r, 0, sizeof(r));
                r.voff = x.voff; r.nhash = x.h1; r.nhash2 = x.h2; r.src = x.src;
                r.pos = x.pos; r.end = x.end; r.tlen = x.tlen; r.mpos = x.mpos; r.mtid = x.mtid;
                r.flag = x.flag; r.l_seq = x.l_seq; r.n_cigar = x.n_cigar; r.mapq = x.mapq; r.l_name = x.l_name;
                r.keep = x.direct ? 2 : 0;
                r.mate_ref = -2;
                r.umask = (uint16_t)UZ_UMASK_ALL;
                T.recs.push_back(r);
              }
            }
            T.n_walked = n_walk;
            mates_in_task(P, T, W, (size_t)i);
        });
    }
    double t2 = now_s();
    P.timing[1] = t2 - t1;
    // ---- mates.  In-task look-ups are exact for mate positions inside the task's reach interval (every record overlapping
    // such a position was walked); anything else goes through the index.
    const size_t n_tasks0 = P.tasks.size();
    // per task a name index would pay for deep pile-ups; the candidates are grouped by name hash first
    struct ByName { std::vector<std::pair<uint64_t, uint32_t>> v; };
    std::vector<ByName> by_name(P.tasks.size()); // (filled only when a mate is left to look up outside its own task)
    auto build_by_name = [&] {
        parallel_slices((int64_t)n_tasks0, workers_for((int64_t)n_tasks0, threads, 16), [&](int64_t i0, int64_t i1, int) {
            for (int64_t i = i0; i < i1; i++) {
                auto &v = by_name[(size_t)i].v;
                const Task &T = P.tasks[(size_t)i];
                v.resize(T.recs.size());
                for (size_t j = 0; j < T.recs.size(); j++) v[j] = {T.recs[j].nhash, (uint32_t)j};
                std::sort(v.begin(), v.end());
            }
        });
    };
    // the task of the first walk whose reach interval holds position `pos` of reference `tid` (they are sorted and disjoint), or -1
    auto covering = [&](int32_t tid_q, int32_t pos_q) -> int64_t {
        size_t lo_i = 0, hi_i = n_tasks0;
        while (lo_i < hi_i) {
            const size_t mid = (lo_i + hi_i) / 2;
            const Task &M = P.tasks[mid];
            if (M.tid < tid_q || (M.tid == tid_q && M.b <= pos_q)) lo_i = mid + 1; else hi_i = mid;
        }
        if (lo_i < n_tasks0 && P.tasks[lo_i].tid == tid_q && P.tasks[lo_i].a <= pos_q && pos_q < P.tasks[lo_i].b) {
            const auto &rv = P.tasks[lo_i].reach; // inside one of its reach intervals?
            auto it = std::upper_bound(rv.begin(), rv.end(), pos_q, [](int32_t key, const std::pair<int32_t, int32_t> &x) { return key < x.second; });
            if (it != rv.end() && it->first <= pos_q) return (int64_t)lo_i;
        }
        return -1;
    };
    // A task of the first walk walked EVERY record overlapping a position inside its reach interval and kept all records of a name
    // as soon as one of them is a fetched one: when it holds the name at all, its answer is the index's answer.
    auto resolve_fast = [&](size_t ti, size_t ri) -> int { // 1: mate_ref set; 0: needs the index
        WRec &x = P.tasks[ti].recs[ri];
        if (!wants_mate(x, n_ref)) { x.mate_ref = -1; return 1; }
        const int64_t tc = covering(x.mtid, x.mpos);
        if (tc < 0) return 0;
        const Task &T = P.tasks[(size_t)tc];
        const auto &v = by_name[(size_t)tc].v;
        auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(x.nhash, (uint32_t)0));
        bool seen = false;
        for (; it != v.end() && it->first == x.nhash; ++it) { // ascending record index = file order
            const WRec &y = T.recs[it->second];
            if (!same_name(P.tasks[ti], x, T, y)) continue;
            seen = true;
            if (is_mate_of(x, y, T.tid)) { x.mate_ref = (tc << 32) | (int64_t)it->second; return 1; }
        }
        if (!seen) return 0;
        x.mate_ref = -1;
        return 1;
    };
    // generation by generation: the members whose mate has not been looked up; a found mate becomes a member
    std::vector<int64_t> frontier;
    {
        std::vector<int64_t> at(P.tasks.size() + 1, 0);
        parallel_slices((int64_t)P.tasks.size(), workers_for((int64_t)P.tasks.size(), threads, 64), [&](int64_t i0, int64_t i1, int) {
            for (int64_t ti = i0; ti < i1; ti++) {
                int64_t c = 0;
                for (const WRec &r : P.tasks[(size_t)ti].recs) c += r.keep != 0 && r.mate_ref == -2;
                at[(size_t)ti + 1] = c;
            }
        });
        for (size_t ti = 0; ti < P.tasks.size(); ti++) at[ti + 1] += at[ti];
        frontier.resize((size_t)at.back());
        parallel_slices((int64_t)P.tasks.size(), workers_for((int64_t)P.tasks.size(), threads, 64), [&](int64_t i0, int64_t i1, int) {
            for (int64_t ti = i0; ti < i1; ti++) {
                int64_t k = at[(size_t)ti];
                const auto &recs = P.tasks[(size_t)ti].recs;
                for (size_t ri = 0; ri < recs.size(); ri++)
                    if (recs[ri].keep != 0 && recs[ri].mate_ref == -2) frontier[(size_t)k++] = ((int64_t)ti << 32) | (int64_t)ri;
            }
        });
    }
    int64_t n_lookups = 0;
    if (!frontier.empty()) build_by_name();
    for (int gen = 0; gen < 64 && !frontier.empty(); gen++) {
        std::vector<Lookup> need;
        {
            const int w = workers_for((int64_t)frontier.size(), threads, 4096);
            std::vector<std::vector<Lookup>> part((size_t)w);
            parallel_slices((int64_t)frontier.size(), w, [&](int64_t i0, int64_t i1, int k) {
                for (int64_t i = i0; i < i1; i++) {
                    const int64_t ref = frontier[(size_t)i];
                    WRec &x = rec_of(P, ref);
                    if (x.mate_ref != -2) continue;
                    if (!resolve_fast((size_t)(ref >> 32), (size_t)(ref & 0xFFFFFFFF))) part[(size_t)k].push_back(Lookup{ref, x.mtid, x.mpos});
                }
            });
            for (auto &v : part) need.insert(need.end(), v.begin(), v.end());
        }
        if (!need.empty()) { // through the index: one extra task per look-up position (walked like a one-base fetch)
            std::sort(need.begin(), need.end(), [](const Lookup &x, const Lookup &y) { return x.mtid < y.mtid || (x.mtid == y.mtid && (x.mpos < y.mpos || (x.mpos == y.mpos && x.who < y.who))); });
            n_lookups += (int64_t)need.size();
            std::vector<size_t> first; // look-ups of one position share a task
            for (size_t k = 0; k < need.size(); k++)
                if (k == 0 || need[k].mtid != need[k - 1].mtid || need[k].mpos != need[k - 1].mpos) first.push_back(k);
            const size_t base = P.tasks.size();
            P.tasks.resize(base + first.size());
            const int w = (int)std::min<int64_t>(threads, (int64_t)first.size());
            std::vector<std::unique_ptr<Inflater>> infs((size_t)std::max(1, w));
            for (auto &p : infs) p.reset(new Inflater());
            parallel_dynamic((int64_t)first.size(), std::max(1, w), [&](int64_t g, int k) {
                Task &T = P.tasks[base + (size_t)g];
                const Lookup &q = need[first[(size_t)g]];
                const size_t k1 = (size_t)g + 1 < first.size() ? first[(size_t)g + 1] : need.size();
                lookup_walk(P, T, q.mtid, q.mpos, *infs[(size_t)k], [&](uint64_t h, uint32_t h2, uint32_t l_name, const uint8_t *name) {
                    for (size_t u = first[(size_t)g]; u < k1; u++) {
                        const WRec &x = rec_of(P, need[u].who);
                        const Task &TX = P.tasks[(size_t)(need[u].who >> 32)];
                        if (x.nhash == h && x.l_name == l_name && (P.desc ? x.nhash2 == h2 : memcmp(TX.names.data() + x.name_at, name, x.l_name) == 0)) return true;
                    }
                    return false;
                });
            });
            // answers: first match in file order
            for (size_t g = 0; g < first.size(); g++) {
                const size_t k1 = g + 1 < first.size() ? first[g + 1] : need.size();
                Task &T = P.tasks[base + g];
                for (size_t u = first[g]; u < k1; u++) {
                    WRec &x = rec_of(P, need[u].who);
                    const Task &TX = P.tasks[(size_t)(need[u].who >> 32)];
                    x.mate_ref = -1;
                    for (size_t j = 0; j < T.recs.size(); j++)
                        if (same_name(TX, x, T, T.recs[j]) && is_mate_of(x, T.recs[j], T.tid)) { x.mate_ref = ((int64_t)(base + g) << 32) | (int64_t)j; break; }
                }
            }
        }
        // the mates found become members; those that were not members yet are the next generation
        std::vector<int64_t> next;
        {
            const int w = workers_for((int64_t)frontier.size(), threads, 8192);
            std::vector<std::vector<int64_t>> part((size_t)w);
            parallel_slices((int64_t)frontier.size(), w, [&](int64_t i0, int64_t i1, int k) {
                for (int64_t i = i0; i < i1; i++) {
                    const int64_t m = rec_of(P, frontier[(size_t)i]).mate_ref;
                    if (m < 0) continue;
                    uint8_t zero = 0; // two members can name the same mate: the first to flip its flag lists it
                    if (__atomic_compare_exchange_n(&rec_of(P, m).keep, &zero, (uint8_t)1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) part[(size_t)k].push_back(m);
                }
            });
            for (auto &v : part) next.insert(next.end(), v.begin(), v.end());
        }
        frontier.swap(next);
    }
    double t3 = now_s();
    static const bool sub_timing = getenv("UZ_STAGE_TIMING") != nullptr; // development aid: where the numbering pass spends its time
    double tn[5] = {t3, 0, 0, 0, 0};
    P.timing[2] = t3 - t2;
    P.io_stats[5] = n_lookups;

    // ---- file order across tasks.  The tasks of the first walk are sorted and their kept records normally do not interleave; a
    // record found through the index (or walked by two tasks: reads longer than the slack) can lie anywhere: then the kept
    // records are sorted by virtual offset and duplicates folded.
    std::vector<int64_t> &order = P.order;
    {
        std::vector<int64_t> t_first(P.tasks.size() + 1, 0);
        parallel_slices((int64_t)P.tasks.size(), workers_for((int64_t)P.tasks.size(), threads, 64), [&](int64_t i0, int64_t i1, int) {
            for (int64_t ti = i0; ti < i1; ti++) {
                int64_t c = 0;
                for (const WRec &r : P.tasks[(size_t)ti].recs) c += r.keep != 0;
                t_first[(size_t)ti + 1] = c;
            }
        });
        for (size_t ti = 0; ti < P.tasks.size(); ti++) t_first[ti + 1] += t_first[ti];
        order.resize((size_t)t_first.back());
        parallel_slices((int64_t)P.tasks.size(), workers_for((int64_t)P.tasks.size(), threads, 64), [&](int64_t i0, int64_t i1, int) {
            for (int64_t ti = i0; ti < i1; ti++) {
                int64_t at = t_first[(size_t)ti];
                const auto &recs = P.tasks[(size_t)ti].recs;
                for (size_t ri = 0; ri < recs.size(); ri++)
                    if (recs[ri].keep) order[(size_t)at++] = ((int64_t)ti << 32) | (int64_t)ri;
            }
        });
        bool sorted = true;
        for (size_t ti = 0; ti + 1 < P.tasks.size() && sorted; ti++) { // kept records of consecutive tasks must ascend
            if (t_first[ti + 1] == t_first[ti]) continue;
            size_t tj = ti + 1;
            while (tj < P.tasks.size() && t_first[tj + 1] == t_first[tj]) tj++;
            if (tj < P.tasks.size() && rec_of(P, order[(size_t)t_first[ti + 1] - 1]).voff >= rec_of(P, order[(size_t)t_first[tj]]).voff) sorted = false;
        }
        if (!sorted) {
            std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return rec_of(P, x).voff < rec_of(P, y).voff; });
            std::vector<int64_t> uniq;
            for (int64_t e : order) {
                if (!uniq.empty() && rec_of(P, uniq.back()).voff == rec_of(P, e).voff) { // the same record kept twice: the one a fetch returned survives
                    WRec &a = rec_of(P, uniq.back()), &b = rec_of(P, e);
                    if (b.keep > a.keep) { P.folded[uniq.back()] = e; a.keep = 0; uniq.back() = e; }
                    else { P.folded[e] = uniq.back(); b.keep = 0; }
                    continue;
                }
                uniq.push_back(e);
            }
            order.swap(uniq);
        }
    }
    const int64_t n = (int64_t)order.size();
    if (n >= ((int64_t)1 << 31)) fail(UZ_IO_E_RANGE, "more than 2^31 - 1 alignment records");
    P.n = n;
    // slices of the output order: every later pass (and the fill) runs over them in parallel, with a serial step over the slices
    const int W = workers_for(n, threads, 8192);
    P.cut.assign((size_t)W + 1, 0);
    for (int k = 0; k <= W; k++) P.cut[(size_t)k] = n * k / W;
    auto slices = [&](auto fn) { parallel_slices(W, W, [&](int64_t s0, int64_t s1, int) { for (int64_t sl = s0; sl < s1; sl++) fn((int)sl, P.cut[(size_t)sl], P.cut[(size_t)sl + 1]); }); };
    slices([&](int, int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; k++) rec_of(P, order[(size_t)k]).gidx = (uint32_t)k; });
    tn[1] = now_s();
    // ---- names -> ids in order of first appearance (hash-sharded, as the table builder of the whole-file decoder does)
    {
        const int SH = 256;
        std::vector<std::vector<int64_t>> hist((size_t)W, std::vector<int64_t>(SH, 0));
        // (the name hashes side by side: the passes below look at nothing else of a record until two hashes are equal, and the records
        // themselves are ~100 bytes apiece behind two indirections)
        std::vector<uint64_t> nh((size_t)n);
        std::vector<uint64_t> nh2(P.desc ? (size_t)n : 0); // descriptor route: the second hash and the length side by side too -- a name IS that triple there
        slices([&](int sl, int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++) {
                const WRec &x = rec_of(P, order[(size_t)k]);
                const uint64_t h = x.nhash; nh[(size_t)k] = h; hist[(size_t)sl][h >> 56]++;
                if (P.desc) nh2[(size_t)k] = (uint64_t)x.nhash2 | ((uint64_t)x.l_name << 32);
            }
        });
        std::vector<int64_t> sh_off(SH + 1, 0);
        int64_t run = 0;
        for (int sft = 0; sft < SH; sft++) {
            sh_off[(size_t)sft] = run;
            for (int w = 0; w < W; w++) { const int64_t c = hist[(size_t)w][(size_t)sft]; hist[(size_t)w][(size_t)sft] = run; run += c; }
        }
        sh_off[SH] = run;
        std::vector<uint32_t> by_shard((size_t)n), first_of((size_t)n);
        slices([&](int sl, int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; k++) by_shard[(size_t)hist[(size_t)sl][nh[(size_t)k] >> 56]++] = (uint32_t)k; });
        parallel_slices(SH, workers_for(SH, threads, 1), [&](int64_t s0, int64_t s1, int) {
            std::vector<int32_t> tab;
            for (int64_t sft = s0; sft < s1; sft++) {
                const int64_t a = sh_off[(size_t)sft], b = sh_off[(size_t)sft + 1];
                size_t cap = 16;
                while (cap < 2 * (size_t)(b - a)) cap <<= 1;
                tab.assign(cap, -1);
                for (int64_t e = a; e < b; e++) { // ascending k inside a shard: the first record met for a name is its first in the file
                    const uint32_t k = by_shard[(size_t)e];
                    const uint64_t hk = nh[k];
                    size_t slot = (size_t)(hk * 0x9E3779B97F4A7C15ULL >> 20) & (cap - 1);
                    for (;;) {
                        const int32_t f = tab[slot];
                        if (f < 0) { tab[slot] = (int32_t)k; first_of[k] = k; break; }
                        if (nh[(size_t)f] == hk && P.desc) {
                            if (nh2[(size_t)f] == nh2[k]) { first_of[k] = (uint32_t)f; break; }
                        } else if (nh[(size_t)f] == hk) { // (then, and only then, the names themselves)
                            const int64_t ref = order[k], rf = order[(size_t)f];
                            if (same_name(P.tasks[(size_t)(ref >> 32)], rec_of(P, ref), P.tasks[(size_t)(rf >> 32)], rec_of(P, rf))) { first_of[k] = (uint32_t)f; break; }
                        }
                        slot = (slot + 1) & (cap - 1);
                    }
                }
            }
        });
        std::vector<int64_t> nfirst((size_t)W + 1, 0);
        slices([&](int sl, int64_t k0, int64_t k1) { int64_t c = 0; for (int64_t k = k0; k < k1; k++) c += first_of[(size_t)k] == (uint32_t)k; nfirst[(size_t)sl + 1] = c; });
        for (int w = 0; w < W; w++) nfirst[(size_t)w + 1] += nfirst[(size_t)w];
        P.name_of_id.assign((size_t)nfirst[(size_t)W], 0);
        slices([&](int sl, int64_t k0, int64_t k1) {
            int64_t id = nfirst[(size_t)sl];
            for (int64_t k = k0; k < k1; k++)
                if (first_of[(size_t)k] == (uint32_t)k) { rec_of(P, order[(size_t)k]).qid = (uint32_t)id; P.name_of_id[(size_t)id] = order[(size_t)k]; id++; }
        });
        slices([&](int, int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++)
                if (first_of[(size_t)k] != (uint32_t)k) rec_of(P, order[(size_t)k]).qid = rec_of(P, order[first_of[(size_t)k]]).qid;
        });
        P.n_qnames = nfirst[(size_t)W];
    }
    tn[2] = now_s();
    // ---- the dictionary of the small columns: combinations numbered in order of first appearance (per slice, then joined in order)
    const Opt &o = P.opt;
    auto tup_of = [&](const WRec &x, uint64_t &key, uint32_t &k2) {
        const bool bases = (x.keep == 2 || o.all_bases);
        uint32_t aux = bases ? x.aux : (x.aux | UZ_AUX_NO_SEQ);
        aux |= (uint32_t)x.simple << UZ_AUX_SIMPLE_SHIFT;
        key = (uint64_t)x.flag | ((uint64_t)x.l_seq << 16) | ((uint64_t)x.n_cigar << 32) | ((uint64_t)x.mapq << 48) | ((uint64_t)(aux & 0xFFu) << 56);
        k2 = (uint32_t)(o.lists ? x.n_low : (uint8_t)0) | ((uint32_t)(o.masks ? x.umask : (uint16_t)0) << 8) | ((uint32_t)x.n_bl << 24);
    };
    if (!P.desc) { // (descriptor route: no link form, no dictionary)
        struct KeyHash { size_t operator()(const std::pair<uint64_t, uint32_t> &k) const { return std::hash<uint64_t>()(k.first * 0x9E3779B97F4A7C15ULL + k.second * 0xC2B2AE3D27D4EB4FULL); } };
        typedef std::unordered_map<std::pair<uint64_t, uint32_t>, uint32_t, KeyHash> Dict;
        std::vector<std::vector<std::pair<uint64_t, uint32_t>>> local((size_t)W);
        std::vector<std::vector<uint32_t>> lidx((size_t)W);
        slices([&](int sl, int64_t k0, int64_t k1) {
            Dict d;
            lidx[(size_t)sl].resize((size_t)(k1 - k0));
            uint64_t lk = ~0ULL; uint32_t lk2 = ~0u, li = 0; // the combination of the record before: most records repeat it
            for (int64_t k = k0; k < k1; k++) {
                uint64_t key; uint32_t k2;
                tup_of(rec_of(P, order[(size_t)k]), key, k2);
                if (key != lk || k2 != lk2) {
                    auto it = d.find({key, k2});
                    if (it == d.end()) { it = d.emplace(std::make_pair(key, k2), (uint32_t)local[(size_t)sl].size()).first; local[(size_t)sl].push_back({key, k2}); }
                    lk = key; lk2 = k2; li = it->second;
                }
                lidx[(size_t)sl][(size_t)(k - k0)] = li;
            }
        });
        Dict dict;
        std::vector<std::vector<uint32_t>> remap((size_t)W);
        for (int w = 0; w < W; w++)
            for (const auto &e : local[(size_t)w]) {
                auto it = dict.find(e);
                if (it == dict.end()) {
                    if (dict.size() >= 65536) fail(UZ_IO_E_RANGE, "more than 65536 combinations of the small columns: stage this batch through the table form");
                    it = dict.emplace(e, (uint32_t)dict.size()).first;
                    P.tup_key.push_back(e.first);
                    P.tup_k2.push_back(e.second);
                }
                remap[(size_t)w].push_back(it->second);
            }
        slices([&](int sl, int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; k++) rec_of(P, order[(size_t)k]).tup = (uint16_t)remap[(size_t)sl][lidx[(size_t)sl][(size_t)(k - k0)]]; });
        P.n_tup = (int64_t)P.tup_key.size();
    }
    tn[3] = now_s();
    // ---- sizes: per slice, then the slices' first offsets
    P.base.assign((size_t)W + 1, SliceBase());
    std::vector<std::vector<int64_t>> cnt((size_t)W, std::vector<int64_t>((size_t)n_ref, 0));
    std::vector<std::vector<int32_t>> span((size_t)W, std::vector<int32_t>((size_t)n_ref, 0));
    std::vector<int> wides((size_t)W, 0);
    slices([&](int sl, int64_t k0, int64_t k1) {
        SliceBase b;
        for (int64_t k = k0; k < k1; k++) {
            const int64_t ref = order[(size_t)k];
            WRec &x = rec_of(P, ref);
            const int32_t tid_k = P.tasks[(size_t)(ref >> 32)].tid;
            if (P.desc) { // the device's table: every CIGAR word, a quality row per record, base rows (all units) of the records with bases
                b.cig += x.n_cigar;
                b.units += UZ_ROW_UNITS(x.l_seq);
                if (x.keep == 2 || o.all_bases) b.seq += UZ_ROW_UNITS(x.l_seq);
                b.names += x.l_name;
                cnt[(size_t)sl][(size_t)tid_k]++;
                span[(size_t)sl][(size_t)tid_k] = std::max(span[(size_t)sl][(size_t)tid_k], x.end - x.pos);
                continue;
            }
            b.omitted += x.simple != 0;
            b.cig += x.simple ? 0 : x.n_cigar;
            b.units += UZ_ROW_UNITS(x.l_seq);
            b.seq += x.n_units; b.exc += x.n_exc; b.qpos += x.n_qpos; b.bl += x.n_bl; b.blu += x.bl_units;
            if (x.l_seq > 256) wides[(size_t)sl] = 1;
            cnt[(size_t)sl][(size_t)tid_k]++;
            span[(size_t)sl][(size_t)tid_k] = std::max(span[(size_t)sl][(size_t)tid_k], x.end - x.pos);
            uint8_t s8, p8;
            int32_t e[4];
            bool has[4];
            b.esc += link_of(P, k, s8, p8, e, has);
        }
        P.base[(size_t)sl + 1] = b;
    });
    for (int w = 0; w < W; w++) P.base[(size_t)w + 1].add(P.base[(size_t)w]);
    P.contig_off.assign((size_t)n_ref + 1, 0);
    P.max_span.assign((size_t)n_ref, 0);
    for (int w = 0; w < W; w++)
        for (int32_t c = 0; c < n_ref; c++) {
            P.contig_off[(size_t)c + 1] += cnt[(size_t)w][(size_t)c];
            P.max_span[(size_t)c] = std::max(P.max_span[(size_t)c], span[(size_t)w][(size_t)c]);
            P.wide |= wides[(size_t)w];
        }
    for (int32_t c = 0; c < n_ref; c++) P.contig_off[(size_t)c + 1] += P.contig_off[(size_t)c];
    for (const Task &T : P.tasks) { P.io_stats[0] += T.file_bytes; P.io_stats[1] += T.n_blocks; P.io_stats[2] += T.n_walked; P.io_stats[6] += T.n_pre; }
    P.io_stats[3] = n;
    P.timing[3] = now_s() - t3;
    if (sub_timing)
        fprintf(stderr, "[uz_stage numbering] %lld records: order %.1f ms | names %.1f ms | dictionary %.1f ms | sizes %.1f ms\n", (long long)n,
                (tn[1] - tn[0]) * 1e3, (tn[2] - tn[1]) * 1e3, (tn[3] - tn[2]) * 1e3, (now_s() - tn[3]) * 1e3);
}

void fill(const uz_stage &P, int threads, uz_reads_packed_view *out) {
    const Opt &o = P.opt;
    const int64_t n = P.n;
    const SliceBase &tot = P.base.back();
    auto need = [&](bool ok, const char *what) { if (!ok) fail(UZ_IO_E_ARG, "uz_stage_fill: the output view %s", what); };
    need(out->seq2 || tot.seq == 0, "needs seq2 (two-bit base rows)");
    need(!out->seq4, "must not set seq4 (the staged form carries two-bit rows)");
    need(out->tup && out->tup_flag && out->tup_l_seq && out->tup_n_cigar && out->tup_mapq && out->tup_aux, "needs the dictionary form (tup, tup_*)");
    need(out->start_d8 && out->pair_d8 && !out->tlen_s && !out->mate_d8 && !out->qname_d8 && !out->start_d && !out->mate_d && !out->qname_d,
         "needs the difference form with eight-bit starts and the pair form of tlen / mate / name id (start_d8, pair_d8)");
    need(tot.esc == 0 || (out->esc16_key && out->esc16_val), "needs the esc16_* list");
    need(!out->end, "must leave `end` out (a BAM record's end is what its CIGAR gives)");
    need(out->cigar_compact != 0, "must set cigar_compact");
    if (o.lists) need(out->tup_n_low && (out->qlow_pos || tot.qpos == 0) && !out->qlow, "needs the list form of the qualities (tup_n_low, qlow_pos)");
    else need(out->qlow && !out->tup_n_low, "needs the quality plane (qlow)");
    if (o.masks) need(out->tup_umask != nullptr, "needs tup_umask");
    if (o.bl) need(out->tup_n_bl && !out->bl_n && (tot.bl == 0 || (out->bl_pos && out->bl_code)) && (!P.wide || out->bl_wide), "needs the list form of the bases (tup_n_bl, bl_pos, bl_code; bl_wide for reads longer than 256 bases)");
    else need(!out->tup_n_bl && !out->bl_n, "must not set bl_n / tup_n_bl (the stage was planned without the list form of the bases)");
    need(!P.wide || out->qlow_pos_wide || !o.lists, "needs qlow_pos_wide (reads longer than 256 bases)");
    if (tot.exc) need(out->exc_rec && out->exc_pos && out->exc_code, "needs the exc_* columns");
    const uz_bamsrc &S = *P.src;
    const int32_t n_ref = (int32_t)S.contigs.size();
    out->n_segs = n; out->n_contigs = n_ref; out->min_base_qual = o.thr; out->n_qnames = (uint32_t)P.n_qnames;
    out->n_cigar_total = tot.cig; out->n_cigar_omitted = tot.omitted; out->n_row_units = tot.units; out->n_seq_units = tot.seq;
    out->n_exc = tot.exc; out->n_qlow_pos = o.lists ? tot.qpos : 0; out->n_tup = P.n_tup; out->n_esc16 = tot.esc;
    out->n_bl = o.bl ? tot.bl : 0; out->n_bl_units = o.bl ? tot.blu : 0;
    auto w = [](const auto *p) { return const_cast<typename std::remove_const<typename std::remove_pointer<decltype(p)>::type>::type *>(p); };
    for (int32_t c = 0; c <= n_ref; c++) w(out->contig_off)[c] = P.contig_off[(size_t)c];
    for (int32_t c = 0; c < n_ref; c++) w(out->max_span)[c] = P.max_span[(size_t)c];
    for (size_t t = 0; t < P.tup_key.size(); t++) {
        const uint64_t key = P.tup_key[t];
        w(out->tup_flag)[t] = (uint16_t)key; w(out->tup_l_seq)[t] = (uint16_t)(key >> 16); w(out->tup_n_cigar)[t] = (uint16_t)(key >> 32);
        w(out->tup_mapq)[t] = (uint8_t)(key >> 48); w(out->tup_aux)[t] = (uint8_t)(key >> 56);
        if (out->tup_n_low) w(out->tup_n_low)[t] = (uint8_t)(P.tup_k2[t] & 0xFF);
        if (out->tup_umask) w(out->tup_umask)[t] = (uint16_t)(P.tup_k2[t] >> 8);
        if (out->tup_n_bl) w(out->tup_n_bl)[t] = (uint8_t)(P.tup_k2[t] >> 24);
    }
    const bool wide = out->qlow_pos_wide != 0;
    const bool blw = out->bl_wide != 0;
    if (o.bl && tot.bl) memset(w(out->bl_code), 0, (size_t)(tot.bl + 3) / 4); // (two-bit fields of neighbouring slices share bytes: OR-ed in below)
    const int W = (int)P.cut.size() - 1;
    parallel_slices(W, std::min(W, resolve_threads(threads)), [&](int64_t s0, int64_t s1, int) {
        for (int64_t sl = s0; sl < s1; sl++) {
            SliceBase at = P.base[(size_t)sl];
            for (int64_t k = P.cut[(size_t)sl]; k < P.cut[(size_t)sl + 1]; k++) {
                const int64_t ref = P.order[(size_t)k];
                const Task &T = P.tasks[(size_t)(ref >> 32)];
                const WRec &x = T.recs[(size_t)(ref & 0xFFFFFFFF)];
                uint8_t s8, p8;
                int32_t e[4];
                bool has[4];
                link_of(P, k, s8, p8, e, has);
                w(out->start_d8)[k] = s8;
                w(out->pair_d8)[k] = p8;
                for (int c = 0; c < 4; c++)
                    if (has[c]) { w(out->esc16_key)[at.esc] = ((uint64_t)k << 2) | (uint64_t)c; w(out->esc16_val)[at.esc] = e[c]; at.esc++; }
                w(out->tup)[k] = x.tup;
                if (!x.simple) { memcpy(w(out->cigar) + at.cig, T.cigars.data() + x.cigar_at, (size_t)x.n_cigar * 4); at.cig += x.n_cigar; }
                const uint8_t *pay = T.pay.data() + x.pay_at;
                if (x.n_units) { memcpy(w(out->seq2) + (size_t)at.seq * UZ_SEQ2_UNIT_BYTES, pay, (size_t)x.n_units * 8); at.seq += x.n_units; }
                pay += (size_t)x.n_units * 8;
                for (int j = 0; j < (int)x.n_bl; j++) {
                    if (blw) { w(out->bl_pos)[2 * at.bl] = pay[3 * j]; w(out->bl_pos)[2 * at.bl + 1] = pay[3 * j + 1]; }
                    else w(out->bl_pos)[at.bl] = pay[3 * j];
                    if (pay[3 * j + 2]) __atomic_fetch_or(w(out->bl_code) + (at.bl >> 2), (uint8_t)(pay[3 * j + 2] << (2 * (at.bl & 3))), __ATOMIC_RELAXED);
                    at.bl++;
                }
                pay += (size_t)x.n_bl * 3;
                for (int j = 0; j < (int)x.n_exc; j++) {
                    w(out->exc_rec)[at.exc] = (uint32_t)k; w(out->exc_pos)[at.exc] = (uint16_t)(pay[4 * j] | (pay[4 * j + 1] << 8)); w(out->exc_code)[at.exc] = pay[4 * j + 2];
                    at.exc++;
                }
                pay += (size_t)x.n_exc * 4;
                for (int j = 0; j < (int)x.n_qpos; j++) {
                    if (wide) { w(out->qlow_pos)[2 * at.qpos] = pay[2 * j]; w(out->qlow_pos)[2 * at.qpos + 1] = pay[2 * j + 1]; }
                    else w(out->qlow_pos)[at.qpos] = pay[2 * j];
                    at.qpos++;
                }
                pay += (size_t)x.n_qpos * 2;
                if (!o.lists) { const size_t ub = (size_t)UZ_ROW_UNITS(x.l_seq) * UZ_QLOW_UNIT_BYTES; memcpy(w(out->qlow) + (size_t)at.units * UZ_QLOW_UNIT_BYTES, pay, ub); }
                at.units += UZ_ROW_UNITS(x.l_seq);
            }
        }
    });
}

template <typename F>
int guarded(F &&fn) {
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

int uz_bamsrc_open(const char *path, const char *bai_path, int64_t head_records, uz_bamsrc **out) {
    if (!path || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_bamsrc *s = new uz_bamsrc();
    const int rc = guarded([&] { open_source(*s, path, bai_path, head_records); });
    if (rc != UZ_IO_OK) { delete s; return rc; }
    *out = s;
    return UZ_IO_OK;
}
void uz_bamsrc_close(uz_bamsrc *s) { delete s; }
int32_t uz_bamsrc_n_contigs(const uz_bamsrc *s) { return s ? (int32_t)s->contigs.size() : 0; }
const char *uz_bamsrc_contig_name(const uz_bamsrc *s, int32_t i) { return (s && i >= 0 && (size_t)i < s->contigs.size()) ? s->contigs[(size_t)i].c_str() : nullptr; }
int32_t uz_bamsrc_contig_length(const uz_bamsrc *s, int32_t i) { return (s && i >= 0 && (size_t)i < s->contig_len.size()) ? s->contig_len[(size_t)i] : -1; }
int64_t uz_bamsrc_tlen_head(const uz_bamsrc *s, int32_t *out, int64_t cap) {
    if (!s || !out || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, (int64_t)s->tlen_head.size());
    memcpy(out, s->tlen_head.data(), (size_t)k * sizeof(int32_t));
    return k;
}
const char *uz_inflate_backend(void) { return libdeflate().ok ? "libdeflate" : "zlib"; }
int uz_io_default_threads(void) { return resolve_threads(0); }
int uz_io_cpu_quota(void) { return cgroup_cpu_quota(); }

int uz_bam_stage_begin(const uz_bamsrc *src, int64_t n_fetch, const int32_t *tid, const int32_t *lo, const int32_t *hi, const uint16_t *extra, int flags,
                       int min_base_qual, int threads, uz_stage **out) {
    if (!src || !out || (n_fetch > 0 && (!tid || !lo || !hi))) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_stage *P = new uz_stage();
    P->src = src;
    P->opt.all_bases = (flags & UZ_STAGE_ALL_BASES) != 0;
    P->opt.lists = !(flags & UZ_STAGE_PLANE);
    P->opt.masks = (flags & UZ_STAGE_UNIT_MASKS) && !P->opt.all_bases && P->opt.lists;
    P->opt.wide_none = P->opt.masks && (flags & UZ_STAGE_WIDE_NO_UNITS);
    P->opt.bl = P->opt.masks && (flags & UZ_STAGE_BASE_LISTS);
    P->small_tasks = (flags & UZ_STAGE_SMALL_TASKS) != 0;
    P->opt.thr = min_base_qual < 0 ? 0 : (min_base_qual > 255 ? 256 : min_base_qual);
    const int rc = guarded([&] { plan_begin(*P, n_fetch, tid, lo, hi, extra, threads); });
    if (rc != UZ_IO_OK) { delete P; return rc; }
    *out = P;
    return UZ_IO_OK;
}
int uz_bam_stage_finish(uz_stage *P) {
    if (!P || !P->begun || P->finished) { last_error = "uz_bam_stage_finish: no plan that was begun and not yet finished"; return UZ_IO_E_ARG; }
    P->finished = true;
    return guarded([&] { plan_finish(*P); });
}
int uz_bam_stage_plan(const uz_bamsrc *src, int64_t n_fetch, const int32_t *tid, const int32_t *lo, const int32_t *hi, const uint16_t *extra, int flags,
                      int min_base_qual, int threads, uz_stage **out) {
    const int rc = uz_bam_stage_begin(src, n_fetch, tid, lo, hi, extra, flags, min_base_qual, threads, out);
    if (rc != UZ_IO_OK) return rc;
    const int rc2 = uz_bam_stage_finish(*out);
    if (rc2 != UZ_IO_OK) { delete *out; *out = nullptr; }
    return rc2;
}

// The blocks the walk of a begun plan will read -- every task's spans from their first block to the block the linear index puts the end
// of the task's reach in -- copied back to back (whole blocks, framing included) into `comp` (cap bytes; NULL: sizes only).
// n_blocks / comp_bytes / out_bytes: how many, their compressed and inflated size; in_off [n_blocks]: where each block's DEFLATE stream
// starts in comp; out_off [n_blocks + 1]: where its bytes belong in the inflated buffer (uz_stage_set_inflated).
int uz_stage_gather_blocks(uz_stage *P, uint8_t *comp, int64_t cap, int64_t *in_off, int64_t *out_off, int64_t *n_blocks, int64_t *comp_bytes,
                           int64_t *out_bytes) {
    if (!P || !P->begun || P->finished) { last_error = "uz_stage_gather_blocks: between uz_bam_stage_begin and uz_bam_stage_finish"; return UZ_IO_E_ARG; }
    return guarded([&] {
        const uz_bamsrc &S = *P->src;
        if (P->n_pre_blocks == 0 && P->pre_bytes == 0) { // list them once
            parallel_slices((int64_t)P->tasks.size(), workers_for((int64_t)P->tasks.size(), resolve_threads(P->threads), 16), [&](int64_t i0, int64_t i1, int) {
                for (int64_t i = i0; i < i1; i++) {
                    Task &T = P->tasks[(size_t)i];
                    T.pre.clear();
                    T.span_stop.assign(T.spans.size(), -1);
                    int64_t last = -1;
                    for (size_t ci = 0; ci < T.spans.size(); ci++) {
                        const Chunk &c = T.spans[ci];
                        int64_t coff = (int64_t)(c.beg >> 16);
                        const int64_t stop = (int64_t)(std::min(c.end, std::max(T.est_end, c.beg)) >> 16); // (inclusive: the block that holds the end)
                        // One block more where the estimate cut the span short: est_end is the first record that OVERLAPS the window behind the task's
                        // reach -- a read that starts in front of the window's edge -- and the walk goes on to the first record that STARTS at or behind
                        // the reach's end, a read length of records further: usually the same block, now and then the next one (round 5: such a task
                        // ended "incomplete" on the device and was walked again by the host, two of 2 059 per feed pass -- with the joins on the device
                        // the one thing left that makes the host inflate and walk).
                        int extra = (int64_t)(c.end >> 16) > stop ? 1 : 0;
                        while (coff <= stop || extra > 0) {
                            if (coff > stop) extra--;
                            BlockHdr h;
                            if (!block_at(S, coff, h)) break;
                            if (coff > last) { T.pre.push_back(PreBlk{coff, 0, h.isize, h.crc, (uint32_t)h.blen, (uint32_t)(h.cdata - (size_t)coff)}); last = coff; }
                            T.span_stop[ci] = coff;
                            coff += (int64_t)h.blen;
                        }
                    }
                }
            });
            int64_t nb = 0, cb = 0, ob = 0;
            for (Task &T : P->tasks)
                for (PreBlk &b : T.pre) { b.at = ob; ob += b.isize; cb += b.blen; nb++; }
            P->n_pre_blocks = nb; P->pre_bytes = ob;
            P->io_stats[7] = cb;
        }
        if (n_blocks) *n_blocks = P->n_pre_blocks;
        if (comp_bytes) *comp_bytes = P->io_stats[7];
        if (out_bytes) *out_bytes = P->pre_bytes;
        if (!comp) return;
        if (cap < P->io_stats[7] || !in_off || !out_off) fail(UZ_IO_E_ARG, "uz_stage_gather_blocks: buffer too small, or null offset arrays");
        // the first block of every task in comp / in the block list
        std::vector<int64_t> c0(P->tasks.size() + 1, 0), b0(P->tasks.size() + 1, 0);
        for (size_t i = 0; i < P->tasks.size(); i++) {
            int64_t cb = 0;
            for (const PreBlk &b : P->tasks[i].pre) cb += (int64_t)b.blen;
            c0[i + 1] = c0[i] + cb; b0[i + 1] = b0[i] + (int64_t)P->tasks[i].pre.size();
        }
        parallel_dynamic((int64_t)P->tasks.size(), resolve_threads(P->threads), [&](int64_t i, int) {
            const Task &T = P->tasks[(size_t)i];
            int64_t at = c0[(size_t)i], k = b0[(size_t)i];
            for (const PreBlk &b : T.pre) { // (the blocks' headers were read when the blocks were listed)
                memcpy(comp + at, S.map + b.coff, b.blen);
                in_off[k] = at + (int64_t)b.hdr;
                out_off[k] = b.at;
                at += (int64_t)b.blen; k++;
            }
        });
        out_off[P->n_pre_blocks] = P->pre_bytes;
    });
}
// the blocks of uz_stage_gather_blocks, inflated (out_off of that call says where each lies); must stay valid until uz_bam_stage_finish
// has returned, which copies the blocks in as it walks and holds each against its CRC-32
int uz_stage_set_inflated(uz_stage *P, const uint8_t *inflated) {
    if (!P || !P->begun || P->finished) { last_error = "uz_stage_set_inflated: between uz_bam_stage_begin and uz_bam_stage_finish"; return UZ_IO_E_ARG; }
    P->inflated = inflated;
    return UZ_IO_OK;
}

void uz_stage_sizes(const uz_stage *P, int64_t out[16]) {
    memset(out, 0, 16 * sizeof(int64_t));
    if (!P) return;
    const SliceBase &t = P->base.back();
    out[0] = P->n; out[1] = t.cig; out[2] = t.omitted; out[3] = t.units; out[4] = t.seq; out[5] = t.exc; out[6] = t.qpos; out[7] = P->wide;
    out[8] = P->n_tup; out[9] = t.esc; out[10] = P->n_qnames; out[11] = P->opt.masks ? 1 : 0;
    out[12] = P->opt.bl ? 1 : 0; out[13] = t.bl; out[14] = t.blu;
}
void uz_stage_io_stats(const uz_stage *P, int64_t out[8]) { for (int k = 0; k < 8; k++) out[k] = P ? P->io_stats[k] : 0; }
void uz_stage_timing(const uz_stage *P, double out[6]) { for (int k = 0; k < 6; k++) out[k] = P ? P->timing[k] : 0.0; }

int uz_stage_fill(const uz_stage *P, int threads, uz_reads_packed_view *out) {
    if (!P || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    if (P->desc) { last_error = "uz_stage_fill: this plan was finished from the device's walk (uz_bam_stage_finish_desc): its records are listed by uz_stage_kept"; return UZ_IO_E_ARG; }
    const double t0 = now_s();
    const int rc = guarded([&] { fill(*P, threads, out); });
    const_cast<uz_stage *>(P)->timing[4] = now_s() - t0;
    return rc;
}

const char *uz_stage_qname(const uz_stage *P, uint32_t id, int32_t *len) {
    if (!P || P->desc || id >= P->name_of_id.size()) return nullptr; // (descriptor route: the name bytes stayed on the device)
    const int64_t ref = P->name_of_id[id];
    const Task &T = P->tasks[(size_t)(ref >> 32)];
    const WRec &x = T.recs[(size_t)(ref & 0xFFFFFFFF)];
    P->name_tmp.assign((const char *)T.names.data() + x.name_at, x.l_name);
    if (len) *len = x.l_name;
    return P->name_tmp.c_str();
}

// many names at once: the bytes of names ids[0 .. n) back to back into buf (no terminators), off[k] .. off[k + 1] the k-th; returns the bytes
// needed (call with cap = 0 for the size), -1 for an id out of range
int64_t uz_stage_qnames(const uz_stage *P, const uint32_t *ids, int64_t n, char *buf, int64_t cap, int64_t *off) {
    if (!P || P->desc || (n > 0 && !ids)) return -1;
    int64_t at = 0;
    for (int64_t k = 0; k < n; k++) {
        if (ids[k] >= P->name_of_id.size()) return -1;
        const int64_t ref = P->name_of_id[ids[k]];
        const Task &T = P->tasks[(size_t)(ref >> 32)];
        const WRec &x = T.recs[(size_t)(ref & 0xFFFFFFFF)];
        if (off) off[k] = at;
        if (buf && at + x.l_name <= cap) memcpy(buf + at, T.names.data() + x.name_at, x.l_name);
        at += x.l_name;
    }
    if (off) off[n] = at;
    return at;
}

// ---- the descriptor route (uz_bamwalk.h): the walk runs on the device, the batch-wide joins here
namespace {
// The device's walk tasks (uz_stage::subs).  Without UZ_STAGE_SMALL_TASKS: the host's tasks as they are.  With it: every group of a task's reach
// intervals that spans at most SUB_EXTENT bases is a task of its own.  Its walk starts where the file's linear index puts the first record that
// overlaps the 16 kb window its first interval begins in -- every record overlapping the group lies at or behind that offset -- and stops at the
// first record that starts at or behind the end of its last interval, as the host's walk of the whole task would.  A record that reaches from one
// group into the next is met by both: uz_stage_merge_subtasks drops the later copy (it starts in front of the earlier group's stop).
void build_subtasks(uz_stage &P) {
    if (!P.subs.empty() || P.tasks.empty()) return;
    const int32_t SUB_EXTENT = 32768;
    P.sub_preflag.assign(P.tasks.size(), 0);
    for (size_t i = 0; i < P.tasks.size(); i++) {
        const Task &T = P.tasks[i];
        const BaiRef &ref = P.src->refs[(size_t)T.tid];
        const int32_t nr = (int32_t)T.reach.size();
        int32_t g0 = 0;
        for (int32_t k = 1; k <= nr; k++) {
            if (k < nr && (!P.small_tasks || T.reach[(size_t)k].second - T.reach[(size_t)g0].first <= SUB_EXTENT)) continue;
            uz_stage::SubTask S;
            S.host = (int32_t)i; S.r0 = g0; S.r1 = k; S.s0 = 0;
            S.tb = k == nr ? T.b : T.reach[(size_t)k - 1].second;
            S.beg = T.spans.empty() ? 0 : T.spans.front().beg;
            if (g0 > 0 && !T.spans.empty()) { // (the task's first group starts where the task starts)
                const size_t w = (size_t)(std::max<int64_t>(T.reach[(size_t)g0].first, 0) >> 14);
                uint64_t off = ref.linear.empty() ? 0 : ref.linear[std::min(w, ref.linear.size() - 1)];
                off = std::max(off, T.spans.front().beg);
                size_t si = 0;
                while (si + 1 < T.spans.size() && T.spans[si].end <= off) si++;
                S.s0 = (int32_t)si;
                S.beg = std::max(off, T.spans[si].beg);
            }
            P.subs.push_back(S);
            g0 = k;
        }
        if (nr == 0) { uz_stage::SubTask S{(int32_t)i, 0, 0, 0, T.b, T.spans.empty() ? 0 : T.spans.front().beg}; P.subs.push_back(S); }
    }
}
} // namespace

void uz_stage_walk_plan_sizes(const uz_stage *P, int64_t out[8]) {
    memset(out, 0, 8 * sizeof(int64_t));
    if (!P) return;
    build_subtasks(*const_cast<uz_stage *>(P));
    out[0] = (int64_t)P->subs.size();
    for (const auto &S : P->subs) out[1] += (int64_t)P->tasks[(size_t)S.host].spans.size() - S.s0;
    for (const Task &T : P->tasks) out[2] += (int64_t)T.reach.size();
    for (const auto &v : P->fx) out[3] += (int64_t)v.size();
    out[4] = P->n_pre_blocks;
    out[5] = (int64_t)P->tasks.size();
}

int uz_stage_walk_plan(uz_stage *P, int32_t *task, int64_t *span, int32_t *reach, int32_t *fetch, int64_t *blk_coff, uint32_t *blk_crc) {
    if (!P || !P->begun || P->finished || !task || !span || !reach || !fetch || !blk_coff) { last_error = "uz_stage_walk_plan: between uz_stage_gather_blocks and the finish, every array set"; return UZ_IO_E_ARG; }
    return guarded([&] {
        if (P->n_pre_blocks == 0 && P->pre_bytes == 0 && !P->tasks.empty()) fail(UZ_IO_E_ARG, "uz_stage_walk_plan: call uz_stage_gather_blocks first (the block table is laid out there)");
        build_subtasks(*P);
        std::vector<int64_t> fbase(P->fx.size() + 1, 0);
        for (size_t t = 0; t < P->fx.size(); t++) {
            fbase[t + 1] = fbase[t] + (int64_t)P->fx[t].size();
            for (size_t k = 0; k < P->fx[t].size(); k++) {
                int32_t *f = fetch + 3 * (fbase[t] + (int64_t)k);
                f[0] = P->fx[t][k].lo; f[1] = P->fx[t][k].hi; f[2] = (int32_t)P->fx[t][k].extra;
            }
        }
        // per host task: where its reach intervals and its blocks lie in the flat arrays
        std::vector<int64_t> rbase(P->tasks.size() + 1, 0), bbase(P->tasks.size() + 1, 0);
        for (size_t i = 0; i < P->tasks.size(); i++) {
            const Task &T = P->tasks[i];
            rbase[i + 1] = rbase[i] + (int64_t)T.reach.size(); bbase[i + 1] = bbase[i] + (int64_t)T.pre.size();
            for (size_t k = 0; k < T.reach.size(); k++) { reach[2 * (rbase[i] + (int64_t)k)] = T.reach[k].first; reach[2 * (rbase[i] + (int64_t)k) + 1] = T.reach[k].second; }
            for (size_t k = 0; k < T.pre.size(); k++) { blk_coff[bbase[i] + (int64_t)k] = T.pre[k].coff; if (blk_crc) blk_crc[bbase[i] + (int64_t)k] = T.pre[k].crc; }
        }
        int64_t si = 0;
        for (size_t u = 0; u < P->subs.size(); u++) {
            const auto &S = P->subs[u];
            const Task &T = P->tasks[(size_t)S.host];
            const int64_t bi = bbase[(size_t)S.host];
            int32_t *tc = task + UZ_WALK_TASK_COLS * u;
            tc[0] = T.tid; tc[1] = S.tb; tc[2] = (int32_t)si; tc[3] = (int32_t)(si + (int64_t)T.spans.size() - S.s0);
            tc[4] = (int32_t)(rbase[(size_t)S.host] + S.r0); tc[5] = (int32_t)(rbase[(size_t)S.host] + S.r1);
            tc[6] = (int32_t)(fbase[(size_t)T.tid] + (int64_t)T.f0); tc[7] = (int32_t)(fbase[(size_t)T.tid] + (int64_t)T.f1); // (every fetch of the host task: a record of this group may be returned by a neighbour's)
            tc[8] = P->fx_max_len[(size_t)T.tid]; tc[9] = S.host;
            for (size_t sp = (size_t)S.s0; sp < T.spans.size(); sp++) {
                const Chunk &c = T.spans[sp];
                const uint64_t beg = sp == (size_t)S.s0 ? S.beg : c.beg;
                int64_t *sc = span + UZ_WALK_SPAN_COLS * si++;
                const int64_t c0 = (int64_t)(beg >> 16), stop = sp < T.span_stop.size() ? T.span_stop[sp] : (int64_t)(std::min(c.end, std::max(T.est_end, c.beg)) >> 16);
                auto lo = std::lower_bound(T.pre.begin(), T.pre.end(), c0, [](const PreBlk &b, int64_t key) { return b.coff < key; });
                auto hi = std::upper_bound(T.pre.begin(), T.pre.end(), stop, [](int64_t key, const PreBlk &b) { return key < b.coff; });
                sc[0] = (int64_t)beg; sc[1] = (int64_t)c.end;
                if (lo == T.pre.end() || lo->coff != c0 || hi <= lo) { // no block there: the walk ends at once, as the host's does at the end of the file ...
                    sc[2] = sc[3] = 0; sc[4] = sc[5] = bi;
                    if (beg != c.beg) P->sub_preflag[(size_t)S.host] = 1; // ... but a sub-task that starts behind what was gathered says nothing: the host walks the task
                    continue;
                }
                sc[2] = lo->at + (int64_t)(beg & 0xFFFF);
                sc[3] = (hi - 1)->at + (int64_t)(hi - 1)->isize;
                sc[4] = bi + (int64_t)(lo - T.pre.begin()); sc[5] = bi + (int64_t)(hi - T.pre.begin());
            }
        }
    });
}

/* The descriptors of the device's walk tasks (sub-tasks of the host's: uz_stage_walk_plan under UZ_STAGE_SMALL_TASKS) joined per host task, in place:
 * the sub-tasks of a task in order, without the records a later sub-task met again (they start in front of the stop of the one before).  In:
 * d [d_first[n_sub]], d_first [n_sub + 1], d_flags / d_walked [n_sub]; out: d compacted, h_first [n_tasks + 1], h_flags / h_walked [n_tasks]. */
int uz_stage_merge_subtasks(const uz_stage *P, uz_walk_desc *d, const int64_t *d_first, const int32_t *d_flags, const int64_t *d_walked, int64_t *h_first,
                            int32_t *h_flags, int64_t *h_walked) {
    if (!P || !d_first || !h_first || !h_flags || !h_walked) { last_error = "uz_stage_merge_subtasks: null argument"; return UZ_IO_E_ARG; }
    return guarded([&] {
        const size_t nt = P->tasks.size();
        for (size_t i = 0; i < nt; i++) { h_flags[i] = P->sub_preflag.empty() ? 0 : (P->sub_preflag[i] ? UZ_WALK_TASK_INCOMPLETE : 0); h_walked[i] = 0; }
        for (size_t u = 0; u < P->subs.size(); u++) { h_flags[(size_t)P->subs[u].host] |= d_flags ? d_flags[u] : 0; h_walked[(size_t)P->subs[u].host] += d_walked ? d_walked[u] : 0; }
        int64_t out = 0;
        size_t u = 0;
        for (size_t i = 0; i < nt; i++) {
            h_first[i] = out;
            int32_t stop_before = INT32_MIN;
            for (; u < P->subs.size() && (size_t)P->subs[u].host == i; u++) {
                if (!h_flags[i])
                    for (int64_t j = d_first[u]; j < d_first[u + 1]; j++) {
                        if (d[j].pos < stop_before) continue; // (the sub-task before walked it, and kept it: it overlaps an interval of this one, so it lay in reach there too)
                        uz_walk_desc x = d[j];
                        x.task = (uint32_t)i;
                        d[out++] = x;
                    }
                stop_before = P->subs[u].tb;
            }
        }
        h_first[nt] = out;
    });
}

int uz_bam_stage_finish_desc(uz_stage *P, const uz_walk_desc *d, const int64_t *d_first, const int32_t *d_flags, const int64_t *d_walked) {
    if (!P || !P->begun || P->finished || !d_first) { last_error = "uz_bam_stage_finish_desc: no plan that was begun and not yet finished"; return UZ_IO_E_ARG; }
    P->finished = true;
    P->desc = true;
    for (Task &T : P->tasks) T.by_hash = true;
    return guarded([&] { plan_finish(*P, d ? d : (const uz_walk_desc *)"", d_first, d_flags, d_walked); });
}

/* the same from the descriptors of the walk plan's SUB-tasks as the device hands them back (d_first [n_sub + 1], d_flags / d_walked [n_sub]): joined per
 * task of the stage as uz_stage_merge_subtasks joins them, without the copy */
int uz_bam_stage_finish_sub(uz_stage *P, const uz_walk_desc *d, const int64_t *d_first, const int32_t *d_flags, const int64_t *d_walked) {
    if (!P || !P->begun || P->finished || !d_first) { last_error = "uz_bam_stage_finish_sub: no plan that was begun and not yet finished"; return UZ_IO_E_ARG; }
    P->finished = true;
    P->desc = true;
    for (Task &T : P->tasks) T.by_hash = true;
    return guarded([&] {
        build_subtasks(*P);
        std::vector<int32_t> hfl(P->tasks.size(), 0);
        for (size_t i = 0; i < P->tasks.size(); i++) hfl[i] = (!P->sub_preflag.empty() && P->sub_preflag[i]) ? UZ_WALK_TASK_INCOMPLETE : 0;
        for (size_t u = 0; u < P->subs.size(); u++) hfl[(size_t)P->subs[u].host] |= d_flags ? d_flags[u] : 0;
        plan_finish(*P, d ? d : (const uz_walk_desc *)"", d_first, hfl.data(), d_walked, true);
    });
}

/* totals: [0] records, [1] CIGAR words, [2] row units, [3] base-row units, [4] query names, [5] aux bytes, [6] tasks the host walked itself, [7] name bytes */
void uz_stage_kept_sizes(const uz_stage *P, int64_t out[8]) {
    memset(out, 0, 8 * sizeof(int64_t));
    if (!P || !P->desc || P->base.empty()) return;
    const SliceBase &t = P->base.back();
    out[0] = P->n; out[1] = t.cig; out[2] = t.units; out[3] = t.seq; out[4] = P->n_qnames; out[7] = t.names;
    for (const Task &T : P->tasks) { out[5] += (int64_t)T.raw.size(); out[6] += !T.raw.empty(); }
}

int uz_stage_kept(const uz_stage *P, int threads, uz_kept_rec *out, int64_t *contig_off, int32_t *max_span, uint8_t *aux, int64_t aux_cap) {
    if (!P || !P->finished || !P->desc || !contig_off || !max_span || (P->n && !out)) { last_error = "uz_stage_kept: a plan finished by uz_bam_stage_finish_desc"; return UZ_IO_E_ARG; }
    return guarded([&] {
        const int32_t n_ref = (int32_t)P->src->contigs.size();
        for (int32_t c = 0; c <= n_ref; c++) contig_off[c] = P->contig_off[(size_t)c];
        for (int32_t c = 0; c < n_ref; c++) max_span[c] = P->max_span[(size_t)c];
        std::vector<int64_t> abase(P->tasks.size() + 1, 0);
        for (size_t i = 0; i < P->tasks.size(); i++) abase[i + 1] = abase[i] + (int64_t)P->tasks[i].raw.size();
        if (abase.back() > aux_cap || (abase.back() && !aux)) fail(UZ_IO_E_ARG, "uz_stage_kept: aux buffer too small");
        for (size_t i = 0; i < P->tasks.size(); i++)
            if (!P->tasks[i].raw.empty()) memcpy(aux + abase[i], P->tasks[i].raw.data(), P->tasks[i].raw.size());
        const int W = (int)P->cut.size() - 1;
        const bool all_bases = P->opt.all_bases;
        parallel_slices(W, std::min(W, resolve_threads(threads)), [&](int64_t s0, int64_t s1, int) {
            for (int64_t sl = s0; sl < s1; sl++) {
                SliceBase at = P->base[(size_t)sl];
                for (int64_t k = P->cut[(size_t)sl]; k < P->cut[(size_t)sl + 1]; k++) {
                    const int64_t ref = P->order[(size_t)k];
                    const size_t ti = (size_t)(ref >> 32);
                    const Task &T = P->tasks[ti];
                    const WRec &x = T.recs[(size_t)(ref & 0xFFFFFFFF)];
                    uz_kept_rec &o = out[k];
                    o.src = T.raw.empty() ? x.src : (UZ_WALK_SRC_AUX | (uint64_t)(abase[ti] + (int64_t)x.pay_at + 4));
                    o.qname = x.qid;
                    const int64_t m = final_ref(*P, x.mate_ref);
                    o.mate = m < 0 ? -1 : (int32_t)rec_of(*P, m).gidx;
                    const uint32_t units = UZ_ROW_UNITS(x.l_seq);
                    const bool bases = x.keep == 2 || all_bases;
                    o.cig_off = (uint32_t)at.cig; o.unit_off = (uint32_t)at.units; o.seq_off = bases ? (uint32_t)at.seq : UZ_KEPT_NO_SEQ; o.name_off = (uint32_t)at.names;
                    at.cig += x.n_cigar; at.units += units; if (bases) at.seq += units; at.names += x.l_name;
                }
            }
        });
    });
}

/* The host's twin of k_bam_walk (csrc/k_bamwalk.hip): the same descriptors from the host's own walk of every task -- what the device's are held
 * against (tests/test_bamwalk_gpu.py), and the descriptor route without a device (tests/test_stage_desc.py).  Call it with out == NULL for the
 * counts (d_first, d_walked), then with the buffer.  src = where the record lies in the buffer uz_stage_gather_blocks lays out. */
int uz_stage_walk_host(uz_stage *P, uz_walk_desc *out, int64_t cap, int64_t *d_first, int64_t *d_walked) {
    if (!P || !P->begun || P->finished || !d_first) { last_error = "uz_stage_walk_host: between uz_stage_gather_blocks and the finish"; return UZ_IO_E_ARG; }
    return guarded([&] {
        const size_t nt = P->tasks.size();
        if (P->twin.size() != nt) {
            // the device's walk tasks (the host's tasks, or their sub-tasks under UZ_STAGE_SMALL_TASKS), each walked by the host's own walk_task on a task
            // of its own -- then joined per host task exactly as the device's descriptors are (uz_stage_merge_subtasks)
            build_subtasks(*P);
            const size_t ns = P->subs.size();
            std::vector<std::vector<uz_walk_desc>> part(ns);
            std::vector<int64_t> walked(ns, 0);
            P->desc = true;
            const int w = (int)std::min<int64_t>(std::max(1, P->threads), std::max<int64_t>(1, (int64_t)ns));
            std::vector<std::unique_ptr<Scratch>> scr((size_t)w);
            parallel_dynamic((int64_t)ns, w, [&](int64_t u, int k) {
                if (!scr[(size_t)k]) scr[(size_t)k].reset(new Scratch());
                const auto &S = P->subs[(size_t)u];
                const Task &H = P->tasks[(size_t)S.host];
                Scratch &W = *scr[(size_t)k];
                Task T; // the sub-task as a task of its own
                T.tid = H.tid; T.b = S.tb; T.f0 = H.f0; T.f1 = H.f1; T.est_end = H.est_end; T.by_hash = true;
                T.reach.assign(H.reach.begin() + S.r0, H.reach.begin() + S.r1);
                T.a = T.reach.empty() ? H.a : T.reach.front().first;
                for (size_t sp = (size_t)S.s0; sp < H.spans.size(); sp++) T.spans.push_back(Chunk{sp == (size_t)S.s0 ? S.beg : H.spans[sp].beg, H.spans[sp].end});
                walk_task(*P, T, W, (size_t)S.host, false);
                walked[(size_t)u] = T.n_walked;
                auto &v = part[(size_t)u];
                v.reserve(W.all.size());
                for (const WRec &r : W.all) {
                    uz_walk_desc x;
                    memset(&x, 0, sizeof(x));
                    x.voff = r.voff; x.h1 = r.nhash; x.h2 = r.nhash2;
                    const int64_t coff = (int64_t)(r.voff >> 16);
                    auto it = std::lower_bound(H.pre.begin(), H.pre.end(), coff, [](const PreBlk &b, int64_t key) { return b.coff < key; });
                    x.src = (it != H.pre.end() && it->coff == coff) ? (uint64_t)(it->at + (int64_t)(r.voff & 0xFFFF) + 4) : ~0ULL; // (a block the gather did not list)
                    x.pos = r.pos; x.end = r.end; x.tlen = r.tlen; x.mpos = r.mpos; x.mtid = r.mtid;
                    x.task = (uint32_t)u; x.flag = r.flag; x.l_seq = r.l_seq; x.n_cigar = r.n_cigar; x.mapq = r.mapq; x.l_name = r.l_name;
                    x.direct = r.keep == 2;
                    v.push_back(x);
                }
            });
            std::vector<int64_t> sf(ns + 1, 0);
            for (size_t u = 0; u < ns; u++) sf[u + 1] = sf[u] + (int64_t)part[u].size();
            std::vector<uz_walk_desc> flat((size_t)sf[ns] + 1);
            for (size_t u = 0; u < ns; u++)
                if (!part[u].empty()) memcpy(flat.data() + sf[u], part[u].data(), part[u].size() * sizeof(uz_walk_desc));
            std::vector<int64_t> hf(nt + 1, 0), hw(nt, 0);
            std::vector<int32_t> hfl(nt, 0), zero(ns, 0);
            const std::vector<uint8_t> keep_flags = P->sub_preflag;
            std::fill(P->sub_preflag.begin(), P->sub_preflag.end(), 0); // (the host's own walk reads any block: nothing is handed back here)
            const int rc = uz_stage_merge_subtasks(P, flat.data(), sf.data(), zero.data(), walked.data(), hf.data(), hfl.data(), hw.data());
            P->sub_preflag = keep_flags;
            if (rc) fail(rc, "%s", last_error.c_str());
            P->twin.assign(nt, {});
            for (size_t i = 0; i < nt; i++) { P->twin[i].assign(flat.begin() + hf[i], flat.begin() + hf[i + 1]); P->tasks[i].n_walked = hw[i]; }
        }
        d_first[0] = 0;
        for (size_t i = 0; i < nt; i++) { d_first[i + 1] = d_first[i] + (int64_t)P->twin[i].size(); if (d_walked) d_walked[i] = P->tasks[i].n_walked; }
        if (!out) return;
        if (cap < d_first[nt]) fail(UZ_IO_E_ARG, "uz_stage_walk_host: descriptor buffer too small");
        for (size_t i = 0; i < nt; i++)
            if (!P->twin[i].empty()) memcpy(out + d_first[i], P->twin[i].data(), P->twin[i].size() * sizeof(uz_walk_desc));
    });
}

/* ---- the joins on the device (include/uz_bamwalk.h: uz_bam_join).  The host's share of that route: the records it has to walk itself, handed to the
 * device as descriptors (task = UZ_WALK_TASK_JOIN | join task: the stage task, or n_tasks + k for the k-th look-up through the index) whose bytes lie
 * in the batch's aux store (src = UZ_WALK_SRC_AUX | offset of the fixed part). */
namespace {
void x_append(uz_stage &P, const Task &T, const WRec &r, const std::vector<uint8_t> &raw, uint32_t jtask, bool direct) {
    uz_walk_desc x;
    memset(&x, 0, sizeof(x));
    x.voff = r.voff; x.h1 = r.nhash; x.h2 = r.nhash2;
    x.src = UZ_WALK_SRC_AUX | (uint64_t)(P.xaux.size() + 4);
    x.pos = r.pos; x.end = r.end; x.tlen = r.tlen; x.mpos = r.mpos; x.mtid = r.mtid;
    x.task = UZ_WALK_TASK_JOIN | jtask; x.flag = r.flag; x.l_seq = r.l_seq; x.n_cigar = r.n_cigar; x.mapq = r.mapq; x.l_name = r.l_name;
    x.direct = direct ? 1 : 0;
    P.xdesc.push_back(x);
    P.xaux.insert(P.xaux.end(), raw.begin() + r.pay_at, raw.begin() + r.pay_at + r.cigar_at); // (keep_raw: pay_at = its block_size field, cigar_at = 4 + block_size)
}
} // namespace

/* d_flags [n walk tasks]: what the device's walk said of every walk task (uz_bam_walk_flags).  h_flags [n tasks of the stage] out: the tasks whose
 * device descriptors are void -- the host walks those itself, here; totals: [0] descriptors, [1] aux bytes the stage holds for the device now */
int uz_stage_walk_flagged(uz_stage *P, const int32_t *d_flags, int32_t *h_flags, int64_t totals[2]) {
    if (!P || !P->begun || P->finished || !h_flags || !totals) { last_error = "uz_stage_walk_flagged: a plan that was begun and not yet finished"; return UZ_IO_E_ARG; }
    return guarded([&] {
        build_subtasks(*P);
        P->desc = true;
        P->joined = true;
        for (Task &T : P->tasks) T.by_hash = true;
        const size_t nt = P->tasks.size();
        for (size_t i = 0; i < nt; i++) h_flags[i] = (!P->sub_preflag.empty() && P->sub_preflag[i]) ? UZ_WALK_TASK_INCOMPLETE : 0;
        for (size_t u = 0; u < P->subs.size(); u++) h_flags[(size_t)P->subs[u].host] |= d_flags ? d_flags[u] : 0;
        Scratch W;
        for (size_t i = 0; i < nt; i++) {
            if (!h_flags[i]) continue;
            Task &T = P->tasks[i];
            walk_task(*P, T, W, i, false);
            // the mate candidates, as finish_task keeps them: the fetched records and those that share a name hash with one
            std::vector<uint64_t> &dn = W.dn;
            dn.clear();
            for (const WRec &r : W.all) if (r.keep == 2) dn.push_back(r.nhash);
            std::sort(dn.begin(), dn.end());
            dn.erase(std::unique(dn.begin(), dn.end()), dn.end());
            for (const WRec &r : W.all)
                if (r.keep == 2 || std::binary_search(dn.begin(), dn.end(), r.nhash)) x_append(*P, T, r, W.tmp.raw, (uint32_t)i, r.keep == 2);
            P->io_stats[0] += T.file_bytes; P->io_stats[1] += T.n_blocks; P->io_stats[2] += T.n_walked; P->io_stats[6] += T.n_pre;
        }
        totals[0] = (int64_t)P->xdesc.size(); totals[1] = (int64_t)P->xaux.size();
    });
}

/* Mates nobody walked: need [n] -- the device's list of members whose mate position lies in no reach interval, or whose name the covering task does
 * not hold.  Every distinct (reference, position) becomes a look-up task (the records overlapping it, through the index; kept: those with an asking
 * name); jtask [n] out: the join task that answers need[k].  totals as above. */
int uz_stage_lookup(uz_stage *P, int64_t n, const uz_need_rec *need, int32_t *jtask, int64_t totals[2]) {
    if (!P || !P->joined || n < 0 || (n && (!need || !jtask)) || !totals) { last_error = "uz_stage_lookup: behind uz_stage_walk_flagged"; return UZ_IO_E_ARG; }
    return guarded([&] {
        const int32_t n_ref = (int32_t)P->src->contigs.size();
        std::vector<int64_t> ord((size_t)n);
        for (int64_t k = 0; k < n; k++) {
            if (need[k].mtid < 0 || need[k].mtid >= n_ref) fail(UZ_IO_E_ARG, "uz_stage_lookup: a mate reference outside the file's");
            ord[(size_t)k] = k;
        }
        std::sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) {
            const uz_need_rec &a = need[x], &b = need[y];
            return a.mtid < b.mtid || (a.mtid == b.mtid && (a.mpos < b.mpos || (a.mpos == b.mpos && x < y)));
        });
        std::vector<size_t> first; // look-ups of one position share a task
        for (size_t k = 0; k < ord.size(); k++)
            if (k == 0 || need[ord[k]].mtid != need[ord[k - 1]].mtid || need[ord[k]].mpos != need[ord[k - 1]].mpos) first.push_back(k);
        const size_t base = P->look_tid.size();
        std::vector<Task> xt(first.size());
        const int w = (int)std::min<int64_t>(std::max(1, P->threads), std::max<int64_t>(1, (int64_t)first.size()));
        std::vector<std::unique_ptr<Inflater>> infs((size_t)w);
        for (auto &p : infs) p.reset(new Inflater());
        parallel_dynamic((int64_t)first.size(), w, [&](int64_t g, int k) {
            const size_t k0 = first[(size_t)g], k1 = (size_t)g + 1 < first.size() ? first[(size_t)g + 1] : ord.size();
            const uz_need_rec &q = need[ord[k0]];
            lookup_walk(*P, xt[(size_t)g], q.mtid, q.mpos, *infs[(size_t)k], [&](uint64_t h, uint32_t h2, uint32_t l_name, const uint8_t *) {
                for (size_t u = k0; u < k1; u++) {
                    const uz_need_rec &x = need[ord[u]];
                    if (x.h1 == h && x.h2 == h2 && x.l_name == l_name) return true;
                }
                return false;
            });
        });
        for (size_t g = 0; g < first.size(); g++) {
            const size_t k1 = g + 1 < first.size() ? first[g + 1] : ord.size();
            const uint32_t jt = (uint32_t)(P->tasks.size() + base + g);
            for (size_t u = first[g]; u < k1; u++) jtask[ord[u]] = (int32_t)jt;
            P->look_tid.push_back(xt[g].tid);
            for (const WRec &r : xt[g].recs) x_append(*P, xt[g], r, xt[g].raw, jt, false);
            P->io_stats[0] += xt[g].file_bytes; P->io_stats[1] += xt[g].n_blocks; P->io_stats[2] += xt[g].n_walked;
        }
        P->io_stats[5] += n;
        totals[0] = (int64_t)P->xdesc.size(); totals[1] = (int64_t)P->xaux.size();
    });
}

/* the descriptors from d0 on and the aux bytes from a0 on, as the two calls above left them; look_tid [look-up tasks so far] or NULL */
int uz_stage_extra(const uz_stage *P, int64_t d0, int64_t a0, uz_walk_desc *desc, uint8_t *aux, int32_t *look_tid) {
    if (!P || d0 < 0 || a0 < 0 || d0 > (int64_t)P->xdesc.size() || a0 > (int64_t)P->xaux.size()) { last_error = "uz_stage_extra: offsets beyond what the stage holds"; return UZ_IO_E_ARG; }
    if (desc && d0 < (int64_t)P->xdesc.size()) memcpy(desc, P->xdesc.data() + d0, (P->xdesc.size() - (size_t)d0) * sizeof(uz_walk_desc));
    if (aux && a0 < (int64_t)P->xaux.size()) memcpy(aux, P->xaux.data() + a0, P->xaux.size() - (size_t)a0);
    if (look_tid && !P->look_tid.empty()) memcpy(look_tid, P->look_tid.data(), P->look_tid.size() * sizeof(int32_t));
    return 0;
}
int64_t uz_stage_n_lookup_tasks(const uz_stage *P) { return P ? (int64_t)P->look_tid.size() : 0; }

/* parity aid (tests/test_stage_desc.py): the kept records of ANY finished plan, in output order -- virtual offset, name id, mate, 1 when the
 * record travels with bases */
int uz_stage_kept_debug(const uz_stage *P, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases) {
    if (!P || !P->finished) { last_error = "uz_stage_kept_debug: a finished plan"; return UZ_IO_E_ARG; }
    return guarded([&] {
        for (int64_t k = 0; k < P->n; k++) {
            const WRec &x = rec_of(*P, P->order[(size_t)k]);
            if (voff) voff[k] = x.voff;
            if (qname) qname[k] = x.qid;
            if (mate) { const int64_t m = final_ref(*P, x.mate_ref); mate[k] = m < 0 ? -1 : (int32_t)rec_of(*P, m).gidx; }
            if (bases) bases[k] = (x.keep == 2 || P->opt.all_bases) ? 1 : 0;
        }
    });
}

/* descriptor route: the record (index in the kept list) that brought name id `ids[k]` first -- its name bytes are the id's (uz_reads_from_bam hands the
 * names of the kept records back in record order); ids == NULL: every id, 0 .. n - 1 */
int uz_stage_name_records(const uz_stage *P, const uint32_t *ids, int64_t n, int64_t *rec) {
    if (!P || !P->finished || !rec || n < 0) { last_error = "uz_stage_name_records: a finished plan"; return UZ_IO_E_ARG; }
    for (int64_t k = 0; k < n; k++) {
        const uint64_t id = ids ? ids[k] : (uint64_t)k;
        if (id >= P->name_of_id.size()) { last_error = "uz_stage_name_records: name id out of range"; return UZ_IO_E_ARG; }
        rec[k] = (int64_t)rec_of(*P, P->name_of_id[id]).gidx;
    }
    return 0;
}

void uz_stage_free(uz_stage *P) { delete P; }

} // extern "C"
