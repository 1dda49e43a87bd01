// io_index.hpp -- BGZF random access and the binning indexes (BAI for BAM, TBI for tabix-indexed text such as .vcf.gz):
// the per-reference layout (bins with chunks of virtual file offsets + a 16 kb linear index) is the same in both files.
// CSI (the index `bcftools index` writes next to a BCF, and `tabix -C` next to a text file): the same bins over a
// scheme of its own width and depth, a left-most offset per bin instead of the linear index.
// Shared by the region decoders of io_bam.cpp (uz_bam_decode_regions) and io_vcf.cpp (uz_vcf_decode_regions).
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "io_common.hpp"

namespace uzio {

struct Chunk { uint64_t beg, end; }; // virtual file offsets (coffset << 16 | uoffset)

struct BaiRef {
    std::vector<std::pair<uint32_t, std::vector<Chunk>>> bins; // sorted by bin number
    std::vector<uint64_t> linear;                              // 16 kb windows
};

// the per-reference part of a BAI / TBI: n_ref x { bins { chunks }, linear index }, starting at d[off]
inline std::vector<BaiRef> parse_index_refs(const uint8_t *d, size_t N, size_t off, int32_t n_ref, const char *path) {
    auto need = [&](size_t k) { if (off + k > N) fail(UZ_IO_E_FORMAT, "truncated index %s", path); };
    // counts are signed in the file: a negative one (a corrupt index) must not wrap into a huge size or move `off` backwards
    auto count = [&](int32_t v, size_t each, const char *what) {
        if (v < 0 || (size_t)v > (N - std::min(off, N)) / each) fail(UZ_IO_E_FORMAT, "corrupt index %s: bad %s count %d", path, what, v);
        return (size_t)v;
    };
    if (n_ref < 0 || (size_t)n_ref > N) fail(UZ_IO_E_FORMAT, "corrupt index %s: bad reference count %d", path, n_ref);
    std::vector<BaiRef> refs((size_t)n_ref);
    for (int32_t r = 0; r < n_ref; r++) {
        need(4);
        const int32_t n_bin = (int32_t)count(rdi32(d + off), 8, "bin"); off += 4;
        for (int32_t b = 0; b < n_bin; b++) {
            need(8);
            const uint32_t bin = rd32(d + off);
            off += 8;
            const int32_t n_chunk = (int32_t)count(rdi32(d + off - 4), 16, "chunk");
            need((size_t)n_chunk * 16);
            std::vector<Chunk> cs;
            if (bin != 37450) { // the pseudo-bin holds counts, not chunks
                for (int32_t k = 0; k < n_chunk; k++) {
                    Chunk c;
                    memcpy(&c.beg, d + off + 16 * (size_t)k, 8);
                    memcpy(&c.end, d + off + 16 * (size_t)k + 8, 8);
                    cs.push_back(c);
                }
                refs[(size_t)r].bins.emplace_back(bin, std::move(cs));
            }
            off += (size_t)n_chunk * 16;
        }
        std::sort(refs[(size_t)r].bins.begin(), refs[(size_t)r].bins.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        need(4);
        off += 4;
        const int32_t n_intv = (int32_t)count(rdi32(d + off - 4), 8, "interval");
        need((size_t)n_intv * 8);
        refs[(size_t)r].linear.resize((size_t)n_intv);
        if (n_intv) memcpy(refs[(size_t)r].linear.data(), d + off, (size_t)n_intv * 8);
        off += (size_t)n_intv * 8;
    }
    return refs;
}

inline std::vector<BaiRef> read_bai(const char *path) {
    Bytes f = read_file(path);
    const uint8_t *d = f.data();
    const size_t N = f.size();
    if (N < 8 || memcmp(d, "BAI\1", 4) != 0) fail(UZ_IO_E_FORMAT, "%s is not a BAI index", path);
    return parse_index_refs(d, N, 8, rdi32(d + 4), path);
}

// bins a region [beg, end) may have records in (SAM spec, reg2bins)
inline void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &out) {
    out.clear();
    if (beg < 0) beg = 0;
    if (end <= beg) end = beg + 1;
    --end;
    out.push_back(0);
    for (uint32_t k = 1 + (uint32_t)(beg >> 26); k <= 1 + (uint32_t)(end >> 26); ++k) out.push_back(k);
    for (uint32_t k = 9 + (uint32_t)(beg >> 23); k <= 9 + (uint32_t)(end >> 23); ++k) out.push_back(k);
    for (uint32_t k = 73 + (uint32_t)(beg >> 20); k <= 73 + (uint32_t)(end >> 20); ++k) out.push_back(k);
    for (uint32_t k = 585 + (uint32_t)(beg >> 17); k <= 585 + (uint32_t)(end >> 17); ++k) out.push_back(k);
    for (uint32_t k = 4681 + (uint32_t)(beg >> 14); k <= 4681 + (uint32_t)(end >> 14); ++k) out.push_back(k);
}

struct Iv { int32_t lo, hi; };

// file access for the region reader
struct FileRd {
    int fd = -1;
    int64_t size = 0;
    explicit FileRd(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) fail(UZ_IO_E_OPEN, "cannot open %s", path);
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); fail(UZ_IO_E_OPEN, "cannot stat %s", path); }
        size = (int64_t)st.st_size;
    }
    ~FileRd() { if (fd >= 0) close(fd); }
    // bytes [off, off + len) clipped to the file; returns the count read
    size_t read_at(int64_t off, uint8_t *dst, size_t len) const {
        size_t got = 0;
        while (got < len && off + (int64_t)got < size) {
            const ssize_t k = pread(fd, dst + got, len - got, (off_t)(off + (int64_t)got));
            if (k <= 0) break;
            got += (size_t)k;
        }
        return got;
    }
};

// Inflates consecutive BGZF blocks starting at compressed offset `coff` until the block holding the virtual offset
// `vend` has been inflated AND the bytes needed by `more()` are there.  out: the inflated bytes of those blocks;
// block_at: (compressed offset, offset in out) per block.
struct Inflated {
    std::vector<uint8_t> bytes;
    std::vector<std::pair<int64_t, size_t>> block_at;
    int64_t next_coff = 0;
};

inline bool inflate_one(const FileRd &f, int64_t coff, Inflated &o, z_stream &z, std::vector<uint8_t> &cbuf, int64_t *file_bytes, int64_t *blocks) {
    uint8_t h[18];
    if (f.read_at(coff, h, 18) != 18) return false; // end of file
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) fail(UZ_IO_E_FORMAT, "not a BGZF block at byte %lld", (long long)coff);
    const size_t xlen = rd16(h + 10);
    cbuf.resize(12 + xlen);
    if (f.read_at(coff, cbuf.data(), 12 + xlen) != 12 + xlen) fail(UZ_IO_E_FORMAT, "truncated BGZF block");
    size_t q = 12, bsize = 0;
    bool found = false;
    while (q + 4 <= 12 + xlen) {
        const size_t slen = rd16(cbuf.data() + q + 2);
        if (cbuf[q] == 'B' && cbuf[q + 1] == 'C' && slen == 2) { bsize = rd16(cbuf.data() + q + 4); found = true; }
        q += 4 + slen;
    }
    if (!found) fail(UZ_IO_E_FORMAT, "BGZF block without a BC field at byte %lld", (long long)coff);
    const size_t blen = bsize + 1;
    cbuf.resize(blen);
    if (f.read_at(coff, cbuf.data(), blen) != blen) fail(UZ_IO_E_FORMAT, "truncated BGZF block at byte %lld", (long long)coff);
    const uint32_t crc = rd32(cbuf.data() + blen - 8), isize = rd32(cbuf.data() + blen - 4);
    const size_t at = o.bytes.size();
    o.block_at.emplace_back(coff, at);
    o.bytes.resize(at + isize);
    if (isize) {
        (void)z; // (the caller's zlib stream is not used any more: the thread's own inflater -- libdeflate when the system has it)
        static thread_local Inflater tl_inf;
        tl_inf.block(cbuf.data() + 12 + xlen, blen - 12 - xlen - 8, o.bytes.data() + at, isize, crc, coff);
    }
    o.next_coff = coff + (int64_t)blen;
    if (file_bytes) *file_bytes += (int64_t)blen;
    if (blocks) *blocks += 1;
    return true;
}

// chunks of the file that can hold records overlapping the intervals of one contig (sorted, merged)
inline void chunks_for(const BaiRef &ref, const std::vector<Iv> &ivs, std::vector<Chunk> &out) {
    std::vector<uint32_t> bins;
    std::vector<Chunk> cs;
    for (const Iv &iv : ivs) {
        reg2bins(iv.lo, iv.hi, bins);
        uint64_t min_off = 0;
        const size_t w = (size_t)(std::max<int64_t>(iv.lo, 0) >> 14);
        if (!ref.linear.empty()) min_off = ref.linear[std::min(w, ref.linear.size() - 1)];
        if (w >= ref.linear.size() && !ref.linear.empty()) min_off = ref.linear.back();
        for (uint32_t b : bins) {
            auto it = std::lower_bound(ref.bins.begin(), ref.bins.end(), b, [](const auto &a, uint32_t key) { return a.first < key; });
            if (it == ref.bins.end() || it->first != b) continue;
            for (const Chunk &c : it->second)
                if (c.end > min_off) cs.push_back(Chunk{std::max(c.beg, min_off), c.end});
        }
    }
    std::sort(cs.begin(), cs.end(), [](const Chunk &a, const Chunk &b) { return a.beg < b.beg || (a.beg == b.beg && a.end < b.end); });
    for (const Chunk &c : cs) {
        // merge chunks that touch or lie in the same compressed block neighbourhood: no record is walked twice
        if (!out.empty() && c.beg <= out.back().end) out.back().end = std::max(out.back().end, c.end);
        else out.push_back(c);
    }
}


// ---- CSI (SAM/CSI spec: CSIv1): magic, min_shift, depth, l_aux, aux, n_ref x { n_bin x { bin, loffset, n_chunk x { beg, end } } }
struct CsiBin { uint32_t bin; uint64_t loff; std::vector<Chunk> chunks; };
struct CsiRef { std::vector<CsiBin> bins; }; // sorted by bin number
struct Csi {
    int32_t min_shift = 14, depth = 5;
    std::vector<uint8_t> aux; // a text file's tabix header (format, columns, names); empty for a BCF
    std::vector<CsiRef> refs;
};

inline Csi parse_csi(const uint8_t *d, size_t N, const char *path) {
    if (N < 16 || memcmp(d, "CSI\1", 4) != 0) fail(UZ_IO_E_FORMAT, "%s is not a CSI index", path);
    Csi x;
    x.min_shift = rdi32(d + 4); x.depth = rdi32(d + 8);
    const int32_t l_aux = rdi32(d + 12);
    if (x.min_shift < 1 || x.min_shift > 30 || x.depth < 1 || x.depth > 9 || x.min_shift + 3 * x.depth > 40) fail(UZ_IO_E_FORMAT, "corrupt index %s: binning %d / %d", path, x.min_shift, x.depth);
    if (l_aux < 0 || 16 + (size_t)l_aux + 4 > N) fail(UZ_IO_E_FORMAT, "truncated index %s", path);
    x.aux.assign(d + 16, d + 16 + l_aux);
    size_t off = 16 + (size_t)l_aux;
    auto need = [&](size_t k) { if (off + k > N) fail(UZ_IO_E_FORMAT, "truncated index %s", path); };
    const int32_t n_ref = rdi32(d + off); off += 4;
    if (n_ref < 0 || (size_t)n_ref > N) fail(UZ_IO_E_FORMAT, "corrupt index %s: bad reference count %d", path, n_ref);
    const uint32_t pseudo = (uint32_t)((((uint64_t)1 << (3 * (x.depth + 1))) - 1) / 7 + 1); // the bin that holds counts, not chunks
    x.refs.resize((size_t)n_ref);
    for (int32_t r = 0; r < n_ref; r++) {
        need(4);
        const int32_t n_bin = rdi32(d + off); off += 4;
        if (n_bin < 0 || (size_t)n_bin > (N - off) / 16) fail(UZ_IO_E_FORMAT, "corrupt index %s: bad bin count %d", path, n_bin);
        for (int32_t b = 0; b < n_bin; b++) {
            need(16);
            CsiBin cb;
            cb.bin = rd32(d + off);
            memcpy(&cb.loff, d + off + 4, 8);
            const int32_t n_chunk = rdi32(d + off + 12);
            off += 16;
            if (n_chunk < 0 || (size_t)n_chunk > (N - off) / 16) fail(UZ_IO_E_FORMAT, "corrupt index %s: bad chunk count %d", path, n_chunk);
            if (cb.bin != pseudo)
                for (int32_t k = 0; k < n_chunk; k++) {
                    Chunk c;
                    memcpy(&c.beg, d + off + 16 * (size_t)k, 8);
                    memcpy(&c.end, d + off + 16 * (size_t)k + 8, 8);
                    cb.chunks.push_back(c);
                }
            off += (size_t)n_chunk * 16;
            if (cb.bin != pseudo) x.refs[(size_t)r].bins.push_back(std::move(cb));
        }
        std::sort(x.refs[(size_t)r].bins.begin(), x.refs[(size_t)r].bins.end(), [](const CsiBin &a, const CsiBin &b) { return a.bin < b.bin; });
    }
    return x;
}

// chunks of the file that can hold records overlapping the intervals of one reference (sorted, merged): the bins of every level that
// overlap an interval (the scheme's reg2bins), cut at the left-most offset of the lowest bin that holds the interval's start -- or, where
// the index has no such bin, of the nearest bin before it on its level, else of its parent, and so on up (the spec's rule for loffset)
inline void chunks_for_csi(const Csi &x, const CsiRef &ref, const std::vector<Iv> &ivs, std::vector<Chunk> &out) {
    std::vector<Chunk> cs;
    const int64_t maxpos = (int64_t)1 << (x.min_shift + 3 * x.depth);
    auto find = [&](uint32_t b) -> const CsiBin * {
        auto it = std::lower_bound(ref.bins.begin(), ref.bins.end(), b, [](const CsiBin &a, uint32_t key) { return a.bin < key; });
        return it != ref.bins.end() && it->bin == b ? &*it : nullptr;
    };
    for (const Iv &iv : ivs) {
        int64_t beg = std::max<int64_t>(iv.lo, 0), end = std::min<int64_t>(iv.hi, maxpos);
        if (beg >= maxpos) continue;
        if (end <= beg) end = beg + 1;
        uint64_t min_off = 0;
        {
            uint32_t bin = (uint32_t)((((uint64_t)1 << (3 * x.depth)) - 1) / 7 + (uint64_t)(beg >> x.min_shift)); // the lowest level's bin of `beg`
            const CsiBin *hit = nullptr;
            while (bin) {
                if ((hit = find(bin))) break;
                const uint32_t parent = (bin - 1) >> 3, first = (parent << 3) + 1;
                bin = bin > first ? bin - 1 : parent;
            }
            if (!bin) hit = find(0);
            min_off = hit ? hit->loff : 0;
        }
        int s = x.min_shift + 3 * x.depth;
        uint64_t t = 0;
        for (int l = 0; l <= x.depth; l++, s -= 3) {
            const uint64_t b0 = t + (uint64_t)(beg >> s), b1 = t + (uint64_t)((end - 1) >> s);
            auto it = std::lower_bound(ref.bins.begin(), ref.bins.end(), (uint32_t)b0, [](const CsiBin &a, uint32_t key) { return a.bin < key; });
            for (; it != ref.bins.end() && it->bin <= b1; ++it)
                for (const Chunk &c : it->chunks)
                    if (c.end > min_off) cs.push_back(Chunk{std::max(c.beg, min_off), c.end});
            t += (uint64_t)1 << (3 * l);
        }
    }
    std::sort(cs.begin(), cs.end(), [](const Chunk &a, const Chunk &b) { return a.beg < b.beg || (a.beg == b.beg && a.end < b.end); });
    for (const Chunk &c : cs) {
        if (!out.empty() && c.beg <= out.back().end) out.back().end = std::max(out.back().end, c.end);
        else out.push_back(c);
    }
}

} // namespace uzio
