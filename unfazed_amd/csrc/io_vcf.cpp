// io_vcf.cpp -- native sites-VCF decoder: text VCF (plain, gzip or bgzip) -> site columns for all
// samples.  Host only; built into libunfazed_io.so.
//
// Stands in for cyvcf2 as the reference uses it in unfazed/informative_site_finder.py:213-339 and
// :571-600 (Variant.start / end / REF / ALT, gt_types with gts012=False, gt_ref_depths,
// gt_alt_depths, gt_quals).  Field semantics are those of unfazed_amd/io_vcf.py (the readable
// statement; tests/test_io_native.py holds the two against each other): genotype codes HOM_REF 0,
// HET 1, UNKNOWN 2, HOM_ALT 3; half-missing calls count with their called allele, haploid calls as
// homozygous; depths from FORMAT/AD (first ALT) falling back to RO / AO, missing -> -1; GQ as a
// float, missing -> -1; end = INFO/END when it parses as an integer, else start + len(REF).
// BCF2 (the binary form, `.bcf`) is decoded into the same columns (decode_bcf below); it has no text lines,
// so uz_vcf_line() is empty for such a file.
#include <map>
#include <memory>

#include "io_common.hpp"
#include "io_index.hpp"

#include <cmath>
#include <cstdlib>

using namespace uzio;

struct uz_vcf {
    Bytes text;
    std::vector<std::string> samples, contigs;
    std::string header; // header lines joined by '\n'
    int64_t n = 0;
    std::vector<int64_t> contig_off;
    std::vector<int32_t> pos, end;
    std::vector<uint8_t> sflags, ref_base, alt_base;
    std::vector<uint8_t> gt;
    std::vector<int32_t> ref_depth, alt_depth;
    std::vector<double> gq;
    std::vector<uint64_t> line_at;
    std::vector<uint32_t> line_len, ref_at, ref_len, alt_at, alt_len; // REF / ALT relative to the line start
    std::vector<uint32_t> chrom_len;
    bool is_bcf = false;
    std::string pool;                 // BCF: comma-joined ALT strings (REF is used in place)
    std::vector<uint64_t> alt_pool_at; // BCF: offset of record i's ALT text in `pool`
    std::vector<uint64_t> info_at;     // BCF: offset of the INFO section of record i in `text`
    std::vector<uint32_t> n_info;
    std::vector<std::string> dict;     // BCF: FILTER / INFO / FORMAT string dictionary
    int64_t io_stats[4] = {0, 0, 0, 0}; // compressed bytes read, BGZF blocks inflated (region decode), lines walked, records kept
};

namespace {

struct Str {
    const char *p;
    size_t n;
    bool eq(const char *s) const { return strlen(s) == n && memcmp(p, s, n) == 0; }
};

// Python int(x) for the plain forms a VCF holds: optional sign, decimal digits
bool parse_int(Str s, long long &out) {
    if (s.n == 0 || s.n > 18) return false;
    size_t i = 0;
    bool neg = false;
    if (s.p[0] == '-' || s.p[0] == '+') { neg = s.p[0] == '-'; i = 1; }
    if (i >= s.n) return false;
    long long v = 0;
    for (; i < s.n; i++) {
        if (s.p[i] < '0' || s.p[i] > '9') return false;
        v = v * 10 + (s.p[i] - '0');
    }
    out = neg ? -v : v;
    return true;
}

long long num_int(Str s) { // io_vcf._num(x, int): "." / "" / unparsable -> -1
    long long v;
    if (s.n == 0 || s.eq(".")) return -1;
    return parse_int(s, v) ? v : -1;
}

double num_float(Str s) { // io_vcf._num(x, float, -1.0)
    if (s.n == 0 || s.eq(".") || s.n > 63) return -1.0;
    char buf[64];
    memcpy(buf, s.p, s.n);
    buf[s.n] = 0;
    char *e = nullptr;
    const double v = strtod(buf, &e);
    if (e == buf || *e != 0) return -1.0;
    return v;
}

// k-th ':'-separated piece of a sample column (or of FORMAT)
bool piece(Str col, int k, Str &out) {
    const char *p = col.p, *end = col.p + col.n;
    for (int i = 0;; i++) {
        const char *q = (const char *)memchr(p, ':', (size_t)(end - p));
        const char *stop = q ? q : end;
        if (i == k) { out = Str{p, (size_t)(stop - p)}; return true; }
        if (!q) return false;
        p = q + 1;
    }
}

int parse_gt(Str g) {
    if (g.eq(".") || g.eq("./.") || g.eq(".|.")) return UZ_GT_UNKNOWN;
    long long al[2] = {-1, -1};
    int na = 0;
    const char *p = g.p, *end = g.p + g.n;
    while (p <= end) {
        const char *q = p;
        while (q < end && *q != '/' && *q != '|') q++;
        if (na < 2) {
            const Str a{p, (size_t)(q - p)};
            long long v = -1;
            if (!a.eq(".")) { if (!parse_int(a, v)) fail(UZ_IO_E_FORMAT, "unparsable genotype allele '%.*s'", (int)a.n, a.p); }
            al[na] = v;
        }
        na++;
        if (q >= end) break;
        p = q + 1;
    }
    if (na == 1) return al[0] < 0 ? UZ_GT_UNKNOWN : (al[0] == 0 ? 0 : 3);
    const long long a = al[0], b = al[1];
    if (a < 0 && b < 0) return UZ_GT_UNKNOWN;
    if (a < 0 || b < 0) { const long long c = b < 0 ? a : b; return c == 0 ? 0 : 1; }
    if (a != b) return 1;
    return a == 0 ? 0 : 3;
}


// ------------------------------------------------------------------ BCF2
struct BcfVal { // one typed value: `n` elements of `type` at `p`
    int type = 0; // 1 int8, 2 int16, 3 int32, 5 float, 7 char, 0 missing / flag
    uint32_t n = 0;
    const uint8_t *p = nullptr;
};
const int BCF_SZ[8] = {0, 1, 2, 4, 0, 4, 0, 1};

const uint8_t *bcf_typed(const uint8_t *p, const uint8_t *end, BcfVal &v) {
    if (p >= end) fail(UZ_IO_E_FORMAT, "truncated BCF record");
    const uint8_t b = *p++;
    v.type = b & 15;
    v.n = b >> 4;
    if (v.n == 15) { // the length follows as a typed integer
        if (p >= end) fail(UZ_IO_E_FORMAT, "truncated BCF record");
        const uint8_t b2 = *p++;
        const int t2 = b2 & 15;
        if (t2 == 1) { v.n = *p; p += 1; } else if (t2 == 2) { v.n = rd16(p); p += 2; } else if (t2 == 3) { v.n = rd32(p); p += 4; }
        else fail(UZ_IO_E_FORMAT, "bad BCF length descriptor");
    }
    if (v.type > 7 || (v.type != 0 && BCF_SZ[v.type] == 0)) fail(UZ_IO_E_FORMAT, "bad BCF value type %d", v.type);
    v.p = p;
    p += (size_t)v.n * BCF_SZ[v.type];
    if (p > end) fail(UZ_IO_E_FORMAT, "BCF value overruns its record");
    return p;
}
// k-th element of an integer vector: value, or missing / end-of-vector
enum { BCF_OK = 0, BCF_MISSING = 1, BCF_EOV = 2 };
int bcf_int(int type, const uint8_t *p, uint32_t k, long long &out) {
    if (type == 1) { const int8_t x = (int8_t)p[k]; if (x == INT8_MIN) return BCF_MISSING; if (x == INT8_MIN + 1) return BCF_EOV; out = x; return BCF_OK; }
    if (type == 2) { int16_t x; memcpy(&x, p + 2 * k, 2); if (x == INT16_MIN) return BCF_MISSING; if (x == INT16_MIN + 1) return BCF_EOV; out = x; return BCF_OK; }
    if (type == 3) { int32_t x; memcpy(&x, p + 4 * k, 4); if (x == INT32_MIN) return BCF_MISSING; if (x == INT32_MIN + 1) return BCF_EOV; out = x; return BCF_OK; }
    return BCF_MISSING;
}
long long bcf_scalar_int(const BcfVal &v) {
    long long x = 0;
    if (v.n == 0 || bcf_int(v.type, v.p, 0, x) != BCF_OK) fail(UZ_IO_E_FORMAT, "bad BCF dictionary key");
    return x;
}

void decode_bcf(uz_vcf &V, int threads) {
    V.is_bcf = true;
    const uint8_t *D = V.text.data();
    const size_t N = V.text.size();
    if (N < 9 || D[4] != 2) fail(UZ_IO_E_FORMAT, "unsupported BCF version");
    const size_t l_text = rd32(D + 5);
    if (9 + l_text > N) fail(UZ_IO_E_FORMAT, "truncated BCF header");
    // header text: dictionaries (IDX= when present, else order of appearance; PASS is 0), contigs, samples
    std::vector<std::string> &dict = V.dict;
    auto dict_set = [&](std::vector<std::string> &d, long idx, const std::string &name) {
        if (idx < 0) { for (auto &x : d) if (x == name) return; idx = (long)d.size(); }
        if ((size_t)idx >= d.size()) d.resize((size_t)idx + 1);
        d[(size_t)idx] = name;
    };
    dict_set(dict, 0, "PASS");
    const char *H = (const char *)D + 9;
    size_t p = 0;
    while (p < l_text) {
        const char *nl = (const char *)memchr(H + p, '\n', l_text - p);
        size_t e = nl ? (size_t)(nl - H) : l_text;
        while (e > p && (H[e - 1] == 0 || H[e - 1] == '\r')) e--;
        if (e > p) {
            const std::string line(H + p, e - p);
            if (!V.header.empty()) V.header.push_back('\n');
            V.header += line;
            auto field = [&](const char *key) -> std::string {
                const std::string k = std::string(key) + "=";
                size_t a = line.find("<" + k);
                if (a == std::string::npos) a = line.find("," + k);
                if (a == std::string::npos) return "";
                a += 1 + k.size();
                size_t b = a;
                while (b < line.size() && line[b] != ',' && line[b] != '>') b++;
                return line.substr(a, b - a);
            };
            const bool is_dict = line.rfind("##INFO=<", 0) == 0 || line.rfind("##FORMAT=<", 0) == 0 || line.rfind("##FILTER=<", 0) == 0;
            if (is_dict || line.rfind("##contig=<", 0) == 0) {
                const std::string id = field("ID"), ix = field("IDX");
                long idx = -1;
                if (!ix.empty()) idx = strtol(ix.c_str(), nullptr, 10);
                if (!id.empty()) dict_set(is_dict ? dict : V.contigs, idx, id);
            } else if (line[0] == '#' && line.size() > 1 && line[1] != '#') {
                size_t a = 0;
                int col = 0;
                while (a <= line.size()) {
                    size_t b = line.find('\t', a);
                    if (b == std::string::npos) b = line.size();
                    if (col >= 9) V.samples.emplace_back(line.substr(a, b - a));
                    col++;
                    a = b + 1;
                }
            }
        }
        p = (nl ? (size_t)(nl - H) : l_text) + 1;
    }
    std::vector<std::string> header_contigs;
    header_contigs.swap(V.contigs); // the table's contigs are those that carry records, in order of appearance
    int k_gt = -1, k_ad = -1, k_ro = -1, k_ao = -1, k_gq = -1;
    for (size_t k = 0; k < dict.size(); k++) {
        if (dict[k] == "GT") k_gt = (int)k; else if (dict[k] == "AD") k_ad = (int)k; else if (dict[k] == "RO") k_ro = (int)k;
        else if (dict[k] == "AO") k_ao = (int)k; else if (dict[k] == "GQ") k_gq = (int)k;
    }
    // record boundaries
    std::vector<uint64_t> rec;
    size_t off = 9 + l_text;
    while (off + 8 <= N) {
        const size_t ls = rd32(D + off), li = rd32(D + off + 4);
        if (ls < 24 || off + 8 + ls + li > N) fail(UZ_IO_E_FORMAT, "truncated BCF record at byte %zu", off);
        rec.push_back(off);
        off += 8 + ls + li;
    }
    const int64_t n = (int64_t)rec.size();
    V.n = n;
    const size_t un = (size_t)n, ns = V.samples.size();
    V.line_at.assign(un, 0); V.line_len.assign(un, 0);
    V.ref_at.resize(un); V.ref_len.resize(un); V.alt_at.assign(un, 0); V.alt_len.resize(un);
    V.chrom_len.assign(un, 0); V.alt_pool_at.resize(un); V.info_at.resize(un); V.n_info.resize(un);
    V.pos.resize(un); V.end.resize(un); V.sflags.resize(un); V.ref_base.resize(un); V.alt_base.resize(un);
    V.gt.assign(ns * un, UZ_GT_UNKNOWN);
    V.ref_depth.assign(ns * un, -1); V.alt_depth.assign(ns * un, -1);
    V.gq.assign(ns * un, -1.0);
    std::vector<int32_t> chrom(un);
    std::vector<std::string> alts(un);
    threads = workers_for(n, threads, 2048);
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; i++) {
            const uint8_t *r = D + rec[(size_t)i];
            const size_t ls = rd32(r), li = rd32(r + 4);
            const uint8_t *q = r + 8, *send = r + 8 + ls, *iend = send + li;
            chrom[(size_t)i] = rdi32(q);
            const int32_t pos = rdi32(q + 4), rlen = rdi32(q + 8);
            const uint32_t nai = rd32(q + 16), nfs = rd32(q + 20);
            const uint32_t n_allele = nai >> 16, n_inf = nai & 0xFFFFu, n_fmt = nfs >> 24, n_smp = nfs & 0xFFFFFFu;
            if (chrom[(size_t)i] < 0 || (size_t)chrom[(size_t)i] >= header_contigs.size()) fail(UZ_IO_E_FORMAT, "BCF contig id out of range");
            if (n_smp != ns && n_fmt) fail(UZ_IO_E_FORMAT, "BCF record with %u samples, header has %zu", n_smp, ns);
            q += 24;
            BcfVal v;
            q = bcf_typed(q, send, v); // ID
            std::string &alt = alts[(size_t)i];
            size_t ref_n = 0, n_alt = 0, alt0_n = 0;
            bool star = false;
            for (uint32_t a = 0; a < n_allele; a++) {
                q = bcf_typed(q, send, v);
                if (a == 0) { V.ref_at[(size_t)i] = (uint32_t)(v.p - r); V.ref_len[(size_t)i] = v.n; ref_n = v.n; V.line_at[(size_t)i] = rec[(size_t)i]; }
                else {
                    if (a > 1) alt.push_back(',');
                    alt.append((const char *)v.p, v.n);
                    if (a == 1) alt0_n = v.n;
                    if (v.n == 1 && v.p[0] == '*') star = true;
                    n_alt++;
                }
            }
            if (n_alt == 0) alt = ".";
            V.alt_len[(size_t)i] = (uint32_t)alt.size();
            q = bcf_typed(q, send, v); // FILTER
            V.info_at[(size_t)i] = (uint64_t)(q - D);
            V.n_info[(size_t)i] = n_inf;
            V.pos[(size_t)i] = pos;
            V.end[(size_t)i] = pos + rlen; // htslib keeps rlen = END - POS when INFO/END is present
            const bool cx = n_alt != 1 || ref_n > 1 || star || alt0_n > 1;
            V.sflags[(size_t)i] = cx ? UZ_SF_COMPLEX : 0;
            V.ref_base[(size_t)i] = cx || ref_n != 1 ? 0 : D[rec[(size_t)i] + V.ref_at[(size_t)i]];
            V.alt_base[(size_t)i] = cx || alt0_n != 1 ? 0 : (uint8_t)alt[0];
            // FORMAT fields
            const uint8_t *f = send;
            BcfVal fv[5]; // GT, AD, RO, AO, GQ
            for (uint32_t k = 0; k < n_fmt; k++) {
                BcfVal key;
                f = bcf_typed(f, iend, key);
                const long long kid = bcf_scalar_int(key);
                if (f >= iend) fail(UZ_IO_E_FORMAT, "truncated BCF FORMAT block");
                BcfVal d; // descriptor: type and per-sample length; the data of all samples follows
                const uint8_t b = *f++;
                d.type = b & 15; d.n = b >> 4;
                if (d.n == 15) { BcfVal L; f = bcf_typed(f, iend, L); d.n = (uint32_t)bcf_scalar_int(L); }
                if (d.type > 7 || (d.type != 0 && BCF_SZ[d.type] == 0)) fail(UZ_IO_E_FORMAT, "bad BCF FORMAT type");
                d.p = f;
                f += (size_t)d.n * BCF_SZ[d.type] * n_smp;
                if (f > iend) fail(UZ_IO_E_FORMAT, "BCF FORMAT data overruns its record");
                if (kid == k_gt) fv[0] = d; else if (kid == k_ad) fv[1] = d; else if (kid == k_ro) fv[2] = d;
                else if (kid == k_ao) fv[3] = d; else if (kid == k_gq) fv[4] = d;
            }
            for (size_t s = 0; s < ns && n_fmt; s++) {
                const size_t o = s * un + (size_t)i;
                auto at = [&](const BcfVal &d) { return d.p + s * (size_t)d.n * BCF_SZ[d.type]; };
                if (fv[0].p && fv[0].n) { // GT: (allele + 1) << 1 | phased; 0 = missing allele
                    long long al[2] = {-1, -1};
                    int na = 0;
                    for (uint32_t k = 0; k < fv[0].n; k++) {
                        long long x;
                        const int st = bcf_int(fv[0].type, at(fv[0]), k, x);
                        if (st == BCF_EOV) break;
                        if (na < 2) al[na] = st == BCF_OK ? (x >> 1) - 1 : -1;
                        na++;
                    }
                    int code = UZ_GT_UNKNOWN;
                    if (na == 1) code = al[0] < 0 ? UZ_GT_UNKNOWN : (al[0] == 0 ? 0 : 3);
                    else if (na >= 2) {
                        const long long a = al[0], b = al[1];
                        if (a < 0 && b < 0) code = UZ_GT_UNKNOWN;
                        else if (a < 0 || b < 0) code = (b < 0 ? a : b) == 0 ? 0 : 1;
                        else if (a != b) code = 1;
                        else code = a == 0 ? 0 : 3;
                    }
                    V.gt[o] = (uint8_t)code;
                }
                bool ad_done = false;
                if (fv[1].p && fv[1].n) { // AD: "." (all missing) falls through to RO / AO, as in the text form
                    long long x0 = -1, x1 = -1;
                    const int s0 = bcf_int(fv[1].type, at(fv[1]), 0, x0);
                    const int s1 = fv[1].n > 1 ? bcf_int(fv[1].type, at(fv[1]), 1, x1) : BCF_EOV;
                    if (!(s0 != BCF_OK && s1 == BCF_EOV)) {
                        V.ref_depth[o] = s0 == BCF_OK ? (int32_t)x0 : -1;
                        V.alt_depth[o] = s1 == BCF_OK ? (int32_t)x1 : -1;
                        ad_done = true;
                    }
                }
                if (!ad_done && fv[2].p && fv[3].p && fv[2].n && fv[3].n) {
                    long long x;
                    V.ref_depth[o] = bcf_int(fv[2].type, at(fv[2]), 0, x) == BCF_OK ? (int32_t)x : -1;
                    V.alt_depth[o] = bcf_int(fv[3].type, at(fv[3]), 0, x) == BCF_OK ? (int32_t)x : -1;
                }
                if (fv[4].p && fv[4].n) {
                    if (fv[4].type == 5) {
                        uint32_t bits;
                        memcpy(&bits, at(fv[4]), 4);
                        float fl;
                        memcpy(&fl, &bits, 4);
                        V.gq[o] = (bits == 0x7F800001u || bits == 0x7F800002u) ? -1.0 : (double)fl;
                    } else {
                        long long x;
                        V.gq[o] = bcf_int(fv[4].type, at(fv[4]), 0, x) == BCF_OK ? (double)x : -1.0;
                    }
                }
            }
        }
    });
    for (int64_t i = 0; i < n; i++) { V.alt_pool_at[(size_t)i] = V.pool.size(); V.pool += alts[(size_t)i]; V.pool.push_back('\0'); }
    // contigs in order of appearance; records must be grouped by contig and sorted inside one
    V.contig_off.clear();
    int32_t last = -1;
    std::vector<uint8_t> seen(header_contigs.size(), 0);
    for (int64_t i = 0; i < n; i++) {
        const int32_t c = chrom[(size_t)i];
        if (c != last) {
            if (seen[(size_t)c]) fail(UZ_IO_E_UNSORTED, "sites records are not grouped by contig: %s", header_contigs[(size_t)c].c_str());
            seen[(size_t)c] = 1;
            V.contigs.push_back(header_contigs[(size_t)c]);
            V.contig_off.push_back(i);
            last = c;
        } else if (V.pos[(size_t)i] < V.pos[(size_t)i - 1])
            fail(UZ_IO_E_UNSORTED, "sites records are not sorted by position on %s", header_contigs[(size_t)c].c_str());
    }
    V.contig_off.push_back(n);
}

void decode_text(uz_vcf &V, int threads);

void decode(uz_vcf &V, const char *path, int threads) {
    {
        Bytes file = read_file(path);
        bool gz = false;
        V.text = inflate_all(file, threads, &gz);
        V.io_stats[0] = (int64_t)file.size();
    }
    decode_text(V, threads);
    V.io_stats[2] = V.io_stats[3] = V.n;
}

// ---------------------------------------------------------------------------------------------------------
// Region decode through the tabix index (NAME.vcf.gz.tbi): what `vcf(region)` hands the reference per DNM
// (informative_site_finder.py:42, :399-420, :566) -- only the BGZF blocks the index names for the intervals are read and
// inflated; the table holds the header and, in file order, the records that overlap an interval.
struct Tbi {
    std::vector<std::string> names;
    std::vector<BaiRef> refs;
};

Tbi read_tbi_file(const char *path, int threads);

// The parsed index of a file is kept for the next call (a session decodes the windows of batch after batch from one sites file; the
// index of a 20 M-site VCF takes longer to inflate and parse than a batch's windows take to decode): keyed by path, size and mtime.
std::shared_ptr<const Tbi> read_tbi(const char *path, int threads) {
    static std::mutex mu;
    static std::map<std::string, std::pair<std::pair<int64_t, int64_t>, std::shared_ptr<const Tbi>>> cache;
    struct stat st;
    if (stat(path, &st) != 0) fail(UZ_IO_E_OPEN, "cannot open %s", path);
    const std::pair<int64_t, int64_t> stamp{(int64_t)st.st_size, (int64_t)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec};
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(path);
        if (it != cache.end() && it->second.first == stamp) return it->second.second;
    }
    auto t = std::make_shared<const Tbi>(read_tbi_file(path, threads));
    std::lock_guard<std::mutex> g(mu);
    if (cache.size() >= 8) cache.clear();
    cache[path] = {stamp, t};
    return t;
}

Tbi read_tbi_file(const char *path, int threads) {
    Bytes f = read_file(path);
    bool gz = false;
    Bytes raw = inflate_all(f, threads, &gz);
    const uint8_t *d = raw.data();
    const size_t N = raw.size();
    if (N < 36 || memcmp(d, "TBI\1", 4) != 0) fail(UZ_IO_E_FORMAT, "%s is not a tabix index", path);
    const int32_t n_ref = rdi32(d + 4), format = rdi32(d + 8), l_nm = rdi32(d + 32);
    if ((format & 0xFFFF) != 2) fail(UZ_IO_E_FORMAT, "%s indexes a file that is not VCF (format %d)", path, format);
    if (n_ref < 0 || l_nm < 0 || 36 + (size_t)l_nm > N) fail(UZ_IO_E_FORMAT, "truncated tabix index %s", path);
    Tbi t;
    size_t a = 36;
    const size_t stop = 36 + (size_t)l_nm;
    while (a < stop && (int32_t)t.names.size() < n_ref) {
        const void *z = memchr(d + a, 0, stop - a);
        const size_t e = z ? (size_t)((const uint8_t *)z - d) : stop;
        t.names.emplace_back((const char *)d + a, e - a);
        a = e + 1;
    }
    if ((int32_t)t.names.size() != n_ref) fail(UZ_IO_E_FORMAT, "tabix index %s: %d names for %d references", path, (int)t.names.size(), n_ref);
    t.refs = parse_index_refs(d, N, stop, n_ref, path);
    return t;
}

// the index next to a sites file: the one named, else <path>.tbi, else <path>.csi (a BCF's index, or a text file's by `tabix -C`)
std::string tbi_path_for(const char *path, const char *tbi_path) {
    if (tbi_path && *tbi_path) return tbi_path;
    const std::string t = std::string(path) + ".tbi", c = std::string(path) + ".csi";
    struct stat st;
    if (stat(t.c_str(), &st) != 0 && stat(c.c_str(), &st) == 0) return c;
    return t;
}

// TBI or CSI behind one face: the names of the references (a BCF's come from its header: bcf_header_contigs) and the chunks of an interval
struct RegionIndex {
    std::shared_ptr<const Tbi> tbi;
    std::shared_ptr<const Csi> csi;
    std::vector<std::string> names;
    size_t n_refs() const { return tbi ? tbi->refs.size() : csi->refs.size(); }
    void chunks(int32_t r, const Iv &iv, std::vector<Chunk> &out) const {
        if (tbi) chunks_for(tbi->refs[(size_t)r], std::vector<Iv>{iv}, out);
        else chunks_for_csi(*csi, csi->refs[(size_t)r], std::vector<Iv>{iv}, out);
    }
};

std::shared_ptr<const Csi> read_csi(const char *path, int threads) { // (kept for the next call like a TBI: read_tbi)
    static std::mutex mu;
    static std::map<std::string, std::pair<std::pair<int64_t, int64_t>, std::shared_ptr<const Csi>>> cache;
    struct stat st;
    if (stat(path, &st) != 0) fail(UZ_IO_E_OPEN, "cannot open %s", path);
    const std::pair<int64_t, int64_t> stamp{(int64_t)st.st_size, (int64_t)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec};
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(path);
        if (it != cache.end() && it->second.first == stamp) return it->second.second;
    }
    Bytes f = read_file(path);
    bool gz = false;
    Bytes raw = inflate_all(f, threads, &gz);
    auto x = std::make_shared<const Csi>(parse_csi(raw.data(), raw.size(), path));
    std::lock_guard<std::mutex> g(mu);
    if (cache.size() >= 8) cache.clear();
    cache[path] = {stamp, x};
    return x;
}

bool index_is_csi(const char *path) { // by its first BGZF block's first bytes
    FileRd f(path);
    Inflated inf;
    z_stream z;
    std::vector<uint8_t> cbuf;
    if (!inflate_one(f, 0, inf, z, cbuf, nullptr, nullptr) || inf.bytes.size() < 4) fail(UZ_IO_E_FORMAT, "%s is not a TBI or CSI index", path);
    return memcmp(inf.bytes.data(), "CSI\1", 4) == 0;
}

// the contigs of a BCF header in the order of their ids (IDX= when present, else order of appearance)
std::vector<std::string> bcf_header_contigs(const uint8_t *D, size_t N) {
    std::vector<std::string> out;
    if (N < 9 || memcmp(D, "BCF\2", 4) != 0) return out;
    const size_t l_text = rd32(D + 5);
    if (9 + l_text > N) fail(UZ_IO_E_FORMAT, "truncated BCF header");
    const char *H = (const char *)D + 9;
    size_t p = 0;
    while (p < l_text) {
        const char *nl = (const char *)memchr(H + p, '\n', l_text - p);
        const size_t e = nl ? (size_t)(nl - H) : l_text;
        if (e - p > 10 && memcmp(H + p, "##contig=<", 10) == 0) {
            const std::string line(H + p, e - p);
            auto field = [&](const char *key) -> std::string {
                const std::string k = std::string(key) + "=";
                size_t a = line.find("<" + k);
                if (a == std::string::npos) a = line.find("," + k);
                if (a == std::string::npos) return "";
                a += 1 + k.size();
                size_t b = a;
                while (b < line.size() && line[b] != ',' && line[b] != '>') b++;
                return line.substr(a, b - a);
            };
            const std::string id = field("ID"), ix = field("IDX");
            if (!id.empty()) {
                long idx = ix.empty() ? -1 : strtol(ix.c_str(), nullptr, 10);
                if (idx < 0) { bool seen = false; for (auto &x : out) seen |= x == id; if (seen) { p = e + 1; continue; } idx = (long)out.size(); }
                if ((size_t)idx >= out.size()) out.resize((size_t)idx + 1);
                out[(size_t)idx] = id;
            }
        }
        p = e + 1;
    }
    return out;
}

// the bytes of a BCF's header (magic, l_text, text) from the head of the file; empty when the file is not a BCF
std::vector<uint8_t> bcf_header_bytes(const FileRd &file, int64_t *file_bytes, int64_t *blocks, int64_t *next_coff) {
    Inflated inf;
    z_stream z;
    std::vector<uint8_t> cbuf;
    if (!inflate_one(file, 0, inf, z, cbuf, file_bytes, blocks)) return {};
    if (inf.bytes.size() < 9 || memcmp(inf.bytes.data(), "BCF\2", 4) != 0) return {};
    const size_t need = 9 + (size_t)rd32(inf.bytes.data() + 5);
    while (inf.bytes.size() < need)
        if (!inflate_one(file, inf.next_coff, inf, z, cbuf, file_bytes, blocks)) fail(UZ_IO_E_FORMAT, "truncated BCF header");
    if (next_coff) *next_coff = inf.next_coff;
    return std::vector<uint8_t>(inf.bytes.begin(), inf.bytes.begin() + (long)need);
}

RegionIndex load_region_index(const char *path, const char *idx_path, int threads) {
    const std::string ip = tbi_path_for(path, idx_path);
    RegionIndex ix;
    if (!index_is_csi(ip.c_str())) {
        ix.tbi = read_tbi(ip.c_str(), threads);
        ix.names = ix.tbi->names;
        return ix;
    }
    ix.csi = read_csi(ip.c_str(), threads);
    const std::vector<uint8_t> &aux = ix.csi->aux;
    if (aux.size() >= 28) { // a text file's tabix header: format, col_seq, col_beg, col_end, meta, skip, l_nm, names
        const int32_t format = rdi32(aux.data()), l_nm = rdi32(aux.data() + 24);
        if ((format & 0xFFFF) != 2) fail(UZ_IO_E_FORMAT, "%s indexes a file that is not VCF (format %d)", ip.c_str(), format);
        if (l_nm < 0 || 28 + (size_t)l_nm > aux.size()) fail(UZ_IO_E_FORMAT, "truncated index %s", ip.c_str());
        size_t a = 28;
        const size_t stop = 28 + (size_t)l_nm;
        while (a < stop) {
            const void *zz = memchr(aux.data() + a, 0, stop - a);
            const size_t e = zz ? (size_t)((const uint8_t *)zz - aux.data()) : stop;
            ix.names.emplace_back((const char *)aux.data() + a, e - a);
            a = e + 1;
        }
    } else { // a BCF: its references are the header's contigs, by id
        FileRd file(path);
        const std::vector<uint8_t> hb = bcf_header_bytes(file, nullptr, nullptr, nullptr);
        if (hb.empty()) fail(UZ_IO_E_FORMAT, "%s has no reference names and %s is not a BCF", ip.c_str(), path);
        ix.names = bcf_header_contigs(hb.data(), hb.size());
    }
    if (ix.names.size() < ix.csi->refs.size()) ix.names.resize(ix.csi->refs.size());
    return ix;
}

void decode_regions(uz_vcf &V, const char *path, const char *tbi_path, int64_t n_iv, const int32_t *iv_ref, const int32_t *iv_lo,
                    const int32_t *iv_hi, int threads) {
    const RegionIndex tbi = load_region_index(path, tbi_path, threads);
    const int32_t n_ref = (int32_t)tbi.n_refs();
    std::vector<std::vector<Iv>> ivs((size_t)n_ref);
    for (int64_t k = 0; k < n_iv; k++) {
        if (iv_ref[k] < 0 || iv_ref[k] >= n_ref) fail(UZ_IO_E_ARG, "interval %lld names reference %d of %d", (long long)k, iv_ref[k], n_ref);
        if (iv_hi[k] > iv_lo[k]) ivs[(size_t)iv_ref[k]].push_back(Iv{std::max(iv_lo[k], 0), iv_hi[k]});
    }
    FileRd file(path);
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) fail(UZ_IO_E_FORMAT, "zlib init failed");
    std::vector<uint8_t> cbuf;
    std::string text;
    int64_t file_bytes = 0, blocks = 0, walked = 0, kept = 0;
    int64_t bcf_body = 0; // a BCF: the compressed offset behind its header's last block (unused)
    int64_t hb_bytes = 0, hb_blocks = 0;
    const std::vector<uint8_t> bcf_head = bcf_header_bytes(file, &hb_bytes, &hb_blocks, &bcf_body);
    const bool is_bcf = !bcf_head.empty();
    if (is_bcf) { file_bytes += hb_bytes; blocks += hb_blocks; } // (a text file's header blocks are read, and counted, below)
    try {
        // header: blocks from the start of the file until a line that does not start with '#'
        if (is_bcf) text.assign((const char *)bcf_head.data(), bcf_head.size());
        else {
            Inflated inf;
            size_t at = 0;
            bool done = false;
            while (!done) {
                while (inf.bytes.size() <= at || !memchr(inf.bytes.data() + at, '\n', inf.bytes.size() - at)) { // (nothing inflated yet: no pointer to hand to memchr)
                    const int64_t next = inf.block_at.empty() ? 0 : inf.next_coff;
                    if (!inflate_one(file, next, inf, z, cbuf, &file_bytes, &blocks)) { done = true; break; }
                }
                if (done) break;
                const uint8_t *nl = (const uint8_t *)memchr(inf.bytes.data() + at, '\n', inf.bytes.size() - at);
                const size_t e = (size_t)(nl - inf.bytes.data());
                if (e > at && inf.bytes[at] != '#') break;
                text.append((const char *)inf.bytes.data() + at, e + 1 - at);
                at = e + 1;
            }
        }
        // work items: (reference, chunk) in file order; every item is walked by one worker into its own text
        struct Item { int32_t r; Chunk ck; };
        std::vector<Item> items;
        std::vector<std::vector<Iv>> merged_of((size_t)n_ref);
        for (int32_t r = 0; r < n_ref; r++) {
            std::vector<Iv> &v = ivs[(size_t)r];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end(), [](const Iv &a, const Iv &b) { return a.lo < b.lo || (a.lo == b.lo && a.hi < b.hi); });
            std::vector<Iv> &merged = merged_of[(size_t)r];
            for (const Iv &iv : v) {
                if (!merged.empty() && iv.lo <= merged.back().hi) merged.back().hi = std::max(merged.back().hi, iv.hi);
                else merged.push_back(iv);
            }
            // chunks per merged interval, then joined per BGZF block: a window costs its own blocks, not the rest of its 16 kb bin
            std::vector<Chunk> chunks;
            for (const Iv &iv : merged) {
                std::vector<Chunk> one;
                tbi.chunks(r, iv, one);
                chunks.insert(chunks.end(), one.begin(), one.end());
            }
            std::sort(chunks.begin(), chunks.end(), [](const Chunk &a, const Chunk &b) { return a.beg < b.beg || (a.beg == b.beg && a.end < b.end); });
            std::vector<Chunk> joined;
            for (const Chunk &c : chunks) {
                if (!joined.empty() && (c.beg >> 16) <= (joined.back().end >> 16)) joined.back().end = std::max(joined.back().end, c.end);
                else joined.push_back(c);
            }
            for (const Chunk &c : joined) items.push_back(Item{r, c});
        }
        std::vector<std::string> texts(items.size());
        std::vector<int64_t> st_bytes(items.size(), 0), st_blocks(items.size(), 0), st_walked(items.size(), 0), st_kept(items.size(), 0);
        parallel_dynamic((int64_t)items.size(), threads, [&](int64_t it, int) {
            const int32_t r = items[(size_t)it].r;
            const Chunk ck = items[(size_t)it].ck;
            const std::vector<Iv> &merged = merged_of[(size_t)r];
            const std::string &name = tbi.names[(size_t)r];
            const int32_t last_hi = merged.back().hi;
            std::string &text = texts[(size_t)it];
            int64_t &file_bytes = st_bytes[(size_t)it], &blocks = st_blocks[(size_t)it], &walked = st_walked[(size_t)it], &kept = st_kept[(size_t)it];
            std::vector<uint8_t> cbuf;
            z_stream z; // (unused by inflate_one: kept for its signature)
            Inflated inf;
            if (!inflate_one(file, (int64_t)(ck.beg >> 16), inf, z, cbuf, &file_bytes, &blocks)) return;
            size_t at = (size_t)(ck.beg & 0xFFFF), blk = 0;
            if (is_bcf) {
                // records of the binary form: l_shared, l_indiv, then CHROM (the header's contig id: the index's reference), POS (0-based), rlen
                for (;;) {
                    bool eof = false;
                    while (at + 8 > inf.bytes.size() && !eof) eof = !inflate_one(file, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks);
                    if (at >= inf.bytes.size()) break;
                    if (at + 8 > inf.bytes.size()) fail(UZ_IO_E_FORMAT, "truncated BCF record");
                    while (blk + 1 < inf.block_at.size() && inf.block_at[blk + 1].second <= at) blk++;
                    if ((((uint64_t)inf.block_at[blk].first << 16) | (uint64_t)(at - inf.block_at[blk].second)) >= ck.end) break;
                    const size_t ls = rd32(inf.bytes.data() + at), li = rd32(inf.bytes.data() + at + 4);
                    if (ls < 24) fail(UZ_IO_E_FORMAT, "bad BCF record (l_shared %zu)", ls);
                    while (at + 8 + ls + li > inf.bytes.size())
                        if (!inflate_one(file, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks)) fail(UZ_IO_E_FORMAT, "truncated BCF record");
                    const uint8_t *q = inf.bytes.data() + at + 8;
                    const int32_t chrom = rdi32(q);
                    const long long pos0 = rdi32(q + 4), rlen = rdi32(q + 8);
                    walked++;
                    if (chrom == r) {
                        if (pos0 >= (long long)last_hi) break;
                        const long long end = pos0 + std::max<long long>(1, rlen);
                        auto iv = std::upper_bound(merged.begin(), merged.end(), pos0, [](long long key, const Iv &x) { return key < (long long)x.hi; });
                        if (iv != merged.end() && (long long)iv->lo < end) {
                            text.append((const char *)inf.bytes.data() + at, 8 + ls + li);
                            kept++;
                        }
                    }
                    at += 8 + ls + li;
                }
                return;
            }
            for (;;) {
                while (at >= inf.bytes.size())
                    if (!inflate_one(file, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks)) return;
                while (blk + 1 < inf.block_at.size() && inf.block_at[blk + 1].second <= at) blk++;
                {
                    const uint64_t voff = ((uint64_t)inf.block_at[blk].first << 16) | (uint64_t)(at - inf.block_at[blk].second);
                    if (voff >= ck.end) break;
                }
                const uint8_t *nl;
                bool eof = false;
                while (!(nl = (const uint8_t *)memchr(inf.bytes.data() + at, '\n', inf.bytes.size() - at)))
                    if (!inflate_one(file, inf.next_coff, inf, z, cbuf, &file_bytes, &blocks)) { eof = true; break; }
                const size_t e = eof ? inf.bytes.size() : (size_t)(nl - inf.bytes.data());
                const char *L = (const char *)inf.bytes.data() + at;
                const size_t len = e - at;
                walked++;
                if (len && L[0] != '#') {
                    // CHROM, POS, REF (columns 1, 2, 4) and INFO/END (column 8)
                    const char *c1 = (const char *)memchr(L, '\t', len);
                    const char *c2 = c1 ? (const char *)memchr(c1 + 1, '\t', len - (size_t)(c1 + 1 - L)) : nullptr;
                    if (c1 && c2 && (size_t)(c1 - L) == name.size() && memcmp(L, name.data(), name.size()) == 0) {
                        long long pos1 = 0;
                        if (parse_int(Str{c1 + 1, (size_t)(c2 - c1 - 1)}, pos1)) {
                            const long long pos0 = pos1 - 1;
                            if (pos0 >= (long long)last_hi) break; // sorted by position: nothing further on can overlap an interval
                            long long end = pos0 + 1;
                            const char *c3 = (const char *)memchr(c2 + 1, '\t', len - (size_t)(c2 + 1 - L));
                            const char *c4 = c3 ? (const char *)memchr(c3 + 1, '\t', len - (size_t)(c3 + 1 - L)) : nullptr;
                            if (c3 && c4) end = pos0 + std::max<long long>(1, (long long)(c4 - c3 - 1));
                            // (INFO/END may reach further: looked for in the INFO column)
                            const char *c5 = c4 ? (const char *)memchr(c4 + 1, '\t', len - (size_t)(c4 + 1 - L)) : nullptr;
                            const char *c6 = c5 ? (const char *)memchr(c5 + 1, '\t', len - (size_t)(c5 + 1 - L)) : nullptr;
                            const char *c7 = c6 ? (const char *)memchr(c6 + 1, '\t', len - (size_t)(c6 + 1 - L)) : nullptr;
                            const char *c8 = c7 ? (const char *)memchr(c7 + 1, '\t', len - (size_t)(c7 + 1 - L)) : nullptr;
                            const char *ie = c8 ? c8 : L + len;
                            for (const char *q = c7 ? c7 : L + len; q && q + 4 < ie;) {
                                const char *h = (const char *)memmem(q, (size_t)(ie - q), "END=", 4);
                                if (!h) break;
                                if (h[-1] == ';' || h[-1] == '\t') {
                                    long long ev = 0;
                                    const char *s0 = h + 4, *s1 = s0;
                                    while (s1 < ie && *s1 >= '0' && *s1 <= '9') s1++;
                                    if (s1 > s0 && parse_int(Str{s0, (size_t)(s1 - s0)}, ev)) end = std::max(end, ev);
                                }
                                q = h + 4;
                            }
                            // the merged intervals are disjoint and ascending: the first one that ends beyond the record's start decides
                            auto iv = std::upper_bound(merged.begin(), merged.end(), pos0, [](long long key, const Iv &x) { return key < (long long)x.hi; });
                            if (iv != merged.end() && (long long)iv->lo < end) {
                                text.append(L, len);
                                text.push_back('\n');
                                kept++;
                            }
                        }
                    }
                }
                if (eof) break;
                at = e + 1;
            }
        });
        // the table's text: the header, then every item's records in file order -- each copied ONCE, by the workers, to its place (appending
        // them to one string and copying that again were two serial passes over the kept text: a quarter of a 20 k-window decode's wall time)
        std::vector<size_t> at(items.size() + 1, text.size());
        for (size_t k = 0; k < items.size(); k++) { at[k + 1] = at[k] + texts[k].size(); file_bytes += st_bytes[k]; blocks += st_blocks[k]; walked += st_walked[k]; kept += st_kept[k]; }
        V.text.alloc(at[items.size()]);
        memcpy(V.text.data(), text.data(), text.size());
        parallel_dynamic((int64_t)items.size(), threads, [&](int64_t k, int) {
            if (!texts[(size_t)k].empty()) memcpy(V.text.data() + at[(size_t)k], texts[(size_t)k].data(), texts[(size_t)k].size());
            std::string().swap(texts[(size_t)k]);
        });
    } catch (...) { inflateEnd(&z); throw; }
    inflateEnd(&z);
    decode_text(V, threads);
    V.io_stats[0] = file_bytes; V.io_stats[1] = blocks; V.io_stats[2] = walked; V.io_stats[3] = kept;
}

void decode_text(uz_vcf &V, int threads) {
    const char *T = (const char *)V.text.data();
    const size_t N = V.text.size();
    if (N >= 5 && memcmp(T, "BCF\2", 4) == 0) { decode_bcf(V, threads); return; }
    // line starts: header first (serial, short), then records
    size_t p = 0;
    bool have_chrom_line = false;
    while (p < N) {
        const char *nl = (const char *)memchr(T + p, '\n', N - p);
        const size_t e = nl ? (size_t)(nl - T) : N;
        if (e == p) { p = e + 1; continue; } // empty line
        if (T[p] != '#') break;
        if (!V.header.empty()) V.header.push_back('\n');
        V.header.append(T + p, e - p);
        if (e - p >= 2 && T[p + 1] != '#') { // #CHROM line: sample names from column 10 on
            have_chrom_line = true;
            int col = 0;
            size_t a = p;
            while (a <= e) {
                const char *tb = (const char *)memchr(T + a, '\t', e - a);
                const size_t b = tb ? (size_t)(tb - T) : e;
                if (col >= 9) V.samples.emplace_back(T + a, b - a);
                col++;
                if (!tb) break;
                a = b + 1;
            }
        }
        p = e + 1;
    }
    (void)have_chrom_line;
    const size_t body = p;
    // record lines (parallel newline scan)
    {
        const int64_t span = (int64_t)(N - body);
        const int W = (int)std::min<int64_t>(threads, std::max<int64_t>(span >> 16, 1));
        std::vector<std::vector<uint64_t>> starts((size_t)W);
        parallel_slices(span, W, [&](int64_t lo, int64_t hi, int w) {
            // a worker owns the lines that START in its slice
            size_t a = body + (size_t)lo;
            const size_t stop = body + (size_t)hi;
            if (lo > 0) {
                const char *nl = (const char *)memchr(T + a - 1, '\n', N - (a - 1));
                if (!nl) return;
                a = (size_t)(nl - T) + 1;
            }
            while (a < stop && a < N) {
                const char *nl = (const char *)memchr(T + a, '\n', N - a);
                const size_t e = nl ? (size_t)(nl - T) : N;
                if (e > a) starts[(size_t)w].push_back(a);
                a = e + 1;
            }
        });
        for (auto &s : starts) V.line_at.insert(V.line_at.end(), s.begin(), s.end());
    }
    const int64_t n = (int64_t)V.line_at.size();
    V.n = n;
    threads = workers_for(n, threads, 2048);
    const size_t un = (size_t)n, ns = V.samples.size();
    V.line_len.resize(un); V.ref_at.resize(un); V.ref_len.resize(un); V.alt_at.resize(un); V.alt_len.resize(un);
    V.chrom_len.resize(un);
    V.pos.resize(un); V.end.resize(un); V.sflags.resize(un); V.ref_base.resize(un); V.alt_base.resize(un);
    V.gt.assign(ns * un, UZ_GT_UNKNOWN);
    V.ref_depth.assign(ns * un, -1); V.alt_depth.assign(ns * un, -1);
    V.gq.assign(ns * un, -1.0);
    parallel_slices(n, threads, [&](int64_t lo, int64_t hi, int) {
        std::vector<Str> f;
        for (int64_t i = lo; i < hi; i++) {
            const size_t a = (size_t)V.line_at[(size_t)i];
            const char *nl = (const char *)memchr(T + a, '\n', N - a);
            const size_t e = nl ? (size_t)(nl - T) : N;
            V.line_len[(size_t)i] = (uint32_t)(e - a);
            f.clear();
            for (size_t q = a;;) {
                const char *tb = (const char *)memchr(T + q, '\t', e - q);
                const size_t b = tb ? (size_t)(tb - T) : e;
                f.push_back(Str{T + q, b - q});
                if (!tb) break;
                q = b + 1;
            }
            if (f.size() < 5) fail(UZ_IO_E_FORMAT, "VCF record with fewer than 5 columns at line %lld", (long long)i);
            long long pos1;
            if (!parse_int(f[1], pos1)) fail(UZ_IO_E_FORMAT, "unparsable POS at record %lld", (long long)i);
            const Str ref = f[3], alt = f[4];
            V.chrom_len[(size_t)i] = (uint32_t)f[0].n;
            V.ref_at[(size_t)i] = (uint32_t)(ref.p - (T + a)); V.ref_len[(size_t)i] = (uint32_t)ref.n;
            V.alt_at[(size_t)i] = (uint32_t)(alt.p - (T + a)); V.alt_len[(size_t)i] = (uint32_t)alt.n;
            const long long start = pos1 - 1;
            long long end = start + (long long)ref.n;
            if (f.size() > 7 && !f[7].eq(".")) { // INFO/END: the last END=... that is present wins, as in a dict
                const char *q = f[7].p, *ie = f[7].p + f[7].n;
                bool has_end = false;
                Str endv{nullptr, 0};
                while (q <= ie) {
                    const char *sc = (const char *)memchr(q, ';', (size_t)(ie - q));
                    const char *stop = sc ? sc : ie;
                    const char *eq = (const char *)memchr(q, '=', (size_t)(stop - q));
                    if (eq && eq - q == 3 && memcmp(q, "END", 3) == 0) { has_end = true; endv = Str{eq + 1, (size_t)(stop - eq - 1)}; }
                    else if (!eq && stop - q == 3 && memcmp(q, "END", 3) == 0) { has_end = true; endv = Str{nullptr, 0}; } // flag: int(True) raises
                    if (!sc) break;
                    q = sc + 1;
                }
                long long ev;
                if (has_end && !endv.p) end = 1; // a bare END flag: int(True)
                else if (has_end && parse_int(endv, ev)) end = ev;
            }
            V.pos[(size_t)i] = (int32_t)start;
            V.end[(size_t)i] = (int32_t)end;
            // complex: len(ALT) != 1 or len(REF) > 1 or "*" in ALT or len(ALT[0]) > 1
            bool cx = ref.n > 1;
            if (alt.eq(".")) cx = true; // no ALT allele
            else {
                size_t n_alt = 1;
                bool star = false;
                const char *q = alt.p, *ae = alt.p + alt.n;
                for (;;) {
                    const char *cm = (const char *)memchr(q, ',', (size_t)(ae - q));
                    const char *stop = cm ? cm : ae;
                    if (stop - q == 1 && *q == '*') star = true;
                    if (!cm) break;
                    n_alt++;
                    q = cm + 1;
                }
                if (n_alt != 1 || star || alt.n > 1) cx = true;
            }
            V.sflags[(size_t)i] = cx ? UZ_SF_COMPLEX : 0;
            V.ref_base[(size_t)i] = cx || ref.n != 1 ? 0 : (uint8_t)ref.p[0];
            V.alt_base[(size_t)i] = cx || alt.n != 1 ? 0 : (uint8_t)alt.p[0];
            if (ns == 0 || f.size() <= 8) continue;
            // FORMAT keys: the last occurrence of a key wins
            int k_gt = -1, k_ad = -1, k_ro = -1, k_ao = -1, k_gq = -1;
            {
                Str pc;
                for (int k = 0; piece(f[8], k, pc); k++) {
                    if (pc.eq("GT")) k_gt = k; else if (pc.eq("AD")) k_ad = k; else if (pc.eq("RO")) k_ro = k;
                    else if (pc.eq("AO")) k_ao = k; else if (pc.eq("GQ")) k_gq = k;
                }
            }
            for (size_t s = 0; s < ns; s++) {
                const Str col = 9 + s < f.size() ? f[9 + s] : Str{".", 1};
                const size_t o = s * un + (size_t)i;
                Str v;
                if (k_gt >= 0 && piece(col, k_gt, v)) V.gt[o] = (uint8_t)parse_gt(v);
                bool ad_done = false;
                if (k_ad >= 0 && piece(col, k_ad, v) && !v.eq(".")) {
                    const char *cm = (const char *)memchr(v.p, ',', v.n);
                    const Str a0{v.p, cm ? (size_t)(cm - v.p) : v.n};
                    V.ref_depth[o] = (int32_t)num_int(a0);
                    if (cm) {
                        const char *c2 = (const char *)memchr(cm + 1, ',', (size_t)(v.p + v.n - cm - 1));
                        const Str a1{cm + 1, c2 ? (size_t)(c2 - cm - 1) : (size_t)(v.p + v.n - cm - 1)};
                        V.alt_depth[o] = (int32_t)num_int(a1);
                    } else V.alt_depth[o] = -1;
                    ad_done = true;
                }
                Str ro, ao;
                if (!ad_done && k_ro >= 0 && k_ao >= 0 && piece(col, k_ro, ro) && piece(col, k_ao, ao)) {
                    V.ref_depth[o] = (int32_t)num_int(ro);
                    const char *cm = (const char *)memchr(ao.p, ',', ao.n);
                    V.alt_depth[o] = (int32_t)num_int(Str{ao.p, cm ? (size_t)(cm - ao.p) : ao.n});
                }
                if (k_gq >= 0 && piece(col, k_gq, v)) V.gq[o] = num_float(v);
            }
        }
    });
    // contigs in order of appearance; records must be grouped by contig and sorted inside one
    V.contig_off.clear();
    for (int64_t i = 0; i < n; i++) {
        const char *c = T + V.line_at[(size_t)i];
        const size_t cl = V.chrom_len[(size_t)i];
        const bool same = !V.contigs.empty() && V.contigs.back().size() == cl && memcmp(V.contigs.back().data(), c, cl) == 0;
        if (!same) {
            const std::string name(c, cl);
            for (auto &k : V.contigs) if (k == name) fail(UZ_IO_E_UNSORTED, "sites records are not grouped by contig: %s", name.c_str());
            V.contigs.push_back(name);
            V.contig_off.push_back(i);
        } else if (V.pos[(size_t)i] < V.pos[(size_t)i - 1])
            fail(UZ_IO_E_UNSORTED, "sites records are not sorted by position on %s", V.contigs.back().c_str());
    }
    V.contig_off.push_back(n);
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

int uz_vcf_decode(const char *path, int threads, uz_vcf **out) {
    if (!path || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    *out = nullptr;
    uz_vcf *h = nullptr;
    const int rc = guarded([&] {
        h = new uz_vcf();
        decode(*h, path, resolve_threads(threads));
    });
    if (rc != UZ_IO_OK) { delete h; return rc; }
    *out = h;
    return UZ_IO_OK;
}

int uz_vcf_decode_regions(const char *path, const char *tbi_path, int64_t n_iv, const int32_t *ref, const int32_t *lo, const int32_t *hi,
                          int threads, uz_vcf **out) {
    return guarded([&] {
        if (!path || !out || n_iv < 0 || (n_iv && (!ref || !lo || !hi))) fail(UZ_IO_E_ARG, "uz_vcf_decode_regions: bad arguments");
        std::unique_ptr<uz_vcf> V(new uz_vcf());
        decode_regions(*V, path, tbi_path, n_iv, ref, lo, hi, resolve_threads(threads));
        *out = V.release();
    });
}

int64_t uz_vcf_index_names(const char *path, const char *tbi_path, char *buf, int64_t cap) {
    int64_t total = -1;
    const int rc = guarded([&] {
        if (!path) fail(UZ_IO_E_ARG, "uz_vcf_index_names: bad arguments");
        const RegionIndex t = load_region_index(path, tbi_path, 1);
        std::string all;
        for (const std::string &n : t.names) { all += n; all.push_back('\0'); }
        if (buf && cap >= (int64_t)all.size()) memcpy(buf, all.data(), all.size());
        total = (int64_t)all.size();
    });
    return rc == UZ_IO_OK ? total : (int64_t)rc;
}

int64_t uz_index_summary(const char *path, int kind, int64_t *out, int64_t cap_refs) {
    int64_t n = -1;
    const int rc = guarded([&] {
        if (!path || (cap_refs > 0 && !out)) fail(UZ_IO_E_ARG, "uz_index_summary: bad arguments");
        std::vector<BaiRef> refs;
        if (kind == 0) refs = read_bai(path);
        else refs = read_tbi_file(path, 1).refs;
        for (size_t r = 0; r < refs.size() && (int64_t)r < cap_refs; r++) {
            uint64_t nb = 0, nc = 0, sb = 0, se = 0, sl = 0;
            for (const auto &b : refs[r].bins) {
                nb++;
                for (const Chunk &c : b.second) { nc++; sb += c.beg; se += c.end; }
            }
            for (uint64_t v : refs[r].linear) sl += v;
            const uint64_t m = (1ULL << 62) - 1;
            int64_t *o = out + 6 * r;
            o[0] = (int64_t)nb; o[1] = (int64_t)nc; o[2] = (int64_t)refs[r].linear.size(); o[3] = (int64_t)(sb & m); o[4] = (int64_t)(se & m); o[5] = (int64_t)(sl & m);
        }
        n = (int64_t)refs.size();
    });
    return rc == UZ_IO_OK ? n : (int64_t)rc;
}

void uz_vcf_io_stats(const uz_vcf *h, int64_t out[4]) {
    for (int k = 0; k < 4; k++) out[k] = h ? h->io_stats[k] : 0;
}

void uz_vcf_free(uz_vcf *h) { delete h; }

int uz_vcf_view_get(const uz_vcf *h, uz_vcf_view *v) {
    if (!h || !v) { last_error = "null argument"; return UZ_IO_E_ARG; }
    memset(v, 0, sizeof(*v));
    v->n_sites = h->n;
    v->n_samples = (int32_t)h->samples.size();
    v->n_contigs = (int32_t)h->contigs.size();
    v->contig_off = h->contig_off.data();
    v->pos = h->pos.data(); v->end = h->end.data();
    v->sflags = h->sflags.data(); v->ref_base = h->ref_base.data(); v->alt_base = h->alt_base.data();
    v->gt = h->gt.data(); v->ref_depth = h->ref_depth.data(); v->alt_depth = h->alt_depth.data(); v->gq = h->gq.data();
    return UZ_IO_OK;
}

const char *uz_vcf_sample(const uz_vcf *h, int32_t i) {
    return (h && i >= 0 && (size_t)i < h->samples.size()) ? h->samples[(size_t)i].c_str() : nullptr;
}
const char *uz_vcf_contig(const uz_vcf *h, int32_t i) {
    return (h && i >= 0 && (size_t)i < h->contigs.size()) ? h->contigs[(size_t)i].c_str() : nullptr;
}
const char *uz_vcf_ref(const uz_vcf *h, int64_t i, int32_t *len) {
    if (!h || i < 0 || i >= h->n) return nullptr;
    if (len) *len = (int32_t)h->ref_len[(size_t)i];
    return (const char *)h->text.data() + h->line_at[(size_t)i] + h->ref_at[(size_t)i];
}
const char *uz_vcf_alt(const uz_vcf *h, int64_t i, int32_t *len) {
    if (!h || i < 0 || i >= h->n) return nullptr;
    if (len) *len = (int32_t)h->alt_len[(size_t)i];
    if (h->is_bcf) return h->pool.data() + h->alt_pool_at[(size_t)i];
    return (const char *)h->text.data() + h->line_at[(size_t)i] + h->alt_at[(size_t)i];
}
const char *uz_vcf_header(const uz_vcf *h, int64_t *len) {
    if (!h) return nullptr;
    if (len) *len = (int64_t)h->header.size();
    return h->header.c_str();
}
int uz_vcf_is_bcf(const uz_vcf *h) { return h && h->is_bcf ? 1 : 0; }

const char *uz_vcf_info(const uz_vcf *h, int64_t i, const char *key, int32_t *len) {
    if (!h || !key || i < 0 || i >= h->n) return nullptr;
    const size_t kl = strlen(key);
    if (h->is_bcf) {
        const uint8_t *D = h->text.data();
        const uint8_t *r = D + h->line_at[(size_t)i];
        const uint8_t *q = D + h->info_at[(size_t)i], *send = r + 8 + rd32(r);
        try {
            for (uint32_t k = 0; k < h->n_info[(size_t)i]; k++) {
                BcfVal kv, v;
                q = bcf_typed(q, send, kv);
                q = bcf_typed(q, send, v);
                const long long id = bcf_scalar_int(kv);
                if (id >= 0 && (size_t)id < h->dict.size() && h->dict[(size_t)id].size() == kl && memcmp(h->dict[(size_t)id].data(), key, kl) == 0) {
                    if (v.type != 7) return nullptr;
                    uint32_t n = v.n;
                    while (n > 0 && v.p[n - 1] == 0) n--; // strings may be NUL padded
                    if (len) *len = (int32_t)n;
                    return (const char *)v.p;
                }
            }
        } catch (const IoError &) {
        }
        return nullptr;
    }
    // text: column 8 of the line, `key=value` separated by ';' (the last occurrence wins, as in a dict)
    const char *L = (const char *)h->text.data() + h->line_at[(size_t)i];
    const char *E = L + h->line_len[(size_t)i];
    const char *f = L;
    for (int col = 0; col < 7 && f; col++) { f = (const char *)memchr(f, '\t', (size_t)(E - f)); if (f) f++; }
    if (!f) return nullptr;
    const char *fe = (const char *)memchr(f, '\t', (size_t)(E - f));
    if (!fe) fe = E;
    const char *best = nullptr;
    int32_t best_len = 0;
    for (const char *q = f; q <= fe;) {
        const char *sc = (const char *)memchr(q, ';', (size_t)(fe - q));
        const char *stop = sc ? sc : fe;
        if ((size_t)(stop - q) > kl && memcmp(q, key, kl) == 0 && q[kl] == '=') { best = q + kl + 1; best_len = (int32_t)(stop - best); }
        else if ((size_t)(stop - q) == kl && memcmp(q, key, kl) == 0) { best = nullptr; best_len = 0; } // a flag
        if (!sc) break;
        q = sc + 1;
    }
    if (best && len) *len = best_len;
    return best;
}

const char *uz_vcf_line(const uz_vcf *h, int64_t i, int32_t *len) {
    if (!h || i < 0 || i >= h->n) return nullptr;
    if (len) *len = (int32_t)h->line_len[(size_t)i];
    return (const char *)h->text.data() + h->line_at[(size_t)i];
}

} // extern "C"
