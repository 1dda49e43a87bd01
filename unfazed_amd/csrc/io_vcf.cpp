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
// BCF is not decoded.
#include "io_common.hpp"

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

void decode(uz_vcf &V, const char *path, int threads) {
    {
        Bytes file = read_file(path);
        bool gz = false;
        V.text = inflate_all(file, threads, &gz);
    }
    const char *T = (const char *)V.text.data();
    const size_t N = V.text.size();
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
    return (const char *)h->text.data() + h->line_at[(size_t)i] + h->alt_at[(size_t)i];
}
const char *uz_vcf_header(const uz_vcf *h, int64_t *len) {
    if (!h) return nullptr;
    if (len) *len = (int64_t)h->header.size();
    return h->header.c_str();
}
const char *uz_vcf_line(const uz_vcf *h, int64_t i, int32_t *len) {
    if (!h || i < 0 || i >= h->n) return nullptr;
    if (len) *len = (int32_t)h->line_len[(size_t)i];
    return (const char *)h->text.data() + h->line_at[(size_t)i];
}

} // extern "C"
