// io_cram.cpp -- native pieces of the CRAM 3.0 decoder (unfazed_amd/io_cram.py): the rANS 4x8 block codec, the one
// inner loop of the format that is hopeless in Python.  Written from the published description of the codec
// (four interleaved 32-bit states, 12-bit frequencies, order-0 and order-1 tables with run-length coded symbol lists);
// the Python decoder in io_cram.py is its twin and the tests hold the two against each other.
#include "io_common.hpp"

namespace {
using namespace uzio;

struct Tab {
    uint16_t freq[256];
    uint16_t cum[256];
    uint8_t look[4096];
    bool used = false;
};

struct Cur {
    const uint8_t *b;
    int64_t p, n;
    uint8_t u8() {
        if (p >= n) fail(UZ_IO_E_FORMAT, "rANS block: table runs past the end of the block");
        return b[p++];
    }
    int peek() const { return p < n ? b[p] : -1; }
};

void read_freqs(Cur &c, Tab &t) {
    memset(t.freq, 0, sizeof(t.freq));
    int rle = 0, j = c.u8();
    do {
        int f = c.u8();
        if (f >= 128) f = ((f & 127) << 8) | c.u8();
        if (j > 255) fail(UZ_IO_E_FORMAT, "rANS block: symbol run leaves the alphabet");
        t.freq[j] = (uint16_t)f;
        if (!rle && c.peek() == j + 1) {
            j = c.u8();
            rle = c.u8();
        } else if (rle) {
            rle--;
            j++;
        } else {
            j = c.u8();
        }
    } while (j);
    int x = 0;
    for (int s = 0; s < 256; s++) {
        const int f = t.freq[s];
        t.cum[s] = (uint16_t)x;
        if (!f) continue;
        if (x + f > 4096) fail(UZ_IO_E_FORMAT, "rANS block: frequencies exceed 4096");
        memset(t.look + x, s, (size_t)f);
        x += f;
    }
    // (a table that sums to less than 4096 leaves slots no well-formed stream lands on; they read as symbol 0)
    if (x < 4096) memset(t.look + x, 0, (size_t)(4096 - x));
    t.used = true;
}

inline void step(uint32_t &x, const Tab &t, uint8_t &sym, const uint8_t *b, int64_t &p, int64_t n) {
    const uint32_t m = x & 0xFFF;
    const uint8_t s = t.look[m];
    sym = s;
    x = t.freq[s] * (x >> 12) + m - t.cum[s];
    while (x < (1u << 23) && p < n) x = (x << 8) | b[p++];
}

void decode(const uint8_t *in, int64_t n_in, uint8_t *out, int64_t n_out) {
    if (n_in < 9) fail(UZ_IO_E_FORMAT, "rANS block shorter than its header");
    const int order = in[0];
    uint32_t csize, rsize;
    memcpy(&csize, in + 1, 4);
    memcpy(&rsize, in + 5, 4);
    if ((int64_t)csize != n_in - 9 || (int64_t)rsize != n_out) fail(UZ_IO_E_FORMAT, "rANS block sizes do not match its header");
    if (n_out == 0) return;
    Cur c{in, 9, n_in};
    if (order == 0) {
        Tab t;
        read_freqs(c, t);
        if (c.p + 16 > n_in) fail(UZ_IO_E_FORMAT, "rANS block: no room for the states");
        uint32_t st[4];
        memcpy(st, in + c.p, 16);
        int64_t p = c.p + 16;
        for (int64_t i = 0; i < n_out; i++) step(st[i & 3], t, out[i], in, p, n_in);
        return;
    }
    if (order != 1) fail(UZ_IO_E_FORMAT, "rANS block of unknown order %d", order);
    std::vector<Tab> tabs(256);
    int rle = 0, i = c.u8();
    do {
        if (i > 255) fail(UZ_IO_E_FORMAT, "rANS block: context run leaves the alphabet");
        read_freqs(c, tabs[(size_t)i]);
        if (!rle && c.peek() == i + 1) {
            i = c.u8();
            rle = c.u8();
        } else if (rle) {
            rle--;
            i++;
        } else {
            i = c.u8();
        }
    } while (i);
    if (c.p + 16 > n_in) fail(UZ_IO_E_FORMAT, "rANS block: no room for the states");
    uint32_t st[4];
    memcpy(st, in + c.p, 16);
    int64_t p = c.p + 16;
    const int64_t q = n_out >> 2;
    int64_t idx[4] = {0, q, 2 * q, 3 * q};
    uint8_t last[4] = {0, 0, 0, 0};
    for (int64_t t = 0; t < q; t++)
        for (int k = 0; k < 4; k++) {
            const Tab &tb = tabs[last[k]];
            if (!tb.used) fail(UZ_IO_E_FORMAT, "rANS block: context without a table");
            step(st[k], tb, out[idx[k]], in, p, n_in);
            last[k] = out[idx[k]++];
        }
    for (int64_t j = idx[3]; j < n_out; j++) {
        const Tab &tb = tabs[last[3]];
        if (!tb.used) fail(UZ_IO_E_FORMAT, "rANS block: context without a table");
        step(st[3], tb, out[j], in, p, n_in);
        last[3] = out[j];
    }
}
} // namespace

extern "C" int uz_rans4x8_decode(const uint8_t *in, int64_t n_in, uint8_t *out, int64_t n_out) {
    try {
        if (!in || (!out && n_out) || n_in < 0 || n_out < 0) fail(UZ_IO_E_ARG, "uz_rans4x8_decode: bad arguments");
        decode(in, n_in, out, n_out);
        return UZ_IO_OK;
    } catch (const IoError &e) {
        last_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        last_error = e.what();
        return UZ_IO_E_FORMAT;
    }
}

// =========================================================================================================
// Record layer of a CRAM 3.0 slice: compression header + slice blocks (already inflated by the caller) -> the slice's
// alignment records as uncompressed BAM records, which the BAM table builder (io_bam.cpp, uz_bam_decode_memory) takes
// from there.  A port of _decode_records() in unfazed_amd/io_cram.py, which stays as the readable statement of the
// layout (and decodes what this does not take: multi-reference slices); tests hold the two against each other.
#include <map>
#include <memory>

namespace {

struct In {
    const uint8_t *b = nullptr;
    int64_t p = 0, n = 0;
    void need(int64_t k) const { if (p + k > n) fail(UZ_IO_E_FORMAT, "CRAM: a data series runs past the end of its block"); }
    int u8() { need(1); return b[p++]; }
    int32_t itf8() {
        need(1);
        const uint32_t b0 = b[p];
        uint32_t v;
        if (b0 < 0x80) { p += 1; return (int32_t)b0; }
        if (b0 < 0xC0) { need(2); v = ((b0 & 0x3F) << 8) | b[p + 1]; p += 2; return (int32_t)v; }
        if (b0 < 0xE0) { need(3); v = ((b0 & 0x1F) << 16) | ((uint32_t)b[p + 1] << 8) | b[p + 2]; p += 3; return (int32_t)v; }
        if (b0 < 0xF0) { need(4); v = ((b0 & 0x0F) << 24) | ((uint32_t)b[p + 1] << 16) | ((uint32_t)b[p + 2] << 8) | b[p + 3]; p += 4; return (int32_t)v; }
        need(5);
        v = ((b0 & 0x0F) << 28) | ((uint32_t)b[p + 1] << 20) | ((uint32_t)b[p + 2] << 12) | ((uint32_t)b[p + 3] << 4) | (b[p + 4] & 0x0F);
        p += 5;
        return (int32_t)v;
    }
    const uint8_t *take(int64_t k) { if (k < 0) fail(UZ_IO_E_FORMAT, "CRAM: negative length"); need(k); const uint8_t *q = b + p; p += k; return q; }
    const uint8_t *until(int stop, int64_t &len) {
        const void *z = p < n ? memchr(b + p, stop, (size_t)(n - p)) : nullptr;
        if (!z) fail(UZ_IO_E_FORMAT, "CRAM: byte array without its stop byte");
        const uint8_t *q = b + p;
        len = (const uint8_t *)z - q;
        p += len + 1;
        return q;
    }
    std::vector<int32_t> itf8_array() {
        const int32_t k = itf8();
        if (k < 0 || k > n) fail(UZ_IO_E_FORMAT, "CRAM: bad array length");
        std::vector<int32_t> v((size_t)k);
        for (auto &x : v) x = itf8();
        return v;
    }
};

struct Bits { // the core block, most significant bit first
    const uint8_t *d = nullptr;
    int64_t p = 0, n = 0; // in bits
    uint32_t bits(int k) {
        if (k <= 0) return 0;
        if (k > 32 || p + k > n) fail(UZ_IO_E_FORMAT, "CRAM: core block exhausted");
        uint64_t v = 0;
        int64_t q = p;
        int left = k;
        while (left > 0) {
            const int in_byte = 8 - (int)(q & 7);
            const int take = left < in_byte ? left : in_byte;
            const uint32_t byte = d[q >> 3];
            v = (v << take) | ((byte >> (in_byte - take)) & ((1u << take) - 1));
            q += take;
            left -= take;
        }
        p = q;
        return (uint32_t)v;
    }
};

struct Enc {
    int kind = 0;
    int32_t a = 0, b = 0;
    std::vector<int32_t> syms, lens;
    std::shared_ptr<Enc> e1, e2;
};

Enc read_encoding(In &in) {
    Enc e;
    e.kind = in.itf8();
    const int32_t n = in.itf8();
    In sub{in.take(n), 0, n};
    switch (e.kind) {
    case 0: break;
    case 1: e.a = sub.itf8(); break;
    case 2: case 8: case 6: case 7: e.a = sub.itf8(); e.b = sub.itf8(); break;
    case 3: e.syms = sub.itf8_array(); e.lens = sub.itf8_array(); break;
    case 4: e.e1 = std::make_shared<Enc>(read_encoding(sub)); e.e2 = std::make_shared<Enc>(read_encoding(sub)); break;
    case 5: e.a = sub.u8(); e.b = sub.itf8(); break;
    case 9: e.a = sub.itf8(); break;
    default: fail(UZ_IO_E_FORMAT, "CRAM: unknown encoding id %d", e.kind);
    }
    return e;
}

struct Streams {
    Bits core;
    std::map<int32_t, In> ext;
    In *cursor(int32_t id) { return &ext[id]; } // (a series that is declared but never used has no block: reading it runs off the end)
};

struct Dec {
    int kind = -1; // -1: the series has no encoding
    bool as_byte = false;
    In *ext = nullptr;
    Bits *core = nullptr;
    int32_t off = 0, nb = 0;
    int stop = 0;
    // HUFFMAN, canonical: entries sorted by (length, symbol)
    std::vector<int32_t> h_sym;
    std::vector<int> h_len;
    std::vector<uint32_t> h_code;
    std::shared_ptr<Dec> len, val;

    int32_t get() {
        switch (kind) {
        case 1: return as_byte ? ext->u8() : ext->itf8();
        case 3: {
            if (h_sym.size() == 1 && h_len[0] == 0) return h_sym[0];
            uint32_t code = 0;
            int have = 0;
            for (size_t i = 0; i < h_sym.size(); i++) {
                if (h_len[i] > have) { code = (code << (h_len[i] - have)) | core->bits(h_len[i] - have); have = h_len[i]; }
                if (h_code[i] == code) return h_sym[i];
            }
            fail(UZ_IO_E_FORMAT, "CRAM: bad HUFFMAN code word");
        }
        case 6: return (int32_t)core->bits(nb) - off;
        case 9: {
            int nz = 0;
            while (core->bits(1) == 0) if (++nz > 31) fail(UZ_IO_E_FORMAT, "CRAM: bad GAMMA code");
            return (int32_t)(((uint32_t)1 << nz) | core->bits(nz)) - off;
        }
        case 7: {
            int u = 0;
            while (core->bits(1) == 1) if (++u > 31) fail(UZ_IO_E_FORMAT, "CRAM: bad SUBEXP code");
            if (u == 0) return (int32_t)core->bits(nb) - off;
            const int t = u + nb - 1;
            if (t > 31) fail(UZ_IO_E_FORMAT, "CRAM: bad SUBEXP code");
            return (int32_t)(((uint32_t)1 << t) | core->bits(t)) - off;
        }
        case 2: case 8: fail(UZ_IO_E_FORMAT, "CRAM: GOLOMB / GOLOMB_RICE codes are not decoded by this build");
        case -1: case 0: fail(UZ_IO_E_FORMAT, "CRAM: a data series without an encoding is read");
        default: fail(UZ_IO_E_FORMAT, "CRAM: encoding %d cannot code a single value", kind);
        }
    }
    void get_n(int64_t n, std::vector<uint8_t> &out) { // n values of a byte series, appended
        if (n < 0) fail(UZ_IO_E_FORMAT, "CRAM: negative length");
        if (kind == 1) { const uint8_t *q = ext->take(n); out.insert(out.end(), q, q + n); return; }
        for (int64_t i = 0; i < n; i++) out.push_back((uint8_t)get());
    }
    void get_array(std::vector<uint8_t> &out) { // one byte array, appended
        if (kind == 4) { const int32_t n = len->get(); val->get_n(n, out); return; }
        if (kind == 5) { int64_t n = 0; const uint8_t *q = ext->until(stop, n); out.insert(out.end(), q, q + n); return; }
        if (kind <= 0) fail(UZ_IO_E_FORMAT, "CRAM: a data series without an encoding is read");
        fail(UZ_IO_E_FORMAT, "CRAM: encoding %d cannot code a byte array", kind);
    }
};

std::shared_ptr<Dec> make_dec(const Enc &e, Streams &st, bool as_byte) {
    auto d = std::make_shared<Dec>();
    d->kind = e.kind;
    d->as_byte = as_byte;
    d->core = &st.core;
    switch (e.kind) {
    case 1: d->ext = st.cursor(e.a); break;
    case 3: {
        if (e.syms.size() != e.lens.size() || e.syms.empty()) fail(UZ_IO_E_FORMAT, "CRAM: bad HUFFMAN code");
        std::vector<size_t> order(e.syms.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return e.lens[x] < e.lens[y] || (e.lens[x] == e.lens[y] && e.syms[x] < e.syms[y]); });
        uint32_t code = 0;
        int prev = e.lens[order[0]];
        for (size_t i : order) {
            const int ln = e.lens[i];
            if (ln < 0 || ln > 31) fail(UZ_IO_E_FORMAT, "CRAM: bad HUFFMAN code length");
            code <<= (ln - prev);
            d->h_sym.push_back(e.syms[i]); d->h_len.push_back(ln); d->h_code.push_back(code);
            code += 1;
            prev = ln;
        }
        break;
    }
    case 4: d->len = make_dec(*e.e1, st, false); d->val = make_dec(*e.e2, st, true); break;
    case 5: d->stop = e.a; d->ext = st.cursor(e.b); break;
    case 6: d->off = e.a; d->nb = e.b; break;
    case 7: d->off = e.a; d->nb = e.b; break;
    case 9: d->off = e.a; break;
    default: break;
    }
    return d;
}

struct CompHdr {
    bool rn = true, ap_delta = true, rr = true;
    uint8_t sm[5] = {0x1B, 0x1B, 0x1B, 0x1B, 0x1B};
    std::vector<std::vector<std::array<uint8_t, 3>>> td;
    std::map<std::string, Enc> series;
    std::map<int32_t, Enc> tags;
    uint8_t subst[256][4];
};

CompHdr read_comp_header(const uint8_t *data, int64_t n) {
    CompHdr h;
    In in{data, 0, n};
    in.itf8();
    const int32_t n_pres = in.itf8();
    bool have_td = false;
    for (int32_t i = 0; i < n_pres; i++) {
        const uint8_t *k = in.take(2);
        if (k[0] == 'R' && k[1] == 'N') h.rn = in.u8() != 0;
        else if (k[0] == 'A' && k[1] == 'P') h.ap_delta = in.u8() != 0;
        else if (k[0] == 'R' && k[1] == 'R') h.rr = in.u8() != 0;
        else if (k[0] == 'S' && k[1] == 'M') memcpy(h.sm, in.take(5), 5);
        else if (k[0] == 'T' && k[1] == 'D') {
            const int32_t len = in.itf8();
            const uint8_t *blob = in.take(len);
            int64_t a = 0;
            while (a < len) {
                const void *z = memchr(blob + a, 0, (size_t)(len - a));
                const int64_t e = z ? (const uint8_t *)z - blob : len;
                std::vector<std::array<uint8_t, 3>> line;
                for (int64_t q = a; q + 3 <= e; q += 3) line.push_back({blob[q], blob[q + 1], blob[q + 2]});
                h.td.push_back(std::move(line));
                a = e + 1;
            }
            have_td = true;
        } else fail(UZ_IO_E_FORMAT, "CRAM: unknown preservation map key %c%c", k[0], k[1]);
    }
    if (!have_td || h.td.empty()) h.td.push_back({});
    in.itf8();
    const int32_t n_ser = in.itf8();
    for (int32_t i = 0; i < n_ser; i++) {
        const uint8_t *k = in.take(2);
        const std::string key((const char *)k, 2);
        h.series[key] = read_encoding(in);
    }
    in.itf8();
    const int32_t n_tag = in.itf8();
    for (int32_t i = 0; i < n_tag; i++) {
        const int32_t key = in.itf8();
        h.tags[key] = read_encoding(in);
    }
    const char *acgtn = "ACGTN";
    for (int c = 0; c < 256; c++) for (int k = 0; k < 4; k++) h.subst[c][k] = 'N';
    for (int r = 0; r < 5; r++) {
        int k = 0;
        for (int x = 0; x < 5; x++) {
            if (x == r) continue;
            h.subst[(uint8_t)acgtn[r]][(h.sm[r] >> (6 - 2 * k)) & 3] = (uint8_t)acgtn[x];
            k++;
        }
    }
    for (int c = 0; c < 256; c++) // a reference base that is not A C G T reads as N
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') memcpy(h.subst[c], h.subst[(uint8_t)'N'], 4);
    return h;
}

struct Rec {
    uint32_t flag = 0, cf = 0;
    int32_t tid = -1, pos1 = 0, aend1 = 0, mapq = 0, rl = 0;
    size_t cig_at = 0, n_cig = 0, seq_at = 0, qual_at = 0, name_at = 0, name_len = 0;
    bool has_seq = false, has_qual = false, has_name = false, has_sa = false, tlen_known = true;
    int32_t mate_line = -1, mtid = -1, mpos1 = 0, tlen = 0;
};

const uint32_t FUNMAP_ = 4, FMUNMAP_ = 8, FREVERSE_ = 16, FMREVERSE_ = 32, FREAD1_ = 64;

void slice_to_bam(const uz_cram_slice &S, std::vector<uint8_t> &out) {
    const CompHdr H = read_comp_header(S.comp_header, S.n_comp_header);
    Streams st;
    st.core.d = S.core; st.core.n = 8 * S.n_core;
    for (int32_t i = 0; i < S.n_ext; i++) st.ext[S.ext_id[i]] = In{S.ext[i], 0, S.n_ext_bytes[i]};
    auto dec = [&](const char *name, bool as_byte) -> std::shared_ptr<Dec> {
        auto it = H.series.find(name);
        if (it == H.series.end()) return std::make_shared<Dec>();
        return make_dec(it->second, st, as_byte);
    };
    auto BF = dec("BF", false), CF = dec("CF", false), RI = dec("RI", false), RL = dec("RL", false), AP = dec("AP", false), RG = dec("RG", false),
         MF = dec("MF", false), NS = dec("NS", false), NP = dec("NP", false), TS = dec("TS", false), NF = dec("NF", false), TL = dec("TL", false),
         FN = dec("FN", false), FP = dec("FP", false), DL = dec("DL", false), RS = dec("RS", false), PD = dec("PD", false), HC = dec("HC", false),
         MQ = dec("MQ", false), FC = dec("FC", true), BA = dec("BA", true), QS = dec("QS", true), BS = dec("BS", true),
         RN = dec("RN", false), IN_ = dec("IN", false), SC = dec("SC", false), BB = dec("BB", false), QQ = dec("QQ", false);
    std::map<int32_t, std::shared_ptr<Dec>> tag_dec;
    if (S.ref_id == -2) fail(UZ_IO_E_ARG, "uz_cram_slice_to_bam: multi-reference slices are decoded by the Python layer");
    const int64_t n = S.n_records;
    if (n < 0) fail(UZ_IO_E_FORMAT, "CRAM: negative record count");
    std::vector<Rec> recs((size_t)n);
    std::vector<uint32_t> cigs;
    std::vector<uint8_t> seqs, quals, names, scratch;
    int64_t last_pos = S.start;
    auto ref_at = [&](int64_t pos0) -> uint8_t {
        const int64_t k = pos0 - S.ref_start0;
        if (!S.ref) fail(UZ_IO_E_ARG, "CRAM: the slice needs its reference bases");
        return (k >= 0 && k < S.n_ref) ? S.ref[k] : (uint8_t)'N';
    };
    for (int64_t i = 0; i < n; i++) {
        Rec &r = recs[(size_t)i];
        r.flag = (uint32_t)BF->get();
        r.cf = (uint32_t)CF->get();
        r.tid = S.ref_id;
        r.rl = RL->get();
        if (r.rl < 0 || r.rl > 0xFFFFFF) fail(UZ_IO_E_FORMAT, "CRAM: bad read length");
        int64_t ap = AP->get();
        if (H.ap_delta) { ap += last_pos; last_pos = ap; }
        r.pos1 = (int32_t)ap;
        (void)RG->get();
        auto read_name = [&] {
            r.name_at = names.size();
            RN->get_array(names);
            r.name_len = names.size() - r.name_at;
            r.has_name = true;
        };
        if (H.rn) read_name();
        if (r.cf & 2) {
            const int32_t mf = MF->get();
            if (!H.rn) read_name();
            r.mtid = NS->get(); r.mpos1 = NP->get(); r.tlen = TS->get();
            if (mf & 1) r.flag |= FMREVERSE_;
            if (mf & 2) r.flag |= FMUNMAP_;
        } else if (r.cf & 4) {
            r.mate_line = (int32_t)(i + NF->get() + 1);
            r.tlen_known = false;
        }
        const int32_t tl = TL->get();
        if (tl < 0 || (size_t)tl >= H.td.size()) fail(UZ_IO_E_FORMAT, "CRAM: tag line outside the tag dictionary");
        for (const auto &ent : H.td[(size_t)tl]) {
            const int32_t key = ((int32_t)ent[0] << 16) | ((int32_t)ent[1] << 8) | ent[2];
            auto it = tag_dec.find(key);
            if (it == tag_dec.end()) {
                auto te = H.tags.find(key);
                if (te == H.tags.end()) fail(UZ_IO_E_FORMAT, "CRAM: tag %c%c has no encoding", ent[0], ent[1]);
                it = tag_dec.emplace(key, make_dec(te->second, st, true)).first;
            }
            scratch.clear();
            it->second->get_array(scratch);
            if (ent[0] == 'S' && ent[1] == 'A') r.has_sa = true;
        }
        r.cig_at = cigs.size();
        auto op = [&](uint32_t code, int64_t ln) {
            if (ln <= 0) return;
            if (cigs.size() > r.cig_at && (cigs.back() & 15) == code) cigs.back() += (uint32_t)ln << 4;
            else cigs.push_back(((uint32_t)ln << 4) | code);
        };
        const bool no_seq = (r.cf & 8) != 0;
        const size_t rl = (size_t)r.rl;
        if (!(r.flag & FUNMAP_)) {
            r.seq_at = seqs.size();
            seqs.resize(r.seq_at + rl, 'N');
            uint8_t *seq = seqs.data() + r.seq_at;
            std::vector<std::pair<int64_t, std::vector<uint8_t>>> qpatch;
            int64_t rpos = ap - 1, spos = 0, prev = 0;
            auto matches = [&](int64_t ln) {
                if (spos + ln > (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read features overrun the read length");
                if (!no_seq) for (int64_t k = 0; k < ln; k++) seq[spos + k] = ref_at(rpos + k);
                op(0, ln);
                rpos += ln; spos += ln;
            };
            auto put = [&](const std::vector<uint8_t> &v) {
                if (spos + (int64_t)v.size() > (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read features overrun the read length");
                memcpy(seq + spos, v.data(), v.size());
            };
            const int32_t fn = FN->get();
            for (int32_t f = 0; f < fn; f++) {
                const int code = FC->get();
                prev += FP->get();
                const int64_t fpos = prev - 1;
                if (fpos > spos) matches(fpos - spos);
                if (fpos < 0 || fpos > (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read feature outside the read");
                switch (code) {
                case 'X': {
                    if (spos >= (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read features overrun the read length");
                    const uint8_t rb = no_seq && !S.ref ? (uint8_t)'N' : ref_at(rpos);
                    seq[spos] = H.subst[rb][BS->get() & 3];
                    op(0, 1); rpos++; spos++;
                    break;
                }
                case 'B': {
                    if (spos >= (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read features overrun the read length");
                    seq[spos] = (uint8_t)BA->get();
                    qpatch.push_back({spos, {(uint8_t)QS->get()}});
                    op(0, 1); rpos++; spos++;
                    break;
                }
                case 'b': { scratch.clear(); BB->get_array(scratch); put(scratch); op(0, (int64_t)scratch.size()); rpos += scratch.size(); spos += scratch.size(); break; }
                case 'I': { scratch.clear(); IN_->get_array(scratch); put(scratch); op(1, (int64_t)scratch.size()); spos += scratch.size(); break; }
                case 'i': {
                    if (spos >= (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: read features overrun the read length");
                    seq[spos] = (uint8_t)BA->get(); op(1, 1); spos++;
                    break;
                }
                case 'S': { scratch.clear(); SC->get_array(scratch); put(scratch); op(4, (int64_t)scratch.size()); spos += scratch.size(); break; }
                case 'D': { const int32_t ln = DL->get(); op(2, ln); rpos += ln; break; }
                case 'N': { const int32_t ln = RS->get(); op(3, ln); rpos += ln; break; }
                case 'H': op(5, HC->get()); break;
                case 'P': op(6, PD->get()); break;
                case 'Q': qpatch.push_back({fpos, {(uint8_t)QS->get()}}); break;
                case 'q': { scratch.clear(); QQ->get_array(scratch); qpatch.push_back({fpos, scratch}); break; }
                default: fail(UZ_IO_E_FORMAT, "CRAM: unknown read feature %d", code);
                }
            }
            if (spos < (int64_t)rl) matches((int64_t)rl - spos);
            r.mapq = MQ->get();
            r.qual_at = quals.size();
            if (r.cf & 1) { QS->get_n((int64_t)rl, quals); r.has_qual = true; }
            else if (!qpatch.empty()) {
                quals.resize(r.qual_at + rl, 0xFF);
                for (auto &pq : qpatch) {
                    if (pq.first < 0 || pq.first + (int64_t)pq.second.size() > (int64_t)rl) fail(UZ_IO_E_FORMAT, "CRAM: quality feature outside the read");
                    memcpy(quals.data() + r.qual_at + pq.first, pq.second.data(), pq.second.size());
                }
                r.has_qual = true;
            }
            r.aend1 = (int32_t)std::max<int64_t>(rpos, ap);
            r.has_seq = !no_seq;
        } else {
            r.seq_at = seqs.size();
            if (!no_seq) { BA->get_n((int64_t)rl, seqs); r.has_seq = true; }
            r.qual_at = quals.size();
            if (r.cf & 1) { QS->get_n((int64_t)rl, quals); r.has_qual = true; }
            r.mapq = 0;
            r.aend1 = (int32_t)ap;
        }
        if (r.has_qual) { // all 0xFF reads as "no qualities"
            bool any = false;
            for (size_t k = 0; k < rl && !any; k++) any = quals[r.qual_at + k] != 0xFF;
            if (!any) r.has_qual = false;
        }
        r.n_cig = cigs.size() - r.cig_at;
    }
    // mates inside the slice: chains through "records to the next fragment"; template length over the chain (htslib's rule)
    for (int64_t i = 0; i < n; i++) {
        Rec &r = recs[(size_t)i];
        if (r.mate_line < 0 || r.tlen_known) continue;
        std::vector<int64_t> chain{ i };
        int64_t j = i;
        while (recs[(size_t)j].mate_line >= 0) {
            const int64_t nxt = recs[(size_t)j].mate_line;
            if (nxt <= j || nxt >= n) fail(UZ_IO_E_FORMAT, "CRAM: a mate link leaves the slice");
            chain.push_back(nxt);
            j = nxt;
        }
        int64_t left = INT64_MAX, right = INT64_MIN;
        for (int64_t k : chain) { left = std::min<int64_t>(left, recs[(size_t)k].pos1); right = std::max<int64_t>(right, recs[(size_t)k].aend1); }
        int left_cnt = 0;
        bool same = true;
        for (int64_t k : chain) { left_cnt += recs[(size_t)k].pos1 == left; same &= recs[(size_t)k].tid == r.tid; }
        for (size_t q = 0; q < chain.size(); q++) {
            Rec &rk = recs[(size_t)chain[q]];
            const Rec &m = recs[(size_t)chain[(q + 1) % chain.size()]];
            if (same) {
                const int64_t t = right - left + 1;
                rk.tlen = (int32_t)((rk.pos1 == left && (left_cnt == 1 || (rk.flag & FREAD1_))) ? t : -t);
            } else rk.tlen = 0;
            rk.mtid = m.tid; rk.mpos1 = m.pos1;
            if (m.flag & FUNMAP_) { rk.flag |= FMUNMAP_; rk.tlen = 0; }
            if (rk.flag & FUNMAP_) rk.tlen = 0;
            if (m.flag & FREVERSE_) rk.flag |= FMREVERSE_;
            if (!rk.has_name) {
                const Rec &head = recs[(size_t)chain[0]];
                if (head.has_name) { rk.name_at = head.name_at; rk.name_len = head.name_len; rk.has_name = true; }
                else rk.mate_line = -3 - (int32_t)chain[0]; // generated below from the head's number
            }
            if (rk.mate_line >= -1) rk.mate_line = -2;
            rk.tlen_known = true;
        }
    }
    // BAM records
    uint8_t code16[256];
    memset(code16, 15, sizeof(code16)); // anything outside BAM's 16-code alphabet reads as N
    {
        const char *alphabet = "=ACMGRSVTWYHKDBN";
        for (int k = 0; k < 16; k++) {
            code16[(uint8_t)alphabet[k]] = (uint8_t)k;
            if (alphabet[k] >= 'A' && alphabet[k] <= 'Z') code16[(uint8_t)(alphabet[k] + 32)] = (uint8_t)k;
        }
    }
    char gen[48];
    for (int64_t i = 0; i < n; i++) {
        const Rec &r = recs[(size_t)i];
        const uint8_t *name = names.data() + r.name_at;
        size_t name_len = r.name_len;
        if (!r.has_name) {
            const int64_t head = r.mate_line <= -3 ? (int64_t)(-3 - r.mate_line) : i;
            name_len = (size_t)snprintf(gen, sizeof(gen), "cram:%lld", (long long)(S.counter + head));
            name = (const uint8_t *)gen;
        }
        if (name_len > 254) fail(UZ_IO_E_RANGE, "CRAM: a read name is longer than 254 bytes");
        const size_t l_seq = r.has_seq ? (size_t)r.rl : 0;
        if (r.n_cig > 0xFFFF) fail(UZ_IO_E_RANGE, "CRAM: more than 65535 CIGAR operations");
        const size_t body = 32 + name_len + 1 + 4 * r.n_cig + (l_seq + 1) / 2 + l_seq + (r.has_sa ? 5 : 0);
        const size_t at = out.size();
        out.resize(at + 4 + body, 0);
        uint8_t *p = out.data() + at;
        auto w32 = [](uint8_t *q, uint32_t v) { memcpy(q, &v, 4); };
        auto w16 = [](uint8_t *q, uint16_t v) { memcpy(q, &v, 2); };
        w32(p, (uint32_t)body);
        w32(p + 4, (uint32_t)r.tid); w32(p + 8, (uint32_t)(r.pos1 - 1));
        p[12] = (uint8_t)(name_len + 1); p[13] = (uint8_t)r.mapq; w16(p + 14, 4680);
        w16(p + 16, (uint16_t)r.n_cig); w16(p + 18, (uint16_t)(r.flag & 0xFFFF)); w32(p + 20, (uint32_t)l_seq);
        w32(p + 24, (uint32_t)r.mtid); w32(p + 28, (uint32_t)(r.mpos1 - 1)); w32(p + 32, (uint32_t)r.tlen);
        uint8_t *q = p + 36;
        memcpy(q, name, name_len); q[name_len] = 0; q += name_len + 1;
        if (r.n_cig) memcpy(q, cigs.data() + r.cig_at, 4 * r.n_cig);
        q += 4 * r.n_cig;
        for (size_t k = 0; k < l_seq; k++) q[k >> 1] |= (uint8_t)(code16[seqs[r.seq_at + k]] << ((k & 1) ? 0 : 4));
        q += (l_seq + 1) / 2;
        if (l_seq) {
            if (r.has_qual) memcpy(q, quals.data() + r.qual_at, l_seq);
            else memset(q, 0xFF, l_seq);
        }
        q += l_seq;
        if (r.has_sa) memcpy(q, "SAZ*\0", 5);
    }
}
} // namespace

extern "C" int uz_cram_slice_to_bam(const uz_cram_slice *s, uint8_t **bam, int64_t *n_bam) {
    try {
        if (!s || !bam || !n_bam || (s->n_ext && (!s->ext_id || !s->ext || !s->n_ext_bytes))) fail(UZ_IO_E_ARG, "uz_cram_slice_to_bam: bad arguments");
        std::vector<uint8_t> out;
        slice_to_bam(*s, out);
        uint8_t *p = (uint8_t *)malloc(out.size() ? out.size() : 1);
        if (!p) fail(UZ_IO_E_RANGE, "out of memory");
        memcpy(p, out.data(), out.size());
        *bam = p;
        *n_bam = (int64_t)out.size();
        return UZ_IO_OK;
    } catch (const IoError &e) {
        last_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        last_error = e.what();
        return UZ_IO_E_FORMAT;
    }
}

extern "C" void uz_io_free(void *p) { free(p); }
