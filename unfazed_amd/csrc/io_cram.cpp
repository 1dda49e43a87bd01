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
