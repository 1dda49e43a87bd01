// k_inflate.hip -- BGZF blocks inflated on the device (DEFLATE, RFC 1951), one wavefront per block.
//
// Why: from files to results the host is the bound -- the box's 16 CPUs inflate 17 GB/s (libdeflate) while the compressed
// windows of a batch would cross the link at 55 GB/s and the device phases what comes out of them two orders of magnitude
// faster.  A BGZF block is an independent DEFLATE stream of at most 64 KiB of output whose LZ77
// window never leaves the block: a batch's windows are tens of thousands of such blocks, enough for every wave slot of
// the chip.
//
// How: the 64 lanes of a wave run the decoder in lockstep on the SAME state (bit buffer, positions: every lane holds a
// copy, so there is no divergence and no broadcast), and differ only where there is parallel work:
//   * input: the wave reads 64 consecutive dwords of the stream with one coalesced load and hands them out with
//     v_readlane as the bit buffer drains (one memory round trip per 256 bytes of input, the next one already in flight);
//   * Huffman tables in LDS: a direct table for codes of up to 10 (literal / length) or 8 (distance) bits, filled by all
//     lanes; the rare longer codes are resolved the canonical way (count / first code per length);
//   * a match is copied by all lanes at once (lane k takes byte k, modulo the distance when source and destination
//     overlap); literals gather in a register window, one byte per lane, and leave it as one coalesced store.
// The output lies in HBM, not in LDS (64 KiB per block would leave two waves per CU): a match reads bytes this wave
// stored earlier, so its loads bypass the L1 (agent-scope loads) and wait for the wave's outstanding stores only when
// the source reaches into bytes stored since the last such wait -- most matches of alignment records point hundreds of
// bytes back and wait for nothing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include "uz_ctx.hpp"

namespace {

#define UZI_LIT_BITS 10
#define UZI_DIST_BITS 8

struct InflateLds {
    uint16_t lit_tab[1 << UZI_LIT_BITS];   // (symbol << 4) | length; 0: a longer code
    uint16_t dist_tab[1 << UZI_DIST_BITS];
    uint16_t lit_sorted[288], dist_sorted[32]; // symbols in canonical order (by length, then value)
    uint16_t lit_cnt[16], dist_cnt[16];        // codes per length
    uint8_t lens[320];                          // code lengths of the block being set up: literal / length code at 0, distance code at 288
    uint8_t cl_lens[32];                        // ... and of the code they are written in
    uint16_t code_of[288];                      // bit-reversed canonical code of every symbol (table fill)
    int work[48];                               // per-length counters of the table set-up
};

__device__ __constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the input stream of one block, seen through two registers per lane: dwords [cbase, cbase + 64) and the 64 after them
struct BitIn {
    const uint32_t *w;  // aligned dwords of the compressed buffer, from the dword that holds the stream's first byte
    int cbase;          // dword index held by lane 0 of `cur`
    uint32_t cur, nxt;  // lane l: w[cbase + l], w[cbase + 64 + l]
    int widx;           // next dword to take
    int last_word;      // the last dword of the compressed buffer (its padding included): no load goes beyond it
    unsigned long long buf;
    int bits;
};
// a BGZF block is at most 64 KiB long, framing included: a decoder that has taken more dwords than that is not reading a block any more
// (an endless run of empty stored blocks, say) and is stopped before it walks out of the buffer
#define UZI_MAX_WORDS (65536 / 4 + 8)
__device__ __forceinline__ uint32_t lane_word(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(lane)); }
__device__ __forceinline__ void bi_open(BitIn &b, const uint8_t *base, long long byte_off, long long buffer_words, int lane) {
    b.w = reinterpret_cast<const uint32_t *>(base) + (byte_off >> 2);
    b.last_word = (int)(buffer_words - 1 - (byte_off >> 2) < 0x7FFFFFFF ? buffer_words - 1 - (byte_off >> 2) : 0x7FFFFFFF);
    b.cbase = 0;
    b.cur = b.w[lane < b.last_word ? lane : b.last_word];
    b.nxt = b.w[64 + lane < b.last_word ? 64 + lane : b.last_word];
    b.widx = 1;
    const int skip = (int)(byte_off & 3) * 8;
    b.buf = (unsigned long long)(lane_word(b.cur, 0) >> skip);
    b.bits = 32 - skip;
}
__device__ __forceinline__ void bi_refill(BitIn &b, int lane) { // afterwards: at least 33 bits
    if (b.bits <= 32) {
        int k = b.widx - b.cbase;
        if (k >= 64) {
            b.cur = b.nxt;
            b.cbase += 64;
            const int at = b.cbase + 64 + lane;
            b.nxt = b.w[at < b.last_word ? at : b.last_word];
            k -= 64;
        }
        b.buf |= (unsigned long long)lane_word(b.cur, k) << b.bits;
        b.bits += 32;
        b.widx++;
    }
}
__device__ __forceinline__ uint32_t bi_peek(const BitIn &b, int n) { return (uint32_t)(b.buf & ((1ULL << n) - 1ULL)); }
__device__ __forceinline__ void bi_drop(BitIn &b, int n) { b.buf >>= n; b.bits -= n; }
__device__ __forceinline__ uint32_t bi_take(BitIn &b, int n) { const uint32_t v = bi_peek(b, n); bi_drop(b, n); return v; }

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// canonical Huffman set-up from lens[0 .. n): counts, symbols in canonical order, the direct table for codes of <= tb bits.
// false: an over-subscribed or (where it matters) incomplete set of lengths
// (complete 2: a code-length code must use its code space up; 1: so must a literal / length code, unless it is ONE code of one bit -- zlib's
// inftrees.c rule ("incomplete set ... max != 1"), which the host inflaters follow; 0: a distance code need not -- the fixed one does not,
// and a block without matches has none.  An unused code then decodes to "no symbol")
// The per-length counters live in LDS (work[48]): indexed by a code length, they would otherwise sit in scratch memory or pin 48 registers
// (the kernel then needs 80 registers and runs 6 waves per SIMD at 86 GB/s instead of 8 at 97).
__device__ bool build_table(const uint8_t *lens, int n, uint16_t *tab, int tb, uint16_t *sorted, uint16_t *cnt, uint16_t *code_of, int *work, int lane,
                            int complete) {
    int *count = work, *offs = work + 16, *next = work + 32;
    __syncthreads();
    if (lane < 16) count[lane] = 0;
    __syncthreads();
    if (lane == 0) {
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        offs[1] = 0; next[1] = 0;
        for (int l = 1; l < 15; l++) { offs[l + 1] = offs[l] + count[l]; next[l + 1] = (next[l] + count[l]) << 1; }
    }
    __syncthreads();
    int left = 1, total = 0;
    for (int l = 1; l < 16; l++) {
        left = (left << 1) - count[l];
        total += count[l];
        if (left < 0) return false;
    }
    if (left > 0 && (complete == 2 || (complete == 1 && !(total == 1 && count[1] == 1)))) return false;
    if (lane < 16) cnt[lane] = (uint16_t)count[lane];
    if (lane == 0) {
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (l) {
                sorted[offs[l]] = (uint16_t)s;
                code_of[s] = (uint16_t)bitrev((uint32_t)next[l], l);
                offs[l]++; next[l]++;
            }
        }
    }
    for (int k = lane; k < (1 << tb); k += 64) tab[k] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) { // every lane fills the replicas of its symbols' codes
        const int l = lens[s];
        if (l && l <= tb) {
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (int k = code_of[s]; k < (1 << tb); k += 1 << l) tab[k] = e;
        }
    }
    __syncthreads();
    return true;
}
// one symbol: the direct table, or the canonical walk for a code longer than the table's index (rare)
__device__ __forceinline__ int decode_sym(BitIn &b, const uint16_t *tab, int tb, const uint16_t *sorted, const uint16_t *cnt) {
    const uint32_t e = tab[bi_peek(b, tb)];
    if (e) { bi_drop(b, (int)(e & 15u)); return (int)(e >> 4); }
    int code = 0, first = 0, index = 0;
    unsigned long long v = b.buf;
    for (int l = 1; l <= 15; l++) {
        code |= (int)(v & 1ULL);
        v >>= 1;
        const int c = cnt[l];
        if (code - c < first) { bi_drop(b, l); return sorted[index + (code - first)]; }
        index += c; first += c;
        first <<= 1; code <<= 1;
    }
    return -1;
}

// err: 1 bad block type / stored length, 2 bad code lengths, 3 bad symbol, 4 output overrun or distance before the block, 5 wrong size,
// 6 the stream runs on past the longest possible BGZF block
#ifndef UZI_WAVES_PER_EU
#define UZI_WAVES_PER_EU 8 // (measured 6 / 8 waves per SIMD: 86 / 97 GB/s of output; the decoder state fits 64 registers)
#endif
__global__ __launch_bounds__(64, UZI_WAVES_PER_EU) void k_bgzf_inflate(int64_t n_blocks, const uint8_t *__restrict__ comp, const int64_t *__restrict__ in_off,
                                                     const int64_t *__restrict__ out_off, uint8_t *out, int64_t comp_words, int32_t *cursor,
                                                     int32_t *err) {
    __shared__ InflateLds L;
    __shared__ int next_block;
    const int lane = threadIdx.x;
    for (;;) {
        __syncthreads();
        if (lane == 0) next_block = atomicAdd(cursor, 1);
        __syncthreads();
        const int64_t blk = next_block;
        if (blk >= n_blocks) return;
        uint8_t *o = out + out_off[blk];
        const int64_t osize = out_off[blk + 1] - out_off[blk];
        BitIn b;
        bi_open(b, comp, in_off[blk], comp_words, lane);
        int64_t pos = 0, safe = 0; // bytes written; bytes whose stores are known to have landed
        int64_t wbase = 0;         // first position of the literal window (pos - wbase bytes wait in it)
        uint32_t wbyte = 0;
        int bad = 0;
        for (int last = 0; !last && !bad;) {
            if (b.widx > UZI_MAX_WORDS) { bad = 6; break; }
            bi_refill(b, lane);
            last = (int)bi_take(b, 1);
            const int type = (int)bi_take(b, 2);
            if (type == 0) { // stored: to the byte boundary, LEN, NLEN, then the bytes
                bi_drop(b, b.bits & 7);
                bi_refill(b, lane);
                const uint32_t len = bi_take(b, 16);
                bi_refill(b, lane);
                const uint32_t nlen = bi_take(b, 16);
                if ((len ^ nlen) != 0xFFFFu || pos + (int64_t)len > osize) { bad = 1; break; }
                // the bytes: drain what the bit buffer holds (whole bytes), then dword by dword through the same window
                uint32_t done = 0;
                while (done < len) {
                    bi_refill(b, lane);
                    const int nb = min(b.bits >> 3, (int)min(len - done, 4u));
                    const uint32_t v = bi_take(b, 8 * nb);
                    if (lane < nb) o[pos + done + lane] = (uint8_t)(v >> (8 * lane));
                    done += (uint32_t)nb;
                }
                pos += len;
                wbase = pos;
                continue;
            }
            if (type == 3) { bad = 1; break; }
            int nlit, ndist;
            if (type == 1) { // fixed code
                for (int s = lane; s < 288; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                if (lane < 32) L.lens[288 + lane] = 5;
                nlit = 288; ndist = 30;
                __syncthreads();
            } else { // dynamic code: the code-length code, then the lengths of both codes
                bi_refill(b, lane);
                nlit = (int)bi_take(b, 5) + 257;
                ndist = (int)bi_take(b, 5) + 1;
                const int ncl = (int)bi_take(b, 4) + 4;
                if (nlit > 286 || ndist > 30) { bad = 2; break; }
                __syncthreads();
                if (lane < 19) L.cl_lens[lane] = 0;
                __syncthreads();
                for (int k = 0; k < ncl; k++) {
                    bi_refill(b, lane);
                    const uint32_t v = bi_take(b, 3);
                    if (lane == 0) L.cl_lens[kClOrder[k]] = (uint8_t)v;
                }
                __syncthreads();
                // (the code-length code sits where the distance code will be: 7-bit direct table, canonical arrays)
                if (!build_table(L.cl_lens, 19, L.dist_tab, 7, L.dist_sorted, L.dist_cnt, L.code_of, L.work, lane, 2)) { bad = 2; break; }
                int n = 0, prev = 0;
                while (n < nlit + ndist) {
                    bi_refill(b, lane);
                    const int s = decode_sym(b, L.dist_tab, 7, L.dist_sorted, L.dist_cnt);
                    if (s < 0) { bad = 2; break; }
                    int rep = 1, val = s;
                    if (s == 16) { if (n == 0) { bad = 2; break; } val = prev; rep = 3 + (int)bi_take(b, 2); }
                    else if (s == 17) { val = 0; rep = 3 + (int)bi_take(b, 3); }
                    else if (s == 18) { val = 0; rep = 11 + (int)bi_take(b, 7); }
                    if (n + rep > nlit + ndist) { bad = 2; break; }
                    for (int k = lane; k < rep; k += 64) { // entry n + k: a literal / length code's length, or a distance code's
                        const int at = n + k;
                        L.lens[at < nlit ? at : 288 + (at - nlit)] = (uint8_t)val;
                    }
                    n += rep;
                    prev = val;
                }
                if (bad) break;
                __syncthreads();
                if (L.lens[256] == 0) { bad = 2; break; } // no end-of-block code
            }
            if (!build_table(L.lens, nlit, L.lit_tab, UZI_LIT_BITS, L.lit_sorted, L.lit_cnt, L.code_of, L.work, lane, 1)) { bad = 2; break; }
            if (!build_table(L.lens + 288, ndist, L.dist_tab, UZI_DIST_BITS, L.dist_sorted, L.dist_cnt, L.code_of, L.work, lane, 0)) { bad = 2; break; }
            // ---- the symbols of the block.  Literals gather in a register window -- lane l holds the byte for position wbase + l -- and
            // leave it as one coalesced store when it is full or a match needs them in memory.
            for (;;) {
                if (b.widx > UZI_MAX_WORDS) { bad = 6; break; }
                bi_refill(b, lane);
                const int s = decode_sym(b, L.lit_tab, UZI_LIT_BITS, L.lit_sorted, L.lit_cnt);
                if (s < 256) {
                    if (s < 0) { bad = 3; break; }
                    if (pos >= osize) { bad = 4; break; }
                    wbyte = (lane == (int)(pos - wbase)) ? (uint32_t)s : wbyte;
                    pos++;
                    if (pos - wbase == 64) { o[wbase + lane] = (uint8_t)wbyte; wbase = pos; }
                    continue;
                }
                if (pos > wbase) { // the window's bytes into memory, in front of whatever comes next
                    if (lane < (int)(pos - wbase)) o[wbase + lane] = (uint8_t)wbyte;
                }
                if (s == 256) { wbase = pos; break; }
                if (s > 285) { bad = 3; break; }
                bi_refill(b, lane);
                // (length and distance from their symbols: RFC 1951's tables are arithmetic -- four codes per extra bit / two per extra bit)
                const int ls = s - 257;
                int len;
                if (ls < 8) len = 3 + ls;
                else if (ls == 28) len = 258;
                else { const int e = (ls >> 2) - 1; len = ((4 + (ls & 3)) << e) + 3 + (int)bi_take(b, e); }
                bi_refill(b, lane);
                const int ds = decode_sym(b, L.dist_tab, UZI_DIST_BITS, L.dist_sorted, L.dist_cnt);
                if (ds < 0 || ds > 29) { bad = 3; break; }
                bi_refill(b, lane);
                int dist;
                if (ds < 4) dist = 1 + ds;
                else { const int e = (ds >> 1) - 1; dist = ((2 + (ds & 1)) << e) + 1 + (int)bi_take(b, e); }
                if ((int64_t)dist > pos || pos + len > osize) { bad = 4; break; }
                const int64_t src = pos - dist;
                if (src + (len < dist ? len : dist) > safe) { // the source reaches into bytes whose stores may still be in flight
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                    asm volatile("" ::: "memory");
                    safe = pos;
                }
                for (int k = lane; k < len; k += 64) {
                    const int sk = dist >= len ? k : k % dist;
                    o[pos + k] = __hip_atomic_load(o + src + sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (past the L1: see the head of the file)
                }
                pos += len;
                wbase = pos;
            }
        }
        if (!bad && pos != osize) bad = 5;
        if (bad && lane == 0) atomicCAS(err, 0, bad | ((int)blk << 4));
    }
}

// ---- CRC-32 of every inflated block against the value its BGZF footer carries (what htslib's bgzf reader checks after inflating a block, and
// what the host's walk checked on blocks the device had inflated: io_stage.cpp Stream::more).  One wavefront per block: the block is cut into 64
// consecutive slices, lane l runs the table CRC over slice l (slicing by four, 16-byte loads; tables in LDS; lane 0 starts from 0xFFFFFFFF, the others
// from 0 -- the register is linear in its start value), and the slices are joined the way zlib's crc32_combine joins two: the CRC register after
// m more zero bytes is a 32 x 32 matrix over GF(2) applied to it, the matrices for 2^j zero bytes (j = 0 .. 16) are squared up once per wave in LDS,
// and lane l applies those its distance from the block's end names.  XOR over the lanes, final inversion, compare.
__device__ __forceinline__ uint32_t gf2_times(const uint32_t *mat, uint32_t vec) {
    uint32_t sum = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) sum ^= ((vec >> i) & 1u) ? mat[i] : 0u;
    return sum;
}
__global__ __launch_bounds__(64) void k_bgzf_crc32(int64_t n_blocks, const uint8_t *__restrict__ out, const int64_t *__restrict__ out_off,
                                                   const uint32_t *__restrict__ want, int32_t *err /* [1]: 0, or 1 + the first bad block seen */) {
    __shared__ uint32_t tab[4][256]; // slicing by four: tab[k][b] = the register after byte b and k more zero bytes
    __shared__ uint32_t mat[19][32]; // [0]: one zero bit, [1]: two, [2]: four; [3 + j]: 2^j zero bytes
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        tab[0][i] = c;
    }
    if (lane < 32) mat[0][lane] = lane == 0 ? 0xEDB88320u : 1u << (lane - 1);
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        for (int i = lane; i < 256; i += 64) { const uint32_t c = tab[k - 1][i]; tab[k][i] = tab[0][c & 0xFFu] ^ (c >> 8); }
        __syncthreads();
    }
    for (int j = 1; j < 19; j++) { // each squared from the one before
        if (lane < 32) mat[j][lane] = gf2_times(mat[j - 1], mat[j - 1][lane]);
        __syncthreads();
    }
    auto step4 = [&](uint32_t c, uint32_t w) {
        c ^= w;
        return tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
    };
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int64_t at = out_off[b];
        const uint32_t n = (uint32_t)(out_off[b + 1] - at);
        const uint8_t *p = out + at;
        // slices whose first byte is 16-byte aligned in memory (all but lane 0's): a lane reads its slice with 16-byte loads, eight of them to a
        // cache line, instead of byte loads 1 KiB apart in every lane
        const uint32_t a0 = (uint32_t)((uintptr_t)p & 15u);
        const uint32_t S = ((((n + 15u) + 63u) / 64u) + 15u) & ~15u;
        uint32_t lo = lane == 0 ? 0u : min(n, (uint32_t)lane * S - a0);
        const uint32_t hi = min(n, ((uint32_t)lane + 1u) * S - a0);
        uint32_t c = lane == 0 ? 0xFFFFFFFFu : 0u;
        while (lo < hi && (((uintptr_t)(p + lo)) & 15u)) { c = tab[0][(c ^ p[lo]) & 0xFFu] ^ (c >> 8); lo++; }
        for (; lo + 16u <= hi; lo += 16u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + lo);
            c = step4(c, v.x); c = step4(c, v.y); c = step4(c, v.z); c = step4(c, v.w);
        }
        for (; lo < hi; lo++) c = tab[0][(c ^ p[lo]) & 0xFFu] ^ (c >> 8);
        uint32_t m = n - hi; // zero bytes behind this slice
        for (int j = 0; m; j++, m >>= 1)
            if (m & 1u) c = gf2_times(mat[3 + j], c);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c ^= __shfl_xor(c, o, 64);
        if (lane == 0 && (c ^ 0xFFFFFFFFu) != want[b]) atomicCAS(err, 0, (int32_t)(b + 1 > 0x7FFFFFFF ? 0x7FFFFFFF : b + 1));
    }
}
} // namespace

// out / out_off / want / err: device memory; *err stays 0 when every block's CRC-32 is the one its footer carries
void uz_launch_crc32(uz_ctx *c, hipStream_t st, int64_t n_blocks, const uint8_t *out, const int64_t *out_off, const uint32_t *want, int32_t *err) {
    if (n_blocks <= 0) return;
    UZ_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_bgzf_crc32, dim3((unsigned)std::min<int64_t>(n_blocks, 256 * 16)), dim3(64), 0, st, n_blocks, out, out_off, want, err);
    UZ_HIP(hipGetLastError());
}


// comp / in_off / out_off / out: device memory (comp padded by 1 KiB past its last byte); out_off[n_blocks] = total bytes
void uz_launch_inflate(uz_ctx *c, hipStream_t st, int64_t n_blocks, const uint8_t *comp, int64_t comp_bytes_padded, const int64_t *in_off,
                       const int64_t *out_off, uint8_t *out, int32_t *cursor_and_err /* [2], zeroed here */) {
    if (n_blocks <= 0) return;
    UZ_HIP(hipMemsetAsync(cursor_and_err, 0, 2 * sizeof(int32_t), st));
    static const int cus = [&] { // (asked once: the query is not cheap; one kind of device per process)
        hipDeviceProp_t prop;
        return hipGetDeviceProperties(&prop, c->device) == hipSuccess ? prop.multiProcessorCount : 256;
    }();
    const int64_t grid = std::min<int64_t>(n_blocks, (int64_t)cus * 32); // every wave slot of the chip: a wave is one block's decoder
    hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)grid), dim3(64), 0, st, n_blocks, comp, in_off, out_off, out, comp_bytes_padded / 4, cursor_and_err, cursor_and_err + 1);
    UZ_HIP(hipGetLastError());
}
