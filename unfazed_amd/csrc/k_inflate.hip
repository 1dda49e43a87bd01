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
    uint16_t lit_tab[1 << UZI_LIT_BITS];   // the literal / length code's direct table (build_table: form 1); before that, the code-length code's (form 0)
    uint32_t dist_tab[1 << UZI_DIST_BITS]; // the distance code's direct table (form 2)
    uint16_t lit_sorted[288], dist_sorted[32]; // symbols in canonical order (by length, then value)
    uint16_t lit_cnt[16], dist_cnt[16];        // codes per length
    uint8_t lens[320];                          // code lengths of the block being set up: literal / length code at 0, distance code at 288
    uint8_t cl_lens[32];                        // ... and of the code they are written in
    uint16_t code_of[288];                      // bit-reversed canonical code of every symbol (table fill); form 1: afterwards every symbol's table entry
    int work[48];                               // per-length counters of the table set-up
    int lit_walk[2];                            // form 1: the canonical walk's `first` and `index` after length UZI_LIT_BITS (where a longer code's walk starts)
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
// every lane holds the same decoder state; a value read from LDS or memory is uniform too, but the compiler cannot know: uni() says so (the value
// moves to a scalar register, what depends on it is scalar arithmetic and scalar branches instead of lane masks)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
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
// form 0: 16-bit entries (symbol << 4) | code length, 0 for a longer code (the code-length code).
// form 1: the literal / length code as the symbol loop reads it -- a literal is (byte << 4) | code length; a length symbol
// 0x8000 | extra bits << 12 | (base length - 3) << 4 | code length (RFC 1951's length table folded into the entry: 3 ... 258 minus 3 fits eight
// bits); end of block and the two symbols that stand for nothing: "7 extra bits" with value 0 / 1; a code longer than the table's index: 0x8000.
// code_of[] holds every symbol's entry afterwards (what a longer code's walk ends in), walk[] where that walk starts.
// form 2: the distance code, 32-bit entries base distance << 8 | extra bits << 4 | code length (RFC 1951's distance table folded in); 0 for a
// longer code and for the two symbols that stand for nothing.
__device__ bool build_table(const uint8_t *lens, int n, void *tab_, int tb, uint16_t *sorted, uint16_t *cnt, uint16_t *code_of, int *work, int lane,
                            int complete, int form = 0, int *walk = nullptr) {
    uint16_t *tab = static_cast<uint16_t *>(tab_);
    uint32_t *tab32 = static_cast<uint32_t *>(tab_);
    int *count = work, *offs = work + 16, *next = work + 32;
    __syncthreads();
    if (lane < 16) count[lane] = 0;
    __syncthreads();
    if (lane == 0) {
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        offs[1] = 0; next[1] = 0;
        for (int l = 1; l < 15; l++) { offs[l + 1] = offs[l] + count[l]; next[l + 1] = (next[l] + count[l]) << 1; }
        if (walk) { // decode_long's `first` and `index` after its iteration l = tb
            int first = 0, index = 0;
            for (int l = 1; l <= tb; l++) { index += count[l]; first += count[l]; first <<= 1; }
            walk[0] = first; walk[1] = index;
        }
    }
    __syncthreads();
    int left = 1, total = 0;
    for (int l = 1; l < 16; l++) {
        const int c = uni(count[l]);
        left = (left << 1) - c;
        total += c;
        if (left < 0) return false;
    }
    if (left > 0 && (complete == 2 || (complete == 1 && !(total == 1 && uni(count[1]) == 1)))) return false;
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
    if (form == 2) for (int k = lane; k < (1 << tb); k += 64) tab32[k] = 0;
    else for (int k = lane; k < (1 << tb); k += 64) tab[k] = form == 1 ? (uint16_t)0x8000 : (uint16_t)0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) { // every lane fills the replicas of its symbols' codes
        const int l = lens[s];
        if (!l) continue;
        uint32_t e = (uint32_t)((s << 4) | l);
        if (form == 1 && s >= 256) {
            const int ls = s - 257;
            int base3 = 0, extra = 7; // end of block (value 0) / a symbol that stands for nothing (286, 287: value 1)
            if (s > 285) base3 = 1;
            else if (s > 256) {
                if (ls < 8) { base3 = ls; extra = 0; }
                else if (ls == 28) { base3 = 255; extra = 0; }
                else { extra = (ls >> 2) - 1; base3 = (4 + (ls & 3)) << extra; }
            }
            e = (uint32_t)(0x8000 | (extra << 12) | (base3 << 4) | l);
        } else if (form == 2) {
            if (s > 29) e = 0;
            else if (s < 4) e = (uint32_t)(((1 + s) << 8) | l);
            else { const int x = (s >> 1) - 1; e = (uint32_t)(((((2 + (s & 1)) << x) + 1) << 8) | (x << 4) | l); }
        }
        if (l <= tb) {
            if (form == 2) { for (int k = code_of[s]; k < (1 << tb); k += 1 << l) tab32[k] = e; }
            else for (int k = code_of[s]; k < (1 << tb); k += 1 << l) tab[k] = (uint16_t)e;
        }
        if (form == 1) code_of[s] = (uint16_t)e; // (this lane's own symbol: its code has just been read)
    }
    __syncthreads();
    return true;
}
// one symbol the canonical way (count / first code per length): what the C++ around the hand-written loops decodes with -- codes of any length
__device__ __forceinline__ int decode_long(BitIn &b, const uint16_t *sorted, const uint16_t *cnt) {
    int code = 0, first = 0, index = 0;
    unsigned long long v = b.buf;
    for (int l = 1; l <= 15; l++) {
        code |= (int)(v & 1ULL);
        v >>= 1;
        const int c = uni((int)cnt[l]);
        if (code - c < first) { bi_drop(b, l); return uni((int)sorted[index + (code - first)]); }
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
        const int64_t blk = uni(next_block);
        if (blk >= n_blocks) return;
        uint8_t *o = out + out_off[blk];
        const int64_t osize64 = out_off[blk + 1] - out_off[blk];
        const uint32_t osize = (uint32_t)(osize64 < 0 ? 0 : (osize64 > (1 << 30) ? (1 << 30) : osize64)); // (a BGZF block holds at most 64 KiB: positions are 32-bit)
        BitIn b;
        bi_open(b, comp, in_off[blk], comp_words, lane);
        uint32_t pos = 0, safe = 0; // bytes written; bytes whose stores are known to have landed
        uint32_t wbase = 0;         // first position of the literal window
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
                if ((len ^ nlen) != 0xFFFFu || pos + len > osize) { bad = 1; break; }
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
                // (the code-length code sits where the other two codes will be: 7-bit direct table in the literal table's place, the distance code's canonical arrays)
                if (!build_table(L.cl_lens, 19, L.lit_tab, 7, L.dist_sorted, L.dist_cnt, L.code_of, L.work, lane, 2)) { bad = 2; break; }
                // the code lengths of both codes, written out by hand like the symbol loop below (and for the same reason: ~500 of these symbols per
                // BGZF block at ~80 scalar instructions each in the compiler's loop were a fifth of the kernel's scalar issue; here ~17).  A symbol below
                // 16 is a length, 16 repeats the one before 3 - 6 times, 17 / 18 write 3 - 10 / 11 - 138 zeros; entry n + k is a literal / length code's
                // length while n + k < nlit, a distance code's behind it (L.lens + 288).  All lanes store (the lanes past the run's end its last entry again).
                // why: 0 done; 1 bad lengths; 2 the next input word lies beyond the 64 the wave holds
                {
                    uint32_t hn = 0, hprev = 0, hwhy = 0;
                    const uint32_t htotal = (uint32_t)(nlit + ndist), hnlit = (uint32_t)nlit, hadj = 288u - (uint32_t)nlit;
                    const uint32_t lds_base = (uint32_t)(uintptr_t)&L;
                    for (;;) {
                        if (b.bits <= 32) {
                            if (b.widx > UZI_MAX_WORDS) { bad = 6; break; }
                            bi_refill(b, lane);
                        }
                        uint32_t e, t_val, t_rep, t_x, t_t, va, ve, vv;
                        unsigned long long buf = b.buf;
                        int bits = b.bits, widx = b.widx;
                        hwhy = 0;
                        hn = uni(hn); hprev = uni(hprev); // (loop-carried through nothing but the statement below: the compiler would keep them in vector registers)
                        asm volatile(
                            ".Luzh_top%=:\n"
                            "  v_bfe_u32 %[va], s44, 0, 7\n"
                            "  v_lshl_add_u32 %[va], %[va], 1, %[vL]\n"
                            "  ds_read_u16 %[ve], %[va] offset:%[o_lit]\n"
                            "  s_waitcnt lgkmcnt(0)\n"
                            "  v_readfirstlane_b32 %[e], %[ve]\n"
                            "  s_and_b32 %[t], %[e], 15\n"
                            "  s_cmp_eq_u32 %[t], 0\n"
                            "  s_cbranch_scc1 .Luzh_bad%=\n"
                            "  s_lshr_b64 s[44:45], s[44:45], %[t]\n"
                            "  s_sub_u32 %[bits], %[bits], %[t]\n"
                            "  s_lshr_b32 %[val], %[e], 4\n"
                            "  s_mov_b32 %[rep], 1\n"
                            "  s_cmpk_lt_u32 %[val], 16\n"
                            "  s_cbranch_scc1 .Luzh_store%=\n"
                            "  s_cmpk_eq_u32 %[val], 16\n"
                            "  s_cbranch_scc1 .Luzh_16%=\n"
                            "  s_cmpk_eq_u32 %[val], 17\n"
                            "  s_cselect_b32 %[x], 3, 7\n"
                            "  s_cselect_b32 %[rep], 3, 11\n"
                            "  s_mov_b32 %[val], 0\n"
                            "  s_branch .Luzh_ext%=\n"
                            ".Luzh_16%=:\n"
                            "  s_cmp_eq_u32 %[n], 0\n"
                            "  s_cbranch_scc1 .Luzh_bad%=\n"
                            "  s_mov_b32 %[val], %[prev]\n"
                            "  s_mov_b32 %[x], 2\n"
                            "  s_mov_b32 %[rep], 3\n"
                            ".Luzh_ext%=:\n"
                            "  s_bfm_b32 %[t], %[x], 0\n"
                            "  s_and_b32 %[t], s44, %[t]\n"
                            "  s_add_u32 %[rep], %[rep], %[t]\n"
                            "  s_lshr_b64 s[44:45], s[44:45], %[x]\n"
                            "  s_sub_u32 %[bits], %[bits], %[x]\n"
                            ".Luzh_store%=:\n"
                            "  s_add_u32 %[t], %[n], %[rep]\n"
                            "  s_cmp_gt_u32 %[t], %[total]\n"
                            "  s_cbranch_scc1 .Luzh_bad%=\n"
                            "  s_mov_b32 %[prev], %[val]\n"
                            "  s_sub_u32 %[x], %[rep], 1\n"
                            "  s_mov_b32 %[e], 0\n"
                            "  v_mov_b32 %[vv], %[val]\n"
                            ".Luzh_w%=:\n"
                            "  v_add_u32 %[va], %[e], %[vlane]\n"
                            "  v_min_u32 %[va], %[x], %[va]\n"
                            "  v_add_u32 %[va], %[n], %[va]\n"
                            "  v_cmp_gt_u32 vcc, %[nlit], %[va]\n"
                            "  v_add_u32 %[ve], %[adj], %[va]\n"
                            "  v_cndmask_b32 %[va], %[ve], %[va], vcc\n"
                            "  v_add_u32 %[va], %[va], %[vL]\n"
                            "  ds_write_b8 %[va], %[vv] offset:%[o_lens]\n"
                            "  s_add_u32 %[e], %[e], 64\n"
                            "  s_cmp_lt_u32 %[e], %[rep]\n"
                            "  s_cbranch_scc1 .Luzh_w%=\n"
                            "  s_mov_b32 %[n], %[t]\n"
                            "  s_cmp_lt_u32 %[n], %[total]\n"
                            "  s_cbranch_scc0 .Luzh_out%=\n"
                            "  s_cmpk_gt_i32 %[bits], 32\n"
                            "  s_cbranch_scc1 .Luzh_top%=\n"
                            "  s_sub_u32 %[t], %[widx], %[cbase]\n"
                            "  s_cmpk_ge_u32 %[t], 64\n"
                            "  s_cbranch_scc1 .Luzh_need%=\n"
                            "  v_readlane_b32 s46, %[cur], %[t]\n"
                            "  s_mov_b32 s47, 0\n"
                            "  s_lshl_b64 s[46:47], s[46:47], %[bits]\n"
                            "  s_or_b64 s[44:45], s[44:45], s[46:47]\n"
                            "  s_add_u32 %[bits], %[bits], 32\n"
                            "  s_add_u32 %[widx], %[widx], 1\n"
                            "  s_branch .Luzh_top%=\n"
                            ".Luzh_bad%=:\n"
                            "  s_mov_b32 %[why], 1\n"
                            "  s_branch .Luzh_out%=\n"
                            ".Luzh_need%=:\n"
                            "  s_mov_b32 %[why], 2\n"
                            ".Luzh_out%=:\n"
                            : "+{s[44:45]}"(buf), [bits] "+s"(bits), [widx] "+s"(widx), [n] "+s"(hn), [prev] "+s"(hprev), [why] "+s"(hwhy), [e] "=&s"(e),
                              [val] "=&s"(t_val), [rep] "=&s"(t_rep), [x] "=&s"(t_x), [t] "=&s"(t_t), [va] "=&v"(va), [ve] "=&v"(ve), [vv] "=&v"(vv)
                            : [vL] "v"(lds_base), [cbase] "s"(b.cbase), [cur] "v"(b.cur), [vlane] "v"(lane), [total] "s"(htotal), [nlit] "s"(hnlit), [adj] "s"(hadj),
                              [o_lit] "i"(offsetof(InflateLds, lit_tab)), [o_lens] "i"(offsetof(InflateLds, lens))
                            : "s46", "s47", "scc", "vcc", "memory");
                        b.buf = buf; b.bits = bits; b.widx = widx;
                        if (hwhy != 2) break;
                    }
                    if (!bad && hwhy == 1) bad = 2;
                }
                if (bad) break;
                __syncthreads();
                if (uni((int)L.lens[256]) == 0) { bad = 2; break; } // no end-of-block code
            }
            // (the distance code first: the literal / length code's build leaves every symbol's entry in code_of[])
            if (!build_table(L.lens + 288, ndist, L.dist_tab, UZI_DIST_BITS, L.dist_sorted, L.dist_cnt, L.code_of, L.work, lane, 0, 2)) { bad = 2; break; }
            if (!build_table(L.lens, nlit, L.lit_tab, UZI_LIT_BITS, L.lit_sorted, L.lit_cnt, L.code_of, L.work, lane, 1, 1, L.lit_walk)) { bad = 2; break; }
            // ---- the symbols of the block.  Literals gather in a register window -- lane l holds the byte for position wbase + l -- and
            // leave it as one coalesced store when it is full or a match needs them in memory.
            //
            // The loop is written out by hand (one asm statement).  Why: every lane holds the same decoder state, so the state lives in scalar
            // registers and the kernel is bound by SCALAR ISSUE -- a SIMD issues one scalar instruction per four cycles whatever its eight waves
            // do, and the compiler's loop (exits unified, conditions kept as lane masks, 64-bit positions, four refill tests per match) spent ~20
            // scalar instructions per literal and ~140 per match: 400 k per 64 KiB block = the 12.7 M cycles a wave lived (rocprofv3, round 5; the
            // generator's BAM is 2.1 k literals + 2.4 k matches per block).  Here: 7 scalar + 6 vector instructions per literal, ~45 + 12 per match
            // whose codes both sit in the direct tables, whose source does not overlap its destination and which is at most 64 bytes long (nine in
            // ten); everything else -- longer codes, long or overlapping matches, the input window moving on, errors -- leaves the statement with a
            // reason (`why`) and whatever it has decoded so far, and is finished by the C++ below.
            //   why: 1 window full; 2 the next input word lies beyond the 64 the wave holds; 3 end of block (consumed); 4 the symbol at hand is not
            //   for the fast path (nothing consumed: a symbol that stands for nothing, a code no symbol has); 5 a match leaves the block (bad 4);
            //   6 length decoded, distance still to come (its code is longer than the table's index, or the input window has to move first)
            // Wait states (gfx940 family): a scalar register a lane select reads (k, wcnt - 1) is written by a scalar instruction; two instructions stand
            // between v_readlane writing a scalar register and the vector instruction that reads it.  (v_writelane cannot take the byte AND the lane from
            // scalar registers -- one constant-bus read per instruction -- so a literal enters the window through v_cmp_eq + v_cndmask.)
            uint32_t wcnt = 0; // literals waiting in the window: pos == wbase + wcnt
            const uint32_t lds_base = (uint32_t)(uintptr_t)&L; // (the tables' address in LDS; the fields are reached through offset: fields of the ds_ instructions)
            for (;;) {
                if (b.bits <= 32) {
                    if (b.widx > UZI_MAX_WORDS) { bad = 6; break; }
                    bi_refill(b, lane);
                }
                uint32_t e, why = 0, len_a, dist_a, t_n, t_x, t_t, va, ve, vr, vt, vp0, vp1;
                {
                    unsigned long long buf = b.buf;
                    int bits = b.bits, widx = b.widx;
                    asm volatile(
                        ".Luz_top%=:\n"
                        "  v_bfe_u32 %[va], s44, 0, 10\n"
                        "  v_lshl_add_u32 %[va], %[va], 1, %[vL]\n"
                        "  ds_read_u16 %[ve], %[va] offset:%[o_lit]\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_readfirstlane_b32 %[e], %[ve]\n"
                        ".Luz_entry%=:\n"
                        "  s_bitcmp1_b32 %[e], 15\n"
                        "  s_cbranch_scc1 .Luz_nonlit%=\n"
                        // a literal: into lane wcnt of the window
                        "  s_and_b32 %[n], %[e], 15\n"
                        "  v_bfe_u32 %[ve], %[ve], 4, 8\n"
                        "  v_cmp_eq_u32 vcc, %[wc], %[vlane]\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[n]\n"
                        "  v_cndmask_b32 %[wb], %[wb], %[ve], vcc\n"
                        "  s_add_u32 %[wc], %[wc], 1\n"
                        "  s_sub_u32 %[bits], %[bits], %[n]\n"
                        "  s_cmpk_eq_u32 %[wc], 64\n"
                        "  s_cbranch_scc1 .Luz_full%=\n"
                        // the end of a symbol: at least 33 bits for the next one
                        ".Luz_next%=:\n"
                        "  s_cmpk_gt_i32 %[bits], 32\n"
                        "  s_cbranch_scc1 .Luz_top%=\n"
                        "  s_sub_u32 %[n], %[widx], %[cbase]\n"
                        "  s_cmpk_ge_u32 %[n], 64\n"
                        "  s_cbranch_scc1 .Luz_need%=\n"
                        "  v_readlane_b32 s46, %[cur], %[n]\n"
                        "  s_mov_b32 s47, 0\n"
                        "  s_lshl_b64 s[46:47], s[46:47], %[bits]\n"
                        "  s_or_b64 s[44:45], s[44:45], s[46:47]\n"
                        "  s_add_u32 %[bits], %[bits], 32\n"
                        "  s_add_u32 %[widx], %[widx], 1\n"
                        "  s_branch .Luz_top%=\n"
                        // not a literal: a length, the end of the block, or a code the table does not hold
                        ".Luz_nonlit%=:\n"
                        "  s_and_b32 %[n], %[e], 15\n"
                        "  s_cmp_eq_u32 %[n], 0\n"
                        "  s_cbranch_scc1 .Luz_long%=\n"
                        "  s_bfe_u32 %[x], %[e], 0x3000c\n"
                        "  s_bfe_u32 %[len], %[e], 0x80004\n"
                        "  s_cmpk_eq_u32 %[x], 7\n"
                        "  s_cbranch_scc1 .Luz_special%=\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[n]\n"
                        "  s_sub_u32 %[bits], %[bits], %[n]\n"
                        "  s_bfm_b32 %[t], %[x], 0\n"
                        "  s_and_b32 %[t], s44, %[t]\n"
                        "  s_add_u32 %[len], %[len], %[t]\n"
                        "  s_add_u32 %[len], %[len], 3\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[x]\n"
                        "  s_sub_u32 %[bits], %[bits], %[x]\n"
                        // the distance: up to 15 + 13 bits
                        "  s_cmpk_gt_i32 %[bits], 32\n"
                        "  s_cbranch_scc1 .Luz_dist%=\n"
                        "  s_sub_u32 %[n], %[widx], %[cbase]\n"
                        "  s_cmpk_ge_u32 %[n], 64\n"
                        "  s_cbranch_scc1 .Luz_havelen%=\n"
                        "  v_readlane_b32 s46, %[cur], %[n]\n"
                        "  s_mov_b32 s47, 0\n"
                        "  s_lshl_b64 s[46:47], s[46:47], %[bits]\n"
                        "  s_or_b64 s[44:45], s[44:45], s[46:47]\n"
                        "  s_add_u32 %[bits], %[bits], 32\n"
                        "  s_add_u32 %[widx], %[widx], 1\n"
                        ".Luz_dist%=:\n"
                        "  v_bfe_u32 %[va], s44, 0, 8\n"
                        "  v_lshl_add_u32 %[va], %[va], 2, %[vL]\n"
                        "  ds_read_b32 %[ve], %[va] offset:%[o_dist]\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_readfirstlane_b32 %[e], %[ve]\n"
                        "  s_and_b32 %[n], %[e], 15\n"
                        "  s_cmp_eq_u32 %[n], 0\n"
                        "  s_cbranch_scc1 .Luz_havelen%=\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[n]\n"
                        "  s_sub_u32 %[bits], %[bits], %[n]\n"
                        "  s_bfe_u32 %[x], %[e], 0x40004\n"
                        "  s_lshr_b32 %[dist], %[e], 8\n"
                        "  s_bfm_b32 %[t], %[x], 0\n"
                        "  s_and_b32 %[t], s44, %[t]\n"
                        "  s_add_u32 %[dist], %[dist], %[t]\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[x]\n"
                        "  s_sub_u32 %[bits], %[bits], %[x]\n"
                        // the window's literals into memory in front of the match (no lane masked off: the lanes behind the fill store its last byte again)
                        "  s_cmp_eq_u32 %[wc], 0\n"
                        "  s_cbranch_scc1 .Luz_noflush%=\n"
                        "  s_add_u32 %[t], %[wbase], %[wc]\n"
                        "  s_cmp_gt_u32 %[t], %[osize]\n"
                        "  s_cbranch_scc1 .Luz_bad4%=\n"
                        "  s_sub_u32 %[n], %[wc], 1\n"
                        "  v_readlane_b32 %[x], %[wb], %[n]\n"
                        "  v_min_u32 %[va], %[n], %[vlane]\n"
                        "  v_cmp_gt_u32 vcc, %[wc], %[vlane]\n"
                        "  v_mov_b32 %[ve], %[x]\n"
                        "  v_cndmask_b32 %[ve], %[ve], %[wb], vcc\n"
                        "  v_add_u32 %[va], %[wbase], %[va]\n"
                        "  global_store_byte %[va], %[ve], s[48:49]\n"
                        "  s_mov_b32 %[wbase], %[t]\n"
                        "  s_mov_b32 %[wc], 0\n"
                        ".Luz_noflush%=:\n"
                        "  s_cmp_gt_u32 %[dist], %[wbase]\n"
                        "  s_cbranch_scc1 .Luz_bad4%=\n"
                        "  s_add_u32 %[t], %[wbase], %[len]\n"
                        "  s_cmp_gt_u32 %[t], %[osize]\n"
                        "  s_cbranch_scc1 .Luz_bad4%=\n"
                        // the copy.  Lane k takes byte k of a 64-byte piece (the lanes past the match's end repeat its last byte); the loads go past the
                        // L1 and wait for the wave's stores only when the source reaches into bytes stored since the last wait -- and every piece waits
                        // for its own load before it stores, so everything in front of the match has landed by then: `safe` moves up to its start
                        "  s_sub_u32 %[n], %[wbase], %[dist]\n"
                        "  s_min_u32 %[x], %[len], %[dist]\n"
                        "  s_add_u32 %[x], %[n], %[x]\n"
                        "  s_cmp_le_u32 %[x], %[safe]\n"
                        "  s_cbranch_scc1 .Luz_safe%=\n"
                        "  s_waitcnt vmcnt(0)\n"
                        ".Luz_safe%=:\n"
                        "  s_mov_b32 %[safe], %[wbase]\n"
                        "  s_sub_u32 %[x], %[len], 1\n"
                        "  s_mov_b32 %[e], 0\n"
                        "  s_cmp_lt_u32 %[dist], %[len]\n"
                        "  s_cbranch_scc1 .Luz_mod%=\n"
                        ".Luz_plain%=:\n"
                        "  v_add_u32 %[va], %[e], %[vlane]\n"
                        "  v_min_u32 %[va], %[x], %[va]\n"
                        "  v_add_u32 %[ve], %[n], %[va]\n"
                        "  global_load_ubyte %[ve], %[ve], s[48:49] sc1\n"
                        "  v_add_u32 %[va], %[wbase], %[va]\n"
                        "  s_add_u32 %[e], %[e], 64\n"
                        "  s_waitcnt vmcnt(0)\n"
                        "  global_store_byte %[va], %[ve], s[48:49]\n"
                        "  s_cmp_lt_u32 %[e], %[len]\n"
                        "  s_cbranch_scc1 .Luz_plain%=\n"
                        "  s_mov_b32 %[wbase], %[t]\n"
                        "  s_branch .Luz_next%=\n"
                        // source and destination overlap (a run): byte k comes from k mod dist -- through a float reciprocal, exact for k < 512 after one
                        // correction either way
                        ".Luz_mod%=:\n"
                        "  v_cvt_f32_u32 %[vr], %[dist]\n"
                        "  v_rcp_f32 %[vr], %[vr]\n"
                        ".Luz_modl%=:\n"
                        "  v_add_u32 %[va], %[e], %[vlane]\n"
                        "  v_min_u32 %[va], %[x], %[va]\n"
                        "  v_cvt_f32_u32 %[ve], %[va]\n"
                        "  v_mul_f32 %[ve], %[ve], %[vr]\n"
                        "  v_cvt_u32_f32 %[ve], %[ve]\n"
                        "  v_mul_u32_u24 %[ve], %[dist], %[ve]\n"
                        "  v_sub_u32 %[ve], %[va], %[ve]\n"
                        "  v_cmp_gt_i32 vcc, 0, %[ve]\n"
                        "  v_add_u32 %[vt], %[dist], %[ve]\n"
                        "  v_cndmask_b32 %[ve], %[ve], %[vt], vcc\n"
                        "  v_cmp_le_u32 vcc, %[dist], %[ve]\n"
                        "  v_subrev_u32 %[vt], %[dist], %[ve]\n"
                        "  v_cndmask_b32 %[ve], %[ve], %[vt], vcc\n"
                        "  v_add_u32 %[ve], %[n], %[ve]\n"
                        "  global_load_ubyte %[ve], %[ve], s[48:49] sc1\n"
                        "  v_add_u32 %[va], %[wbase], %[va]\n"
                        "  s_add_u32 %[e], %[e], 64\n"
                        "  s_waitcnt vmcnt(0)\n"
                        "  global_store_byte %[va], %[ve], s[48:49]\n"
                        "  s_cmp_lt_u32 %[e], %[len]\n"
                        "  s_cbranch_scc1 .Luz_modl%=\n"
                        "  s_mov_b32 %[wbase], %[t]\n"
                        "  s_branch .Luz_next%=\n"
                        // a code longer than the table's index: the canonical walk (decode_long) from length 11 on -- its state after length 10 was worked out
                        // when the table was built -- ends at the symbol's table entry (code_of[]), and the symbol goes the way of all the others
                        ".Luz_long%=:\n"
                        "  ds_read_b32 %[vp0], %[vL] offset:%[o_walk]\n"
                        "  ds_read_b32 %[vp1], %[vL] offset:%[o_walk1]\n"
                        "  s_brev_b32 %[t], s44\n"
                        "  s_lshr_b32 %[t], %[t], 22\n"
                        "  s_lshl_b32 %[t], %[t], 1\n"
                        "  s_mov_b32 %[n], 10\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_readfirstlane_b32 %[x], %[vp0]\n"
                        "  v_readfirstlane_b32 %[len], %[vp1]\n"
                        ".Luz_walk%=:\n"
                        "  s_lshr_b32 %[dist], s44, %[n]\n"
                        "  s_and_b32 %[dist], %[dist], 1\n"
                        "  s_or_b32 %[t], %[t], %[dist]\n"
                        "  s_add_u32 %[n], %[n], 1\n"
                        "  v_mov_b32 %[va], %[n]\n"
                        "  v_lshl_add_u32 %[va], %[va], 1, %[vL]\n"
                        "  ds_read_u16 %[ve], %[va] offset:%[o_cnt]\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_readfirstlane_b32 %[dist], %[ve]\n"
                        "  s_sub_i32 %[e], %[t], %[dist]\n"
                        "  s_cmp_lt_i32 %[e], %[x]\n"
                        "  s_cbranch_scc1 .Luz_found%=\n"
                        "  s_add_u32 %[len], %[len], %[dist]\n"
                        "  s_add_u32 %[x], %[x], %[dist]\n"
                        "  s_lshl_b32 %[x], %[x], 1\n"
                        "  s_lshl_b32 %[t], %[t], 1\n"
                        "  s_cmpk_lt_u32 %[n], 15\n"
                        "  s_cbranch_scc1 .Luz_walk%=\n"
                        "  s_branch .Luz_generic%=\n"
                        ".Luz_found%=:\n"
                        "  s_sub_u32 %[t], %[t], %[x]\n"
                        "  s_add_u32 %[t], %[t], %[len]\n"
                        "  v_mov_b32 %[va], %[t]\n"
                        "  v_lshl_add_u32 %[va], %[va], 1, %[vL]\n"
                        "  ds_read_u16 %[ve], %[va] offset:%[o_sorted]\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_lshl_add_u32 %[va], %[ve], 1, %[vL]\n"
                        "  ds_read_u16 %[ve], %[va] offset:%[o_entry]\n"
                        "  s_waitcnt lgkmcnt(0)\n"
                        "  v_readfirstlane_b32 %[e], %[ve]\n"
                        "  s_branch .Luz_entry%=\n"
                        // the ways out
                        ".Luz_special%=:\n" // (value 0: the end of the block; 1: a symbol that stands for nothing -- the C++ reports it)
                        "  s_cmp_lg_u32 %[len], 0\n"
                        "  s_cbranch_scc1 .Luz_generic%=\n"
                        "  s_lshr_b64 s[44:45], s[44:45], %[n]\n"
                        "  s_sub_u32 %[bits], %[bits], %[n]\n"
                        "  s_mov_b32 %[why], 3\n"
                        "  s_branch .Luz_out%=\n"
                        ".Luz_full%=:\n"
                        "  s_mov_b32 %[why], 1\n"
                        "  s_branch .Luz_out%=\n"
                        ".Luz_need%=:\n"
                        "  s_mov_b32 %[why], 2\n"
                        "  s_branch .Luz_out%=\n"
                        ".Luz_generic%=:\n"
                        "  s_mov_b32 %[why], 4\n"
                        "  s_branch .Luz_out%=\n"
                        ".Luz_bad4%=:\n"
                        "  s_mov_b32 %[why], 5\n"
                        "  s_branch .Luz_out%=\n"
                        ".Luz_havelen%=:\n"
                        "  s_mov_b32 %[why], 6\n"
                        ".Luz_out%=:\n"
                        : "+{s[44:45]}"(buf), [bits] "+s"(bits), [widx] "+s"(widx), [wc] "+s"(wcnt), [wbase] "+s"(wbase), [safe] "+s"(safe), [wb] "+v"(wbyte),
                          [why] "+s"(why), [e] "=&s"(e), [len] "=&s"(len_a), [dist] "=&s"(dist_a), [n] "=&s"(t_n), [x] "=&s"(t_x), [t] "=&s"(t_t),
                          [va] "=&v"(va), [ve] "=&v"(ve), [vr] "=&v"(vr), [vt] "=&v"(vt), [vp0] "=&v"(vp0), [vp1] "=&v"(vp1)
                        : [vL] "v"(lds_base), [cbase] "s"(b.cbase), [cur] "v"(b.cur), [vlane] "v"(lane), [osize] "s"(osize), "{s[48:49]}"(o),
                          [o_lit] "i"(offsetof(InflateLds, lit_tab)), [o_dist] "i"(offsetof(InflateLds, dist_tab)), [o_walk] "i"(offsetof(InflateLds, lit_walk)), [o_walk1] "i"(offsetof(InflateLds, lit_walk) + 4),
                          [o_cnt] "i"(offsetof(InflateLds, lit_cnt)), [o_sorted] "i"(offsetof(InflateLds, lit_sorted)), [o_entry] "i"(offsetof(InflateLds, code_of))
                        : "s46", "s47", "scc", "vcc", "memory");
                    b.buf = buf; b.bits = bits; b.widx = widx;
                }
                if (why == 1) { // the window is full: one coalesced store
                    if (wbase + 64u > osize) { bad = 4; break; }
                    o[wbase + lane] = (uint8_t)wbyte;
                    wbase += 64; wcnt = 0;
                    continue;
                }
                if (why == 2) continue; // (the refill at the loop's head moves the input window on)
                if (why == 5) { bad = 4; break; }
                int len = (int)len_a, dist = (int)dist_a;
                if (why == 4) { // the symbol at hand the general way (the canonical walk decodes a code of any length)
                    const int s = decode_long(b, L.lit_sorted, L.lit_cnt);
                    if (s < 0) { bad = 3; break; }
                    if (s < 256) {
                        wbyte = (uint32_t)lane == wcnt ? (uint32_t)s : wbyte;
                        wcnt++;
                        if (wcnt == 64) {
                            if (wbase + 64u > osize) { bad = 4; break; }
                            o[wbase + lane] = (uint8_t)wbyte;
                            wbase += 64; wcnt = 0;
                        }
                        continue;
                    }
                    if (s == 256) why = 3;
                    else {
                        if (s > 285) { bad = 3; break; }
                        bi_refill(b, lane);
                        // (length and distance from their symbols: RFC 1951's tables are arithmetic -- four codes per extra bit / two per extra bit)
                        const int ls = s - 257;
                        if (ls < 8) len = 3 + ls;
                        else if (ls == 28) len = 258;
                        else { const int x = (ls >> 2) - 1; len = ((4 + (ls & 3)) << x) + 3 + (int)bi_take(b, x); }
                    }
                }
                if (wcnt) { // the window's bytes into memory, in front of whatever comes next.  No lane is masked off (a divergent branch here would have
                    // the compiler restructure the whole symbol loop around lane masks): the lanes behind the window's fill store its last byte again
                    if (wbase + wcnt > osize) { bad = 4; break; }
                    const uint32_t last_b = lane_word(wbyte, (int)wcnt - 1);
                    const uint32_t wl = (uint32_t)lane < wcnt ? (uint32_t)lane : wcnt - 1u;
                    o[wbase + wl] = (uint8_t)((uint32_t)lane < wcnt ? wbyte : last_b);
                    wbase += wcnt; wcnt = 0;
                }
                if (why == 3) break; // the end of the block
                const uint32_t pos = wbase;
                { // the distance (the canonical walk: the direct table's entries are made for the fast path)
                    bi_refill(b, lane);
                    const int ds = decode_long(b, L.dist_sorted, L.dist_cnt);
                    if (ds < 0 || ds > 29) { bad = 3; break; }
                    bi_refill(b, lane);
                    if (ds < 4) dist = 1 + ds;
                    else { const int x = (ds >> 1) - 1; dist = ((2 + (ds & 1)) << x) + 1 + (int)bi_take(b, x); }
                }
                if ((uint32_t)dist > pos || pos + (uint32_t)len > osize) { bad = 4; break; }
                const uint32_t src = pos - (uint32_t)dist;
                if (src + (uint32_t)(len < dist ? len : dist) > safe) { // the source reaches into bytes whose stores may still be in flight
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                    asm volatile("" ::: "memory");
                    safe = pos;
                }
                // lane k takes byte k of the match; where source and destination overlap (a run: quality strings are full of them) the source index
                // wraps at the distance -- k mod dist through a float reciprocal, exact for k < 512 after one correction (an integer division by a
                // variable is ~40 instructions).  Uniform loops over 64-byte pieces; the lanes past the match's end repeat its last byte: no lane masks
                const int lastk = len - 1;
                if (dist >= len) {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (int k0 = 0; k0 < len; k0 += 64) {
                        const int k = min(k0 + lane, lastk);
                        o[pos + k] = __hip_atomic_load(o + src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (past the L1: see the head of the file)
                    }
                } else if (dist == 1) {
                    const uint8_t v = __hip_atomic_load(o + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (int k0 = 0; k0 < len; k0 += 64) o[pos + min(k0 + lane, lastk)] = v;
                } else {
                    const float rd = 1.0f / (float)dist;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (int k0 = 0; k0 < len; k0 += 64) {
                        const int k = min(k0 + lane, lastk);
                        const int q = (int)((float)k * rd);
                        int r = k - q * dist;
                        r = r < 0 ? r + dist : (r >= dist ? r - dist : r);
                        o[pos + k] = __hip_atomic_load(o + src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                wbase = pos + (uint32_t)len;
            }
            if (bad) break;
            pos = wbase;
        }
        if (!bad && (int64_t)pos != osize64) bad = 5;
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
