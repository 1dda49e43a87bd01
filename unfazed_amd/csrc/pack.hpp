// pack.hpp -- the packed record format (include/uz_types.h, uz_reads_packed_view): BAM's 4-bit base codes and
// the one-bit "quality below the threshold" plane.  Shared by the device code, the host-side decoders and the
// CPU twin of the kernel body, so that there is one statement of the geometry.
#pragma once
#include <cstddef>
#include <cstdint>

#include "uz_types.h"

#if defined(__HIPCC__)
#define UZP_HD __host__ __device__ static inline
#else
#define UZP_HD static inline
#endif

// "=ACMGRSVTWYHKDBN" as two little-endian 64-bit constants: code c -> ASCII (what pysam's query_sequence shows)
UZP_HD uint8_t uz_nt16_ascii(uint32_t code) {
    const uint64_t lo = 0x565352474D43413DULL; // '=' 'A' 'C' 'M' 'G' 'R' 'S' 'V'  (byte 0 = '=')
    const uint64_t hi = 0x4E42444B48595754ULL;        // 'T' 'W' 'Y' 'H' 'K' 'D' 'B' 'N'
    const uint64_t w = (code & 8u) ? hi : lo;
    return (uint8_t)(w >> (8u * (code & 7u)));
}

// ASCII -> code, 0xFF for a character outside the alphabet (exact characters only: the device compares bases
// with REF / ALT characters as the reference compares strings, so nothing is folded)
UZP_HD uint8_t uz_ascii_nt16(uint8_t ch) {
    switch (ch) {
    case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3;
    case 'G': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7;
    case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11;
    case 'K': return 12; case 'D': return 13; case 'B': return 14; case 'N': return 15;
    default: return 0xFF;
    }
}

// base k of the row that starts at unit `unit`: its BAM code / its character
UZP_HD uint32_t uz_seq4_code(const uint8_t *seq4, uint32_t unit, int k) {
    const uint8_t b = seq4[(size_t)unit * UZ_SEQ4_UNIT_BYTES + (size_t)(k >> 1)];
    return (k & 1) ? (uint32_t)(b & 15u) : (uint32_t)(b >> 4);
}
UZP_HD uint8_t uz_seq4_base(const uint8_t *seq4, uint32_t unit, int k) { return uz_nt16_ascii(uz_seq4_code(seq4, unit, k)); }
// 1 iff the quality of base k is below the threshold the plane was built with
UZP_HD uint32_t uz_qlow_bit(const uint8_t *qlow, uint32_t unit, int k) {
    return (uint32_t)(qlow[(size_t)unit * UZ_QLOW_UNIT_BYTES + (size_t)(k >> 3)] >> (k & 7)) & 1u;
}

// end of an alignment as htslib's bam_endpos gives it: what a BAM decoder writes into the `end` column, and what the device
// derives when the staged form leaves the column out
UZP_HD int32_t uz_bam_endpos(int32_t start, uint32_t flag, uint32_t n_cigar, const uint32_t *cigar) {
    if ((flag & 4u) || n_cigar == 0) return start + 1;
    int64_t rl = 0;
    for (uint32_t k = 0; k < n_cigar; k++) {
        const uint32_t op = cigar[k] & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int64_t)(cigar[k] >> 4); // M D N = X
    }
    return (int32_t)(start + (rl > 0 ? rl : 1));
}

// a CIGAR of one M / = / X operation spanning the read -> 1 / 2 / 3 (UZ_AUX_SIMPLE_*), 0 otherwise; and back to its word
UZP_HD uint32_t uz_cigar_simple_code(uint32_t n_cigar, uint32_t word, uint32_t l_seq) {
    if (n_cigar != 1 || (word >> 4) != l_seq) return 0;
    const uint32_t op = word & 15u;
    return op == 0 ? 1u : (op == 7 ? 2u : (op == 8 ? 3u : 0u));
}
UZP_HD uint32_t uz_cigar_simple_word(uint32_t code, uint32_t l_seq) { return (l_seq << 4) | (code == 1 ? 0u : (code == 2 ? 7u : 8u)); }

// ---- the two-bit rows of the host link (uz_reads_packed_view.seq2)
// A 0, C 1, G 2, T 3; 0xFF for any other character
UZP_HD uint8_t uz_ascii_seq2(uint8_t ch) {
    switch (ch) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    default: return 0xFF;
    }
}
// one byte of a seq2 row (four bases, first base in bits 7-6) -> the two bytes of the seq4 row that hold them
// (little-endian 16 bits: low byte = bases 0 and 1).  Code c becomes the BAM code 1 << c (A 1, C 2, G 4, T 8).
UZP_HD uint32_t uz_seq2_expand_byte(uint32_t v) {
    const uint32_t n0 = 1u << ((v >> 6) & 3u), n1 = 1u << ((v >> 4) & 3u), n2 = 1u << ((v >> 2) & 3u), n3 = 1u << (v & 3u);
    return ((n0 << 4) | n1) | (((n2 << 4) | n3) << 8);
}
// base k of the seq2 row that starts at unit `unit` (ASCII; a listed exception is NOT applied here)
UZP_HD uint8_t uz_seq2_base(const uint8_t *seq2, uint32_t unit, int k) {
    const uint8_t b = seq2[(size_t)unit * UZ_SEQ2_UNIT_BYTES + (size_t)(k >> 2)];
    return uz_nt16_ascii(1u << ((b >> (6 - 2 * (k & 3))) & 3u));
}

// Host-side packing of one record's rows (decoders, tests, the CPU twin).  seq: ASCII, qual: bytes; dst rows
// hold UZ_ROW_UNITS(l_seq) units and are fully written (padding zero).  Returns 0, or -1 for a character
// outside the alphabet.
static inline int uz_pack_rows_host(const uint8_t *seq, const uint8_t *qual, int l_seq, int min_base_qual, uint8_t *seq4_row,
                                    uint8_t *qlow_row) {
    const uint32_t units = UZ_ROW_UNITS(l_seq);
    for (uint32_t b = 0; b < units * UZ_SEQ4_UNIT_BYTES; b++) seq4_row[b] = 0;
    for (uint32_t b = 0; qlow_row && b < units * UZ_QLOW_UNIT_BYTES; b++) qlow_row[b] = 0;
    int bad = 0;
    for (int k = 0; k < l_seq; k++) {
        const uint8_t c = uz_ascii_nt16(seq[k]);
        if (c == 0xFF) bad = -1;
        seq4_row[k >> 1] |= (uint8_t)((c & 15u) << ((k & 1) ? 0 : 4));
        if (qlow_row && (int)qual[k] < min_base_qual) qlow_row[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
    return bad;
}

// The same with two-bit base rows: bases other than A/C/G/T are handed to `exc(pos, code)` (code = BAM 4-bit) and stored as 0.
// Returns 0, or -1 for a character outside BAM's alphabet.
template <typename F>
static inline int uz_pack_rows_host2(const uint8_t *seq, const uint8_t *qual, int l_seq, int min_base_qual, uint8_t *seq2_row,
                                     uint8_t *qlow_row, F &&exc) {
    const uint32_t units = UZ_ROW_UNITS(l_seq);
    for (uint32_t b = 0; b < units * UZ_SEQ2_UNIT_BYTES; b++) seq2_row[b] = 0;
    for (uint32_t b = 0; qlow_row && b < units * UZ_QLOW_UNIT_BYTES; b++) qlow_row[b] = 0;
    int bad = 0;
    for (int k = 0; k < l_seq; k++) {
        const uint8_t c2 = uz_ascii_seq2(seq[k]);
        if (c2 != 0xFF) seq2_row[k >> 2] |= (uint8_t)(c2 << (6 - 2 * (k & 3)));
        else {
            const uint8_t c4 = uz_ascii_nt16(seq[k]);
            if (c4 == 0xFF) bad = -1; else exc(k, c4);
        }
        if (qlow_row && (int)qual[k] < min_base_qual) qlow_row[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
    return bad;
}
