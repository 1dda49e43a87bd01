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

// base k of the row that starts at unit `unit` (ASCII)
UZP_HD uint8_t uz_seq4_base(const uint8_t *seq4, uint32_t unit, int k) {
    const uint8_t b = seq4[(size_t)unit * UZ_SEQ4_UNIT_BYTES + (size_t)(k >> 1)];
    return uz_nt16_ascii((k & 1) ? (uint32_t)(b & 15u) : (uint32_t)(b >> 4));
}
// 1 iff the quality of base k is below the threshold the plane was built with
UZP_HD uint32_t uz_qlow_bit(const uint8_t *qlow, uint32_t unit, int k) {
    return (uint32_t)(qlow[(size_t)unit * UZ_QLOW_UNIT_BYTES + (size_t)(k >> 3)] >> (k & 7)) & 1u;
}

// Host-side packing of one record's rows (decoders, tests, the CPU twin).  seq: ASCII, qual: bytes; dst rows
// hold UZ_ROW_UNITS(l_seq) units and are fully written (padding zero).  Returns 0, or -1 for a character
// outside the alphabet.
static inline int uz_pack_rows_host(const uint8_t *seq, const uint8_t *qual, int l_seq, int min_base_qual, uint8_t *seq4_row,
                                    uint8_t *qlow_row) {
    const uint32_t units = UZ_ROW_UNITS(l_seq);
    for (uint32_t b = 0; b < units * UZ_SEQ4_UNIT_BYTES; b++) seq4_row[b] = 0;
    for (uint32_t b = 0; b < units * UZ_QLOW_UNIT_BYTES; b++) qlow_row[b] = 0;
    int bad = 0;
    for (int k = 0; k < l_seq; k++) {
        const uint8_t c = uz_ascii_nt16(seq[k]);
        if (c == 0xFF) bad = -1;
        seq4_row[k >> 1] |= (uint8_t)((c & 15u) << ((k & 1) ? 0 : 4));
        if ((int)qual[k] < min_base_qual) qlow_row[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
    return bad;
}
