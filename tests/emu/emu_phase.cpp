// emu_phase.cpp -- CPU emulation harness of the per-DNM read-stage kernel body
// (unfazed_amd/csrc/phase_body.hpp compiled with -DUZ_EMU: one lane, phases run
// sequentially).  DEBUGGING AID for the authoring container (no GPU there): it lets
// tests/test_emu_phase.py run the exact kernel logic against the oracle on CPU.
// Never loaded by the product.
#define UZ_EMU 1
#include <cstdlib>
#include <cstring>
#include <vector>

#include "phase_body.hpp"
#ifdef UZ_EMU_STATS
extern "C" { long long uz_emu_stats[16]; long long *uz_emu_log = nullptr; }
#endif

extern "C" int emu_phase(const uz_params *P, const uz_sites_view *S, const uz_reads_view *Rv, const uz_dnms_view *D,
                         const int64_t *cand_off, const int32_t *cand_idx, const uint8_t *cand_flags,
                         const int64_t *het_off, const int32_t *het_idx, int32_t *status, int32_t *counts,
                         int32_t *origin, int32_t *evidence, long long *list_start, int32_t *list_len, int32_t *pool,
                         long long pool_cap, long long *pool_used, const uint8_t *no_seq /* optional: records staged without bases */,
                         int32_t *base_err_out /* optional: 1 when the bases of such a record were requested */,
                         const uint16_t *umask_in /* optional: staged 32-base units per record (UZ_UMASK_ALL: all) -- the others are left out of the rows */,
                         const int64_t *bl_off /* optional, with umask_in: [n + 1] listed bases per record (uz_types.h bl_*) -- a record with a list keeps, inside its staged units, ONLY those bases (the others read as code 0) and is marked UZ_UMASK_LISTED, as the device's header build leaves it */,
                         const uint16_t *bl_pos) {
    // the table in the packed form the device holds (built here on the host from the ASCII view)
    const int64_t n = Rv->n_segs;
    std::vector<uint8_t> nlow((size_t)n + 1);
    std::vector<uint16_t> qs((size_t)n + 1);
    std::vector<uint16_t> umask((size_t)n + 1, (uint16_t)UZ_UMASK_ALL); // every unit of every row
    std::vector<RecA> ra((size_t)n + 1);
    std::vector<RecB> rb((size_t)n + 1);
    std::vector<uint32_t> fm((size_t)n + 1), qoff((size_t)n + 1), cigar;
    std::vector<uint8_t> seq4, qlow;
    uint64_t coff = 0, uoff = 0;
    for (int64_t i = 0; i < n; i++) {
        uz_pack_rec(ra[i], rb[i], Rv->start[i], Rv->end[i], (uint32_t)coff, (no_seq && no_seq[i]) ? UZ_NO_SEQ_OFF : (uint32_t)uoff, Rv->mate[i],
                    Rv->qname[i], Rv->l_seq[i], Rv->n_cigar[i], Rv->tlen[i]);
        fm[i] = uz_pack_fm(Rv->flag[i], Rv->mapq[i], Rv->aux[i]);
        qoff[i] = (uint32_t)uoff;
        for (int k = 0; k < (int)Rv->n_cigar[i]; k++) cigar.push_back(Rv->cigar[(size_t)Rv->cigar_off[i] + k]);
        const uint32_t units = UZ_ROW_UNITS(Rv->l_seq[i]);
        seq4.resize((size_t)(uoff + units) * UZ_SEQ4_UNIT_BYTES);
        qlow.resize((size_t)(uoff + units) * UZ_QLOW_UNIT_BYTES);
        if (uz_pack_rows_host(Rv->seq + ((size_t)Rv->sq_off16[i] << 4), Rv->qual + ((size_t)Rv->sq_off16[i] << 4), Rv->l_seq[i],
                              P->min_gt_qual, seq4.data() + (size_t)uoff * UZ_SEQ4_UNIT_BYTES,
                              qlow.data() + (size_t)uoff * UZ_QLOW_UNIT_BYTES) != 0)
            return -2;
        uint32_t kept = units;
        if (umask_in && umask_in[i] != UZ_UMASK_ALL) { // keep the staged units only, back to back (as the device's rows are)
            umask[i] = umask_in[i];
            kept = 0;
            for (uint32_t u = 0; u < units; u++)
                if ((umask_in[i] >> u) & 1u) {
                    memmove(seq4.data() + (size_t)(uoff + kept) * UZ_SEQ4_UNIT_BYTES, seq4.data() + (size_t)(uoff + u) * UZ_SEQ4_UNIT_BYTES, UZ_SEQ4_UNIT_BYTES);
                    memmove(qlow.data() + (size_t)(uoff + kept) * UZ_QLOW_UNIT_BYTES, qlow.data() + (size_t)(uoff + u) * UZ_QLOW_UNIT_BYTES, UZ_QLOW_UNIT_BYTES);
                    kept++;
                }
            if (bl_off && bl_off[i + 1] > bl_off[i]) { // only the listed bases survive
                std::vector<uint8_t> keepn((size_t)kept * UZ_SEQ4_UNIT_BYTES, 0);
                for (int64_t e = bl_off[i]; e < bl_off[i + 1]; e++) {
                    const uint32_t q = bl_pos[e], u = q >> 5;
                    if (u > 14 || !((umask_in[i] >> u) & 1u)) return -3; // a listed base outside the mask
                    const uint32_t row = (uint32_t)__builtin_popcount(umask_in[i] & ((1u << u) - 1u));
                    keepn[(size_t)row * UZ_SEQ4_UNIT_BYTES + ((q & 31) >> 1)] |= (uint8_t)((q & 1) ? 0x0F : 0xF0);
                }
                for (size_t b = 0; b < keepn.size(); b++) seq4[(size_t)uoff * UZ_SEQ4_UNIT_BYTES + b] &= keepn[b];
                umask[i] = (uint16_t)(umask_in[i] | UZ_UMASK_LISTED);
            }
        }
        {   // as the staged form has it (list form of the quality plane): the count of every record; a quality row only for a record
            // that carries its bases and has at most UZ_QLOW_LIST_MAX low ones -- asking for a bit of any other sets base_err = 2
            int low = 0;
            const uint8_t *ql = Rv->qual + ((size_t)Rv->sq_off16[i] << 4);
            for (int k = 0; k < (int)Rv->l_seq[i]; k++) low += (int)ql[k] < P->min_gt_qual;
            nlow[i] = (uint8_t)(low > 255 ? 255 : low);
            if ((no_seq && no_seq[i]) || low > UZ_QLOW_LIST_MAX) qoff[i] = UZ_NO_QLOW_OFF;
        }
        coff += Rv->n_cigar[i];
        uoff += kept;
    }
    cigar.push_back(0); seq4.resize(seq4.size() + 64); qlow.resize(qlow.size() + 64);
    RD R;
    R.contig_off = Rv->contig_off; R.max_span = Rv->max_span; R.n_contigs = Rv->n_contigs;
    R.ra = ra.data(); R.rb = rb.data(); R.fm = fm.data(); R.cigar = cigar.data(); R.seq4 = seq4.data(); R.qlow = qlow.data();
    R.qs = qs.data(); R.min_map_qual = P->min_map_qual; R.qoff = qoff.data(); R.nlow = nlow.data(); R.umask = umask.data();
    int32_t base_err = 0;
    R.err = &base_err;
    std::vector<int32_t> coarse((size_t)(n >> 12) + 2);
    for (int64_t k = 0; (k << 12) < n; k++) coarse[k] = Rv->start[k << 12];
    R.coarse = coarse.data();
    std::vector<int32_t> mid_idx((size_t)(n >> 6) + 2);
    for (int64_t k = 0; (k << 6) < n; k++) mid_idx[k] = Rv->start[k << 6];
    R.mid = mid_idx.data();
    std::vector<int32_t> mid8((size_t)(n >> 3) + 16);
    for (int64_t k = 0; (k << 3) < n; k++) mid8[k] = Rv->start[k << 3];
    R.mid8 = mid8.data();
    for (int64_t i = 0; i < n; i++) { // the QC word as the header build makes it
        int nonmatch = 0, none = 0;
        for (int k = 0; k < (int)rb[i].n_cigar; k++) uz_cigar_op_counts(cigar[(size_t)ra[i].cigar_off + k], nonmatch, none);
        qs[i] = uz_qs_word(fm[i] & 0xFFFFu, fm[i] >> 24, (fm[i] >> 16) & 0xFFu, nlow[i], rb[i].n_cigar, nonmatch, none);
    }
    PhaseArgs a;
    memset(&a, 0, sizeof(a));
    a.n = D->n;
    a.min_gt_qual = P->min_gt_qual; a.readlen = P->readlen; a.no_extended = P->no_extended;
    a.read_goal = P->read_goal; a.evidence_min_ratio = P->evidence_min_ratio; a.cutoff = D->cutoff;
    a.split_error_margin = P->split_error_margin;
    a.spos = S->pos; a.sref = S->ref_base; a.salt = S->alt_base;
    a.cand_off = cand_off; a.het_off = het_off; a.cand_idx = cand_idx; a.het_idx = het_idx; a.cand_flags = cand_flags;
    a.rcontig = D->rcontig; a.dstart = D->start; a.dend = D->end; a.dflags = D->dflags; a.vartype = D->vartype; a.allele_off = D->allele_off; a.alleles = D->alleles;
    a.R = R;
    a.status = status; a.counts = counts; a.origin = origin; a.evidence = evidence;
    a.want_lists = 1; a.pool = pool; a.pool_cap = (unsigned long long)pool_cap;
    unsigned long long cursor = 0;
    a.pool_cursor = &cursor; a.list_start = list_start; a.list_len = list_len;
    std::vector<int32_t> pre_win((size_t)4 * D->n + 4), pre_ha((size_t)het_off[D->n] + 2), pre_hl((size_t)het_off[D->n] + 2);
    a.pre_win = pre_win.data(); a.pre_ha = pre_ha.data(); a.pre_hl = pre_hl.data();
    long long mA = 0, mT = 0, mH = 0, mC = 0, mM = 2;
    for (int d = 0; d < D->n; d++) {
        int32_t b[5];
        long long tp = 0;
        int mhp = 0;
        uz_phase_bounds(a, d, b, 0, 1, tp, mhp);
        b[1] = (int32_t)(tp > 0x7FFFFFF0LL ? 0x7FFFFFF0LL : tp);
        b[4] = mhp;
        if (b[0] > mA) mA = b[0];
        if (b[1] > mT) mT = b[1];
        if (b[2] > mH) mH = b[2];
        if (b[3] > mC) mC = b[3];
        long long M = (long long)b[1] + 4LL * b[0] * (b[4] + 1);
        if (M > mM) mM = M;
    }
    Caps caps;
    caps.A = (int)mA; caps.T = (int)mT; caps.H = (int)mH; caps.C = (int)mC; caps.I = (int)(4 * mA);
    long long p2 = 1; while (p2 < mM) p2 <<= 1;
    caps.M = (int)p2;
    const size_t bytes = uz_scratch_layout(caps, a.so);
    std::vector<uint8_t> scratch(bytes + 256, 0xCD);
    a.scratch = scratch.data(); a.scratch_per_wg = bytes; a.caps = caps;
    WgShared sh;
    // exercise both builds of the body: the HBM build alone, the arena build with a tiny arena (gives most DNMs up at one
    // of its checks -> they are redone by the HBM build, as the device does), the arena build with a roomy arena
    static const int arena_sizes[3] = {0, 3072, 65536};
    std::vector<uint8_t> arena(65536 + 64);
    for (int d = 0; d < D->n; d++) {
        a.lds_arena_bytes = arena_sizes[d % 3];
#ifdef UZ_EMU_STATS
        a.lds_arena_bytes = 65536; // the statistics are those of the arena build (scripts/phase_sizes.py)
#endif
        int given_up = 1;
        if (a.lds_arena_bytes) given_up = uz_phase_dnm<true>(&a, scratch.data(), &sh, arena.data(), d);
        if (given_up) uz_phase_dnm<false>(&a, scratch.data(), &sh, nullptr, d);
    }
    *pool_used = (long long)cursor;
    if (base_err_out) *base_err_out = base_err;
    return 0;
}
