"""phase_svs -- same call surface as reference unfazed/sv_phaser.py:427-493.

Allele-balance phasing of DEL / DUP (run_cnv_phasing, :357-423) runs on the device
(K1 CNV class codes + K2 whole-region window emit).  The read-backed half for SVs
(run_read_phasing -> collect_reads_sv, read_collector.py:435-602: split / discordant /
clipped read evidence) is the first "next" row of SURVEY.md 8(f) and is NOT built yet:
requesting it raises instead of silently returning allele-balance evidence only, unless the
caller opts in with `allele_balance_only=True`."""
from __future__ import annotations

from . import abi, session


def phase_svs(
    dnms, kids, pedigrees, sites, threads, build, no_extended, multiread_proc_min, quiet_mode,
    ab_homref, ab_homalt, ab_het, min_gt_qual, min_depth, search_dist, insert_size_max_sample,
    stdevs, min_map_qual, readlen, split_error_margin, evidence_min_ratio=10, allele_balance_only=False,
):
    host = session.host_for(sites, insert_size_max_sample)
    params = abi.make_params(
        search_dist=search_dist, min_gt_qual=min_gt_qual, min_depth=min_depth, min_map_qual=min_map_qual,
        readlen=readlen, split_error_margin=split_error_margin, no_extended=no_extended,
        insert_size_max_sample=insert_size_max_sample, evidence_min_ratio=evidence_min_ratio,
        ab_homref=ab_homref, ab_homalt=ab_homalt, ab_het=ab_het,
    )
    cnv_records = host.run_cnv_phasing(dnms, pedigrees, threads, build, multiread_proc_min, quiet_mode, params)
    if not allele_balance_only:
        raise NotImplementedError(
            "read-backed SV phasing (collect_reads_sv) is not built yet; pass allele_balance_only=True "
            "to get the allele-balance records of DEL/DUP events only"
        )
    return cnv_records
