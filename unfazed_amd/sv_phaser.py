"""phase_svs -- same call surface as reference unfazed/sv_phaser.py:427-493: allele-balance
phasing of DEL / DUP (run_cnv_phasing, :357-423) merged with read-backed phasing of the SV
breakpoints (run_read_phasing, :176-266 -> collect_reads_sv, read_collector.py:435-602: split,
discordant and clipped read evidence), both through the C ABI."""
from __future__ import annotations

from . import abi, session


def phase_svs(
    dnms, kids, pedigrees, sites, threads, build, no_extended, multiread_proc_min, quiet_mode,
    ab_homref, ab_homalt, ab_het, min_gt_qual, min_depth, search_dist, insert_size_max_sample,
    stdevs, min_map_qual, readlen, split_error_margin, evidence_min_ratio=10, allele_balance_only=False,
):
    with session.DEVICE_LOCK, session.no_gc_pauses():
        for bam, ref in {(dn.get("bam", ""), dn.get("cram_ref")) for dn in dnms}:  # a CRAM is decoded against the FASTA its DNMs carry (unfazed.py:270, read_collector.py:372-373)
            session.set_cram_reference(bam, ref)
        host = session.host_for(sites, insert_size_max_sample, dnms=dnms, search_dist=search_dist)
        params = abi.make_params(
            search_dist=search_dist, min_gt_qual=min_gt_qual, min_depth=min_depth, min_map_qual=min_map_qual,
            readlen=readlen, split_error_margin=split_error_margin, no_extended=no_extended,
            insert_size_max_sample=insert_size_max_sample, evidence_min_ratio=evidence_min_ratio,
            ab_homref=ab_homref, ab_homalt=ab_homalt, ab_het=ab_het,
        )
        cnv_records = host.run_cnv_phasing(dnms, pedigrees, threads, build, multiread_proc_min, quiet_mode, params,
                                           annotate=allele_balance_only)
        if allele_balance_only:
            return cnv_records
        read_records = host.run_read_phasing(
            dnms, pedigrees, threads, build, no_extended, multiread_proc_min, quiet_mode, params,
            search_dist, insert_size_max_sample, stdevs, readlen, sv=True,
        )
        for key in cnv_records:  # sv_phaser.py:484-492
            if key not in read_records:
                read_records[key] = cnv_records[key]
            else:
                read_records[key]["cnv_dad_sites"] = cnv_records[key]["cnv_dad_sites"]
                read_records[key]["cnv_mom_sites"] = cnv_records[key]["cnv_mom_sites"]
                read_records[key]["evidence_type"] += "," + cnv_records[key]["cnv_evidence_type"]
        return read_records
