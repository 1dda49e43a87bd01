"""phase_snvs -- same call surface as reference unfazed/snv_phaser.py:356-399; the work
goes through the C ABI (unfazed_amd.engine.HipEngine) instead of the thread pool over
multithread_read_phasing.  `threads` only selects which of the reference's two behaviours
around swallowed worker exceptions is mirrored (see hostpath)."""
from __future__ import annotations

from . import abi, session


def phase_snvs(
    dnms, kids, pedigrees, sites, threads, build, no_extended, multithread_proc_min, quiet_mode,
    ab_homref, ab_homalt, ab_het, min_gt_qual, min_depth, search_dist, insert_size_max_sample,
    stdevs, min_map_qual, readlen, split_error_margin, evidence_min_ratio=10,
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
        return host.run_read_phasing(
            dnms, pedigrees, threads, build, no_extended, multithread_proc_min, quiet_mode, params,
            search_dist, insert_size_max_sample, stdevs, readlen,
        )
