"""The command line fed with BCF inputs (--sites and --dnms) gives the BED of the text-VCF inputs."""
import os
import subprocess
import sys

import pytest

from synth.small import SmallConfig, make_small
from tests.bcfio import write_bcf
from tests.filesio import dump_dataset
from unfazed_amd.io_vcf import read_vcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bcf_inputs_match_text_inputs(tmp_path, hip_lib):
    ds = make_small(SmallConfig(seed=77, n_dnms=10))
    paths = dump_dataset(ds, str(tmp_path))
    kid = list(ds.pedigrees)[0]
    smp, recs, _ = read_vcf(paths["sites"])
    sites_bcf = str(tmp_path / "sites.bcf")
    write_bcf(sites_bcf, smp, recs, ds.contigs, use_idx=True)
    smp2, recs2, _ = read_vcf(paths["dnm_vcf"])
    dnm_bcf = str(tmp_path / "dnms.bcf")
    write_bcf(dnm_bcf, smp2, recs2, ds.contigs)
    base = ["-p", paths["ped"], "--build", "38", "-t", "1", "-q", "-o", "bed", "--bam-pairs", "%s:%s" % (kid, paths["bams"][kid])]
    outs = []
    for dn, st in ((paths["dnm_vcf"], paths["sites"]), (dnm_bcf, sites_bcf)):
        r = subprocess.run([sys.executable, "-m", "unfazed_amd", "-d", dn, "-s", st] + base, cwd=ROOT, check=True,
                           capture_output=True, text=True)
        outs.append(r.stdout)
    assert outs[0] == outs[1]
    assert len(outs[0].strip().split("\n")) >= 2
    # ... and with the BCF's CSI index next to it (bcftools index): the sites of the batch's windows only, the same BED
    from tests.filesio import write_csi
    from unfazed_amd import io_native
    write_csi(sites_bcf)
    assert io_native.tabix_index_path(sites_bcf) == sites_bcf + ".csi"
    r = subprocess.run([sys.executable, "-m", "unfazed_amd", "-d", dnm_bcf, "-s", sites_bcf] + base, cwd=ROOT, check=True, capture_output=True, text=True)
    assert r.stdout == outs[0]
    r = subprocess.run([sys.executable, "-m", "unfazed_amd", "-d", dnm_bcf, "-s", sites_bcf] + base[:-4] + ["-o", "vcf"] + base[-2:],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "BCF" in (r.stderr + r.stdout)
