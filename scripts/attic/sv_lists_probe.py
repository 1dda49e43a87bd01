"""development aid: the SV goldens with the list form of the qualities forced onto every table -- does the kernel's guard ever fire?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_golden import SV, WIDE_SV, check_sv_golden, check_wide_sv
from unfazed_amd import session
from unfazed_amd.engine import HipEngine, UnfazedHipError
real = HipEngine.upload_reads
def forced(self, reads, min_base_qual=None, point_only=False, fetches=None, all_bases=False):
    return real(self, reads, min_base_qual=min_base_qual, point_only=True)
HipEngine.upload_reads = forced
eng = HipEngine(0)
session.set_backend(eng)
ok = bad = 0
for path in SV:
    session._READS.clear(); session._HOSTS.clear()
    try:
        check_sv_golden(eng, path); ok += 1
    except UnfazedHipError as e:
        bad += 1; print("REFUSED", os.path.basename(path), str(e)[:120])
for name in WIDE_SV:
    session._READS.clear(); session._HOSTS.clear()
    try:
        check_wide_sv(eng, name); ok += 1
    except UnfazedHipError as e:
        bad += 1; print("REFUSED", name, str(e)[:120])
print("sv goldens with lists forced: %d reproduced, %d refused" % (ok, bad))
