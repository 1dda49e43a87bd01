"""Development aid (GPU box): build the library with -DUZ_PHASE_TIMING (lane 0 adds the shader-clock ticks between phase
boundaries of k_phase) into build_variants/ and run the resident bench pass with it; the breakdown goes to stderr."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unfazed_amd import build  # noqa: E402

out = os.path.join(ROOT, "build_variants", "libunfazed_hip_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
build.build(extra_flags=["-DUZ_PHASE_TIMING"] + sys.argv[1:], out=out)  # cross-compiles in the authoring container; fresh builds travel
env = dict(os.environ, UZ_HIP_LIB=out)
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--no-staged", "--no-cpu", "--steps", "2", "--warmup", "1"], env=env))
