"""Builds the in-tree native libraries:
  unfazed_amd/libunfazed_hip.so  the HIP kernels + C ABI, hipcc for gfx950 (cross-compiles without a GPU)
  unfazed_amd/libunfazed_io.so   the host-side BAM / VCF decoders (g++, zlib, threads)
Both are git-ignored but travel with the tree."""
import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["abi.hip", "k_sites.hip", "k_reads.hip", "k_inflate.hip", "k_bamwalk.hip", "k_bamjoin.hip"]
DEFAULT_FLAGS = []
LIB = os.path.join(_HERE, "libunfazed_hip.so")


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over the device sources (csrc/*.hip, *.hpp, include/*.h): profiles record it, and bench.py quotes a
    profile's counters only while the kernels it measured are the kernels of this build"""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(_HERE, "..", "include")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(inc, "uz_types.h")) +
                    glob.glob(os.path.join(inc, "unfazed_hip.h"))):
        if os.path.basename(f).startswith("io_"):
            continue  # host-only decoders
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, extra_flags=(), out=None):
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(_HERE, "..", "include")
    srcs = [os.path.join(csrc, s) for s in SRC]
    # every header a source can include: editing phase_body.hpp (the k_phase body) must rebuild the library
    deps = srcs + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(inc, "*.h"))
    lib = out or LIB
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(deps):
        return lib
    # several ranks of one node may get here at once (bench.py --gpus N): build under a file lock
    import fcntl
    lock = open(lib + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(deps):
            return lib
        return _compile(lib, srcs, inc, csrc, extra_flags, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _compile(lib, srcs, inc, csrc, extra_flags, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-parameter", "-I", inc, "-I", csrc] + (list(extra_flags) if extra_flags else DEFAULT_FLAGS) + srcs + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    return lib


IO_SRC = ["io_bam.cpp", "io_vcf.cpp", "io_pack.cpp", "io_stage.cpp"]
IO_LIB = os.path.join(_HERE, "libunfazed_io.so")


def build_io(force=False, verbose=False, out=None):
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(_HERE, "..", "include")
    srcs = [os.path.join(csrc, s) for s in IO_SRC]
    deps = srcs + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(inc, "*.h"))
    lib = out or IO_LIB
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(deps):
        return lib
    import fcntl
    lock = open(lib + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(deps):
            return lib
        cxx = os.environ.get("CXX", "g++")
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-parameter",
               "-I", inc, "-I", csrc] + srcs + ["-lz", "-ldl", "-o", lib + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(lib + ".tmp", lib)
        return lib
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_io(force="--force" in sys.argv, verbose=True))
