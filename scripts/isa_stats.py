#!/usr/bin/env python
"""Static figures of every gfx950 kernel of the library, from the compiler's own assembly (`hipcc -S`): registers, spills, LDS, scratch, and the
instruction mix that the reviews asked about (SGPR spill traffic = v_readlane / v_writelane, cross-lane steps through LDS = ds_bpermute, DPP and
v_permlane* forms, waits, branches).  Needs no GPU.
    python scripts/isa_stats.py [--json profiles/rNN_isa_stats.json] [--kernel k_phase] [--keep-asm DIR]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def stats_of(asm_text):
    kernels = {}
    # bodies: from "NAME:" (a .globl / .type @function symbol) to its ".Lfunc_end"
    for m in re.finditer(r"^(\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = [ln.strip().split()[0] for ln in body.split("\n") if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        c = lambda pred: sum(1 for x in ins if pred(x))
        dpp = sum(1 for ln in body.split("\n") if ("row_shr" in ln or "row_shl" in ln or "row_ror" in ln or "quad_perm" in ln or "row_bcast" in ln or "row_mirror" in ln
                                                    or "row_half_mirror" in ln or "wave_shr" in ln or "row_newbcast" in ln or "row_share" in ln) and ln.startswith("\t"))
        kernels[name] = {
            "instructions": len(ins),
            "valu": c(lambda x: x.startswith("v_")),
            "salu": c(lambda x: x.startswith("s_") and not x.startswith(("s_waitcnt", "s_load", "s_buffer_load", "s_cbranch", "s_branch", "s_nop", "s_barrier"))),
            "smem": c(lambda x: x.startswith(("s_load", "s_buffer_load"))),
            "v_readlane": c(lambda x: x.startswith("v_readlane")), "v_writelane": c(lambda x: x.startswith("v_writelane")),
            "v_readfirstlane": c(lambda x: x.startswith("v_readfirstlane")),
            "ds_bpermute": c(lambda x: x.startswith("ds_bpermute")), "ds_permute": c(lambda x: x.startswith("ds_permute")), "ds_swizzle": c(lambda x: x.startswith("ds_swizzle")),
            "dpp": dpp, "v_permlane": c(lambda x: x.startswith("v_permlane")),
            "ds_other": c(lambda x: x.startswith("ds_") and not x.startswith(("ds_bpermute", "ds_permute", "ds_swizzle"))),
            "global": c(lambda x: x.startswith("global_")), "flat": c(lambda x: x.startswith("flat_")), "scratch": c(lambda x: x.startswith(("scratch_", "buffer_"))),
            "s_waitcnt": c(lambda x: x.startswith("s_waitcnt")), "s_barrier": c(lambda x: x.startswith("s_barrier")),
            "branches": c(lambda x: x.startswith(("s_cbranch", "s_branch"))),
        }
    # metadata (amdhsa.kernels)
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", asm_text, re.S):
        blk = m.group(0)
        nm = re.search(r"\.name:\s*(\S+)", blk).group(1)
        k = kernels.setdefault(nm, {})
        for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "agpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                    "max_flat_workgroup_size"):
            mm = re.search(r"\." + key + r":\s*(\d+)", blk)
            if mm:
                k[key] = int(mm.group(1))
        k["is_kernel"] = True
    return {n: v for n, v in kernels.items() if v.get("is_kernel")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--kernel", default=None, help="only kernels whose demangled name contains this")
    ap.add_argument("--keep-asm", default=None)
    ap.add_argument("--flags", default="", help="extra compiler flags (quoted)")
    args = ap.parse_args()
    from unfazed_amd import build
    csrc = os.path.join(ROOT, "unfazed_amd", "csrc")
    inc = os.path.join(ROOT, "include")
    d = args.keep_asm or tempfile.mkdtemp(prefix="uzisa_")
    os.makedirs(d, exist_ok=True)
    out = {}
    for src in build.SRC:
        s = os.path.join(d, src.replace(".hip", ".s"))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", inc, "-I", csrc, "-S", "--cuda-device-only",
               os.path.join(csrc, src), "-o", s] + args.flags.split()
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        st = stats_of(open(s).read())
        dm = demangle(list(st))
        for n, v in st.items():
            v["source"] = src
            out[re.sub(r"\(anonymous namespace\)::", "", dm.get(n, n)).split("(")[0]] = v
    if args.kernel:
        out = {k: v for k, v in out.items() if args.kernel in k}
    cols = ["vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "instructions", "valu",
            "v_readlane", "v_writelane", "ds_bpermute", "dpp", "v_permlane", "s_waitcnt", "s_barrier", "branches"]
    print("%-34s" % "kernel" + " ".join("%9s" % c[:9] for c in cols))
    for n in sorted(out):
        print("%-34s" % n[:34] + " ".join("%9s" % out[n].get(c, "") for c in cols))
    if args.json:
        json.dump({"kernel_source_sha": build.kernel_source_hash(), "compiler": "hipcc -O3 --offload-arch=gfx950 -S (ROCm 7.2.0)", "kernels": out}, open(args.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
