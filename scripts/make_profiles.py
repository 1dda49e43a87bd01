"""Copy the judged summaries of a profiling round (scripts/profile_round.sh TAG, run on the GPU box) from gpurun_out/TAG
into profiles/ (tracked): kernel statistics, the per-kernel PMC means, and the derived files bench.py reads
(k1_traffic.json, phase_latency.json, phase_issue.json).   usage: make_profiles.py TAG ROUND"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unfazed_amd.build import kernel_source_hash  # noqa: E402

tag, rnd = sys.argv[1], sys.argv[2]
# the device sources the profile was collected on: the bench line records it (roofline.kernel_source_sha); bench.py quotes these
# counters only while it equals the hash of the build it runs
KSHA = None
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_bench_kernel_stats.csv" % rnd))
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, "%s_pmc_summary.json" % rnd))
for extra, name in (("staged_kernel_stats.csv", "%s_bench_staged_kernel_stats.csv"), ("bench_staged_under_rocprof.json", "%s_bench_staged_under_rocprof.json"),
                    ("feed_kernel_stats.csv", "%s_bench_feed_kernel_stats.csv"), ("bench_feed_under_rocprof.json", "%s_bench_feed_under_rocprof.json"),
                    ("feed_pmc.txt", "%s_feed_pmc.txt"), ("inflate_pmc.json", "%s_inflate_pmc.json"), ("resident_timeline.json", "%s_resident_timeline.json"),
                    ("bench_line.json", "%s_bench_line.json"), ("bench_line_shard12500.json", "%s_bench_line_shard12500.json")):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, name % rnd))
line = json.loads([x for x in open(os.path.join(src, "stats.log")) if x.startswith("{")][-1])
KSHA = (line.get("roofline") or {}).get("kernel_source_sha") or kernel_source_hash()
json.dump(line, open(os.path.join(dst, "%s_bench_under_rocprof.json" % rnd), "w"), indent=1)
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}


full_size = json.load(open(os.path.join(src, "full_size_launches.json"))) if os.path.exists(os.path.join(src, "full_size_launches.json")) else {}
if full_size:
    json.dump(full_size, open(os.path.join(dst, "%s_full_size_launches.json" % rnd), "w"), indent=1)


def avg_ns(pat):
    # (k_phase<true> runs twice per batch since round 5: its full-size launches alone, from the trace -- scripts/profile_round.sh)
    if pat.startswith("k_phase<true>") and "k_phase" in full_size:
        return float(full_size["k_phase"]["avg_ns"]), int(full_size["k_phase"]["full_size_launches"])
    for k, r in stats.items():
        if pat in k:
            return float(r["AverageNs"]), int(r["Calls"])
    return None, 0


def c(kernel, name):
    return pmc[kernel]["counters_per_launch"][name]["mean"]


how = ("rocprofv3 --kernel-trace --pmc <counters> in separate passes (scripts/profile_round.sh) over `bench.py --no-staged --no-cpu "
       "--steps 3 --warmup 1` (100 k DNMs, 20 M sites); per launch, mean over the launches of the run")
# K1: pure 16-byte streams -> the guide's x2 correction of FETCH_SIZE applies (MI355X_MICROARCH.md, HBM)
f, w = c("k_site_scan", "FETCH_SIZE") * 1024, c("k_site_scan", "WRITE_SIZE") * 1024
n_sites = line["config"]["sites"]
json.dump({"kernel": pmc["k_site_scan"]["full_name"], "n_sites": n_sites, "FETCH_SIZE_KB_raw": f / 1024, "WRITE_SIZE_KB_raw": w / 1024,
           "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": int(2 * f + w),
           "algorithmic_bytes_per_launch": 20 * n_sites, "avg_ns_rocprof": avg_ns("k_site_scan")[0], "how": how, "kernel_source_sha": KSHA,
           "source": "profiles/%s_pmc_summary.json" % rnd}, open(os.path.join(dst, "k1_traffic.json"), "w"), indent=1)
# k_phase: latency model from the SQ / TCP counters
waves = c("k_phase", "SQ_WAVES")
loads = c("k_phase", "SQ_INSTS_VMEM_RD")
wait_q = c("k_phase", "SQ_WAIT_ANY")
wave_q = c("k_phase", "SQ_WAVE_CYCLES")
ns, calls = avg_ns("k_phase<true>(")
if ns is None:
    ns, calls = avg_ns("k_phase(")
json.dump({"kernel": "k_phase", "dnms": line["config"]["dnms_per_gpu"], "waves_resident": waves,
           "vmem_read_instructions_per_wave": loads / waves,
           "wave_cycles_parked_frac": wait_q / wave_q, "active_frac": c("k_phase", "SQ_ACTIVE_INST_ANY") / wave_q,
           "issue_stall_frac": c("k_phase", "SQ_WAIT_INST_ANY") / wave_q,
           "parked_cycles_per_read_instruction": 4 * wait_q / loads,
           "l1_miss_latency_cycles": c("k_phase", "TCP_TCC_READ_REQ_LATENCY_sum") / c("k_phase", "TCP_TCC_READ_REQ_sum"),
           "l1_accesses_per_dnm": c("k_phase", "TCP_TOTAL_CACHE_ACCESSES_sum") / line["config"]["dnms_per_gpu"],
           "l2_requests_per_dnm": c("k_phase", "TCC_REQ_sum") / line["config"]["dnms_per_gpu"],
           "l2_hit_rate": c("k_phase", "TCC_HIT_sum") / (c("k_phase", "TCC_HIT_sum") + c("k_phase", "TCC_MISS_sum")),
           "fetch_KB_raw_per_dnm": c("k_phase", "FETCH_SIZE") / line["config"]["dnms_per_gpu"],
           "lds_instructions_per_wave": c("k_phase", "SQ_INSTS_LDS") / waves,
           "avg_ns_rocprof": ns, "vgpr": pmc["k_phase"]["vgpr"], "lds_static_bytes": pmc["k_phase"]["lds"], "scratch": pmc["k_phase"]["scratch"],
           "how": how + "; SQ_* in quad-cycles (x4 = shader cycles)", "source": "profiles/%s_pmc_summary.json" % rnd},
          open(os.path.join(dst, "phase_latency.json"), "w"), indent=1)
print(open(os.path.join(dst, "phase_latency.json")).read())
# k_phase: what bounds it.  Instruction issue: wave-instructions by class, the share of the VALU pipes' cycles they take (a
# wave64 VALU instruction occupies its SIMD's 16-lane pipe for 4 cycles; 1024 SIMDs), lane utilisation, instruction cache.
cp = pmc["k_phase"]["counters_per_launch"]
if "SQ_INSTS_VALU" in cp and "GRBM_GUI_ACTIVE" in cp:
    cyc = c("k_phase", "GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
    n_simd = 1024
    valu, salu = c("k_phase", "SQ_INSTS_VALU"), c("k_phase", "SQ_INSTS_SALU")
    dn = line["config"]["dnms_per_gpu"]
    st = line["calls"]["status_counts"]
    busy = dn - st[2] if len(st) > 2 else dn  # DNMs with candidate sites: the others end at once
    issue = {"kernel": "k_phase<true> (LDS build)", "dnms": dn, "dnms_with_candidates": busy,
             "dnms_redone_by_hbm_build": line["calls"].get("dnms_redone_by_hbm_build_of_k_phase"),
             "wave_instructions_per_launch": {"valu": valu, "salu": salu, "branch": c("k_phase", "SQ_INSTS_BRANCH"),
                                              "vmem": c("k_phase", "SQ_INSTS_VMEM"), "lds": c("k_phase", "SQ_INSTS_LDS"),
                                              "smem": c("k_phase", "SQ_INSTS_SMEM")},
             "valu_wave_instructions_per_dnm_with_candidates": valu / busy,
             "shader_cycles_per_launch": cyc, "shader_clock_GHz": cyc / ns, "waves_resident": waves, "waves_per_simd": waves / n_simd,
             "valu_pipe_busy_frac": 4.0 * valu / (n_simd * cyc), "scalar_pipe_busy_frac": 4.0 * salu / (n_simd * cyc),
             "valu_lane_utilisation": c("k_phase", "SQ_THREAD_CYCLES_VALU") / (64.0 * c("k_phase", "SQ_ACTIVE_INST_VALU")),
             "wave_cycles_parked_frac": wait_q / wave_q, "wave_active_frac": c("k_phase", "SQ_ACTIVE_INST_ANY") / wave_q,
             "icache_miss_rate": c("k_phase", "SQC_ICACHE_MISSES") / max(1.0, c("k_phase", "SQC_ICACHE_REQ")),
             "ta_busy_frac": c("k_phase", "TA_TA_BUSY_sum") / (256.0 * cyc),
             "valu_floor_ms": 4.0 * valu / n_simd / (cyc / ns) / 1e6, "avg_ms_rocprof": ns / 1e6,
             "vgpr": pmc["k_phase"]["vgpr"], "scratch": pmc["k_phase"]["scratch"],
             "how": how + "; SQ_INSTS_* count wave-instructions, SQ_ACTIVE_* / SQ_WAIT_* are in quad-cycles",
             "source": "profiles/%s_pmc_summary.json" % rnd}
    issue["kernel_source_sha"] = KSHA
    json.dump(issue, open(os.path.join(dst, "phase_issue.json"), "w"), indent=1)
    print(json.dumps(issue, indent=1))


# the register question: rocprofv3's VGPR column against the code object's own metadata (scripts/isa_stats.py -> profiles/<round>_isa_stats.json,
# which needs no GPU: run it before this script) -- and what each caps
try:
    isa = json.load(open(os.path.join(dst, "%s_isa_stats.json" % rnd)))
    isa = isa.get("kernels", isa)
    kp = next(v for k, v in isa.items() if "k_phase<true>" in k)
    pi_path = os.path.join(dst, "phase_issue.json")
    pi = json.load(open(pi_path))
    total = int(kp.get("vgpr_count", 0)) + int(kp.get("agpr_count", 0))
    alloc = (total + 7) // 8 * 8
    pi["vgpr_rocprof_column"] = pi.get("vgpr")
    pi["vgpr_count_code_object"] = int(kp.get("vgpr_count", 0))
    pi["agpr_count_code_object"] = int(kp.get("agpr_count", 0))
    pi["sgpr_spill_count_code_object"] = int(kp.get("sgpr_spill_count", 0))
    pi["waves_per_simd_the_registers_allow"] = 512 // alloc if alloc else None
    pi["occupancy_note"] = ("the code object's metadata is what the hardware allocates by (blocks of 8 of a SIMD's 512 registers per lane): %d -> %d -> %d waves per SIMD, "
                            "the same four the LDS arenas leave room for (16 single-wave workgroups per CU) -- BOTH cap the read stage at four, not the arena alone; "
                            "a fifth wave needs <= 96 registers AND arenas <= 8 KB.  rocprofv3's VGPR_Count column (%s) is not that number" % (total, alloc, 512 // alloc, pi.get("vgpr")))
    json.dump(pi, open(pi_path, "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print("no ISA statistics beside the profile (%s): phase_issue.json keeps rocprofv3's column only" % e, file=sys.stderr)
