"""profiles/<round>_build_pmc.json from the counters scripts/build_pmc.sh collected (gpurun_out/TAG/build_pmc.json): per-record figures of the
header build of the staged pass (k_pack_link).   usage: build_pmc_summary.py TAG ROUND"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
j = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "build_pmc.json")))
k = j["k_pack_link"]
c = {n: v["mean"] for n, v in k["counters_per_launch"].items()}
waves = c["SQ_WAVES"]
recs = waves * 256 / 4  # a workgroup of four waves builds a span of 1 024 records: 256 records per wave-round, four rounds
recs = waves / 4 * 1024
isa = {}
try:
    isa = json.load(open(os.path.join(ROOT, "profiles", "%s_isa_stats.json" % rnd)))["kernels"].get("k_pack_link", {})
except Exception:
    pass
out = {
    "kernel": "k_pack_link (header build of the link form, csrc/k_reads.hip)",
    "kernel_source_sha": j.get("bench", {}).get("kernel_source_sha"),
    "how": "scripts/build_pmc.sh: rocprofv3 --kernel-trace --pmc in six passes over `bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 1 --warmup 1` (the staged pass of "
           "100 k DNMs: %s chunks; under --pmc kernels run one at a time: solo durations); per launch, mean over the launches; scripts/build_pmc_summary.py" % j.get("bench", {}).get("chunks"),
    "records_per_launch_mean": recs,
    "launches": k["counters_per_launch"]["SQ_WAVES"]["launches"],
    "solo_duration_ns": k.get("solo_duration_ns"),
    "solo_ns_per_record_median": round(k["solo_duration_ns"]["median"] / recs, 5) if k.get("solo_duration_ns") else None,
    "per_record": {
        "valu_wave_instructions_per_64": c["SQ_INSTS_VALU"] / recs * 64, "salu_per_64": c["SQ_INSTS_SALU"] / recs * 64, "branch_per_64": c["SQ_INSTS_BRANCH"] / recs * 64,
        "lds_per_64": c["SQ_INSTS_LDS"] / recs * 64, "vmem_read_per_64": c["SQ_INSTS_VMEM_RD"] / recs * 64, "vmem_write_per_64": c["SQ_INSTS_VMEM_WR"] / recs * 64,
        "hbm_bytes_read_fetch_x2": c["FETCH_SIZE"] * 1024 * 2 / recs, "hbm_bytes_written": c["WRITE_SIZE"] * 1024 / recs,
        "algorithmic_bytes_in": 6.3, "algorithmic_bytes_in_note": "what the kernel reads per record: the link columns but the dictionary index (4.3 B) + the 16-bit index k_tup_expand rebuilt in HBM (2 B)", "algorithmic_bytes_out": "RecA 16 + RecB 16 + fm 4 + qoff 4 + nlow 1 + umask 2 + qs 2 + one CIGAR word 4 + the listed bases' row (16 per record with bases) + its quality row (4) ~ 60",
    },
    "wave_cycles_active_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_cycles_parked_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "simd_quad_cycles_taken_by_instructions_frac": c["SQ_ACTIVE_INST_ANY"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024 / 4) if False else None,
    "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
    "tcp_accesses_per_record": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / recs,
    "registers": {"vgpr": isa.get("vgpr_count", k.get("vgpr")), "sgpr_spilled": isa.get("sgpr_spill_count"), "lds_bytes": k.get("lds"), "workgroups_per_cu": 3},
    "counters_per_launch": k["counters_per_launch"],
}
# instructions in flight against the issue slots of the chip's 1 024 SIMDs: SQ_ACTIVE_INST_ANY is in quad-cycles summed over waves; GRBM_GUI_ACTIVE is
# summed over the eight XCDs
cycles = c["GRBM_GUI_ACTIVE"] / 8.0
out["simd_quad_cycles_taken_by_instructions_frac"] = c["SQ_ACTIVE_INST_ANY"] / (cycles / 4.0 * 1024)
if "k_pack_rec" in j and j["k_pack_rec"].get("solo_duration_ns"):
    out["note_other_launches"] = "k_pack_rec in the same run: the generator's 187 M-record resident table (%d launch(es), %.2f ms)" % (
        j["k_pack_rec"]["solo_duration_ns"]["launches"], j["k_pack_rec"]["solo_duration_ns"]["median"] / 1e6)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_build_pmc.json" % rnd), "w"), indent=1)
print(json.dumps({a: b for a, b in out.items() if a != "counters_per_launch"}, indent=1))
