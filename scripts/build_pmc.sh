#!/bin/bash
# Counters of the header build of the staged pass (k_pack_link; solo: --pmc serialises kernels) -> gpurun_out/TAG/build_pmc.json
#   scripts/build_pmc2.sh TAG [env assignments...]
TAG=${1:-bpmc2}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"; rm -rf "$OUT/bp_"*
ONLY='--kernel-include-regex k_pack_link|k_pack_rec'
ARGS="$ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 1 --warmup 1"
run() { n=$1; shift; rocprofv3 $ONLY --output-format csv --kernel-trace --pmc "$@" -d "$OUT/bp_$n" -o run -- python3 $ARGS > "$OUT/bp_$n.log" 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT GRBM_GUI_ACTIVE
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run e FETCH_SIZE
run f WRITE_SIZE
cd "$ROOT"
python3 scripts/pmc_summary.py "$OUT/build_pmc.json" "$OUT"/bp_a "$OUT"/bp_b "$OUT"/bp_c "$OUT"/bp_d "$OUT"/bp_e "$OUT"/bp_f > /dev/null 2>&1
# solo durations of the launches (the trace of pass a) and the records they built (the bench line of the same run)
python3 - "$OUT" <<'P'
import csv, glob, json, sys
out = sys.argv[1]
j = json.load(open(out + "/build_pmc.json"))
f = glob.glob(out + "/bp_a/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
for key, pat in (("k_pack_link", "k_pack_link("), ("k_pack_rec", "k_pack_rec<")):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if pat in r["Kernel_Name"])
    if d and key in j:
        j[key]["solo_duration_ns"] = {"launches": len(d), "median": d[len(d) // 2], "min": d[0], "max": d[-1]}
line = [x for x in open(out + "/bp_a.log") if x.startswith("{")]
if line:
    b = json.loads(line[-1])
    j["bench"] = {"read_records_staged_per_step": b["link"]["read_records_staged"], "chunks": b["link"]["chunks"], "bytes_per_step": b["link"]["bytes_per_step"],
                  "kernel_source_sha": (b.get("roofline") or {}).get("kernel_source_sha")}
json.dump(j, open(out + "/build_pmc.json", "w"), indent=1)
print(json.dumps({k: v.get("solo_duration_ns") for k, v in j.items() if isinstance(v, dict) and "solo_duration_ns" in v}))
P
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +2M -delete
