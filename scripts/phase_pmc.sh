#!/bin/bash
# Counters of the read stage alone (k_phase), resident pass: scripts/phase_pmc.sh TAG [env assignments...]  -> gpurun_out/TAG/pmc_rows.txt
TAG=${1:-phasepmc}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-staged --no-cpu --steps 3 --warmup 1 --feed-dnms 0 --no-config5"
ONLY='--kernel-include-regex k_phase<'
run() { n=$1; shift; rocprofv3 $ONLY --output-format csv --kernel-trace --pmc "$@" -d "$OUT/$n" -o run -- python3 $ARGS > "$OUT/$n.log" 2>&1; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run lds SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_IFETCH
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TA_DATA_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
cd "$ROOT"
for n in sq insts lds tcp ta tcc fetch; do python3 scripts/pmc_rows.py "$OUT/$n"; done > "$OUT/pmc_rows.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +4M -delete
cat "$OUT/pmc_rows.txt"
