#!/bin/bash
# development aid (GPU box): counters of the header build of the staged pass (solo: --pmc serialises kernels)  scripts/build_pmc2.sh [env...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/bpmc2"; rm -rf "$OUT"; mkdir -p "$OUT"
ONLY='--kernel-include-regex k_pack_link|k_pack_rec'
ARGS="$ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 1 --warmup 1"
run() { n=$1; shift; rocprofv3 $ONLY --output-format csv --kernel-trace --pmc "$@" -d "$OUT/$n" -o run -- python3 $ARGS > "$OUT/$n.log" 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT GRBM_GUI_ACTIVE
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run e FETCH_SIZE
run f WRITE_SIZE
cd "$ROOT"
for n in a b c d e f; do python3 scripts/pmc_rows.py "$OUT/$n" | grep -v "false, false\|k_pack_rec<false" ; done
rm -rf "$OUT"
