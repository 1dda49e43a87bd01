#!/bin/bash
# One profiling round on the GPU box (run through gpurun): kernel statistics and the PMC passes the bench line and DESIGN.md quote.
# usage: scripts/profile_round.sh TAG   -> gpurun_out/TAG/{stats,sq,tcp,tcc,fetch,write}/..., summaries in gpurun_out/TAG/
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-staged --no-cpu --steps 3 --warmup 1 --feed-dnms 0 --no-config5"
# counters only for the product's kernels (the synthetic generator's launches would be serialised and counted too)
ONLY='--kernel-include-regex k_phase|k_site_scan|k_window|k_pack_rec|k_pack_link|k_expand_seq2|k_cnv|k_bounds_reduce'
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS -d $OUT/sq -o run -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU -d $OUT/insts -o run -- python3 $ARGS > $OUT/insts.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_MISSES TA_TA_BUSY_sum -d $OUT/misc -o run -- python3 $ARGS > $OUT/misc.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/tcp -o run -- python3 $ARGS > $OUT/tcp.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/tcc -o run -- python3 $ARGS > $OUT/tcc.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run -- python3 $ARGS > $OUT/write.log 2>&1
cd $ROOT
python3 scripts/pmc_summary.py $OUT/pmc_summary.json $OUT/sq $OUT/insts $OUT/misc $OUT/tcp $OUT/tcc $OUT/fetch $OUT/write > $OUT/pmc_summary.txt 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# (k_phase<true> is launched twice per batch: the statistics file averages both; the full-size launches alone, from the trace)
python3 - "$OUT" <<'P'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
res = {}
for name, pat in (("k_phase", "k_phase<true>("), ("k_pack_link", "k_pack_link("), ("k_phase_bounds", "k_phase_bounds("), ("k_site_scan", "k_site_scan<")):
    rr = [r for r in rows if pat in r["Kernel_Name"]]
    if not rr:
        continue
    gcol = "Grid_Size_X" if "Grid_Size_X" in rr[0] else "Grid_Size"
    gmax = max(int(r[gcol]) for r in rr)
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rr if int(r[gcol]) == gmax]
    res[name] = {"full_size_launches": len(d), "all_launches": len(rr), "grid": gmax, "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
json.dump(res, open(out + "/full_size_launches.json", "w"), indent=1)
print(json.dumps(res))
P
grep "^{" $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
# keep the merged output small: the raw traces are not needed
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
# the staged pass (the headline `value`): kernel statistics of the command the driver runs, without the CPU legs
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/staged -o run -- python3 $ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 5 --warmup 1 > $OUT/staged.log 2>&1
cd $ROOT
find $OUT/staged -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/staged_kernel_stats.csv
grep "^{" $OUT/staged.log | tail -1 > $OUT/bench_staged_under_rocprof.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
# the feed pass (files -> results: BGZF inflate, CRC-32, record walk, descriptor filter, extract on the device): kernel statistics, then counters of its kernels
cd /tmp
UZ_BENCH_NO_PRODUCT=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/feed -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 > $OUT/feed.log 2>&1
cd $ROOT
find $OUT/feed -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/feed_kernel_stats.csv
grep "^{" $OUT/feed.log | tail -1 > $OUT/bench_feed_under_rocprof.json
cd /tmp
FONLY='--kernel-include-regex k_bam_walk|k_bam_extract|k_desc_filter|k_tab_insert|k_bgzf_crc32|k_bgzf_inflate|k_join_|k_final_'
UZ_BENCH_NO_PRODUCT=1 rocprofv3 $FONLY --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/feedpmc -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 > $OUT/feedpmc.log 2>&1
UZ_BENCH_NO_PRODUCT=1 rocprofv3 $FONLY --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/feedfetch -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 > $OUT/feedfetch.log 2>&1
UZ_BENCH_NO_PRODUCT=1 rocprofv3 $FONLY --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/feedwrite -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 > $OUT/feedwrite.log 2>&1
cd $ROOT
(python3 scripts/pmc_rows.py $OUT/feedpmc; python3 scripts/pmc_rows.py $OUT/feedfetch; python3 scripts/pmc_rows.py $OUT/feedwrite) > $OUT/feed_pmc.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
# the BGZF inflate alone (scripts/inflate_probe.py: every block of the feed pass's BAM in one launch): duration and instruction counters of the
# whole-file launches -> inflate_pmc.json (per 64 KiB block: scalar / vector / branch / LDS / memory instructions; the scalar-issue floor)
cd /tmp
rocprofv3 --kernel-include-regex k_bgzf_inflate --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/inflate -o run -- python3 $ROOT/scripts/inflate_probe.py 20000 > $OUT/inflate.log 2>&1
cd $ROOT
python3 - "$OUT" <<'P'
import csv, collections, glob, json, re, sys
out = sys.argv[1]
f = glob.glob(out + "/inflate/**/*counter_collection.csv", recursive=True)
d = collections.defaultdict(dict)
for r in (csv.DictReader(open(f[0])) if f else []):
    d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    d[r["Dispatch_Id"]]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
log = open(out + "/inflate.log").read()
m = re.search(r"BAM: (\d+) records, ([\d.]+) MB -> ([\d.]+) MB, (\d+) blocks", log)
g = re.search(r"device: (\d+) blocks, ([\d.]+) ms per launch = ([\d.]+) GB/s", log)
res = {"probe": g.group(0) if g else None}
if d and m:
    blocks = int(m.group(4))
    big = max(v.get("SQ_INSTS_SALU", 0) for v in d.values())
    full = [v for v in d.values() if v.get("SQ_INSTS_SALU", 0) > 0.9 * big]
    mean = lambda k: sum(v[k] for v in full) / len(full)
    per = {k.replace("SQ_INSTS_", "").lower() + "_per_block": round(mean(k) / blocks, 1) for k in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")}
    ns = mean("ns")
    cyc_per_simd_block = ns * 1e-9 * 2.35e9 * 1024 / blocks
    res.update(blocks=blocks, whole_file_launches=len(full), launch_ms_under_counters=round(ns / 1e6, 2), out_GB=float(m.group(3)) / 1e3, **per,
               simd_cycles_per_block=round(cyc_per_simd_block), scalar_issue_floor_frac=round(per["salu_per_block"] * 4 / cyc_per_simd_block, 3),
               note="one wavefront per block, 8 per SIMD; a SIMD issues one scalar instruction per four cycles: salu_per_block x 4 against the cycles a SIMD spends per block (2.35 GHz assumed)")
json.dump(res, open(out + "/inflate_pmc.json", "w"), indent=1)
print(json.dumps(res))
P
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete


# one resident step as a timeline (every kernel and copy in order, the gaps between them): resident_timeline.json
cd $ROOT
bash scripts/resident_timeline.sh > $OUT/resident_timeline.txt 2>&1
cp gpurun_out/resident_timeline.json $OUT/resident_timeline.json 2>/dev/null
# the driver's command, last: the bench line of the round
cd $ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.log 2> $OUT/bench_line.err
grep "^{" $OUT/bench_line.log | tail -1 > $OUT/bench_line.json
python3 bench.py --no-cpu --no-config5 --feed-dnms 0 --dnms 12500 --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > $OUT/bench_line_shard12500.json
du -sh $OUT
