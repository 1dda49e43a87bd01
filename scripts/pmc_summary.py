"""Per-kernel averages of rocprofv3 --pmc passes -> one JSON (kept under profiles/).
usage: pmc_summary.py OUT.json DIR [DIR ...]   (every *counter_collection.csv below the DIRs is read)
Per launch: the counter value summed over the rows of one dispatch (rocprofv3 writes one row per counter per dispatch);
reported: the mean over the launches of each kernel, and the launch count."""
import collections
import csv
import glob
import json
import sys

KERNELS = {"k_phase": "k_phase<true>(", "k_phase_hbm": "k_phase<false>(", "k_phase_r1": "k_phase(", "k_seg_qc": "k_seg_qc(", "k_site_scan": "k_site_scan<", "k_mark_ranges": "k_mark_ranges(",
           "k_phase_bounds": "k_phase_bounds(", "k_window": "k_window<", "k_pack_rec": "k_pack_rec<", "k_pack_link": "k_pack_link("}


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> dispatch -> value
    grid = collections.defaultdict(dict)  # kernel -> dispatch -> grid size
    meta = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                for short, pat in KERNELS.items():
                    if pat in name:
                        key = (f, row["Dispatch_Id"])
                        c = acc[short][row["Counter_Name"]]
                        c[key] = c.get(key, 0.0) + float(row["Counter_Value"])
                        grid[short][key] = int(row.get("Grid_Size") or 0)
                        meta[short] = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                                       "scratch": int(row["Scratch_Size"]), "workgroup": int(row["Workgroup_Size"]), "full_name": name[:120]}
    out = {}
    for short, counters in acc.items():
        out[short] = dict(meta[short])
        # the read stage launches k_phase<true> twice per batch since round 5 (the whole batch, then the few DNMs it gave up, with larger arenas):
        # the counters of a kernel are those of its FULL-SIZE launches (the largest grid seen)
        gmax = max(grid[short].values()) if grid[short] else 0
        full = {k for k, g in grid[short].items() if g == gmax}
        out[short]["grid_size"] = gmax
        out[short]["counters_per_launch"] = {}
        for c, v in sorted(counters.items()):
            vv = {k: x for k, x in v.items() if k in full} or v
            out[short]["counters_per_launch"][c] = {"mean": sum(vv.values()) / len(vv), "launches": len(vv)}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
