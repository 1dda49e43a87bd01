"""Print per-kernel averages of a rocprofv3 --pmc counter_collection.csv (diagnostics)."""
import csv, glob, sys, collections
root = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
