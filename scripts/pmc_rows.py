"""Print per-kernel averages of a rocprofv3 --pmc counter_collection.csv (diagnostics): per launch the counter's rows summed, then the mean over launches."""
import csv, glob, sys, collections
root = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat in k:
            d = acc[k][row["Counter_Name"]]
            key = (f, row["Dispatch_Id"])
            d[key] = d.get(key, 0.0) + float(row["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s launches=%d mean=%.6g" % (c, len(v), sum(v.values()) / len(v)))
