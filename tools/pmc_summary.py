"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        full = r["Kernel_Name"].split("(")[0]
        if "gnr::" not in full:
            continue
        k = full[full.index("gnr::"):][:56]          # (the last 40 characters until round 5: the longer template lists lost their "gnr::")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, d))
    for k, cs in agg.items():
        print("  %-56s n=%d" % (k, len(next(iter(cs.values())))), " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
