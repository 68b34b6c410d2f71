"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, d))
    for k, cs in agg.items():
        if "gnr::" not in k:
            continue
        print("  %-42s n=%d" % (k, len(next(iter(cs.values())))), " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
