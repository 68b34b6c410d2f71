"""Lists the launches of the last iteration in a rocprofv3 kernel_trace.csv of tools/n1_trace.py, in launch order:
name, grid, duration, gap to the previous kernel's end."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# iteration starts: the rgb_conv_kernel launch on the input feature map (the smallest grid of that kernel)
gx = [int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) for r in rows]
rgb = [i for i, n in enumerate(names) if "rgb_conv_kernel" in n or "rgb_conv_split_kernel" in n]
gmin = min(gx[i] for i in rgb) if rgb else 0
starts = [i for i in rgb if gx[i] == gmin]
s = starts[-1] if starts else 0
prev_end = None
tot = 0
for r in rows[s:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("gnr::", "")
    if n.startswith("at::native"):
        n = "torch:" + n.split("<")[0].split("::")[-1]
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    print("%-46s grid %-18s %8.1f us  gap %6.1f" % (n[:46], grid, (en - st) / 1e3, gap))
    tot += en - st
    prev_end = en
print("kernel time %.1f us over %d launches; span %.1f us" % (tot / 1e3, len(rows) - s, (int(rows[-1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3))
