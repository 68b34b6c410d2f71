"""Idle time inside a training step from a rocprofv3 kernel_trace.csv: steps are delimited by the fwd16 / fwd3 training
kernel; for the LAST full step prints every gap > 2 us between consecutive kernels (end -> next start), the idle total and
the kernel-time total.   python tools/step_gaps.py kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if re.search(r"fwd(16|3)_kernel<true>", r["Kernel_Name"])]
if len(marks) < 2:
    sys.exit("need at least two training-forward launches")
a, b = marks[-2], marks[-1]
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").replace("gnr::", "")[:60]
idle = busy = 0
prev = None
for r in rows[a:b + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None:
        gap = st - prev[1]
        if gap > 2000:
            print("gap %8.1f us   after %-50s before %s" % (gap / 1e3, short(prev[0]), short(r["Kernel_Name"])))
        if gap > 0:
            idle += gap
    if r is not rows[b]:
        busy += en - st
    prev = (r["Kernel_Name"], max(en, prev[1]) if prev else en)
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print("step span %.3f ms, kernel time %.3f ms, idle %.3f ms, %d launches" % (span / 1e6, busy / 1e6, idle / 1e6, b - a))
