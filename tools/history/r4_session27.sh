#!/bin/bash
# Round 4, GPU session 27: idle / low-activity time in front of the training forward in the cfg4 step (rocprofv3 kernel trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s27
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o run -- python $R/bench.py --config cfg4 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
cd $R
python tools/step_gaps.py $O/kernel_trace.csv | tee $O/gaps.txt
python - <<'P'
import csv, re
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r4s27/kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if "fwd16_kernel<true>" in r["Kernel_Name"]]
b=marks[-1]
# the 40 launches in front of the last training forward: name, duration, grid
t_end=int(rows[b]["Start_Timestamp"])
print("launches in the 3 ms before the training forward (start offset us, duration us, workgroups, name):")
for r in rows[max(0,b-80):b]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    if t_end-st < 3_000_000:
        wg=int(r["Grid_Size_X"])*int(r.get("Grid_Size_Y",1) or 1)*int(r.get("Grid_Size_Z",1) or 1)//max(1,int(r["Workgroup_Size_X"])*int(r.get("Workgroup_Size_Y",1) or 1)*int(r.get("Workgroup_Size_Z",1) or 1))
        print("%9.1f %8.1f %7d  %s" % ((st-t_end)/1e3,(en-st)/1e3,wg,re.sub(r"\(.*","",r["Kernel_Name"])[:90]))
P
