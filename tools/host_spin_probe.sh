#!/bin/bash
# Who burns the host cores while the cfg2b step runs?  (cpu_ms 1.7 x ms_per_step.)
#   tools/host_spin_probe.sh <name> [extra bench args...]
# Starts bench.py in the background, waits until the timed steps run, then samples (a) per-thread CPU time from
# /proc/<pid>/task/*/stat twice, 5 s apart, and (b) native backtraces of every thread with rocgdb, 4 times.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
O=$R/gpurun_out/$NAME
mkdir -p $O
cd $R
python bench.py --steps 30 --warmup 2 --no-alt --no-cpu-baseline --no-one-call "$@" > $O/bench.json 2> $O/bench.err &
PID=$!
# the pre-flight line marks the end of the imports; the workload is built and the two warm-up steps run in the next seconds
for i in $(seq 1 300); do grep -q hipMemGetInfo $O/bench.err 2>/dev/null && break; sleep 1; done
sleep ${PROBE_DELAY:-10}
threads() { for t in /proc/$PID/task/*; do
    read -r -a f < $t/stat 2>/dev/null || continue
    echo "$(basename $t) $(cat $t/comm 2>/dev/null | tr ' ' '_') ${f[13]} ${f[14]}"; done; }
threads > $O/t0.txt; sleep 5; threads > $O/t1.txt
python - $O/t0.txt $O/t1.txt <<'P' > $O/threads.txt
import sys
a = {l.split()[0]: l.split() for l in open(sys.argv[1])}
b = {l.split()[0]: l.split() for l in open(sys.argv[2])}
rows = []
for tid, r in b.items():
    if tid in a:
        du, ds = int(r[2]) - int(a[tid][2]), int(r[3]) - int(a[tid][3])
        rows.append((du + ds, tid, r[1], du, ds))
print("# per-thread CPU over 5.0 s (clock ticks of 10 ms): total tid comm user sys")
for r in sorted(rows, reverse=True)[:12]:
    print("%5d %8s %-20s user %4d sys %4d  = %.2f cores" % (r[0], r[1], r[2], r[3], r[4], r[0] / 500.0))
P
cat $O/threads.txt
for i in 1 2 3 4; do
  timeout 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "thread apply all bt 14" > $O/bt_$i.txt 2>&1
  sleep 1
done
wait $PID
echo "bench rc=$?"
tail -c 400 $O/bench.json
# the hot threads' stacks, condensed
python - $O <<'P'
import re, sys, glob, collections
top = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + "/bt_*.txt")):
    txt = open(f).read()
    for blk in re.split(r"\nThread \d+ ", txt)[1:]:
        frames = re.findall(r"#\d+\s+(?:0x[0-9a-f]+ in )?([^\s(]+)", blk)
        top[" < ".join(frames[:7])] += 1
for k, v in top.most_common(14):
    print(v, k[:400])
P
