#!/bin/bash
# Who burns the host cores while the cfg2b step runs?  (cpu_ms 1.7 x ms_per_step.)
#   tools/host_spin_probe.sh <name> [extra bench args...]
# Starts bench.py in the background, waits until the timed steps run, then samples per-thread CPU time, state and kernel wait
# channel from /proc/<pid>/task/* twice, 5 s apart (no debugger: /proc only).  PROBE_STACKS=1 adds native backtraces (rocgdb).
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
    echo "$(basename $t) $(cat $t/comm 2>/dev/null | tr ' ' '_') ${f[13]} ${f[14]} ${f[2]} $(cat $t/wchan 2>/dev/null || echo -)"; done; }
threads > $O/t0.txt; sleep 5; threads > $O/t1.txt
python - $O/t0.txt $O/t1.txt <<'P' > $O/threads.txt
import sys
a = {l.split()[0]: l.split() for l in open(sys.argv[1])}
b = {l.split()[0]: l.split() for l in open(sys.argv[2])}
rows = []
for tid, r in b.items():
    if tid in a:
        du, ds = int(r[2]) - int(a[tid][2]), int(r[3]) - int(a[tid][3])
        rows.append((du + ds, tid, r[1], du, ds, r[4], r[5] if len(r) > 5 else "-"))
print("# per-thread CPU over 5.0 s (clock ticks of 10 ms): total tid comm user sys state wchan; %d threads" % len(b))
for r in sorted(rows, reverse=True)[:12]:
    print("%5d %8s %-20s user %4d sys %4d  = %.2f cores  state %s wchan %s" % (r[0], r[1], r[2], r[3], r[4], r[0] / 500.0, r[5], r[6]))
P
cat $O/threads.txt
if [ "${PROBE_STACKS:-0}" = 1 ]; then
  for i in 1 2 3; do timeout 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "thread apply all bt 14" > $O/bt_$i.txt 2>&1; sleep 1; done
fi
wait $PID
echo "bench rc=$?"
python - $O/bench.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.1f  host %s  sync %s" % (d["ms_per_step"], [(e["leg"], round(e["ms"], 1), round(e["cpu_ms"], 1), round(e["cpu_share"], 2)) for e in d["host"]["host_enqueue_ms_per_step"]], d["host"].get("sync")))
P
