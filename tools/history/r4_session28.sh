#!/bin/bash
# Round 4, GPU session 28: does ANY long different load in front of the training forward cost it its clock, or only our backward?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s28
mkdir -p $O
cd $R
{
timeout 200 python tools/stage_loop.py fwd --rays 8192 --seconds 4 2>&1 | grep stage_loop
timeout 200 python tools/stage_loop.py alt --rays 8192 --seconds 5 2>&1 | grep stage_loop
for reps in 30 250; do timeout 200 python tools/stage_loop.py fwd --rays 8192 --seconds 5 --pre gemm --pre-reps $reps 2>&1 | grep stage_loop; done
for reps in 30 400; do timeout 200 python tools/stage_loop.py fwd --rays 8192 --seconds 5 --pre copy --pre-reps $reps 2>&1 | grep stage_loop; done
timeout 200 python tools/stage_loop.py bwd --rays 8192 --seconds 4 2>&1 | grep stage_loop
} | tee $O/pre.txt
