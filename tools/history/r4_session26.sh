#!/bin/bash
# Round 4, GPU session 26: what costs the training forward its clock after a backward -- power or a low-activity window in front of it?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s26
mkdir -p $O
cd $R
{
for g in 0 50 200 1000 5000 20000; do timeout 200 python tools/stage_loop.py fwd --rays 8192 --seconds 4 --gap-us $g 2>&1 | grep stage_loop; done
for pre in none gemm copy; do timeout 200 python tools/stage_loop.py alt --rays 8192 --seconds 5 --pre $pre 2>&1 | grep stage_loop; done
for g in 0 200 5000; do timeout 200 python tools/stage_loop.py fwd --rays 8192 --seconds 4 --gap-us $g --nosave 2>&1 | grep stage_loop; done
} | tee $O/gap.txt
