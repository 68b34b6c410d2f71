#!/bin/bash
# Round 4, GPU session 20: L2 pre-touch in the blur-fused feat_layers GEMMs (GNR_C16_TOUCH_AHEAD): A/B of the distance, then tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s20
mkdir -p $O
export TMPDIR=/tmp
cd $R
rm -f gpurun_out/ab/n1_summary.txt
for d in 0 2 3 5; do
  bash tools/ab_n1.sh touch$d "-DGNR_C16_TOUCH_AHEAD=$d" 2>&1 | tail -1
done
cp gpurun_out/ab/n1_summary.txt $O/touch_ab.txt
# back to the shipped build
python -m gazenerf_amd.build > $O/rebuild.log 2>&1; tail -1 $O/rebuild.log
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q 2>&1 | tail -2
bash tools/n1_trace.sh r4s20/b7 --batch 7 --iters 5 > /dev/null 2>&1
grep -E "N1 B" $O/b7/wall.log; grep -E "true>|kernel time" $O/b7/launches.txt
rm -rf $O/b7/prof gpurun_out/ab/*/prof
echo done
