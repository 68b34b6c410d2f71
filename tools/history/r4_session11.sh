#!/bin/bash
# Round 4, GPU session 11: fused un-shuffle + RGB branch in the feat_layers epilogue -- parity, then B = 1 / B = 7 traces.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s11
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
bash tools/n1_trace.sh r4s11/b1_graph --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
N1_GRAPH=0 bash tools/n1_trace.sh r4s11/b1_eager --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
bash tools/n1_trace.sh r4s11/b7 --batch 7 --iters 5 > /dev/null 2>&1
for n in b1_graph b1_eager b7; do echo "== $n"; cat $O/$n/wall.log | grep "N1 B"; cat $O/$n/launches.txt | grep -v "torch:"; done
rm -rf $O/*/prof
echo done
