#!/bin/bash
# Round 4, GPU session 31: conv16_blur_lds_kernel -- one or two k-blocks of global loads in flight (GNR_BLUR_LDS_DEPTH).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s31
mkdir -p $O
export TMPDIR=/tmp GNR_ALLOW_EXPERIMENTAL_LIB=1
cd $R
for d in 2 1; do
  if [ $d = 1 ]; then python -m gazenerf_amd.build --no-torch-ext > $O/d$d.build.log 2>&1
  else GNR_EXTRA_FILES="gnr_conv16.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_BLUR_LDS_DEPTH=$d" python -m gazenerf_amd.build --no-torch-ext > $O/d$d.build.log 2>&1; fi
  echo "== depth $d"
  timeout 600 python -m pytest tests/test_upsample.py -m gpu -x -q 2>&1 | tail -1
  for i in 1 2; do
    bash tools/n1_trace.sh r4s31/d${d}_$i --batch 7 --iters 5 > /dev/null 2>&1
    grep -E "blur_lds|kernel time" $O/d${d}_$i/launches.txt | awk '{printf "%s %s | ", $1, $(NF-3)}'; echo
  done
  rm -rf $O/*/prof
done
