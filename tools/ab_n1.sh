#!/bin/bash
# Timing experiments on the upsampler's GEMM: rebuild gnr_conv16.hip with extra -D switches (results may be WRONG), run
# the upsampler forward at B = 7 under the kernel trace, keep the conv16 launch times.
# usage: tools/ab_n1.sh <tag> "<flags>" [n1_trace.py args]
set -u
export GNR_ALLOW_EXPERIMENTAL_LIB=1      # _lib.load() refuses a library built with timing switches otherwise
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FLAGSX=$2; shift 2
mkdir -p $R/gpurun_out/ab
cd $R
GNR_EXTRA_FILES="gnr_conv16.hip" GNR_EXTRA_HIPCC_FLAGS="$FLAGSX" python -m gazenerf_amd.build --no-torch-ext > gpurun_out/ab/$TAG.build.log 2>&1
tools/n1_trace.sh ab/n1_$TAG --batch 7 --iters 3 --fwd-only "$@" > /dev/null 2>&1
echo "$TAG: $(grep N1 gpurun_out/ab/n1_$TAG/wall.log | cut -c1-40) | $(grep conv16_kernel gpurun_out/ab/n1_$TAG/launches.txt | awk '{printf "%s ", $(NF-3)}')" | tee -a gpurun_out/ab/n1_summary.txt
