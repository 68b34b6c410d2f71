#!/bin/bash
# PMC counters of the upsampler's GEMM launches (separate rocprofv3 --pmc passes, kernel-trace only), per dispatch of
# the last forward: SQ busy / wait / MFMA-busy shares, HBM-side bytes.   usage: tools/n1_pmc.sh <name> [n1_trace.py args]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $R/tools/n1_trace.py "$@" > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python $R/tools/n1_trace.py "$@" > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python $R/tools/n1_trace.py "$@" > $OUT/write.log 2>&1
python $R/tools/n1_pmc_list.py $OUT | tee $OUT/pmc.txt
