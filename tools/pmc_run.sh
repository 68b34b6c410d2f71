#!/bin/bash
# PMC collection on the GPU box: separate passes (SQ / FETCH / WRITE), kernel-trace only.
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <bench args...>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $R/bench.py "$@" --no-cpu-baseline --no-one-call > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python $R/bench.py "$@" --no-cpu-baseline --no-one-call > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python $R/bench.py "$@" --no-cpu-baseline --no-one-call > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python $R/bench.py "$@" --no-cpu-baseline --no-one-call > $OUT/sq2.log 2>&1
ls -R $OUT | head -40
