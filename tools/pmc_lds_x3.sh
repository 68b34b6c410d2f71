#!/bin/bash
# LDS-pipe utilisation of the bf16x3 chain kernels at the bench's launch size: is the structure
# "LDS-bandwidth-bound by construction" or parked in s_waitcnt / barrier?  Separate --pmc passes, --kernel-trace only.
# usage (GPU box): tools/pmc_lds_x3.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*LDS[A-Z0-9_]*\|SQ_BUSY_CU_CYCLES\|SQ_ACTIVE_INST_LDS\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAIT_INST_LDS" | sort -u > $OUT/lds_counters_available.txt
ARGS="--precision bf16x3 --steps 1 --warmup 1 --no-cpu-baseline --no-one-call --no-alt"
pass() {   # name, counters...
  local n=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python $R/bench.py $ARGS > $OUT/$n.log 2>&1
  echo "pass $n rc $?"
}
pass a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass b SQ_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass c SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
python $R/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
