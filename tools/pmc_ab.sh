#!/bin/bash
# usage: run_pmc.sh <tag> <env assignments...>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
env "$@" python $R/tools/gpu_fwd_time.py > $OUT/fwd_time.log 2>&1
timeout 600 env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-one-call > $OUT/sq.log 2>&1
timeout 600 env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-one-call > $OUT/sq2.log 2>&1
timeout 600 env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sqf -o sqf -- python $R/bench.py --steps 1 --warmup 0 --mode fwd --no-cpu-baseline --no-alt --no-one-call > $OUT/sqf.log 2>&1
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
timeout 600 env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-one-call > $OUT/fetch.log 2>&1
timeout 600 env "$@" rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-one-call > $OUT/write.log 2>&1
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
