#!/bin/bash
# Round 4, GPU session 4: where do the forward's extra ~300 W come from (training forward vs its inference kernel vs a build
# without the per-layer dumps), and the bench line through a one-rank RCCL group.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s4
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-one-call --force-dist > $O/bench_forced_rccl_ws1.json 2> $O/bench_forced_rccl_ws1.err
tail -c 400 $O/bench_forced_rccl_ws1.json
timeout 200 python tools/smi_sample.py $O/smi_fwd_save.csv -- python tools/stage_loop.py fwd --seconds 8 2>> $O/err.txt | tee -a $O/fwd_power.txt
timeout 200 python tools/smi_sample.py $O/smi_fwd_nosave.csv -- python tools/stage_loop.py fwd --seconds 8 --nosave 2>> $O/err.txt | tee -a $O/fwd_power.txt
export GNR_ALLOW_EXPERIMENTAL_LIB=1
for v in "nodump:-DGNR_ABL16=2" "nofeat:-DGNR_ABL16=4" "burst8:-DGNR_DUMP_BURST=8" "temporal:-DGNR_TEMPORAL_DUMP_TIMING=1"; do
  tag=${v%%:*}; fl=${v#*:}
  GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="$fl" python -m gazenerf_amd.build --no-torch-ext > $O/build_$tag.log 2>&1
  echo "== $tag ($fl)" | tee -a $O/fwd_power.txt
  timeout 200 python tools/smi_sample.py $O/smi_fwd_$tag.csv -- python tools/stage_loop.py fwd --seconds 8 2>> $O/err.txt | tee -a $O/fwd_power.txt
done
echo done
