#!/bin/bash
# One GPU session on a gpurun box: named steps, run in the order given, everything under gpurun_out/<name>/.
#   tools/session.sh <name> <step> [<step> ...]
# (replaces the per-session scripts of rounds 1-4, kept under tools/history/).  Steps:
#   tests            python -m pytest tests -m gpu -x -q                          -> pytest.log
#   tests:<expr>     ... -k <expr>                                                -> pytest_k.log
#   smoke            __graft_entry__.smoke()
#   n1               upsampler alone under the kernel trace, B = 1 graph + B = 7  -> n1_launches.txt
#   n1pmc            per-launch counters of the B = 7 forward+backward            -> n1_pmc.txt
#   bench            python bench.py --steps 20 --warmup 5 (the driver's command) -> bench_default.json
#   bench_fwd        python bench.py --mode fwd --steps 5 --warmup 2              -> bench_fwd.json
#   cfg4, cfg4x3     python bench.py --config cfg4 [--precision bf16x3]           -> bench_cfg4[_bf16x3].json
#   stats            rocprofv3 --kernel-trace --stats of the default bench        -> stats/kernel_stats.csv
#   configs          tools/run_configs.py under rocprofv3 --kernel-trace --stats  -> configs.txt, configs_kernel_stats.csv
#   sq               SQ counters of the hot-path kernels at the bench's tile      -> sq/summary.txt
#   traffic          tools/pmc_capture.py (FETCH_SIZE / WRITE_SIZE per launch)    -> traffic.txt
#   rccl1            the forced one-rank RCCL bench lines                         -> bench_forced_rccl_ws1*.json
#   ws8              bench.py --gpus 8 on one GPU over gloo (cfg4 + strong)       -> bench_ws8_*.json
#   ws8diag:<n>      n repeats of the FULL-SIZE 8-rank cfg4 rehearsal (the run that went red on the driver's box in round 5), then n at
#                    the test-only --side 16, every rank's stderr kept; a failed repeat's cause lines -> ws8diag.txt
#   guard            tests/guard/run_guarded.py, long forms (every scenario, both bindings, with and without --poison) -> guard.txt
#   canary           libgnr.so rebuilt with -DGNR_CANARY=1 (gaps with a pattern between the carved regions of every workspace, filled before and
#                    compared after every entry point: csrc/gnr_canary.h), the two fuzzers + the parity / upsampler / network tests on it, then
#                    -DGNR_CANARY=2 (a deliberate 64-float overrun inside a GEMM's scratch) must FAIL; product build restored   -> canary.txt
#   softstart:<n,n,..>  the training forward's clock after a backward with the first-round workgroups' start RAMPED (-DGNR_SOFTSTART=n on
#                    gnr_fwd16.hip, experimental builds; 0 = the product build): tools/stage_loop.py alt at 8192 rays + bench.py --config cfg4 -> softstart.txt
#   hostspin[:args]  tools/host_spin_probe.sh: per-thread CPU time / state / wait channel of the default bench while its steps run -> hostspin/
#   syncab           the default bench under --sync auto | blocking | yield, 6 steps each: ms_per_step and host.cpu_share   -> syncab.txt
#   noise            tests/diagnostics/grad_noise_draws.py                        -> grad_noise_draws.txt
#   dropterm         the bf16x3 gate on a build with one cross term dropped       -> grad_gate_dropped_term.txt
#   x3outlier        anatomy of the gate's outlier draw (bf16x3 vs fp32 kernels)  -> x3_outlier_draw13.txt
#   abn1:<tag>:<flags>  rebuild gnr_conv16.hip + gnr_upsample.hip with extra -D flags (experimental build), N1 B = 7 trace,
#                    restore the product build                                    -> ab_<tag>_launches.txt
#   n1tile:<mt,nt>   N1 B = 7 trace with a GEMM instance pinned (gnr_set_conv16_tile)   -> tile_<mt>x<nt>_launches.txt
#   x3energy         J per step of the bf16x3 leg (tools/x3_energy.py)            -> x3_energy.txt
#   x3ab:<tag>:<flags>  the same on gnr_fwd3.hip / gnr_bwd3.hip rebuilt with extra -D flags -> x3_energy_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
O=$R/gpurun_out/$NAME
mkdir -p $O
export TMPDIR=/tmp
cd $R
jline() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'value %.1f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'up', d.get('upsampler', {}).get('fwdbwd_ms'),
          'frac %.3f' % d['roofline']['frac'], 'step_frac', d['roofline'].get('step_frac'), 'enq/cpu', [(round(e['ms'], 2), round(e.get('cpu_ms', 0), 2)) for e in d.get('host', {}).get('host_enqueue_ms_per_step', [])], d.get('build'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
P
}
for STEP in "$@"; do
  echo "=== $STEP ($(date +%H:%M:%S))"
  case $STEP in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log;;
    tests:*) timeout 1500 python -m pytest tests -m gpu -x -q -k "${STEP#tests:}" > $O/pytest_k.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_k.log; tail -15 $O/pytest_k.log;;
    smoke) python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2;;
    n1)
      bash tools/n1_trace.sh $NAME/b1_graph --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
      bash tools/n1_trace.sh $NAME/b7 --batch 7 --iters 5 > /dev/null 2>&1
      { echo "# tools/session.sh $NAME n1 ($(python -c 'from gazenerf_amd import _lib; print(_lib.build_info())')): tools/n1_trace.sh <name> --batch 1 --iters 9 --fwd-only (HIP-graph replay) and --batch 7 --iters 5"
        for n in b1_graph b7; do echo "== $n"; grep "N1 B" $O/$n/wall.log; grep -v "torch:" $O/$n/launches.txt; done; } > $O/n1_launches.txt
      grep -E "N1 B|kernel time" $O/n1_launches.txt; rm -rf $O/*/prof;;
    n1pmc)
      bash tools/n1_pmc.sh $NAME/n1pmc --batch 7 --iters 2 > /dev/null 2>&1
      { echo "# tools/session.sh $NAME n1pmc ($(python -c 'from gazenerf_amd import _lib; print(_lib.build_info())')): tools/n1_pmc.sh --batch 7 --iters 2 -- every launch of the LAST forward+backward;"
        echo "# SQ counters, FETCH_SIZE (as counted: the gfx950 x2 for 16-B/lane streaming reads is not applied) and WRITE_SIZE from three separate --pmc passes, --kernel-trace only"
        cat $O/n1pmc/pmc.txt; } > $O/n1_pmc.txt
      rm -rf $O/n1pmc/sq $O/n1pmc/fetch $O/n1pmc/write; tail -n +3 $O/n1_pmc.txt | cut -c1-200;;
    bench) timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; jline $O/bench_default.json;;
    bench_fwd) timeout 900 python bench.py --mode fwd --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fwd.json 2> $O/bench_fwd.err; jline $O/bench_fwd.json;;
    cfg4) timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; jline $O/bench_cfg4.json;;
    cfg4x3) timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --precision bf16x3 > $O/bench_cfg4_bf16x3.json 2> $O/bench_cfg4_bf16x3.err; jline $O/bench_cfg4_bf16x3.json;;
    stats) bash tools/prof_stats.sh $NAME/stats > /dev/null 2>&1; rm -rf $O/stats/prof; head -12 $O/stats/kernel_stats.csv | cut -c1-160;;
    configs)
      ( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfgprof -o run -- python $R/tools/run_configs.py --reps 3 > $O/configs.txt 2> $O/configs.err )
      find $O/cfgprof -name "*kernel_stats.csv" -exec cp {} $O/configs_kernel_stats.csv \; ; rm -rf $O/cfgprof; cat $O/configs.txt | tail -20;;
    sq) bash tools/pmc_ab.sh $NAME/sq X=1 > /dev/null 2>&1; rm -rf $O/sq/sq $O/sq/sq2 $O/sq/sqf $O/sq/fetch $O/sq/write; head -60 $O/sq/summary.txt | cut -c1-220;;
    traffic) timeout 1500 python tools/pmc_capture.py $NAME/traffic > $O/traffic.txt 2>&1; tail -25 $O/traffic.txt | cut -c1-200;;
    rccl1)
      GNR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-one-call > $O/bench_forced_rccl_ws1.json 2> $O/bench_forced_rccl_ws1.err; jline $O/bench_forced_rccl_ws1.json
      GNR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_forced_rccl_ws1_cfg4.json 2> $O/bench_forced_rccl_ws1_cfg4.err; jline $O/bench_forced_rccl_ws1_cfg4.json;;
    ws8)
      GNR_BENCH_BACKEND=gloo GNR_BENCH_DEVICE=0 timeout 1200 python bench.py --gpus 8 --config cfg4 --steps 5 --warmup 2 > $O/bench_ws8_gloo_cfg4.json 2> $O/bench_ws8_gloo_cfg4.err; jline $O/bench_ws8_gloo_cfg4.json
      GNR_BENCH_BACKEND=gloo GNR_BENCH_DEVICE=0 timeout 1200 python bench.py --gpus 8 --scaling strong --micro 4096 --steps 3 --warmup 1 --no-alt > $O/bench_ws8_gloo_strong.json 2> $O/bench_ws8_gloo_strong.err; jline $O/bench_ws8_gloo_strong.json
      grep -v "^\[W\|^W0\|^\*\*\*" $O/bench_ws8_gloo_cfg4.err | head -60 > $O/bench_ws8_gloo_cfg4_stderr_head.txt;;
    ws8diag:*)
      N=${STEP#ws8diag:}
      { echo "# tools/session.sh $NAME ws8diag:$N ($(python -c 'from gazenerf_amd import _lib; print(_lib.build_info())')): bench.py --gpus 8 --config cfg4 [--side 16] --steps 2 --warmup 1, eight ranks on one GPU over gloo"
        for SIDE in 64 16; do for i in $(seq 1 $N); do
          GNR_BENCH_BACKEND=gloo GNR_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --config cfg4 --side $SIDE --steps 2 --warmup 1 > $O/ws8diag_${SIDE}_$i.json 2> $O/ws8diag_${SIDE}_$i.err
          RC=$?
          echo "side $SIDE repeat $i rc=$RC $(python -c "import json,sys; d=json.loads(open('$O/ws8diag_${SIDE}_$i.json').read().strip().splitlines()[-1]); print('ms/step %.2f' % d['ms_per_step'], 'allreduce ms %.2f' % d['allreduce']['ms'])" 2>/dev/null)"
          grep "peak .* GiB allocated" $O/ws8diag_${SIDE}_$i.err | sed 's/^bench.py: /    /' | head -8       # per-rank memory: 8 ranks share ONE device here
          if [ $RC -ne 0 ]; then grep -n "failed:\|Error\|error\|hipMemGetInfo\|HSA_STATUS\|out of memory" $O/ws8diag_${SIDE}_$i.err | head -40; else rm -f $O/ws8diag_${SIDE}_$i.err; fi
        done; done; } > $O/ws8diag.txt 2>&1
      cat $O/ws8diag.txt | cut -c1-300;;
    guard)
      { echo "# tools/session.sh $NAME guard ($(python -c 'from gazenerf_amd import _lib; print(_lib.build_info())')): tests/guard/run_guarded.py <scenario> ... -- one JSON line per run"
        for B in ctypes torch_ext; do for P in "" "--poison"; do
          for SC in hot_path upsample; do timeout 1500 python tests/guard/run_guarded.py $SC --binding $B $P --cases 40 --seed 11 2> $O/guard_${SC}_${B}${P}.err; done
          for SC in many_images aux network_step; do timeout 1500 python tests/guard/run_guarded.py $SC --binding $B $P 2> $O/guard_${SC}_${B}${P}.err; done
        done; done; } > $O/guard.txt 2>&1
      python - $O/guard.txt <<'P'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["scenario"], d["binding"], "poison" if d["poison"] else "-", "ok" if d.get("ok") else "FAIL", "violations", d["violations"], "calls checked", d["calls_checked"],
              "allocations", d["allocations"], "peak GiB %.1f" % (d["peak_live_bytes"] / 2**30), d.get("error", ""))
P
      ;;
    syncab)
      { echo "# tools/session.sh $NAME syncab: python bench.py --steps 6 --warmup 2 --no-alt --no-cpu-baseline --no-one-call --sync <mode>; then cfg4 --steps 10 --warmup 3"
        # the HIP runtime's own knobs for the same wait (names from `strings libamdhip64.so`; semantics to be read off the numbers)
        for E in "DEBUG_CLR_MAX_BATCH_SIZE=100000" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=100000" "ROC_ACTIVE_WAIT_TIMEOUT=0" "ROC_CPU_WAIT_FOR_SIGNAL=1" "ROC_SIGNAL_POOL_SIZE=4096" "HIP_FORCE_DEV_KERNARG=1" "ROC_AQL_QUEUE_SIZE=65536"; do
          env $E timeout 600 python bench.py --steps 6 --warmup 2 --no-alt --no-cpu-baseline --no-one-call > $O/syncab_env.json 2> $O/syncab_env.err
          python -c "
import json; d=json.loads(open('$O/syncab_env.json').read().strip().splitlines()[-1]); e=d['host']['host_enqueue_ms_per_step'][0]
print('cfg2b env %-40s ms/step %8.2f  enqueue ms %8.2f  cpu ms %8.2f  cpu_share %.2f' % ('$E', d['ms_per_step'], e['ms'], e['cpu_ms'], e['cpu_share']))" 2>&1 | tail -1
        done
        for M in auto blocking yield auto blocking; do
          timeout 600 python bench.py --steps 6 --warmup 2 --no-alt --no-cpu-baseline --no-one-call --sync $M > $O/syncab_$M.json 2> $O/syncab_$M.err
          python -c "
import json; d=json.loads(open('$O/syncab_$M.json').read().strip().splitlines()[-1]); e=d['host']['host_enqueue_ms_per_step'][0]
print('cfg2b sync %-8s ms/step %8.2f  enqueue ms %8.2f  cpu ms %8.2f  cpu_share %.2f  rc %s' % ('$M', d['ms_per_step'], e['ms'], e['cpu_ms'], e['cpu_share'], d['host']['sync']['hipSetDevice_hipSetDeviceFlags_rc']))"
        done
        for M in auto blocking; do
          timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --sync $M > $O/syncab_cfg4_$M.json 2> $O/syncab_cfg4_$M.err
          python -c "
import json; d=json.loads(open('$O/syncab_cfg4_$M.json').read().strip().splitlines()[-1]); e=d['host']['host_enqueue_ms_per_step'][0]
print('cfg4  sync %-8s ms/step %8.2f  enqueue ms %8.2f  cpu ms %8.2f  cpu_share %.2f' % ('$M', d['ms_per_step'], e['ms'], e['cpu_ms'], e['cpu_share']))"
        done; } > $O/syncab.txt 2>&1
      cat $O/syncab.txt;;
    canary)
      CF="gnr_api.hip,gnr_bwd.hip,gnr_upsample.hip,gnr_wgrad.hip"
      { echo "# tools/session.sh $NAME canary: carve-internal canaries (gazenerf_amd/csrc/gnr_canary.h)"
        GNR_EXTRA_FILES="$CF" GNR_EXTRA_HIPCC_FLAGS="-DGNR_CANARY=1" python -m gazenerf_amd.build --no-torch-ext > $O/canary_build1.log 2>&1; echo "build -DGNR_CANARY=1 rc=$?"
        export GNR_ALLOW_EXPERIMENTAL_LIB=1
        timeout 300 python tests/diagnostics/canary_selftest.py expect-clean 2> $O/canary_clean.err | tail -3
        timeout 1500 python tests/diagnostics/fuzz_hot_path_split.py 40 5 2> $O/canary_fuzz_hot.err | tail -3
        timeout 1500 python tests/diagnostics/fuzz_upsample.py 40 5 2> $O/canary_fuzz_up.err | tail -3
        timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_upsample.py tests/test_network.py -m gpu -q -k "not graph" 2>&1 | tail -6      # (a canary build synchronises inside the entry points: not capturable)
        GNR_EXTRA_FILES="$CF" GNR_EXTRA_HIPCC_FLAGS="-DGNR_CANARY=2" python -m gazenerf_amd.build --no-torch-ext > $O/canary_build2.log 2>&1; echo "build -DGNR_CANARY=2 rc=$?"
        timeout 300 python tests/diagnostics/canary_selftest.py expect-hit 2> $O/canary_hit.err | tail -4
        unset GNR_ALLOW_EXPERIMENTAL_LIB
        python -m gazenerf_amd.build --no-torch-ext > $O/canary_restore.log 2>&1
        python -c "from gazenerf_amd import _lib; print('restored:', _lib.build_info())"; } > $O/canary.txt 2>&1
      cat $O/canary.txt | cut -c1-300;;
    softstart:*)
      { echo "# tools/session.sh $NAME $STEP: -DGNR_SOFTSTART=n (gnr_chain16.h: first-round workgroup i sleeps i n / 64 x 3.4 us); stage_loop alt 8192 rays, then cfg4"
        for N in $(echo ${STEP#softstart:} | tr ',' ' '); do
          if [ "$N" = 0 ]; then python -m gazenerf_amd.build --no-torch-ext > $O/softstart_build_$N.log 2>&1; unset GNR_ALLOW_EXPERIMENTAL_LIB
          else GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_SOFTSTART=$N" python -m gazenerf_amd.build --no-torch-ext > $O/softstart_build_$N.log 2>&1; export GNR_ALLOW_EXPERIMENTAL_LIB=1; fi
          echo "== GNR_SOFTSTART=$N ($(python -c 'from gazenerf_amd import _lib; print(_lib.build_info())' 2>/dev/null))"
          timeout 300 python tools/stage_loop.py alt --rays 8192 --seconds 6 2> $O/softstart_loop_$N.err | tail -2
          timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/softstart_cfg4_$N.json 2> $O/softstart_cfg4_$N.err
          python -c "
import json; d=json.loads(open('$O/softstart_cfg4_$N.json').read().strip().splitlines()[-1])
print('cfg4 ms/step %.2f ' % d['ms_per_step'] + ' '.join('%s %.3f ms %.0f MHz' % (s['stage'], s['avg_ms'], s.get('clock_mhz') or 0) for s in d['stages']))"
        done
        unset GNR_ALLOW_EXPERIMENTAL_LIB
        python -m gazenerf_amd.build --no-torch-ext > $O/softstart_restore.log 2>&1
        python -c "from gazenerf_amd import _lib; print('restored:', _lib.build_info())"; } > $O/softstart.txt 2>&1
      cat $O/softstart.txt | cut -c1-260;;
    hostspin*) A=${STEP#hostspin}; A=${A#:}; bash tools/host_spin_probe.sh $NAME/hostspin${A:+_}${A// /_} $A 2>&1 | tail -40;;
    noise) timeout 2400 python tests/diagnostics/grad_noise_draws.py > $O/grad_noise_draws.txt 2> $O/grad_noise_draws.err; tail -8 $O/grad_noise_draws.txt | cut -c1-250;;
    dropterm)
      GNR_EXTRA_FILES="gnr_bwd3.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_ABLATE=128" python -m gazenerf_amd.build --no-torch-ext > $O/dropterm_build.log 2>&1
      { echo "# tools/session.sh $NAME dropterm: libgnr.so rebuilt with -DGNR_ABLATE=128 on gnr_bwd3.hip (the W_lo x a_hi cross term dropped from ONE layer of"
        echo "# bwd3_chain_kernel, RGB_layer_1^T), then tests/diagnostics/grad_noise_draws.py: the gate of tests/test_parity_gpu.py must FAIL"
        GNR_ALLOW_EXPERIMENTAL_LIB=1 timeout 2400 python tests/diagnostics/grad_noise_draws.py 2> $O/dropterm.err; } > $O/grad_gate_dropped_term.txt
      python -m gazenerf_amd.build > $O/dropterm_rebuild.log 2>&1          # back to the product build
      python -c "from gazenerf_amd import _lib; print('restored:', _lib.build_info())"
      grep -c "FAIL\|  bf16x3" $O/grad_gate_dropped_term.txt; grep "^gate" $O/grad_gate_dropped_term.txt | cut -c1-60;;
    abn1:*)
      T=${STEP#abn1:}; TAG=${T%%:*}; FL=${T#*:}
      GNR_EXTRA_FILES="gnr_conv16.hip,gnr_upsample.hip" GNR_EXTRA_HIPCC_FLAGS="$FL" python -m gazenerf_amd.build --no-torch-ext > $O/ab_$TAG.build.log 2>&1
      GNR_ALLOW_EXPERIMENTAL_LIB=1 bash tools/n1_trace.sh $NAME/ab_$TAG --batch 7 --iters 5 > /dev/null 2>&1
      { echo "# $FL"; grep "N1 B" $O/ab_$TAG/wall.log; grep -v "torch:" $O/ab_$TAG/launches.txt; } > $O/ab_${TAG}_launches.txt; rm -rf $O/ab_$TAG/prof
      python -m gazenerf_amd.build --no-torch-ext > $O/ab_restore.log 2>&1
      grep -E "N1 B|kernel time|blur_lds|unshuffle_kernel|rgb_bwd_blur|946176" $O/ab_${TAG}_launches.txt;;
    n1tile:*)
      TL=${STEP#n1tile:}
      N1_CONV16_TILE=$TL bash tools/n1_trace.sh $NAME/tile_${TL/,/x} --batch 7 --iters 5 > /dev/null 2>&1
      { echo "# N1_CONV16_TILE=$TL (gnr_set_conv16_tile)"; grep "N1 B" $O/tile_${TL/,/x}/wall.log; grep -v "torch:" $O/tile_${TL/,/x}/launches.txt; } > $O/tile_${TL/,/x}_launches.txt; rm -rf $O/tile_${TL/,/x}/prof
      grep -E "N1 B|kernel time|unshuffle" $O/tile_${TL/,/x}_launches.txt;;
    x3outlier) timeout 600 python tests/diagnostics/gpu_x3_outlier.py 13 > $O/x3_outlier_draw13.txt 2> $O/x3_outlier.err; cut -c1-230 $O/x3_outlier_draw13.txt | tail -22;;
    x3energy) timeout 900 python tools/x3_energy.py > $O/x3_energy.txt 2> $O/x3_energy.err; tail -12 $O/x3_energy.txt;;
    x3ab:*)
      T=${STEP#x3ab:}; TAG=${T%%:*}; FL=${T#*:}
      GNR_EXTRA_FILES="gnr_fwd3.hip,gnr_bwd3.hip" GNR_EXTRA_HIPCC_FLAGS="$FL" python -m gazenerf_amd.build --no-torch-ext > $O/x3ab_$TAG.build.log 2>&1
      GNR_ALLOW_EXPERIMENTAL_LIB=1 timeout 900 python tools/x3_energy.py > $O/x3_energy_$TAG.txt 2> $O/x3_energy_$TAG.err; tail -12 $O/x3_energy_$TAG.txt
      python -m gazenerf_amd.build --no-torch-ext > $O/x3ab_restore.log 2>&1;;
    *) echo "unknown step $STEP";;
  esac
done
echo "=== done ($(date +%H:%M:%S))"
