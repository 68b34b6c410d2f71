#!/bin/bash
# Round 4, GPU session 22: cfg4 with the reference trainer's default loss switches (--ref-loss: L1 + VGG-perceptual) and with the
# PatchGAN terms on top; kernel stats of the --ref-loss step (what the loss costs beside the hot path).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s22
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --ref-loss > $O/bench_cfg4_refloss.json 2> $O/bench_cfg4_refloss.err
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --ref-loss --gan-loss > $O/bench_cfg4_refloss_gan.json 2> $O/bench_cfg4_refloss_gan.err
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --ref-loss --precision bf16x3 > $O/bench_cfg4_refloss_bf16x3.json 2> $O/bench_cfg4_refloss_bf16x3.err
bash tools/prof_stats.sh r4s22/stats --config cfg4 --steps 5 --warmup 2 --ref-loss > /dev/null 2>&1
rm -rf $O/stats/prof
python - <<'P'
import json
for f in ("bench_cfg4","bench_cfg4_refloss","bench_cfg4_refloss_gan","bench_cfg4_refloss_bf16x3"):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r4s22/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), round(d['ms_per_step'],3), round(d['outside_hot_path_ms'],3), d['config']['loss'][:60])
    except Exception as e: print(f, 'ERR', e)
P
head -25 $O/stats/kernel_stats.csv | cut -c1-160
for f in $O/*.err; do tail -n 2 $f; done
echo done
