#!/bin/bash
# Round 4, GPU session 25: is it the dump stores' power that costs the training forward its clock after a backward?  tools/stage_loop.py
# alt (forward, backward, forward, ... as a training loop issues them) at cfg4's launch size, shipped build vs experimental builds of
# gnr_fwd16.hip without the per-layer dumps (-DGNR_ABL16=2: the backward then reads stale buffers -- timing only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s25
mkdir -p $O
export TMPDIR=/tmp GNR_ALLOW_EXPERIMENTAL_LIB=1
cd $R
run() { # tag flags
  if [ -n "$2" ]; then GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="$2" python -m gazenerf_amd.build --no-torch-ext > $O/$1.build.log 2>&1
  else python -m gazenerf_amd.build --no-torch-ext > $O/$1.build.log 2>&1; fi
  for rays in 8192 32768; do
    echo "== $1 alt rays=$rays"; timeout 300 python tools/stage_loop.py alt --rays $rays --seconds 6 2>&1 | grep -v amdgpu | tail -4
  done
  echo "== $1 fwd alone rays=8192"; timeout 300 python tools/stage_loop.py fwd --rays 8192 --seconds 5 2>&1 | grep -v amdgpu | tail -2
}
run shipped "" 2>&1 | tee $O/shipped.txt
run nodump "-DGNR_ABL16=2" 2>&1 | tee $O/nodump.txt
run burst1 "-DGNR_DUMP_BURST=1" 2>&1 | tee $O/burst1.txt
python -m gazenerf_amd.build --no-torch-ext > $O/restore.build.log 2>&1
echo done
