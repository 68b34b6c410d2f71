#!/bin/bash
# Build timing-only variants of the bf16x3 kernel (GNR_ABLATE bits, see gnr_fwd3.hip) into tools/ubench/abl/.
# usage: tools/ablate_fwd3.sh 0 1 2 4 8 ...   then on the GPU box: python tools/gpu_fwd3_ablate.py
set -e
cd "$(dirname "$0")/.."
python -m gazenerf_amd.build >/dev/null
mkdir -p tools/ubench/abl
OBJS=$(ls gazenerf_amd/csrc/build/*.o | grep -v gnr_fwd3.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGNR_EXPERIMENTAL_BUILD=1 -DGNR_ABLATE=$v ${EXTRA_FLAGS} -c gazenerf_amd/csrc/gnr_fwd3.hip -o tools/ubench/abl/fwd3_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/abl/libgnr_abl$v.so $OBJS tools/ubench/abl/fwd3_$v.o ) &
done
wait
ls tools/ubench/abl/*.so
