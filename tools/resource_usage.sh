#!/bin/bash
# Register / scratch usage of every kernel in gazenerf_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage), one
# line per kernel: name, VGPRs, AGPRs, scratch bytes per lane, SGPR / VGPR spills, LDS bytes.   tools/resource_usage.sh [file.hip ...]
cd "$(dirname "$0")/../gazenerf_amd/csrc" || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(*.hip)
for f in "${files[@]}"; do
    extra=""; [ "$f" = gnr_wgrad.hip ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
        sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass.*//' |
        awk '/Function Name/ {if (n) print n, v, a, sc, ss, vs, l; n=$3} / VGPRs:/ {v="vgpr " $2} /AGPRs:/ {a="agpr " $2}
             /ScratchSize/ {sc="scratch " $NF} /SGPRs Spill/ {ss="sspill " $3} /VGPRs Spill/ {vs="vspill " $3} /LDS Size/ {l="lds " $NF}
             END {if (n) print n, v, a, sc, ss, vs, l}' | while read -r name rest; do echo "$(echo "$name" | c++filt | cut -c1-70) | $rest"; done
done
# M0 audit for the kernels whose inline asm overwrites M0 without saving it (gnr_chain3.h, ring_issue_piece): every line
# mentioning m0 must be one of those writes.
for f in gnr_fwd3.hip gnr_bwd3.hip; do
    n=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -o - "$f" 2>/dev/null | grep -w "m0" | grep -vcE "^\s*s_mov_b32 m0, (s[0-9]+|vcc_lo|vcc_hi|ttmp[0-9]+)\s*$")
    echo "$f: foreign M0 uses: $n"
done
