#!/bin/bash
# A/B of library builds on ONE box: bash tools/ab_libs.sh [suffix ...]   (ectrans_amd/libectrans_mi.so.<suffix>; "" = the product build)
cd ${GRAFT_REPO_ROOT:-$PWD}
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(base "")
for rep in 1 2; do
  for sfx in "${libs[@]}"; do
    lib=$PWD/ectrans_amd/libectrans_mi.so${sfx:+.$sfx}
    echo "== ${sfx:-product}"; EMI_LIB=$lib python tools/gpu_perf.py 1279 137 10 4 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
  done
done
