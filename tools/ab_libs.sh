#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== base"; EMI_LIB=$PWD/ectrans_amd/libectrans_mi.so.base python tools/gpu_perf.py 1279 137 10 4 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== new"; python tools/gpu_perf.py 1279 137 10 4 2>&1 | grep -v amdgpu.ids | tail -3
done
