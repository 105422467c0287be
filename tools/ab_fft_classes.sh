#!/bin/bash
# per-kernel FFT times of two library builds on ONE box:  bash tools/ab_fft_classes.sh SUFFIX   (ectrans_amd/libectrans_mi.so.SUFFIX against the product build)
cd ${GRAFT_REPO_ROOT:-$PWD}
for sfx in "$1" ""; do
  tag=ab_${sfx:-product}
  EMI_LIB=$PWD/ectrans_amd/libectrans_mi.so${sfx:+.$sfx} bash tools/fft_class_times.sh $tag > gpurun_out/$tag.txt 2>&1
  echo "== ${sfx:-product}"; grep "ms per" gpurun_out/$tag.txt | sort -k2 -n -r | head -${2:-10}; grep total gpurun_out/$tag.txt
done
