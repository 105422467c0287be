#!/bin/bash
# round-4 experiment 1: (a) four-waves-per-SIMD Legendre loop in the probe, (b) tile order of the Legendre launches vs fetched bytes
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
echo "== probe"; PROBE_ONLY_W4=1 timeout 300 ./tools/leg_loop_probe 2>&1 | tee $O/r4_probe_w4.txt
for cfg in "base" "EMI_LEG_DIR_ORDER=1" "EMI_LEG_INV_ORDER=1"; do
  echo "== $cfg"
  if [ "$cfg" = base ]; then bash tools/leg_traffic.sh r4_order_base; else env $cfg bash tools/leg_traffic.sh r4_order_${cfg//=/_}; fi
done 2>&1 | tee $O/r4_exp1.txt
