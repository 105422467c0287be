#!/bin/bash
# Round-end evidence on the GPU box:  bash tools/collect_profiles.sh TAG
# (run through gpurun; writes gpurun_out/TAG_*; copy what should be judged into profiles/)
TAG=${1:-r2d}
O=gpurun_out
# EMI_COLLECT_PRECISION=4: the same collection for the fp32 library (files TAG_fp32_*, traffic file TAG_pmc_traffic_fp32.json)
PREC=${EMI_COLLECT_PRECISION:-8}
PSFX=""; PLBL="fp64"
if [ "$PREC" = 4 ]; then PSFX=_fp32; PLBL="fp32"; TAGP=${TAG}; TAG=${TAG}_fp32; fi
mkdir -p $O
export TMPDIR=/tmp
BENCH="python3 bench.py --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran"
# 1. kernel trace + stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o ${TAG} -- $BENCH > $O/${TAG}_stats.log 2>&1
# 2. HBM traffic: two PMC passes, kernel-trace only (1 step: the counters are per launch)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAGP:-$TAG}_pmc_traffic${PSFX}.json --label "on bench.py --precision $PREC --steps 1 --warmup 0, TCo1279 137Lx10 $PLBL, 1x MI355X" > $O/${TAG}_pmc_traffic.txt 2>&1
# 3. SQ counters (MFMA busy, LDS conflicts, occupancy) in their own pass
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/${TAG}_pmc_sq1 -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/${TAG}_pmc_sq2 -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_sq2.log 2>&1
# 3b. L2 hit rate per kernel: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_pmc_l2 -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_l2.log 2>&1
# 4. the bench line itself, with the CPU baseline, outside any profiler
python3 bench.py --precision $PREC $( [ "$PREC" = 4 ] && echo --no-cpu-baseline --no-api-level ) > $O/${TAG}_bench.json 2> $O/${TAG}_bench_err.log
tail -c 3000 $O/${TAG}_bench.json
cat $O/${TAG}_pmc_traffic.txt
find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs head -12
# 5. SQ summary per kernel (MFMA busy, LDS conflicts, VALU / wait split of the FFT kernels)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_sq3 -- python3 bench.py --precision $PREC --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran > $O/${TAG}_pmc_sq3.log 2>&1
python3 tools/pmc_sq_summary.py $O/${TAG}_pmc_sq_summary.csv $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 $O/${TAG}_pmc_sq3 $O/${TAG}_pmc_l2 > $O/${TAG}_pmc_sq_summary.txt 2>&1
cat $O/${TAG}_pmc_sq_summary.txt
find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_tco1279_kernel_stats.csv
