#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r4d > gpurun_out/r4d_collect.log 2>&1
bash tools/pmc_fft.sh r4d >> gpurun_out/r4d_collect.log 2>&1
python3 bench.py --precision 4 --no-cpu-baseline --no-api-level > gpurun_out/r4d_bench_tco1279_fp32.json 2>/dev/null
python3 bench.py --nsmax 399 --nfld 4 --no-cpu-baseline --no-api-level > gpurun_out/r4d_bench_tco399.json 2>/dev/null
tail -c 1500 gpurun_out/r4d_bench.json
