#!/bin/bash
# round-end collection on the GPU box (through gpurun): tests, profiles of the bench command, counters of BOTH libraries, the other single-GPU configurations
TAG=${1:-r6z}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$EMI_FINAL_SKIP_TESTS" ]; then
python3 -m pytest tests -x -q -m gpu > $O/${TAG}_gputests.log 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/${TAG}_gputests.log | tail -3
fi
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.txt 2>&1
bash tools/pmc_fft.sh $TAG > $O/${TAG}_pmc_fft_stdout.txt 2>&1
bash tools/pmc_ea.sh $TAG > $O/${TAG}_pmc_ea_stdout.txt 2>&1
# fp32 library: HBM traffic of the Legendre launches (two passes) and the FFT counter table
B32="python3 bench.py --precision 4 --steps 1 --warmup 0 --no-cpu-baseline --no-api-level --no-dense-timing --no-fortran"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fp32_pmc_fetch -- $B32 > $O/${TAG}_fp32_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_fp32_pmc_write -- $B32 > $O/${TAG}_fp32_pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/${TAG}_fp32_pmc_fetch $O/${TAG}_fp32_pmc_write $O/${TAG}_pmc_traffic_fp32.json --label "on bench.py --precision 4 --steps 1 --warmup 0, TCo1279 137Lx10 fp32, 1x MI355X (k_leg_dir loads 8 B per lane: its 2 x FETCH_SIZE is an upper bound)" > $O/${TAG}_pmc_traffic_fp32.txt 2>&1
bash tools/pmc_fft.sh $TAG 4 > $O/${TAG}_pmc_fft_fp32_stdout.txt 2>&1
# the bench line quotes the counter files of ITS build (source hash): put the ones just collected where bench.py looks (the box's copy of the repo)
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_fft.json $O/${TAG}_pmc_traffic_fp32.json $O/${TAG}_pmc_fft_fp32.json profiles/ 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_tco1279_driver_cmd.json 2> $O/${TAG}_bench_driver_err.log
python3 bench.py --nsmax 399 --nfld 4 --steps 20 --warmup 5 --no-fortran > $O/${TAG}_bench_tco399.json 2>> $O/${TAG}_bench_driver_err.log
python3 bench.py --precision 4 --no-cpu-baseline --no-api-level > $O/${TAG}_bench_tco1279_fp32.json 2>> $O/${TAG}_bench_driver_err.log
python3 bench.py --nsmax 2559 --precision 4 --steps 3 --warmup 1 --no-cpu-baseline --no-api-level > $O/${TAG}_bench_tco2559_fp32.json 2>> $O/${TAG}_bench_driver_err.log
for f in tco1279_driver_cmd tco399 tco1279_fp32 tco2559_fp32; do python3 - <<PY
import json
try:
    j=json.loads(open("$O/${TAG}_bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(j["value"],3), "pairs/s", round(j["ms_per_step"],2), "ms", j["phase_ms_per_step"], "frac", round(j["roofline"]["frac"],3), "traffic", j["roofline"]["traffic"], "dense", j.get("dense_timing",{}).get("ms_per_step"), "fortran", (j.get("fortran_device_resident") or {}).get("ms_per_pair"), "fft_launches", j.get("fft_launches_per_direction"))
except Exception as e: print("$f failed", e)
PY
done
