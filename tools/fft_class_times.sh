# per-kernel FFT times of one TCo1279 pair:  bash tools/fft_class_times.sh TAG [env assignments...]   (through gpurun)
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o ${TAG} -- python3 tools/gpu_perf.py ${EMI_PERF_N:-1279} ${EMI_PERF_NLEV:-137} ${EMI_PERF_NFLD:-10} 2 > gpurun_out/${TAG}_stats.log 2>&1
f=$(find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${TAG}_kernel_stats.csv
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "k_fft" in n:
        m=re.search(r'(k_fft_\w+(<\d+>)?)',n)
        ms=float(r["TotalDurationNs"])/1e6/2
        tot+=ms
        print("%-22s %7.2f ms per pair-half (calls %s)"%(m.group(1),ms,r["Calls"]))
print("FFT total per pair %.1f ms"%tot)
PY
tail -3 gpurun_out/${TAG}_stats.log
