TAG=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o ${TAG} -- python3 tools/gpu_perf.py 2559 137 10 1 4 > gpurun_out/${TAG}_stats.log 2>&1
f=$(find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$f")))
tot=0
for r in rows:
    n=r["Name"]
    if "k_fft" in n:
        m=re.search(r'(k_fft_\w+(<\d+>)?)',n)
        ms=float(r["TotalDurationNs"])/1e6
        tot+=ms
        if ms>5: print("%-22s %7.2f ms (calls %s)"%(m.group(1),ms,r["Calls"]))
print("FFT total per pair %.1f ms"%tot)
PY
tail -2 gpurun_out/${TAG}_stats.log
