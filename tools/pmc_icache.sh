# instruction-cache counters of the FFT kernels:  bash tools/pmc_icache.sh TAG   (through gpurun; kernel trace + one counter pass)
TAG=${1:-ic}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${TAG}_ic -- python3 tools/gpu_perf.py 1279 137 10 1 8 > gpurun_out/${TAG}_ic.log 2>&1
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/${TAG}_ic/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_fft_\w+(<\d+>)?|k_leg_\w+)',r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
print("kernel icache_req hit_rate misses/wave ifetch/wave")
for k,a in sorted(acc.items(), key=lambda kv:-kv[1]["SQC_ICACHE_REQ"])[:16]:
    w=max(a["SQ_WAVES"],1)
    print("%-20s %.3g %.3f %.1f %.1f"%(k,a["SQC_ICACHE_REQ"],a["SQC_ICACHE_HITS"]/max(a["SQC_ICACHE_REQ"],1),(a["SQC_ICACHE_MISSES"]+a["SQC_ICACHE_MISSES_DUPLICATE"])/w,a["SQ_IFETCH"]/w))
PY
