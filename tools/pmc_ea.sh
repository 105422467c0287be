# L2 memory-side (EA) read requests of the Legendre launches, split by destination: bash tools/pmc_ea.sh TAG   (through gpurun)
# is the traffic of k_leg_dir / k_leg_inv served by HBM or by the Infinity Cache?  TCC_EA0_RDREQ = all read requests the L2s send
# to the fabric, _32B / _64B / _128B their sizes, _DRAM those routed to memory (HBM behind the Infinity Cache), with the stall counters
# of the memory credit path.  One PMC pass per group of four TCC counters, kernel trace only.
TAG=${1:-ea}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${TAG}_$PASS -- python3 tools/gpu_perf.py 1279 137 10 1 8 > gpurun_out/${TAG}_$PASS.log 2>&1; }
PASS=ea1; run TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum
PASS=ea2; run TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
PASS=ea3; run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
PASS=ea4; run TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); n=collections.defaultdict(int)
for p in ("ea1","ea2","ea3","ea4"):
    for f in glob.glob("gpurun_out/${TAG}_%s/**/*counter_collection.csv"%p,recursive=True):
        for r in csv.DictReader(open(f)):
            m=re.search(r'(k_leg_\w+|k_fft_\w+(<\d+>)?|k_prepack_inv|k_postpack_dir)',r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
for f in glob.glob("gpurun_out/${TAG}_ea1/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_leg_\w+|k_fft_\w+(<\d+>)?|k_prepack_inv|k_postpack_dir)',r["Kernel_Name"])
        if m: dur[m.group(1)]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; n[m.group(1)]+=1
with open("gpurun_out/${TAG}_pmc_ea.txt","w") as fh:
    names=sorted({c for a in acc.values() for c in a})
    fh.write("counters: "+" ".join(names)+"\n")
    for k in sorted(acc,key=lambda k:-dur[k])[:12]:
        a=acc[k]
        fh.write("%-20s %7.2f ms (%d launches)  "%(k,dur[k],n[k])+"  ".join("%s=%.4g"%(c.replace("TCC_EA0_","").replace("_sum",""),a[c]) for c in names)+"\n")
        rd=a.get("TCC_EA0_RDREQ_sum",0.0)
        if rd:
            b32,b64,b128=a.get("TCC_EA0_RDREQ_32B_sum",0.0),a.get("TCC_EA0_RDREQ_64B_sum",0.0),a.get("TCC_EA0_RDREQ_128B_sum",0.0)
            other=rd-b32-b64-b128
            fh.write("    read requests %.4g: 32B %.3g 64B %.3g 128B %.3g other %.3g; bytes if other = 64 B: %.1f GB, if other = 128 B: %.1f GB; to DRAM %.1f %% of the requests (32B ones: %.3g); L2 hit rate %.1f %%\n"%(
                rd,b32,b64,b128,other,(b32*32+b64*64+b128*128+other*64)/1e9,(b32*32+b64*64+b128*128+other*128)/1e9,100*a.get("TCC_EA0_RDREQ_DRAM_sum",0)/rd,a.get("TCC_EA0_RDREQ_DRAM_32B_sum",0),
                100*a.get("TCC_HIT_sum",0)/max(1.0,a.get("TCC_HIT_sum",0)+a.get("TCC_MISS_sum",0))))
print(open("gpurun_out/${TAG}_pmc_ea.txt").read())
PY
tail -2 gpurun_out/${TAG}_ea1.log
