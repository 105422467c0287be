# SQ counters of one mode of tools/fft_r16_probe:  bash tools/pmc_probe.sh MODE  -> gpurun_out/pmc_probe_MODE/
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
M=$1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_probe_$M/a -- tools/fft_r16_probe 32768 3 $M > gpurun_out/pmc_probe_$M.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_probe_$M/b -- tools/fft_r16_probe 32768 3 $M >> gpurun_out/pmc_probe_$M.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/pmc_probe_$M/c -- tools/fft_r16_probe 32768 3 $M >> gpurun_out/pmc_probe_$M.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_probe_$M/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
n=len([1 for f in glob.glob("gpurun_out/pmc_probe_$M/a/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "k_conv" in r["Kernel_Name"] and r["Counter_Name"]=="SQ_WAVES"])
with open("gpurun_out/pmc_probe_$M.txt","w") as fh:
    for k in sorted(acc): fh.write("%s %g\n"%(k,acc[k]))
    fh.write("dispatch_rows %d\n"%n)
print(open("gpurun_out/pmc_probe_$M.txt").read())
PY
