# SQ counters of the FFT kernels of one TCo1279 pair:  bash tools/pmc_fft.sh TAG [PRECISION = 8 | 4]   (through gpurun; three counter passes, kernel trace only)
TAG=${1:-hot}
PREC=${2:-8}
FP=F64; SFX=""; [ "$PREC" = 4 ] && FP=F32 && SFX=_fp32
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${TAG}_pmc1 -- python3 tools/gpu_perf.py 1279 137 10 1 $PREC > gpurun_out/${TAG}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/${TAG}_pmc2 -- python3 tools/gpu_perf.py 1279 137 10 1 $PREC > gpurun_out/${TAG}_pmc2.log 2>&1
# pass 3: what the vector ALU issues -- double-precision arithmetic against everything else (integer, address, compare / select, moves)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_$FP SQ_INSTS_VALU_ADD_$FP SQ_INSTS_VALU_MUL_$FP SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d gpurun_out/${TAG}_pmc3 -- python3 tools/gpu_perf.py 1279 137 10 1 $PREC > gpurun_out/${TAG}_pmc3.log 2>&1
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float))
dur=collections.defaultdict(float)
acc3=collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("gpurun_out/${TAG}_pmc1","gpurun_out/${TAG}_pmc2","gpurun_out/${TAG}_pmc3"):
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            m=re.search(r'(k_fft_\w+(<\d+>)?)',r["Kernel_Name"])
            if m: (acc3 if d.endswith("pmc3") else acc)[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
def nonfp(k):
    a=acc3[k]
    if not a.get("SQ_INSTS_VALU"): return float("nan")
    return 1.0-(a["SQ_INSTS_VALU_FMA_${FP}"]+a["SQ_INSTS_VALU_ADD_${FP}"]+a["SQ_INSTS_VALU_MUL_${FP}"])/a["SQ_INSTS_VALU"]
for f in glob.glob("gpurun_out/${TAG}_pmc1/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_fft_\w+(<\d+>)?)',r["Kernel_Name"])
        if m: dur[m.group(1)]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
with open("gpurun_out/${TAG}_pmc_fft${SFX}.txt","w") as fh:
    fh.write("kernel ms clkGHz valu/wave lds/wave vmem/wave | share of SIMD time: valu lds any | wave life: wait_any wait_inst | lds conflict | VALU: fma add mul (${FP}; a packed fp32 instruction counts once) int32 int64 cvt per wave, non-fp share\n")
    for k in sorted(acc, key=lambda k:-dur[k]):
        a=acc[k]
        if not a.get("SQ_WAVES"): continue
        w=a["SQ_WAVES"]; cyc=a["SQ_BUSY_CYCLES"]/32.0
        simd=cyc*1024.0/4.0   # quad-cycles of all SIMDs
        b=acc3[k]; w3=max(b.get("SQ_INSTS_VALU",0.0)/max(a["SQ_INSTS_VALU"]/w,1.0),1.0)  # waves of pass 3 (same launches)
        fh.write("%-20s %6.2f %5.2f %6.0f %5.0f %5.0f | %4.2f %4.2f %4.2f | %4.2f %4.2f | %5.3f | %5.0f %5.0f %5.0f %5.0f %5.0f %4.0f  %4.2f\n"%(k,dur[k],cyc/dur[k]/1e6 if dur[k] else 0,
            a["SQ_INSTS_VALU"]/w,a["SQ_INSTS_LDS"]/w,(a["SQ_INSTS_VMEM_RD"]+a["SQ_INSTS_VMEM_WR"])/w,
            a["SQ_ACTIVE_INST_VALU"]/simd,a["SQ_ACTIVE_INST_LDS"]/simd,a["SQ_ACTIVE_INST_ANY"]/simd,
            a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"],a["SQ_WAIT_INST_ANY"]/a["SQ_WAVE_CYCLES"],
            a["SQ_LDS_BANK_CONFLICT"]/max(1.0,a["SQ_LDS_IDX_ACTIVE"]),
            b["SQ_INSTS_VALU_FMA_${FP}"]/w3,b["SQ_INSTS_VALU_ADD_${FP}"]/w3,b["SQ_INSTS_VALU_MUL_${FP}"]/w3,b["SQ_INSTS_VALU_INT32"]/w3,b["SQ_INSTS_VALU_INT64"]/w3,b["SQ_INSTS_VALU_CVT"]/w3,nonfp(k)))
print(open("gpurun_out/${TAG}_pmc_fft${SFX}.txt").read())
# machine-readable summary for bench.py's `fft_bound` block: duration-weighted issue shares of all FFT launches of one pair,
# stamped with the hash of the library sources (bench.py quotes it only for the build it was taken on)
import json, sys
sys.path.insert(0, ".")
import ectrans_amd
tot = sum(dur.values())
w = lambda key, den: sum(dur[k] * acc[k][key] / (acc[k]["SQ_BUSY_CYCLES"] / 32.0 * 1024.0 / 4.0 if den == "simd" else acc[k]["SQ_WAVE_CYCLES"]) for k in dur if acc[k].get("SQ_WAVES")) / tot
js = {"source_hash": ectrans_amd.source_hash(), "workload": "tools/gpu_perf.py 1279 137 10 (TCo1279, KF = 1645, precision ${PREC}), one pair, all k_fft_* launches", "precision": ${PREC},
      "fft_ms_per_pair": tot, "simd_issue_share": {"valu": w("SQ_ACTIVE_INST_VALU", "simd"), "lds": w("SQ_ACTIVE_INST_LDS", "simd"), "any": w("SQ_ACTIVE_INST_ANY", "simd")},
      "wave_life_share": {"wait_any": w("SQ_WAIT_ANY", "wave"), "wait_inst_any": w("SQ_WAIT_INST_ANY", "wave")},
      "nonfp_valu_share": sum(dur[k] * nonfp(k) for k in dur if acc3[k].get("SQ_INSTS_VALU")) / max(sum(dur[k] for k in dur if acc3[k].get("SQ_INSTS_VALU")), 1e-9),
      "nonfp_valu_share_six_heaviest": {k: nonfp(k) for k in sorted(dur, key=lambda k: -dur[k])[:6]},
      "kernels": {k: {"ms": dur[k], "valu_per_wave": acc[k]["SQ_INSTS_VALU"] / acc[k]["SQ_WAVES"], "lds_per_wave": acc[k]["SQ_INSTS_LDS"] / acc[k]["SQ_WAVES"],
                      "clock_GHz": acc[k]["SQ_BUSY_CYCLES"] / 32.0 / dur[k] / 1e6,
                      "valu_share": acc[k]["SQ_ACTIVE_INST_VALU"] / (acc[k]["SQ_BUSY_CYCLES"] / 32.0 * 256.0), "any_share": acc[k]["SQ_ACTIVE_INST_ANY"] / (acc[k]["SQ_BUSY_CYCLES"] / 32.0 * 256.0),
                      "nonfp_valu_share": nonfp(k),
                      "lds_bank_conflict_share": acc[k]["SQ_LDS_BANK_CONFLICT"] / max(1.0, acc[k]["SQ_LDS_IDX_ACTIVE"])} for k in dur if acc[k].get("SQ_WAVES")},
      "method": "rocprofv3 --pmc, two passes (SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_BUSY_CYCLES; SQ_INSTS_* / SQ_LDS_*), kernel trace only; shares = counter (quad-cycles, summed over SIMDs) / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs / 4), weighted by launch duration"}
json.dump(js, open("gpurun_out/${TAG}_pmc_fft${SFX}.json", "w"), indent=1)
print(json.dumps({k: js[k] for k in ("fft_ms_per_pair", "simd_issue_share", "wave_life_share", "nonfp_valu_share", "nonfp_valu_share_six_heaviest", "source_hash")}))
PY
