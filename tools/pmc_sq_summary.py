"""Per-kernel sums of the SQ counters of rocprofv3 --pmc passes:  python tools/pmc_sq_summary.py OUT.csv DIR [DIR ...]

One row per kernel: every counter summed over the kernel's dispatches (and over the XCDs/SEs rocprofv3 reports),
plus the number of launches.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / 1024 divided
by the kernel's duration in cycles = SQ_BUSY_CYCLES (summed over the 32 shader engines) / 32."""
import csv, glob, os, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if not ("emi_f" in k or k.startswith("k_")):
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add((d, row.get("Dispatch_Id", "")))
names = sorted({c for k in acc for c in acc[k]})
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel"] + names + ["launches"])
    for k in sorted(acc):
        nl = max(1, len({x for x in launches[k] if x[0] == dirs[0]}))
        w.writerow([k] + ["%g" % acc[k].get(c, 0.0) for c in names] + [nl])
for k in ("emi_f64::k_leg_inv", "emi_f64::k_leg_dir"):
    if k in acc and acc[k].get("SQ_BUSY_CYCLES"):
        a = acc[k]
        print("%s: MFMA pipe busy %.1f %% of the kernel time; executed MFMA flops %.4g (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512)" % (
            k, 100.0 * (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / (a["SQ_BUSY_CYCLES"] / 32.0), a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512.0))
