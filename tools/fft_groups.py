"""Per-launch-group FFT kernel durations from a rocprofv3 kernel trace:
    python tools/fft_groups.py TRACE_DIR      (after rocprofv3 --kernel-trace --output-format csv -d TRACE_DIR -- python3 tools/gpu_perf.py ...)"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "k_fft" in n:
        wg = int(r["Workgroup_Size_X"])
        acc[(n, wg, int(r["Grid_Size_X"]) // wg, r["LDS_Block_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(acc):
    v = acc[k]
    print("%-20s T=%4d blocks=%8d lds=%6s  n=%d  min %.2f ms" % (k[0], k[1], k[2], k[3], len(v), min(v)))
