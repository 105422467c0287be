#!/bin/bash
# Legendre launch time and fetched bytes for one tile-map configuration (environment as given):
#   EMI_LEG_DIR_MGROUPS=2 bash tools/leg_traffic.sh TAG        (through gpurun; writes gpurun_out/TAG_*)
TAG=${1:-legmap}
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python3 tools/gpu_perf.py 1279 137 10 3 2>&1 | tail -2 | head -1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 tools/gpu_perf.py 1279 137 10 1 > $O/${TAG}_fetch.log 2>&1
python3 - "$O/${TAG}_fetch" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(float)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_leg_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"]) * 1024.0 * 2.0
for k in sorted(acc):
    print("   %-24s read %.1f GB per launch (2 x FETCH_SIZE)" % (k, acc[k] / 1e9))
PY
