#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels from the device assembly (hipcc --cuda-device-only -S): per kernel the number of
floating-point vector instructions (v_*_f64 / f32, v_pk_*_f32, MFMA) against every other vector-ALU instruction (integer, address,
compare, select, move, readlane ...), LDS, vector memory and scalar instructions.  VERDICT r4 #2 counts the same on the shipped object.
usage: isa_mix.py file.s [name-filter ...]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        if re.search(r"_f(64|32|16)\b|_f(64|32)_", op) and not op.startswith(("v_cmp", "v_cvt", "v_cndmask")):
            return "valu_fp"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def main():
    path, filt = sys.argv[1], sys.argv[2:]
    cur, stats, ops = None, {}, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            stats[cur], ops[cur] = Counter(), Counter()
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        stats[cur][c] += 1
        if c == "valu_other":
            ops[cur][op] += 1
    import subprocess
    names = list(stats)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    print("%-60s %7s %7s %6s %6s %6s %7s  nonfp/valu" % ("kernel", "fp", "nonfp", "mfma", "lds", "vmem", "scalar"))
    for n, d in zip(names, dem):
        short = re.sub(r"\(.*", "", d).replace("void ", "")
        if filt and not any(f in short for f in filt):
            continue
        s = stats[n]
        tot = s["valu_fp"] + s["valu_other"]
        if tot + s["mfma"] == 0:
            continue
        print("%-60s %7d %7d %6d %6d %6d %7d  %.2f" % (short[:60], s["valu_fp"], s["valu_other"], s["mfma"], s["lds"], s["vmem"], s["scalar"], s["valu_other"] / max(tot, 1)))
        if filt:
            print("      " + ", ".join("%s %d" % kv for kv in ops[n].most_common(14)))


if __name__ == "__main__":
    main()
