"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [--label TEXT]

Corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies
the 128-byte requests of wide (16 B/lane) coalesced reads at 64 bytes, so read bytes = 2 x FETCH_SIZE for
kernels whose global loads are all that wide (every kernel of this library loads real2/double2 pairs);
WRITE_SIZE is exact for 16 B/lane stores.  Infinity-Cache hits are counted as traffic, so the figures
are an upper bound of the HBM bytes."""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            name = name.replace("void ", "").strip()
            acc[name] += float(row["Counter_Value"]) * 1024.0
            cnt[name] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main():
    fd, wd, out = sys.argv[1:4]
    label = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--label" else ""
    rd, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ectrans_amd
    res = {"source_hash": ectrans_amd.source_hash(),  # bench.py quotes this file only for the build it was taken on
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) " + label,
           "corrections": "KiB -> bytes; read bytes = 2 x FETCH_SIZE (gfx950, 16 B/lane loads); WRITE_SIZE exact; "
                          "Infinity-Cache hits included (upper bound of HBM bytes)",
           "kernels": {}}
    for k in sorted(set(rd) | set(wr)):
        if not ("emi_f" in k or k.startswith("k_")):
            continue
        r, n = rd.get(k, (0.0, 0))
        w, _ = wr.get(k, (0.0, 0))
        res["kernels"][k] = {"launches_profiled": n, "fetch_size_bytes_raw_per_launch": r,
                             "read_bytes_per_launch": 2 * r, "write_bytes_per_launch": w,
                             "hbm_bytes_per_launch": 2 * r + w}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-28s n=%3d read %8.2f GB write %8.2f GB" % (k, v["launches_profiled"], v["read_bytes_per_launch"] / 1e9,
                                                            v["write_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
