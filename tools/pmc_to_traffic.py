#!/usr/bin/env python3
"""tools/pmc_to_traffic.py FETCH_DIR WRITE_DIR KERNEL_SUBSTRING OUT.json [n] [config] [extra-kernel ...]
With PMC_KERNEL_STATS=<kernel_stats.csv of a rocprofv3 --kernel-trace --stats run of the same program on the same box> every record
also carries avg_launch_us: bytes and duration of ONE box, which is the only pairing bench.py quotes a PMC fraction for.
Turns two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other: they do not fit one pass on
gfx950) into the per-launch HBM byte count bench.py reads: FETCH_SIZE x2 (gfx950 counts 128-B read requests as
64 B; MI355X_MICROARCH.md, re-checked by tools/pmc_calib.hip), WRITE_SIZE as read, both in KiB."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


def main():
    fd, wd, key, out = sys.argv[1:5]
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 149
    config = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    extra = sys.argv[7:]
    F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    dur = {}
    if os.environ.get("PMC_KERNEL_STATS") and os.path.exists(os.environ["PMC_KERNEL_STATS"]):
        for r in csv.DictReader(open(os.environ["PMC_KERNEL_STATS"])):
            dur[r["Name"]] = float(r["AverageNs"]) / 1e3

    def one(sub):
        fk = [k for k in F if sub in k]
        wk = [k for k in W if sub in k]
        if not fk or not wk:
            raise SystemExit(f"kernel {sub!r} not found in the passes")
        (nf, f), (nw, w) = F[fk[0]], W[wk[0]]
        rec = {"kernel": fk[0][:120], "launches": [nf, nw], "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
               "fetch_correction": 2.0, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
        dk = [k for k in dur if sub in k]
        if dk:
            rec["avg_launch_us"] = dur[dk[0]]
        return rec

    j = {"n": n, "config": config}
    j.update(one(key))
    j["calibration"] = ("tools/pmc_calib.hip: FETCH_SIZE reads exactly 1/2 of a streamed byte count on gfx950, "
                        "WRITE_SIZE reads stores and atomics exactly")
    j["source"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, averaged over the launches"
    for e in extra:
        j.setdefault("other_kernels", {})[e] = one(e)
    with open(out, "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
