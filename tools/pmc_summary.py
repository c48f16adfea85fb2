#!/usr/bin/env python3
"""Aggregate the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic of each kernel.

  export SK_CHOL_CHAIN_SERVER=0   # counter collection serialises kernels: same launches, none of them resident
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

gfx950 corrections applied: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so
it is doubled (upper bound: the 8-B-per-lane C-tile loads are uncalibrated); WRITE_SIZE is exact.
Both counters are in KiB.  Infinity-Cache hits are included in FETCH_SIZE."""
import collections
import csv
import glob
import json
import sys


def agg(d, counter):
    path = glob.glob(d + "/*/*counter_collection.csv")[0]
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[n][0] += float(r["Counter_Value"])
            out[n][1] += 1
    return out


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    res = {}
    for n in sorted(f):
        fetch = f[n][0] / f[n][1] * 1024.0
        write = w[n][0] / max(1, w[n][1]) * 1024.0
        res[n] = {"launches": f[n][1], "fetch_size_bytes_per_launch_raw": fetch, "write_size_bytes_per_launch": write,
                  "hbm_bytes_per_launch_corrected": 2.0 * fetch + write}
    # the trailing update is launched as either of two tilings: one combined per-launch figure
    names = [n for n in res if n.startswith("sk::syrk_trailing_")]
    if names:
        launches = sum(f[n][1] for n in names)
        fetch = sum(f[n][0] for n in names) / launches * 1024.0
        write = sum(w[n][0] for n in names) / max(1, sum(w[n][1] for n in names)) * 1024.0
        res["syrk_trailing (both tilings)"] = {"launches": launches, "fetch_size_bytes_per_launch_raw": fetch, "write_size_bytes_per_launch": write,
                                               "hbm_bytes_per_launch_corrected": 2.0 * fetch + write}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)", "kernels": res},
              open(sys.argv[3], "w"), indent=1)
    for n in ("syrk_trailing (both tilings)", "sk::syrk_trailing_f64_kernel", "sk::syrk_trailing_thin_f64_kernel"):
        if n in res:
            print(n, json.dumps(res[n]))


if __name__ == "__main__":
    main()
