#!/usr/bin/env python3
"""Counter-derived fp64-MFMA utilisation per kernel from one rocprofv3 pass (SURVEY.md sections 5 / 8(d)):

  export SK_CHOL_CHAIN_SERVER=0   # counter collection serialises kernels: the same launches, none of them resident
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- \
      python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone
  python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma profiles/r02_pmc_mfma.json

SQ_INSTS_VALU_MFMA_MOPS_F64 counts fp64 MFMA operations in units of 512 flops (MI355X_MICROARCH.md / rocprofv3 counter
description: "MOPS" = 512-flop units); flops = 512 * MOPS.  Kernel durations come from the kernel trace of the same pass
(End - Start), so utilisation = flops / duration / 78.6 TFLOP/s is what the counters saw under collection (kernels run one
at a time there).  The tool also prints flops-per-launch against the algorithmic count where it knows it."""
import collections
import csv
import glob
import json
import sys

PEAK = 78.6e12


def main():
    d, out = sys.argv[1], sys.argv[2]
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[n]:
            seen[n].add(r["Dispatch_Id"])
            acc[n]["_seconds"] += dur.get(r["Dispatch_Id"], 0.0)
            acc[n]["_launches"] += 1
    res = {}
    for n, a in sorted(acc.items()):
        mops = a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        if mops <= 0:
            continue
        flops = 512.0 * mops
        res[n] = {"launches": int(a["_launches"]), "mfma_f64_flops_per_launch": flops / a["_launches"], "seconds_per_launch": a["_seconds"] / a["_launches"],
                  "tflops": flops / a["_seconds"] * 1e-12 if a["_seconds"] > 0 else None,
                  "mfma_util_of_78.6": flops / a["_seconds"] / PEAK if a["_seconds"] > 0 else None,
                  "sq_busy_cycles_per_launch": a.get("SQ_BUSY_CYCLES", 0.0) / a["_launches"]}
    json.dump({"method": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES (one pass), 512 flops per MOPS unit; durations from the kernel "
                         "trace of the same pass; SK_CHOL_CHAIN_SERVER=0", "kernels": res}, open(out, "w"), indent=1)
    for n, v in res.items():
        print(n, json.dumps(v))


if __name__ == "__main__":
    main()
