#!/usr/bin/env python3
"""Aggregate the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic of each kernel.

  # counter collection serialises kernels: the passes run with --no-resident-kernels (sk_options_set_resident_kernels(o, 0): the
  # same factorisation plan of the UNDISSECTED system, one launch per step, nothing that waits for another kernel)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone --no-resident-kernels > bench_under_pmc.json
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-alone --no-resident-kernels
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r04_pmc_traffic.json bench_under_pmc.json

With the bench line of one of the passes as fourth argument the summary is LIKE FOR LIKE (round-3 verdict, item 5): the SYRK's
counter bytes per launch beside the algorithmic C-tile bytes per launch OF THAT PLAN, and per-iteration totals of the SYRK,
the Schur assembly, the Jacobian phase and the back-substitution beside SURVEY.md section 8(d)'s algorithmic bytes.

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
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)", "kernels": res}
    # per-iteration totals: an iteration = one linear solve (bal_point_block_kernel runs once per solve)
    iters = res.get("sk::bal_point_block_kernel", {}).get("launches", 0)
    groups = {
        "syrk": ["sk::syrk_trailing_f64_kernel", "sk::syrk_trailing_thin_f64_kernel"],
        "cholesky_other": ["sk::potrf128_kernel", "sk::trsm_gemm_f64_kernel", "sk::trsm_gemm_thin_f64_kernel", "sk::gemm_update_f64_kernel", "sk::gemm_update_thin_f64_kernel",
                           "sk::gemm_diag_f64_kernel", "sk::bs_step_kernel", "sk::bs_resident_kernel", "sk::copy_row_kernel"],
        "schur_assembly": ["sk::bal_point_block_kernel", "sk::bal_obs_precompute_kernel", "sk::bal_cam_diag_kernel", "sk::bal_pair_kernel", "sk::bal_pair_long_kernel",
                           "sk::zero_envelope_kernel", "sk::bal_finish_all_kernel"],
        "jacobian_phase": ["sk::bal_eval_jac_kernel", "sk::bal_cam_records_kernel", "sk::bal_cam_reduce_kernel", "sk::bal_pt_reduce_kernel", "sk::grad_max_xnorm_kernel"],
        "back_substitution_and_cost": ["sk::bal_cam_step_kernel", "sk::bal_obs_backsub_kernel", "sk::bal_point_backsub_kernel", "sk::bal_eval_cost_kernel"],
    }
    if iters > 0:
        totals = {}
        for g, names in groups.items():
            b = 0.0
            for n, r in res.items():
                if any(n == m or n.startswith(m + "<") for m in names):
                    b += r["hbm_bytes_per_launch_corrected"] * r["launches"]
            totals[g] = b / iters
        out["linear_solves_in_the_pass"] = iters
        out["hbm_bytes_per_iteration"] = totals
    if len(sys.argv) > 4:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import bench_record
        line = bench_record.load(sys.argv[4])  # (the run's detailed record: bench.py's side file, SK_BENCH_DETAILS)
        roof = line["roofline"]
        n_obs = line["config"].get("observations")
        alg = roof.get("algorithmic_c_tile_bytes_per_launch")
        sy = res.get("syrk_trailing (both tilings)")
        if alg and sy:
            out["syrk_like_for_like"] = {
                "plan": roof.get("plan"), "launches_per_iteration": roof.get("launches_per_iteration"),
                "counter_bytes_per_launch": sy["hbm_bytes_per_launch_corrected"], "algorithmic_c_tile_bytes_per_launch": alg,
                "ratio": sy["hbm_bytes_per_launch_corrected"] / alg,
                "note": "same plan on both sides: the pass's own bench line (sk_options_set_resident_kernels(o, 0): undissected, launch by launch)"}
        if n_obs and iters > 0:
            env = line["config"].get("envelope_bytes", 0.0)
            out["algorithmic_bytes_per_iteration"] = {"jacobian_phase": 328.0 * n_obs, "schur_assembly": 208.0 * n_obs + env, "back_substitution_and_cost": 136.0 * n_obs,
                                                      "note": "SURVEY.md section 8(d): 328 / 208 (+ the envelope written once) / 136 B per observation"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for n in ("syrk_trailing (both tilings)", "sk::syrk_trailing_f64_kernel", "sk::syrk_trailing_thin_f64_kernel"):
        if n in res:
            print(n, json.dumps(res[n]))


if __name__ == "__main__":
    main()
