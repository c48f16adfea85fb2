#!/usr/bin/env python3
"""BASELINE.json config 5 on one MI355X: synthetic dense NLLS, m residuals x n parameters,
DENSE_NORMAL_CHOLESKY.  Prints iterations, per-phase times and the achieved fp64-MFMA rate of
the J^T J SYRK (HIP events around its launches)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1000000)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--seed", type=int, default=5)
    args = ap.parse_args()
    m, n = args.m, args.n
    rng = np.random.default_rng(args.seed)
    x_star = rng.normal(size=n)
    t0 = time.time()
    y = sk.api.synth_dense_targets(args.seed, m, n, x_star) + rng.normal(0, 1e-3, m)
    consts = np.stack([np.full(m, float(args.seed)), np.arange(m, dtype=np.float64), y], axis=1)
    print("targets on the GPU: %.1f s" % (time.time() - t0), flush=True)
    x = sk.DoubleArray(n)
    problem = sk.Problem()
    problem.addDenseRows(10, consts, None, x, n)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_NORMAL_CHOLESKY)
    o.setMaxNumIterations(1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    t0 = time.time()
    s = sk.StepSolver(o, problem)
    print("setup + iteration 0 (Jacobian %.1f GB): %.2f s" % (8e-9 * m * n, time.time() - t0), flush=True)
    s.setKernelTiming(1)
    t0 = time.time()
    for _ in range(args.iters):
        s.step()
    dt = (time.time() - t0) / args.iters
    sec, launches = s.kernelSeconds("syrk_gram")
    flops = s.stat("jtj_flops_algorithmic")  # m n (n + 1), SURVEY.md section 8(d) — not the padded tiles (syrkFlopsPerSolve)
    summ = sk.Solver.Summary()
    s.finish(summ)
    out = {"config": "synthetic dense NLLS m=%d n=%d DENSE_NORMAL_CHOLESKY" % (m, n), "seconds_per_iteration": dt,
           "iterations_per_second": 1.0 / dt, "syrk_gram_tflops": flops * launches / sec * 1e-12 if sec > 0 else None,
           "syrk_gram_frac_of_78.6": flops * launches / sec * 1e-12 / 78.6 if sec > 0 else None,
           "syrk_gram_seconds_per_launch": sec / max(1, launches),
           "costs": [it["cost"] for it in summ.iterations()], "max_abs_error_vs_planted": float(np.abs(x.toArray(n) - x_star).max()),
           "phases_s_per_iteration": {k: summ.phaseSeconds(i) / max(1, len(summ.iterations()) - 1) for i, k in enumerate(
               ["jacobian_eval", "normal_equations", "cholesky", "step", "cost_eval"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
