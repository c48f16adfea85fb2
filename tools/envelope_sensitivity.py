#!/usr/bin/env python3
"""How much of the headline rests on the banded structure of the synthetic Ladybug problem: LM iterations/s and
the block-envelope fill of the reduced camera system against the fraction of tracks that are seen from two distant
windows of the trajectory (loop closures; skeres_amd.bal.generate(long_range_fraction=...)).

  python tools/envelope_sensitivity.py > profiles/r02_envelope_sensitivity.txt

Every row is a fresh problem of the exact Ladybug-1723 shape (C=1723, P=156502, N=678718, seed 1723) solved with the
default options (automatic plan); `full` is the same problem with the envelope off (every 128-block factored)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402


def run(workload, frac, envelope, steps, warmup):
    prob = bal.generate_named(workload, seed=1723, perturb=(1e-2, 1e-1, 1e-1), long_range_fraction=frac)
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(1, prob.observations, None, params, offs)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    o.setCholeskyEnvelope(envelope)
    s = sk.StepSolver(o, problem)
    for _ in range(warmup):
        s.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    dt = (time.perf_counter() - t0) / steps
    st = {k: s.stat(k) for k in ("envelope_fill", "camera_order", "cholesky_flops_plan", "cholesky_flops_full", "cholesky_columns_resident")}
    summ = sk.Solver.Summary()
    s.finish(summ)
    its = summ.iterations()
    n_ok = sum(it["step_is_successful"] for it in its[1:])
    return dt, st, n_ok, len(its) - 1, summ.phaseSeconds(2) / max(1, len(its) - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ladybug-1723-156502")
    ap.add_argument("--fractions", type=float, nargs="+", default=[0.0, 0.005, 0.02, 0.10])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    print("# %s, seed 1723, default options; %d timed LM iterations after %d warm-up" % (args.workload, args.steps, args.warmup))
    print("# long-range tracks | envelope fill | camera order | resident columns | plan TFlop | ms/iteration | it/s | Cholesky ms | successful steps")
    for f in args.fractions:
        dt, st, n_ok, n_it, chol = run(args.workload, f, True, args.steps, args.warmup)
        print("%6.1f %%   fill %.3f   order %d   resident %3d   %.3f TFlop   %7.2f ms   %6.1f it/s   chol %6.2f ms   %d/%d" % (
            100 * f, st["envelope_fill"], int(st["camera_order"]), int(st["cholesky_columns_resident"]), 1e-12 * st["cholesky_flops_plan"],
            1e3 * dt, 1.0 / dt, 1e3 * chol, n_ok, n_it), flush=True)
    dt, st, n_ok, n_it, chol = run(args.workload, 0.0, False, max(3, args.steps // 3), 1)
    print("  full     fill %.3f   order %d   resident %3d   %.3f TFlop   %7.2f ms   %6.1f it/s   chol %6.2f ms   %d/%d" % (
        st["envelope_fill"], int(st["camera_order"]), int(st["cholesky_columns_resident"]), 1e-12 * st["cholesky_flops_plan"], 1e3 * dt, 1.0 / dt,
        1e3 * chol, n_ok, n_it), flush=True)


if __name__ == "__main__":
    main()
