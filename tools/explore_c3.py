#!/usr/bin/env python3
"""Developer tool: run the Ladybug-1723-shaped solve on the GPU with a given
start perturbation, print the iteration log and the per-kernel HIP-event times."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402

KERNELS = ["bal_eval_jac", "bal_cam_records", "bal_cam_reduce", "memset_S", "bal_cam_diag", "bal_pair", "gemm_diag_update", "gemm_panel_update", "potrf128", "gemm_trsm", "gemm_syrk_next",
           "gemm_syrk", "backsolve", "bal_eval_cost"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ladybug-1723-156502")
    ap.add_argument("--perturb", type=float, nargs=3, default=[5e-2, 5e-1, 5e-1])
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--group", type=int, default=0)
    ap.add_argument("--no-lookahead", action="store_true")
    ap.add_argument("--dense-factor", action="store_true", help="factor every block (no envelope)")
    ap.add_argument("--timing", type=int, default=1, help="0 none, 1 all named kernels (slows the chain), 2 SYRK only")
    args = ap.parse_args()
    prob = bal.generate_named(args.workload, seed=1723, perturb=tuple(args.perturb))
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
    o.setCholeskyTuning(args.group, not args.no_lookahead)
    o.setCholeskyEnvelope(not args.dense_factor)
    t0 = time.time()
    s = sk.StepSolver(o, problem)
    print("setup + iteration 0: %.2f s" % (time.time() - t0))
    print("plan: " + ", ".join("%s %g" % (k, s.stat(k)) for k in ("dissected", "dissection_head_cameras", "dissection_separator_cameras", "dissection_tail_cameras",
                                                                 "dissection_model_us_plain", "dissection_model_us", "cholesky_columns_resident", "envelope_fill")))
    s.step()
    s.setKernelTiming(args.timing)
    t0 = time.time()
    for _ in range(args.iters):
        s.step()
    dt = time.time() - t0
    print("%d iterations: %.3f ms each" % (args.iters, 1e3 * dt / args.iters))
    for k in KERNELS:
        sec, n = s.kernelSeconds(k)
        print("  %-18s %6d launches  %9.3f ms total  %8.3f ms/iter  %8.1f us/launch" % (k, n, 1e3 * sec, 1e3 * sec / args.iters,
                                                                                       1e6 * sec / max(1, n)))
    summ = sk.Solver.Summary()
    s.finish(summ)
    n_it = max(1, len(summ.iterations()) - 1)
    print("device ms per iteration: " + ", ".join("%s %.2f" % (n, 1e3 * summ.phaseSeconds(k) / n_it) for k, n in enumerate(
        ["jacobian", "assembly (incl. jacobian)", "factor + solve", "back-substitution", "cost", "all-reduce"])))
    for i, it in enumerate(summ.iterations()):
        print("%3d cost %.6e  ok %d  rho %.3f  radius %.2e  |g| %.2e" % (i, it["cost"], it["step_is_successful"], it["relative_decrease"],
                                                                        it["trust_region_radius"], it["gradient_max_norm"]))


if __name__ == "__main__":
    main()
