#!/usr/bin/env python3
"""Developer tool: stability soak on one GPU — many LM iterations of the Ladybug-1723-shaped solve in one solver, then
solvers created and destroyed in a loop (device memory must come back), then the chain's give-up counter: how often a
factorisation was lost to a time-out of the resident chain (should be 0 on an unshared device)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402


def make(prob):
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(1, prob.observations, None, params, offs)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(100000)
    o.setFunctionTolerance(0.0); o.setGradientTolerance(0.0); o.setParameterTolerance(0.0)
    return sk.StepSolver(o, problem), problem, params


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    revisits = [(200, 900, 40, 150), (450, 1300, 40, 150), (700, 1600, 40, 150)] if (len(sys.argv) > 2 and sys.argv[2] == "revisits") else ()
    prob = bal.generate_named("ladybug-1723-156502", seed=1723, perturb=(5e-2, 5e-1, 5e-1), revisits=revisits)
    s, problem, params = make(prob)
    print("plan: retained points %d, border cameras %d, dissected %d, block columns under the resident chain %d" % (
        s.stat("retained_points"), s.stat("border_cameras"), s.stat("dissected"), s.stat("cholesky_columns_resident")), flush=True)
    t0 = time.time()
    worst, last = 0.0, time.time()
    for i in range(iters):
        s.step()
        now = time.time()
        worst, last = max(worst, now - last), now
        if i % 250 == 249:
            print("%5d iterations, %.2f ms each so far, slowest single step %.1f ms" % (i + 1, 1e3 * (now - t0) / (i + 1), 1e3 * worst), flush=True)
    summ = sk.Solver.Summary()
    s.finish(summ)
    its = summ.iterations()
    print("costs: first %.6e last %.6e; invalid steps %d of %d" % (its[0]["cost"], its[-1]["cost"], sum(1 for it in its if not it.get("step_is_valid", 1)), len(its)))
    del s, problem, params
    free0 = torch.cuda.mem_get_info()[0]
    small = bal.generate(64, 4000, 20000, seed=5)
    for k in range(40):
        s, problem, params = make(small)
        for _ in range(3):
            s.step()
        s.finish(sk.Solver.Summary())
        del s, problem, params
    free1 = torch.cuda.mem_get_info()[0]
    print("free device memory before / after 40 solver lifetimes: %.1f / %.1f MB" % (free0 / 1e6, free1 / 1e6))


if __name__ == "__main__":
    main()
