#!/usr/bin/env python3
"""Developer tool: the Ladybug-1723-shaped solve with the Snavely body RECORDED (skeres_amd.TracedCostFunctor) against
the registered device functor: costs per iteration and the time of the evaluation kernels."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from skeres_amd.examples.traced_functors import TracedSnavelyReprojectionError  # noqa: E402

prob = bal.generate_named("ladybug-1723-156502", seed=1723, perturb=(5e-2, 5e-1, 5e-1))
offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
f = TracedSnavelyReprojectionError(0.0, 0.0)
ins, consts, nregs, outs = f.tape()
print("tape: %d instructions, %d registers, %d literals" % (ins.shape[0], nregs, consts.size))
for traced in (False, True):
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    if traced:
        problem.addResidualBlocksTraced(f, prob.observations, None, params, offs)
    else:
        problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, None, params, offs)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(1000)
    o.setFunctionTolerance(0.0); o.setGradientTolerance(0.0); o.setParameterTolerance(0.0)
    s = sk.StepSolver(o, problem)
    s.step()
    s.setKernelTiming(1)
    t0 = time.time()
    for _ in range(8):
        s.step()
    dt = (time.time() - t0) / 8
    summ = sk.Solver.Summary()
    j, jn = s.kernelSeconds("bal_eval_jac")
    c, cn = s.kernelSeconds("bal_eval_cost")
    s.finish(summ)
    print("%-10s %.2f ms per iteration (all kernels timed); eval_jac %.1f us x %d, eval_cost %.1f us x %d; costs %s" % (
        "recorded" if traced else "registered", 1e3 * dt, 1e6 * j / max(1, jn), jn, 1e6 * c / max(1, cn), cn,
        " ".join("%.9e" % it["cost"] for it in summ.iterations()[:4])))
