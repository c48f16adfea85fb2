#!/usr/bin/env python3
"""Developer tool: wall time of sk_solver_create (upload, camera order, envelope, retained points, dissection, pair lists, queue trial,
iteration 0) for a named workload, twice in one process (the second without the device's one-time queue trial)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402

NAMES = {"ladybug": ("ladybug-1723-156502", 1723), "venice": ("venice-1778-993923", 1778), "bal49": ("problem-49-7776", 49)}
name, seed = NAMES[sys.argv[1] if len(sys.argv) > 1 else "ladybug"]
prob = bal.generate_named(name, seed=seed, perturb=(1e-2, 1e-1, 1e-1))
for k in range(2):
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    t0 = time.time()
    problem.addResidualBlocks(1, prob.observations, None, params, offs)
    t1 = time.time()
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    s = sk.StepSolver(o, problem)
    t2 = time.time()
    for _ in range(3):
        s.step()
    t3 = time.time()
    print("%s run %d: addResidualBlocks %.3f s, solver create (set-up + iteration 0) %.3f s, three iterations %.4f s" % (name, k, t1 - t0, t2 - t1, t3 - t2))
    del s
