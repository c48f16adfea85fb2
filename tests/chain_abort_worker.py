"""Subprocess of tests/test_gpu_parity.py::test_resident_chain_timeout_is_reported_and_refactored.

Fault injection: SK_CHAIN_TEST_WITHHOLD_MARKER=<block column> withholds, once, the SYRK-completion marker that the
column launch of that block column waits for (chol_kernels.hip) — the launch gives up after the chain's time-out.  The
column that times out is the LAST one of a resident run when the marker withheld is the one a run's last column needs,
which is the case the server cannot notice (it has nothing left to wait for): the column launch itself must report
info = 2.  The solver must then (1) say so on stderr, (2) factor the same system again launch by launch in the same
iteration, so that (3) the trajectory is the one of an undisturbed solve.  Runs in a process of its own because the
time-out switches the resident chain off for the rest of the process."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402


def main():
    prob = bal.generate_named("ladybug-1723-156502", seed=1723, perturb=(1e-2, 1e-1, 1e-1))
    problem, params, loss = bal_problem_to_sk(prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    options.setMaxNumIterations(3)
    solver = sk.StepSolver(options, problem)
    resident_before = solver.stat("cholesky_columns_resident")
    t0 = time.time()
    while not solver.step():
        pass
    dt = time.time() - t0
    resident_after = solver.stat("cholesky_columns_resident")
    summary = sk.Solver.Summary()
    solver.finish(summary)
    its = summary.iterations()
    print(json.dumps({"costs": [it["cost"] for it in its], "valid": [int(it["step_is_valid"]) for it in its],
                      "step_norms": [it["step_norm"] for it in its], "resident_before": resident_before, "resident_after": resident_after,
                      "seconds": dt, "x_norm": float(np.linalg.norm(params.toArray(prob.num_parameters)))}))


if __name__ == "__main__":
    main()
