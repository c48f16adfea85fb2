"""Subprocess of tests/test_gpu_parity.py::test_event_forms_and_event_free_forms_are_the_same_arithmetic: one solve of a 900-camera
sequence under the lock-step dissection with retained points; prints the iteration costs and a checksum of the parameters as hex
floats.  The developer variables that select the form of the chain's starts / joins / back-substitutions (SK_CHAIN_EARLY_SERVER,
SK_BS_PAIR, SK_BS_SPREAD: read once per process) come from the environment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402


def main():
    prob = bal.generate(900, 9000, 40000, seed=9)
    problem, params, loss = bal_problem_to_sk(prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    options.setRetainedPoints("on", 12)
    options.setCholeskyDissection("auto")
    options.setMaxNumIterations(6)
    solver = sk.StepSolver(options, problem)
    dissected, resident = solver.stat("dissected"), solver.stat("cholesky_columns_resident")
    while not solver.step():
        pass
    summary = sk.Solver.Summary()
    solver.finish(summary)
    x = params.toArray(prob.num_parameters)
    print("FORMS dissected=%d resident=%d costs=%s x=%s" % (dissected, resident, ",".join(float(it["cost"]).hex() for it in summary.iterations()),
                                                            float(np.sum(x * np.arange(1, x.size + 1) % 7)).hex()))


if __name__ == "__main__":
    main()
