"""Shared helpers for the parity tests."""
import numpy as np

import skeres_amd as sk


def bal_problem_to_sk(prob, x0=None):
    """Build a skeres_amd Problem the way EX/SimpleBundleAdjuster.scala:134-145 does,
    through the bulk addResidualBlocks call.  Returns (problem, params DoubleArray, loss)."""
    x0 = prob.parameters if x0 is None else x0
    params = sk.RichDoubleArray.fromArray(x0)
    problem = sk.Problem()
    loss = sk.PredefinedLossFunctions.trivialLoss()
    offs = np.stack([9 * prob.camera_index.astype(np.int64),
                     9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, loss, params, offs)
    return problem, params, loss


def solve_bal_gpu(prob, x0=None, **opts):
    problem, params, loss = bal_problem_to_sk(prob, x0)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    for k, v in opts.items():
        getattr(options, k)(v)
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    return params.toArray(prob.num_parameters), summary


CURVE_DATA = None


def curve_fitting_data():
    """The 67 (x, y) samples of EX/CurveFitting.scala:22-90 (tests/golden/curve_fitting_data.txt)."""
    global CURVE_DATA
    if CURVE_DATA is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curve_fitting_data.txt")
        CURVE_DATA = np.loadtxt(path)
    return CURVE_DATA
