"""Shared helpers for the parity tests."""
import numpy as np

import skeres_amd as sk


def bal_problem_to_sk(prob, x0=None, loss=None):
    """Build a skeres_amd Problem the way EX/SimpleBundleAdjuster.scala:134-145 does,
    through the bulk addResidualBlocks call.  Returns (problem, params DoubleArray, loss)."""
    x0 = prob.parameters if x0 is None else x0
    params = sk.RichDoubleArray.fromArray(x0)
    problem = sk.Problem()
    loss = loss if loss is not None else sk.PredefinedLossFunctions.trivialLoss()
    offs = np.stack([9 * prob.camera_index.astype(np.int64),
                     9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, loss, params, offs)
    return problem, params, loss


def solve_bal_gpu(prob, x0=None, loss=None, **opts):
    problem, params, loss = bal_problem_to_sk(prob, x0, loss)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    for k, v in opts.items():
        getattr(options, k)(*v) if isinstance(v, tuple) else getattr(options, k)(v)
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


def robust_curve_fitting_data():
    """The 67 (x, y) samples of EX/RobustCurveFitting.scala:21-90 (tests/golden/robust_curve_fitting_data.txt)."""
    import os
    return np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robust_curve_fitting_data.txt"))


def sk_loss(spec):
    """oracle-style loss tuple -> skeres_amd LossFunction (None -> None)."""
    L = sk.PredefinedLossFunctions
    if spec is None:
        return None
    kind = spec[0]
    if kind == "composed":
        return L.composedLoss(sk_loss(spec[1]), sk_loss(spec[2]))
    if kind == "scaled":
        return L.scaledLoss(sk_loss(spec[1]), spec[2])
    if kind == "tolerant":
        return L.tolerantLoss(spec[1], spec[2])
    return {"huber": L.huberLoss, "softlone": L.softLOneLoss, "cauchy": L.cauchyLoss, "tukey": L.tukeyLoss}[kind](spec[1])
