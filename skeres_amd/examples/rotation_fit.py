"""No counterpart among the reference's examples: a use of what its Problem inherits from ceres::Problem but none of its
programs exercises — a local parameterization (PredefinedLocalParameterizations.quaternion(), ceres.i:203-205).

Fit a rotation to noisy point pairs.  The unknown is a unit quaternion (w, x, y, z); with the parameterization the
minimiser works in the 3-dimensional tangent space and every step is applied through Plus, so the iterate stays on
the unit sphere; without it the fourth direction (the norm, which the residuals do not see) is a free parameter."""
import sys

import numpy as np

import skeres_amd as sk


def main(argv=(), seed=4, n=12, noise=0.02):
    rng = np.random.default_rng(seed)
    q_true = np.array([0.7, -0.4, 0.5, 0.3])
    q_true /= np.linalg.norm(q_true)
    points = rng.normal(size=(n, 3))
    R = sk.Rotation.quaternionToRotation(sk.Quaternion(*q_true))  # the device code device functors call (csrc/rotation.hpp)
    Rm = np.array([[float(R(i, j)) for j in range(3)] for i in range(3)])
    targets = points @ Rm.T + noise * rng.normal(size=(n, 3))

    q = sk.DoubleArray(4)
    for i, v in enumerate([1.0, 0.0, 0.0, 0.0]):
        q.set(i, v)
    problem = sk.Problem()
    problem.addParameterBlock(q, 4, sk.PredefinedLocalParameterizations.quaternion())
    costs = [sk.QuaternionRotationError(p, t).toAutoDiffCostFunction() for p, t in zip(points, targets)]
    for cost in costs:
        problem.addResidualBlock(cost, None, q)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_QR)
    options.setMinimizerProgressToStdout(bool(argv))
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    got = np.array([q.get(i) for i in range(4)])
    if got[0] * q_true[0] < 0:
        got = -got  # q and -q are the same rotation
    print(summary.briefReport())
    print("estimate  %s  (norm %.15f)" % (np.array2string(got, precision=6), np.linalg.norm(got)))
    print("truth     %s" % np.array2string(q_true, precision=6))
    return got, q_true


if __name__ == "__main__":
    main(sys.argv[1:])
