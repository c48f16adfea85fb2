"""Mirror of examples/.../CurveFitting.scala:100-133: y = exp(m x + c) through 67 noisy samples."""
import os
import sys

import numpy as np

import skeres_amd as sk

# (x, y) samples: the data table of the reference example (CurveFitting.scala:22-90) shipped with the example (the reference embeds it in the source)
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "curve_fitting_data.txt")


def main(argv=()):
    sk.ceres.initGoogleLogging("CurveFitting")
    data = np.loadtxt(_DATA)
    m = sk.DoubleArray(1)
    m.set(0, 0.0)
    c = sk.DoubleArray(1)
    c.set(0, 0.0)
    loss = sk.PredefinedLossFunctions.trivialLoss()
    problem = sk.Problem()
    for x, y in data:
        problem.addResidualBlock(sk.ExponentialResidual(x, y).toAutoDiffCostFunction(), loss, m, c)
    options = sk.Solver.Options()
    options.setMaxNumIterations(25)
    options.setLinearSolverType(sk.LinearSolverType.DENSE_QR)
    options.setMinimizerProgressToStdout(True)
    print("Initial: 0.0, 0.0")
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    final_x = [float(m.get(0)), float(c.get(0))]
    print(summary.briefReport())
    print("Final: %s" % ", ".join(repr(v) for v in final_x))
    return final_x


if __name__ == "__main__":
    main(sys.argv)
