"""Mirror of examples/.../Powell.scala:55-91 (Powell's singular function, four scalar blocks)."""
import sys

import skeres_amd as sk


def main(argv=()):
    sk.ceres.initGoogleLogging("Powell")
    initial_x = [3.0, -1.0, 0.0, 1.0]
    xs = [sk.DoubleArray(1) for _ in range(4)]
    for a, v in zip(xs, initial_x):
        a.set(0, v)
    x1, x2, x3, x4 = xs
    loss = sk.PredefinedLossFunctions.trivialLoss()
    problem = sk.Problem()
    problem.addResidualBlock(sk.PowellF1().toAutoDiffCostFunction(), loss, x1, x2)
    problem.addResidualBlock(sk.PowellF2().toAutoDiffCostFunction(), loss, x3, x4)
    problem.addResidualBlock(sk.PowellF3().toAutoDiffCostFunction(), loss, x2, x3)
    problem.addResidualBlock(sk.PowellF4().toAutoDiffCostFunction(), loss, x1, x4)
    options = sk.Solver.Options()
    options.setMinimizerType(sk.MinimizerType.TRUST_REGION)
    options.setMinimizerProgressToStdout(True)
    options.setMaxNumIterations(100)
    options.setLinearSolverType(sk.LinearSolverType.DENSE_QR)
    print("Initial: %s" % ", ".join(str(v) for v in initial_x))
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    xout = [float(a.get(0)) for a in xs]
    print(summary.briefReport())
    print("Final: %s" % ", ".join("%.3g" % v for v in xout))
    return xout


if __name__ == "__main__":
    main(sys.argv)
