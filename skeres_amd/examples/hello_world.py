"""Mirror of examples/.../HelloWorld.scala:10-41: minimise 1/2 (10 - x)^2 from x = 0.5 with an autodiff functor."""
import sys

import skeres_amd as sk


def main(argv=()):
    sk.ceres.initGoogleLogging("HelloWorld")
    initial_x = 0.5
    x = sk.DoubleArray(1)
    x.set(0, initial_x)
    problem = sk.Problem()
    cost = sk.HelloCostFunctor().toAutoDiffCostFunction()
    loss = sk.PredefinedLossFunctions.trivialLoss()
    problem.addResidualBlock(cost, loss, x)
    options = sk.Solver.Options()
    options.setMinimizerProgressToStdout(True)
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    print(summary.briefReport())
    return float(x.get(0))


if __name__ == "__main__":
    main(sys.argv)
