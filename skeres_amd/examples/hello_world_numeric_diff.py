"""Mirror of examples/.../HelloWorldNumericDiff.scala:6-35: the HelloWorld problem with a numerically
differentiated host functor (central differences) — the director path end to end."""
import sys

import skeres_amd as sk


class HelloCostFunctor(sk.NumericDiffCostFunctor):  # HelloWorldNumericDiff.scala:7-9
    def __init__(self):
        super().__init__(1, 1)

    def apply(self, x):
        return [10.0 - x[0]]


def main(argv=()):
    sk.ceres.initGoogleLogging("HelloWorld")
    initial_x = 0.5
    x = sk.RichDoubleArray.ofSize(1)
    x.set(0, initial_x)
    problem = sk.Problem()
    cost = HelloCostFunctor().toNumericDiffCostFunction(sk.NumericDiffMethodType.CENTRAL)
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
