"""Mirror of examples/.../PowellAnalytic.scala:6-131: Powell's function with hand-written Jacobians in
SizedCostFunction subclasses (host cost functions: the director path).

Two deliberate differences from the reference text, both in the Jacobian writes: the reference stores
the derivative w.r.t. the SECOND parameter block with `jacobians.set(0, 1, ...)`, i.e. one element past
the end of block 0's 1x1 Jacobian, and never fills block 1's; here each derivative goes to its own block
(`jacobians[1][0, 0]`), which is what the mathematics (and the AutoDiff twin, Powell.scala) needs.
F2a is sqrt(5) (x3 - x4) as PowellAnalytic.scala:34 has it (Powell.scala's F2 is sqrt(5) x3 - x4)."""
import math
import sys

import skeres_amd as sk


class _Pair(sk.SizedCostFunction):
    def __init__(self):
        super().__init__(1, 1, 1)

    def evaluate(self, parameters, residuals, jacobians):
        r, d0, d1 = self.f(parameters[0][0], parameters[1][0])
        residuals[0] = r
        if jacobians is not None:
            if jacobians[0] is not None:
                jacobians[0][0, 0] = d0
            if jacobians[1] is not None:
                jacobians[1][0, 0] = d1
        return True


class F1a(_Pair):  # PowellAnalytic.scala:8-23
    def f(self, x1, x2):
        return x1 + 10 * x2, 1.0, 10.0


class F2a(_Pair):  # :25-42
    def f(self, x3, x4):
        s5 = math.sqrt(5)
        return s5 * (x3 - x4), s5, -s5


class F3a(_Pair):  # :44-60
    def f(self, x2, x3):
        d = x2 - 2.0 * x3
        return d * d, 2.0 * d, -4.0 * d


class F4a(_Pair):  # :62-81
    def f(self, x1, x4):
        s10 = math.sqrt(10.0)
        d = x1 - x4
        return s10 * d * d, s10 * 2 * d, -s10 * 2 * d


def main(argv=()):
    sk.ceres.initGoogleLogging("Powell")
    initial_x = [3.0, -1.0, 0.0, 1.0]
    xs = [sk.DoubleArray(1) for _ in range(4)]
    for a, v in zip(xs, initial_x):
        a.set(0, v)
    x1, x2, x3, x4 = xs
    loss = sk.PredefinedLossFunctions.trivialLoss()
    problem = sk.Problem()
    problem.addResidualBlock(F1a(), loss, x1, x2)
    problem.addResidualBlock(F2a(), loss, x3, x4)
    problem.addResidualBlock(F3a(), loss, x2, x3)
    problem.addResidualBlock(F4a(), loss, x1, x4)
    options = sk.Solver.Options()
    options.setMinimizerType(sk.MinimizerType.TRUST_REGION)
    options.setMinimizerProgressToStdout(True)
    options.setMaxNumIterations(100)
    options.setLinearSolverType(sk.LinearSolverType.DENSE_QR)

    def state_str(v):
        return ", ".join("x%d = %r" % (i + 1, float(xi)) for i, xi in enumerate(v))
    print("Initial: " + state_str(initial_x))
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    xout = [float(a.get(0)) for a in xs]
    print(summary.briefReport())
    print("Final: " + state_str(xout))
    return xout


if __name__ == "__main__":
    main(sys.argv)
