"""CORE/CostFunctionToFunctor.scala and the host side of CORE/AutodiffCostFunction.scala (functors whose generic body
is host code).  The reference has no spec for CostFunctionToFunctor; the cases below pin its documented behaviour
(:66-122: residuals over doubles; residuals + chain rule over Jets) on the reference's own known-answer functors
(TEST/AutodiffCostFuntionSpec.scala:14-138) and on analytic derivatives.  Host logic only."""
import math

import numpy as np
import pytest

import skeres_amd as sk
from skeres_amd import rotation as R
from skeres_amd.rotation import Jet


class BinaryScalarHost(sk.HostAutoDiffCostFunctor):  # AutodiffCostFuntionSpec.scala:14-26, body on the host
    def __init__(self, a):
        super().__init__(1, 2, 2)
        self.a = a

    def apply(self, x, y):
        return [x[0] * y[0] + x[1] * y[1] - self.a]


class BinaryVector3Host(sk.HostAutoDiffCostFunctor):  # :55-69
    def __init__(self, a):
        super().__init__(3, 2, 2)
        self.a = a

    def apply(self, x, y):
        return [x[0] * y[0] + x[1] * y[1] - self.a, x[0] * y[0] - x[1] * y[1] - self.a, x[0] * y[1] + x[1] * y[0] + self.a * self.a * 13]


class TenParameterHost(sk.HostAutoDiffCostFunctor):  # :111-119
    def __init__(self):
        super().__init__(1, *([1] * 10))

    def apply(self, *x):
        s = x[0][0]
        for xi in x[1:]:
            s = s + xi[0]
        return [s]


def evaluate(cost, blocks, want_jacobians=True):
    p = [np.array(b, dtype=np.float64) for b in blocks]
    r = np.zeros(cost.numResiduals())
    J = [np.zeros((cost.numResiduals(), len(b))) for b in blocks] if want_jacobians else None
    assert cost.evaluate(p, r, J)
    return r, J


def test_host_autodiff_reproduces_the_reference_known_answers():
    r, J = evaluate(BinaryScalarHost(1.0).toAutoDiffCostFunction(), [[1.0, 2.0], [3.0, 4.0]])
    assert r.tolist() == [10.0] and J[0].tolist() == [[3.0, 4.0]] and J[1].tolist() == [[1.0, 2.0]]  # :28-51
    r0, _ = evaluate(BinaryScalarHost(1.0).toAutoDiffCostFunction(), [[1.0, 2.0], [3.0, 4.0]], want_jacobians=False)
    assert r0.tolist() == r.tolist()  # :39-40: the null-Jacobian call returns the same residuals
    r, J = evaluate(BinaryVector3Host(1.0).toAutoDiffCostFunction(), [[1.0, 2.0], [3.0, 4.0]])
    assert r.tolist() == [10.0, -6.0, 23.0]
    assert J[0].tolist() == [[3.0, 4.0], [3.0, -4.0], [4.0, 3.0]] and J[1].tolist() == [[1.0, 2.0], [1.0, -2.0], [2.0, 1.0]]
    r, J = evaluate(TenParameterHost().toAutoDiffCostFunction(), [[float(i)] for i in range(10)])
    assert r.tolist() == [45.0] and all(j.tolist() == [[1.0]] for j in J)  # :121-138


def test_jet_functions_against_analytic_derivatives():
    x = Jet(0.7, 0, 2)
    y = Jet(-1.3, 1, 2)
    cases = [
        (R.sqrt(x * x + y * y), math.hypot(0.7, 1.3), [0.7 / math.hypot(0.7, 1.3), -1.3 / math.hypot(0.7, 1.3)]),
        (R.exp(x) * R.sin(y), math.exp(0.7) * math.sin(-1.3), [math.exp(0.7) * math.sin(-1.3), math.exp(0.7) * math.cos(-1.3)]),
        (R.atan2(y, x), math.atan2(-1.3, 0.7), [1.3 / (0.49 + 1.69), 0.7 / (0.49 + 1.69)]),
        (x / y, 0.7 / -1.3, [1 / -1.3, -0.7 / 1.69]),
        (2.0 / x - y ** 3, 2 / 0.7 + 1.3 ** 3, [-2 / 0.49, -3 * 1.69]),
        (R.log(x) + R.cos(y) - R.acos(x) + R.asin(x) * R.atan(y),
         math.log(0.7) + math.cos(-1.3) - math.acos(0.7) + math.asin(0.7) * math.atan(-1.3),
         [1 / 0.7 + 1 / math.sqrt(1 - 0.49) + math.atan(-1.3) / math.sqrt(1 - 0.49), -math.sin(-1.3) + math.asin(0.7) / (1 + 1.69)]),
    ]
    for got, real, grad in cases:
        assert got.real == pytest.approx(real, rel=1e-15, abs=1e-15)
        assert got.infinitesimal.tolist() == pytest.approx(grad, rel=1e-14, abs=1e-15)
    assert (x < y) is False and (abs(y)).real == 1.3 and (abs(y)).infinitesimal.tolist() == [0.0, -1.0]


class AnalyticF4(sk.SizedCostFunction):  # PowellAnalytic.scala:62-81, derivatives in their own blocks
    def __init__(self):
        super().__init__(1, 1, 1)

    def evaluate(self, parameters, residuals, jacobians):
        d = parameters[0][0] - parameters[1][0]
        residuals[0] = math.sqrt(10.0) * d * d
        if jacobians is not None:
            if jacobians[0] is not None:
                jacobians[0][0, 0] = 2 * math.sqrt(10.0) * d
            if jacobians[1] is not None:
                jacobians[1][0, 0] = -2 * math.sqrt(10.0) * d
        return True


class Failing(sk.SizedCostFunction):
    def __init__(self):
        super().__init__(1, 1)

    def evaluate(self, parameters, residuals, jacobians):
        return False


def test_cost_function_to_functor_over_doubles_and_jets():
    functor = sk.CostFunctionToFunctor(AnalyticF4())
    assert isinstance(functor, sk.HostAutoDiffCostFunctor) and functor.kNumResiduals == 1 and functor.N == (1, 1)
    assert sk.DynamicCostFunctionToFunctor is sk.CostFunctionToFunctor
    # doubles: the cost function's residuals (CostFunctionToFunctor.scala:66-77)
    assert functor([3.0], [1.0]) == [pytest.approx(math.sqrt(10.0) * 4.0)]
    # Jets carrying derivatives w.r.t. three underlying parameters: chain rule (:79-122)
    a, b = Jet(3.0, [1.0, 2.0, 0.0]), Jet(1.0, [0.0, -1.0, 5.0])
    out = functor([a], [b])[0]
    g = 2 * math.sqrt(10.0) * 2.0
    assert out.real == pytest.approx(math.sqrt(10.0) * 4.0)
    assert out.infinitesimal.tolist() == pytest.approx([g * 1.0, g * 2.0 - g * -1.0, -g * 5.0])
    # back through the host autodiff: the wrapped cost function's own Jacobians
    r, J = evaluate(functor.toAutoDiffCostFunction(), [[3.0], [1.0]])
    assert r.tolist() == [pytest.approx(math.sqrt(10.0) * 4.0)] and J[0][0, 0] == pytest.approx(g) and J[1][0, 0] == pytest.approx(-g)
    # a failing evaluation gives an empty result (:73-74, :87-88), which the cost function reports as failure
    failing = sk.CostFunctionToFunctor(Failing())
    assert failing([1.0]) == [] and failing([Jet(1.0, 0, 1)]) == []
    assert not failing.toAutoDiffCostFunction().evaluate([np.array([1.0])], np.zeros(1), [np.zeros((1, 1))])
    with pytest.raises(ValueError):
        functor([1.0, 2.0], [1.0])


class Composite(sk.HostAutoDiffCostFunctor):
    """A functor that calls a wrapped cost function from its generic body — what CostFunctionToFunctor is for
    (ceres' cost_function_to_functor.h use case): residual = exp(0.1 f4(x1, x4)) + x1 x4."""

    def __init__(self):
        super().__init__(1, 1, 1)
        self.f4 = sk.CostFunctionToFunctor(AnalyticF4())

    def apply(self, x1, x4):
        return [R.exp(0.1 * self.f4(x1, x4)[0]) + x1[0] * x4[0]]


def test_wrapped_cost_function_inside_a_generic_functor():
    r, J = evaluate(Composite().toAutoDiffCostFunction(), [[0.4], [-0.2]])
    d, s10 = 0.6, math.sqrt(10.0)
    e = math.exp(0.1 * s10 * d * d)
    assert r[0] == pytest.approx(e - 0.08, rel=1e-15)
    assert J[0][0, 0] == pytest.approx(e * 0.1 * 2 * s10 * d - 0.2, rel=1e-14)
    assert J[1][0, 0] == pytest.approx(-e * 0.1 * 2 * s10 * d + 0.4, rel=1e-14)
