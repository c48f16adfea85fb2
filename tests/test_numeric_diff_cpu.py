"""The reference's NumericDiffCostFunctionSpec (core/src/test/scala/.../NumericDiffCostFunctionSpec.scala),
restated against the Python host mirror of CORE/NumericDiffCostFunction.scala.  Host logic only: a
numerically differentiated cost function evaluates caller code on the host in the reference too."""
import numpy as np
import pytest

import skeres_amd as sk


def expect_close(x, y, tol):  # TestUtil.scala:6-22
    d = abs(x - y)
    rel = d if (x == 0.0 or y == 0.0) else d / max(abs(x), abs(y))
    return rel <= tol


class EasyFunctor(sk.NumericDiffCostFunctor):  # NumericDiffCostFunctionSpec.scala:81-97
    def __init__(self):
        super().__init__(3, 5, 5)

    def apply(self, x1, x2):
        y0 = float(np.dot(x1, x2))
        return [y0, y0 * y0, float(np.dot(x2, x2))]


class TranscendentalFunctor(sk.NumericDiffCostFunctor):  # :158-175
    def __init__(self):
        super().__init__(2, 5, 5)

    def apply(self, x1, x2):
        d = float(np.dot(x1, x2))
        return [np.sin(d), np.exp(-d / 10.0)]


@pytest.mark.parametrize("method,tol", [(sk.NumericDiffMethodType.FORWARD, 2e-5), (sk.NumericDiffMethodType.CENTRAL, 3e-9)])
def test_easy_case(method, tol):  # :26-78
    cost = EasyFunctor().toNumericDiffCostFunction(method)
    x1 = [1e-64, 2.0, 3.0, 4.0, 5.0]  # x1[0] deliberately small: behaviour near zero
    x2 = [9.0, 9.0, 5.0, 5.0, 1.0]
    parameters = sk.RichDoubleMatrix.fromArrays(x1, x2)
    jacobians = sk.RichDoubleMatrix.ofSize(2, 15)
    residuals = sk.RichDoubleArray.ofSize(3)
    assert cost.evaluate(parameters, residuals, jacobians)
    res = residuals.toArray(3)
    assert list(res) == EasyFunctor().apply(np.array(x1), np.array(x2))
    dydx1, dydx2 = jacobians.getRow(0).toArray(15), jacobians.getRow(1).toArray(15)
    for i in range(5):
        assert expect_close(x2[i], dydx1[i], tol) and expect_close(x1[i], dydx2[i], tol)
        assert expect_close(2 * x2[i] * res[0], dydx1[5 + i], tol) and expect_close(2 * x1[i] * res[0], dydx2[5 + i], tol)
        assert expect_close(0.0, dydx1[10 + i], tol) and expect_close(2 * x2[i], dydx2[10 + i], tol)


K_TESTS = [([1.0, 2.0, 3.0, 4.0, 5.0], [9.0, 9.0, 5.0, 5.0, 1.0]), ([0.0, 2.0, 3.0, 0.0, 5.0], [9.0, 9.0, 5.0, 5.0, 1.0]),
           ([1.0, 2.0, 3.0, 1.0, 5.0], [0.0, 9.0, 0.0, 5.0, 0.0]), ([0.0] * 5, [9.0, 9.0, 5.0, 5.0, 1.0]),
           ([1.0, 2.0, 3.0, 4.0, 5.0], [0.0] * 5), ([0.0] * 5, [0.0] * 5)]  # :100-107


@pytest.mark.parametrize("method,tol", [(sk.NumericDiffMethodType.FORWARD, 2.0e-5), (sk.NumericDiffMethodType.CENTRAL, 2.0e-7)])
def test_transcendental_case(method, tol):  # :109-155
    cost = TranscendentalFunctor().toNumericDiffCostFunction(method)
    for x1, x2 in K_TESTS:
        parameters = sk.RichDoubleMatrix.fromArrays(x1, x2)
        jacobians = sk.RichDoubleMatrix.ofSize(2, 10)
        residuals = sk.RichDoubleArray.ofSize(2)
        assert cost.evaluate(parameters, residuals, jacobians)
        dydx1, dydx2 = jacobians.getRow(0).toArray(10), jacobians.getRow(1).toArray(10)
        d = float(np.dot(x1, x2))
        for j in range(5):
            assert expect_close(x2[j] * np.cos(d), dydx1[j], tol) and expect_close(x1[j] * np.cos(d), dydx2[j], tol)
            assert expect_close(-x2[j] * np.exp(-d / 10.0) / 10.0, dydx1[5 + j], tol)
            assert expect_close(-x1[j] * np.exp(-d / 10.0) / 10.0, dydx2[5 + j], tol)


def test_contract_details():
    with pytest.raises(ValueError, match="RIDDERS"):
        EasyFunctor().toNumericDiffCostFunction(sk.NumericDiffMethodType.RIDDERS)
    with pytest.raises(ValueError):
        sk.NumericDiffCostFunctor(0, 1)

    class Failing(sk.NumericDiffCostFunctor):  # an empty result signals failure (NumericDiffCostFunction.scala:100-103)
        def __init__(self):
            super().__init__(1, 1)

        def apply(self, x):
            return [] if x[0] > 1.0 else [x[0]]
    cost = Failing().toNumericDiffCostFunction(sk.NumericDiffMethodType.CENTRAL)
    res = sk.RichDoubleArray.ofSize(1)
    assert not cost.evaluate(sk.RichDoubleMatrix.fromArrays([2.0]), res, None)
    assert cost.evaluate(sk.RichDoubleMatrix.fromArrays([0.5]), res, None) and res.get(0) == 0.5  # residuals only
    jac = sk.RichDoubleMatrix.ofSize(1, 1)
    assert not cost.evaluate(sk.RichDoubleMatrix.fromArrays([1.0]), res, jac)  # the forward probe fails
    # the step is relative with a sqrt(epsilon) floor (:83-93, :117-119)
    opts = sk.NumericDiffOptions()
    assert opts.getRelativeStepSize() == 1e-6
    assert cost.numResiduals() == 1 and cost.parameterBlockSizes() == [1]
