"""PredefinedLocalParameterizations (ceres.i:186-210): the oracle's restatement (oracle/parameterization.hpp) against
properties that do not depend on it — central differences of Plus for every Jacobian, the manifolds' invariants, the
documented special cases — and the host-side argument checks of the C ABI.  The reference holds no test for them
(parity unpinned: Ceres is an un-vendored dependency)."""
import numpy as np
import pytest

import oracle
import skeres_amd as sk

CASES = [(("identity",), 3), (("subset", [1]), 3), (("subset", [0, 3]), 5), (("quaternion",), 4), (("homogeneous",), 2),
         (("homogeneous",), 4), (("homogeneous",), 7), (("constant",), 3)]


def points(size, rng, kind):
    xs = [rng.normal(size=size) for _ in range(6)]
    if kind == "homogeneous":  # both branches of the Householder vector: last entry negative / positive, and x = +-e_n
        e = np.zeros(size)
        e[-1] = 1.0
        xs += [e * 2.5, -e * 0.5, np.abs(xs[0]), -np.abs(xs[1])]
    if kind == "quaternion":
        xs = [x / np.linalg.norm(x) for x in xs]
    return xs


@pytest.mark.parametrize("p,size", CASES)
def test_jacobian_is_the_derivative_of_plus_at_zero(p, size):
    rng = np.random.default_rng(7)
    ls = oracle.parameterization_local_size(p, size)
    for x in points(size, rng, p[0]):
        J = oracle.parameterization_jacobian(p, x)
        assert J.shape == (size, ls)
        h = 1e-6
        for c in range(ls):
            e = np.zeros(ls)
            e[c] = h
            fd = (oracle.parameterization_plus(p, x, e) - oracle.parameterization_plus(p, x, -e)) / (2 * h)
            np.testing.assert_allclose(J[:, c], fd, atol=2e-9 * max(1.0, np.linalg.norm(x)))
        np.testing.assert_array_equal(oracle.parameterization_plus(p, x, np.zeros(ls)), x)  # Plus(x, 0) = x exactly


def test_manifold_invariants_and_special_cases():
    rng = np.random.default_rng(11)
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        d = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 0.5, 3.0])
        qp = oracle.parameterization_plus(("quaternion",), q, d)
        assert abs(np.linalg.norm(qp) - 1.0) < 1e-14  # stays a unit quaternion
        # q_plus = [cos|d|, sin|d| d/|d|] (x) q: for q = identity the update itself
        ident = oracle.parameterization_plus(("quaternion",), [1.0, 0.0, 0.0, 0.0], d)
        nd = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        np.testing.assert_allclose(ident, np.concatenate([[np.cos(nd)], np.sin(nd) / nd * d]), rtol=2e-15, atol=1e-300)
        for n in (2, 3, 6):
            x = rng.normal(size=n) * 3.0
            dd = rng.normal(size=n - 1) * 0.7
            xp = oracle.parameterization_plus(("homogeneous",), x, dd)
            assert abs(np.linalg.norm(xp) - np.linalg.norm(x)) < 1e-13 * np.linalg.norm(x)  # the norm is kept
            # the Jacobian's columns are tangent to the sphere through x
            np.testing.assert_allclose(oracle.parameterization_jacobian(("homogeneous",), x).T @ x, 0.0, atol=1e-13 * (x @ x))
    # x = |x| e_n: Householder reflection is the identity, Plus moves along the first n-1 axes
    xp = oracle.parameterization_plus(("homogeneous",), [0.0, 0.0, 2.0], [0.2, 0.0])
    np.testing.assert_allclose(xp, [2.0 * np.sin(0.1), 0.0, 2.0 * np.cos(0.1)], rtol=1e-15, atol=1e-16)
    # the quaternion Jacobian at the identity: d/d delta of [cos|d|, sin|d| d/|d|] = [0; I]
    np.testing.assert_array_equal(oracle.parameterization_jacobian(("quaternion",), [1.0, 0.0, 0.0, 0.0]), np.vstack([np.zeros((1, 3)), np.eye(3)]))
    np.testing.assert_array_equal(oracle.parameterization_plus(("subset", [1]), [1.0, 2.0, 3.0], [10.0, 20.0]), [11.0, 2.0, 23.0])
    np.testing.assert_array_equal(oracle.parameterization_plus(("constant",), [1.0, 2.0], []), [1.0, 2.0])


def test_c_abi_sizes_and_argument_checks():
    P = sk.PredefinedLocalParameterizations
    assert (P.identity(5).globalSize(), P.identity(5).localSize()) == (5, 5)
    assert (P.subset(4, [0, 2]).globalSize(), P.subset(4, [0, 2]).localSize()) == (4, 2)
    assert (P.quaternion().globalSize(), P.quaternion().localSize()) == (4, 3)
    assert (P.homogeneousVector(4).globalSize(), P.homogeneousVector(4).localSize()) == (4, 3)
    for bad in (lambda: P.subset(3, [3]), lambda: P.subset(3, [-1]), lambda: P.subset(3, [1, 1]), lambda: P.homogeneousVector(1),
                lambda: P.identity(0), lambda: P.identity(17)):
        with pytest.raises(ValueError):
            bad()
    # Problem: a parameterization must match the block it is set on; blocks must exist
    problem = sk.Problem()
    x, y = sk.DoubleArray(4), sk.DoubleArray(3)
    problem.addParameterBlock(x, 4, P.quaternion())
    problem.addParameterBlock(y, 3)
    with pytest.raises(ValueError):
        problem.setParameterization(y, P.quaternion())
    with pytest.raises(ValueError):
        problem.addParameterBlock(x, 3)  # registered with another size
    with pytest.raises(ValueError):
        problem.setParameterBlockConstant(sk.DoubleArray(2))  # not part of the problem
    problem.setParameterization(y, P.subset(3, [0]))
    problem.setParameterBlockConstant(y)
    problem.setParameterBlockVariable(y)
    problem.setParameterization(y, None)
