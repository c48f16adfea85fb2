"""Pins the oracle's Rotation restatement (oracle/rotation.hpp) with the reference's own RotationSpec
(tests/rotation_spec.py), and checks the host-side pieces of skeres_amd.rotation that need no device."""
import numpy as np
import pytest

import oracle
import skeres_amd as sk
from rotation_spec import run_all, JavaRandom


@pytest.fixture(scope="module", autouse=True)
def _oracle_built():
    oracle.build()


def test_reference_rotation_spec_on_the_oracle():
    assert run_all(lambda op, a, rm, k: oracle.rotation_apply(op, a, rm, k))


def test_matrix_adapters():  # RotationSpec.scala:224-247
    a = [1, 2, 3, 4, 5, 6, 7, 8, 9]
    m = sk.ColumnMajorMatrixAdapter3x3(a)
    assert m.data == a and m.numRows == 3 and m.numCols == 3 and m.rowStride == 1 and m.colStride == 3
    assert all(m(i, j) == a[j * 3 + i] for i in range(3) for j in range(3))
    m = sk.RowMajorMatrixAdapter3x3(a)
    assert m.data == a and m.numRows == 3 and m.numCols == 3 and m.rowStride == 3 and m.colStride == 1
    assert all(m(i, j) == a[i * 3 + j] for i in range(3) for j in range(3))


def test_remaining_functions_against_numpy():
    rnd = JavaRandom(7)
    d = rnd.doubles(8 * 200).reshape(200, 8) * 2 - 1
    z, w = d[:, :4], d[:, 4:]
    zw = oracle.rotation_apply(10, d)  # Hamilton product
    ref = np.stack([z[:, 0] * w[:, 0] - np.sum(z[:, 1:] * w[:, 1:], axis=1),
                    *(z[:, 0, None] * w[:, 1:] + w[:, 0, None] * z[:, 1:] + np.cross(z[:, 1:], w[:, 1:])).T], axis=1)
    np.testing.assert_allclose(zw, ref, rtol=1e-14, atol=1e-15)
    xy = d[:, :6]
    np.testing.assert_allclose(oracle.rotation_apply(11, xy), np.cross(xy[:, :3], xy[:, 3:]), rtol=1e-14, atol=1e-16)  # the true cross product
    np.testing.assert_allclose(oracle.rotation_apply(12, xy)[:, 0], np.sum(xy[:, :3] * xy[:, 3:], axis=1), rtol=1e-14, atol=1e-16)
    # a non-unit quaternion rotates like its normalisation; row-major and column-major outputs are transposes
    qp = d[:, :7]
    qn = qp.copy()
    qn[:, :4] /= np.linalg.norm(qn[:, :4], axis=1)[:, None]
    np.testing.assert_allclose(oracle.rotation_apply(9, qp), oracle.rotation_apply(8, qn), rtol=1e-12, atol=1e-14)
    Rr = oracle.rotation_apply(7, qp[:, :4], row_major=True).reshape(-1, 3, 3)
    Rc = oracle.rotation_apply(7, qp[:, :4], row_major=False).reshape(-1, 3, 3)
    np.testing.assert_array_equal(Rr, Rc.transpose(0, 2, 1))
    with pytest.raises(ValueError):
        oracle.rotation_apply(7, [[0, 0, 0, 0]])  # `require(norm != 0)`, Rotation.scala:372
    # rotation matrix -> quaternion, all four branches (trace >= 0 and each dominant diagonal entry)
    for aa in ([0.3, -0.2, 0.1], [3.0, 0.1, 0.1], [0.1, 3.0, 0.1], [0.1, 0.1, 3.0]):
        R = oracle.rotation_apply(4, [aa])
        q = oracle.rotation_apply(2, R)[0]
        q2 = oracle.rotation_apply(0, [aa])[0]
        assert np.allclose(q, q2, atol=1e-14) or np.allclose(q, -q2, atol=1e-14)
