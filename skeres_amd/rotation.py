"""Host-side mirror of CORE/Rotation.scala:14-522 over ``sk_rotation_apply``: the functions run ON THE
DEVICE (the code device functors call, csrc/rotation.hpp), for plain doubles and for Jets.

    Rotation.angleAxisToQuaternion(aa) -> Quaternion          Rotation.scala:72
    Rotation.quaternionToAngleAxis(q) -> [3]                  :104
    Rotation.rotationMatrixToQuaternion(R) / ...ToAngleAxis   :156-204   (R: 9 values column-major, or a MatrixAdapter)
    Rotation.angleAxisToRotationMatrix(aa) -> ColumnMajorMatrixAdapter3x3   :206
    Rotation.eulerAnglesToRotationMatrix(ea) -> RowMajorMatrixAdapter3x3    :264
    Rotation.quaternionToScaledRotation(q) / quaternionToRotation(q) -> RowMajorMatrixAdapter3x3   :322-381
    Rotation.unitQuaternionRotatePoint / quaternionRotatePoint / quaternionProduct / crossProduct /
    dotProduct / angleAxisRotatePoint                         :393-522

Values are floats or ``Jet`` (spire.math.Jet as the reference uses it: real part + infinitesimal vector).
"""
import ctypes as C

import numpy as np

from .api import SkeresError, _check, _dp, lib

(ANGLE_AXIS_TO_QUATERNION, QUATERNION_TO_ANGLE_AXIS, ROTATION_MATRIX_TO_QUATERNION, ROTATION_MATRIX_TO_ANGLE_AXIS,
 ANGLE_AXIS_TO_ROTATION_MATRIX, EULER_ANGLES_TO_ROTATION_MATRIX, QUATERNION_TO_SCALED_ROTATION, QUATERNION_TO_ROTATION,
 UNIT_QUATERNION_ROTATE_POINT, QUATERNION_ROTATE_POINT, QUATERNION_PRODUCT, CROSS_PRODUCT, DOT_PRODUCT,
 ANGLE_AXIS_ROTATE_POINT) = range(14)
IN_LEN = (3, 4, 9, 9, 3, 3, 4, 4, 7, 7, 8, 6, 6, 6)
OUT_LEN = (4, 3, 4, 3, 9, 9, 9, 9, 3, 3, 4, 3, 1, 3)


class Jet:
    """spire.math.Jet[Double]: ``Jet(x, k)`` = x + e_k (needs a dimension), ``Jet(x, [..])`` explicit.

    Arithmetic and the elementary functions below follow spire's Jet (Jet.scala [ext]; the rules are the chain rule):
    enough to write a generic functor on the host — ``HostAutoDiffCostFunctor`` in api.py — as the reference writes
    them over ``T: Field: Trig: NRoot``.  Comparisons look at the real part only, as spire's ``Order[Jet]`` does."""

    __slots__ = ("real", "infinitesimal")

    def __init__(self, real, inf, dim=None):
        self.real = float(real)
        if np.isscalar(inf):
            if dim is None:
                raise ValueError("Jet(x, k) needs dim (the reference's implicit JetDim)")
            v = np.zeros(dim)
            v[int(inf)] = 1.0
            self.infinitesimal = v
        else:
            self.infinitesimal = np.asarray(inf, dtype=np.float64)

    def __repr__(self):
        return "Jet(%r, %r)" % (self.real, self.infinitesimal.tolist())

    def __float__(self):
        return self.real

    @staticmethod
    def _split(other):
        return (other.real, other.infinitesimal) if isinstance(other, Jet) else (float(other), 0.0)

    def __neg__(self):
        return Jet(-self.real, -self.infinitesimal)

    def __pos__(self):
        return self

    def __add__(self, other):
        r, v = Jet._split(other)
        return Jet(self.real + r, self.infinitesimal + v)

    __radd__ = __add__

    def __sub__(self, other):
        r, v = Jet._split(other)
        return Jet(self.real - r, self.infinitesimal - v)

    def __rsub__(self, other):
        r, v = Jet._split(other)
        return Jet(r - self.real, v - self.infinitesimal)

    def __mul__(self, other):
        r, v = Jet._split(other)
        return Jet(self.real * r, self.infinitesimal * r + v * self.real)

    __rmul__ = __mul__

    def __truediv__(self, other):
        r, v = Jet._split(other)
        inv = 1.0 / r
        q = self.real * inv
        return Jet(q, (self.infinitesimal - v * q) * inv)

    def __rtruediv__(self, other):
        r, v = Jet._split(other)
        inv = 1.0 / self.real
        q = r * inv
        return Jet(q, (v - self.infinitesimal * q) * inv)

    def __pow__(self, p):
        if isinstance(p, Jet):
            return exp(p * log(self))
        return Jet(self.real ** p, self.infinitesimal * (p * self.real ** (p - 1)))

    def __lt__(self, other):
        return self.real < Jet._split(other)[0]

    def __le__(self, other):
        return self.real <= Jet._split(other)[0]

    def __gt__(self, other):
        return self.real > Jet._split(other)[0]

    def __ge__(self, other):
        return self.real >= Jet._split(other)[0]

    def __abs__(self):
        return -self if self.real < 0.0 else self

    # spire-style method names, as generic reference code calls them
    def sqrt(self):
        return sqrt(self)


def _unary(f, df):
    def g(x):
        if isinstance(x, Jet):
            return Jet(f(x.real), x.infinitesimal * df(x.real))
        return f(float(x))
    return g


sqrt = _unary(np.sqrt, lambda x: 0.5 / np.sqrt(x))
exp = _unary(np.exp, np.exp)
log = _unary(np.log, lambda x: 1.0 / x)
sin = _unary(np.sin, np.cos)
cos = _unary(np.cos, lambda x: -np.sin(x))
tan = _unary(np.tan, lambda x: 1.0 / np.cos(x) ** 2)
asin = _unary(np.arcsin, lambda x: 1.0 / np.sqrt(1.0 - x * x))
acos = _unary(np.arccos, lambda x: -1.0 / np.sqrt(1.0 - x * x))
atan = _unary(np.arctan, lambda x: 1.0 / (1.0 + x * x))


def atan2(y, x):
    if not isinstance(y, Jet) and not isinstance(x, Jet):
        return float(np.arctan2(y, x))
    (yr, yv), (xr, xv) = Jet._split(y), Jet._split(x)
    d = xr * xr + yr * yr
    return Jet(np.arctan2(yr, xr), (yv * xr - xv * yr) / d)


class Quaternion:
    """spire.math.Quaternion(r, i, j, k) of floats or Jets."""

    def __init__(self, r, i=0.0, j=0.0, k=0.0):
        self.r, self.i, self.j, self.k = r, i, j, k

    def parts(self):
        return [self.r, self.i, self.j, self.k]

    def abs(self):
        return float(np.sqrt(sum(float(v) ** 2 for v in self.parts())))

    def normalize(self):
        n = self.abs()
        return Quaternion(*[v / n for v in self.parts()])

    @property
    def isReal(self):
        return self.i == 0.0 and self.j == 0.0 and self.k == 0.0

    def __repr__(self):
        return "Quaternion(%r, %r, %r, %r)" % tuple(self.parts())


class MatrixAdapter:  # Rotation.scala:14-57
    def __init__(self, data, rowStride, colStride):
        self.data, self.rowStride, self.colStride = list(data), rowStride, colStride

    @property
    def isRowMajor(self):
        return self.colStride == 1

    @property
    def isColMajor(self):
        return self.rowStride == 1

    @property
    def numRows(self):
        return len(self.data) // self.rowStride if self.isRowMajor else len(self.data) // self.numCols

    @property
    def numCols(self):
        return len(self.data) // self.colStride if self.isColMajor else len(self.data) // self.numRows

    def __call__(self, i, j):
        return self.data[i * self.rowStride + j * self.colStride]

    def set(self, i, j, x):
        self.data[i * self.rowStride + j * self.colStride] = x


class RowMajorMatrixAdapter3x3(MatrixAdapter):
    def __init__(self, data):
        super().__init__(data, 3, 1)


class ColumnMajorMatrixAdapter3x3(MatrixAdapter):
    def __init__(self, data):
        super().__init__(data, 1, 3)


def apply(op, values, row_major=False, jet_dim=0):
    """Batch form: [n, in_len] (jet_dim K > 0: [n, in_len, 1 + K]) -> [n, out_len(, 1 + K)], on the device."""
    a = np.ascontiguousarray(values, dtype=np.float64)
    n = a.shape[0]
    want = (n, IN_LEN[op]) + ((1 + jet_dim,) if jet_dim else ())
    if a.shape != want:
        raise ValueError("rotation op %d expects an array of shape %r, got %r" % (op, want, a.shape))
    out = np.zeros((n, OUT_LEN[op]) + ((1 + jet_dim,) if jet_dim else ()))
    _check(lib().sk_rotation_apply(int(op), int(bool(row_major)), int(jet_dim), a.ctypes.data_as(_dp), n, out.ctypes.data_as(_dp)))
    return out


def _one(op, values, row_major=False):
    """One item; values is a flat list of floats or Jets; returns a list of the same kind."""
    values = list(values)
    if any(isinstance(v, Jet) for v in values):
        dim = next(len(v.infinitesimal) for v in values if isinstance(v, Jet))
        a = np.zeros((1, len(values), 1 + dim))
        for e, v in enumerate(values):
            if isinstance(v, Jet):
                a[0, e, 0], a[0, e, 1:] = v.real, v.infinitesimal
            else:
                a[0, e, 0] = float(v)
        out = apply(op, a, row_major, dim)[0]
        return [Jet(o[0], o[1:].copy()) for o in out]
    out = apply(op, np.asarray([values], dtype=np.float64), row_major)[0]
    return [float(v) for v in out]


def _matrix_arg(R):
    if isinstance(R, MatrixAdapter):
        if R.isColMajor and R.colStride == 3:
            return R.data, False
        if R.isRowMajor and R.rowStride == 3:
            return R.data, True
        raise ValueError("only 3x3 row-major or column-major adapters")
    return list(R), False  # a bare array is COLUMN MAJOR (Rotation.scala:151-160, :195-196)


class Rotation:
    @staticmethod
    def angleAxisToQuaternion(angleAxis):
        return Quaternion(*_one(ANGLE_AXIS_TO_QUATERNION, angleAxis))

    @staticmethod
    def quaternionToAngleAxis(quaternion):
        return _one(QUATERNION_TO_ANGLE_AXIS, quaternion.parts())

    @staticmethod
    def rotationMatrixToQuaternion(R):
        data, rm = _matrix_arg(R)
        return Quaternion(*_one(ROTATION_MATRIX_TO_QUATERNION, data, rm))

    @staticmethod
    def rotationMatrixToAngleAxis(R):
        data, rm = _matrix_arg(R)
        return _one(ROTATION_MATRIX_TO_ANGLE_AXIS, data, rm)

    @staticmethod
    def angleAxisToRotationMatrix(angleAxis):
        return ColumnMajorMatrixAdapter3x3(_one(ANGLE_AXIS_TO_ROTATION_MATRIX, angleAxis, False))

    @staticmethod
    def eulerAnglesToRotationMatrix(euler):
        return RowMajorMatrixAdapter3x3(_one(EULER_ANGLES_TO_ROTATION_MATRIX, euler, True))

    @staticmethod
    def quaternionToScaledRotation(q):
        return RowMajorMatrixAdapter3x3(_one(QUATERNION_TO_SCALED_ROTATION, q.parts(), True))

    @staticmethod
    def quaternionToRotation(q):
        try:
            return RowMajorMatrixAdapter3x3(_one(QUATERNION_TO_ROTATION, q.parts(), True))
        except SkeresError as e:
            if "zero quaternion" in str(e):
                raise ValueError("requirement failed") from e  # Rotation.scala:372
            raise

    @staticmethod
    def unitQuaternionRotatePoint(q, pt):
        return _one(UNIT_QUATERNION_ROTATE_POINT, q.parts() + list(pt))

    @staticmethod
    def quaternionRotatePoint(q, pt):
        return _one(QUATERNION_ROTATE_POINT, q.parts() + list(pt))

    @staticmethod
    def quaternionProduct(z, w):
        return Quaternion(*_one(QUATERNION_PRODUCT, z.parts() + w.parts()))

    @staticmethod
    def crossProduct(x, y):
        return _one(CROSS_PRODUCT, list(x) + list(y))

    @staticmethod
    def dotProduct(x, y):
        return _one(DOT_PRODUCT, list(x) + list(y))[0]

    @staticmethod
    def angleAxisRotatePoint(angleAxis, pt):
        return _one(ANGLE_AXIS_ROTATE_POINT, list(angleAxis) + list(pt))
