"""The reference's RotationSpec (core/src/test/scala/.../RotationSpec.scala:248-655; itself a port of
ceres' rotation_test.cc) restated over a batch backend ``apply(op, array, row_major, jet_dim)``:
``oracle.rotation_apply`` (CPU, tests/test_rotation_cpu.py) and ``skeres_amd.rotation.apply`` (the device
code, tests/test_gpu_parity.py).  The 10 000-trial loops use the reference's inputs: scala.util.Random(5)
is java.util.Random(5), reproduced below.  Comments cite the spec line of each case."""
import numpy as np

kPi = np.pi
kEpsilon = float(np.spacing(1.0))
kTolerance = 10 * kEpsilon          # :16
kLooseTolerance = 1e-9              # :17
kNumTrials = 10000
kSmallTinyCutoff = int(2 * np.log10(kEpsilon))          # :18  (-31)
kTinyZeroLimit = int(1 + np.log10(5e-324))              # :19  (-322)

(AA2Q, Q2AA, R2Q, R2AA, AA2R, EA2R, Q2SR, Q2R, UQRP, QRP, QPROD, CROSS, DOT, AARP) = range(14)


class JavaRandom:
    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def _next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        return self.seed >> (48 - bits)

    def nextDouble(self):
        return ((self._next(26) << 27) + self._next(27)) * (1.0 / (1 << 53))

    def doubles(self, n):
        return np.array([self.nextDouble() for _ in range(n)])


def near_array(expected, left, tol=kTolerance):
    return np.all(np.abs(np.asarray(expected) - np.asarray(left)) <= tol, axis=-1)


def near_quaternion(expected, left):
    return near_array(expected, left) | near_array(expected, -np.asarray(left))


def normalized_quaternion(q):
    return np.abs(1 - np.sum(np.asarray(q) ** 2, axis=-1)) <= kTolerance


def near_angle_axis(expected, left):  # :77-92
    expected, left = np.atleast_2d(expected), np.atleast_2d(left)
    e = np.linalg.norm(expected, axis=1)
    d = np.linalg.norm(left - expected, axis=1)
    dflip = np.linalg.norm(left + expected, axis=1)
    near_pi = np.abs(e - kPi) < kLooseTolerance
    with np.errstate(divide="ignore", invalid="ignore"):
        delta = np.where(e > 0, np.where(near_pi, np.minimum(d, dflip), d) / e, np.linalg.norm(left, axis=1))
    return delta <= kLooseTolerance


def orthonormal(data):  # :134-149, on the 9 values as stored
    m = np.asarray(data).reshape(-1, 3, 3)
    g = np.einsum("nci,ndi->ncd", m, m)
    return np.all(np.abs(g - np.eye(3)) <= kTolerance, axis=(1, 2))


def is_near(x, y):  # :160-171, elementwise on arrays
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    d = np.abs(x - y)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = d / np.maximum(np.abs(x), np.abs(y))
    return np.where((x == 0) | (y == 0), d <= kTolerance, rel <= kTolerance)


def run_all(apply):
    """Every case of the spec; raises AssertionError with the case's line number."""
    one = lambda op, v, rm=False: apply(op, np.asarray([v], dtype=float), rm, 0)[0]

    # ---- :249-278 angle-axis -> quaternion
    q = one(AA2Q, [0, 0, 0])
    assert normalized_quaternion(q) and q[1] == q[2] == q[3] == 0.0, 249
    for theta, line in ((1e-2, 256), (5e-324 ** 0.75, 264)):
        q = one(AA2Q, [theta, 0, 0])
        assert normalized_quaternion(q) and near_quaternion([np.cos(theta / 2), np.sin(theta / 2), 0, 0], q), line
    q = one(AA2Q, [kPi / 2, 0, 0])
    assert normalized_quaternion(q) and near_quaternion([0.5 * np.sqrt(2), 0.5 * np.sqrt(2), 0, 0], q), 272
    # ---- :279-312 quaternion -> angle-axis
    assert near_angle_axis([0, 0, 0], one(Q2AA, [1, 0, 0, 0])), 279
    assert near_angle_axis([0, kPi, 0], one(Q2AA, [0, 0, 1, 0])), 284
    assert near_angle_axis([0, 0, kPi / 3], one(Q2AA, [np.sqrt(3) / 2, 0, 0, 0.5])), 289
    for theta, line in ((1e-2, 294), (5e-324 ** 0.75, 300)):
        assert near_angle_axis([theta, 0, 0], one(Q2AA, [np.cos(theta / 2), np.sin(theta / 2), 0, 0])), line
    half = 0.75 * kPi
    assert np.linalg.norm(one(Q2AA, [np.cos(half), np.sin(half), 0, 0])) <= kPi, 306
    # ---- :313-326 angle-axis -> quaternion -> angle-axis, Random(5)
    rnd = JavaRandom(5)
    draws = rnd.doubles(4 * kNumTrials).reshape(kNumTrials, 4)
    tmp = draws[:, :3] * 2 - 1
    theta = kPi * (2 * draws[:, 3] - 1)
    aa = tmp * (theta / np.linalg.norm(tmp, axis=1))[:, None]
    q = apply(AA2Q, aa, False, 0)
    assert np.all(normalized_quaternion(q)), 321
    assert np.all(near_angle_axis(aa, apply(Q2AA, q, False, 0))), 323
    # ---- :327-339 quaternion -> angle-axis -> quaternion
    rnd = JavaRandom(5)
    t4 = rnd.doubles(4 * kNumTrials).reshape(kNumTrials, 4) * 2 - 1
    qn = t4 / np.linalg.norm(t4, axis=1)[:, None]
    assert np.all(normalized_quaternion(qn)), 334
    assert np.all(near_quaternion(qn, apply(AA2Q, apply(Q2AA, qn, False, 0), False, 0))), 337
    # ---- :340-409 angle-axis <-> rotation matrix (column major), fixed cases
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    for aa1, expected, back, line in (([0, 0, 0], eye, False, 340), ([1e-24, 2e-24, 3e-24], eye, False, 347),
                                      ([kPi / 2, 0, 0], [1, 0, 0, 0, 0, 1, 0, -1, 0], True, 354),
                                      ([0, kPi, 0], [-1, 0, 0, 0, 1, 0, 0, 0, -1], True, 364),
                                      ([0, 0, kPi / 3], [0.5, np.sqrt(3) / 2, 0, -np.sqrt(3) / 2, 0.5, 0, 0, 0, 1], True, 399)):
        R = one(AA2R, aa1)
        assert orthonormal(R)[0] and near_array(expected, R), line
        if back:
            assert near_angle_axis(aa1, one(R2AA, R)), line
    rnd = JavaRandom(5)  # :373-388 near pi
    d = rnd.doubles(4 * kNumTrials).reshape(kNumTrials, 4)
    tmp = d[:, :3] * 2 - 1
    theta = kPi - 1e-8 * d[:, 3]
    aa = tmp * (theta / np.linalg.norm(tmp, axis=1))[:, None]
    assert np.all(near_angle_axis(aa, apply(R2AA, apply(AA2R, aa, False, 0), False, 0))), 373
    inm = [1, 0, 0, 0, -1, 0, 0, 0, -1]  # :389-398 exactly pi about X
    aa1 = one(R2AA, inm)
    assert near_angle_axis([kPi, 0, 0], aa1), 389
    R = one(AA2R, aa1)
    assert orthonormal(R)[0] and near_array(inm, R), 394
    for scale, line in ((1.0, 410), (1e-16, 424)):  # :410-437 random, and random near zero
        rnd = JavaRandom(5)
        d = rnd.doubles(4 * kNumTrials).reshape(kNumTrials, 4)
        tmp = d[:, :3] * 2 - 1
        theta = scale * (kPi * 2 * d[:, 3] - kPi)
        aa = tmp * (theta / np.linalg.norm(tmp, axis=1))[:, None]
        R = apply(AA2R, aa, False, 0)
        assert np.all(orthonormal(R)), line
        assert np.all(near_angle_axis(aa, apply(R2AA, R, False, 0))), line
    # ---- :438-458 Euler angles
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            for z in (-1, 0, 1):
                if sum(v != 0 for v in (x, y, z)) <= 1:
                    aa_col = one(AA2R, [x, y, z])            # column major
                    aa_row = np.asarray(aa_col).reshape(3, 3).T.ravel()  # transpose3x3 -> "row major adapter"
                    ea = one(EA2R, np.degrees([x, y, z]), True)
                    assert orthonormal(aa_row)[0] and orthonormal(ea)[0] and near_array(ea, aa_row), 438
    rnd = JavaRandom(5)
    ea = 360.0 * (rnd.doubles(3 * kNumTrials).reshape(kNumTrials, 3) * 2.0 - 1.0)
    assert np.all(orthonormal(apply(EA2R, ea, True, 0))), 451
    # ---- :459-562 with jets.  Layout [n, len, 1 + K]
    def jets(rows):  # rows: list of (real, [inf...])
        return np.array([[[r] + list(v) for r, v in rows]], dtype=float)
    for i in range(-2, kSmallTinyCutoff - 1, -1):  # :459-477
        theta = 10.0 ** i
        s, c = np.sin(theta / 2), np.cos(theta / 2)
        got = apply(AA2Q, jets([(theta, [1, 0, 0]), (0, [0, 1, 0]), (0, [0, 0, 1])]), False, 3)[0]
        exp = jets([(c, [-s / 2, 0, 0]), (s, [c / 2, 0, 0]), (0, [0, s / theta, 0]), (0, [0, 0, s / theta])])[0]
        assert np.all(is_near(exp, got)), (459, i)
    exp0 = jets([(1, [0, 0, 0]), (0, [0.5, 0, 0]), (0, [0, 0.5, 0]), (0, [0, 0, 0.5])])[0]
    for i in range(kSmallTinyCutoff, kTinyZeroLimit - 1, -1):  # :478-497
        theta = 10.0 ** i
        got = apply(AA2Q, jets([(theta, [1, 0, 0]), (0, [0, 1, 0]), (0, [0, 0, 1])]), False, 3)[0]
        assert np.all(is_near(exp0, got)), (478, i)
    got = apply(AA2Q, jets([(0, [1, 0, 0]), (0, [0, 1, 0]), (0, [0, 0, 1])]), False, 3)[0]  # :498-509
    assert np.all(is_near(exp0, got)), 498
    for i in range(-2, kSmallTinyCutoff - 1, -1):  # :510-527
        theta = 10.0 ** i
        s, c = np.sin(theta / 2), np.cos(theta / 2)
        got = apply(Q2AA, jets([(c, [1, 0, 0, 0]), (s, [0, 1, 0, 0]), (0, [0, 0, 1, 0]), (0, [0, 0, 0, 1])]), False, 4)[0]
        exp = jets([(theta, [-2 * s, 2 * c, 0, 0]), (0, [0, 0, theta / s, 0]), (0, [0, 0, 0, theta / s])])[0]
        assert np.all(is_near(exp, got)), (510, i)
    for i in range(kSmallTinyCutoff, kTinyZeroLimit // 2 - 1, -1):  # :528-549 (the reference stops at kTinyZeroLimit / 2)
        theta = 10.0 ** i
        s, c = np.sin(theta / 2), np.cos(theta / 2)
        got = apply(Q2AA, jets([(c, [1, 0, 0, 0]), (s, [0, 1, 0, 0]), (0, [0, 0, 1, 0]), (0, [0, 0, 0, 1])]), False, 4)[0]
        exp = jets([(theta, [-theta, 2.0, 0, 0]), (0, [0, 0, 2.0, 0]), (0, [0, 0, 0, 2.0])])[0]
        assert np.all(is_near(exp, got)), (528, i)
    got = apply(Q2AA, jets([(1, [1, 0, 0, 0]), (0, [0, 1, 0, 0]), (0, [0, 0, 1, 0]), (0, [0, 0, 0, 1])]), False, 4)[0]  # :550-562
    exp = jets([(0, [0, 2.0, 0, 0]), (0, [0, 0, 2.0, 0]), (0, [0, 0, 0, 2.0])])[0]
    assert np.all(is_near(exp, got)), 550
    # ---- :563-593 canned quaternion -> (scaled) rotation, row major
    q = [+0.1956830471754074, -0.0150618562474847, +0.7634572982788086, -0.3019454777240753]
    Q = [-0.6355194033477252, 0.0951730541682254, 0.3078870197911186, -0.1411693904792992, 0.5297609702153905, -0.4551502574482019,
         -0.2896955822708862, -0.4669396571547050, -0.4536309793389248]
    NQ = [-0.8918859164053080, 0.1335655625725649, 0.4320876677394745, -0.1981166751680096, 0.7434648665444399, -0.6387564287225856,
          -0.4065578619806013, -0.6553016349046693, -0.6366242786393164]
    assert near_array(Q, one(Q2SR, q, True)), 563
    assert near_array(NQ, one(Q2R, q, True)), 590
    # ---- :594-615 rotate by a quaternion == rotate by its matrix
    rnd = JavaRandom(5)
    d = rnd.doubles(7 * kNumTrials).reshape(kNumTrials, 7)
    quat = d[:, :4] / np.linalg.norm(d[:, :4], axis=1)[:, None]
    assert np.all(normalized_quaternion(quat)), 601
    p = 10.0 * (2 * d[:, 4:] - 1)
    r1 = apply(UQRP, np.concatenate([quat, p], axis=1), False, 0)
    R = apply(Q2R, quat, True, 0).reshape(-1, 3, 3)
    assert np.all(near_array(np.einsum("nij,nj->ni", R, p), r1, kLooseTolerance)), 594
    # ---- :616-655 rotate by an angle-axis == rotate by its matrix (and near zero angle)
    for scale, line in ((kPi, 616), (1.0e-16, 636)):
        rnd = JavaRandom(5)
        d = rnd.doubles(7 * kNumTrials).reshape(kNumTrials, 7)
        theta = scale * (2 * d[:, 0] - 1)
        v = 2 * d[:, 1:4] - 1
        aa = theta[:, None] * (v / np.linalg.norm(v, axis=1)[:, None])
        p = 10.0 * (2 * d[:, 4:] - 1)
        r1 = apply(AARP, np.concatenate([aa, p], axis=1), False, 0)
        Rc = apply(AA2R, aa, False, 0).reshape(-1, 3, 3)  # column major: Rc[n, j, i] = R(i, j)
        assert np.all(near_array(np.einsum("nji,nj->ni", Rc, p), r1, kLooseTolerance)), line
    return True
