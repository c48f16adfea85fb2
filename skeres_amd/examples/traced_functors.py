"""Functor bodies written generically, as a user of the reference writes them (CORE/CostFunctor.scala:40-51), and
handed to the device as recordings (``skeres_amd.TracedCostFunctor``): none of these uses the device functor registry.

    TracedSnavelyReprojectionError   EX/SimpleBundleAdjuster.scala:79-119 over Rotation.angleAxisRotatePoint
                                     (CORE/Rotation.scala:449-522, whose `if` on theta^2 becomes a ``where``)
    TracedExponentialResidual        EX/CurveFitting.scala:92-98
    TracedPowell                     EX/Powell.scala:14-53, the four residuals as one functor over four 1-blocks
    TracedPinholeReprojectionError   the bundle adjuster's functor with the camera's intrinsics CAPTURED instead of optimised: a
                                     (2; 6, 3) block shape — DENSE_SCHUR on a shape other than the reference's (2; 9, 3)
"""
import numpy as np

from .. import TracedCostFunctor
from .. import tape as T


def angle_axis_rotate_point(w, pt):
    """Rotation.angleAxisRotatePoint for any T; the branch at CORE/Rotation.scala:458 as a select of two recorded arms."""
    theta2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]

    def far():  # CORE/Rotation.scala:459-492
        theta = T.sqrt(theta2)
        s, c = T.sin(theta), T.cos(theta)
        ti = 1.0 / theta
        wn = [w[0] * ti, w[1] * ti, w[2] * ti]
        x = [wn[1] * pt[2] - wn[2] * pt[1], wn[2] * pt[0] - wn[0] * pt[2], wn[0] * pt[1] - wn[1] * pt[0]]
        tmp = ((wn[0] * pt[0] + wn[1] * pt[1]) + wn[2] * pt[2]) * (1.0 - c)
        return [(pt[i] * c + x[i] * s) + wn[i] * tmp for i in range(3)]

    def near():  # :493-521
        return [pt[0] + (w[1] * pt[2] - w[2] * pt[1]), pt[1] + (w[2] * pt[0] - w[0] * pt[2]), pt[2] + (w[0] * pt[1] - w[1] * pt[0])]
    return T.where(theta2 > float(np.finfo(np.float64).eps), far, near)


class TracedSnavelyReprojectionError(TracedCostFunctor):
    def __init__(self, observedX, observedY):
        super().__init__(2, 9, 3, captured=(observedX, observedY))

    def apply(self, camera, point):
        ox, oy = self.captured_values()
        p = angle_axis_rotate_point(camera[0:3], point)
        p = [p[0] + camera[3], p[1] + camera[4], p[2] + camera[5]]
        xp, yp = (-p[0]) / p[2], (-p[1]) / p[2]
        r2 = xp * xp + yp * yp
        distortion = 1.0 + r2 * (camera[7] + camera[8] * r2)
        fd = camera[6] * distortion
        return [fd * xp - ox, fd * yp - oy]


class TracedExponentialResidual(TracedCostFunctor):
    def __init__(self, x, y):
        super().__init__(1, 1, 1, captured=(x, y))

    def apply(self, m, c):
        x, y = self.captured_values()
        return [y - T.exp(m[0] * x + c[0])]


class TracedPowell(TracedCostFunctor):
    def __init__(self):
        super().__init__(4, 1, 1, 1, 1)

    def apply(self, x1, x2, x3, x4):
        a, b, c, d = x1[0], x2[0], x3[0], x4[0]
        return [a + 10.0 * b, float(np.sqrt(5.0)) * (c - d), (b - 2.0 * c) * (b - 2.0 * c), float(np.sqrt(10.0)) * (a - d) * (a - d)]


class TracedPinholeReprojectionError(TracedCostFunctor):
    """EX/SimpleBundleAdjuster.scala:79-119 over a camera of SIX parameters (angle-axis, translation); focal length and the two
    distortion coefficients are captured doubles of the closure, like the observation.  The same residual as
    SnavelyReprojectionError on a camera whose intrinsics are held constant."""

    def __init__(self, observedX, observedY, focal, k1, k2, residuals=2):
        super().__init__(residuals, 6, 3, captured=(observedX, observedY, focal, k1, k2))

    def apply(self, camera, point):
        ox, oy, focal, k1, k2 = self.captured_values()
        p = angle_axis_rotate_point(camera[0:3], point)
        p = [p[0] + camera[3], p[1] + camera[4], p[2] + camera[5]]
        xp, yp = (-p[0]) / p[2], (-p[1]) / p[2]
        r2 = xp * xp + yp * yp
        distortion = 1.0 + r2 * (k1 + k2 * r2)
        fd = focal * distortion
        return [fd * xp - ox, fd * yp - oy][: self.kNumResiduals]
