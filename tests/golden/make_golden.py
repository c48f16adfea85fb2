#!/usr/bin/env python3
"""Generates tests/golden/*.json — independent cross-check vectors for the oracle.

The reference itself cannot run anywhere in this pipeline (Scala + SWIG/JNI +
native Ceres; no JVM, no Ceres), so these vectors do NOT come from it.  They
come from tools that share no code with either the oracle or the product:

  snavely_jacobians.json   SymPy closed-form differentiation of the Snavely
                           reprojection model, evaluated with 50-digit mpmath
  optima.json              SciPy least_squares (trf, exact Jacobians from SymPy /
                           analytic formulas) minimisers of CurveFitting, Powell
                           and a tiny bundle-adjustment problem
  lm_step.json             ONE Levenberg-Marquardt step of a 16-camera bundle-adjustment problem, computed with none of
                           the oracle's or the product's code: residuals and Jacobians from the SymPy closed form
                           evaluated with 40-digit mpmath, Jacobi scaling, LM diagonal and the damped normal
                           equations (J^T J + D^2) y = J^T r formed in extended precision and solved with NumPy +
                           iterative refinement; candidate cost again from the closed form.  Pins the linear
                           algebra of one iteration (oracle: tests/test_oracle_kat.py; GPU: tests/test_gpu_parity.py)
  robust_optima.json       SciPy least_squares with loss = 'cauchy' / 'huber' / 'soft_l1'
                           (the same rho as Ceres' CauchyLoss / HuberLoss / SoftLOneLoss with
                           f_scale = a) on the RobustCurveFitting samples and on a tiny
                           bundle-adjustment problem with gross outliers

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np
import sympy as sym
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from skeres_amd import bal  # noqa: E402  (generator only: numpy, no native code)


def snavely_symbolic():
    c = sym.symbols("c0:9", real=True)
    X = sym.symbols("X0:3", real=True)
    ox, oy = sym.symbols("ox oy", real=True)
    aa = sym.Matrix(c[0:3])
    pt = sym.Matrix(X)
    theta = sym.sqrt(aa.dot(aa))
    w = aa / theta
    p = pt * sym.cos(theta) + w.cross(pt) * sym.sin(theta) + w * (w.dot(pt)) * (1 - sym.cos(theta))
    p = p + sym.Matrix(c[3:6])
    xp, yp = -p[0] / p[2], -p[1] / p[2]
    r2 = xp * xp + yp * yp
    d = 1 + r2 * (c[7] + c[8] * r2)
    res = sym.Matrix([c[6] * d * xp - ox, c[6] * d * yp - oy])
    J = res.jacobian(list(c) + list(X))
    return list(c) + list(X) + [ox, oy], res, J


def make_snavely():
    mp.mp.dps = 50
    syms, res, J = snavely_symbolic()
    f_res = sym.lambdify(syms, res, "mpmath")
    f_jac = sym.lambdify(syms, J, "mpmath")
    rng = np.random.default_rng(20261003)
    cases = []
    for k in range(12):
        cam = np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 1, 3), [rng.uniform(400, 1200)],
                              [rng.normal(0, 1e-6)], [rng.normal(0, 1e-11)]])
        pt = rng.normal(0, 1, 3) + np.array([0, 0, -6.0])
        obs = rng.normal(0, 150, 2)
        args = [mp.mpf(float(v)) for v in list(cam) + list(pt) + list(obs)]
        r = f_res(*args)
        Jm = f_jac(*args)
        cases.append({"camera": cam.tolist(), "point": pt.tolist(), "observed": obs.tolist(),
                      "residuals": [float(r[i]) for i in range(2)],
                      "jacobian": [[float(Jm[i, j]) for j in range(12)] for i in range(2)]})
    with open(os.path.join(HERE, "snavely_jacobians.json"), "w") as f:
        json.dump({"source": "sympy closed form + mpmath 50 digits", "cases": cases}, f, indent=1)


def make_optima():
    out = {}
    data = np.loadtxt(os.path.join(HERE, "curve_fitting_data.txt"))
    xs, ys = data[:, 0], data[:, 1]

    def f(p): return ys - np.exp(p[0] * xs + p[1])
    def j(p): e = np.exp(p[0] * xs + p[1]); return np.stack([-xs * e, -e], axis=1)
    s = least_squares(f, [0.0, 0.0], jac=j, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    out["curve_fitting"] = {"x": s.x.tolist(), "cost": float(0.5 * np.sum(s.fun ** 2))}

    s5, s10 = np.sqrt(5.0), np.sqrt(10.0)

    def fp(x): return np.array([x[0] + 10 * x[1], s5 * x[2] - x[3], (x[1] - 2 * x[2]) ** 2, s10 * (x[0] - x[3]) ** 2])
    s = least_squares(fp, [3.0, -1.0, 0.0, 1.0], method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    out["powell"] = {"x": s.x.tolist(), "cost": float(0.5 * np.sum(s.fun ** 2))}

    prob = bal.generate(4, 24, 90, seed=42)
    C, P = prob.num_cameras, prob.num_points

    def fb(x):
        cams = x[:9 * C].reshape(-1, 9)[prob.camera_index]
        pts = x[9 * C:].reshape(-1, 3)[prob.point_index]
        pred, _ = bal.snavely_project(cams, pts)
        return (pred - prob.observations).ravel()
    s = least_squares(fb, prob.parameters, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=400)
    out["tiny_bal"] = {"shape": [C, P, prob.num_observations], "seed": 42, "cost": float(0.5 * np.sum(s.fun ** 2)),
                       "initial_cost": float(0.5 * np.sum(fb(prob.parameters) ** 2))}
    with open(os.path.join(HERE, "optima.json"), "w") as f:
        json.dump({"source": "scipy.optimize.least_squares (trf)", **out}, f, indent=1)


def make_robust_optima():
    out = {}
    data = np.loadtxt(os.path.join(HERE, "robust_curve_fitting_data.txt"))
    xs, ys = data[:, 0], data[:, 1]

    def f(p): return ys - np.exp(p[0] * xs + p[1])
    def j(p): e = np.exp(p[0] * xs + p[1]); return np.stack([-xs * e, -e], axis=1)

    def rho_cost(kind, a, r2):  # 1/2 sum rho(s), Ceres conventions
        if kind == "cauchy":
            return 0.5 * np.sum(a * a * np.log1p(r2 / (a * a)))
        if kind == "huber":
            return 0.5 * np.sum(np.where(r2 <= a * a, r2, 2 * a * np.sqrt(r2) - a * a))
        return 0.5 * np.sum(2 * a * a * (np.sqrt(1 + r2 / (a * a)) - 1))

    cf = {}
    for kind, a in (("cauchy", 0.5), ("huber", 0.3), ("soft_l1", 0.4)):
        s = least_squares(f, [0.0, 0.0], jac=j, method="trf", loss=kind, f_scale=a, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        cf[kind] = {"a": a, "x": s.x.tolist(), "cost": float(rho_cost(kind, a, s.fun ** 2)), "scipy_cost": float(s.cost)}
    out["robust_curve_fitting"] = cf

    # tiny BAL with gross outliers in the observations.  Ceres applies one rho(|r|^2) per residual BLOCK
    # (2 residuals), which SciPy's per-residual `loss` cannot express; so each block is fed as the single
    # residual sqrt(rho_huber(s)) — 1/2 sum of its squares IS the robust cost — and minimised plainly.
    C, P, N, seed = 4, 24, 80, 11
    prob = bal.generate(C, P, N, seed=seed)
    rng = np.random.default_rng(5)
    obs = prob.observations.copy()
    bad = rng.choice(N, 6, replace=False)
    obs[bad] += rng.normal(0, 40.0, (6, 2))
    a = 2.0

    def blocks(x):
        cams = x[:9 * C].reshape(C, 9)[prob.camera_index]
        pts = x[9 * C:].reshape(P, 3)[prob.point_index]
        pred, _ = bal.snavely_project(cams, pts)
        return pred - obs

    def fr(x):  # one residual per block: sqrt(rho_huber(s)), so that 1/2 sum fr^2 = 1/2 sum rho(s)
        s2 = np.sum(blocks(x) ** 2, axis=1)
        return np.sqrt(np.where(s2 <= a * a, s2, 2 * a * np.sqrt(s2) - a * a))

    def cost(x):
        return 0.5 * np.sum(fr(x) ** 2)
    s = least_squares(fr, prob.parameters, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=3000)
    # sqrt(rho) is not smooth at s = 0: polish the robust cost itself (C^1) with a quasi-Newton method
    from scipy.optimize import minimize
    xm = s.x
    for _ in range(6):
        xm = minimize(cost, xm, method="L-BFGS-B", options={"maxiter": 20000, "maxfun": 2000000, "ftol": 1e-16, "gtol": 1e-9}).x

    class _S:  # what the record below reads
        fun = fr(xm)
    s = _S
    out["tiny_bal_huber"] = {"shape": [C, P, N], "seed": seed, "a": a, "observations": obs.tolist(),
                             "cost": float(0.5 * np.sum(s.fun ** 2)),
                             "initial_cost": float(0.5 * np.sum(fr(prob.parameters) ** 2))}
    with open(os.path.join(HERE, "robust_optima.json"), "w") as f:
        json.dump({"source": "scipy.optimize.least_squares (robust losses)", **out}, f, indent=1)


def make_lm_step():
    """One LM step as Ceres 1.x defines it (constants: initial radius 1e4, LM diagonal
    clamp [1e-6, 1e32], Jacobi scaling 1 / (1 + |J_j|)) on bal.generate(16, 60, 300, seed=8)."""
    mp.mp.dps = 40
    C, P, N, seed = 16, 60, 300, 8
    prob = bal.generate(C, P, N, seed=seed)
    syms, res, J = snavely_symbolic()
    f_res = sym.lambdify(syms, res, "mpmath")
    f_jac = sym.lambdify(syms, J, "mpmath")
    n = 9 * C + 3 * P

    def evaluate(x, want_jac):
        r = np.zeros(2 * N, dtype=np.longdouble)
        Jd = np.zeros((2 * N, n), dtype=np.longdouble) if want_jac else None
        cost = mp.mpf(0)
        for o in range(N):
            c, q = int(prob.camera_index[o]), int(prob.point_index[o])
            args = [mp.mpf(float(v)) for v in list(x[9 * c:9 * c + 9]) + list(x[9 * C + 3 * q:9 * C + 3 * q + 3]) + list(prob.observations[o])]
            rv = f_res(*args)
            cost += rv[0] * rv[0] + rv[1] * rv[1]
            r[2 * o], r[2 * o + 1] = np.longdouble(str(rv[0])), np.longdouble(str(rv[1]))
            if want_jac:
                Jm = f_jac(*args)
                for i in range(2):
                    for j in range(9):
                        Jd[2 * o + i, 9 * c + j] = np.longdouble(str(Jm[i, j]))
                    for j in range(3):
                        Jd[2 * o + i, 9 * C + 3 * q + j] = np.longdouble(str(Jm[i, 9 + j]))
        return float(cost / 2), r, Jd

    x0 = prob.parameters.astype(np.float64)
    cost0, r, Jd = evaluate(x0, True)
    g = Jd.T @ r
    scale = 1.0 / (1.0 + np.sqrt(np.sum(Jd * Jd, axis=0)))
    Js = Jd * scale
    diag = np.sum(Js * Js, axis=0)
    radius = np.longdouble(1e4)
    D2 = np.minimum(np.maximum(diag, np.longdouble(1e-6)), np.longdouble(1e32)) / radius
    A = Js.T @ Js + np.diag(D2)
    b = Js.T @ r
    A64 = A.astype(np.float64)
    y = np.linalg.solve(A64, b.astype(np.float64)).astype(np.longdouble)
    for _ in range(6):  # iterative refinement against the extended-precision system
        y = y + np.linalg.solve(A64, (b - A @ y).astype(np.float64)).astype(np.longdouble)
    step_scaled = -y
    mr = Js @ step_scaled
    mcc = float(-np.dot(mr, r + mr / 2))
    delta = (step_scaled * scale).astype(np.float64)
    x1 = x0 + delta
    cost1, _, _ = evaluate(x1, False)
    rho = (cost0 - cost1) / mcc
    out = {"source": "sympy closed form + mpmath 40 digits; numpy extended precision + iterative refinement",
           "shape": [C, P, N], "seed": seed, "camera_index": prob.camera_index.tolist(), "point_index": prob.point_index.tolist(),
           "observations": prob.observations.tolist(), "x0": x0.tolist(),
           "initial_cost": cost0, "gradient_max_norm": float(np.max(np.abs(g))), "initial_radius": 1e4,
           "delta": delta.tolist(), "step_norm": float(np.sqrt(np.sum((x1 - x0) ** 2))), "model_cost_change": mcc,
           "candidate_cost": cost1, "relative_decrease": rho,
           "radius_after": float(1e4 / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)),
           "refinement_residual": float(np.max(np.abs(b - A @ y)) / np.max(np.abs(b)))}
    with open(os.path.join(HERE, "lm_step.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lm_step":
        make_lm_step()
        sys.exit(0)
    make_snavely()
    make_optima()
    make_robust_optima()
    make_lm_step()
    print("golden fixtures written to", HERE)
