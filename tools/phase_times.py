#!/usr/bin/env python3
"""Per-phase milliseconds per LM iteration of a named BAL workload (developer tool).
usage: phase_times.py [ladybug|venice|bal49] [steps] [key=value solver options: border=off dissection=off retained=off|on:12 revisits=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402

NAMES = {"ladybug": ("ladybug-1723-156502", 1723), "venice": ("venice-1778-993923", 1778), "bal49": ("problem-49-7776", 49)}
PHASES = ["jacobian_eval", "schur_assemble", "cholesky", "back_substitute", "cost_eval", "allreduce"]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    kw = dict(a.split("=") for a in sys.argv[3:])
    name, seed = NAMES[which]
    gen = {}
    if kw.pop("revisits", None):
        gen["revisits"] = [(200, 900, 40, 150), (450, 1300, 40, 150), (700, 1600, 40, 150)]
    prob = bal.generate_named(name, seed=seed, perturb=(1e-2, 1e-1, 1e-1), **gen)
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    loss = sk.PredefinedLossFunctions.trivialLoss()
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, loss, params, offs)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(steps + 1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    if "border" in kw:
        o.setCholeskyBorder(kw["border"])
    if "dissection" in kw:
        o.setCholeskyDissection(kw["dissection"])
    if "retained" in kw:
        o.setRetainedPoints(*(kw["retained"].split(":")[:1] + [int(v) for v in kw["retained"].split(":")[1:]]))
    solver = sk.StepSolver(o, problem)
    for _ in range(2):
        solver.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.step()
    summ = sk.Solver.Summary()
    solver.finish(summ)
    dt = time.perf_counter() - t0
    its = summ.iterations()
    n = max(1, len(its) - 1)
    ph = {k: 1e3 * summ.phaseSeconds(i) / n for i, k in enumerate(PHASES)}
    print("  cut: head %d tail %d separator %d" % (solver.stat("dissection_head_cameras"), solver.stat("dissection_tail_cameras"), solver.stat("dissection_separator_cameras")))
    print("  retained points %d (model %.0f us, %.0f without), border cameras %d, dissected %d, envelope fill %.3f, resident columns %d" % (
        solver.stat("retained_points"), solver.stat("retained_model_us"), solver.stat("retained_model_us_without"), solver.stat("border_cameras"),
        solver.stat("dissected"), solver.stat("envelope_fill"), solver.stat("cholesky_columns_resident")))
    print("%s %s: %.3f ms/step | A %.3f  B %.3f  C %.3f  D %.3f + %.3f | final cost %.9e" % (
        which, " ".join(sys.argv[3:]), 1e3 * dt / steps, ph["jacobian_eval"], ph["schur_assemble"], ph["cholesky"], ph["back_substitute"], ph["cost_eval"], its[-1]["cost"]), flush=True)


if __name__ == "__main__":
    main()
