#!/usr/bin/env python3
"""Developer tool: retained points, the border of loop-closure cameras and the lock-step dissection in combination, over a spread of
problem shapes — three LM iterations of every plan against the all-eliminated, undissected solve of the same problem (1e-8 on the
costs: elimination orders differ in rounding, nothing else).  Prints one line per case; exit code 1 on a mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402


def run(prob, **kw):
    problem, params, loss = bal_problem_to_sk(prob)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(3)
    o.setGraphReplay(False)
    for k, v in kw.items():
        getattr(o, k)(*v) if isinstance(v, tuple) else getattr(o, k)(v)
    solver = sk.StepSolver(o, problem)
    st = {k: int(solver.stat(k)) for k in ("retained_points", "border_cameras", "dissected", "cholesky_columns_resident")}
    while not solver.step():
        pass
    s = sk.Solver.Summary()
    solver.finish(s)
    return [it["cost"] for it in s.iterations()], params.toArray(prob.num_parameters), st


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    bad = 0
    shapes = [(340, 3400, 15000), (450, 5000, 21000), (520, 5200, 23000), (640, 6400, 28000), (800, 8000, 34000), (1000, 12000, 50000), (1200, 14000, 60000)]
    for C, P, N in shapes:
        seed = int(rng.integers(1, 1000))
        for revisits in ((), "two"):
            rv = [(C // 10, C // 2, C // 50 + 6, 30), (C // 4, (3 * C) // 4, C // 50 + 6, 30)] if revisits else ()
            prob = bal.generate(C, P, N, seed=seed, revisits=rv)
            if C in (520, 1000):
                # the cameras in a scrambled order (and the residual blocks shuffled): neither first appearance nor the memory order of the
                # camera blocks is banded — the reverse Cuthill-McKee candidate has to find the sequence
                perm = rng.permutation(C)
                cams = prob.parameters[:9 * C].reshape(C, 9)
                x0 = np.concatenate([cams[np.argsort(perm)].ravel(), prob.parameters[9 * C:]])
                order = rng.permutation(prob.num_observations)
                prob = bal.BalProblem(C, P, perm[prob.camera_index][order].astype(np.int32), prob.point_index[order].astype(np.int32), prob.observations[order], x0)
            ref, x_ref, st_ref = run(prob, setRetainedPoints="off", setCholeskyDissection="off", setCholeskyBorder="off")
            for name, kw in (("auto", {}), ("retained on:6", {"setRetainedPoints": ("on", 6)}), ("retained on:24", {"setRetainedPoints": ("on", 24)}),
                             ("retained on:24, undissected", {"setRetainedPoints": ("on", 24), "setCholeskyDissection": "off"}),
                             ("retained on:12, border on", {"setRetainedPoints": ("on", 12), "setCholeskyBorder": "on"})):
                if "border on" in name and not revisits:
                    continue
                c, x, st = run(prob, **kw)
                err = max(abs(a / b - 1.0) for a, b in zip(c, ref)) if len(c) == len(ref) else 1.0
                dx = float(np.abs(x - x_ref).max() / max(1.0, np.abs(x_ref).max()))
                ok = err <= 1e-8 and dx <= 1e-6
                bad += 0 if ok else 1
                print("%s C=%d seed=%d revisits=%d %-30s retained %3d border %3d dissected %d resident %3d | cost err %.1e  dx %.1e" % (
                    "ok  " if ok else "FAIL", C, seed, len(rv), name, st["retained_points"], st["border_cameras"], st["dissected"], st["cholesky_columns_resident"], err, dx), flush=True)
    print("FUZZ_RETAINED_%s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
