#!/usr/bin/env python3
"""Developer tool: timeline of the resident panel chain (SK_CHAIN_STAMPS) on the Ladybug-1723-shaped solve.
Per block column: when the server started / finished potrf, and when tile 0 of the column launch started, saw potrf,
finished the TRSM, saw X(j+1,j) and syrk(j-1), finished next(j) — microseconds relative to the server's start."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402


def main():
    path = os.path.join(tempfile.gettempdir(), "sk_chain_stamps.txt")
    for f in (path, path + ".pair"):
        if os.path.exists(f):
            os.remove(f)
    os.environ["SK_CHAIN_STAMPS"] = path
    which = sys.argv[1] if len(sys.argv) > 1 else "ladybug-1723-156502"
    if which.startswith("band:"):  # a synthetic camera band of another size: band:C,P,N,seed
        C_, P_, N_, seed_ = (int(v) for v in which[5:].split(","))
        prob = bal.generate(C_, P_, N_, seed=seed_)
    else:
        prob = bal.generate_named(which, seed=1723, perturb=(1e-2, 1e-1, 1e-1))
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(1, prob.observations, None, params, offs)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    o.setCholeskyDissection(sys.argv[2] if len(sys.argv) > 2 else "off")  # (the stamps are per block column of ONE front)
    s = sk.StepSolver(o, problem)
    for _ in range(4):
        s.step()
    files = [f for f in (path + ".pair", path) if os.path.exists(f)]  # (a lock-step dissection: the two leaf fronts, then the root)
    raws = [np.loadtxt(f) for f in files]
    origin = min(r[:, 1].min() for r in raws)
    for f, r in zip(files, raws):
        report(r)
        # (one clock for all of them: where a factorisation's chain begins and ends against the other's)
        print("on the common clock: first potrf start %+.1f us, last potrf done %+.1f us, last tile done %+.1f us\n"
              % ((r[:, 1].min() - origin) * 0.01, (r[:, 2].max() - origin) * 0.01, (r[:, 7].max() - origin) * 0.01))
        os.remove(f)


def report(raw):
    t0 = raw[:, 1].min()
    names = ["potrf start", "potrf done", "col start", "saw potrf", "trsm done", "saw X+syrk", "next done"]
    # (two fronts in lock-step: the partner front's block columns are numbered from 512)
    for front, rows in (("head / only front", raw[raw[:, 0] < 512]), ("partner front (lock-step)", raw[raw[:, 0] >= 512])):
        if not len(rows):
            continue
        cols, st = rows[:, 0].astype(int), rows[:, 1:]
        us = (st - t0) * 0.01
        print("%s\ncol  " % front + "  ".join("%11s" % n for n in names) + "   cycle  trailing rows")
        for j in range(len(us)):
            cyc = us[j, 0] - us[j - 1, 0] if j else 0.0
            print("%3d  " % (cols[j] % 512) + "  ".join("%11.1f" % us[j, i] for i in range(7)) + "   %6.1f  %4d" % (cyc, int(st[j, 7])))
        d = np.diff(us[:, 0])
        seg = np.stack([us[:, 1] - us[:, 0], us[:, 3] - us[:, 1], us[:, 4] - us[:, 3], us[:, 5] - us[:, 4], us[:, 6] - us[:, 5]], axis=1)
        nxt = us[1:, 0] - us[:-1, 6]
        print("total %.1f us, mean cycle %.1f us (median %.1f); medians: potrf %.1f, done->seen %.1f, trsm tile %.1f, X seen %.1f, next tile %.1f, next potrf start %.1f"
              % (us[-1, 1] - us[0, 0], d.mean(), np.median(d), *np.median(seg, axis=0), np.median(nxt)))


if __name__ == "__main__":
    main()
