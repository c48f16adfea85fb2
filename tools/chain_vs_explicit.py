#!/usr/bin/env python3
"""Developer tool: the factorisation plans of sk_solve on a block-banded SPD matrix, against each other and against the
residual of A x = b.  Factors the same matrix (a) launch by launch with an explicit SYRK depth and (b) under the automatic
plan (resident panel chain + launch-by-launch groups), prints where the two factors differ block by block.

  python tools/chain_vs_explicit.py --workload ladybug-1723-156502      # the block envelope of the bench problem
  python tools/chain_vs_explicit.py --nblk 40 --band 6                   # a synthetic band
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402


def ladybug_envelope(workload):
    p = bal.generate_named(workload, seed=1723)
    C = p.num_cameras
    n = 9 * C
    nblk = (n + 1 + 127) // 128
    cam = p.camera_index.astype(np.int64)
    cmin = np.full(p.num_points, C)
    np.minimum.at(cmin, p.point_index, cam)
    first_col = np.arange(nblk)
    col = (9 * cmin[p.point_index]) // 128
    for row in ((9 * cam) // 128, (9 * cam + 8) // 128):
        np.minimum.at(first_col, row, col)
    return n, envelope_last(first_col)


def envelope_last(first_col):
    nblk = len(first_col)
    last = np.arange(nblk)
    last[nblk - 1] = nblk - 1
    for i in range(nblk - 1):
        c = min(first_col[i], i)
        last[c] = max(last[c], i)
    last = np.maximum.accumulate(last)
    last[nblk - 2] = min(last[nblk - 2], nblk - 2)
    return last.astype(np.int32)


def banded_spd(n, last, seed):
    """Diagonally dominant symmetric matrix, non-zero exactly inside the block envelope (lower triangle filled)."""
    rng = np.random.default_rng(seed)
    nblk = len(last)
    A = np.zeros((n, n))
    for c in range(nblk):
        c0, c1 = 128 * c, min(n, 128 * (c + 1))
        r1 = min(n, 128 * (min(last[c], nblk - 2) + 1))
        if c0 >= n:
            break
        A[c0:r1, c0:c1] = rng.normal(0, 1.0, (r1 - c0, c1 - c0))
    A = np.tril(A)
    rowsum = np.abs(A).sum(axis=1) + np.abs(A).sum(axis=0)
    A[np.arange(n), np.arange(n)] = rowsum + 1.0 + rng.uniform(0, 1, n)
    return A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default=None)
    ap.add_argument("--nblk", type=int, default=40)
    ap.add_argument("--band", type=int, default=6)
    ap.add_argument("--group", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    if args.workload:
        n, last = ladybug_envelope(args.workload)
    else:
        n = 128 * args.nblk - 1
        first_col = np.maximum(0, np.arange(args.nblk) - args.band)
        last = envelope_last(first_col)
    nblk = len(last)
    print("n = %d, %d block columns, envelope heights %s" % (n, nblk, list((last - np.arange(nblk))[:nblk])))
    A = banded_spd(n, last, args.seed)
    b = np.random.default_rng(args.seed + 1).normal(size=n)
    Af = A + np.tril(A, -1).T if n <= 6000 else None
    x_e, L_e = sk.api.cholesky_solve(A, b, want_L=True, group=args.group, last=last, automatic_plan=False)
    print("explicit group %d: max |L| %.3e" % (args.group, np.abs(L_e).max()))
    for rep in range(args.reps):
        x_a, L_a = sk.api.cholesky_solve(A, b, want_L=True, group=0, last=last, automatic_plan=True)
        d = np.abs(L_a - L_e)
        print("automatic plan (run %d): max |L_auto - L_explicit| = %.3e, |x_auto - x_explicit| / |x| = %.3e" % (
            rep, d.max(), np.linalg.norm(x_a - x_e) / np.linalg.norm(x_e)))
        bad = []
        for i in range(nblk):
            for j in range(i + 1):
                blk = d[128 * i:128 * (i + 1), 128 * j:128 * (j + 1)]
                if blk.size and blk.max() > 1e-9 * max(1.0, np.abs(L_e[128 * i:128 * (i + 1), 128 * j:128 * (j + 1)]).max()):
                    bad.append((i, j, float(blk.max())))
        print("  blocks (row, col, max diff) that differ: %d%s" % (len(bad), "" if not bad else "; first: %s" % bad[:12]))
        if bad:
            cols = sorted({j for _, j, _ in bad})
            print("  block columns with differences: %s" % cols[:40])
    if Af is not None:
        for name, x in (("explicit", x_e), ("automatic", x_a)):
            print("  %s: |A x - b| / |b| = %.3e" % (name, np.linalg.norm(Af @ x - b) / np.linalg.norm(b)))
        Lnp = np.linalg.cholesky(Af)
        print("  vs numpy: explicit %.3e, automatic %.3e" % (np.abs(L_e - Lnp).max(), np.abs(L_a - Lnp).max()))


if __name__ == "__main__":
    main()
