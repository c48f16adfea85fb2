"""Subprocess of tests/test_gpu_parity.py::test_resident_pairs_plan_vs_numpy_and_oracle: the factorisation plan with RESIDENT
PAIRS (SK_CHAIN_PAIR_MAX_TRAILING=56 in the environment — read once per process; off by default because it does not pay,
chol_kernels.hip) against numpy on envelopes that put pairs between resident runs, after single resident columns, before
them and at an odd run's end, and against the oracle's trajectory on a 400-camera problem."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import solve_bal_gpu  # noqa: E402
import oracle  # noqa: E402

SHAPES = {
    "resident-then-wide-then-resident": [7] * 8 + [40 - i for i in range(30)] + [9] * 24,
    "odd-resident-run": [5] * 7 + [36 - i for i in range(27)] + [8] * 13,
    "two-wide-parts": [6] * 6 + [30 - i for i in range(20)] + [7] * 9 + [28 - i for i in range(19)] + [6] * 10,
    "wide-from-the-start": [44 - i // 2 for i in range(40)] + [8] * 12,
}


def main():
    assert os.environ.get("SK_CHAIN_PAIR_MAX_TRAILING") == "56"
    for name, heights in SHAPES.items():
        nblk = len(heights) + 1
        first_col = np.arange(nblk)
        for c, h in enumerate(heights):
            for r in range(c, min(nblk - 1, c + h + 1)):
                first_col[r] = min(first_col[r], c)
        last = np.arange(nblk)
        for i in range(nblk - 1):
            c = min(first_col[i], i)
            last[c] = max(last[c], i)
        last = np.maximum.accumulate(last)
        last[nblk - 2] = min(last[nblk - 2], nblk - 2)
        last[nblk - 1] = nblk - 1
        last = last.astype(np.int32)
        n = 128 * nblk - 70
        rng = np.random.default_rng(len(name))
        A = np.zeros((n, n))
        for c in range(nblk):
            c0, c1, r1 = 128 * c, min(n, 128 * (c + 1)), min(n, 128 * (min(last[c], nblk - 2) + 1))
            if c0 < n:
                A[c0:r1, c0:c1] = rng.normal(0, 1.0, (r1 - c0, c1 - c0))
        A = np.tril(A)
        A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + np.abs(A).sum(axis=0) + 1.0 + rng.uniform(0, 1, n)
        b = rng.normal(size=n)
        Af = A + np.tril(A, -1).T
        Lnp = np.linalg.cholesky(Af)
        xnp = np.linalg.solve(Af, b)
        x, L = sk.api.cholesky_solve(A, b, want_L=True, last=last, group=0, automatic_plan=True)
        assert np.abs(L - Lnp).max() <= 1e-11 * np.abs(Lnp).max(), (name, np.abs(L - Lnp).max())
        assert np.linalg.norm(x - xnp) <= 1e-11 * np.linalg.norm(xnp), name
    prob = bal.generate(400, 30000, 140000, seed=77)
    x_gpu, sg = solve_bal_gpu(prob, setMaxNumIterations=3)
    x_cpu, so = oracle.solve_bal(400, 30000, prob.camera_index, prob.point_index, prob.observations, prob.parameters,
                                 oracle.default_options(linear_solver_type=oracle.DENSE_SCHUR, num_threads=8, max_num_iterations=3))
    for k, it in enumerate(sg.iterations()):
        assert abs(it["cost"] - so.iterations[k].cost) <= 1e-10 * so.iterations[k].cost, (k, it["cost"], so.iterations[k].cost)
    print("PAIR_PLAN_OK")


if __name__ == "__main__":
    main()
