"""Subprocess of tests/test_gpu_parity.py::test_resident_backsolve_is_bitwise_the_launch_by_launch_one: solves a banded and
a dense system through sk_cholesky_solve_ex and runs three LM iterations of a bundle-adjustment problem, and stores every
solution in argv[1] (.npz).  The parent runs it twice — as it is (the back-substitution of the reduced system is ONE resident
launch, bs_resident_kernel) and with SK_BS_RESIDENT=0 (one launch per block step, bs_step_kernel) — and compares bits: the
environment is read once per process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import solve_bal_gpu  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(77)
    # block-banded, 30 block columns, envelope six blocks high (as tests/test_gpu_parity.py::_banded_spd)
    nblk, n = 31, 128 * 31 - 70
    last = np.minimum(np.arange(nblk) + 6, nblk - 2).astype(np.int32)
    last[nblk - 1] = nblk - 1
    A = np.zeros((n, n))
    for c in range(nblk):
        c0, c1, r1 = 128 * c, min(n, 128 * (c + 1)), min(n, 128 * (min(last[c], nblk - 2) + 1))
        if c0 < n:
            A[c0:r1, c0:c1] = rng.normal(0, 1.0, (r1 - c0, c1 - c0))
    A = np.tril(A)
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + np.abs(A).sum(axis=0) + 1.0
    b = rng.normal(size=n)
    out["banded"], _ = sk.api.cholesky_solve(A, b, want_L=True, last=last, group=0, automatic_plan=True)
    # dense
    m = 1500
    U = rng.normal(size=(m, 40))
    D = U @ U.T
    D[np.arange(m), np.arange(m)] += 10.0
    out["dense"], _ = sk.api.cholesky_solve(D, rng.normal(size=m), want_L=True, group=2)
    prob = bal.generate(400, 30000, 140000, seed=77)
    x, summary = solve_bal_gpu(prob, setMaxNumIterations=3)
    out["bal_x"] = x
    out["bal_costs"] = np.array([it["cost"] for it in summary.iterations()])
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
