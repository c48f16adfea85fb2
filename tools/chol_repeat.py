#!/usr/bin/env python3
"""Developer tool: several cholesky_solve calls in one process (context re-creation), with progress prints."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skeres_amd as sk  # noqa: E402

for n, group in [(5, 4), (127, 4), (128, 4), (300, 1), (700, 2), (1500, 3)]:
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 20))
    A = G @ G.T + n * np.eye(n)
    b = rng.normal(size=n)
    print("solve", n, group, flush=True)
    x = sk.api.cholesky_solve(A, b, group=group)
    print("  err %.2e" % np.abs(x - np.linalg.solve(A, b)).max(), flush=True)
