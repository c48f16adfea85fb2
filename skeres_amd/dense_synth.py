"""Synthetic dense problem of BASELINE.json config 5 ("10k params x 1M residuals"):
r_i(x) = tanh(a_i . x) - y_i, a_ij ~ N(0, 1/n) from the counter-based generator of
skeres_amd/csrc/synth.hpp (restated here in numpy for small sizes), x* ~ N(0, 1),
y = tanh(A x*) + N(0, sigma^2), start x0 = 0."""
import numpy as np

_M1, _M2, _G, _K = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15, 0xD6E8FEB86659FD93
_MASK = (1 << 64) - 1


def _mix64(z):
    z = (z + np.uint64(_G))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
    return z ^ (z >> np.uint64(31))


def unit_rows(seed, rows, n):
    """unit-variance draws unit(seed, i, j) for i in rows, j in [0, n): array [len(rows), n]."""
    with np.errstate(over="ignore"):
        i = np.asarray(rows, dtype=np.uint64)[:, None]
        j = np.arange(n, dtype=np.uint64)[None, :]
        h = _mix64(np.uint64(seed) ^ ((i * np.uint64(n) + j) * np.uint64(_K)))
    s = ((h & np.uint64(0xffff)) + ((h >> np.uint64(16)) & np.uint64(0xffff)) + ((h >> np.uint64(32)) & np.uint64(0xffff))
         + ((h >> np.uint64(48)) & np.uint64(0xffff))).astype(np.float64) + 2.0
    return (s * (1.0 / 65536.0) - 2.0) * 1.7320508075688772


def generate(m, n, seed=5, sigma=1e-3, chunk=4096):
    """(consts [m, 3] = (seed, row, y), x_star).  Host-side: meant for m * n up to ~1e8."""
    rng = np.random.default_rng(seed)
    x_star = rng.normal(size=n)
    y = np.empty(m)
    for r0 in range(0, m, chunk):
        rows = np.arange(r0, min(m, r0 + chunk))
        y[rows] = np.tanh(unit_rows(seed, rows, n) @ x_star / np.sqrt(n))
    y += rng.normal(0, sigma, m)
    consts = np.stack([np.full(m, float(seed)), np.arange(m, dtype=np.float64), y], axis=1)
    return consts, x_star
