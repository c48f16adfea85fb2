"""Bundle-adjustment-in-the-large (BAL) problem container, text reader/writer and
a seeded synthetic generator with the exact (cameras, points, observations)
shape of the named datasets.

Mirrors ``BalProblem`` of the reference
(examples/src/main/scala/org/somelightprojections/skeres/examples/SimpleBundleAdjuster.scala:18-77):
one contiguous parameter array, cameras ``[0, 9C)`` then points ``[9C, 9C+3P)``;
camera block = ``[angle_axis(3), translation(3), focal, k1, k2]``.

Real BAL files are not available offline, so every BAL configuration in this
repository is synthetic and shape-exact.
"""
from dataclasses import dataclass

import numpy as np

# (cameras, points, observations) of the datasets BASELINE.json names.
SHAPES = {
    "problem-49-7776": (49, 7776, 31843),
    "ladybug-1723-156502": (1723, 156502, 678718),
    "venice-1778-993923": (1778, 993923, 5001946),
}


@dataclass
class BalProblem:
    num_cameras: int
    num_points: int
    camera_index: np.ndarray  # int32 [N]
    point_index: np.ndarray   # int32 [N]
    observations: np.ndarray  # float64 [N, 2]
    parameters: np.ndarray    # float64 [9C + 3P]

    @property
    def num_observations(self):
        return int(self.camera_index.shape[0])

    @property
    def num_parameters(self):
        return 9 * self.num_cameras + 3 * self.num_points

    def cameras(self):
        return self.parameters[: 9 * self.num_cameras].reshape(-1, 9)

    def points(self):
        return self.parameters[9 * self.num_cameras:].reshape(-1, 3)

    # -- BAL text format (SimpleBundleAdjuster.scala:37-76) ------------------
    def to_file(self, path):
        with open(path, "w") as f:
            f.write("%d %d %d\n" % (self.num_cameras, self.num_points, self.num_observations))
            for c, p, (x, y) in zip(self.camera_index, self.point_index, self.observations):
                f.write("%d %d %.16e %.16e\n" % (c, p, x, y))
            for v in self.parameters:
                f.write("%.16e\n" % v)

    @staticmethod
    def from_file(path):
        with open(path) as f:
            tok = f.read().split()
        C, P, N = int(tok[0]), int(tok[1]), int(tok[2])
        body = np.array(tok[3:3 + 4 * N], dtype=np.float64).reshape(N, 4)
        params = np.array(tok[3 + 4 * N: 3 + 4 * N + 9 * C + 3 * P], dtype=np.float64)
        if params.shape[0] != 9 * C + 3 * P:
            raise ValueError("truncated BAL file: expected %d parameters" % (9 * C + 3 * P))
        return BalProblem(C, P, body[:, 0].astype(np.int32), body[:, 1].astype(np.int32),
                          np.ascontiguousarray(body[:, 2:4]), params)


def snavely_project(cameras, points):
    """Vectorised SnavelyReprojectionError prediction (numpy; generator only)."""
    aa, t = cameras[:, 0:3], cameras[:, 3:6]
    f, k1, k2 = cameras[:, 6], cameras[:, 7], cameras[:, 8]
    theta2 = np.sum(aa * aa, axis=1)
    theta = np.sqrt(np.maximum(theta2, 1e-300))
    w = aa / theta[:, None]
    c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
    wxp = np.cross(w, points)
    p_rot = points * c + wxp * s + w * (np.sum(w * points, axis=1)[:, None] * (1 - c))
    small = theta2 <= np.finfo(np.float64).eps
    if small.any():
        p_rot[small] = points[small] + np.cross(aa[small], points[small])
    p = p_rot + t
    xp, yp = -p[:, 0] / p[:, 2], -p[:, 1] / p[:, 2]
    r2 = xp * xp + yp * yp
    d = 1.0 + r2 * (k1 + k2 * r2)
    return np.stack([f * d * xp, f * d * yp], axis=1), p[:, 2]


def generate(num_cameras, num_points, num_observations, seed=1723, sigma_px=0.5,
             perturb=(1e-2, 1e-1, 1e-1), long_range_fraction=0.0, revisits=()):
    """Seeded synthetic BAL problem with exactly the requested shape.

    Cameras sit on a noisy trajectory along +x looking down -z at a slab of
    points (Snavely convention: points in front have p_z < 0).  Each point is
    seen by k_p >= 2 cameras, k_p drawn from a truncated log-normal with a
    heavy tail (up to C/4) and adjusted so that sum k_p == N; its cameras are
    a spatially local, strictly increasing run around the point's position on
    the trajectory, so the reduced camera matrix has BAL-like band + fill.

    ``long_range_fraction`` (loop closures): that fraction of the points is seen from TWO distant windows of the
    trajectory instead of one — the second half of the point's camera run is moved to a window drawn uniformly from
    the part of the trajectory the first half does not cover (a street revisited).  Every such track couples two
    far-apart groups of cameras, which is what widens the envelope of the reduced camera system; the sensitivity of
    the factorisation to it is reported in profiles/ (tools/envelope_sensitivity.py).  0 leaves the problem exactly
    as it was (the extra draws come from a generator of their own).

    ``revisits`` (LOCALISED loop closures): ``[(first_a, first_b, width, tracks), ...]`` — the trajectory comes back, with
    the ``width`` cameras from ``first_b`` on, to the place the ``width`` cameras from ``first_a`` on have seen: ``tracks``
    of the short tracks that lie inside window a keep the first half of their cameras there and have the second half
    moved to the same offsets inside window b (a street driven twice; real sequences revisit a few places, they do not
    scatter tracks over the whole trajectory as ``long_range_fraction`` does).  Shapes (C, P, N and every track length)
    stay exactly what they were.
    """
    C, P, N = int(num_cameras), int(num_points), int(num_observations)
    if N < 2 * P:
        raise ValueError("need at least 2 observations per point")
    if N > P * C:
        raise ValueError("more observations than (camera, point) pairs")
    rng = np.random.default_rng(seed)
    kmax = max(2, min(C, max(4, C // 4, int(np.ceil(N / P)) + 2)))
    mean_extra = N / P - 2.0
    sig = 1.2
    mu = np.log(max(mean_extra, 1e-3)) - 0.5 * sig * sig
    k = 2 + np.floor(rng.lognormal(mu, sig, size=P)).astype(np.int64)
    k = np.clip(k, 2, kmax)
    # adjust to hit N exactly
    for _ in range(64):
        diff = N - int(k.sum())
        if diff == 0:
            break
        if diff > 0:
            cand = np.flatnonzero(k < kmax)
            pick = rng.choice(cand, size=min(diff, cand.size), replace=False)
            k[pick] += 1
        else:
            cand = np.flatnonzero(k > 2)
            pick = rng.choice(cand, size=min(-diff, cand.size), replace=False)
            k[pick] -= 1
    if int(k.sum()) != N:
        raise RuntimeError("could not match observation count")

    # ground truth cameras
    spacing = 0.02  # long tracks (up to ~C/2 cameras wide) must stay inside the field of view
    cam_x = spacing * np.arange(C) + rng.normal(0, 0.02, C)
    centers = np.stack([cam_x, rng.normal(0, 0.3, C), 6.0 + rng.normal(0, 0.3, C)], axis=1)
    aa = rng.normal(0, 0.08, (C, 3))
    cams = np.zeros((C, 9))
    cams[:, 0:3] = aa
    cams[:, 6] = rng.uniform(400, 1200, C)
    cams[:, 7] = rng.normal(0, 3e-7, C)
    cams[:, 8] = rng.normal(0, 6e-13, C)
    # t = -R c
    Rc, _ = _rotate(aa, centers)
    cams[:, 3:6] = -Rc

    # visibility: strictly increasing local camera runs
    pt_start = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(k, out=pt_start[1:])
    gaps = rng.integers(1, 4, size=N).astype(np.int64)
    gaps[pt_start[:-1]] = 0
    run = np.cumsum(gaps)
    run -= np.repeat(run[pt_start[:-1]], k)          # offsets within the point's run
    span = run[pt_start[1:] - 1]                      # last offset per point
    span = np.minimum(span, C - 1)
    anchor = rng.uniform(0, 1, P)
    base = np.floor(anchor * (C - span)).astype(np.int64)   # base + span <= C - 1
    cam_idx = np.repeat(base, k) + run
    # runs longer than C-1 (only when k ~ C): wrap into range keeping distinctness
    over = cam_idx > C - 1
    if over.any():
        # rebuild those points as evenly spread distinct cameras
        bad_pts = np.unique(np.repeat(np.arange(P), k)[over])
        for p in bad_pts:
            kp = int(k[p])
            cam_idx[pt_start[p]:pt_start[p + 1]] = np.sort(rng.choice(C, size=kp, replace=False))
    if long_range_fraction > 0.0:
        rng_lr = np.random.default_rng([seed, 0x10c5])
        chosen = np.flatnonzero(rng_lr.uniform(0, 1, P) < long_range_fraction)
        u = rng_lr.uniform(0, 1, P)
        for p in chosen:
            a, b = int(pt_start[p]), int(pt_start[p + 1])
            h = a + (b - a) // 2                     # second half: observations [h, b)
            first = cam_idx[a:h]
            off = cam_idx[h:b] - cam_idx[h]          # offsets inside the second window
            span2 = int(off[-1])
            lo, hi = int(first[0]), int(first[-1])
            len_l = max(0, lo - span2)               # window start positions left of the first half: [0, lo - span2)
            len_r = max(0, (C - span2) - (hi + 1))   # ... and right of it: [hi + 1, C - span2)
            if len_l + len_r <= 0:
                continue                             # the first half leaves no room: the track stays local
            t = int(u[p] * (len_l + len_r))
            base2 = t if t < len_l else hi + 1 + (t - len_l)
            cam_idx[h:b] = base2 + off
    for ri, (first_a, first_b, width, tracks) in enumerate(revisits):
        first_a, first_b, width, tracks = int(first_a), int(first_b), int(width), int(tracks)
        if not (0 <= first_a and first_a + width <= C and 0 <= first_b and first_b + width <= C and width >= 2):
            raise ValueError("revisit %d: windows must lie inside the trajectory" % ri)
        if abs(first_a - first_b) < width:
            raise ValueError("revisit %d: the two windows overlap" % ri)
        rng_rv = np.random.default_rng([seed, 0x7e15, ri])
        lo = cam_idx[pt_start[:-1]]
        hi = cam_idx[pt_start[1:] - 1]
        inside = np.flatnonzero((lo >= first_a) & (hi < first_a + width) & (k >= 2))
        if inside.size < tracks:
            raise ValueError("revisit %d: only %d tracks lie inside window a" % (ri, inside.size))
        for p in np.sort(rng_rv.choice(inside, size=tracks, replace=False)):
            a, b = int(pt_start[p]), int(pt_start[p + 1])
            h = a + (b - a + 1) // 2                 # second half: observations [h, b)
            cam_idx[h:b] = cam_idx[h:b] - first_a + first_b
    pt_idx = np.repeat(np.arange(P, dtype=np.int64), k)

    # points near the centroid of their cameras' x positions
    cx_sum = np.add.reduceat(cam_x[cam_idx], pt_start[:-1])
    px = cx_sum / k + rng.normal(0, 0.5, P)
    pts = np.stack([px, rng.normal(0, 1.0, P), np.clip(rng.normal(0, 1.0, P), -3, 3)], axis=1)

    proj, depth = snavely_project(cams[cam_idx], pts[pt_idx])
    assert (depth < 0).all(), "generator produced a point behind its camera"
    obs = proj + rng.normal(0, sigma_px, (N, 2))

    # initial parameters = truth perturbed
    cams0 = cams.copy()
    cams0[:, 0:3] += rng.uniform(-perturb[0], perturb[0], (C, 3))
    cams0[:, 3:6] += rng.uniform(-perturb[1], perturb[1], (C, 3))
    pts0 = pts + rng.uniform(-perturb[2], perturb[2], (P, 3))

    # BAL files list observations point-major in practice; shuffle lightly so
    # nothing downstream can rely on input order.
    order = rng.permutation(N)
    return BalProblem(C, P, cam_idx[order].astype(np.int32), pt_idx[order].astype(np.int32),
                      np.ascontiguousarray(obs[order]),
                      np.concatenate([cams0.ravel(), pts0.ravel()]))


def generate_named(name, seed=1723, **kw):
    C, P, N = SHAPES[name]
    return generate(C, P, N, seed=seed, **kw)


def _rotate(aa, pts):
    theta = np.sqrt(np.sum(aa * aa, axis=1))
    w = aa / np.maximum(theta, 1e-300)[:, None]
    c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
    out = pts * c + np.cross(w, pts) * s + w * (np.sum(w * pts, axis=1)[:, None] * (1 - c))
    return out, theta
