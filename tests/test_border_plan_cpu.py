"""Loop closures: which cameras the solver orders into the trailing border of the reduced camera system
(sk_problem_border_plan — the host logic of BalSolver::setup, no device needed; DESIGN.md section 4, "Bordered envelope")."""
import numpy as np
import pytest

import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk


def _plan(prob, mode):
    problem, params, loss = bal_problem_to_sk(prob)
    r = problem.borderPlan(mode)
    pos = r["position"]
    C = prob.num_cameras
    # a permutation of the cameras: equal cameras get equal positions, different cameras different ones
    cam_pos = np.full(C, -1, dtype=np.int64)
    cam_pos[prob.camera_index] = pos
    assert sorted(cam_pos.tolist()) == list(range(C))
    assert np.array_equal(cam_pos[prob.camera_index], pos)
    return r, cam_pos


def test_generator_with_revisits_keeps_the_shape_and_couples_the_two_windows():
    base = bal.generate(400, 30000, 140000, seed=77)
    prob = bal.generate(400, 30000, 140000, seed=77, revisits=[(60, 250, 12, 40), (120, 330, 12, 40)])
    assert (prob.num_cameras, prob.num_points, prob.num_observations) == (400, 30000, 140000)
    # the same track lengths, and tracks that see both windows of a revisit
    assert np.array_equal(np.bincount(prob.point_index, minlength=30000), np.bincount(base.point_index, minlength=30000))
    for a, b in ((60, 250), (120, 330)):
        in_a = np.zeros(30000, bool)
        in_b = np.zeros(30000, bool)
        ca, cb = prob.camera_index, prob.camera_index
        in_a[prob.point_index[(ca >= a) & (ca < a + 12)]] = True
        in_b[prob.point_index[(cb >= b) & (cb < b + 12)]] = True
        assert (in_a & in_b).sum() >= 40
    with pytest.raises(ValueError):
        bal.generate(400, 30000, 140000, seed=77, revisits=[(60, 65, 12, 40)])  # overlapping windows


def test_no_visits_no_border():
    prob = bal.generate(400, 30000, 140000, seed=77)
    for mode in ("auto", "off"):
        r, cam_pos = _plan(prob, mode)
        assert r["border_cameras"] == 0 and r["model_us"] == r["model_us_plain"] > 0
    # the band's own order: the BAL numbering
    assert np.array_equal(cam_pos, np.arange(400))


def test_revisiting_cameras_go_to_the_border():
    revisits = [(60, 350, 12, 40), (200, 520, 12, 40)]
    prob = bal.generate(600, 6000, 26000, seed=9, revisits=revisits)
    r_off, _ = _plan(prob, "off")
    r_on, cam_pos = _plan(prob, "on")
    r_auto, cam_pos_auto = _plan(prob, "auto")
    assert np.array_equal(cam_pos, cam_pos_auto)  # (the model takes it by itself here)
    nb = r_on["border_cameras"]
    assert r_off["border_cameras"] == 0 and 1 <= nb <= 24
    border = np.flatnonzero(cam_pos >= 600 - nb)
    # one window of every revisit, whole: the cameras of window b (or of window a) that the moved tracks see
    in_b = np.concatenate([np.arange(b, b + w) for a, b, w, t in revisits])
    in_a = np.concatenate([np.arange(a, a + w) for a, b, w, t in revisits])
    assert set(border) <= set(in_b) or set(border) <= set(in_a)
    # the band keeps its order
    band = np.flatnonzero(cam_pos < 600 - nb)
    assert np.all(np.diff(cam_pos[band]) > 0)
    # and its width: the envelope with the border is smaller than the plain one, and the model says so
    assert r_on["envelope_fill"] < r_off["envelope_fill"]
    assert r_on["model_us"] < r_on["model_us_plain"]
    assert r_on["gap"] >= 4


def test_auto_takes_the_border_at_ladybug_size_and_leaves_the_plain_problem_alone():
    """BASELINE.json configs[2] shape with three places revisited: the chain model prefers the border (>= 10 %)."""
    prob = bal.generate_named("ladybug-1723-156502", revisits=[(200, 900, 40, 150), (450, 1300, 40, 150), (700, 1600, 40, 150)])
    r, cam_pos = _plan(prob, "auto")
    assert 100 <= r["border_cameras"] <= 124
    assert r["model_us"] < 0.7 * r["model_us_plain"] and r["envelope_fill"] < 0.45
    plain = bal.generate_named("ladybug-1723-156502")
    r0, cam_pos0 = _plan(plain, "auto")
    assert r0["border_cameras"] == 0 and abs(r0["envelope_fill"] - 0.3693) < 1e-3
