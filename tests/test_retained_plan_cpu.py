"""Retained points: which points DENSE_SCHUR keeps in the reduced system instead of eliminating them (sk_problem_retained_plan — the
host logic of BalSolver::setup, no device needed; DESIGN.md section 4, "Retained points")."""
import numpy as np

import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk


def _spans(prob, problem):
    """span of every point's cameras in the order the cameras have inside the reduced system (no border)"""
    pos = problem.borderPlan("off")["position"].astype(np.int64)
    P, C = prob.num_points, prob.num_cameras
    lo = np.full(P, C)
    hi = np.full(P, -1)
    np.minimum.at(lo, prob.point_index, pos)
    np.maximum.at(hi, prob.point_index, pos)
    return hi - lo


def test_a_short_band_retains_nothing_and_a_forced_count_is_exact():
    prob = bal.generate(400, 30000, 140000, seed=77)
    problem, params, loss = bal_problem_to_sk(prob)
    auto = problem.retainedPlan("auto")
    assert auto["retained_points"] == 0 and not auto["retained_of_block"].any() and auto["model_us"] == auto["model_us_without"] > 0
    assert problem.retainedPlan("off")["retained_points"] == 0
    on = problem.retainedPlan("on", 9)
    assert on["retained_points"] == 9
    kept = np.unique(prob.point_index[on["retained_of_block"] == 1])
    assert len(kept) == 9
    # a retained point is retained in every residual block that has it
    assert np.array_equal(on["retained_of_block"] == 1, np.isin(prob.point_index, kept))
    # the widest tracks, in the cameras' order inside the reduced system
    span = _spans(prob, problem)
    assert span[kept].min() >= np.delete(span, kept).max()
    # a count that is not a multiple of three: the multiple below it (three points share a nine-row pseudo-camera)
    assert problem.retainedPlan("on", 8)["retained_points"] == 6


def test_ladybug_shaped_sequence_retains_its_landmarks():
    """Ladybug-1723: the twelve points with the longest tracks (98 .. 392 observations) set the envelope's height; without them
    every block column of the reduced system is chain-bound and the chain model falls by more than a third."""
    prob = bal.generate_named("ladybug-1723-156502", seed=1723)
    problem, params, loss = bal_problem_to_sk(prob)
    r = problem.retainedPlan("auto")
    assert 3 <= r["retained_points"] <= 48 and r["retained_points"] % 3 == 0
    assert r["model_us"] < 0.7 * r["model_us_without"]
    kept = np.unique(prob.point_index[r["retained_of_block"] == 1])
    k = np.bincount(prob.point_index, minlength=prob.num_points)
    assert k[kept].min() >= 50 and k[kept].max() == k.max()
    # with loop closures on top (three places seen twice): landmarks are retained AND the revisiting cameras go to the border
    rev = bal.generate_named("ladybug-1723-156502", seed=1723, revisits=[(200, 900, 40, 150), (450, 1300, 40, 150), (700, 1600, 40, 150)])
    problem2, _, _ = bal_problem_to_sk(rev)
    r2 = problem2.retainedPlan("auto")
    assert r2["retained_points"] >= 3 and r2["model_us"] < 0.7 * r2["model_us_without"]
    # ... and with the border of cameras forbidden, the retained points take the revisits' tracks too (a border of points instead of cameras)
    r3 = problem2.retainedPlan("auto", 0, "off")
    assert r3["retained_points"] > r2["retained_points"] and r3["model_us"] < 0.9 * r3["model_us_without"]


def test_scattered_loop_closures_are_retained_as_points():
    """Ladybug-1723 with 0.5 % of its tracks seen from two distant windows (bench.py's record `loop_closures`): the doubling counts of
    the widest tracks cannot find the set (782 tracks: 768 leave the envelope full, 1536 cost twice the rows); the tracks with a jump
    in their camera list, retained at their exact number, bring the chain model from 25.8 ms (a border of 237 cameras) to a third."""
    prob = bal.generate_named("ladybug-1723-156502", seed=1723, long_range_fraction=0.005)
    problem, params, loss = bal_problem_to_sk(prob)
    r = problem.retainedPlan("auto")
    assert 700 <= r["retained_points"] <= 900 and r["retained_points"] % 3 == 0
    assert r["model_us"] < 0.4 * r["model_us_without"]
