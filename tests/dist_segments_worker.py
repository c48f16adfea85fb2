"""Worker of tests/test_distributed_cpu.py::test_segmented_plan_and_substructuring_on_the_cpu: one of WORLD_SIZE gloo ranks on
the CPU.  The SEGMENTED distribution's plan comes from the PRODUCT's host logic (sk_problem_segment_plan: which camera is in
which segment or separator, which rank owns which point); the arithmetic it implies — every rank eliminates its own segment
from the reduced camera system of its own points, the separators' system is summed over the ranks, solved, and every rank
back-substitutes its interior — is carried out here with the CPU oracle's reduced systems and numpy, and must give the
solution of the whole reduced system.  No GPU involved: this pins the decomposition, not the kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = bal.generate(160, 5000, 11500, seed=7)   # short tracks: separators of a few cameras, room for eight segments
    C, P = prob.num_cameras, prob.num_points
    problem, _, _ = bal_problem_to_sk(prob)
    R, part, owner = problem.segmentPlan(world, forced=True)
    assert 2 <= R <= world, R
    if len(sys.argv) > 1:
        assert R == int(sys.argv[1]), (R, sys.argv[1])
    cam, pt = prob.camera_index, prob.point_index
    # ---- the plan: a camera has ONE part; no point is seen from two segments; a point's blocks have one owner, its segment if it has one
    part_of_cam = np.full(C, 10 ** 6)
    np.minimum.at(part_of_cam, cam, part)
    assert np.array_equal(part_of_cam[cam], part)
    seg_lo, seg_hi = np.full(P, 10 ** 6), np.full(P, -1)
    in_seg = part >= 0
    np.minimum.at(seg_lo, pt[in_seg], part[in_seg])
    np.maximum.at(seg_hi, pt[in_seg], part[in_seg])
    touched = seg_hi >= 0
    assert np.array_equal(seg_lo[touched], seg_hi[touched])
    own_lo, own_hi = np.full(P, 10 ** 6), np.full(P, -1)
    np.minimum.at(own_lo, pt, owner)
    np.maximum.at(own_hi, pt, owner)
    assert np.array_equal(own_lo, own_hi) and own_hi.max() < R
    assert np.array_equal(own_lo[touched], seg_lo[touched])
    assert set(np.unique(part)) == set(range(-(R - 1), R))  # every segment and every separator has cameras

    # ---- the arithmetic: segment `rank` eliminated on this rank, the separators' system summed over the ranks
    rng = np.random.default_rng(0)
    D = rng.uniform(0.5, 2.0, 9 * C + 3 * P)
    x = prob.parameters
    n = 9 * C
    idx = lambda cams: (9 * np.asarray(cams)[:, None] + np.arange(9)[None, :]).ravel()  # noqa: E731
    B = idx(np.flatnonzero(part_of_cam < 0))
    mine = owner == rank
    if rank < R:
        S, rhs = oracle.bal_reduced_system(C, P, cam[mine], pt[mine], prob.observations[mine], x, D, add_Dc=False)
        S = np.tril(S) + np.tril(S, -1).T
        I = idx(np.flatnonzero(part_of_cam == rank))
        other = np.setdiff1d(np.arange(n), np.concatenate([I, B]))
        assert not S[other].any() and not S[:, other].any()  # this rank's points see its segment and the separators only
        S_II = S[np.ix_(I, I)] + np.diag(D[I] ** 2)
        X = np.linalg.solve(S_II, np.column_stack([S[np.ix_(I, B)], rhs[I]]))
        root = S[np.ix_(B, B)] - S[np.ix_(B, I)] @ X[:, :-1]
        g = rhs[B] - S[np.ix_(B, I)] @ X[:, -1]
    else:  # a replica adds zeros
        root, g = np.zeros((B.size, B.size)), np.zeros(B.size)
    payload = torch.from_numpy(np.concatenate([root.ravel(), g]))
    dist.all_reduce(payload)  # the one collective of the linear solve: the separators' system (a few MB at full size)
    root_sum = payload[: B.size * B.size].numpy().reshape(B.size, B.size) + np.diag(D[B] ** 2)
    y_B = np.linalg.solve(root_sum, payload[B.size * B.size:].numpy())
    y = np.zeros(n)
    if rank < R:
        y[I] = X[:, -1] - X[:, :-1] @ y_B
    if rank == 0:
        y[B] = y_B
    ty = torch.from_numpy(y)
    dist.all_reduce(ty)  # every camera's solution from the rank that owns it
    S_full, rhs_full = oracle.bal_reduced_system(C, P, cam, pt, prob.observations, x, D, add_Dc=True)
    S_full = np.tril(S_full) + np.tril(S_full, -1).T
    y_ref = np.linalg.solve(S_full, rhs_full)
    assert np.linalg.norm(ty.numpy() - y_ref) <= 1e-9 * np.linalg.norm(y_ref), np.linalg.norm(ty.numpy() - y_ref) / np.linalg.norm(y_ref)
    dist.barrier()
    if rank == 0:
        print("DIST_SEGMENTS_OK world=%d segments=%d separator_unknowns=%d" % (world, R, B.size))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
