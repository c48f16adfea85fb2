"""Worker of tests/test_distributed_cpu.py: one of WORLD_SIZE gloo ranks on the CPU.

Checks the decomposition the multi-GPU path relies on: with the
points partitioned by the PRODUCT's host logic (sk_problem_point_partition),
the per-rank reduced camera systems — computed here by the CPU oracle, since no
GPU exists on this box — sum (all-reduce) to the reduced system of the whole
problem, and the per-rank costs sum to the whole cost."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = bal.generate(12, 300, 1500, seed=5)
    C, P = prob.num_cameras, prob.num_points
    problem, _, _ = bal_problem_to_sk(prob)
    cuts, nc, npts, point_of_block = problem.pointPartition(world)
    assert (nc, npts) == (C, P)
    # the generator numbers points in order of first appearance differently from the file order:
    # map through point_of_block (point number the solver assigns to each residual block)
    mine = (point_of_block >= cuts[rank]) & (point_of_block < cuts[rank + 1])
    counts = torch.tensor([int(mine.sum())], dtype=torch.int64)
    dist.all_reduce(counts)
    assert int(counts.item()) == prob.num_observations  # every observation on exactly one rank

    rng = np.random.default_rng(0)
    D = rng.uniform(0.5, 2.0, 9 * C + 3 * P)
    x = prob.parameters
    S, rhs = oracle.bal_reduced_system(C, P, prob.camera_index[mine], prob.point_index[mine], prob.observations[mine], x, D,
                                       add_Dc=False)
    r, _, _, cost = oracle.bal_evaluate(C, P, prob.camera_index[mine], prob.point_index[mine], prob.observations[mine], x,
                                        jacobians=False)
    payload = torch.from_numpy(np.concatenate([S.ravel(), rhs, [cost]]))
    dist.all_reduce(payload)  # the one collective of the linear solve: sum of (S | rhs | cost)
    n = 9 * C
    S_sum, rhs_sum, cost_sum = payload[: n * n].numpy().reshape(n, n), payload[n * n: n * n + n].numpy(), float(payload[-1])

    S_full, rhs_full = oracle.bal_reduced_system(C, P, prob.camera_index, prob.point_index, prob.observations, x, D, add_Dc=False)
    _, _, _, cost_full = oracle.bal_evaluate(C, P, prob.camera_index, prob.point_index, prob.observations, x, jacobians=False)
    scale = np.abs(S_full).max()
    assert np.abs(np.tril(S_sum) - np.tril(S_full)).max() <= 1e-12 * scale, np.abs(np.tril(S_sum) - np.tril(S_full)).max() / scale
    assert np.abs(rhs_sum - rhs_full).max() <= 1e-12 * np.abs(rhs_full).max()
    assert abs(cost_sum - cost_full) <= 1e-12 * cost_full
    # adding D_c^2 once after the reduction gives the system the single-GPU path factors
    S_ref, _ = oracle.bal_reduced_system(C, P, prob.camera_index, prob.point_index, prob.observations, x, D, add_Dc=True)
    S_sum = S_sum + np.diag(D[:n] ** 2)
    assert np.abs(np.tril(S_sum) - np.tril(S_ref)).max() <= 1e-12 * scale
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
