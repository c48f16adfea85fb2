"""Worker of tests/test_gpu_parity.py::test_venice_1778_at_full_size_sharded_over_two_ranks: BASELINE.json configs[3] AS IT IS DEFINED
— BAL Venice-1778-993923 (C = 1778, P = 993 923, N = 5 001 946) with the residual blocks sharded over ranks and the reduced
system all-reduced — at FULL size, on a world of 2 ranks that share GPU 0 and exchange through gloo (the host-staged hook of
dist_gpu_worker2.py: RCCL refuses two ranks on one device).  One LM iteration against the single-GPU trajectory; the ranks'
parameters bit for bit; the balance of the point partition (sum of squared track lengths per rank) and what travels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dist_gpu_worker2 import HostStagedAllReduce  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import skeres_amd as sk
    from skeres_amd import bal
    from helpers import bal_problem_to_sk

    mode = sys.argv[1] if len(sys.argv) > 1 else "sharded"
    prob = bal.generate_named("venice-1778-993923", seed=1778, perturb=(1e-2, 1e-1, 1e-1))

    def run(distributed):
        problem, params, loss = bal_problem_to_sk(prob)
        options = sk.Solver.Options()
        options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
        options.setMaxNumIterations(1)
        hook = None
        if distributed:
            hook = HostStagedAllReduce()
            options.setDistributed(rank, world, hook)
            options.setDistributionMode({"auto": 0, "sharded": 1}[mode])
        solver = sk.StepSolver(options, problem)
        info = {"used": solver.distribution()[0] if distributed else "single", "allreduce_bytes": solver.stat("allreduce_bytes"),
                "full_triangle": solver.stat("allreduce_bytes_full_triangle"), "fill": solver.stat("envelope_fill"), "blocks": solver.stat("reduced_system_blocks"), "problem": problem}
        while not solver.step():
            pass
        summary = sk.Solver.Summary()
        solver.finish(summary)
        return params.toArray(prob.num_parameters), summary, info
    x_plain, s_plain, _ = run(False)
    x, s, info = run(True)
    a, b = s.iterations(), s_plain.iterations()
    assert len(a) == len(b) == 2
    for u, v in zip(a, b):
        for k, tol in (("cost", 1e-10), ("step_norm", 1e-9), ("gradient_max_norm", 1e-8)):
            assert abs(u[k] - v[k]) <= tol * max(abs(v[k]), 1e-300), (k, u[k], v[k])
    assert np.linalg.norm(x - x_plain) <= 1e-9 * np.linalg.norm(x_plain - prob.parameters)
    t = torch.from_numpy(x.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)  # every rank ends with ALL parameters, bit for bit the same
    # the point partition: contiguous runs of (nearly) equal sum of squared track lengths
    cuts, C, P, point_of_block = info["problem"].pointPartition(world)
    k = np.bincount(point_of_block, minlength=P).astype(np.float64)
    per_rank = np.array([np.sum(k[cuts[r]:cuts[r + 1]] ** 2) for r in range(world)])
    imbalance = per_rank.max() / per_rank.mean()
    assert imbalance <= 1.05, (imbalance, per_rank)
    # what travels per iteration in sharded mode: the lower-triangular 128-blocks inside the envelope, nothing else
    nblk = int(info["blocks"])  # (the reduced system's block rows: the cameras' — and the retained points', three to a nine-row pseudo-camera)
    assert nblk >= (9 * C + 1 + 127) // 128
    blocks = info["fill"] * 0.5 * nblk * (nblk + 1)
    if info["used"] == "sharded":
        assert abs(info["allreduce_bytes"] - blocks * 128 * 128 * 8) <= 1e-6 * info["allreduce_bytes"], (info["allreduce_bytes"], blocks)
        assert info["allreduce_bytes"] < 0.6 * info["full_triangle"]
    if mode == "sharded":
        assert info["used"] == "sharded"
    dist.barrier()
    if rank == 0:
        print("DIST_VENICE_OK world=%d mode=%s used=%s imbalance=%.4f allreduce_MB=%.1f of %.1f" % (
            world, mode, info["used"], imbalance, info["allreduce_bytes"] / 1e6, info["full_triangle"] / 1e6))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
