"""Worker of test_distributed_hook_path_world_of_one_matches_plain_solve: the whole multi-GPU
code path (partition, all-reduce hook over torch.distributed/RCCL, scalar exchange, gather of the
points) on ONE GPU with a process group of size 1.  A sum over one rank is the identity, so the
trajectory must equal the plain solve bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import skeres_amd as sk
    from skeres_amd import bal, dist as sk_dist
    from helpers import bal_problem_to_sk, solve_bal_gpu

    prob = bal.generate(16, 600, 2600, seed=11)
    x_plain, s_plain = solve_bal_gpu(prob)
    problem, params, loss = bal_problem_to_sk(prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    options.setStream(torch.cuda.current_stream().cuda_stream)
    hook = sk_dist.attach(options, problem, 0, 1)
    summary = sk.Solver.Summary()
    sk.ceres.solve(options, problem, summary)
    torch.cuda.synchronize()
    assert hook.calls >= 3 * (summary.numIterations() - 1), hook.calls  # column norms, reduced system, scalars
    a = [it["cost"] for it in summary.iterations()]
    b = [it["cost"] for it in s_plain.iterations()]
    assert a == b, (a, b)
    assert np.array_equal(params.toArray(prob.num_parameters), x_plain)
    dist.destroy_process_group()
    print("DIST_GPU_OK calls=%d iterations=%d" % (hook.calls, summary.numIterations()))


if __name__ == "__main__":
    main()
