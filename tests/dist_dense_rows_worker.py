"""Worker of test_dense_rows_sharded_over_ranks: BASELINE.json config 5's path (dense rows over one block,
DENSE_NORMAL_CHOLESKY) with the ROWS sharded over the ranks: every rank forms J^T J of its rows
with the MFMA SYRK, the lower block triangle is all-reduced, the Cholesky runs replicated.  The ranks share GPU 0 and
exchange through gloo (host-staged hook), as tests/dist_gpu_worker2.py does."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dist_gpu_worker2 import HostStagedAllReduce  # noqa: E402


def solve(sk, consts, n, hook=None, rank=0, world=1):
    x = sk.DoubleArray(n)
    problem = sk.Problem()
    problem.addDenseRows(10, consts, None, x, n)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_NORMAL_CHOLESKY)
    o.setMaxNumIterations(50)
    if hook is not None:
        o.setDistributed(rank, world, hook)
    summary = sk.Solver.Summary()
    solver = sk.StepSolver(o, problem)
    mode = solver.distribution()[0]
    while not solver.step():
        pass
    solver.finish(summary)
    return x.toArray(n), summary, mode


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import skeres_amd as sk
    from skeres_amd import dense_synth
    m, n = (int(v) for v in sys.argv[1].split(","))
    consts, x_star = dense_synth.generate(m, n, seed=5)
    x_plain, s_plain, _ = solve(sk, consts, n)
    hook = HostStagedAllReduce()
    x, s, mode = solve(sk, consts, n, hook, rank, world)
    assert mode == "sharded", mode
    assert hook.calls >= 4 * (s.numIterations() - 1)
    a = [it["cost"] for it in s.iterations()]
    b = [it["cost"] for it in s_plain.iterations()]
    assert abs(len(a) - len(b)) <= 1, (len(a), len(b))
    for k in range(min(6, len(a), len(b))):
        assert abs(a[k] - b[k]) <= 1e-10 * max(b[k], 1e-300), (k, a[k], b[k])  # the sums are grouped by rank: tolerance, not bits
    assert np.abs(x - x_plain).max() <= 1e-8 * max(1.0, np.abs(x_plain).max())
    assert np.abs(x - x_star).max() < 0.1
    t = torch.from_numpy(x.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)  # replicated Cholesky of identical all-reduced sums: the ranks agree bit for bit
    dist.barrier()
    if rank == 0:
        print("DIST_DENSE_ROWS_OK world=%d m=%d n=%d iterations=%d" % (world, m, n, s.numIterations()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
