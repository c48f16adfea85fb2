"""Worker of test_native_rccl_hook_world_of_one_matches_plain_solve: the multi-GPU path through the LIBRARY'S OWN RCCL hook
(sk_allreduce_rccl_init / sk_allreduce_rccl_fn: no torch, no Python in the collective) with a communicator of one rank.  A
sum over one rank is the identity: the trajectory must equal the plain solve bit for bit.  (Two ranks need two GPUs: RCCL
refuses two ranks on one device; the multi-rank arithmetic is covered over gloo by tests/dist_gpu_worker2.py.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import skeres_amd as sk
    from skeres_amd import bal
    from helpers import bal_problem_to_sk, solve_bal_gpu
    assert "torch" not in sys.modules
    prob = bal.generate(16, 600, 2600, seed=11)
    x_plain, s_plain = solve_bal_gpu(prob)
    for rep in range(2):  # (twice: a communicator created, used, destroyed, and another one after it)
        problem, params, loss = bal_problem_to_sk(prob)
        options = sk.Solver.Options()
        options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
        rccl = sk.api.RcclAllReduce(0, 1, sk.api.RcclAllReduce.unique_id())
        options.setDistributedRccl(0, 1, rccl)
        options.setDistributionMode(1)
        summary = sk.Solver.Summary()
        sk.ceres.solve(options, problem, summary)
        assert rccl.calls >= 3 * (summary.numIterations() - 1), rccl.calls  # column norms, reduced system, scalars
        a = [it["cost"] for it in summary.iterations()]
        b = [it["cost"] for it in s_plain.iterations()]
        assert a == b, (a, b)
        assert np.array_equal(params.toArray(prob.num_parameters), x_plain)
        calls = rccl.calls
        rccl.close()
    assert "torch" not in sys.modules
    print("DIST_RCCL_OK calls=%d iterations=%d" % (calls, summary.numIterations()))


if __name__ == "__main__":
    main()
