"""Worker of tests/test_gpu_parity.py::test_*_on_one_gpu_with_a_real_exchange: the multi-GPU decompositions with a REAL
exchange between the ranks of a world of 2 to 5.  All ranks use GPU 0 (the test box has one GPU), so RCCL cannot carry the traffic
(it refuses two ranks on one device); the all-reduce hook stages through the host and sums with gloo.
Everything else — point partition, sharded evaluation and Schur assembly, the three exchanges per
iteration, the replicated factorisation, the gather of the point blocks — is the production path."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class HostStagedAllReduce:
    """sk_allreduce_fn over gloo: device -> pinned host -> gloo sum -> device."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self.calls = 0

    def __call__(self, ptr, count, stream):
        self.calls += 1
        assert self.hip.hipStreamSynchronize(stream) == 0
        host = torch.empty(count, dtype=torch.float64)
        assert self.hip.hipMemcpy(host.data_ptr(), ptr, 8 * count, 2) == 0  # device -> host
        dist.all_reduce(host)
        assert self.hip.hipMemcpy(ptr, host.data_ptr(), 8 * count, 1) == 0  # host -> device


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # torch initialises HIP before the library does
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import skeres_amd as sk
    from skeres_amd import bal
    from helpers import bal_problem_to_sk, solve_bal_gpu

    mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
    fields = sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "600", "2600", "11"]
    revisits = len(fields) > 4 and fields[4] == "rev"  # loop closures: the revisiting cameras go to a border of the reduced system (round 4)
    kept = len(fields) > 4 and fields[4] in ("kept", "kept2", "keptN")  # retained points: the widest tracks stay in the reduced system (round 4)
    kept2 = len(fields) > 4 and fields[4] == "kept2"  # ... in a SEGMENTED world cut in two: their pseudo-cameras are members of the one separator (round 5)
    keptN = len(fields) > 4 and fields[4] == "keptN"  # ... cut into a segment per rank: the pseudo-cameras are a border of the root and of every segment's front
    shape = [int(v) for v in fields[:4]]
    prob = bal.generate(shape[0], shape[1], shape[2], seed=shape[3], revisits=[(60, 350, 12, 40), (200, 520, 12, 40)] if revisits else ())
    x_plain, s_plain = solve_bal_gpu(prob, **({"setCholeskyBorder": "off"} if revisits else ({"setRetainedPoints": "off"} if kept else {})))
    problem, params, loss = bal_problem_to_sk(prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    hook = HostStagedAllReduce()
    options.setDistributed(rank, world, hook)
    options.setDistributionMode({"auto": 0, "sharded": 1, "replicated": 2, "segmented": 3}[mode])
    if revisits:
        options.setCholeskyBorder("on")
    if kept:
        options.setRetainedPoints("on", 12)
    if kept2:
        options.setMaxSegments(2)
    summary = sk.Solver.Summary()
    solver = sk.StepSolver(options, problem)
    used, t_allreduce, t_saved = solver.distribution()
    if revisits:
        # every rank orders the same cameras into the border (or set-up fails: the ranks compare a hash of order, envelope and tail
        # profile), and the packed all-reduce follows the bordered envelope
        assert solver.stat("border_cameras") >= 1 and solver.stat("dissected") == 0
        assert solver.stat("allreduce_bytes") < solver.stat("allreduce_bytes_full_triangle")
    if kept:
        # every rank retains the same twelve points (the hash the ranks compare covers them); whichever rank owns one writes its rows
        assert solver.stat("retained_points") == 12 and solver.stat("dissected") == (1 if kept2 or keptN else 0)
    if shape[0] >= 200 and mode == "sharded":
        # a camera sequence long enough for a band: only the blocks inside the envelope travel
        assert solver.stat("allreduce_bytes") < 0.9 * solver.stat("allreduce_bytes_full_triangle"), (solver.stat("allreduce_bytes"), solver.stat("allreduce_bytes_full_triangle"))
    while not solver.step():
        pass
    solver.finish(summary)
    segments = int(solver.stat("segments"))
    if mode == "segmented":
        # the camera sequence is cut: rank r's device eliminates segment r (the last one back to front); what travels is the
        # separators' system.  argv[3]: how many segments the sequence must have been cut into (ranks beyond: replicas)
        assert used == "segmented" and solver.stat("dissected") == 1
        if len(sys.argv) > 3:
            assert segments == int(sys.argv[3]), (segments, sys.argv[3])
        assert 2 <= segments <= world
        if shape[0] >= 200:  # (tracks of up to a quarter of the cameras: every further separator is a quarter of the system wide)
            assert solver.stat("allreduce_bytes") < (0.6 if segments == 2 else 1.0) * solver.stat("allreduce_bytes_full_triangle")
    elif mode != "auto":
        assert used == mode, (used, mode)
    else:  # a 2600-observation problem: the host-staged all-reduce costs far more than sharding saves
        assert used == "replicated" and t_allreduce > t_saved > 0.0, (used, t_allreduce, t_saved)
    if used == "replicated":
        # two exchanges at set-up (do all ranks have the look-ahead queues? they must factor by one plan; did all ranks pick the same
        # camera order with the memory-order candidate in play?), then only the probe of AUTO
        assert hook.calls == 2 + ((3 + 1) if mode == "auto" else 0), hook.calls
    a = [it["cost"] for it in summary.iterations()]
    b = [it["cost"] for it in s_plain.iterations()]
    assert abs(len(a) - len(b)) <= 1, (len(a), len(b), a[:8], b[:8])
    for k in range(min(5, len(a), len(b))):
        assert abs(a[k] - b[k]) <= 1e-10 * b[k], (k, a[k], b[k])  # summation order differs: tolerance, not bits
    assert abs(summary.finalCost() - s_plain.finalCost()) <= 1e-9 * s_plain.finalCost()
    x = params.toArray(prob.num_parameters)
    # every rank ends with ALL parameters (cameras replicated, points gathered) and the ranks agree bit for bit
    t = torch.from_numpy(x.copy())
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    assert np.abs(x - x_plain).max() <= 1e-6 * max(1.0, np.abs(x_plain).max())
    dist.barrier()
    if rank == 0:
        print("DIST_GPU2_OK world=%d mode=%s segments=%d calls=%d iterations=%d" % (world, mode, segments, hook.calls, summary.numIterations()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
