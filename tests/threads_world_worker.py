"""A world of N ranks as N THREADS of ONE process sharing GPU 0 — what makes worlds of 8 (and more) possible on a box that allows
six processes on its card.  Every thread is a rank: its own Problem, its own solver, its own stream; the all-reduce hook
copies the rank's buffer to the host, meets the other ranks at a barrier, sums the buffers in rank order (every rank the
same sum: the ranks stay bitwise equal) and copies the sum back.  The solvers factor CONCURRENTLY on the device's shared
look-ahead streams (enqueued as one unit per factorisation: chol_kernels.hip, DeviceQueues::enqueue_mutex).

  python tests/threads_world_worker.py <world> <mode: segmented | sharded | rows> <shape> [segments | 0] [scrambled | kept]

Used by tests/test_gpu_parity.py::test_world_of_eight_ranks_as_threads_*."""
import ctypes
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk, solve_bal_gpu  # noqa: E402


class ThreadWorld:
    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.bufs = [None] * n
        self.calls = [0] * n
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

    def hook(self, rank):
        def allreduce(ptr, count, stream):
            self.calls[rank] += 1
            # Copies on the SOLVER'S stream, never on the null stream: the factorisation's CU-masked streams are blocking streams, and
            # a null-stream operation (a plain hipMemcpy) of one thread waits for another rank's resident potrf server — which
            # waits for launches that queue up behind that very operation: every chain in flight would sit out its 1 s time-out.
            host = np.empty(count)
            assert self.hip.hipMemcpyAsync(host.ctypes.data, ptr, 8 * count, 2, stream) == 0
            assert self.hip.hipStreamSynchronize(stream) == 0
            self.bufs[rank] = host
            self.barrier.wait(timeout=300)
            total = self.bufs[0].copy()
            for r in range(1, self.n):
                assert self.bufs[r].shape == total.shape, "the ranks all-reduce buffers of different sizes"
                total += self.bufs[r]
            self.barrier.wait(timeout=300)  # every rank has read every buffer before any rank's next call replaces one
            assert self.hip.hipMemcpyAsync(ptr, total.ctypes.data, 8 * count, 1, stream) == 0
            assert self.hip.hipStreamSynchronize(stream) == 0
        return allreduce


def run_ranks(world, body):
    """body(rank, hook) -> result, on `world` threads; returns the results (raises the first exception)."""
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def main(rank):
        try:
            results[rank] = body(rank, tw.hook(rank))
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, e))
            tw.barrier.abort()
    threads = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    return results, tw


def main():
    world, mode, shape = int(sys.argv[1]), sys.argv[2], [int(v) for v in sys.argv[3].split(",")]
    if mode == "rows":
        from skeres_amd import dense_synth
        from dist_dense_rows_worker import solve
        m, n = shape
        consts, x_star = dense_synth.generate(m, n, seed=5)
        x_plain, s_plain, _ = solve(sk, consts, n)
        results, tw = run_ranks(world, lambda rank, hook: solve(sk, consts, n, hook, rank, world))
        for x, s, used in results:
            assert used == "sharded"
            a, b = [it["cost"] for it in s.iterations()], [it["cost"] for it in s_plain.iterations()]
            assert abs(len(a) - len(b)) <= 1
            for k in range(min(6, len(a), len(b))):
                assert abs(a[k] - b[k]) <= 1e-10 * max(b[k], 1e-300), (k, a[k], b[k])
            assert np.array_equal(x, results[0][0])  # the ranks agree bit for bit
        print("THREADS_WORLD_OK world=%d mode=rows iterations=%d" % (world, results[0][1].numIterations()))
        return
    prob = bal.generate(shape[0], shape[1], shape[2], seed=shape[3])
    x_plain, s_plain = solve_bal_gpu(prob)

    scrambled = len(sys.argv) > 5 and sys.argv[5] == "scrambled"
    kept = len(sys.argv) > 5 and sys.argv[5] == "kept"  # twelve retained points: their pseudo-cameras border the root and every segment's front

    def body(rank, hook):
        if scrambled and rank == 1:
            # this rank keeps its cameras in REVERSE order in memory: its memory-order candidate for the camera sequence differs from
            # the other ranks', the ranks notice (a hash of order and envelope) and all fall back to the rank-invariant candidates
            C = prob.num_cameras
            x0 = prob.parameters.copy()
            x0[:9 * C] = prob.parameters[:9 * C].reshape(C, 9)[::-1].ravel()
            params = sk.RichDoubleArray.fromArray(x0)
            problem = sk.Problem()
            loss = sk.PredefinedLossFunctions.trivialLoss()
            offs = np.stack([9 * (C - 1 - prob.camera_index.astype(np.int64)), 9 * C + 3 * prob.point_index.astype(np.int64)], axis=1)
            problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, loss, params, offs)
        else:
            problem, params, loss = bal_problem_to_sk(prob)
        options = sk.Solver.Options()
        options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
        options.setDistributed(rank, world, hook)
        options.setDistributionMode({"sharded": 1, "segmented": 3}[mode])
        if kept:
            options.setRetainedPoints("on", 12)
        solver = sk.StepSolver(options, problem)
        if kept:
            assert 3 <= solver.stat("retained_points") <= 12, solver.stat("retained_points")
        used = solver.distribution()[0]
        segments = int(solver.stat("segments"))
        while not solver.step():
            pass
        summary = sk.Solver.Summary()
        solver.finish(summary)
        x = params.toArray(prob.num_parameters)
        if scrambled and rank == 1:
            C = prob.num_cameras
            x[:9 * C] = x[:9 * C].reshape(C, 9)[::-1].ravel()
        return x, summary, used, segments
    results, tw = run_ranks(world, body)
    b = [it["cost"] for it in s_plain.iterations()]
    for x, summary, used, segments in results:
        assert used == mode, (used, mode)
        if mode == "segmented" and len(sys.argv) > 4 and int(sys.argv[4]) > 0:
            assert segments == int(sys.argv[4]), (segments, sys.argv[4])
        a = [it["cost"] for it in summary.iterations()]
        assert abs(len(a) - len(b)) <= 1, (len(a), len(b), a[:6], b[:6])
        for k in range(min(5, len(a), len(b))):
            assert abs(a[k] - b[k]) <= 1e-10 * b[k], (k, a[k], b[k])
        assert abs(summary.finalCost() - s_plain.finalCost()) <= 1e-9 * s_plain.finalCost()
        assert np.array_equal(x, results[0][0])  # every rank ends with ALL parameters, bit for bit the same
        assert np.abs(x - x_plain).max() <= 1e-6 * max(1.0, np.abs(x_plain).max())
    print("THREADS_WORLD_OK world=%d mode=%s segments=%d calls=%d iterations=%d" % (world, mode, results[0][3], tw.calls[0], results[0][1].numIterations()))


if __name__ == "__main__":
    main()
