"""torch.distributed plumbing for the multi-GPU path.

The solver needs ONE collective: an in-place sum of a run of doubles in device
memory (``sk_allreduce_fn``).  ``TorchAllReduce`` provides it over
``torch.distributed`` (backend ``nccl`` == RCCL on ROCm, one process per GPU):

* the big reduced-system buffer is a torch tensor handed to the solver
  (``Solver.Options.setReduceBuffer``), so its all-reduce is zero-copy;
* the few small vectors live in solver-owned memory and go through a staging
  tensor with two device-to-device copies on the solver's stream.

PyTorch is plumbing only (device memory, streams, process group).
"""
import ctypes


class TorchAllReduce:
    def __init__(self, big=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.big = torch, dist, big
        self.base = big.data_ptr() if big is not None else 0
        self.nbytes = big.numel() * 8 if big is not None else 0
        self.stage = None
        self.calls = 0
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.hip.hipMemcpyAsync.restype = ctypes.c_int

    def __call__(self, ptr, count, stream):
        """In-place sum of `count` doubles at device address `ptr`, ordered with the work already enqueued on the
        solver's `stream` and before what the solver enqueues next: the collective is issued under that very stream
        (torch.cuda.ExternalStream), whatever torch's current stream happens to be — a solver left on its private
        stream (no options.setStream) would otherwise race with the copies and pack kernels around the collective."""
        torch, dist = self.torch, self.dist
        self.calls += 1
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream else _NullContext()
        with ctx:
            if self.base and self.base <= ptr and ptr + 8 * count <= self.base + self.nbytes:
                off = (ptr - self.base) // 8
                dist.all_reduce(self.big[off:off + count])
                return
            if self.stage is None or self.stage.numel() < count:
                self.stage = torch.empty(max(count, 1 << 16), dtype=torch.float64, device="cuda")
            st = self.stage[:count]
            if self.hip.hipMemcpyAsync(st.data_ptr(), ptr, 8 * count, 3, stream) != 0:  # 3 == hipMemcpyDeviceToDevice
                raise RuntimeError("hipMemcpyAsync D2D failed")
            dist.all_reduce(st)
            if self.hip.hipMemcpyAsync(ptr, st.data_ptr(), 8 * count, 3, stream) != 0:
                raise RuntimeError("hipMemcpyAsync D2D failed")


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def attach(options, problem, rank, world):
    """Allocate the reduce buffer with torch, register it and the hook on `options`.
    The hook issues its collectives on the stream the solver hands it, so the solver may run on any stream
    (options.setStream or its private one).  Returns the hook (keep it alive as long as the solver)."""
    import torch
    from . import api
    nbytes = api.lib().sk_reduce_buffer_bytes(options._h, problem._h)
    big = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda")
    options.setReduceBuffer(big.data_ptr(), nbytes)
    hook = TorchAllReduce(big)
    options.setDistributed(rank, world, hook)
    return hook
