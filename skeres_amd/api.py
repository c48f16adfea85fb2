"""ctypes binding of libskeres_amd.so that mirrors the reference's Scala API.

Names, argument meaning and error behaviour follow the reference so the parity
tests read like its own specs (core/src/test/scala/.../AutodiffCostFuntionSpec.scala).
``require`` failures of the Scala side surface as ``ValueError`` here
(IllegalArgumentException there); native failures as ``SkeresError``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libskeres_amd.so")


class SkeresError(RuntimeError):
    pass


_dp = C.POINTER(C.c_double)
_dpp = C.POINTER(_dp)
_ip = C.POINTER(C.c_int)
EVALUATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dpp, _dp, _dpp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None

_SIGS = {
    "sk_version": (C.c_char_p, []),
    "sk_last_error": (C.c_char_p, []),
    "sk_init_logging": (None, [C.c_char_p]),
    "sk_device_count": (C.c_int, []),
    "sk_array_new": (_dp, [C.c_int]),
    "sk_array_free": (None, [_dp]),
    "sk_array_getitem": (C.c_double, [_dp, C.c_int]),
    "sk_array_setitem": (None, [_dp, C.c_int, C.c_double]),
    "sk_array_slice": (_dp, [_dp, C.c_int]),
    "sk_array_copy_in": (None, [_dp, _dp, C.c_int]),
    "sk_array_copy_out": (None, [_dp, _dp, C.c_int]),
    "sk_matrix_is_null": (C.c_int, [_dpp]),
    "sk_matrix_row": (_dp, [_dpp, C.c_int]),
    "sk_ptrvec_new": (C.c_void_p, []),
    "sk_ptrvec_free": (None, [C.c_void_p]),
    "sk_ptrvec_add": (None, [C.c_void_p, _dp]),
    "sk_ptrvec_size": (C.c_int, [C.c_void_p]),
    "sk_ptrvec_get": (_dp, [C.c_void_p, C.c_int]),
    "sk_ptrvec_set": (None, [C.c_void_p, C.c_int, _dp]),
    "sk_ptrvec_to_pointer_pointer": (_dpp, [C.c_void_p]),
    "sk_rotation_apply": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp]),
    "sk_loss_trivial": (C.c_void_p, []),
    "sk_loss_huber": (C.c_void_p, [C.c_double]),
    "sk_local_parameterization_identity": (C.c_void_p, [C.c_int]),
    "sk_local_parameterization_subset": (C.c_void_p, [C.c_int, _ip, C.c_int]),
    "sk_local_parameterization_quaternion": (C.c_void_p, []),
    "sk_local_parameterization_homogeneous_vector": (C.c_void_p, [C.c_int]),
    "sk_local_parameterization_free": (None, [C.c_void_p]),
    "sk_local_parameterization_global_size": (C.c_int, [C.c_void_p]),
    "sk_local_parameterization_local_size": (C.c_int, [C.c_void_p]),
    "sk_local_parameterization_plus": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, _dp]),
    "sk_local_parameterization_compute_jacobian": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    "sk_problem_add_parameter_block": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_void_p]),
    "sk_problem_set_parameterization": (C.c_int, [C.c_void_p, _dp, C.c_void_p]),
    "sk_problem_set_parameter_block_constant": (C.c_int, [C.c_void_p, _dp]),
    "sk_problem_set_parameter_block_variable": (C.c_int, [C.c_void_p, _dp]),
    "sk_loss_soft_l_one": (C.c_void_p, [C.c_double]),
    "sk_loss_cauchy": (C.c_void_p, [C.c_double]),
    "sk_loss_tukey": (C.c_void_p, [C.c_double]),
    "sk_loss_tolerant": (C.c_void_p, [C.c_double, C.c_double]),
    "sk_loss_composed": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "sk_loss_scaled": (C.c_void_p, [C.c_void_p, C.c_double]),
    "sk_loss_evaluate": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    "sk_loss_free": (None, [C.c_void_p]),
    "sk_cost_function_new_autodiff": (C.c_void_p, [C.c_int, _dp, C.c_int]),
    "sk_cost_function_new_callback": (C.c_void_p, [EVALUATE_FN, C.c_void_p, C.c_int, _ip, C.c_int]),
    "sk_cost_function_free": (None, [C.c_void_p]),
    "sk_cost_function_num_residuals": (C.c_int, [C.c_void_p]),
    "sk_cost_function_num_parameter_blocks": (C.c_int, [C.c_void_p]),
    "sk_cost_function_parameter_block_size": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_cost_function_evaluate": (C.c_int, [C.c_void_p, _dpp, _dp, _dpp]),
    "sk_problem_new": (C.c_void_p, []),
    "sk_problem_free": (None, [C.c_void_p]),
    "sk_problem_add_residual_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _dpp, C.c_int, _ip]),
    "sk_problem_add_residual_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_void_p, _dpp]),
    "sk_problem_add_residual_blocks_tape": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, C.c_void_p, _dpp]),
    "sk_cost_function_new_tape": (C.c_void_p, [C.c_int, _ip, C.c_int, _ip, C.c_int, _dp, C.c_int, C.c_int, _ip, _dp, C.c_int]),
    "sk_problem_add_dense_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_void_p, _dp, C.c_int]),
    "sk_problem_num_residual_blocks": (C.c_int, [C.c_void_p]),
    "sk_problem_num_parameter_blocks": (C.c_int, [C.c_void_p]),
    "sk_problem_num_parameters": (C.c_int, [C.c_void_p]),
    "sk_problem_num_residuals": (C.c_int, [C.c_void_p]),
    "sk_options_new": (C.c_void_p, []),
    "sk_options_free": (None, [C.c_void_p]),
    "sk_options_set_linear_solver_type": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_minimizer_type": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_max_num_iterations": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_minimizer_progress_to_stdout": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_function_tolerance": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_gradient_tolerance": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_parameter_tolerance": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_initial_trust_region_radius": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_max_trust_region_radius": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_min_trust_region_radius": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_min_relative_decrease": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_min_lm_diagonal": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_max_lm_diagonal": (C.c_int, [C.c_void_p, C.c_double]),
    "sk_options_set_jacobi_scaling": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_max_num_consecutive_invalid_steps": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_device": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_cholesky_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "sk_options_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sk_options_set_distributed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]),
    "sk_options_set_reduce_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "sk_reduce_buffer_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p]),
    "sk_summary_new": (C.c_void_p, []),
    "sk_summary_free": (None, [C.c_void_p]),
    "sk_summary_initial_cost": (C.c_double, [C.c_void_p]),
    "sk_summary_final_cost": (C.c_double, [C.c_void_p]),
    "sk_summary_num_iterations": (C.c_int, [C.c_void_p]),
    "sk_summary_num_successful_steps": (C.c_int, [C.c_void_p]),
    "sk_summary_num_unsuccessful_steps": (C.c_int, [C.c_void_p]),
    "sk_summary_termination_type": (C.c_int, [C.c_void_p]),
    "sk_summary_message": (C.c_char_p, [C.c_void_p]),
    "sk_summary_brief_report": (C.c_char_p, [C.c_void_p]),
    "sk_summary_full_report": (C.c_char_p, [C.c_void_p]),
    "sk_summary_num_logged_iterations": (C.c_int, [C.c_void_p]),
    "sk_summary_iteration_field": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "sk_summary_phase_seconds": (C.c_double, [C.c_void_p, C.c_int]),
    "sk_summary_linear_solver_type_used": (C.c_int, [C.c_void_p]),
    "sk_summary_linear_solver_type_given": (C.c_int, [C.c_void_p]),
    "sk_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sk_solver_create": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "sk_solver_free": (None, [C.c_void_p]),
    "sk_solver_step": (C.c_int, [C.c_void_p, _ip]),
    "sk_solver_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sk_solver_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_solver_kernel_seconds": (C.c_double, [C.c_void_p, C.c_char_p, _ip]),
    "sk_solver_syrk_flops_per_solve": (C.c_double, [C.c_void_p]),
    "sk_solver_syrk_c_bytes_per_solve": (C.c_double, [C.c_void_p]),
    "sk_solver_distribution": (C.c_int, [C.c_void_p, _dp, _dp]),
    "sk_solver_stat": (C.c_int, [C.c_void_p, C.c_char_p, _dp]),
    "sk_last_status": (C.c_int, []),
    "sk_cholesky_solve_dissected": (C.c_int, [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sk_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "sk_allreduce_rccl_init": (C.c_void_p, [C.c_int, C.c_int, C.c_void_p]),
    "sk_allreduce_rccl_create": (C.c_void_p, [C.c_void_p]),
    "sk_allreduce_rccl_free": (None, [C.c_void_p]),
    "sk_allreduce_rccl_calls": (C.c_long, [C.c_void_p]),
    "sk_allreduce_rccl_fn": (C.c_void_p, []),
    "sk_cholesky_solve_segments": (C.c_int, [C.c_int, _dp, _dp, _dp, C.c_int, _ip, C.c_int, C.c_int]),
    "sk_cholesky_solve_ex": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip, C.c_int]),
    "sk_problem_border_plan": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), _ip, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sk_cholesky_solve_bordered": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int]),
    "sk_options_set_distribution_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_cholesky_envelope": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_cholesky_dissection": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_cholesky_border": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_retained_points": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "sk_problem_retained_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sk_options_set_resident_kernels": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_graph_replay": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_options_set_max_segments": (C.c_int, [C.c_void_p, C.c_int]),
    "sk_cholesky_solve": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int]),
    "sk_synth_dense_targets": (C.c_int, [C.c_double, C.c_int, C.c_int, _dp, _dp]),
    "sk_problem_point_partition": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _ip]),
    "sk_problem_segment_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _ip]),
}


def exported_symbols():
    """Every entry point include/skeres_amd.h declares."""
    return sorted(_SIGS)


def lib():
    """Load libskeres_amd.so (raises loudly when it has not been built).  SKERES_AMD_LIBRARY=<path> loads another build of it
    instead — the fault-injection build of the tests (libskeres_amd_testing.so, `make -C skeres_amd/csrc testing`)."""
    global _lib
    if _lib is None:
        global _LIB_PATH
        _LIB_PATH = os.environ.get("SKERES_AMD_LIBRARY", _LIB_PATH)
        if not os.path.exists(_LIB_PATH):
            raise SkeresError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C skeres_amd/csrc). There is no CPU fallback." % _LIB_PATH)
        L = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SkeresError("status %d: %s" % (rc, lib().sk_last_error().decode()))


def device_count():
    return lib().sk_device_count()


# ---------------------------------------------------------------------------
# native memory helpers
# ---------------------------------------------------------------------------
class DoubleArray:
    """com.google.ceres.DoubleArray (ceres.i:95-96): a native double[n] the caller owns."""

    def __init__(self, n=None, _ptr=None, _owner=None):
        if _ptr is not None:
            self._p, self._own, self._keep = _ptr, False, _owner
        else:
            self._p, self._own, self._keep = lib().sk_array_new(int(n)), True, None
            if not self._p:
                raise MemoryError("sk_array_new(%r)" % n)
        self.n = n

    def __del__(self):
        if getattr(self, "_own", False) and self._p and _lib is not None:
            _lib.sk_array_free(self._p)
            self._p = None

    def getitem(self, i):
        return lib().sk_array_getitem(self._p, i)

    def setitem(self, i, x):
        lib().sk_array_setitem(self._p, i, float(x))

    get, set = getitem, setitem  # RichDoubleArray.get / set (CORE/RichDoubleArray.scala:20,27)

    def cast(self):
        return self._p

    toPointer = cast

    @staticmethod
    def frompointer(p, owner=None):
        return DoubleArray(_ptr=p, _owner=owner)

    def slice(self, start):  # RichDoubleArray.slice (CORE/RichDoubleArray.scala:52)
        return DoubleArray(_ptr=lib().sk_array_slice(self._p, int(start)), _owner=self)

    def copyFrom(self, values):  # CORE/RichDoubleArray.scala:36-39 (bulk)
        a = np.ascontiguousarray(values, dtype=np.float64)
        lib().sk_array_copy_in(self._p, a.ctypes.data_as(_dp), a.size)
        return self

    def toArray(self, length):  # CORE/RichDoubleArray.scala:65-69 (bulk)
        out = np.empty(int(length), dtype=np.float64)
        lib().sk_array_copy_out(self._p, out.ctypes.data_as(_dp), out.size)
        return out

    def isNull(self):
        return not bool(self._p)


class RichDoubleArray:
    """CORE/RichDoubleArray.scala:73-74 factories."""

    @staticmethod
    def ofSize(n):
        return DoubleArray(n)

    @staticmethod
    def fromArray(a):
        return DoubleArray(len(a)).copyFrom(a)


class StdVectorDoublePointer:
    """com.google.ceres.StdVectorDoublePointer (ceres.i:82)."""

    def __init__(self):
        self._v = lib().sk_ptrvec_new()
        self._keep = []

    def __del__(self):
        if getattr(self, "_v", None) and _lib is not None:
            _lib.sk_ptrvec_free(self._v)
            self._v = None

    def add(self, arr):
        self._keep.append(arr)
        lib().sk_ptrvec_add(self._v, arr.cast() if arr is not None else _dp())

    def size(self):
        return lib().sk_ptrvec_size(self._v)

    def toPointerPointer(self):
        return lib().sk_ptrvec_to_pointer_pointer(self._v)


class RichDoubleMatrix:
    """View over a native double** (CORE/RichDoubleMatrix.scala:32-99)."""

    def __init__(self, rows):
        self.rows = rows  # list of DoubleArray or None

    @staticmethod
    def ofSize(num_rows, num_cols):
        return RichDoubleMatrix([DoubleArray(num_cols) for _ in range(num_rows)])

    @staticmethod
    def fromArrays(*arrays):
        return RichDoubleMatrix([RichDoubleArray.fromArray(a) for a in arrays])

    def isNull(self):
        return False

    def hasRow(self, i):
        return self.rows[i] is not None

    def getRow(self, i):
        return self.rows[i]

    def get(self, i, j):
        return self.rows[i].get(j)

    def set(self, i, j, x):
        self.rows[i].set(j, x)

    def _as_pp(self):
        return (_dp * len(self.rows))(*[(r.cast() if r is not None else _dp()) for r in self.rows])


# ---------------------------------------------------------------------------
# cost functions
# ---------------------------------------------------------------------------
class CostFunction:
    """com.google.ceres.CostFunction director (ceres.i:48).  Subclass and override
    ``evaluate(parameters, residuals, jacobians) -> bool`` for a host-side cost
    function; ``parameters``/``jacobians`` are lists of numpy views (``jacobians``
    is None, or holds None for rows the solver does not want)."""

    def __init__(self):
        self._h = None
        self._num_residuals = 0
        self._block_sizes = []

    def setNumResiduals(self, n):
        self._num_residuals = int(n)

    def numResiduals(self):
        return self._num_residuals

    def parameterBlockSizes(self):
        return list(self._block_sizes)

    def evaluate(self, parameters, residuals, jacobians):  # pragma: no cover - abstract
        raise NotImplementedError

    def _handle(self):
        if self._h is None:
            nres, sizes = self._num_residuals, self._block_sizes

            def tramp(_user, params, res, jacs):
                try:
                    p = [np.ctypeslib.as_array(params[i], shape=(sizes[i],)) for i in range(len(sizes))]
                    r = np.ctypeslib.as_array(res, shape=(nres,))
                    j = None
                    if jacs:
                        j = [np.ctypeslib.as_array(jacs[i], shape=(nres, sizes[i])) if jacs[i] else None
                             for i in range(len(sizes))]
                    return 1 if self.evaluate(p, r, j) else 0
                except Exception:  # an exception must not unwind through native frames
                    import traceback
                    traceback.print_exc()
                    return 0

            self._tramp = EVALUATE_FN(tramp)
            bs = (C.c_int * len(sizes))(*sizes)
            self._h = lib().sk_cost_function_new_callback(self._tramp, None, nres, bs, len(sizes))
            if not self._h:
                raise ValueError(lib().sk_last_error().decode())
        return self._h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.sk_cost_function_free(self._h)
            self._h = None


class SizedCostFunction(CostFunction):
    """CORE/SizedCostFunction.scala:6-14."""

    def __init__(self, kNumResiduals, *N):
        super().__init__()
        if any(n < 0 for n in N):
            raise ValueError("Negative block size detected. Block size are: %s" % ", ".join(map(str, N)))
        if any(not (N[i] == 0 or N[i - 1] > 0) for i in range(1, len(N))):
            raise ValueError("Zero block cannot precede a non-zero block. Block sizes are (ignore trailing 0's): "
                             + ", ".join(map(str, N)))
        self.kNumResiduals = kNumResiduals
        self.N = tuple(N)
        self.setNumResiduals(kNumResiduals)
        self._block_sizes = [n for n in N]


class NumericDiffMethodType:  # ceres::NumericDiffMethodType (ceres/types.h [ext]) as SWIG exposes it
    CENTRAL, FORWARD, RIDDERS = range(3)


class NumericDiffOptions:
    """ceres::NumericDiffOptions (ceres.i): Ceres' documented defaults."""

    def __init__(self):
        self._relative_step_size = 1e-6
        self._ridders_relative_initial_step_size = 1e-2
        self._max_num_ridders_extrapolations = 10
        self._ridders_epsilon = 1e-12

    def getRelativeStepSize(self):
        return self._relative_step_size

    def setRelativeStepSize(self, v):
        self._relative_step_size = float(v)

    def getRiddersRelativeInitialStepSize(self):
        return self._ridders_relative_initial_step_size

    def setRiddersRelativeInitialStepSize(self, v):
        self._ridders_relative_initial_step_size = float(v)

    def getMaxNumRiddersExtrapolations(self):
        return self._max_num_ridders_extrapolations

    def getRiddersEpsilon(self):
        return self._ridders_epsilon


EpsilonDouble = float(np.spacing(1.0))  # CORE/package.scala:15, ulp(1.0)


def _host_views(cost, parameters, residuals, jacobians):
    """Arguments of CostFunction.evaluate as numpy views, whichever way the caller passed them:
    the solver's director trampoline hands numpy views already; user code (the reference's specs)
    hands RichDoubleMatrix / DoubleArray pointer objects."""
    nres, sizes = cost.numResiduals(), cost.parameterBlockSizes()
    if isinstance(parameters, RichDoubleMatrix):
        p = [np.ctypeslib.as_array(parameters.getRow(i).cast(), shape=(sizes[i],)) for i in range(len(sizes))]
    else:
        p = parameters
    r = np.ctypeslib.as_array(residuals.cast(), shape=(nres,)) if isinstance(residuals, DoubleArray) else residuals
    if jacobians is None or (isinstance(jacobians, RichDoubleMatrix) and jacobians.isNull()):
        j = None
    elif isinstance(jacobians, RichDoubleMatrix):
        j = [np.ctypeslib.as_array(jacobians.getRow(i).cast(), shape=(nres, sizes[i])) if jacobians.hasRow(i) else None
             for i in range(len(sizes))]
    else:
        j = jacobians
    return p, r, j


class CostFunctor:
    """CORE/CostFunctor.scala:25-29."""

    def __init__(self, kNumResiduals, *N):
        if kNumResiduals <= 0:
            raise ValueError("Nonpositive number of residuals specified: %d" % kNumResiduals)
        if any(n <= 0 for n in N):
            raise ValueError("Nonpositive parameter block sizes specified: %s" % ", ".join(map(str, N)))
        self.kNumResiduals = kNumResiduals
        self.N = tuple(N)


class NumericDiffCostFunctor(CostFunctor):
    """CORE/CostFunctor.scala:40-51.  Override ``apply(*x)`` (x[i]: numpy array of block i) to return the
    residuals, or an empty sequence to signal failure."""

    def toNumericDiffCostFunction(self, method, options=None):
        return NumericDiffCostFunction(method, options if options is not None else NumericDiffOptions(), self)

    def apply(self, *x):  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self, *x):
        return self.apply(*x)


class NumericDiffCostFunction(SizedCostFunction):
    """CORE/NumericDiffCostFunction.scala:69-162: Jacobians by forward or central differences of a host
    functor.  A host-side cost function by nature (the functor is caller code): the solver reaches it
    through the director path, exactly like any other CostFunction subclass."""

    def __init__(self, method, options, costFunctor):
        if method == NumericDiffMethodType.RIDDERS:
            raise ValueError("requirement failed: RIDDERS method not yet implemented")  # NumericDiffCostFunction.scala:74
        super().__init__(costFunctor.kNumResiduals, *costFunctor.N)
        self.method, self.options, self.costFunctor = method, options, costFunctor

    def evaluate(self, parameters, residuals, jacobians):
        p, r, j = _host_views(self, parameters, residuals, jacobians)
        f, N, nres = self.costFunctor, self.costFunctor.N, self.kNumResiduals
        min_step = np.sqrt(EpsilonDouble)  # :83-93: never below sqrt(epsilon)
        x = [np.array(p[i], dtype=np.float64) for i in range(len(N))]  # :95-98 private copies
        y_c = np.asarray(f(*x), dtype=np.float64)
        if y_c.size == 0:
            return False
        r[:] = y_c
        if j is None:
            return True
        scale = self.options.getRelativeStepSize()
        for i in range(len(N)):
            xi = x[i]
            steps = np.maximum(np.abs(xi) * scale, min_step)  # :117-119
            Ji = np.zeros((nres, N[i]))
            for col in range(N[i]):
                xic, sc = xi[col], steps[col]
                xi[col] = xic + sc
                y_f = np.asarray(f(*x), dtype=np.float64)
                xi[col] = xic
                if y_f.size == 0:
                    return False
                if self.method == NumericDiffMethodType.FORWARD:
                    Ji[:, col] = (y_f - y_c) * (1.0 / sc)
                else:
                    xi[col] = xic - sc
                    y_b = np.asarray(f(*x), dtype=np.float64)
                    xi[col] = xic
                    if y_b.size == 0:
                        return False
                    Ji[:, col] = (y_f - y_b) * (0.5 / sc)
            if j[i] is not None:
                j[i][:, :] = Ji
        return True


class AutoDiffCostFunctor:
    """CORE/CostFunctor.scala:31-51.  A functor whose generic body lives in the
    device functor registry (``FUNCTOR_ID``); ``consts`` are the doubles the
    Scala closure captures."""
    FUNCTOR_ID = None

    def __init__(self, kNumResiduals, *N, consts=()):
        if kNumResiduals <= 0:
            raise ValueError("Nonpositive number of residuals specified: %d" % kNumResiduals)
        if any(n <= 0 for n in N):
            raise ValueError("Nonpositive parameter block sizes specified: %s" % ", ".join(map(str, N)))
        self.kNumResiduals = kNumResiduals
        self.N = tuple(N)
        self.consts = tuple(float(c) for c in consts)

    def toAutoDiffCostFunction(self):
        return AutoDiffCostFunction(self)


class AutoDiffCostFunction(SizedCostFunction):
    """CORE/AutodiffCostFunction.scala:69-135 over a device functor."""

    def __init__(self, costFunctor):
        super().__init__(costFunctor.kNumResiduals, *costFunctor.N)
        self.costFunctor = costFunctor
        if costFunctor.FUNCTOR_ID is None:
            raise ValueError("functor %s has no device body registered" % type(costFunctor).__name__)

    def _handle(self):
        if self._h is None:
            c = np.asarray(self.costFunctor.consts, dtype=np.float64)
            self._h = lib().sk_cost_function_new_autodiff(self.costFunctor.FUNCTOR_ID,
                                                         c.ctypes.data_as(_dp) if c.size else _dp(), c.size)
            if not self._h:
                raise ValueError(lib().sk_last_error().decode())
            if lib().sk_cost_function_num_residuals(self._h) != self.kNumResiduals or \
                    [lib().sk_cost_function_parameter_block_size(self._h, i)
                     for i in range(lib().sk_cost_function_num_parameter_blocks(self._h))] != list(self.N):
                raise ValueError("functor sizes do not match its device body")
        return self._h

    def evaluate(self, parameters, residuals, jacobians):
        """evaluate(DoublePointerPointer, DoublePointer, DoublePointerPointer): Boolean
        — runs the device functor on the GPU for this one residual block."""
        pp = parameters._as_pp()
        jp = jacobians._as_pp() if jacobians is not None else None
        rc = lib().sk_cost_function_evaluate(self._handle(), pp, residuals.cast(), jp)
        if rc < 0:
            raise SkeresError(lib().sk_last_error().decode())
        return bool(rc)


class HostAutoDiffCostFunctor(CostFunctor):
    """A generic functor whose body is host code (CORE/CostFunctor.scala:31-38, the reference's only kind: its functors
    run on the JVM).  Override ``apply(*x)``: x[i] is the list of block i's values, floats or ``Jet``s (rotation.Jet
    has the arithmetic and ``skeres_amd.rotation.sqrt / exp / sin / ...`` the functions); return the residuals, or an
    empty sequence to signal failure.  Functors with a body in the device registry (``AutoDiffCostFunctor``) are
    evaluated on the GPU; this one reaches the solver through the director path, like any host ``CostFunction``."""

    def toAutoDiffCostFunction(self):
        return HostAutoDiffCostFunction(self)

    def apply(self, *x):  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self, *x):
        return self.apply(*x)


class HostAutoDiffCostFunction(SizedCostFunction):
    """CORE/AutodiffCostFunction.scala:69-135 on the host: residuals from the functor over doubles; Jacobians from
    one evaluation over Jets of dimension sum(N), block i's k-th value seeded with e_(offset_i + k)."""

    def __init__(self, costFunctor):
        super().__init__(costFunctor.kNumResiduals, *costFunctor.N)
        self.costFunctor = costFunctor

    def evaluate(self, parameters, residuals, jacobians):
        from .rotation import Jet
        p, r, j = _host_views(self, parameters, residuals, jacobians)
        f, N, nres = self.costFunctor, self.costFunctor.N, self.kNumResiduals
        if j is None:
            y = f(*[[float(v) for v in p[i]] for i in range(len(N))])
            if len(y) == 0:
                return False
            r[:] = [float(v) for v in y]
            return True
        dim, x, off = int(sum(N)), [], 0
        for i in range(len(N)):
            x.append([Jet(p[i][k], off + k, dim) for k in range(N[i])])
            off += N[i]
        y = f(*x)
        if len(y) == 0:
            return False
        off = 0
        for i in range(len(N)):
            if j[i] is not None:
                for row in range(nres):
                    j[i][row, :] = y[row].infinitesimal[off:off + N[i]] if isinstance(y[row], Jet) else 0.0
            off += N[i]
        r[:] = [float(v) for v in y]
        return True


class TracedCostFunctor(HostAutoDiffCostFunctor):
    """A generic functor WITHOUT a body in the device registry that still runs on the GPU: ``apply`` is run once on
    recording values (skeres_amd/tape.py: the ``T`` of CORE/CostFunctor.scala:40-51 instantiated a third time) and the
    recorded instruction list is evaluated on the device per residual block (include/skeres_amd.h:
    sk_cost_function_new_tape).  Write ``apply`` over generic values — arithmetic, ``skeres_amd.tape.sqrt / exp / log /
    sin / cos / ... / atan2`` and ``skeres_amd.tape.where(a > b, then, otherwise)`` for a data-dependent branch — and it
    also works over floats and ``rotation.Jet`` (``toHostAutoDiffCostFunction``: the director path, for comparison).
    ``captured``: the doubles the closure captures (the reference's ``observedX``, ``observedY``); inside ``apply`` read
    them with ``self.captured_values()``, which gives floats, or the recording's values while recording."""

    def __init__(self, kNumResiduals, *N, captured=()):
        super().__init__(kNumResiduals, *N)
        self.captured = tuple(float(c) for c in captured)
        self._traced_captured = None
        self._tape = None

    def captured_values(self):
        return list(self._traced_captured) if self._traced_captured is not None else list(self.captured)

    def tape(self):
        """(instructions [n, 5] int32, literals, number of registers, output operands): recorded once per functor object."""
        if self._tape is None:
            from . import tape as _tape
            self._tape = _tape.record(self, self.N, len(self.captured))
        return self._tape

    def withCaptured(self, *captured):
        """The same functor around other captured doubles (another observation): shares this one's recording."""
        import copy
        if len(captured) != len(self.captured):
            raise ValueError("the functor captures %d doubles, %d given" % (len(self.captured), len(captured)))
        self.tape()
        g = copy.copy(self)
        g.captured = tuple(float(c) for c in captured)
        return g

    def toAutoDiffCostFunction(self):
        return TracedCostFunction(self)

    def toHostAutoDiffCostFunction(self):
        return HostAutoDiffCostFunction(self)


class TracedCostFunction(SizedCostFunction):
    """CORE/AutodiffCostFunction.scala:69-135 over a recorded functor: evaluated on the device."""

    def __init__(self, costFunctor):
        super().__init__(costFunctor.kNumResiduals, *costFunctor.N)
        self.costFunctor = costFunctor

    def _handle(self):
        if self._h is None:
            f = self.costFunctor
            ins, consts, nregs, outs = f.tape()
            sizes = np.asarray(f.N, dtype=np.int32)
            cap = np.asarray(f.captured, dtype=np.float64)
            ins = np.ascontiguousarray(ins, dtype=np.int32)
            self._h = lib().sk_cost_function_new_tape(
                f.kNumResiduals, sizes.ctypes.data_as(_ip), len(f.N), ins.ctypes.data_as(_ip), ins.shape[0],
                consts.ctypes.data_as(_dp) if consts.size else _dp(), consts.size, int(nregs), outs.ctypes.data_as(_ip),
                cap.ctypes.data_as(_dp) if cap.size else _dp(), cap.size)
            if not self._h:
                raise ValueError(lib().sk_last_error().decode())
        return self._h

    def evaluate(self, parameters, residuals, jacobians):
        pp = parameters._as_pp()
        jp = jacobians._as_pp() if jacobians is not None else None
        rc = lib().sk_cost_function_evaluate(self._handle(), pp, residuals.cast(), jp)
        if rc < 0:
            raise SkeresError(lib().sk_last_error().decode())
        return bool(rc)


class CostFunctorAdapter(HostAutoDiffCostFunctor):
    """CORE/CostFunctionToFunctor.scala:50-123: a CostFunction as a generic functor, so that it can be called from
    inside another functor.  Over doubles it evaluates the cost function's residuals (:66-77); over Jets it evaluates
    residuals and Jacobians at the real parts and applies the chain rule, output[i] = residual[i] +
    sum_j J[i][j] * infinitesimal(input[j]) (:79-122).  An evaluation that fails gives an empty result, as there."""

    def __init__(self, cost):
        super().__init__(cost.numResiduals(), *cost.parameterBlockSizes())
        self.cost = cost

    def apply(self, *x):
        from .rotation import Jet
        N, nres = self.N, self.kNumResiduals
        if len(x) != len(N) or any(len(x[i]) != N[i] for i in range(len(N))):
            raise ValueError("Invalid sizes for x")
        jets = any(isinstance(v, Jet) for xi in x for v in xi)
        p = [np.array([float(v) for v in xi], dtype=np.float64) for xi in x]
        r = np.zeros(nres)
        if not jets:
            return list(r) if self.cost.evaluate(p, r, None) else []
        J = [np.zeros((nres, n)) for n in N]
        if not self.cost.evaluate(p, r, J):
            return []
        dim = max(len(v.infinitesimal) for xi in x for v in xi if isinstance(v, Jet))
        out = []
        for i in range(nres):
            inf = np.zeros(dim)
            for b in range(len(N)):
                for k in range(N[b]):
                    if isinstance(x[b][k], Jet):
                        inf += J[b][i, k] * x[b][k].infinitesimal
            out.append(Jet(r[i], inf))
        return out


def CostFunctionToFunctor(cost):
    """CORE/CostFunctionToFunctor.scala:13-15."""
    return CostFunctorAdapter(cost)


DynamicCostFunctionToFunctor = CostFunctionToFunctor  # :18-20: "only to preserve the ceres-solver class names"


class SnavelyReprojectionError(AutoDiffCostFunctor):  # EX/SimpleBundleAdjuster.scala:79-119
    FUNCTOR_ID = 1

    def __init__(self, observedX, observedY):
        super().__init__(2, 9, 3, consts=(observedX, observedY))


class ExponentialResidual(AutoDiffCostFunctor):  # EX/CurveFitting.scala:92-98
    FUNCTOR_ID = 2

    def __init__(self, x, y):
        super().__init__(1, 1, 1, consts=(x, y))


class PowellF1(AutoDiffCostFunctor):  # EX/Powell.scala:14-21
    FUNCTOR_ID = 3

    def __init__(self):
        super().__init__(1, 1, 1)


class PowellF2(AutoDiffCostFunctor):
    FUNCTOR_ID = 4

    def __init__(self):
        super().__init__(1, 1, 1)


class PowellF3(AutoDiffCostFunctor):
    FUNCTOR_ID = 5

    def __init__(self):
        super().__init__(1, 1, 1)


class PowellF4(AutoDiffCostFunctor):
    FUNCTOR_ID = 6

    def __init__(self):
        super().__init__(1, 1, 1)


class BinaryScalarCost(AutoDiffCostFunctor):  # TEST/AutodiffCostFuntionSpec.scala:14-26
    FUNCTOR_ID = 7

    def __init__(self, a):
        super().__init__(1, 2, 2, consts=(a,))


class BinaryVector3Cost(AutoDiffCostFunctor):  # TEST/AutodiffCostFuntionSpec.scala:55-69
    FUNCTOR_ID = 8

    def __init__(self, a):
        super().__init__(3, 2, 2, consts=(a,))


class TenParameterCost(AutoDiffCostFunctor):  # TEST/AutodiffCostFuntionSpec.scala:111-119
    FUNCTOR_ID = 9

    def __init__(self):
        super().__init__(1, *([1] * 10))


class QuaternionRotationError(AutoDiffCostFunctor):
    """r = R(q) p - t for a quaternion block q = (w, x, y, z), normalised first (no reference counterpart: a registered
    functor with a block of size 4 for the local parameterizations to act on)."""
    FUNCTOR_ID = 12

    def __init__(self, p, t):
        super().__init__(3, 4, consts=tuple(p) + tuple(t))


class HelloCostFunctor(AutoDiffCostFunctor):  # EX/HelloWorld.scala:11-14
    FUNCTOR_ID = 11

    def __init__(self):
        super().__init__(1, 1)


class LossFunction:
    """com.google.ceres.LossFunction as built by PredefinedLossFunctions (ceres.i:159-184)."""

    def __init__(self, h):
        if not h:
            raise SkeresError(lib().sk_last_error().decode())
        self._h = h

    def evaluate(self, sq_norm):
        """LossFunction::Evaluate on the device: rows (rho, rho', rho'') for every squared norm given."""
        s = np.ascontiguousarray(np.atleast_1d(sq_norm), dtype=np.float64)
        rho = np.zeros((len(s), 3))
        _check(lib().sk_loss_evaluate(self._h, s.ctypes.data_as(_dp), len(s), rho.ctypes.data_as(_dp)))
        return rho

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.sk_loss_free(self._h)
            self._h = None


_Loss = LossFunction


class PredefinedLossFunctions:  # ceres.i:168-184
    @staticmethod
    def trivialLoss():
        return LossFunction(lib().sk_loss_trivial())

    @staticmethod
    def huberLoss(a):
        return LossFunction(lib().sk_loss_huber(a))

    @staticmethod
    def softLOneLoss(a):
        return LossFunction(lib().sk_loss_soft_l_one(a))

    @staticmethod
    def cauchyLoss(a):
        return LossFunction(lib().sk_loss_cauchy(a))

    @staticmethod
    def tukeyLoss(a):
        return LossFunction(lib().sk_loss_tukey(a))

    @staticmethod
    def tolerantLoss(a, b):
        return LossFunction(lib().sk_loss_tolerant(a, b))

    @staticmethod
    def composedLoss(f, g):
        return LossFunction(lib().sk_loss_composed(f._h if f is not None else None, g._h if g is not None else None))

    @staticmethod
    def scaledLoss(rho, a):
        return LossFunction(lib().sk_loss_scaled(rho._h if rho is not None else None, a))


class LocalParameterization:
    """com.google.ceres.LocalParameterization as built by PredefinedLocalParameterizations (ceres.i:186-210)."""

    def __init__(self, h):
        if not h:
            raise ValueError(lib().sk_last_error().decode())
        self._h = h

    def globalSize(self):
        return lib().sk_local_parameterization_global_size(self._h)

    def localSize(self):
        return lib().sk_local_parameterization_local_size(self._h)

    def plus(self, x, delta):
        """LocalParameterization::Plus on the device; x [global] or [n, global], delta [local] or [n, local]."""
        x2 = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        d2 = np.ascontiguousarray(np.atleast_2d(delta), dtype=np.float64).reshape(len(x2), -1)
        if d2.shape[1] == 0:
            d2 = np.zeros((len(x2), 1))
        out = np.zeros_like(x2)
        _check(lib().sk_local_parameterization_plus(self._h, x2.ctypes.data_as(_dp), d2.ctypes.data_as(_dp), len(x2), out.ctypes.data_as(_dp)))
        return out if np.ndim(x) == 2 else out[0]

    def computeJacobian(self, x):
        """LocalParameterization::ComputeJacobian on the device: [global, local] (or [n, global, local])."""
        x2 = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        J = np.zeros((len(x2), self.globalSize(), self.localSize()))
        if J.size:
            _check(lib().sk_local_parameterization_compute_jacobian(self._h, x2.ctypes.data_as(_dp), len(x2), J.ctypes.data_as(_dp)))
        return J if np.ndim(x) == 2 else J[0]

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.sk_local_parameterization_free(self._h)
            self._h = None


class PredefinedLocalParameterizations:  # ceres.i:186-210
    @staticmethod
    def identity(size):
        return LocalParameterization(lib().sk_local_parameterization_identity(int(size)))

    @staticmethod
    def subset(size, constant_parameters):
        c = np.ascontiguousarray(list(constant_parameters), dtype=np.int32)
        return LocalParameterization(lib().sk_local_parameterization_subset(int(size), c.ctypes.data_as(_ip) if c.size else _ip(), c.size))

    @staticmethod
    def quaternion():
        return LocalParameterization(lib().sk_local_parameterization_quaternion())

    @staticmethod
    def homogeneousVector(size):
        return LocalParameterization(lib().sk_local_parameterization_homogeneous_vector(int(size)))


# ---------------------------------------------------------------------------
# Problem / Solver
# ---------------------------------------------------------------------------
class LinearSolverType:
    DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR = range(7)


class MinimizerType:
    LINE_SEARCH, TRUST_REGION = 0, 1


class TerminationType:
    CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE = range(5)


class Problem:
    """CORE/Problem.scala:16-33 — never owns cost / loss objects; keeps Python
    references so the GC cannot collect objects native code still points at."""

    def __init__(self):
        self._h = lib().sk_problem_new()
        self._costs, self._losses, self._arrays = [], [], []

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.sk_problem_free(self._h)
            self._h = None

    def addResidualBlock(self, cost, loss, *x):
        self._costs.append(cost)
        self._losses.append(loss)
        self._arrays.extend(x)
        pp = (_dp * len(x))(*[xi.cast() for xi in x])
        rid = C.c_int()
        rc = lib().sk_problem_add_residual_block(self._h, cost._handle(), loss._h if loss is not None else None,
                                                 pp, len(x), C.byref(rid))
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)
        return rid.value

    # ceres::Problem members the reference's Problem inherits from the SWIG-wrapped class (CORE/Problem.scala:16)
    def addParameterBlock(self, values, size, parameterization=None):
        self._arrays.append(values)
        rc = lib().sk_problem_add_parameter_block(self._h, values.cast(), int(size), parameterization._h if parameterization is not None else None)
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def setParameterization(self, values, parameterization):
        rc = lib().sk_problem_set_parameterization(self._h, values.cast(), parameterization._h if parameterization is not None else None)
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def setParameterBlockConstant(self, values):
        rc = lib().sk_problem_set_parameter_block_constant(self._h, values.cast())
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def setParameterBlockVariable(self, values):
        rc = lib().sk_problem_set_parameter_block_variable(self._h, values.cast())
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def addResidualBlocks(self, functor_id, consts, loss, base, offsets):
        """Bulk form of the loop at EX/SimpleBundleAdjuster.scala:139-145.
        ``base`` is a DoubleArray, ``offsets`` an int array [n, num_blocks] of
        element offsets of each parameter block inside ``base``."""
        self._arrays.append(base)
        self._losses.append(loss)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = offsets.shape[0]
        addr = C.cast(base.cast(), C.c_void_p).value
        ptrs = np.ascontiguousarray(addr + 8 * offsets.ravel(), dtype=np.uint64)
        consts = np.ascontiguousarray(consts, dtype=np.float64)
        rc = lib().sk_problem_add_residual_blocks(self._h, int(functor_id), int(n), consts.ctypes.data_as(_dp),
                                                  loss._h if loss is not None else None,
                                                  C.cast(ptrs.ctypes.data, _dpp))
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def addResidualBlocksTraced(self, costFunctor, captured, loss, base, offsets):
        """The same for a recorded functor (``TracedCostFunctor``): n residual blocks of its body, block b with the
        captured doubles ``captured[b]`` ([n, len(costFunctor.captured)])."""
        cost = costFunctor.toAutoDiffCostFunction() if isinstance(costFunctor, TracedCostFunctor) else costFunctor
        self._costs.append(cost)
        self._arrays.append(base)
        self._losses.append(loss)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = offsets.shape[0]
        addr = C.cast(base.cast(), C.c_void_p).value
        ptrs = np.ascontiguousarray(addr + 8 * offsets.ravel(), dtype=np.uint64)
        captured = np.ascontiguousarray(captured, dtype=np.float64)
        rc = lib().sk_problem_add_residual_blocks_tape(self._h, cost._handle(), int(n), captured.ctypes.data_as(_dp) if captured.size else _dp(),
                                                       loss._h if loss is not None else None, C.cast(ptrs.ctypes.data, _dpp))
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def addDenseRows(self, functor_id, consts, loss, x, n):
        """num_rows residual blocks of a dense-row functor over the single parameter block x[0..n)
        (BASELINE.json config 5); consts is [num_rows, 3] = (seed, row index, y)."""
        self._arrays.append(x)
        self._losses.append(loss)
        consts = np.ascontiguousarray(consts, dtype=np.float64)
        rc = lib().sk_problem_add_dense_rows(self._h, int(functor_id), int(consts.shape[0]), consts.ctypes.data_as(_dp),
                                             loss._h if loss is not None else None, x.cast(), int(n))
        if rc == 1:
            raise ValueError(lib().sk_last_error().decode())
        _check(rc)

    def segmentPlan(self, max_segments, forced=True):
        """(number of segments, camera part of every residual block — its segment, or -k for separator k —, owner rank of every
        block's point): the segmented distribution's plan (sk_problem_segment_plan; host logic, no device needed)."""
        nb = self.numResidualBlocks()
        n = C.c_int(0)
        part, owner = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
        _check(lib().sk_problem_segment_plan(self._h, int(max_segments), int(bool(forced)), C.byref(n), part.ctypes.data_as(_ip), owner.ctypes.data_as(_ip)))
        return n.value, part, owner

    def borderPlan(self, mode="auto"):
        """The camera order of the reduced system and the border of loop-closure cameras as set-up derives them
        (sk_problem_border_plan; host logic, no device needed): a dict with border_cameras, position (of every residual
        block's camera inside the reduced system), gap, model_us, model_us_plain, envelope_fill."""
        mode = {"auto": 0, "on": 1, "off": 2}.get(mode, mode)
        nb = self.numResidualBlocks()
        n, gap = C.c_int(0), C.c_int(0)
        pos = np.zeros(nb, dtype=np.int32)
        us, us_plain, fill = C.c_double(0), C.c_double(0), C.c_double(0)
        _check(lib().sk_problem_border_plan(self._h, int(mode), C.byref(n), pos.ctypes.data_as(_ip), C.byref(gap), C.byref(us), C.byref(us_plain), C.byref(fill)))
        return {"border_cameras": n.value, "position": pos, "gap": gap.value, "model_us": us.value, "model_us_plain": us_plain.value,
                "envelope_fill": fill.value}

    def retainedPlan(self, mode="auto", max_points=0, border="auto"):
        """The points DENSE_SCHUR keeps in the reduced system instead of eliminating them, as set-up chooses them
        (sk_problem_retained_plan; host logic, no device needed): a dict with retained_points, retained_of_block (1 per residual
        block whose point is retained), model_us, model_us_without."""
        modes = {"auto": 0, "on": 1, "off": 2}
        nb = self.numResidualBlocks()
        n = C.c_int(0)
        flag = np.zeros(nb, dtype=np.int32)
        us, us_without = C.c_double(0), C.c_double(0)
        _check(lib().sk_problem_retained_plan(self._h, int(modes.get(mode, mode)), int(max_points), int(modes.get(border, border)), C.byref(n),
                                              flag.ctypes.data_as(_ip), C.byref(us), C.byref(us_without)))
        return {"retained_points": n.value, "retained_of_block": flag, "model_us": us.value, "model_us_without": us_without.value}

    def pointPartition(self, world):
        """(cuts[world+1], num_cameras, num_points, point_of_block[num residual blocks]):
        how sk_solve shards this problem over `world` ranks (host logic only)."""
        cuts = np.zeros(world + 1, dtype=np.int32)
        pob = np.zeros(self.numResidualBlocks(), dtype=np.int32)
        nc, npts = C.c_int(), C.c_int()
        rc = lib().sk_problem_point_partition(self._h, int(world), cuts.ctypes.data_as(_ip), C.byref(nc), C.byref(npts),
                                              pob.ctypes.data_as(_ip))
        _check(rc)
        return cuts, nc.value, npts.value, pob

    def numResidualBlocks(self):
        return lib().sk_problem_num_residual_blocks(self._h)

    def numParameterBlocks(self):
        return lib().sk_problem_num_parameter_blocks(self._h)

    def numParameters(self):
        return lib().sk_problem_num_parameters(self._h)

    def numResiduals(self):
        return lib().sk_problem_num_residuals(self._h)


class Solver:
    class Options:
        def __init__(self):
            self._h = lib().sk_options_new()
            self._keep = []

        def __del__(self):
            if getattr(self, "_h", None) and _lib is not None:
                _lib.sk_options_free(self._h)
                self._h = None

        def _set(self, name, v):
            rc = getattr(lib(), "sk_options_set_" + name)(self._h, v)
            if rc == 1:
                raise ValueError(lib().sk_last_error().decode())
            _check(rc)

        def setLinearSolverType(self, t): self._set("linear_solver_type", int(t))
        def setMinimizerType(self, t): self._set("minimizer_type", int(t))
        def setMaxNumIterations(self, n): self._set("max_num_iterations", int(n))
        def setMinimizerProgressToStdout(self, on): self._set("minimizer_progress_to_stdout", int(bool(on)))
        def setFunctionTolerance(self, v): self._set("function_tolerance", float(v))
        def setGradientTolerance(self, v): self._set("gradient_tolerance", float(v))
        def setParameterTolerance(self, v): self._set("parameter_tolerance", float(v))
        def setInitialTrustRegionRadius(self, v): self._set("initial_trust_region_radius", float(v))
        def setMaxTrustRegionRadius(self, v): self._set("max_trust_region_radius", float(v))
        def setMinTrustRegionRadius(self, v): self._set("min_trust_region_radius", float(v))
        def setMinRelativeDecrease(self, v): self._set("min_relative_decrease", float(v))
        def setMinLmDiagonal(self, v): self._set("min_lm_diagonal", float(v))
        def setMaxLmDiagonal(self, v): self._set("max_lm_diagonal", float(v))
        def setJacobiScaling(self, on): self._set("jacobi_scaling", int(bool(on)))
        def setMaxNumConsecutiveInvalidSteps(self, n): self._set("max_num_consecutive_invalid_steps", int(n))
        def setDevice(self, d): self._set("device", int(d))

        def setCholeskyTuning(self, group=0, lookahead=True):
            _check(lib().sk_options_set_cholesky_tuning(self._h, int(group), int(bool(lookahead))))
        def setStream(self, s): self._set("stream", C.c_void_p(int(s)))

        def setDistributed(self, rank, world, allreduce):
            """allreduce(device_ptr:int, count:int, stream:int) -> None: in-place sum over ranks."""
            def tramp(_u, ptr, count, stream):
                try:
                    allreduce(int(ptr), int(count), int(stream or 0))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = ALLREDUCE_FN(tramp)
            self._keep.append(cb)
            _check(lib().sk_options_set_distributed(self._h, int(rank), int(world), cb, None))

        def setDistributedRccl(self, rank, world, rccl):
            """The library's own RCCL hook (sk_allreduce_rccl_fn) instead of a Python callback: `rccl` is an RcclAllReduce."""
            fn = C.cast(lib().sk_allreduce_rccl_fn(), ALLREDUCE_FN)
            self._keep.append(rccl)
            _check(lib().sk_options_set_distributed(self._h, int(rank), int(world), fn, rccl._h))

        def setCholeskyEnvelope(self, on):
            """DENSE_SCHUR: factor only the blocks inside the reduced system's block envelope (default on; bit-identical)."""
            _check(lib().sk_options_set_cholesky_envelope(self._h, int(bool(on))))

        def setCholeskyDissection(self, mode):
            """DENSE_SCHUR: two-way dissection of the camera sequence: "auto" (default) / "on" / "off" (or 0 / 1 / 2)."""
            mode = {"auto": 0, "on": 1, "off": 2}.get(mode, mode)
            _check(lib().sk_options_set_cholesky_dissection(self._h, int(mode)))

        def setResidentKernels(self, on):
            """0: no kernel of this solver waits for another (the same plans, launch by launch): counter-collection runs."""
            _check(lib().sk_options_set_resident_kernels(self._h, int(bool(on))))

        def setGraphReplay(self, on):
            """0: launch-bound problems enqueue every launch instead of replaying their iteration as a hipGraph."""
            _check(lib().sk_options_set_graph_replay(self._h, int(bool(on))))

        def setMaxSegments(self, n):
            """Several ranks: cut the camera sequence into at most n segments (0: one per rank)."""
            _check(lib().sk_options_set_max_segments(self._h, int(n)))

        def setCholeskyBorder(self, mode):
            """DENSE_SCHUR: loop-closure cameras ordered into a trailing border: "auto" (default) / "on" / "off" (or 0 / 1 / 2)."""
            mode = {"auto": 0, "on": 1, "off": 2}.get(mode, mode)
            _check(lib().sk_options_set_cholesky_border(self._h, int(mode)))

        def setRetainedPoints(self, mode, max_points=0):
            """DENSE_SCHUR: the points with the widest tracks stay in the reduced system instead of being eliminated:
            "auto" (default) / "on" / "off" (or 0 / 1 / 2); at most max_points of them (0: the library's limit)."""
            mode = {"auto": 0, "on": 1, "off": 2}.get(mode, mode)
            _check(lib().sk_options_set_retained_points(self._h, int(mode), int(max_points)))

        def setDistributionMode(self, mode):
            """0 auto (default), 1 sharded, 2 replicated: what a world > 1 does (include/skeres_amd.h)."""
            _check(lib().sk_options_set_distribution_mode(self._h, int(mode)))

        def setReduceBuffer(self, ptr, nbytes):
            _check(lib().sk_options_set_reduce_buffer(self._h, C.c_void_p(int(ptr)), int(nbytes)))

    class Summary:
        def __init__(self):
            self._h = lib().sk_summary_new()

        def __del__(self):
            if getattr(self, "_h", None) and _lib is not None:
                _lib.sk_summary_free(self._h)
                self._h = None

        def initialCost(self): return lib().sk_summary_initial_cost(self._h)
        def finalCost(self): return lib().sk_summary_final_cost(self._h)
        def numIterations(self): return lib().sk_summary_num_iterations(self._h)
        def numSuccessfulSteps(self): return lib().sk_summary_num_successful_steps(self._h)
        def numUnsuccessfulSteps(self): return lib().sk_summary_num_unsuccessful_steps(self._h)
        def terminationType(self): return lib().sk_summary_termination_type(self._h)
        def message(self): return lib().sk_summary_message(self._h).decode()
        def briefReport(self): return lib().sk_summary_brief_report(self._h).decode()
        def fullReport(self): return lib().sk_summary_full_report(self._h).decode()
        def phaseSeconds(self, k): return lib().sk_summary_phase_seconds(self._h, k)
        def linearSolverTypeUsed(self): return lib().sk_summary_linear_solver_type_used(self._h)
        def linearSolverTypeGiven(self): return lib().sk_summary_linear_solver_type_given(self._h)

        def iterations(self):
            names = ["cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease",
                     "trust_region_radius", "step_is_valid", "step_is_successful"]
            return [{nm: lib().sk_summary_iteration_field(self._h, i, k) for k, nm in enumerate(names)}
                    for i in range(lib().sk_summary_num_logged_iterations(self._h))]


class ceres:
    """Module-level functions of the SWIG module (ceres.i)."""

    @staticmethod
    def initGoogleLogging(name):
        lib().sk_init_logging(name.encode())

    @staticmethod
    def solve(options, problem, summary):
        _check(lib().sk_solve(options._h, problem._h, summary._h))


class StepSolver:
    """Stepping form of ceres.solve (sk_solver_create / step / finish)."""

    def __init__(self, options, problem):
        self._keep = (options, problem)
        self._h = lib().sk_solver_create(options._h, problem._h)
        if not self._h:
            raise SkeresError("status %d: %s" % (lib().sk_last_status(), lib().sk_last_error().decode()))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.sk_solver_free(self._h)
            self._h = None

    def step(self):
        done = C.c_int()
        _check(lib().sk_solver_step(self._h, C.byref(done)))
        return bool(done.value)

    def finish(self, summary):
        _check(lib().sk_solver_finish(self._h, summary._h))

    def setKernelTiming(self, on):
        """0/False off, 1/True every named launch (diagnostic), 2 only the dominant kernel."""
        lib().sk_solver_set_kernel_timing(self._h, int(on))

    def kernelSeconds(self, name):
        n = C.c_int()
        s = lib().sk_solver_kernel_seconds(self._h, name.encode(), C.byref(n))
        return s, n.value

    def syrkFlopsPerSolve(self):
        return lib().sk_solver_syrk_flops_per_solve(self._h)

    def syrkCBytesPerSolve(self):
        return lib().sk_solver_syrk_c_bytes_per_solve(self._h)

    def stat(self, name):
        """A named figure of the solver's plan (sk_solver_stat), e.g. "envelope_fill", "cholesky_flops_full"."""
        v = C.c_double()
        _check(lib().sk_solver_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def distribution(self):
        """("sharded" | "replicated", measured all-reduce seconds, estimated seconds of work sharding removes per iteration)."""
        a, b = C.c_double(), C.c_double()
        mode = lib().sk_solver_distribution(self._h, C.byref(a), C.byref(b))
        return {1: "sharded", 2: "replicated", 3: "segmented"}.get(mode, "sharded"), a.value, b.value


def cholesky_solve(A, b, want_L=False, group=0, last=None, automatic_plan=False):
    """Dense SPD solve on the GPU (sk_cholesky_solve / sk_cholesky_solve_ex): `last` = block envelope (one entry per
    128-block column of the padded matrix), automatic_plan = the grouping DENSE_SCHUR uses by default."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    x = np.empty(n)
    L = np.empty((n, n)) if want_L else None
    lp = _ip()
    if last is not None:
        last = np.ascontiguousarray(last, dtype=np.int32)
        if last.shape[0] != (n + 1 + 127) // 128:
            raise ValueError("last needs one entry per 128-block column of the padded matrix")
        lp = last.ctypes.data_as(_ip)
    _check(lib().sk_cholesky_solve_ex(n, A.ctypes.data_as(_dp), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp),
                                      L.ctypes.data_as(_dp) if want_L else _dp(), int(group), lp, int(bool(automatic_plan))))
    return (x, L) if want_L else x


def cholesky_solve_bordered(A, b, border_begin, want_L=False, group=0, automatic_plan=False):
    """Dense SPD solve on the GPU with a bordered block envelope (sk_cholesky_solve_bordered): rows from `border_begin` on are
    the border (they may couple with any column), the rows before it a block-banded matrix; the envelope is A's own."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    x = np.empty(n)
    L = np.empty((n, n)) if want_L else None
    _check(lib().sk_cholesky_solve_bordered(n, A.ctypes.data_as(_dp), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp),
                                            L.ctypes.data_as(_dp) if want_L else _dp(), int(group), int(border_begin), int(bool(automatic_plan))))
    return (x, L) if want_L else x


def cholesky_solve_dissected(A, b, head, tail_begin, group=0, automatic_plan=False):
    """A x = b by two-way dissection on the GPU (sk_cholesky_solve_dissected): rows [0, head) and [tail_begin, n) are
    eliminated side by side (the tail back to front), the separator in between last."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    x = np.empty(n)
    _check(lib().sk_cholesky_solve_dissected(n, A.ctypes.data_as(_dp), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp), int(head), int(tail_begin),
                                             int(group), int(bool(automatic_plan))))
    return x


class RcclAllReduce:
    """Native RCCL all-reduce for the multi-GPU path (sk_allreduce_rccl_*): no torch involved.  Rank 0 makes the 128-byte
    id (RcclAllReduce.unique_id()) and hands it to the others; every rank then constructs RcclAllReduce(rank, world, id) on
    its own device."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().sk_rccl_unique_id(buf))
        return buf.raw

    def __init__(self, rank, world, unique_id):
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        self._h = lib().sk_allreduce_rccl_init(int(rank), int(world), self._id)
        if not self._h:
            raise SkeresError("status 6: %s" % lib().sk_last_error().decode())

    @property
    def calls(self):
        return int(lib().sk_allreduce_rccl_calls(self._h))

    def close(self):
        if self._h:
            lib().sk_allreduce_rccl_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cholesky_solve_segments(A, b, cuts, group=0, automatic_plan=False):
    """A x = b by multi-way dissection on the GPU (sk_cholesky_solve_segments): `cuts` = [(begin, end) of each separator],
    ascending; len(cuts) + 1 segments, each eliminated on its own, the separators' system last."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = A.shape[0]
    x = np.empty(n)
    c = np.ascontiguousarray(np.asarray(cuts, dtype=np.int32).reshape(-1))
    _check(lib().sk_cholesky_solve_segments(n, A.ctypes.data_as(_dp), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp), len(cuts) + 1,
                                            c.ctypes.data_as(_ip), int(group), int(bool(automatic_plan))))
    return x


def synth_dense_targets(seed, m, n, x_star):
    """tanh(A x_star) of the synthetic dense problem, computed on the GPU (sk_synth_dense_targets)."""
    xs = np.ascontiguousarray(x_star, dtype=np.float64)
    y = np.empty(int(m))
    _check(lib().sk_synth_dense_targets(float(seed), int(m), int(n), xs.ctypes.data_as(_dp), y.ctypes.data_as(_dp)))
    return y
