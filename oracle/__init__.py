"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for ``oracle/liboracle.so`` (the CPU restatement of the
reference's hot path; see oracle/jet.hpp, functors.hpp, lm.cpp headers for
what each piece restates and how it is pinned).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product package
``skeres_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

DENSE_NORMAL_CHOLESKY, DENSE_QR, DENSE_SCHUR = 0, 1, 3
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
MAX_LOG = 256

# functor ids (oracle/functors.hpp; same numbering as include/skeres_amd.h)
SNAVELY, EXPONENTIAL, POWELL_F1, POWELL_F2, POWELL_F3, POWELL_F4 = 1, 2, 3, 4, 5, 6
BINARY_SCALAR, BINARY_VECTOR3, TEN_PARAMETER = 7, 8, 9
HELLO_WORLD = 11  # EX/HelloWorld.scala:11-14
QUATERNION_ROTATION = 12  # r = R(q) p - t (no reference counterpart; local-parameterization tests)
SYNTH_TANH_ROW = 10  # BASELINE.json config 5 (dense rows over one block of any size)


class Options(C.Structure):
    _fields_ = [
        ("linear_solver_type", C.c_int), ("max_num_iterations", C.c_int),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("jacobi_scaling", C.c_int),
        ("max_num_consecutive_invalid_steps", C.c_int), ("num_threads", C.c_int),
        ("cholesky_envelope", C.c_int),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("cost", C.c_double), ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double), ("step_norm", C.c_double),
        ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
        ("step_is_valid", C.c_int), ("step_is_successful", C.c_int), ("time_s", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("num_unsuccessful_steps", C.c_int), ("termination_type", C.c_int),
        ("num_logged", C.c_int), ("num_threads_used", C.c_int),
        ("total_time_s", C.c_double), ("t_linear_assemble_s", C.c_double),
        ("t_linear_cholesky_s", C.c_double), ("t_linear_backsub_s", C.c_double),
        ("message", C.c_char * 256), ("iterations", Iteration * MAX_LOG),
    ]

    def costs(self):
        return [self.iterations[i].cost for i in range(self.num_logged)]

    def iteration_seconds(self):
        """Wall-clock seconds of iterations 1 .. num_logged-1 (iteration 0 is the initial evaluation)."""
        t = [self.iterations[i].time_s for i in range(self.num_logged)]
        return [b - a for a, b in zip(t[:-1], t[1:])]


_lib = None


def build():
    """Compile the oracle (g++; seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.or_options_default.argtypes = [C.POINTER(Options)]
        L.or_functor_info.argtypes = [C.c_int, ip, ip, ip, ip]
        L.or_evaluate.argtypes = [C.c_int, dp, C.POINTER(dp), dp, C.POINTER(dp)]
        L.or_angle_axis_rotate_point.argtypes = [dp, dp, dp]
        L.or_angle_axis_to_rotation_matrix.argtypes = [dp, dp]
        L.or_solve.argtypes = [C.c_int, ip, dp, C.c_int, ip, dp, ip, ip, ip,
                               C.POINTER(Options), C.POINTER(Summary)]
        L.or_solve_bal.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp,
                                   C.POINTER(Options), C.POINTER(Summary)]
        L.or_bal_evaluate.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, dp, dp, dp, dp]
        L.or_bal_reduced_system.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, dp, C.c_int, dp, dp]
        L.or_rotation_apply.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]
        L.or_loss_evaluate.argtypes = [dp, C.c_int, C.c_double, dp]
        L.or_solve_loss.argtypes = [C.c_int, ip, dp, C.c_int, ip, dp, ip, ip, ip, dp, ip,
                                    C.POINTER(Options), C.POINTER(Summary)]
        L.or_solve_bal_loss.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, C.c_int, dp,
                                        C.POINTER(Options), C.POINTER(Summary)]
        L.or_solve_bal_masks.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, C.c_int, ip, ip, dp,
                                         C.POINTER(Options), C.POINTER(Summary)]
        L.or_solve_param.argtypes = [C.c_int, ip, dp, C.c_int, ip, dp, ip, ip, ip, dp, ip, ip, ip, ip,
                                     C.POINTER(Options), C.POINTER(Summary)]
        L.or_parameterization_local_size.argtypes = [C.c_int, C.c_int, C.c_int]
        L.or_parameterization_plus.argtypes = [C.c_int, C.c_int, ip, C.c_int, dp, dp, dp]
        L.or_parameterization_jacobian.argtypes = [C.c_int, C.c_int, ip, C.c_int, dp, dp]
        L.or_dense_rows_cost.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, dp]
        L.or_cholesky_lower.argtypes = [dp, C.c_int, C.c_int]
        L.or_cholesky_solve.argtypes = [dp, C.c_int, dp]
        L.or_cholesky_lower_envelope.argtypes = [dp, C.c_int, C.c_int, ip]
        L.or_cholesky_solve_envelope.argtypes = [dp, C.c_int, dp, ip]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_options(**kw):
    o = Options()
    lib().or_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def functor_info(fid):
    nr, nb, nc = C.c_int(), C.c_int(), C.c_int()
    sizes = (C.c_int * 16)()
    if not lib().or_functor_info(fid, C.byref(nr), C.byref(nb), C.byref(nc), sizes):
        raise ValueError("unknown functor id %d" % fid)
    return nr.value, [sizes[i] for i in range(nb.value)], nc.value


def evaluate(fid, consts, params, want_jacobians=True, null_rows=()):
    """AutoDiffCostFunction.evaluate for one residual block.

    Returns (ok, residuals, jacobians) with jacobians[i] row-major
    kNumResiduals x N(i), or None where the row pointer was null."""
    nres, sizes, _ = functor_info(fid)
    consts = np.ascontiguousarray(consts, dtype=np.float64) if len(consts) else np.zeros(1)
    blocks = [np.ascontiguousarray(p, dtype=np.float64) for p in params]
    dp = C.POINTER(C.c_double)
    pp = (dp * len(blocks))(*[_dp(b) for b in blocks])
    res = np.zeros(nres)
    jacs = None
    jp = None
    if want_jacobians:
        jacs = [None if i in null_rows else np.zeros(nres * n) for i, n in enumerate(sizes)]
        jp = (dp * len(blocks))(*[(_dp(j) if j is not None else dp()) for j in jacs])
    ok = lib().or_evaluate(fid, _dp(consts), pp, _dp(res), jp)
    if jacs is not None:
        jacs = [None if j is None else j.reshape(nres, -1) for j in jacs]
    return bool(ok), res, jacs


def angle_axis_rotate_point(aa, pt):
    aa = np.ascontiguousarray(aa, dtype=np.float64)
    pt = np.ascontiguousarray(pt, dtype=np.float64)
    out = np.zeros(3)
    lib().or_angle_axis_rotate_point(_dp(aa), _dp(pt), _dp(out))
    return out


def angle_axis_to_rotation_matrix(aa):
    aa = np.ascontiguousarray(aa, dtype=np.float64)
    R = np.zeros(9)
    lib().or_angle_axis_to_rotation_matrix(_dp(aa), _dp(R))
    return R.reshape(3, 3)


# ---- robust losses (oracle/loss.hpp) ----------------------------------------------
# A loss is None (trivial) or a tuple: ("huber", a), ("softlone", a), ("cauchy", a), ("tukey", a),
# ("tolerant", a, b), ("composed", f, g), ("scaled", rho_or_None, a).
_LOSS_TYPES = {"trivial": 0, "huber": 1, "softlone": 2, "cauchy": 3, "tukey": 4, "tolerant": 5, "composed": 6, "scaled": 7}


def _flatten_loss(loss, nodes):
    """Append `loss` to `nodes` (rows of 5: type, a, b, f, g; children first); returns its root index, -1 for trivial."""
    if loss is None or loss[0] == "trivial":
        return -1
    kind = loss[0]
    if kind == "composed":
        g = _flatten_loss(loss[2], nodes)
        f = _flatten_loss(loss[1], nodes)
        nodes.append([6.0, 0.0, 0.0, float(f), float(g)])
    elif kind == "scaled":
        f = _flatten_loss(loss[1], nodes)
        nodes.append([7.0, float(loss[2]), 0.0, float(f), -1.0])
    else:
        nodes.append([float(_LOSS_TYPES[kind]), float(loss[1]), float(loss[2]) if kind == "tolerant" else 0.0, -1.0, -1.0])
    return len(nodes) - 1


def loss_evaluate(loss, s):
    """(rho, rho', rho'') of `loss` at squared norm s."""
    nodes = []
    root = _flatten_loss(loss, nodes)
    arr = np.asarray(nodes if nodes else [[0.0] * 5], dtype=np.float64)
    rho = np.zeros(3)
    lib().or_loss_evaluate(_dp(arr), root, float(s), _dp(rho))
    return rho


def solve(block_sizes, x0, residual_blocks, options=None):
    """residual_blocks: list of (functor_id, consts, [param block indices]) or
    (functor_id, consts, [param block indices], loss)."""
    if any(len(rb) > 3 and rb[3] is not None for rb in residual_blocks):
        return _solve_loss(block_sizes, x0, residual_blocks, options)
    residual_blocks = [rb[:3] for rb in residual_blocks]
    o = options or default_options()
    bs = np.asarray(block_sizes, dtype=np.int32)
    x = np.array(x0, dtype=np.float64).copy()
    fids = np.asarray([rb[0] for rb in residual_blocks], dtype=np.int32)
    consts, coff, pidx, poff = [], [], [], [0]
    for _, c, idx in residual_blocks:
        coff.append(len(consts))
        consts.extend(c)
        pidx.extend(idx)
        poff.append(len(pidx))
    consts = np.asarray(consts + [0.0], dtype=np.float64)
    coff = np.asarray(coff, dtype=np.int32)
    pidx = np.asarray(pidx, dtype=np.int32)
    poff = np.asarray(poff, dtype=np.int32)
    s = Summary()
    rc = lib().or_solve(len(bs), _ip(bs), _dp(x), len(fids), _ip(fids), _dp(consts), _ip(coff),
                        _ip(pidx), _ip(poff), C.byref(o), C.byref(s))
    if rc != 0:
        raise RuntimeError("or_solve failed: %d" % rc)
    return x, s


P_IDENTITY, P_SUBSET, P_QUATERNION, P_HOMOGENEOUS_VECTOR, P_CONSTANT = range(5)  # oracle/parameterization.hpp


def _param_args(p):
    """p: None | ("identity",) | ("subset", [constant indices]) | ("quaternion",) | ("homogeneous",) | ("constant",)"""
    if p is None:
        return -1, []
    kind = {"identity": P_IDENTITY, "subset": P_SUBSET, "quaternion": P_QUATERNION, "homogeneous": P_HOMOGENEOUS_VECTOR,
            "constant": P_CONSTANT}[p[0]]
    return kind, list(p[1]) if kind == P_SUBSET else []


def parameterization_local_size(p, size):
    kind, const = _param_args(p)
    return lib().or_parameterization_local_size(kind, size, len(const))


def parameterization_plus(p, x, delta):
    kind, const = _param_args(p)
    x = np.ascontiguousarray(x, dtype=np.float64)
    d = np.ascontiguousarray(list(delta) + [0.0], dtype=np.float64)
    c = np.asarray(const + [0], dtype=np.int32)
    out = np.zeros_like(x)
    lib().or_parameterization_plus(kind, len(x), _ip(c), len(const), _dp(x), _dp(d), _dp(out))
    return out


def parameterization_jacobian(p, x):
    kind, const = _param_args(p)
    x = np.ascontiguousarray(x, dtype=np.float64)
    c = np.asarray(const + [0], dtype=np.int32)
    ls = parameterization_local_size(p, len(x))
    J = np.zeros((len(x), max(ls, 1)))
    lib().or_parameterization_jacobian(kind, len(x), _ip(c), len(const), _dp(x), _dp(J))
    return J[:, :ls]


def solve_param(block_sizes, x0, residual_blocks, parameterizations, options=None):
    """As solve(), with one parameterization (see _param_args) or None per parameter block."""
    o = options or default_options()
    bs = np.asarray(block_sizes, dtype=np.int32)
    x = np.array(x0, dtype=np.float64).copy()
    fids = np.asarray([rb[0] for rb in residual_blocks], dtype=np.int32)
    consts, coff, pidx, poff, nodes, roots, seen = [], [], [], [0], [], [], {}
    for rb in residual_blocks:
        coff.append(len(consts))
        consts.extend(rb[1])
        pidx.extend(rb[2])
        poff.append(len(pidx))
        loss = rb[3] if len(rb) > 3 else None
        key = repr(loss)
        if key not in seen:
            seen[key] = _flatten_loss(loss, nodes)
        roots.append(seen[key])
    ptype, pcoff, pconst = [], [0], []
    for p in parameterizations:
        kind, const = _param_args(p)
        ptype.append(kind)
        pconst.extend(const)
        pcoff.append(len(pconst))
    consts = np.asarray(consts + [0.0], dtype=np.float64)
    nodes = np.asarray(nodes if nodes else [[0.0] * 5], dtype=np.float64)
    arrs = [np.asarray(a, dtype=np.int32) for a in (coff, pidx, poff, roots, ptype, pcoff, pconst + [0])]
    s = Summary()
    rc = lib().or_solve_param(len(bs), _ip(bs), _dp(x), len(fids), _ip(fids), _dp(consts), _ip(arrs[0]), _ip(arrs[1]), _ip(arrs[2]),
                              _dp(nodes), _ip(arrs[3]), _ip(arrs[4]), _ip(arrs[5]), _ip(arrs[6]), C.byref(o), C.byref(s))
    if rc != 0:
        raise RuntimeError("or_solve_param failed: %d" % rc)
    return x, s


def _solve_loss(block_sizes, x0, residual_blocks, options=None):
    o = options or default_options()
    bs = np.asarray(block_sizes, dtype=np.int32)
    x = np.array(x0, dtype=np.float64).copy()
    fids = np.asarray([rb[0] for rb in residual_blocks], dtype=np.int32)
    consts, coff, pidx, poff, nodes, roots, seen = [], [], [], [0], [], [], {}
    for rb in residual_blocks:
        coff.append(len(consts))
        consts.extend(rb[1])
        pidx.extend(rb[2])
        poff.append(len(pidx))
        loss = rb[3] if len(rb) > 3 else None
        key = repr(loss)
        if key not in seen:
            seen[key] = _flatten_loss(loss, nodes)
        roots.append(seen[key])
    consts = np.asarray(consts + [0.0], dtype=np.float64)
    coff = np.asarray(coff, dtype=np.int32)
    pidx = np.asarray(pidx, dtype=np.int32)
    poff = np.asarray(poff, dtype=np.int32)
    nodes = np.asarray(nodes if nodes else [[0.0] * 5], dtype=np.float64)
    roots = np.asarray(roots, dtype=np.int32)
    s = Summary()
    rc = lib().or_solve_loss(len(bs), _ip(bs), _dp(x), len(fids), _ip(fids), _dp(consts), _ip(coff),
                             _ip(pidx), _ip(poff), _dp(nodes), _ip(roots), C.byref(o), C.byref(s))
    if rc != 0:
        raise RuntimeError("or_solve_loss failed: %d" % rc)
    return x, s


def solve_bal(C_, P_, cam_idx, pt_idx, obs, x0, options=None, loss=None, cam_mask=None, pt_mask=None):
    """cam_mask / pt_mask: per camera / point, bit k set = coordinate k held constant (0x1ff / 0x7: a constant block)."""
    o = options or default_options(linear_solver_type=DENSE_SCHUR)
    cam = np.ascontiguousarray(cam_idx, dtype=np.int32)
    pt = np.ascontiguousarray(pt_idx, dtype=np.int32)
    ob = np.ascontiguousarray(obs, dtype=np.float64)
    x = np.array(x0, dtype=np.float64).copy()
    s = Summary()
    nodes = []
    root = _flatten_loss(loss, nodes)
    arr = np.asarray(nodes if nodes else [[0.0] * 5], dtype=np.float64)
    null = C.POINTER(C.c_int)()
    cm = np.ascontiguousarray(cam_mask, dtype=np.int32) if cam_mask is not None else None
    pm = np.ascontiguousarray(pt_mask, dtype=np.int32) if pt_mask is not None else None
    rc = lib().or_solve_bal_masks(C_, P_, len(cam), _ip(cam), _ip(pt), _dp(ob), _dp(arr), root, _ip(cm) if cm is not None else null,
                                  _ip(pm) if pm is not None else null, _dp(x), C.byref(o), C.byref(s))
    if rc != 0:
        raise RuntimeError("or_solve_bal failed: %d" % rc)
    return x, s


def bal_evaluate(C_, P_, cam_idx, pt_idx, obs, x, jacobians=True):
    cam = np.ascontiguousarray(cam_idx, dtype=np.int32)
    pt = np.ascontiguousarray(pt_idx, dtype=np.int32)
    ob = np.ascontiguousarray(obs, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(cam)
    r = np.zeros(2 * n)
    F = np.zeros(18 * n) if jacobians else None
    E = np.zeros(6 * n) if jacobians else None
    cost = C.c_double()
    null = C.POINTER(C.c_double)()
    ok = lib().or_bal_evaluate(C_, P_, n, _ip(cam), _ip(pt), _dp(ob), _dp(x), _dp(r),
                               _dp(F) if jacobians else null, _dp(E) if jacobians else null,
                               C.byref(cost))
    if not ok:
        raise RuntimeError("or_bal_evaluate failed")
    if jacobians:
        return r.reshape(n, 2), F.reshape(n, 2, 9), E.reshape(n, 2, 3), cost.value
    return r.reshape(n, 2), None, None, cost.value


def dense_rows_cost(consts, x, num_threads=0):
    """1/2 sum r^2 of the dense rows (functor SYNTH_TANH_ROW; consts [m, 3]) at x — cost-only, threaded, no Jacobian."""
    consts = np.ascontiguousarray(consts, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = C.c_double(0.0)
    rc = lib().or_dense_rows_cost(_dp(consts), consts.shape[0], _dp(x), x.shape[0], num_threads, C.byref(out))
    if rc:
        raise RuntimeError("dense-row evaluation failed")
    return out.value


def cholesky_lower(A, num_threads=1, last_row=None):
    """last_row: column envelope (n ints, non-decreasing, last_row[j] >= j): the entries below it are structural zeros."""
    L = np.array(A, dtype=np.float64, order="C").copy()
    if last_row is None:
        info = lib().or_cholesky_lower(_dp(L), L.shape[0], num_threads)
    else:
        lr = np.ascontiguousarray(last_row, dtype=np.int32)
        info = lib().or_cholesky_lower_envelope(_dp(L), L.shape[0], num_threads, _ip(lr))
    return info, np.tril(L)


def cholesky_solve(L, b, last_row=None):
    """Solve L L^T y = b (L lower, row-major); with last_row the structural zeros left of each row's envelope are skipped."""
    L = np.ascontiguousarray(L, dtype=np.float64)
    y = np.array(b, dtype=np.float64).copy()
    if last_row is None:
        lib().or_cholesky_solve(_dp(L), L.shape[0], _dp(y))
    else:
        lr = np.ascontiguousarray(last_row, dtype=np.int32)
        lib().or_cholesky_solve_envelope(_dp(L), L.shape[0], _dp(y), _ip(lr))
    return y


def bal_reduced_system(C_, P_, cam_idx, pt_idx, obs, x, D, add_Dc=True):
    """Reduced camera system (S lower, rhs) of the given observations."""
    cam = np.ascontiguousarray(cam_idx, dtype=np.int32)
    pt = np.ascontiguousarray(pt_idx, dtype=np.int32)
    ob = np.ascontiguousarray(obs, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = 9 * C_
    S = np.zeros((n, n))
    rhs = np.zeros(n)
    rc = lib().or_bal_reduced_system(C_, P_, len(cam), _ip(cam), _ip(pt), _dp(ob), _dp(x), _dp(D), int(add_Dc), _dp(S), _dp(rhs))
    if rc != 0:
        raise RuntimeError("or_bal_reduced_system failed: %d" % rc)
    return S, rhs


ROTATION_IN = (3, 4, 9, 9, 3, 3, 4, 4, 7, 7, 8, 6, 6, 6)
ROTATION_OUT = (4, 3, 4, 3, 9, 9, 9, 9, 3, 3, 4, 3, 1, 3)


def rotation_apply(op, values, row_major=False, jet_dim=0):
    """Rotation op `op` (ids of include/skeres_amd.h: sk_rotation_op) on an [n, in_len] array
    (jet_dim K > 0: [n, in_len, 1 + K]); returns [n, out_len(, 1 + K)]."""
    a = np.ascontiguousarray(values, dtype=np.float64)
    n = a.shape[0]
    shape = (n, ROTATION_OUT[op]) + ((1 + jet_dim,) if jet_dim else ())
    out = np.zeros(shape)
    rc = lib().or_rotation_apply(op, int(row_major), jet_dim, _dp(a), n, _dp(out))
    if rc != 0:
        raise ValueError("or_rotation_apply failed: %d" % rc)
    return out
