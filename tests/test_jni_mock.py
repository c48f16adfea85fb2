"""bindings/jni/skeres_amd_jni.c EXECUTED, without a JVM: the thunks are compiled against tests/jni_stub/jni.h together with
tests/jni_stub/mock_jvm.c — a functional mock of the JNIEnv / JavaVM members they use (arrays that are copied on Get and written
back on Release unless JNI_ABORT, a recorded pending exception, a "Java object" whose evaluateNative is a callback, counters for
pinned arrays, global references and calls made with an exception pending) — linked to libskeres_amd.so and driven through
ctypes.  It is a MOCK: it pins the logic of the thunks (marshalling, release modes, status -> exception, the director
trampoline), nothing about a real JVM.  CPU tests cover what needs no device; `-m gpu` tests solve through the thunks
against the oracle (a 16-camera bundle adjustment; the reference's curve fitting through the director trampoline)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K_INT, K_LONG, K_DOUBLE = 3, 4, 5
EVALUATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64)


class Jni:
    """The mock library: `call("skArrayNew", n)` runs Java_com_google_ceres_SkeresNative_skArrayNew(env, NULL, n)."""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        L = self.lib
        L.mock_env.restype = C.c_void_p
        L.mock_new_array.restype = C.c_void_p
        L.mock_new_array.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mock_array_data.restype = C.c_void_p
        L.mock_array_data.argtypes = [C.c_void_p]
        L.mock_new_string.restype = C.c_void_p
        L.mock_new_string.argtypes = [C.c_char_p]
        L.mock_string_chars.restype = C.c_char_p
        L.mock_string_chars.argtypes = [C.c_void_p]
        L.mock_new_director_object.restype = C.c_void_p
        L.mock_new_director_object.argtypes = [EVALUATE_FN, C.c_void_p]
        L.mock_new_plain_object.restype = C.c_void_p
        L.mock_object_global_refs.argtypes = [C.c_void_p]
        L.mock_exception_class.restype = C.c_char_p
        L.mock_exception_message.restype = C.c_char_p
        self.env = C.c_void_p(L.mock_env())
        assert L.mock_load() == 0x00010006
        self.keep = []

    def call(self, name, *args, restype=C.c_int64):
        fn = getattr(self.lib, "Java_com_google_ceres_SkeresNative_" + name)
        fn.restype = restype
        conv = []
        for a in args:
            if isinstance(a, float):
                conv.append(C.c_double(a))
            elif isinstance(a, (int, np.integer)):
                conv.append(C.c_int64(int(a)))
            else:
                conv.append(a)
        return fn(self.env, None, *conv)

    def i32(self, *args):  # (jint arguments are passed in 64-bit registers either way; kept for readability)
        return args

    def array(self, kind, values):
        dtype = {K_INT: np.int32, K_LONG: np.int64, K_DOUBLE: np.float64}[kind]
        v = np.ascontiguousarray(values, dtype=dtype)
        h = C.c_void_p(self.lib.mock_new_array(kind, v.size, v.ctypes.data_as(C.c_void_p)))
        return h

    def array_values(self, h, kind, n):
        dtype = {K_INT: np.int32, K_LONG: np.int64, K_DOUBLE: np.float64}[kind]
        p = self.lib.mock_array_data(h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()

    def string(self, h):
        return self.lib.mock_string_chars(C.c_void_p(h)).decode()

    def exception(self):
        if not self.lib.mock_exception_pending():
            return None
        e = (self.lib.mock_exception_class().decode(), self.lib.mock_exception_message().decode())
        self.lib.mock_exception_clear()
        return e

    def clean(self):
        """nothing left pinned, attached or referenced, no call made with an exception pending"""
        L = self.lib
        return (L.mock_pinned(), L.mock_attached(), L.mock_jni_violations()) == (0, 0, 0)


@pytest.fixture(scope="module")
def jni(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("jni") / "libskeres_amd_jni_mock.so")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "bindings", "jni", "skeres_amd_jni.c"), os.path.join(ROOT, "tests", "jni_stub", "mock_jvm.c"),
           "-L", os.path.join(ROOT, "skeres_amd"), "-lskeres_amd", "-Wl,-rpath," + os.path.join(ROOT, "skeres_amd"), "-o", out]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return Jni(out)


def test_native_arrays_slices_matrices_and_pointer_vectors_through_the_thunks(jni):
    """DoubleArray / DoubleArraySlice / DoubleMatrix / StdVectorDoublePointer as the Scala layer calls them (ceres.i:79-125):
    TEST/DoubleArraySliceSpec.scala:8-23 and TEST/RichDoubleMatrixSpec.scala:37-65, restated over the thunks."""
    a = jni.call("skArrayNew", 10)
    assert a and jni.exception() is None
    for i in range(10):
        jni.call("skArraySetitem", a, i, float(i), restype=None)
    s = jni.call("skArraySlice", a, 4)
    assert s == a + 4 * 8
    for i in range(6):
        assert jni.call("skArrayGetitem", s, i, restype=C.c_double) == 4 + i
        jni.call("skArraySetitem", s, i, 10.0 * (4 + i), restype=None)
    assert [jni.call("skArrayGetitem", a, i, restype=C.c_double) for i in range(10)] == [0, 1, 2, 3, 40, 50, 60, 70, 80, 90]
    # bulk copies: one crossing each way
    src = jni.array(K_DOUBLE, np.arange(10) * 3.0)
    jni.call("skArrayCopyIn", a, src, 10, restype=None)
    dst = jni.array(K_DOUBLE, np.zeros(10))
    jni.call("skArrayCopyOut", a, dst, 10, restype=None)
    assert np.array_equal(jni.array_values(dst, K_DOUBLE, 10), np.arange(10) * 3.0)
    # a copy that overruns the Java array raises instead of reading past it
    jni.call("skArrayCopyIn", a, jni.array(K_DOUBLE, np.zeros(4)), 10, restype=None)
    assert jni.exception()[0] == "java/lang/ArrayIndexOutOfBoundsException"
    # std::vector<double*> and the double** it hands out
    b = jni.call("skArrayNew", 6)
    v = jni.call("skPtrvecNew")
    jni.call("skPtrvecAdd", v, a, restype=None)
    jni.call("skPtrvecAdd", v, 0, restype=None)
    jni.call("skPtrvecSet", v, 1, b, restype=None)
    assert jni.call("skPtrvecSize", v, restype=C.c_int) == 2 and jni.call("skPtrvecGet", v, 1) == b
    m = jni.call("skPtrvecToPointerPointer", v)
    assert not jni.call("skMatrixIsNull", m, restype=C.c_ubyte) and jni.call("skMatrixIsNull", 0, restype=C.c_ubyte)
    assert jni.call("skMatrixRow", m, 0) == a and jni.call("skMatrixRow", m, 1) == b
    jni.call("skPtrvecFree", v, restype=None)
    jni.call("skArrayFree", a, restype=None)
    jni.call("skArrayFree", b, restype=None)
    assert jni.clean()


def test_status_codes_become_java_exceptions(jni):
    """SK_ERR_INVALID_ARGUMENT -> IllegalArgumentException (what the Scala `require`s throw, CORE/CostFunctor.scala:32-33), anything
    else -> RuntimeException, both with sk_last_error() as the message; a NULL handle throws as well."""
    o = jni.call("skOptionsNew")
    assert jni.call("skOptionsSetMaxNumIterations", o, 7, restype=C.c_int) == 0 and jni.exception() is None
    assert jni.call("skOptionsSetLinearSolverType", o, 99, restype=C.c_int) != 0
    cls, msg = jni.exception()
    assert cls == "java/lang/RuntimeException" and "not implemented" in msg  # (SK_ERR_UNSUPPORTED)
    assert jni.call("skOptionsSetCholeskyBorder", o, 1, restype=C.c_int) == 0 and jni.call("skOptionsSetCholeskyBorder", o, 5, restype=C.c_int) != 0
    assert jni.call("skOptionsSetRetainedPoints", o, 1, 12, restype=C.c_int) == 0 and jni.call("skOptionsSetRetainedPoints", o, 7, 0, restype=C.c_int) != 0
    assert jni.exception()[0] == "java/lang/IllegalArgumentException"
    # an unknown device functor: a NULL handle and an exception
    assert jni.call("skCostFunctionNewAutodiff", 12345, jni.array(K_DOUBLE, [1.0])) == 0
    assert jni.exception() is not None
    # a subset parameterization with an index out of range
    assert jni.call("skLocalParameterizationSubset", 3, jni.array(K_INT, [0, 7])) == 0
    assert jni.exception()[0] == "java/lang/IllegalArgumentException"
    good = jni.call("skLocalParameterizationSubset", 9, jni.array(K_INT, [6, 7, 8]))
    assert good and jni.exception() is None
    jni.call("skLocalParameterizationFree", good, restype=None)
    jni.call("skOptionsFree", o, restype=None)
    assert jni.string(jni.call("skVersion")) and jni.clean()


def _build_curve_fitting_through_the_director(jni, evaluate):
    """EX/CurveFitting.scala:100-117 with every block a JVM cost function (SizedCostFunction -> skDirectorNew +
    skCostFunctionNewCallback): returns (problem, m, c, handles to free)."""
    from helpers import curve_fitting_data
    m, c = jni.call("skArrayNew", 1), jni.call("skArrayNew", 1)
    for p in (m, c):
        jni.call("skArraySetitem", p, 0, 0.0, restype=None)
    problem = jni.call("skProblemNew")
    loss = jni.call("skLossTrivial")
    sizes = jni.array(K_INT, [1, 1])
    costs = []
    for k, (x, y) in enumerate(curve_fitting_data()):
        cb = EVALUATE_FN(lambda user, pp, rr, jj, x=x, y=y: evaluate(x, y, pp, rr, jj))
        jni.keep.append(cb)
        obj = C.c_void_p(jni.lib.mock_new_director_object(cb, None))
        director = jni.call("skDirectorNew", obj)
        cost = jni.call("skCostFunctionNewCallback", director, 1, sizes)
        assert director and cost and jni.lib.mock_object_global_refs(obj) == 1
        v = jni.call("skPtrvecNew")
        jni.call("skPtrvecAdd", v, m, restype=None)
        jni.call("skPtrvecAdd", v, c, restype=None)
        jni.call("skProblemAddResidualBlock", problem, cost, loss, v)
        jni.call("skPtrvecFree", v, restype=None)
        costs.append((cost, director, obj))
    assert jni.exception() is None
    assert jni.call("skProblemNumResidualBlocks", problem, restype=C.c_int) == 67 and jni.call("skProblemNumParameterBlocks", problem, restype=C.c_int) == 2
    return problem, m, c, loss, costs


def _exponential_residual(x, y, pp, rr, jj):
    """y - exp(m x + c) and its derivatives, written into the native buffers the solver hands over (EX/CurveFitting.scala:92-98
    through CORE/AutodiffCostFunction.scala:74-134: `jacobians` and each of its rows may be null)."""
    params = C.cast(pp, C.POINTER(C.POINTER(C.c_double)))
    e = np.exp(params[0][0] * x + params[1][0])
    C.cast(rr, C.POINTER(C.c_double))[0] = y - e
    if jj:
        rows = C.cast(jj, C.POINTER(C.POINTER(C.c_double)))
        if rows[0]:
            rows[0][0] = -x * e
        if rows[1]:
            rows[1][0] = -e
    return 1


def test_without_a_device_the_solve_throws_and_leaves_nothing_pinned(jni):
    import skeres_amd as sk
    if sk.device_count() > 0:
        pytest.skip("a device is present: the solve runs (see the gpu tests)")
    problem, m, c, loss, costs = _build_curve_fitting_through_the_director(jni, _exponential_residual)
    o, s = jni.call("skOptionsNew"), jni.call("skSummaryNew")
    assert jni.call("skSolve", o, problem, s, restype=C.c_int) != 0
    cls, msg = jni.exception()
    assert cls == "java/lang/RuntimeException" and "no HIP device" in msg
    for cost, director, obj in costs:
        jni.call("skCostFunctionFree", cost, restype=None)
        jni.call("skDirectorFree", director, restype=None)
        assert jni.lib.mock_object_global_refs(obj) == 0
    assert jni.lib.mock_global_refs() == 0 and jni.clean()


def test_director_needs_the_evaluate_method(jni):
    obj = C.c_void_p(jni.lib.mock_new_plain_object())
    assert jni.call("skDirectorNew", obj) == 0
    assert jni.exception()[0] == "java/lang/NoSuchMethodError"
    assert jni.lib.mock_global_refs() == 0


# ---- with a device: solves through the thunks, against the oracle -------------------------------------------------------------
@pytest.mark.gpu
def test_bundle_adjustment_through_the_thunks_vs_oracle(jni):
    """EX/SimpleBundleAdjuster.scala:126-155 as the Scala layer of bindings/ drives it: one native array (skArrayNew + skArrayCopyIn),
    the set-up loop in one crossing (skProblemAddResidualBlocks with element offsets), DENSE_SCHUR, skSolve, the summary's
    getters and strings, skArrayCopyOut — a 16-camera problem against the oracle's trajectory."""
    import oracle
    import skeres_amd as sk
    from skeres_amd import bal
    if sk.device_count() < 1:
        pytest.fail("GPU tests need a HIP device")
    prob = bal.generate(16, 600, 2600, seed=11)
    n = prob.num_parameters
    base = jni.call("skArrayNew", n)
    jni.call("skArrayCopyIn", base, jni.array(K_DOUBLE, prob.parameters), n, restype=None)
    problem, loss = jni.call("skProblemNew"), jni.call("skLossTrivial")
    offs = np.stack([9 * prob.camera_index.astype(np.int64), 9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    rc = jni.call("skProblemAddResidualBlocks", problem, 1, prob.num_observations, jni.array(K_DOUBLE, prob.observations.ravel()), loss, base,
                  jni.array(K_LONG, offs.ravel()), restype=C.c_int)
    assert rc == 0 and jni.exception() is None
    o, s = jni.call("skOptionsNew"), jni.call("skSummaryNew")
    assert jni.call("skOptionsSetLinearSolverType", o, 3, restype=C.c_int) == 0  # DENSE_SCHUR
    assert jni.call("skSolve", o, problem, s, restype=C.c_int) == 0 and jni.exception() is None
    out = jni.array(K_DOUBLE, np.zeros(n))
    jni.call("skArrayCopyOut", base, out, n, restype=None)
    x_gpu = jni.array_values(out, K_DOUBLE, n)
    x_cpu, so = oracle.solve_bal(prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index, prob.observations, prob.parameters,
                                 oracle.default_options(linear_solver_type=oracle.DENSE_SCHUR))
    assert abs(jni.call("skSummaryInitialCost", s, restype=C.c_double) - so.initial_cost) <= 1e-10 * so.initial_cost
    assert abs(jni.call("skSummaryFinalCost", s, restype=C.c_double) - so.final_cost) <= 1e-9 * so.final_cost
    costs = so.costs()
    for k in range(min(5, len(costs))):
        assert abs(jni.call("skSummaryIterationField", s, k, 0, restype=C.c_double) - costs[k]) <= 1e-10 * costs[k]
    assert jni.call("skSummaryTerminationType", s, restype=C.c_int) == so.termination_type
    assert np.linalg.norm(x_gpu - x_cpu) <= 1e-6 * np.linalg.norm(x_cpu)
    assert "Ceres Solver Report" in jni.string(jni.call("skSummaryBriefReport", s)) and "DENSE_SCHUR" in jni.string(jni.call("skSummaryFullReport", s))
    for h, free in ((s, "skSummaryFree"), (o, "skOptionsFree"), (problem, "skProblemFree"), (loss, "skLossFree"), (base, "skArrayFree")):
        jni.call(free, h, restype=None)
    assert jni.clean()


@pytest.mark.gpu
@pytest.mark.parametrize("attached", [True, False])
def test_curve_fitting_through_the_director_trampoline_vs_oracle(jni, attached):
    """The reference's director path (ceres.i:48): every residual block of EX/CurveFitting.scala a JVM cost function whose
    evaluate the native solver calls back through jvm_evaluate — here the mock's callback — with the exact native signature;
    DENSE_QR, 25 iterations, against the oracle.  attached=False: the up-call arrives on a thread the JVM does not know
    (GetEnv fails): the trampoline attaches and detaches around it."""
    import oracle
    from helpers import curve_fitting_data
    jni.lib.mock_set_thread_attached(1 if attached else 0)
    try:
        problem, m, c, loss, costs = _build_curve_fitting_through_the_director(jni, _exponential_residual)
        o, s = jni.call("skOptionsNew"), jni.call("skSummaryNew")
        jni.call("skOptionsSetMaxNumIterations", o, 25, restype=C.c_int)
        jni.call("skOptionsSetLinearSolverType", o, 1, restype=C.c_int)  # DENSE_QR
        assert jni.call("skSolve", o, problem, s, restype=C.c_int) == 0 and jni.exception() is None
    finally:
        jni.lib.mock_set_thread_attached(1)
    blocks = [(2, [x, y], [0, 1]) for x, y in curve_fitting_data()]
    xo, so = oracle.solve([1, 1], [0.0, 0.0], blocks, oracle.default_options(linear_solver_type=oracle.DENSE_QR, max_num_iterations=25))
    got = [jni.call("skArrayGetitem", m, 0, restype=C.c_double), jni.call("skArrayGetitem", c, 0, restype=C.c_double)]
    np.testing.assert_allclose(got, xo, rtol=1e-7)
    assert abs(jni.call("skSummaryFinalCost", s, restype=C.c_double) - so.final_cost) <= 1e-9 * so.final_cost
    for cost, director, obj in costs:
        jni.call("skCostFunctionFree", cost, restype=None)
        jni.call("skDirectorFree", director, restype=None)
    assert jni.lib.mock_global_refs() == 0 and jni.clean()


@pytest.mark.gpu
def test_an_evaluate_that_throws_fails_the_solve_and_its_exception_survives(jni):
    """evaluate() throws on its third call: the trampoline reports the block as not evaluable, makes NO further call into the
    environment while the exception is pending (undefined behaviour under the JNI specification), and skSolve
    returns with THAT exception pending, not one of its own."""
    calls = {"n": 0}

    def evaluate(x, y, pp, rr, jj):
        calls["n"] += 1
        return -1 if calls["n"] >= 3 else _exponential_residual(x, y, pp, rr, jj)
    problem, m, c, loss, costs = _build_curve_fitting_through_the_director(jni, evaluate)
    o, s = jni.call("skOptionsNew"), jni.call("skSummaryNew")
    jni.call("skOptionsSetLinearSolverType", o, 1, restype=C.c_int)
    rc = jni.call("skSolve", o, problem, s, restype=C.c_int)
    # (like ceres::Solve, the native solve reports a cost function that cannot be evaluated at the starting point through the summary —
    # termination FAILURE — not through its status)
    assert rc != 0 or jni.call("skSummaryTerminationType", s, restype=C.c_int) == 2
    assert calls["n"] == 3  # no up-call after the one that threw
    assert jni.exception() == ("java/lang/IllegalStateException", "evaluate threw")
    assert jni.lib.mock_jni_violations() == 0
