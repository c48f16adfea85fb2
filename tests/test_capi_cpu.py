"""CPU-only checks of the C ABI and the host logic (no compute calls: no GPU here).

* the library loads and exports every symbol include/skeres_amd.h declares;
* native-memory helpers (ports of DoubleArraySliceSpec / RichDoubleArraySpec /
  RichDoubleMatrixSpec of the reference);
* argument validation (the Scala side's `require`s) and Problem bookkeeping;
* BAL text format round trip; compute entry points fail LOUDLY without a device.
"""
import os
import re

import numpy as np
import pytest

import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    return built


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "skeres_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", header))
    declared -= {"sk_evaluate_fn", "sk_allreduce_fn"}
    assert len(declared) > 80
    L = sk.lib()
    missing = [name for name in sorted(declared) if not hasattr(L, name)]
    assert not missing, missing
    # and the Python binding covers all of them
    assert set(sk.api.exported_symbols()) == declared


def test_version_and_device_count():
    assert b"skeres_amd" in sk.lib().sk_version()
    assert sk.device_count() >= 0


# ---- DoubleArraySliceSpec.scala:8-23, RichDoubleArraySpec.scala:8-33 ----------------
def test_double_array_slice():
    a = sk.DoubleArray(10)
    for i in range(10):
        a.setitem(i, float(i))
    s = a.slice(3)
    assert [s.get(i) for i in range(7)] == [3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0]
    s.set(0, -1.0)
    assert a.get(3) == -1.0  # a slice is a view, not a copy


def test_rich_double_array_round_trips():
    a = sk.RichDoubleArray.fromArray([1.0, 2.0, 3.0])
    assert list(a.toArray(3)) == [1.0, 2.0, 3.0]
    b = sk.RichDoubleArray.ofSize(3)
    b.copyFrom([4.0, 5.0, 6.0])
    assert [b.get(i) for i in range(3)] == [4.0, 5.0, 6.0]
    assert not a.isNull()


# ---- RichDoubleMatrixSpec.scala:8-94 --------------------------------------------------
def test_rich_double_matrix():
    m = sk.RichDoubleMatrix.ofSize(3, 4)
    for i in range(3):
        for j in range(4):
            m.set(i, j, 10 * i + j)
    assert m.get(2, 3) == 23 and m.hasRow(1) and not m.isNull()
    assert list(m.getRow(1).toArray(4)) == [10, 11, 12, 13]
    m2 = sk.RichDoubleMatrix.fromArrays([1.0, 2.0], [3.0, 4.0, 5.0])
    assert m2.get(1, 2) == 5.0
    v = sk.StdVectorDoublePointer()
    assert v.size() == 0 and not v.toPointerPointer()  # DoubleMatrix.toPointerPointer: NULL when empty (ceres.i:121-123)
    v.add(m.getRow(0))
    assert v.size() == 1 and bool(v.toPointerPointer())


# ---- validation ---------------------------------------------------------------------------
def test_cost_functor_validation():
    with pytest.raises(ValueError):  # CORE/CostFunctor.scala:32
        sk.AutoDiffCostFunctor(0, 1)
    with pytest.raises(ValueError):  # CORE/CostFunctor.scala:33
        sk.AutoDiffCostFunctor(1, 2, 0)
    with pytest.raises(ValueError):  # CORE/SizedCostFunction.scala:7
        sk.SizedCostFunction(1, -1)
    with pytest.raises(ValueError):  # CORE/SizedCostFunction.scala:8-11
        sk.SizedCostFunction(1, 0, 2)
    cf = sk.SnavelyReprojectionError(1.0, 2.0).toAutoDiffCostFunction()
    assert cf.numResiduals() == 2 and cf.parameterBlockSizes() == [9, 3]


def test_problem_bookkeeping_and_errors():
    m, c, bad = sk.DoubleArray(1), sk.DoubleArray(1), sk.DoubleArray(9)
    loss = sk.PredefinedLossFunctions.trivialLoss()
    p = sk.Problem()
    rid0 = p.addResidualBlock(sk.ExponentialResidual(0.0, 1.0).toAutoDiffCostFunction(), loss, m, c)
    rid1 = p.addResidualBlock(sk.ExponentialResidual(1.0, 2.0).toAutoDiffCostFunction(), loss, m, c)
    assert (rid0, rid1) == (0, 1)
    assert (p.numResidualBlocks(), p.numParameterBlocks(), p.numParameters(), p.numResiduals()) == (2, 2, 2, 2)
    with pytest.raises(ValueError):  # wrong number of parameter blocks
        p.addResidualBlock(sk.ExponentialResidual(0.0, 1.0).toAutoDiffCostFunction(), loss, m)
    with pytest.raises(ValueError):  # same pointer, different size
        p.addResidualBlock(sk.SnavelyReprojectionError(0.0, 0.0).toAutoDiffCostFunction(), loss, bad, m)
    with pytest.raises(ValueError):  # duplicate parameter block inside one residual block
        p.addResidualBlock(sk.ExponentialResidual(0.0, 1.0).toAutoDiffCostFunction(), loss, m, m)


def test_options_validation():
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(25)
    o.setMinimizerProgressToStdout(True)
    with pytest.raises(sk.SkeresError):
        o.setLinearSolverType(sk.LinearSolverType.SPARSE_NORMAL_CHOLESKY)
    with pytest.raises(sk.SkeresError):
        o.setMinimizerType(sk.MinimizerType.LINE_SEARCH)
    with pytest.raises(ValueError):
        o.setMaxNumIterations(-1)
    with pytest.raises(ValueError):
        o.setFunctionTolerance(-1.0)
    # the library's own options of the reduced system: modes are checked
    for setter in (o.setCholeskyBorder, o.setRetainedPoints):
        for mode in ("auto", "on", "off"):
            setter(mode)
        with pytest.raises(sk.SkeresError):
            setter(7)
    with pytest.raises(sk.SkeresError):
        o.setRetainedPoints("on", -3)


def test_reduce_buffer_has_room_for_the_pseudo_cameras_of_retained_points():
    """sk_reduce_buffer_bytes (what skeres_amd/dist.py allocates for the all-reduce of the reduced system): the packed lower block
    triangle of the cameras' system — and, unless retained points are off, of as many nine-row pseudo-cameras as the options allow."""
    from skeres_amd import api, bal
    from helpers import bal_problem_to_sk
    prob = bal.generate(40, 300, 1500, seed=3)
    problem, params, loss = bal_problem_to_sk(prob)

    def nbytes(mode, max_points=0):
        o = sk.Solver.Options()
        o.setRetainedPoints(mode, max_points)
        return api.lib().sk_reduce_buffer_bytes(o._h, problem._h)

    def tri(cams):
        nblk = (9 * cams + 1 + 127) // 128
        return 128 * 128 * 8 * nblk * (nblk + 1) // 2
    assert nbytes("off") == tri(40)
    # AUTO retains nothing below 64 cameras: the plain size (it used to add 512 pseudo-cameras whatever the problem);
    # ON with a count: that many points (as far as the problem has wide tracks), three to a pseudo-camera — what the plan
    # sk_problem_retained_plan reports, from host data alone
    assert nbytes("auto") == tri(40)
    kept = problem.retainedPlan("on", 30)["retained_points"]
    assert 3 <= kept <= 30 and nbytes("on", 30) == tri(40 + (kept + 2) // 3)


def test_predefined_loss_functions_construction_and_validation():
    # ceres.i:159-184: constructors exist for every predefined loss; bad parameters and over-deep nesting are errors.
    # (No evaluation here: LossFunction.evaluate runs on the device.)
    L = sk.PredefinedLossFunctions
    losses = [L.trivialLoss(), L.huberLoss(1.0), L.softLOneLoss(0.5), L.cauchyLoss(0.5), L.tukeyLoss(2.0), L.tolerantLoss(1.0, 0.3)]
    losses.append(L.composedLoss(losses[1], losses[3]))
    losses.append(L.scaledLoss(losses[2], 0.5))
    losses.append(L.scaledLoss(None, 2.0))
    assert all(l._h for l in losses)
    for bad in (lambda: L.huberLoss(0.0), lambda: L.cauchyLoss(-1.0), lambda: L.tolerantLoss(1.0, 0.0)):
        with pytest.raises(sk.SkeresError):
            bad()
    deep = L.cauchyLoss(1.0)
    for _ in range(4):
        deep = L.composedLoss(deep, L.trivialLoss())
    with pytest.raises(sk.SkeresError, match="nested deeper"):
        L.composedLoss(deep, None)
    # a problem accepts any of them per residual block; dense rows take ONE for all their rows (refused at solve time otherwise)
    m, c = sk.DoubleArray(1), sk.DoubleArray(1)
    problem = sk.Problem()
    for l in losses:
        problem.addResidualBlock(sk.ExponentialResidual(1.0, 2.0).toAutoDiffCostFunction(), l, m, c)
    assert problem.numResidualBlocks() == len(losses)
    x = sk.DoubleArray(8)
    rows = sk.Problem()
    rows.addDenseRows(10, np.zeros((4, 3)), L.huberLoss(1.0), x, 8)
    assert rows.numResidualBlocks() == 4
    if sk.device_count() == 0:
        with pytest.raises(sk.SkeresError, match="no HIP device"):
            L.huberLoss(1.0).evaluate([1.0])


def test_compute_fails_loudly_without_a_device():
    if sk.device_count() > 0:
        pytest.skip("a GPU is present")
    prob = bal.generate(4, 24, 90, seed=42)
    problem, params, loss = bal_problem_to_sk(prob)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    with pytest.raises(sk.SkeresError, match="no HIP device"):
        sk.ceres.solve(o, problem, sk.Solver.Summary())
    assert np.array_equal(params.toArray(prob.num_parameters), prob.parameters)  # untouched: no silent CPU path
    with pytest.raises(sk.SkeresError, match="no HIP device"):
        sk.api.cholesky_solve(np.eye(4), np.ones(4))
    cf = sk.BinaryScalarCost(1.0).toAutoDiffCostFunction()
    with pytest.raises(sk.SkeresError, match="no HIP device"):
        cf.evaluate(sk.RichDoubleMatrix.ofSize(2, 2), sk.DoubleArray(1), None)


def test_dense_schur_rejects_non_bundle_shapes():
    m, c = sk.DoubleArray(1), sk.DoubleArray(1)
    p = sk.Problem()
    p.addResidualBlock(sk.ExponentialResidual(0.0, 1.0).toAutoDiffCostFunction(), None, m, c)
    with pytest.raises(sk.SkeresError, match="DENSE_SCHUR"):
        p.pointPartition(2)


# ---- host logic: sharding ------------------------------------------------------------------
def test_point_partition_covers_every_point_once_and_balances_squares():
    prob = bal.generate(30, 2000, 9500, seed=9)
    problem, _, _ = bal_problem_to_sk(prob)
    for world in (1, 2, 3, 8):
        cuts, nc, npts, pob = problem.pointPartition(world)
        assert nc == 30 and npts == 2000
        assert cuts[0] == 0 and cuts[-1] == npts and np.all(np.diff(cuts) >= 0)
        k2 = np.bincount(pob, minlength=npts).astype(np.int64) ** 2
        loads = [k2[cuts[r]:cuts[r + 1]].sum() for r in range(world)]
        assert sum(loads) == k2.sum()
        assert max(loads) <= k2.sum() / world + k2.max()  # contiguous cut: within one point of ideal


# ---- BAL text format (SimpleBundleAdjuster.scala:37-76) ---------------------------------
def test_bal_text_round_trip(tmp_path):
    prob = bal.generate(3, 12, 30, seed=1)
    path = tmp_path / "problem.txt"
    prob.to_file(str(path))
    back = bal.BalProblem.from_file(str(path))
    assert (back.num_cameras, back.num_points, back.num_observations) == (3, 12, 30)
    assert np.array_equal(back.camera_index, prob.camera_index) and np.array_equal(back.point_index, prob.point_index)
    np.testing.assert_array_equal(back.observations, prob.observations)
    np.testing.assert_array_equal(back.parameters, prob.parameters)
    first = open(path).readline().split()
    assert first == ["3", "12", "30"]


def test_generator_shapes_are_exact_and_visible():
    for name in ("problem-49-7776",):
        C, P, N = bal.SHAPES[name]
        prob = bal.generate_named(name)
        assert (prob.num_cameras, prob.num_points, prob.num_observations) == (C, P, N)
        k = np.bincount(prob.point_index, minlength=P)
        assert k.min() >= 2
        # no duplicate (camera, point) observation
        key = prob.camera_index.astype(np.int64) * P + prob.point_index
        assert np.unique(key).size == N
        _, depth = bal.snavely_project(prob.cameras()[prob.camera_index], prob.points()[prob.point_index])
        assert (depth < 0).all()  # in front of the camera (Snavely sign convention)


def test_header_is_valid_c99_and_the_signatures_link(tmp_path):
    """tests/abi_smoke.c: include/skeres_amd.h compiled as C99 (-pedantic -Werror) and linked against libskeres_amd.so;
    memory helpers, Problem bookkeeping, option validation, and sk_solve -> SK_ERR_NO_DEVICE without a GPU (with one,
    the tiny problem is solved).  What a JNI shim generated from the header would bind (ceres.i:95-125)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "skeres_amd")
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-lskeres_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + rocm_lib])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = rocm_lib + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout


def test_per_device_queue_table_logic(tmp_path):
    """tests/device_table_test.cpp: the table that owns the factorisation's queues per device (csrc/device_table.hpp) —
    one entry per device, created once under concurrent requests, stable addresses, failed creations retried."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "device_table_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "device_table_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "device_table_test ok" in out.stdout, out.stdout + out.stderr


def test_failures_through_a_null_handle_carry_a_typed_status():
    # sk_solver_create can only return NULL; the status is recorded (sk_last_status), never parsed from the message
    prob = bal.generate(4, 20, 80, seed=1)
    problem, params, loss = bal_problem_to_sk(prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    if sk.device_count() == 0:
        with pytest.raises(sk.SkeresError, match="status 2"):  # SK_ERR_NO_DEVICE
            sk.StepSolver(options, problem)
        assert sk.lib().sk_last_status() == 2
    # a configuration that is refused before any device is touched: dense rows under DENSE_QR
    x = sk.DoubleArray(4)
    p2 = sk.Problem()
    p2.addDenseRows(10, np.zeros((8, 3)), None, x, 4)
    o2 = sk.Solver.Options()
    o2.setLinearSolverType(sk.LinearSolverType.DENSE_QR)
    with pytest.raises(sk.SkeresError, match="status 4"):  # SK_ERR_UNSUPPORTED
        sk.StepSolver(o2, p2)
    assert sk.lib().sk_last_status() == 4
    summary = sk.Solver.Summary()
    with pytest.raises(sk.SkeresError, match="status 4"):
        sk.ceres.solve(o2, p2, summary)


def test_example_data_tables_are_the_fixture_tables():
    # the examples ship their own copy of the reference's sample tables (CurveFitting.scala:22-90,
    # RobustCurveFitting.scala:21-90); product code does not reach into tests/
    for name in ("curve_fitting_data.txt", "robust_curve_fitting_data.txt"):
        a = open(os.path.join(ROOT, "tests", "golden", name)).read()
        b = open(os.path.join(ROOT, "skeres_amd", "examples", "data", name)).read()
        assert a == b
    for f in ("curve_fitting.py", "robust_curve_fitting.py"):
        assert "tests" not in open(os.path.join(ROOT, "skeres_amd", "examples", f)).read().split("_DATA")[1].split("\n")[0]
