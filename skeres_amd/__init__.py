"""skeres_amd — MI355X-native Levenberg-Marquardt behind the skeres API.

Python host-side mirror of the reference's public names (package
``org.somelightprojections.skeres`` + SWIG module ``com.google.ceres``) over the
C ABI of ``libskeres_amd.so`` (include/skeres_amd.h):

    AutoDiffCostFunctor / toAutoDiffCostFunction   CORE/CostFunctor.scala:40-51
    SizedCostFunction / CostFunction.evaluate      CORE/SizedCostFunction.scala:6-14
    NumericDiffCostFunctor / NumericDiffCostFunction   CORE/NumericDiffCostFunction.scala:69-162 (host side)
    Problem.addResidualBlock                       CORE/Problem.scala:20-27
    DoubleArray, RichDoubleArray, RichDoubleMatrix CORE/RichDoubleArray.scala, RichDoubleMatrix.scala
    PredefinedLossFunctions (trivial, huber, softLOne, cauchy, tukey, tolerant, composed, scaled)  ceres.i:159-184
    Solver.Options / Solver.Summary / ceres.solve  EX/SimpleBundleAdjuster.scala:147-154

There is no CPU fallback: every compute call needs the HIP library and a gfx950
device and raises otherwise.
"""
from .api import (  # noqa: F401
    SkeresError, lib, device_count,
    DoubleArray, RichDoubleArray, RichDoubleMatrix, StdVectorDoublePointer,
    CostFunction, SizedCostFunction, CostFunctor, AutoDiffCostFunctor, AutoDiffCostFunction,
    NumericDiffCostFunctor, NumericDiffCostFunction, NumericDiffMethodType, NumericDiffOptions,
    HostAutoDiffCostFunctor, HostAutoDiffCostFunction, TracedCostFunctor, TracedCostFunction, CostFunctorAdapter, CostFunctionToFunctor, DynamicCostFunctionToFunctor,
    SnavelyReprojectionError, ExponentialResidual, PowellF1, PowellF2, PowellF3, PowellF4,
    BinaryScalarCost, BinaryVector3Cost, TenParameterCost, HelloCostFunctor, QuaternionRotationError,
    LocalParameterization, PredefinedLocalParameterizations,
    PredefinedLossFunctions, LossFunction, Problem, Solver, LinearSolverType, MinimizerType, TerminationType,
    ceres, StepSolver,
)
from . import bal  # noqa: F401
from . import rotation  # noqa: F401
from . import tape  # noqa: F401
from .rotation import Rotation, Jet, Quaternion, MatrixAdapter, RowMajorMatrixAdapter3x3, ColumnMajorMatrixAdapter3x3  # noqa: F401
