package com.google.ceres

// The classes the reference gets from SWIG (ceres.i -> com.google.ceres.*), written by hand over SkeresNative so that
// EX/*.scala and CORE/*.scala keep their imports (`import com.google.ceres._`).  Only what the reference's sources use.
// Not compiled in this repository's image (no scalac); see bindings/README.md.

/** SWIGTYPE_p_double: a native `double*`. */
final class DoublePointer(val address: Long) extends AnyVal { def isNull: Boolean = address == 0L }
/** SWIGTYPE_p_p_double: a native `double**`. */
final class DoublePointerPointer(val address: Long) extends AnyVal { def isNull: Boolean = address == 0L }

/** carrays.i's DoubleArray (ceres.i:95-96): owns `n` native doubles unless made by `frompointer`. */
class DoubleArray private (val address: Long, owned: Boolean) {
  def this(n: Int) = this(SkeresNative.skArrayNew(n), true)
  def getitem(i: Int): Double = SkeresNative.skArrayGetitem(address, i)
  def setitem(i: Int, v: Double): Unit = SkeresNative.skArraySetitem(address, i, v)
  def cast: DoublePointer = new DoublePointer(address)
  override def finalize(): Unit = if (owned) SkeresNative.skArrayFree(address)
}
object DoubleArray { def frompointer(p: DoublePointer): DoubleArray = new DoubleArray(p.address, false) }

/** ceres.i:99-107 */
object DoubleArraySlice { def get(buffer: DoublePointer, start: Int): DoublePointer = new DoublePointer(SkeresNative.skArraySlice(buffer.address, start)) }

/** ceres.i:113-125 */
object DoubleMatrix {
  def isNull(m: DoublePointerPointer): Boolean = m.address == 0L || SkeresNative.skMatrixIsNull(m.address)
  def row(m: DoublePointerPointer, i: Int): DoublePointer = new DoublePointer(SkeresNative.skMatrixRow(m.address, i))
  def toPointerPointer(v: StdVectorDoublePointer): DoublePointerPointer = v.data
}

/** std::vector<double*> (ceres.i:82) */
class StdVectorDoublePointer {
  val handle: Long = SkeresNative.skPtrvecNew()
  /** std::vector<double*>(n): n null pointers (CORE/RichDoubleMatrix.scala:74) */
  def this(n: Int) = { this(); var i = 0; while (i < n) { SkeresNative.skPtrvecAdd(handle, 0L); i += 1 } }
  def add(p: DoublePointer): Unit = SkeresNative.skPtrvecAdd(handle, p.address)
  def set(i: Int, p: DoublePointer): Unit = SkeresNative.skPtrvecSet(handle, i, p.address)
  def get(i: Int): DoublePointer = new DoublePointer(SkeresNative.skPtrvecGet(handle, i))
  def size: Int = SkeresNative.skPtrvecSize(handle)
  /** the vector's storage as a `double**` (DoubleMatrix.toPointerPointer, ceres.i:121-123): valid while the vector lives and is not resized */
  def data: DoublePointerPointer = new DoublePointerPointer(SkeresNative.skPtrvecToPointerPointer(handle))
  override def finalize(): Unit = SkeresNative.skPtrvecFree(handle)
}

/** std::vector<int> (ceres.i:79): what CostFunction's block sizes are to the JVM — `size` is a long and elements are read with
  * `get`, as SWIG's proxy has them (CORE/CostFunctionToFunctor.scala:24-25, CORE/SizedCostFunction.scala:13). */
class StdVectorInt {
  private val values = scala.collection.mutable.ArrayBuffer.empty[Int]
  def add(x: Int): Unit = values += x
  def get(i: Int): Int = values(i)
  def set(i: Int, x: Int): Unit = values(i) = x
  def size: Long = values.length.toLong
  def isEmpty: Boolean = values.isEmpty
  def clear(): Unit = values.clear()
  def toArray: Array[Int] = values.toArray
}

object Ownership extends Enumeration { val DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP = Value }
object LinearSolverType extends Enumeration { val DENSE_NORMAL_CHOLESKY = Value(0); val DENSE_QR = Value(1); val DENSE_SCHUR = Value(3) }
object MinimizerType extends Enumeration { val LINE_SEARCH = Value(0); val TRUST_REGION = Value(1) }
/** ceres::NumericDiffMethodType (ceres/types.h through ceres.i:137).  A class with members, as SWIG's Java enums are: the reference
  * uses the name as a TYPE (CORE/NumericDiffCostFunction.scala:69, CORE/CostFunctor.scala:62) and imports its members
  * (`import NumericDiffMethodType._`, CORE/NumericDiffCostFunction.scala:80). */
sealed abstract class NumericDiffMethodType(val swigValue: Int)
object NumericDiffMethodType {
  case object CENTRAL extends NumericDiffMethodType(0)
  case object FORWARD extends NumericDiffMethodType(1)
  case object RIDDERS extends NumericDiffMethodType(2)
}
/** ceres::NumericDiffOptions (ceres/numeric_diff_options.h through ceres.i:39,143) with the defaults of Ceres 1.x; plain JVM state —
  * numerically differentiated functors are evaluated on the JVM (CORE/NumericDiffCostFunction.scala:75-162) and reach the solver
  * through the director path. */
class NumericDiffOptions {
  private var relativeStepSize = 1e-6
  private var riddersRelativeInitialStepSize = 1e-2
  private var maxNumRiddersExtrapolations = 10
  private var riddersEpsilon = 1e-12
  private var riddersStepShrinkFactor = 2.0
  def getRelativeStepSize: Double = relativeStepSize
  def setRelativeStepSize(v: Double): Unit = relativeStepSize = v
  def getRiddersRelativeInitialStepSize: Double = riddersRelativeInitialStepSize
  def setRiddersRelativeInitialStepSize(v: Double): Unit = riddersRelativeInitialStepSize = v
  def getMaxNumRiddersExtrapolations: Int = maxNumRiddersExtrapolations
  def setMaxNumRiddersExtrapolations(v: Int): Unit = maxNumRiddersExtrapolations = v
  def getRiddersEpsilon: Double = riddersEpsilon
  def setRiddersEpsilon(v: Double): Unit = riddersEpsilon = v
  def getRiddersStepShrinkFactor: Double = riddersStepShrinkFactor
  def setRiddersStepShrinkFactor(v: Double): Unit = riddersStepShrinkFactor = v
}
object TerminationType extends Enumeration { val CONVERGENCE = Value(0); val NO_CONVERGENCE = Value(1); val FAILURE = Value(2) }

/** ceres::LossFunction as an opaque handle; the JVM proxy owns it (`%newobject`, ceres.i:160-167). */
class LossFunction(val handle: Long) { override def finalize(): Unit = SkeresNative.skLossFree(handle) }
/** ceres.i:168-184 */
object PredefinedLossFunctions {
  def trivialLoss: LossFunction = new LossFunction(SkeresNative.skLossTrivial())
  def huberLoss(a: Double) = new LossFunction(SkeresNative.skLossHuber(a))
  def softLOneLoss(a: Double) = new LossFunction(SkeresNative.skLossSoftLOne(a))
  def cauchyLoss(a: Double) = new LossFunction(SkeresNative.skLossCauchy(a))
  def tukeyLoss(a: Double) = new LossFunction(SkeresNative.skLossTukey(a))
  def tolerantLoss(a: Double, b: Double) = new LossFunction(SkeresNative.skLossTolerant(a, b))
  def composedLoss(f: LossFunction, g: LossFunction) = new LossFunction(SkeresNative.skLossComposed(f.handle, g.handle))
  def scaledLoss(rho: LossFunction, a: Double) = new LossFunction(SkeresNative.skLossScaled(rho.handle, a))
}

class LocalParameterization(val handle: Long) { override def finalize(): Unit = SkeresNative.skLocalParameterizationFree(handle) }
/** ceres.i:192-210 */
object PredefinedLocalParameterizations {
  def identity(size: Int) = new LocalParameterization(SkeresNative.skLocalParameterizationIdentity(size))
  def subset(size: Int, constantParameters: Array[Int]) = new LocalParameterization(SkeresNative.skLocalParameterizationSubset(size, constantParameters))
  def quaternion = new LocalParameterization(SkeresNative.skLocalParameterizationQuaternion())
  def homogeneousVector(size: Int) = new LocalParameterization(SkeresNative.skLocalParameterizationHomogeneousVector(size))
}

/** ceres::CostFunction.  The reference's is a SWIG director (ceres.i:48); here the subclass decides how it reaches native
  * code (SizedCostFunction.nativeHandle): a device functor id, a recorded body, or the director trampoline. */
abstract class CostFunction {
  protected var numResidualsValue = 0
  protected val blockSizes = new StdVectorInt
  def setNumResiduals(n: Int): Unit = numResidualsValue = n
  def numResiduals(): Int = numResidualsValue
  def mutableParameterBlockSizes(): StdVectorInt = blockSizes
  def parameterBlockSizes(): StdVectorInt = blockSizes
  /** CostFunction::Evaluate: `jacobians` may be null, and so may any of its rows. */
  def evaluate(parameters: DoublePointerPointer, residuals: DoublePointer, jacobians: DoublePointerPointer): Boolean
  /** what the native director calls (jvm_evaluate in skeres_amd_jni.c) */
  final def evaluateNative(parameters: Long, residuals: Long, jacobians: Long): Boolean =
    evaluate(new DoublePointerPointer(parameters), new DoublePointer(residuals), new DoublePointerPointer(jacobians))
  /** native sk_cost_function* (created on first use) */
  def nativeHandle: Long
}

/** ceres::Problem, renamed as in ceres.i:73 */
class CeresProblem(options: CeresProblem.Options) {
  val handle: Long = SkeresNative.skProblemNew()
  def addResidualBlock(cost: CostFunction, loss: LossFunction, x: StdVectorDoublePointer): Long =
    SkeresNative.skProblemAddResidualBlock(handle, cost.nativeHandle, if (loss == null) 0L else loss.handle, x.handle)
  def addParameterBlock(values: DoublePointer, size: Int): Unit = SkeresNative.skProblemAddParameterBlock(handle, values.address, size, 0L)
  def addParameterBlock(values: DoublePointer, size: Int, p: LocalParameterization): Unit = SkeresNative.skProblemAddParameterBlock(handle, values.address, size, p.handle)
  def setParameterization(values: DoublePointer, p: LocalParameterization): Unit = SkeresNative.skProblemSetParameterization(handle, values.address, p.handle)
  def setParameterBlockConstant(values: DoublePointer): Unit = SkeresNative.skProblemSetParameterBlockConstant(handle, values.address)
  def setParameterBlockVariable(values: DoublePointer): Unit = SkeresNative.skProblemSetParameterBlockVariable(handle, values.address)
  def numResidualBlocks: Int = SkeresNative.skProblemNumResidualBlocks(handle)
  def numParameterBlocks: Int = SkeresNative.skProblemNumParameterBlocks(handle)
  def numParameters: Int = SkeresNative.skProblemNumParameters(handle)
  def numResiduals: Int = SkeresNative.skProblemNumResiduals(handle)
  override def finalize(): Unit = SkeresNative.skProblemFree(handle)
}
object CeresProblem {
  /** The native problem never owns cost or loss functions (sk_problem_new): the setters exist for source compatibility. */
  class Options { def setCostFunctionOwnership(o: Ownership.Value): Unit = (); def setLossFunctionOwnership(o: Ownership.Value): Unit = () }
}

object Solver {
  class Options {
    val handle: Long = SkeresNative.skOptionsNew()
    def setLinearSolverType(t: LinearSolverType.Value): Unit = SkeresNative.skOptionsSetLinearSolverType(handle, t.id)
    def setMinimizerType(t: MinimizerType.Value): Unit = SkeresNative.skOptionsSetMinimizerType(handle, t.id)
    def setMaxNumIterations(n: Int): Unit = SkeresNative.skOptionsSetMaxNumIterations(handle, n)
    def setMinimizerProgressToStdout(on: Boolean): Unit = SkeresNative.skOptionsSetMinimizerProgressToStdout(handle, if (on) 1 else 0)
    def setFunctionTolerance(v: Double): Unit = SkeresNative.skOptionsSetFunctionTolerance(handle, v)
    def setGradientTolerance(v: Double): Unit = SkeresNative.skOptionsSetGradientTolerance(handle, v)
    def setParameterTolerance(v: Double): Unit = SkeresNative.skOptionsSetParameterTolerance(handle, v)
    def setInitialTrustRegionRadius(v: Double): Unit = SkeresNative.skOptionsSetInitialTrustRegionRadius(handle, v)
    def setMaxTrustRegionRadius(v: Double): Unit = SkeresNative.skOptionsSetMaxTrustRegionRadius(handle, v)
    def setMinTrustRegionRadius(v: Double): Unit = SkeresNative.skOptionsSetMinTrustRegionRadius(handle, v)
    def setMinRelativeDecrease(v: Double): Unit = SkeresNative.skOptionsSetMinRelativeDecrease(handle, v)
    def setMinLmDiagonal(v: Double): Unit = SkeresNative.skOptionsSetMinLmDiagonal(handle, v)
    def setMaxLmDiagonal(v: Double): Unit = SkeresNative.skOptionsSetMaxLmDiagonal(handle, v)
    def setJacobiScaling(on: Boolean): Unit = SkeresNative.skOptionsSetJacobiScaling(handle, if (on) 1 else 0)
    def setMaxNumConsecutiveInvalidSteps(n: Int): Unit = SkeresNative.skOptionsSetMaxNumConsecutiveInvalidSteps(handle, n)
    // MI355X additions (no counterpart in ceres::Solver::Options)
    def setDevice(hipDevice: Int): Unit = SkeresNative.skOptionsSetDevice(handle, hipDevice)
    def setDistributionMode(mode: Int): Unit = SkeresNative.skOptionsSetDistributionMode(handle, mode)
    /** this JVM is rank `rank` of `world`, one per GPU; the collective is the library's own RCCL all-reduce */
    def setDistributedRccl(rank: Int, world: Int, uniqueId: Array[Byte]): Unit = {
      rccl = SkeresNative.skAllreduceRcclInit(rank, world, uniqueId)
      SkeresNative.skOptionsSetDistributedRccl(handle, rank, world, rccl)
    }
    private var rccl = 0L
    override def finalize(): Unit = { SkeresNative.skOptionsFree(handle); if (rccl != 0L) SkeresNative.skAllreduceRcclFree(rccl) }
  }
  class Summary {
    val handle: Long = SkeresNative.skSummaryNew()
    def initialCost: Double = SkeresNative.skSummaryInitialCost(handle)
    def finalCost: Double = SkeresNative.skSummaryFinalCost(handle)
    def numSuccessfulSteps: Int = SkeresNative.skSummaryNumSuccessfulSteps(handle)
    def numUnsuccessfulSteps: Int = SkeresNative.skSummaryNumUnsuccessfulSteps(handle)
    def terminationType: TerminationType.Value = TerminationType(SkeresNative.skSummaryTerminationType(handle))
    // (they differ when DENSE_SCHUR met a problem without the 2 / (9, 3) block structure: its alternate, DENSE_QR, was used — as Ceres does)
    def linearSolverTypeUsed: LinearSolverType.Value = LinearSolverType(SkeresNative.skSummaryLinearSolverTypeUsed(handle))
    def linearSolverTypeGiven: LinearSolverType.Value = LinearSolverType(SkeresNative.skSummaryLinearSolverTypeGiven(handle))
    def message: String = SkeresNative.skSummaryMessage(handle)
    def briefReport(): String = SkeresNative.skSummaryBriefReport(handle)
    def fullReport(): String = SkeresNative.skSummaryFullReport(handle)
    override def finalize(): Unit = SkeresNative.skSummaryFree(handle)
  }
}

/** the module class of ceres.i: free functions */
object ceres {
  def initGoogleLogging(name: String): Unit = SkeresNative.skInitLogging(name)
  def solve(options: Solver.Options, problem: CeresProblem, summary: Solver.Summary): Unit =
    SkeresNative.skSolve(options.handle, problem.handle, summary.handle)
}
