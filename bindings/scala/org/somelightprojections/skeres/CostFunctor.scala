package org.somelightprojections.skeres

import com.google.ceres.{NumericDiffMethodType, NumericDiffOptions}
import scala.reflect.ClassTag
import scala.{specialized => sp}
import spire.algebra._

// CHANGED with respect to CORE/CostFunctor.scala:31-51: an AutoDiffCostFunctor may name a device body.  Everything a user
// functor already does keeps compiling and running: `apply` is unchanged.

/** Sizes of a functor: `kNumResiduals` outputs from parameter blocks of sizes `N` (CORE/CostFunctor.scala:31-34). */
abstract class CostFunctor(val kNumResiduals: Int, val N: Int*) {
  require(kNumResiduals > 0, s"Nonpositive number of residuals specified: $kNumResiduals")
  require(N.forall(_ > 0), s"Nonpositive parameter block sizes specified: ${N.mkString(", ")}")
}

abstract class AutoDiffCostFunctor(kNumResiduals: Int, N: Int*) extends CostFunctor(kNumResiduals, N: _*) {
  /** The generic residual (CORE/CostFunctor.scala:50): an empty result reports failure. */
  def apply[@sp(Double) T: Field: Trig: NRoot: Order: ClassTag](x: Array[T]*): Array[T]

  def toAutoDiffCostFunction = AutoDiffCostFunction(this)

  // ---- where the body runs when the solver evaluates it (new) ----
  /** Id of this functor's body in the device registry (`sk_functor_id` of include/skeres_amd.h: 1 = SnavelyReprojectionError,
    * 2 = ExponentialResidual, ...), or None.  With an id, the solver evaluates the functor on the GPU and never calls back. */
  def deviceFunctorId: Option[Int] = None
  /** The doubles this closure captures, in the order the device body (or the recording) expects them. */
  def deviceConstants: Array[Double] = Array.empty
  /** No device body: may `apply` be run once with the recording T (Recording.scala) and interpreted on the GPU?  False for a
    * functor that reports failure with an empty result or branches with a plain `if` on its arguments; such functors are
    * called back on the JVM once per residual block and evaluation, as the reference does with every functor (ceres.i:48). */
  def recordable: Boolean = false
}

/** Unchanged in shape (CORE/CostFunctor.scala:57-71): numerically differentiated functors are evaluated on the JVM. */
abstract class NumericDiffCostFunctor(kNumResiduals: Int, N: Int*) extends CostFunctor(kNumResiduals, N: _*) {
  /** A numerically differentiated cost function computed through this functor (CORE/CostFunctor.scala:61-64): the reference's own
    * NumericDiffCostFunction.scala, unchanged — it extends SizedCostFunction and so reaches the solver through the director path. */
  def toNumericDiffCostFunction(method: NumericDiffMethodType, options: Option[NumericDiffOptions] = None) =
    NumericDiffCostFunction(method, options.getOrElse(new NumericDiffOptions), this)

  def apply(x: Array[Double]*): Array[Double]
}
