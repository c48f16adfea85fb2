package org.somelightprojections.skeres.examples

import org.somelightprojections.skeres._

// What changes in the reference's example programs: NOTHING in their main() bodies; each functor with a body in the device
// registry gains two one-line overrides.  The generic `apply` stays as it is in the reference (EX/SimpleBundleAdjuster.scala:81-118,
// EX/CurveFitting.scala:93-97) — it is still what runs for T = Double / Jet on the JVM (TEST/AutodiffCostFuntionSpec-style
// direct calls of evaluate) — and is therefore not repeated here: mix these traits into the reference's classes.

/** class SnavelyReprojectionError(observedX, observedY) extends AutoDiffCostFunctor(2, 9, 3) with SnavelyOnDevice */
trait SnavelyOnDevice { self: AutoDiffCostFunctor =>
  def observedX: Double
  def observedY: Double
  override def deviceFunctorId: Option[Int] = Some(1)                        // SK_FUNCTOR_SNAVELY_REPROJECTION
  override def deviceConstants: Array[Double] = Array(observedX, observedY)
}

/** class ExponentialResidual(x, y) extends AutoDiffCostFunctor(1, 1, 1) with ExponentialOnDevice */
trait ExponentialOnDevice { self: AutoDiffCostFunctor =>
  def x: Double
  def y: Double
  override def deviceFunctorId: Option[Int] = Some(2)                        // SK_FUNCTOR_EXPONENTIAL_RESIDUAL
  override def deviceConstants: Array[Double] = Array(x, y)
}

/** The bundle adjuster's set-up loop (EX/SimpleBundleAdjuster.scala:139-145) in one native call instead of N x 6 crossings. */
object BundleAdjusterSetup {
  /** cameraIndex / pointIndex / observations as BalProblem holds them (:18-34); parameters = the one native array */
  def addAll(problem: Problem, loss: com.google.ceres.LossFunction, numCameras: Int, cameraIndex: Array[Int], pointIndex: Array[Int],
             observations: Array[Double], parameters: DoublePointer): Unit = {
    val n = cameraIndex.length
    val offsets = new Array[Long](2 * n)
    var i = 0
    while (i < n) { offsets(2 * i) = 9L * cameraIndex(i); offsets(2 * i + 1) = 9L * numCameras + 3L * pointIndex(i); i += 1 }
    val functor = new AutoDiffCostFunctor(2, 9, 3) with SnavelyOnDevice {
      val observedX = 0.0; val observedY = 0.0   // (per-block values travel in `observations`)
      def apply[T: spire.algebra.Field: spire.algebra.Trig: spire.algebra.NRoot: spire.algebra.Order: scala.reflect.ClassTag](x: Array[T]*): Array[T] =
        throw new UnsupportedOperationException("evaluated on the device")
    }
    problem.addResidualBlocks(functor, n, observations, loss, parameters, offsets)
  }
}
