package org.somelightprojections.skeres

import com.google.ceres.SkeresNative
import spire.implicits._
import spire.math.{Jet, JetDim}

// CHANGED with respect to CORE/AutodiffCostFunction.scala:69-135: three ways to a native handle instead of one director.
// evaluate() keeps the reference's contract — it is what TEST/AutodiffCostFuntionSpec.scala calls directly, and what the
// native solver calls back for a functor that has neither a device body nor a recording.
case class AutoDiffCostFunction(costFunctor: AutoDiffCostFunctor)
  extends SizedCostFunction(costFunctor.kNumResiduals, costFunctor.N: _*) {

  implicit val jetDimension = JetDim(costFunctor.N.sum)

  override protected def makeNativeHandle(): Long = costFunctor.deviceFunctorId match {
    case Some(id) =>                          // the body is in the device registry: no upcall, ever
      SkeresNative.skCostFunctionNewAutodiff(id, costFunctor.deviceConstants)
    case None if costFunctor.recordable =>    // run apply[Rec] once; the GPU interprets the tape per residual block
      val t = Recording.of(costFunctor)
      SkeresNative.skCostFunctionNewTape(kNumResiduals, costFunctor.N.toArray, t.instructions, t.constants, t.numRegisters, t.outputs,
                                         costFunctor.deviceConstants)
    case None => super.makeNativeHandle()     // director path: evaluate() below, on the JVM
  }

  /** CostFunction::Evaluate on the JVM.  Cost-only when `jacobians` is null: the functor over Doubles.  Otherwise the functor
    * over Jets of dimension sum(N), scalar j of block i seeded with the k-th unit infinitesimal, k counting scalars in block
    * order; residuals are the real parts; block i's Jacobian is row-major kNumResiduals x N(i), skipped when its row pointer
    * is null.  An empty functor result is `false`. */
  override def evaluate(parameters: DoublePointerPointer, residuals: DoublePointer, jacobians: DoublePointerPointer): Boolean = {
    val sizes = costFunctor.N
    val blocks = sizes.length
    if (jacobians.isNull) {
      val x = new Array[Array[Double]](blocks)
      var i = 0
      while (i < blocks) { x(i) = parameters.getRow(i).toArray(sizes(i)); i += 1 }
      val y = costFunctor(x: _*)
      if (y.isEmpty) return false
      residuals.copyFrom(y)
      true
    } else {
      val jx = new Array[Array[Jet[Double]]](blocks)
      var seed = 0
      var i = 0
      while (i < blocks) {
        val values = parameters.getRow(i).toArray(sizes(i))
        jx(i) = Array.tabulate(sizes(i)) { j => val jet = Jet[Double](values(j), seed); seed += 1; jet }
        i += 1
      }
      val jy = costFunctor(jx: _*)
      if (jy.isEmpty) return false
      residuals.copyFrom(jy.map(_.real))
      var offset = 0
      i = 0
      while (i < blocks) {
        val n = sizes(i)
        if (jacobians.hasRow(i)) {
          val block = new Array[Double](kNumResiduals * n)
          var r = 0
          while (r < kNumResiduals) { System.arraycopy(jy(r).infinitesimal, offset, block, r * n, n); r += 1 }
          jacobians.copyRowFrom(i, block)   // one crossing per block instead of one per entry
        }
        offset += n
        i += 1
      }
      true
    }
  }
}
