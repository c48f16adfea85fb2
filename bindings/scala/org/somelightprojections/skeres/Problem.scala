package org.somelightprojections.skeres

import com.google.ceres._
import scala.collection.mutable

// CHANGED with respect to CORE/Problem.scala:6-33 only in what it extends (the hand-written CeresProblem) and in the bulk
// form of the set-up loop; ownership is as in the reference — the Problem never owns cost or loss functions, and keeps
// JVM references to them for its own lifetime.
object Problem {
  class Options extends CeresProblem.Options {
    setCostFunctionOwnership(Ownership.DO_NOT_TAKE_OWNERSHIP)
    setLossFunctionOwnership(Ownership.DO_NOT_TAKE_OWNERSHIP)
  }
}

class Problem(opts: Problem.Options) extends CeresProblem(opts) {
  def this() = this(new Problem.Options())

  /** CORE/Problem.scala:20-27: parameter blocks are identified by address. */
  def addResidualBlock(cost: CostFunction, loss: LossFunction, x: DoublePointer*): ResidualBlockId = {
    costs += cost
    losses += loss
    val xv = new StdVectorDoublePointer()
    x.foreach(xv.add)
    addResidualBlock(cost, loss, xv)
  }

  /** The set-up loop of EX/SimpleBundleAdjuster.scala:139-145 in one JNI crossing: `n` residual blocks of ONE functor class
    * (a device body or a recording), block b with the captured doubles consts(b * k .. ) and parameter blocks at the element
    * offsets offsets(b * blocks .. ) of the one native array `base` (BalProblem's layout, :18-34).  5 M observations are then
    * one call instead of 5 M x (3 + 3) crossings. */
  def addResidualBlocks(functor: AutoDiffCostFunctor, n: Int, consts: Array[Double], loss: LossFunction, base: DoublePointer, offsets: Array[Long]): Unit = {
    losses += loss
    functor.deviceFunctorId match {
      case Some(id) => SkeresNative.skProblemAddResidualBlocks(handle, id, n, consts, loss.handle, base.address, offsets)
      case None =>
        val cost = functor.toAutoDiffCostFunction
        costs += cost
        SkeresNative.skProblemAddResidualBlocksTape(handle, cost.nativeHandle, n, consts, loss.handle, base.address, offsets)
    }
  }

  private val costs = mutable.ListBuffer.empty[CostFunction]
  private val losses = mutable.ListBuffer.empty[LossFunction]
}
