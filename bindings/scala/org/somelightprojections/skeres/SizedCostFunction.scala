package org.somelightprojections.skeres

import com.google.ceres.{CostFunction, SkeresNative}

// CHANGED with respect to CORE/SizedCostFunction.scala:6-14: the base class is no SWIG director; a native handle is made
// on first use.  The default is the director path (sk_cost_function_new_callback with the JNI trampoline): the native
// solver calls evaluate() on the JVM for this block, with the exact signature of the reference (ceres.i:48).
abstract class SizedCostFunction(val kNumResiduals: Int, val N: Int*) extends CostFunction {
  require(N.forall(_ >= 0), s"Negative block size detected. Block size are: ${N.mkString(", ")}")
  require(N.indices.tail.forall(i => N(i) == 0 || N(i - 1) > 0),
    "Zero block cannot precede a non-zero block. Block sizes are (ignore trailing 0's): " + N.mkString(", "))
  setNumResiduals(kNumResiduals)
  N.foreach(mutableParameterBlockSizes.add)

  private var director = 0L
  /** how this cost function reaches native code; AutoDiffCostFunction overrides it for device and recorded bodies */
  protected def makeNativeHandle(): Long = {
    director = SkeresNative.skDirectorNew(this)
    SkeresNative.skCostFunctionNewCallback(director, kNumResiduals, N.toArray)
  }
  lazy val nativeHandle: Long = makeNativeHandle()
  override def finalize(): Unit = {
    SkeresNative.skCostFunctionFree(nativeHandle)
    if (director != 0L) SkeresNative.skDirectorFree(director)
  }
}
