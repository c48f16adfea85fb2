package org.somelightprojections.skeres

import com.google.ceres.{DoubleArray, DoubleArraySlice, SkeresNative}

/** CORE/RichDoubleArray.scala with the same members; copyFrom / toArray cross JNI once (sk_array_copy_in / _out) instead of
  * once per element (CORE/RichDoubleArray.scala:36-39, 65-69).  No bounds checks, as in the reference. */
case class RichDoubleArray(a: DoubleArray) {
  def get(i: Int): Double = a.getitem(i)
  def set(i: Int, x: Double): Unit = a.setitem(i, x)
  def copyFrom(from: Array[Double]): DoubleArray = { SkeresNative.skArrayCopyIn(a.address, from, from.length); a }
  def isNull: Boolean = a.address == 0L
  def slice(start: Int): DoublePointer = DoubleArraySlice.get(a.cast, start)
  def toPointer: DoublePointer = a.cast
  def toArray(length: Int): Array[Double] = { val out = new Array[Double](length); SkeresNative.skArrayCopyOut(a.address, out, length); out }
}

object RichDoubleArray {
  /** native memory handed to the caller as a bare pointer: caller-owned for the life of the program, as in the reference
    * (CORE/RichDoubleArray.scala:73) — the DoubleArray that allocated it is deliberately leaked */
  def ofSize(n: Int): DoublePointer = new DoublePointer(SkeresNative.skArrayNew(n))
  def fromArray(a: Array[Double]): DoublePointer = { val p = ofSize(a.length); SkeresNative.skArrayCopyIn(p.address, a, a.length); p }
}
