package org.somelightprojections.skeres

import com.google.ceres.{DoubleArray, DoubleMatrix, StdVectorDoublePointer}

/** View over a native `double**` (CORE/RichDoubleMatrix.scala:32-60): the same members. */
case class RichDoubleMatrix(data: DoublePointerPointer) {
  def isNull: Boolean = DoubleMatrix.isNull(data)
  def hasRow(i: Int): Boolean = !DoubleMatrix.row(data, i).isNull
  def getRow(i: Int): DoubleArray = DoubleArray.frompointer(DoubleMatrix.row(data, i))
  def get(i: Int, j: Int): Double = getRow(i).getitem(j)
  def set(i: Int, j: Int, x: Double): Unit = getRow(i).setitem(j, x)
  def copyRowFrom(i: Int, from: Array[Double]): Unit = copyRowFrom(i, from, 0, from.length)
  /** the leading (end - begin) elements of row i from from[begin, end) (CORE/RichDoubleMatrix.scala:58-60) */
  def copyRowFrom(i: Int, from: Array[Double], begin: Int, end: Int): Unit = RichDoubleArray(getRow(i)).copyFrom(from.slice(begin, end))
}

/** The factories of CORE/RichDoubleMatrix.scala:65-98, used by every spec of the reference that calls `evaluate`
  * (TEST/AutodiffCostFuntionSpec.scala:28,43,71,91; TEST/NumericDiffCostFunctionSpec.scala:37-39,121-122;
  * TEST/RichDoubleMatrixSpec.scala:14,45,68,81).  The blocks and the pointer vector are native memory that stays allocated for
  * the life of the program, as in the reference (its StdVectorDoublePointer proxies are dropped as well: the `double**` they
  * hand out must outlive them, so the vectors made here are deliberately never freed). */
object RichDoubleMatrix {
  private val keep = scala.collection.mutable.ArrayBuffer.empty[StdVectorDoublePointer]  // (a vector's storage IS the double**)

  /** numRows native blocks of numColumns doubles each; not necessarily contiguous. */
  def ofSize(numRows: Int, numColumns: Int): DoublePointerPointer =
    if (numRows == 0 || numColumns == 0) {
      fromStdVector(null)
    } else {
      val pointerVector = new StdVectorDoublePointer(numRows)
      (0 until numRows).foreach(i => pointerVector.set(i, RichDoubleArray.ofSize(numColumns)))
      fromStdVector(pointerVector)
    }

  /** the blocks vec points to, as a `double**` (ceres.i:121-123); null gives the null matrix */
  def fromStdVector(vec: StdVectorDoublePointer): DoublePointerPointer =
    if (vec == null) new DoublePointerPointer(0L) else { keep.synchronized { keep += vec }; DoubleMatrix.toPointerPointer(vec) }

  /** a.length native blocks initialised with the given arrays (copied: the arrays may go out of scope) */
  def fromArrays(a: Array[Double]*): DoublePointerPointer = {
    val v = new StdVectorDoublePointer()
    a.foreach(ai => v.add(RichDoubleArray.fromArray(ai)))
    fromStdVector(v)
  }

  /** an empty collection of native memory blocks */
  lazy val empty: DoublePointerPointer = ofSize(0, 0)
}
