package org.somelightprojections.skeres

import com.google.ceres.{DoubleArray, DoubleMatrix}

/** View over a native `double**` (CORE/RichDoubleMatrix.scala:32-60): the members AutoDiffCostFunction.evaluate uses. */
case class RichDoubleMatrix(data: DoublePointerPointer) {
  def isNull: Boolean = DoubleMatrix.isNull(data)
  def hasRow(i: Int): Boolean = !DoubleMatrix.row(data, i).isNull
  def getRow(i: Int): DoubleArray = DoubleArray.frompointer(DoubleMatrix.row(data, i))
  def get(i: Int, j: Int): Double = getRow(i).getitem(j)
  def set(i: Int, j: Int, x: Double): Unit = getRow(i).setitem(j, x)
  def copyRowFrom(i: Int, from: Array[Double]): Unit = RichDoubleArray(getRow(i)).copyFrom(from)
}
