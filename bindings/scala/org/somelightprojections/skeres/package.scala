package org.somelightprojections

import scala.language.implicitConversions
import spire.algebra.Order
import spire.math._
import spire.implicits._

// Same names as the reference's package object (CORE/package.scala:9-28); the pointer types are the hand-written ones of
// bindings/scala/com/google/ceres/Native.scala instead of SWIG's SWIGTYPE_p_*.
package object skeres {
  type DoublePointerPointer = com.google.ceres.DoublePointerPointer
  type DoublePointer = com.google.ceres.DoublePointer
  type ResidualBlockId = Long

  val EpsilonDouble: Double = ulp(1.0D)

  implicit def doublePointerPointerToRichDoubleMatrix(p: DoublePointerPointer): RichDoubleMatrix = RichDoubleMatrix(p)
  implicit def doublePointerToRichDoubleArray(p: DoublePointer): RichDoubleArray = RichDoubleArray(com.google.ceres.DoubleArray.frompointer(p))
  implicit def doubleArraytoRichDoubleArray(a: com.google.ceres.DoubleArray): RichDoubleArray = RichDoubleArray(a)

  /** Jets are ordered by their real parts: this is what decides every branch inside a generic functor (CORE/package.scala:27). */
  implicit val jetDoubleOrder: Order[Jet[Double]] = Order.by[Jet[Double], Double](_.real)
}
