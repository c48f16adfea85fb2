package org.somelightprojections.skeres

import scala.collection.mutable.ArrayBuffer
import scala.reflect.ClassTag
import spire.algebra._

// NEW (no counterpart in the reference).  `apply` of an AutoDiffCostFunctor is generic over Field / Trig / NRoot / Order
// (CORE/CostFunctor.scala:50), and the reference instantiates it with Double and Jet[Double].  `Rec` is a third T: its
// arithmetic appends an instruction to a tape and returns a fresh virtual register.  Running `apply` ONCE on Rec
// parameters yields the tape that libskeres_amd interprets on the GPU per residual block (sk_cost_function_new_tape,
// csrc/tape.hpp) — a functor without a body in the device registry still runs on the device, where the reference calls the
// JVM back for every block (ceres.i:48).  skeres_amd/tape.py is this file in Python, and tests/test_traced_functors.py
// its tests; the instruction format is include/skeres_amd.h's sk_tape_opcode / sk_tape_operand_kind.
//
// A data-dependent branch cannot be taken while recording (spire's Order compares real parts, CORE/Rotation.scala:458): a
// recordable functor states it as Recording.where(cond)(thenArm)(elseArm) — both arms go on the tape, a SELECT picks one per
// evaluation; for T = Double / Jet the same helper is a plain `if`.  Comparing two Rec values any other way throws.

object TapeOp {
  val MOV = 0; val ADD = 1; val SUB = 2; val MUL = 3; val DIV = 4; val NEG = 5; val SQRT = 6; val EXP = 7; val LOG = 8; val SIN = 9; val COS = 10
  val TAN = 11; val ASIN = 12; val ACOS = 13; val ATAN = 14; val ATAN2 = 15; val ABS = 16; val LT = 17; val LE = 18; val SELECT = 19
  val REGISTER = 0; val PARAMETER = 1; val CAPTURED = 2; val CONSTANT = 3
  def operand(kind: Int, index: Int): Int = (kind << 28) | index
}

/** a value of the recording: an operand of the tape under construction */
final class Rec(val tape: Recorder, val kind: Int, val index: Int)

final class Recorder {
  import TapeOp._
  private val ins = ArrayBuffer.empty[Array[Int]]   // op, virtual destination, (kind, index) x 3 flattened; -1 = absent
  val consts = ArrayBuffer.empty[Double]
  private var virtual = 0
  def const(x: Double): Rec = {
    val bits = java.lang.Double.doubleToRawLongBits(x)
    var i = consts.indexWhere(c => java.lang.Double.doubleToRawLongBits(c) == bits)
    if (i < 0) { consts += x; i = consts.length - 1 }
    new Rec(this, CONSTANT, i)
  }
  def emit(op: Int, args: Rec*): Rec = {
    require(args.forall(_.tape eq this), "values of two recordings mixed")
    val row = Array.fill(8)(-1)
    row(0) = op; row(1) = virtual
    args.zipWithIndex.foreach { case (a, k) => row(2 + 2 * k) = a.kind; row(3 + 2 * k) = a.index }
    ins += row
    virtual += 1
    new Rec(this, REGISTER, virtual - 1)
  }
  /** Drops what no output depends on and re-numbers the virtual registers by liveness (a register dies at its last use:
    * the device keeps a thread's registers in LDS, so few matter).  Returns what sk_cost_function_new_tape takes. */
  def finish(outputs: Seq[Rec]): Recording.Tape = {
    val needed = scala.collection.mutable.Set.empty[Int] ++ outputs.filter(_.kind == REGISTER).map(_.index)
    val keep = new Array[Boolean](ins.length)
    for (i <- ins.indices.reverse if needed(ins(i)(1))) {
      keep(i) = true
      for (k <- 0 until 3 if ins(i)(2 + 2 * k) == REGISTER) needed += ins(i)(3 + 2 * k)
    }
    val live = ins.indices.filter(keep(_)).map(ins(_))
    val lastUse = scala.collection.mutable.Map.empty[Int, Int]
    for ((row, i) <- live.zipWithIndex; k <- 0 until 3 if row(2 + 2 * k) == REGISTER) lastUse(row(3 + 2 * k)) = i
    outputs.filter(_.kind == REGISTER).foreach(o => lastUse(o.index) = live.length)
    val phys = scala.collection.mutable.Map.empty[Int, Int]
    val free = scala.collection.mutable.Stack.empty[Int]
    var nregs = 0
    val out = ArrayBuffer.empty[Int]
    for ((row, i) <- live.zipWithIndex) {
      val enc = Array(0, 0, 0)
      for (k <- 0 until 3 if row(2 + 2 * k) >= 0)
        enc(k) = if (row(2 + 2 * k) == REGISTER) operand(REGISTER, phys(row(3 + 2 * k))) else operand(row(2 + 2 * k), row(3 + 2 * k))
      for (k <- 0 until 3 if row(2 + 2 * k) == REGISTER && lastUse.get(row(3 + 2 * k)).contains(i) && phys.contains(row(3 + 2 * k)))
        free.push(phys.remove(row(3 + 2 * k)).get)   // a register read here for the last time may hold the result
      val r = if (free.nonEmpty) free.pop() else { nregs += 1; nregs - 1 }
      phys(row(1)) = r
      out ++= Seq(row(0), r, enc(0), enc(1), enc(2))
    }
    val outOps = outputs.map(o => if (o.kind == REGISTER) operand(REGISTER, phys(o.index)) else operand(o.kind, o.index))
    Recording.Tape(out.toArray, consts.toArray, nregs, outOps.toArray)
  }
}

object Recording {
  import TapeOp._
  final case class Tape(instructions: Array[Int], constants: Array[Double], numRegisters: Int, outputs: Array[Int])

  /** the typeclass instances `apply[T]` asks for, for T = Rec */
  implicit object RecField extends Field[Rec] {
    private def lift(t: Recorder, x: Double) = t.const(x)
    def zero: Rec = throw new UnsupportedOperationException("Rec literals need a tape: use fromDouble through a value (x * 0)")
    def one: Rec = zero
    def plus(a: Rec, b: Rec): Rec = a.tape.emit(ADD, a, b)
    override def minus(a: Rec, b: Rec): Rec = a.tape.emit(SUB, a, b)
    def times(a: Rec, b: Rec): Rec = a.tape.emit(MUL, a, b)
    def div(a: Rec, b: Rec): Rec = a.tape.emit(DIV, a, b)
    def negate(a: Rec): Rec = a.tape.emit(NEG, a)
    def quot(a: Rec, b: Rec): Rec = div(a, b)
    def mod(a: Rec, b: Rec): Rec = throw new UnsupportedOperationException("mod cannot be recorded")
    def gcd(a: Rec, b: Rec): Rec = throw new UnsupportedOperationException("gcd cannot be recorded")
    /** literals of the body mixed with a Rec (x * 2.0, 1.0 + r2 * ...) arrive through spire's Double => T conversions */
    def literal(like: Rec, x: Double): Rec = lift(like.tape, x)
  }
  implicit object RecTrig extends Trig[Rec] {
    def e: Rec = RecField.zero; def pi: Rec = RecField.zero
    def exp(a: Rec): Rec = a.tape.emit(EXP, a); def expm1(a: Rec): Rec = a.tape.emit(SUB, exp(a), a.tape.const(1.0))
    def log(a: Rec): Rec = a.tape.emit(LOG, a); def log1p(a: Rec): Rec = log(a.tape.emit(ADD, a, a.tape.const(1.0)))
    def sin(a: Rec): Rec = a.tape.emit(SIN, a); def cos(a: Rec): Rec = a.tape.emit(COS, a); def tan(a: Rec): Rec = a.tape.emit(TAN, a)
    def asin(a: Rec): Rec = a.tape.emit(ASIN, a); def acos(a: Rec): Rec = a.tape.emit(ACOS, a); def atan(a: Rec): Rec = a.tape.emit(ATAN, a)
    def atan2(y: Rec, x: Rec): Rec = y.tape.emit(ATAN2, y, x)
    def sinh(a: Rec): Rec = { val ep = exp(a); val em = exp(a.tape.emit(NEG, a)); a.tape.emit(MUL, a.tape.emit(SUB, ep, em), a.tape.const(0.5)) }
    def cosh(a: Rec): Rec = { val ep = exp(a); val em = exp(a.tape.emit(NEG, a)); a.tape.emit(MUL, a.tape.emit(ADD, ep, em), a.tape.const(0.5)) }
    def tanh(a: Rec): Rec = a.tape.emit(DIV, sinh(a), cosh(a))
    def toRadians(a: Rec): Rec = a.tape.emit(MUL, a, a.tape.const(math.Pi / 180.0)); def toDegrees(a: Rec): Rec = a.tape.emit(MUL, a, a.tape.const(180.0 / math.Pi))
  }
  implicit object RecNRoot extends NRoot[Rec] {
    def nroot(a: Rec, n: Int): Rec = if (n == 2) a.tape.emit(SQRT, a) else fpow(a, a.tape.const(1.0 / n))
    def fpow(a: Rec, b: Rec): Rec = RecTrig.exp(a.tape.emit(MUL, b, RecTrig.log(a)))   // spire's fpow for a positive base
  }
  /** Order cannot answer while recording. */
  implicit object RecOrder extends Order[Rec] {
    def compare(a: Rec, b: Rec): Int =
      throw new UnsupportedOperationException("a comparison of recorded values cannot decide an `if`: state the branch as Recording.where(a, b)(...)")
  }

  /** `if (a > b) thenArm else elseArm` of a generic body, for every T.  Recorded: both arms on the tape, one SELECT per value. */
  def whereGreater[T: Order: ClassTag](a: T, b: T)(thenArm: => Array[T])(elseArm: => Array[T]): Array[T] = (a, b) match {
    case (ra: Rec, rb: Rec) =>
      val cond = ra.tape.emit(LT, rb, ra)
      val (u, v) = (thenArm.asInstanceOf[Array[Rec]], elseArm.asInstanceOf[Array[Rec]])
      require(u.length == v.length, s"the two arms of a where return ${u.length} and ${v.length} values")
      u.zip(v).map { case (x, y) => ra.tape.emit(SELECT, cond, x, y) }.asInstanceOf[Array[T]]
    case _ => if (implicitly[Order[T]].gt(a, b)) thenArm else elseArm
  }

  /** Runs functor.apply once on recorded parameters (operand PARAMETER k, k counting scalars in block order, as
    * AutoDiffCostFunction seeds its Jets) and captured doubles (operand CAPTURED i). */
  def of(functor: AutoDiffCostFunctor): Tape = {
    val tape = new Recorder
    var k = 0
    val x = functor.N.map { n => val block = Array.tabulate(n)(j => new Rec(tape, PARAMETER, k + j)); k += n; block }
    val y = functor.apply[Rec](x: _*)
    require(y.length == functor.kNumResiduals,
      s"the functor returned ${y.length} residuals while recording, kNumResiduals is ${functor.kNumResiduals} (an empty result cannot be recorded)")
    tape.finish(y)
  }
}
