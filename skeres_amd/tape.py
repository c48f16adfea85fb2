"""Recording a generic functor body once, for the device (include/skeres_amd.h: sk_cost_function_new_tape).

The reference's residuals are written generically,

    def apply[T: Field: Trig: NRoot: Order: ClassTag](x: Array[T]*): Array[T]      (CORE/CostFunctor.scala:40-51)

and instantiated with T = Double and T = Jet[Double].  ``Traced`` below is a third T: its arithmetic appends an
instruction to a tape and returns a fresh virtual register.  Running ``apply`` once on
``Traced`` parameters yields the tape the device interpreter evaluates per residual block (csrc/tape.hpp) — so a
functor WITHOUT a body in the device registry still runs on the GPU, where ``HostAutoDiffCostFunctor`` is called
back on the host for every block.

    class MyResidual(sk.TracedCostFunctor):
        def __init__(self, ox, oy):
            super().__init__(2, 9, 3, captured=(ox, oy))        # kNumResiduals, N(0), N(1); what the closure captures
        def apply(self, cam, X):                                # generic in T: floats, rotation.Jet or Traced
            ox, oy = self.captured_values()                     # the captured doubles as T
            ...
            return [fx * xp - ox, fy * yp - oy]

A data-dependent branch (spire's Order compares real parts, CORE/Rotation.scala:458) cannot be taken while
recording: write it as ``tape.where(a > b, then, otherwise)`` — both arms are recorded, a select picks one per
evaluation (and ``where`` on plain floats / Jets just picks; pass the arms as callables when the one not taken must
not run on the host).  ``bool(a > b)`` on traced values raises.
"""
import numpy as np

(MOV, ADD, SUB, MUL, DIV, NEG, SQRT, EXP, LOG, SIN, COS, TAN, ASIN, ACOS, ATAN, ATAN2, ABS, LT, LE, SELECT) = range(20)
REGISTER, PARAMETER, CAPTURED, CONSTANT = range(4)
_UNARY = {"sqrt": SQRT, "exp": EXP, "log": LOG, "sin": SIN, "cos": COS, "tan": TAN, "asin": ASIN, "acos": ACOS, "atan": ATAN}


def _operand(kind, index):
    return (kind << 28) | index


class Recorder:
    """The tape under construction: instructions over virtual registers, literals, and at the end a register
    allocation (a virtual register dies at its last use; the device keeps a thread's registers in LDS, so few matter)."""

    def __init__(self):
        self.ins = []        # [op, dst (virtual), a, b, c] with operands as (kind, index) or None
        self.consts = []
        self._const_index = {}
        self.num_virtual = 0

    def const(self, x):
        x = float(x)
        k = np.float64(x).tobytes()
        if k not in self._const_index:
            self._const_index[k] = len(self.consts)
            self.consts.append(x)
        return Traced(self, (CONSTANT, self._const_index[k]))

    def lift(self, x):
        if isinstance(x, Traced):
            if x.rec is not self:
                raise ValueError("values of two recordings mixed")
            return x
        return self.const(x)

    def emit(self, op, *args):
        dst = self.num_virtual
        self.num_virtual += 1
        ops = [self.lift(a).operand for a in args] + [None] * (3 - len(args))
        self.ins.append([op, dst] + ops)
        return Traced(self, (REGISTER, dst))

    def finish(self, outputs):
        """(instructions as int32 [n, 5], constants, number of registers, output operands) with physical registers."""
        outs = [self.lift(o).operand for o in outputs]
        # drop what no output depends on (the recording may compute values it never returns)
        needed, keep = set(o[1] for o in outs if o[0] == REGISTER), [False] * len(self.ins)
        for i in range(len(self.ins) - 1, -1, -1):
            if self.ins[i][1] in needed:
                keep[i] = True
                for o in self.ins[i][2:]:
                    if o is not None and o[0] == REGISTER:
                        needed.add(o[1])
        ins = [self.ins[i] for i in range(len(self.ins)) if keep[i]]
        last_use = {}
        for i, (_, _, *ops) in enumerate(ins):
            for o in ops:
                if o is not None and o[0] == REGISTER:
                    last_use[o[1]] = i
        for o in outs:
            if o[0] == REGISTER:
                last_use[o[1]] = len(ins)
        phys, free, nregs, out_ins = {}, [], 0, []
        for i, (op, dst, *ops) in enumerate(ins):
            enc = [0, 0, 0]
            for k, o in enumerate(ops):
                if o is not None:
                    enc[k] = _operand(REGISTER, phys[o[1]]) if o[0] == REGISTER else _operand(*o)
            for o in ops:  # registers read here for the last time may hold the result
                if o is not None and o[0] == REGISTER and last_use.get(o[1]) == i and o[1] in phys:
                    free.append(phys.pop(o[1]))
            if free:
                r = free.pop()
            else:
                r = nregs
                nregs += 1
            phys[dst] = r
            out_ins.append([op, r] + enc)
        out_ops = [_operand(REGISTER, phys[o[1]]) if o[0] == REGISTER else _operand(*o) for o in outs]
        return (np.asarray(out_ins, dtype=np.int32).reshape(-1, 5), np.asarray(self.consts, dtype=np.float64), nregs,
                np.asarray(out_ops, dtype=np.int32))


class TracedCondition:
    """The result of comparing traced values: usable in ``where`` only."""

    def __init__(self, value):
        self.value = value

    def __bool__(self):
        raise TypeError("a comparison of traced values cannot decide a Python `if`: both arms must be recorded — "
                        "write skeres_amd.tape.where(condition, then, otherwise)")


class Traced:
    """The recording T.  Arithmetic as on rotation.Jet / float; comparisons give a TracedCondition."""
    __slots__ = ("rec", "operand")

    def __init__(self, rec, operand):
        self.rec, self.operand = rec, operand

    def __neg__(self):
        return self.rec.emit(NEG, self)

    def __pos__(self):
        return self

    def __add__(self, o):
        return self.rec.emit(ADD, self, o)

    def __radd__(self, o):
        return self.rec.emit(ADD, o, self)

    def __sub__(self, o):
        return self.rec.emit(SUB, self, o)

    def __rsub__(self, o):
        return self.rec.emit(SUB, o, self)

    def __mul__(self, o):
        return self.rec.emit(MUL, self, o)

    def __rmul__(self, o):
        return self.rec.emit(MUL, o, self)

    def __truediv__(self, o):
        return self.rec.emit(DIV, self, o)

    def __rtruediv__(self, o):
        return self.rec.emit(DIV, o, self)

    def __pow__(self, p):
        if isinstance(p, (int, np.integer)) and 0 <= int(p) <= 8:
            if p == 0:
                return self.rec.const(1.0)
            y = self
            for _ in range(int(p) - 1):
                y = y * self
            return y
        return exp(p * log(self))  # spire's fpow for a positive base

    def __abs__(self):
        return self.rec.emit(ABS, self)

    def __lt__(self, o):
        return TracedCondition(self.rec.emit(LT, self, o))

    def __le__(self, o):
        return TracedCondition(self.rec.emit(LE, self, o))

    def __gt__(self, o):
        return TracedCondition(self.rec.emit(LT, o, self))

    def __ge__(self, o):
        return TracedCondition(self.rec.emit(LE, o, self))

    def sqrt(self):  # spire-style method, as rotation.Jet has it
        return sqrt(self)

    def _tape_unary(self, name):
        return self.rec.emit(_UNARY[name], self)


def where(condition, then, otherwise):
    """``if (condition) then else otherwise`` of a generic body.  ``then`` / ``otherwise``: values, lists of values, or
    callables returning either — pass callables when an arm must not be evaluated on the host unless taken (1 / theta at
    theta = 0 raises on floats).  Recorded: BOTH arms go on the tape and a select picks, element by element."""
    if isinstance(condition, TracedCondition):
        a = then() if callable(then) else then
        b = otherwise() if callable(otherwise) else otherwise
        rec = condition.value.rec
        if isinstance(a, (list, tuple)):
            if len(a) != len(b):
                raise ValueError("the two arms of a where() return %d and %d values" % (len(a), len(b)))
            return [rec.emit(SELECT, condition.value, u, v) for u, v in zip(a, b)]
        return rec.emit(SELECT, condition.value, a, b)
    arm = then if condition else otherwise
    return arm() if callable(arm) else arm


def _generic(name):
    def f(x):
        if isinstance(x, Traced):
            return x._tape_unary(name)
        from . import rotation
        return getattr(rotation, name)(x)
    f.__name__ = name
    return f


sqrt, exp, log, sin, cos, tan, asin, acos, atan = (_generic(n) for n in ("sqrt", "exp", "log", "sin", "cos", "tan", "asin", "acos", "atan"))


def atan2(y, x):
    for v in (y, x):
        if isinstance(v, Traced):
            return v.rec.emit(ATAN2, y, x)
    from . import rotation
    return rotation.atan2(y, x)


def record(functor, N, num_captured):
    """Runs ``functor.apply`` once on traced parameters; returns what sk_cost_function_new_tape takes."""
    rec = Recorder()
    x, k = [], 0
    for n in N:
        x.append([Traced(rec, (PARAMETER, k + j)) for j in range(n)])
        k += n
    functor._traced_captured = [Traced(rec, (CAPTURED, i)) for i in range(num_captured)]
    try:
        y = functor.apply(*x)
    finally:
        functor._traced_captured = None
    if len(y) != functor.kNumResiduals:
        raise ValueError("the functor returned %d residuals while recording, kNumResiduals is %d (an empty result — the "
                         "reference's failure signal — cannot be recorded)" % (len(y), functor.kNumResiduals))
    return rec.finish(list(y))
