"""Host-side scalars M31 / CM31 / QM31 (Python ints) mirroring the reference's field classes
(packages/core/src/fields/{m31,cm31,qm31}.ts).  Only constants and single values live here —
columns never round-trip through these objects (SURVEY.md §7 "object-array boundary")."""
from __future__ import annotations

P = 2147483647


class M31:
    __slots__ = ("value",)

    def __init__(self, value: int):
        if not (0 <= value < P):          # m31.ts:108-113: from_u32_unchecked range-checks in TS
            raise ValueError("M31 value out of range")
        self.value = int(value)

    # constructors (m31.ts:39-44,252-267)
    @staticmethod
    def from_(v: int) -> "M31":
        if v < 0:
            return M31.reduce(2 * P - abs(v))
        return M31.reduce(v)

    @staticmethod
    def from_u32_unchecked(v: int) -> "M31":
        return M31(v)

    @staticmethod
    def reduce(x: int) -> "M31":          # m31.ts:89-101 (x < P^2)
        return M31(((((x >> 31) + x + 1) >> 31) + x) & P)

    @staticmethod
    def partialReduce(x: int) -> "M31":   # m31.ts:60-64
        return M31(x - P if x >= P else x)

    @staticmethod
    def zero() -> "M31":
        return M31(0)

    @staticmethod
    def one() -> "M31":
        return M31(1)

    def add(self, o: "M31") -> "M31":
        return M31((self.value + o.value) % P)

    def sub(self, o: "M31") -> "M31":
        return M31((self.value - o.value) % P)

    def mul(self, o: "M31") -> "M31":
        return M31(self.value * o.value % P)

    def neg(self) -> "M31":
        return M31((-self.value) % P)

    def square(self) -> "M31":
        return self.mul(self)

    def double(self) -> "M31":
        return self.add(self)

    def inverse(self) -> "M31":           # m31.ts:137-142
        if self.value == 0:
            raise ZeroDivisionError("0 has no inverse")
        return M31(pow(self.value, P - 2, P))

    def isZero(self) -> bool:
        return self.value == 0

    def equals(self, o) -> bool:
        return isinstance(o, M31) and self.value == o.value

    __eq__ = equals

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return f"M31({self.value})"


class CM31:
    __slots__ = ("real", "imag")

    def __init__(self, real: M31, imag: M31):
        self.real, self.imag = real, imag

    @staticmethod
    def from_u32_unchecked(a: int, b: int) -> "CM31":
        return CM31(M31(a), M31(b))

    def add(self, o): return CM31(self.real.add(o.real), self.imag.add(o.imag))
    def sub(self, o): return CM31(self.real.sub(o.real), self.imag.sub(o.imag))
    def neg(self): return CM31(self.real.neg(), self.imag.neg())

    def mul(self, o):                     # cm31.ts:139-149
        return CM31(self.real.mul(o.real).sub(self.imag.mul(o.imag)), self.real.mul(o.imag).add(self.imag.mul(o.real)))

    def mulM31(self, m: M31): return CM31(self.real.mul(m), self.imag.mul(m))
    def complexConjugate(self): return CM31(self.real, self.imag.neg())

    def inverse(self):                    # cm31.ts:237-251
        n = self.real.square().add(self.imag.square())
        if n.isZero():
            raise ZeroDivisionError("0 has no inverse")
        ni = n.inverse()
        return CM31(self.real.mul(ni), self.imag.neg().mul(ni))

    def tup(self): return (self.real.value, self.imag.value)
    def equals(self, o): return self.tup() == o.tup()
    __eq__ = equals
    def __hash__(self): return hash(self.tup())
    def __repr__(self): return f"CM31{self.tup()}"


_R = None


class QM31:
    """(a + bi) + (c + di)u, u^2 = 2 + i (qm31.ts:9,29)."""
    __slots__ = ("c0", "c1")

    def __init__(self, c0: CM31, c1: CM31):
        self.c0, self.c1 = c0, c1

    @staticmethod
    def from_u32_unchecked(a, b, c, d) -> "QM31":
        return QM31(CM31.from_u32_unchecked(a, b), CM31.from_u32_unchecked(c, d))

    @staticmethod
    def from_(m: M31) -> "QM31":
        return QM31(CM31(m, M31.zero()), CM31(M31.zero(), M31.zero()))

    @staticmethod
    def from_partial_evals(evals) -> "QM31":
        """qm31.ts:168-174: sum_k evals[k] * basis_k with basis (1, i, u, iu)."""
        res = evals[0]
        for k, e in enumerate(evals[1:], 1):
            res = res.add(e.mul(QM31.from_u32_unchecked(*[1 if j == k else 0 for j in range(4)])))
        return res

    @staticmethod
    def zero(): return QM31.from_u32_unchecked(0, 0, 0, 0)
    @staticmethod
    def one(): return QM31.from_u32_unchecked(1, 0, 0, 0)

    def add(self, o): return QM31(self.c0.add(o.c0), self.c1.add(o.c1))
    def sub(self, o): return QM31(self.c0.sub(o.c0), self.c1.sub(o.c1))
    def neg(self): return QM31(self.c0.neg(), self.c1.neg())

    def mul(self, o):                     # qm31.ts:223-233
        r = CM31.from_u32_unchecked(2, 1)
        return QM31(self.c0.mul(o.c0).add(r.mul(self.c1.mul(o.c1))), self.c0.mul(o.c1).add(self.c1.mul(o.c0)))

    def mulM31(self, m: M31): return QM31(self.c0.mulM31(m), self.c1.mulM31(m))
    def mul_cm31(self, c: CM31): return QM31(self.c0.mul(c), self.c1.mul(c))
    mulCM31 = mul_cm31
    def square(self): return self.mul(self)
    def double(self): return self.add(self)

    def pow(self, e: int):
        r, b = QM31.one(), self
        while e:
            if e & 1:
                r = r.mul(b)
            b = b.mul(b)
            e >>= 1
        return r

    def inverse(self):                    # qm31.ts:282-305
        if self.to_m31_array() == (0, 0, 0, 0):
            raise ZeroDivisionError("0 has no inverse")
        b2 = self.c1.mul(self.c1)
        ib2 = CM31(b2.imag.neg(), b2.real)
        denom = self.c0.mul(self.c0).sub(b2.add(b2).add(ib2))
        di = denom.inverse()
        return QM31(self.c0.mul(di), self.c1.mul(di).neg())

    def complexConjugate(self, ts_compat=None):
        """Rust: (c0, -c1).  The TS port conjugates each CM31 instead (qm31.ts:433-435): ts_compat=True."""
        from .semantics import ts_compat as _resolve
        if _resolve(ts_compat):
            return QM31(self.c0.complexConjugate(), self.c1.complexConjugate())
        return QM31(self.c0, self.c1.neg())

    def to_m31_array(self): return (*self.c0.tup(), *self.c1.tup())
    tup = to_m31_array
    def equals(self, o): return self.tup() == o.tup()
    __eq__ = equals
    def __hash__(self): return hash(self.tup())
    def __repr__(self): return f"QM31{self.tup()}"


def as_q4(v) -> tuple:
    """QM31 | 4-sequence -> 4-tuple of ints."""
    return v.tup() if isinstance(v, QM31) else tuple(int(x) for x in v)
