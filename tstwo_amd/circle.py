"""Host-side circle-group bookkeeping mirroring the reference (packages/core/src/circle.ts,
poly/circle/{domain,canonic}.ts, poly/line.ts): indices are integers mod 2^31; points are only
materialised for the handful of scalars the launch code needs.  No column data here."""
from __future__ import annotations

from .fields import M31, P, QM31

M31_CIRCLE_LOG_ORDER = 31
_MASK = (1 << 31) - 1
MAX_CIRCLE_DOMAIN_LOG_SIZE = M31_CIRCLE_LOG_ORDER - 1


class CirclePoint:
    """circle.ts:19-134 over M31 (ints) or QM31 objects."""
    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x, self.y = x, y

    def add(self, o):
        return CirclePoint(self.x.mul(o.x).sub(self.y.mul(o.y)), self.x.mul(o.y).add(self.y.mul(o.x)))

    def double(self): return self.add(self)
    def conjugate(self): return CirclePoint(self.x, self.y.neg())
    neg = conjugate
    def antipode(self): return CirclePoint(self.x.neg(), self.y.neg())

    def complexConjugate(self, ts_compat=None):
        return CirclePoint(self.x.complexConjugate(ts_compat), self.y.complexConjugate(ts_compat))

    def mul(self, scalar: int, one, zero):
        res, cur = CirclePoint(one, zero), self
        while scalar > 0:
            if scalar & 1:
                res = res.add(cur)
            cur = cur.double()
            scalar >>= 1
        return res

    def equals(self, o): return self.x == o.x and self.y == o.y

    @staticmethod
    def get_random_point(channel) -> "CirclePoint":
        """circle.ts:126-133: a secure-field circle point from one drawn felt t: ((1-t^2)/(1+t^2), 2t/(1+t^2))."""
        t = channel.draw_felt()
        t2 = t.square()
        inv = t2.add(QM31.one()).inverse()
        return CirclePoint(QM31.one().sub(t2).mul(inv), t.double().mul(inv))
    __eq__ = equals
    def __repr__(self): return f"CirclePoint({self.x}, {self.y})"


M31_CIRCLE_GEN = CirclePoint(M31(2), M31(1268011823))                      # circle.ts:137
SECURE_FIELD_CIRCLE_GEN = CirclePoint(QM31.from_u32_unchecked(1, 0, 478637715, 513582971),
                                      QM31.from_u32_unchecked(992285211, 649143431, 740191619, 1186584352))


class CirclePointIndex:
    """circle.ts:152-196."""
    __slots__ = ("value",)

    def __init__(self, value: int):
        self.value = int(value)

    @staticmethod
    def zero(): return CirclePointIndex(0)
    @staticmethod
    def generator(): return CirclePointIndex(1)

    @staticmethod
    def subgroup_gen(log_size: int):
        if log_size > M31_CIRCLE_LOG_ORDER:
            raise ValueError("log_size too large")
        return CirclePointIndex((1 << (M31_CIRCLE_LOG_ORDER - log_size)) & _MASK)

    def reduce(self): return CirclePointIndex(self.value & _MASK)
    def to_point(self): return M31_CIRCLE_GEN.mul(self.value & _MASK, M31.one(), M31.zero())
    def add(self, o): return CirclePointIndex((self.value + o.value) & _MASK)
    def sub(self, o): return CirclePointIndex((self.value - o.value) & _MASK)
    def mul(self, k: int): return CirclePointIndex((self.value * k) & _MASK)
    def neg(self): return CirclePointIndex((-self.value) & _MASK)

    def half(self):
        if self.value & 1:
            raise ValueError("not even")
        return CirclePointIndex(self.value >> 1)

    def __eq__(self, o): return self.value == o.value
    def __repr__(self): return f"CirclePointIndex({self.value})"


class Coset:
    """circle.ts:199-291 (points are computed lazily)."""
    __slots__ = ("initial_index", "log_size", "step_size")

    def __init__(self, initial_index: CirclePointIndex, log_size: int):
        if log_size > M31_CIRCLE_LOG_ORDER:
            raise ValueError("log_size too large")
        self.initial_index = initial_index.reduce()
        self.log_size = log_size
        self.step_size = CirclePointIndex.subgroup_gen(log_size)

    new = staticmethod(lambda initial_index, log_size: Coset(initial_index, log_size))

    @staticmethod
    def subgroup(log_size): return Coset(CirclePointIndex.zero(), log_size)
    @staticmethod
    def odds(log_size): return Coset(CirclePointIndex.subgroup_gen(log_size + 1), log_size)
    @staticmethod
    def half_odds(log_size): return Coset(CirclePointIndex.subgroup_gen(log_size + 2), log_size)

    @property
    def initial(self): return self.initial_index.to_point()
    @property
    def step(self): return self.step_size.to_point()

    def size(self): return 1 << self.log_size
    def logSize(self): return self.log_size

    def double(self):
        if self.log_size <= 0:
            raise ValueError("log_size must be >0 to double")
        return Coset(self.initial_index.mul(2), self.log_size - 1)

    def repeated_double(self, n):
        c = self
        for _ in range(n):
            c = c.double()
        return c

    def is_doubling_of(self, other: "Coset") -> bool:
        return self.log_size <= other.log_size and self.equals(other.repeated_double(other.log_size - self.log_size))

    def equals(self, o):
        return (self.initial_index.value == o.initial_index.value and self.step_size.value == o.step_size.value
                and self.log_size == o.log_size)

    __eq__ = equals

    def index_at(self, i): return self.initial_index.add(self.step_size.mul(i))
    def at(self, i): return self.index_at(i).to_point()
    def shift(self, s: CirclePointIndex): return Coset(self.initial_index.add(s), self.log_size)

    def conjugate(self):
        # Rust negates initial AND step; the TS port only negates the initial index (circle.ts:288-290,
        # SURVEY.md App. B-1).  Nothing on the hot path iterates a conjugated coset, so only the index matters.
        return Coset(self.initial_index.neg(), self.log_size)

    def iter(self):
        cur, step = self.initial, self.step
        for _ in range(self.size()):
            yield cur
            cur = cur.add(step)

    def __repr__(self): return f"Coset(initial={self.initial_index.value}, log_size={self.log_size})"


class CircleDomain:
    """poly/circle/domain.ts:12-148."""
    __slots__ = ("halfCoset",)

    def __init__(self, half_coset: Coset):
        self.halfCoset = half_coset

    new = staticmethod(lambda half_coset: CircleDomain(half_coset))

    def logSize(self): return self.halfCoset.log_size + 1
    log_size = logSize
    def size(self): return 1 << self.logSize()

    def indexAt(self, i: int) -> CirclePointIndex:
        if i < 0:
            raise ValueError("i must be a non-negative integer")
        h = self.halfCoset.size()
        return self.halfCoset.index_at(i) if i < h else self.halfCoset.index_at(i - h).neg()

    index_at = indexAt
    def at(self, i: int): return self.indexAt(i).to_point()
    def isCanonic(self): return self.halfCoset.initial_index.mul(4).value == self.halfCoset.step_size.value

    def split(self, log_parts: int):
        if log_parts > self.halfCoset.log_size:
            raise ValueError("logParts cannot exceed half coset log size")
        sub = CircleDomain(Coset(self.halfCoset.initial_index, self.halfCoset.log_size - log_parts))
        return sub, [self.halfCoset.step_size.mul(i) for i in range(1 << log_parts)]

    def shift(self, s): return CircleDomain(self.halfCoset.shift(s))
    def __eq__(self, o): return self.halfCoset == o.halfCoset


class CanonicCoset:
    """poly/circle/canonic.ts:24-130."""
    __slots__ = ("coset",)

    def __init__(self, log_size: int):
        if log_size <= 0:
            raise ValueError("log_size must be a positive integer")
        self.coset = Coset.odds(log_size)

    new = staticmethod(lambda log_size: CanonicCoset(log_size))

    def log_size(self): return self.coset.log_size
    logSize = log_size
    def size(self): return self.coset.size()
    def half_coset(self): return Coset.half_odds(self.log_size() - 1)
    halfCoset = half_coset
    def circle_domain(self): return CircleDomain(self.half_coset())
    circleDomain = circle_domain
    def index_at(self, i): return self.coset.index_at(i)
    indexAt = index_at
    def at(self, i): return self.coset.at(i)
    def initial_index(self): return self.coset.initial_index          # canonic.ts:78-88
    initialIndex = initial_index
    def step_size(self): return self.coset.step_size
    stepSize = step_size
    def step(self): return self.coset.step


class LineDomain:
    """poly/line.ts:18-115 — x-coordinates of a coset."""
    __slots__ = ("_coset",)

    def __init__(self, coset: Coset, _checked: bool = False):
        if not _checked:
            size = coset.size()
            if size == 2:
                if coset.initial.x.isZero():
                    raise ValueError("coset x-coordinates not unique")
            elif size > 2:
                # ord(initial) must be at least 4 * ord(step) unless initial is the identity (line.ts:44-53)
                def log_order(idx: int) -> int:
                    idx &= _MASK
                    return 0 if idx == 0 else 31 - ((idx & -idx).bit_length() - 1)
                init = coset.initial_index.value
                if init != 0 and not (log_order(init) >= log_order(coset.step_size.value) + 2):
                    raise ValueError("coset x-coordinates not unique")
        self._coset = coset

    new = staticmethod(lambda coset: LineDomain(coset))

    def coset(self): return self._coset
    def size(self): return self._coset.size()
    def logSize(self): return self._coset.log_size
    log_size = logSize
    def at(self, i): return self._coset.at(i).x
    def double(self): return LineDomain(self._coset.double(), _checked=True)
    def __eq__(self, o): return self._coset == o._coset


def bit_reverse_index(i: int, log_size: int) -> int:
    """utils.ts:15-22."""
    r = 0
    for _ in range(log_size):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


import functools  # noqa: E402


@functools.lru_cache(maxsize=64)
def bit_reverse_perm(log_size: int):
    """[bit_reverse_index(i, log_size) for i < 2^log_size] as a read-only int64 array (cached: the host side of a FRI proof asks
    for the same few sizes again and again)."""
    import numpy as np
    idx = np.arange(1 << log_size, dtype=np.int64)
    r = np.zeros_like(idx)
    for _ in range(log_size):
        r = (r << 1) | (idx & 1)
        idx >>= 1
    r.setflags(write=False)
    return r
