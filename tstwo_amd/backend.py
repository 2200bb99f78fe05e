"""HipBackend — the MI355X twin of the reference's CpuBackend behind the same Backend / Column /
ColumnOps surface (packages/core/src/backend/index.ts:12-31,53-74; backend/cpu/index.ts:18-157).

Columns live in HBM as little-endian u32 words (M31.intoSlice layout); `M31` objects are only created
by at()/toCpu().  Everything computes on the GPU through the C ABI (tstwo_amd._lib); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .circle import CanonicCoset
from .fields import M31, P, QM31, as_q4


def _vp(ptr):
    return C.c_void_p(ptr)


def _as_u32(data) -> np.ndarray:
    """Accepts numpy arrays, ints or M31 objects (value semantics: always copies)."""
    if isinstance(data, np.ndarray):
        arr = np.array(data, dtype=np.uint32, copy=True)
    else:
        arr = np.fromiter((d.value if isinstance(d, M31) else int(d) for d in data), dtype=np.uint32)
    if arr.size and int(arr.max()) >= P:
        raise ValueError("M31 value out of range")
    return arr


class HipColumn:
    """Column<M31> in device memory (backend/index.ts:53-74; CpuColumn backend/cpu/index.ts:85-157)."""

    def __init__(self, data=None, *, _buf: L.DeviceBuffer | None = None, _len: int = 0):
        if _buf is not None:
            self.buf, self._len = _buf, _len
        else:
            arr = _as_u32(data if data is not None else [])
            self.buf = L.DeviceBuffer(max(arr.nbytes, 16))
            self._len = int(arr.size)
            if arr.size:
                self.buf.upload(arr)

    @property
    def ptr(self) -> int:
        return self.buf.ptr

    @staticmethod
    def zeros(length: int) -> "HipColumn":
        b = L.DeviceBuffer(max(4 * length, 16))
        b.zero()
        return HipColumn(_buf=b, _len=length)

    @staticmethod
    def uninitialized(length: int) -> "HipColumn":
        return HipColumn(_buf=L.DeviceBuffer(max(4 * length, 16)), _len=length)

    @staticmethod
    def from_numpy(arr: np.ndarray) -> "HipColumn":
        return HipColumn(arr)

    def clone(self) -> "HipColumn":
        out = HipColumn.uninitialized(self._len)
        L.call("tstwo_copy", _vp(out.ptr), _vp(self.ptr), 4 * self._len)
        return out

    def len(self) -> int:
        return self._len

    __len__ = len

    def isEmpty(self) -> bool:
        return self._len == 0

    def to_numpy(self) -> np.ndarray:
        return self.buf.download(np.uint32, self._len) if self._len else np.empty(0, dtype=np.uint32)

    def toCpu(self) -> list:
        return [M31(int(v)) for v in self.to_numpy()]

    def _check(self, i: int):
        if not (0 <= i < self._len):
            raise IndexError(f"Index {i} out of bounds for column of length {self._len}")

    def at(self, i: int) -> M31:
        self._check(i)
        return M31(int(self.buf.download(np.uint32, 1, 4 * i)[0]))

    def set(self, i: int, v) -> None:
        self._check(i)
        self.buf.upload(np.array([v.value if isinstance(v, M31) else int(v)], dtype=np.uint32), 4 * i)


class SecureColumnByCoords:
    """SoA of 4 M31 columns holding QM31 values (fields/secure_columns.ts:124-217), device resident."""

    def __init__(self, columns):
        columns = list(columns)
        if len(columns) != 4 or len({c.len() for c in columns}) != 1:
            raise ValueError("SecureColumnByCoords needs 4 coordinate columns of one length")
        self.columns = columns

    @staticmethod
    def zeros(n): return SecureColumnByCoords([HipColumn.zeros(n) for _ in range(4)])
    @staticmethod
    def uninitialized(n): return SecureColumnByCoords([HipColumn.uninitialized(n) for _ in range(4)])

    @staticmethod
    def from_(values) -> "SecureColumnByCoords":
        vals = [as_q4(v) for v in values]
        return SecureColumnByCoords([HipColumn(np.array([v[k] for v in vals], dtype=np.uint32)) for k in range(4)])

    @staticmethod
    def from_numpy(cols4) -> "SecureColumnByCoords":
        return SecureColumnByCoords([HipColumn(c) for c in cols4])

    def len(self): return self.columns[0].len()
    __len__ = len
    def isEmpty(self): return self.len() == 0
    def at(self, i) -> QM31: return QM31.from_u32_unchecked(*[c.at(i).value for c in self.columns])

    def set(self, i, v) -> None:
        for c, x in zip(self.columns, as_q4(v)):
            c.set(i, x)

    def gather(self, positions) -> list:
        """[self.at(p) for p in positions] with one device gather (tstwo_gather_words) instead of 4 reads per element."""
        positions = list(positions)
        if not positions:
            return []
        n, k = self.len(), len(positions)
        for p in positions:
            if p < 0 or p >= n:
                raise IndexError(f"Index {p} out of bounds for column of length {n}")
        srcs = (L.vp * (4 * k))(*[c.ptr for c in self.columns for _ in range(k)])
        idx = (C.c_uint64 * (4 * k))(*(positions * 4))
        out = np.empty(4 * k, dtype=np.uint32)
        L.call("tstwo_gather_words", srcs, idx, 1, 4 * k, out.ctypes.data_as(L.u32p))
        rows = out.reshape(4, k).T.tolist()
        return [QM31.from_u32_unchecked(*r) for r in rows]

    def to_numpy(self):
        # the four coordinate columns in one round trip (tstwo_download_many; large columns are fetched one by one inside it)
        if all(isinstance(c, HipColumn) for c in self.columns) and self.columns[0]._len:
            return L.download_many([(c.buf.ptr, c._len) for c in self.columns])
        return [c.to_numpy() for c in self.columns]
    def to_vec(self): return [QM31.from_u32_unchecked(*map(int, t)) for t in zip(*self.to_numpy())]
    toCpu = to_vec
    def ptrs(self): return L.p4([c.ptr for c in self.columns])
    def clone(self): return SecureColumnByCoords([c.clone() for c in self.columns])


class HipBackend:
    """Backend (backend/index.ts:12-31).  One instance per process = one GPU (LOCAL_RANK)."""

    name = "HipBackend"

    def __init__(self, device: int | None = None):
        if device is not None:
            L.init(device)
        else:
            L.ensure_init()

    # --- ColumnOps
    def bitReverseColumn(self, col) -> None:
        """In-place bit reversal (backend/cpu/index.ts:62-79); throws "length is not power of two"."""
        cols = col.columns if isinstance(col, SecureColumnByCoords) else [col]
        L.call("tstwo_bit_reverse", L.ptr_array([c.ptr for c in cols]), len(cols), cols[0].len())

    def createBaseFieldColumn(self, data) -> HipColumn:
        return HipColumn(data)

    def createSecureFieldColumn(self, data) -> SecureColumnByCoords:
        return SecureColumnByCoords.from_(data)

    # --- field column ops (bench/m31.bench.ts workload, on columns)
    def _binop(self, name, a: HipColumn, b: HipColumn | None) -> HipColumn:
        if b is not None and a.len() != b.len():
            raise ValueError("column length mismatch")
        out = HipColumn.uninitialized(a.len())
        if b is None:
            L.call(name, _vp(a.ptr), _vp(out.ptr), a.len())
        else:
            L.call(name, _vp(a.ptr), _vp(b.ptr), _vp(out.ptr), a.len())
        return out

    def add(self, a, b): return self._binop("tstwo_m31_add", a, b)
    def sub(self, a, b): return self._binop("tstwo_m31_sub", a, b)
    def mul(self, a, b): return self._binop("tstwo_m31_mul", a, b)
    def neg(self, a): return self._binop("tstwo_m31_neg", a, None)

    def batchInverse(self, col):
        """batchInverse (fields/fields.ts:165-180); throws "0 has no inverse" when an element is zero."""
        if isinstance(col, SecureColumnByCoords):
            out = SecureColumnByCoords.uninitialized(col.len())
            L.call("tstwo_qm31_batch_inverse", col.ptrs(), out.ptrs(), col.len())
            return out
        out = HipColumn.uninitialized(col.len())
        L.call("tstwo_m31_batch_inverse", _vp(col.ptr), _vp(out.ptr), col.len())
        return out

    def secureMul(self, a: SecureColumnByCoords, b: SecureColumnByCoords) -> SecureColumnByCoords:
        out = SecureColumnByCoords.uninitialized(a.len())
        L.call("tstwo_qm31_mul", a.ptrs(), b.ptrs(), out.ptrs(), a.len())
        return out

    # --- helpers used by bench / drivers
    @staticmethod
    def canonic_half_coset_initial(log_size: int) -> int:
        return CanonicCoset(log_size).half_coset().initial_index.value

    def sync(self) -> None:
        L.sync()


def shard_columns(n_columns: int, world_size: int, rank: int) -> list:
    """Column sharding of SURVEY.md §8(e): rank g owns the contiguous block [g*C/W, (g+1)*C/W)."""
    per = n_columns // world_size
    extra = n_columns % world_size
    start = rank * per + min(rank, extra)
    return list(range(start, start + per + (1 if rank < extra else 0)))
