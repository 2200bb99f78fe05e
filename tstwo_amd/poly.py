"""PolyOps on the GPU: HipCirclePoly / HipCircleEvaluation with the reference's static-method dispatch
(packages/core/src/backend/cpu/circle.ts:17-208, poly/circle/{poly,evaluation,ops,secure_poly}.ts,
poly/twiddles.ts, poly/line.ts:241-329)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .backend import HipBackend, HipColumn, SecureColumnByCoords, _vp
from .circle import CanonicCoset, CircleDomain, CirclePoint, Coset, LineDomain
from .fields import M31, QM31, as_q4


class TwiddleTree:
    """poly/twiddles.ts:10-29 with device buffers."""

    def __init__(self, root_coset: Coset, twiddles: HipColumn, itwiddles: HipColumn):
        self.rootCoset = root_coset
        self.root_coset = root_coset
        self.twiddles = twiddles
        self.itwiddles = itwiddles

    @property
    def log_size(self) -> int:
        return self.rootCoset.log_size


def get_twiddle_dbls(tree: TwiddleTree, inverse: bool = False) -> list:
    """The SimdBackend's twiddle format (backend/simd/fft/index.ts:161-203, Rust get_twiddle_dbls / get_itwiddle_dbls):
    per layer l, the doubled x-coordinates (as u32, not reduced) of the first half of coset.repeated_double(l) in
    bit-reversed order — i.e. level l of the tree times two.  SimdBackend is a CPU backend, so these are host arrays
    (one download of the tree); lets a SimdBackend reuse a tree generated on the GPU."""
    buf = (tree.itwiddles if inverse else tree.twiddles).to_numpy()
    out, off = [], 0
    for l in range(tree.rootCoset.log_size):
        n = 1 << (tree.rootCoset.log_size - 1 - l)
        out.append((buf[off:off + n].astype(np.uint64) * 2 & 0xFFFFFFFF).astype(np.uint32))
        off += n
    return out


getTwiddleDbls = get_twiddle_dbls


def precompute_twiddles(coset: Coset) -> TwiddleTree:
    """precomputeTwiddles (backend/cpu/circle.ts:210-239), generated on the device."""
    n = coset.size()
    tw, itw = HipColumn.uninitialized(n), HipColumn.uninitialized(n)
    L.call("tstwo_twiddles_build", coset.initial_index.value, coset.log_size, _vp(tw.ptr), _vp(itw.ptr))
    return TwiddleTree(coset, tw, itw)


def _swap57(col: HipColumn) -> None:
    """The reference's log_size == 3 output swap (backend/cpu/circle.ts:123-131,145-151), compat only."""
    v = col.to_numpy()
    v[5], v[7] = v[7], v[5]
    col.buf.upload(v)


class HipCirclePoly:
    """CirclePoly<HipBackend> (poly/circle/poly.ts:9-73) + the PolyOps statics of backend/cpu/circle.ts:39-208."""

    compatLog3Swap = False   # True reproduces the reference's TEMPORARY WORKAROUND output order at log 3

    def __init__(self, coeffs):
        self.coeffs = coeffs if isinstance(coeffs, HipColumn) else HipColumn(coeffs)
        n = self.coeffs.len()
        if n == 0 or n & (n - 1):
            raise ValueError("coeffs length must be a power of two")
        self._log_size = n.bit_length() - 1

    new = staticmethod(lambda coeffs: HipCirclePoly(coeffs))

    def logSize(self) -> int:
        return self._log_size

    log_size = logSize

    # ---- instance API (poly.ts:34-59): dispatch through the class statics
    def evaluate(self, domain: CircleDomain) -> "HipCircleEvaluation":
        tw = type(self).precomputeTwiddles(domain.halfCoset)
        return type(self).evaluate_static(self, domain, tw)

    def evaluateWithTwiddles(self, domain, twiddles): return type(self).evaluate_static(self, domain, twiddles)
    def evalAtPoint(self, point) -> QM31: return type(self).eval_at_point(self, point)
    def extend(self, log_size): return type(self).extend_static(self, log_size)

    # ---- PolyOps statics
    @staticmethod
    def precomputeTwiddles(coset: Coset) -> TwiddleTree:
        return precompute_twiddles(coset)

    @staticmethod
    def eval_at_point(poly: "HipCirclePoly", point: CirclePoint) -> QM31:
        out = (C.c_uint32 * 4)()
        L.call("tstwo_eval_at_point", _vp(poly.coeffs.ptr), poly.logSize(), L.u32x(as_q4(point.x)), L.u32x(as_q4(point.y)), out)
        return QM31.from_u32_unchecked(*out)

    @staticmethod
    def eval_at_point_batch(polys, point: CirclePoint) -> list:
        """eval_at_point of several polynomials of ONE size at one point: one launch sequence, one read-back
        (tstwo_eval_at_point_batch).  Same values as [p.evalAtPoint(point) for p in polys]."""
        polys = list(polys)
        if not polys:
            return []
        lg = polys[0].logSize()
        if any(p.logSize() != lg for p in polys):
            raise ValueError("eval_at_point_batch: polynomials must have one size")
        out = (C.c_uint32 * (4 * len(polys)))()
        L.call("tstwo_eval_at_point_batch", L.ptr_array([p.coeffs.ptr for p in polys]), len(polys), lg,
               L.u32x(as_q4(point.x)), L.u32x(as_q4(point.y)), out)
        return [QM31.from_u32_unchecked(*out[4 * i:4 * i + 4]) for i in range(len(polys))]

    def _n_significant(self) -> int:
        c = self.coeffs.to_numpy()
        nz = np.flatnonzero(c)
        return int(nz[-1]) + 1 if nz.size else 0

    def isInFftSpace(self, log_fft_size: int) -> bool:
        """poly.ts:56-63: at most 2^log_fft_size leading coefficients are non-zero."""
        return self._n_significant() <= (1 << log_fft_size)

    def isInFriSpace(self, log_fft_size: int) -> bool:
        """poly.ts:66-73 (is_in_fri_space): one more coefficient than the FFT space."""
        return self._n_significant() <= (1 << log_fft_size) + 1

    is_in_fft_space, is_in_fri_space = isInFftSpace, isInFriSpace

    @staticmethod
    def extend_static(poly: "HipCirclePoly", log_size: int) -> "HipCirclePoly":
        if log_size < poly.logSize():
            raise ValueError("log size too small")
        out = HipColumn.uninitialized(1 << log_size)
        L.call("tstwo_poly_extend", _vp(poly.coeffs.ptr), poly.logSize(), _vp(out.ptr), log_size)
        return HipCirclePoly(out)

    @staticmethod
    def evaluate_static(poly: "HipCirclePoly", domain: CircleDomain, twiddles: TwiddleTree) -> "HipCircleEvaluation":
        return evaluate_polynomials([poly], domain, twiddles)[0]

    @staticmethod
    def interpolate(eval_: "HipCircleEvaluation", twiddles: TwiddleTree) -> "HipCirclePoly":
        return interpolate_columns([eval_], twiddles)[0]


def _check_tree(domain: CircleDomain, twiddles: TwiddleTree) -> None:
    if not domain.halfCoset.is_doubling_of(twiddles.rootCoset):
        raise ValueError("twiddle tree mismatch")                     # backend/cpu/circle.ts:89-91,140-142


def evaluate_polynomials(polys, domain: CircleDomain, twiddles: TwiddleTree) -> list:
    """PolyOps.evaluatePolynomials (poly/circle/ops.ts:89-101), batched: one launch sequence for all columns."""
    _check_tree(domain, twiddles)
    n = domain.log_size()
    polys = list(polys)
    for p in polys:
        if n < p.logSize():
            raise ValueError("log size too small")
    outs = [HipColumn.uninitialized(1 << n) for _ in polys]
    # extend + evaluate per group of equal-sized polynomials (tstwo_cfft_evaluate_extended never materialises the zero
    # padding when the extension is 1-2 bits); same-size groups share one launch sequence
    by_log = {}
    for i, p in enumerate(polys):
        by_log.setdefault(p.logSize(), []).append(i)
    for lg, idxs in by_log.items():
        L.call("tstwo_cfft_evaluate_extended", L.ptr_array([polys[i].coeffs.ptr for i in idxs]), lg,
               L.ptr_array([outs[i].ptr for i in idxs]), len(idxs), n, domain.halfCoset.initial_index.value,
               _vp(twiddles.twiddles.ptr), twiddles.log_size)
    if HipCirclePoly.compatLog3Swap and n == 3:
        for c in outs:
            _swap57(c)
    return [HipCircleEvaluation(domain, c) for c in outs]


def interpolate_columns(evals, twiddles: TwiddleTree) -> list:
    """PolyOps.interpolateColumns (poly/circle/ops.ts:73-82): every column is interpolated on ITS OWN domain (the default
    calls interpolateWithTwiddles per element); columns sharing a domain share one batched launch sequence."""
    evals = list(evals)
    outs = [None] * len(evals)
    groups = {}
    for i, e in enumerate(evals):
        groups.setdefault((e.domain.log_size(), e.domain.halfCoset.initial_index.value), []).append(i)
    for (n, initial), idxs in groups.items():
        domain = evals[idxs[0]].domain
        _check_tree(domain, twiddles)
        if HipCirclePoly.compatLog3Swap and n == 3:
            cols = [evals[i].values.clone() for i in idxs]
            for c in cols:
                _swap57(c)
            L.call("tstwo_cfft_interpolate", L.ptr_array([c.ptr for c in cols]), len(cols), n, initial,
                   _vp(twiddles.itwiddles.ptr), twiddles.log_size)
        else:
            # value semantics (the evaluations survive) without a separate clone: the first pass reads evals, writes cols
            cols = [HipColumn.uninitialized(1 << n) for _ in idxs]
            L.call("tstwo_cfft_interpolate_to", L.ptr_array([evals[i].values.ptr for i in idxs]), L.ptr_array([c.ptr for c in cols]),
                   len(cols), n, initial, _vp(twiddles.itwiddles.ptr), twiddles.log_size)
        for i, c in zip(idxs, cols):
            outs[i] = HipCirclePoly(c)
    return outs


class HipCircleEvaluation:
    """CircleEvaluation<HipBackend, M31, BitReversedOrder> (poly/circle/evaluation.ts:98-177)."""

    def __init__(self, domain: CircleDomain, values):
        self.domain = domain
        self.values = values if isinstance(values, HipColumn) else HipColumn(values)
        if self.values.len() != domain.size():
            raise ValueError("evaluation length does not match the domain size")

    new = staticmethod(lambda domain, values: HipCircleEvaluation(domain, values))

    @staticmethod
    def precomputeTwiddles(coset): return precompute_twiddles(coset)
    @staticmethod
    def to_cpu(values: HipColumn): return values.toCpu()
    @staticmethod
    def bitReverseColumn(col: HipColumn): HipBackend().bitReverseColumn(col)

    def interpolate(self) -> HipCirclePoly:
        return HipCirclePoly.interpolate(self, precompute_twiddles(self.domain.halfCoset))

    def interpolateWithTwiddles(self, twiddles) -> HipCirclePoly: return HipCirclePoly.interpolate(self, twiddles)
    def toCpu(self): return self.values.toCpu()

    def bitReverse(self) -> "HipCircleEvaluation":
        """evaluation.ts:122-128 (NaturalOrder -> BitReversedOrder): a permuted copy."""
        out = self.values.clone()
        HipBackend().bitReverseColumn(out)
        return HipCircleEvaluation(self.domain, out)

    bitReverseBack = bitReverse                     # evaluation.ts:130-136: the same involution
    bit_reverse, bit_reverse_back = bitReverse, bitReverse

    def deref(self) -> HipColumn:                   # evaluation.ts:174-176
        return self.values

    def coset_sub_evaluation(self, offset: int, step: int) -> "CosetSubEvaluation":
        return CosetSubEvaluation(self.values, offset, step)


class CosetSubEvaluation:
    """CosetSubEvaluation (evaluation.ts:182-196): a strided, wrapping view `values[(offset + i*step) & (len-1)]` of a
    device column.  `at(i)` reads one word; `gather(indices)` fetches many with one device gather."""

    def __init__(self, evaluation: HipColumn, offset: int, step: int):
        self.evaluation, self.offset, self.step = evaluation, offset, step

    def _idx(self, index: int) -> int:
        return (self.offset + index * self.step) & (self.evaluation.len() - 1)

    def at(self, index: int) -> M31:
        return self.evaluation.at(self._idx(index))

    get = at

    def gather(self, indices) -> list:
        idx = [self._idx(i) for i in indices]
        if not idx:
            return []
        out = np.empty(len(idx), dtype=np.uint32)
        L.call("tstwo_gather_words", L.ptr_array([self.evaluation.ptr] * len(idx)), (C.c_uint64 * len(idx))(*idx), 1, len(idx),
               out.ctypes.data_as(L.u32p))
        return [M31(int(v)) for v in out]


class SecureEvaluation:
    """SecureEvaluation<HipBackend, BitReversedOrder> (poly/circle/secure_poly.ts:46-81): 4 coordinate columns."""

    def __init__(self, domain: CircleDomain, values: SecureColumnByCoords):
        if values.len() != domain.size():
            raise ValueError("evaluation length does not match the domain size")
        self.domain, self.values = domain, values

    def len(self): return self.values.len()

    def intoCoordinateEvals(self) -> list:
        """secure_poly.ts:61-64."""
        return [HipCircleEvaluation(self.domain, c) for c in self.values.columns]

    def interpolateWithTwiddles(self, twiddles) -> "SecureCirclePoly":
        """secure_poly.ts:73-80: the 4 coordinate columns are interpolated in one batched launch sequence."""
        return SecureCirclePoly(interpolate_columns(self.intoCoordinateEvals(), twiddles))

    into_coordinate_evals, interpolate_with_twiddles = intoCoordinateEvals, interpolateWithTwiddles


class SecureCirclePoly:
    """SecureCirclePoly<HipBackend> (poly/circle/secure_poly.ts:11-44): a QM31 polynomial as 4 coordinate CirclePolys."""

    def __init__(self, polys):
        polys = list(polys)
        if len(polys) != 4:
            raise ValueError("SecureCirclePoly needs 4 coordinate polynomials")
        self.polys = polys

    def __iter__(self): return iter(self.polys)
    def __getitem__(self, i): return self.polys[i]
    def __len__(self): return 4
    def logSize(self) -> int: return self.polys[0].logSize()
    log_size = logSize
    def intoCoordinatePolys(self) -> list: return self.polys
    into_coordinate_polys = intoCoordinatePolys

    def evalColumnsAtPoint(self, point) -> list:
        """secure_poly.ts:20-22 — one batched launch for the 4 coordinates."""
        return HipCirclePoly.eval_at_point_batch(self.polys, point)

    def evalAtPoint(self, point, ts_compat=None) -> QM31:
        """Rust: from_partial_evals of the coordinate evaluations.  The TS port returns coordinate 0 only
        (secure_poly.ts:14-18): ts_compat=True reproduces that."""
        cols = self.evalColumnsAtPoint(point)
        from .semantics import ts_compat as _resolve
        return cols[0] if _resolve(ts_compat) else QM31.from_partial_evals(cols)

    eval_at_point, eval_columns_at_point = evalAtPoint, evalColumnsAtPoint

    def evaluateWithTwiddles(self, domain: CircleDomain, twiddles) -> SecureEvaluation:
        """secure_poly.ts:28-39."""
        evs = evaluate_polynomials(self.polys, domain, twiddles)
        return SecureEvaluation(domain, SecureColumnByCoords([e.values for e in evs]))

    evaluate_with_twiddles = evaluateWithTwiddles


def domain_line_twiddles_from_tree(domain, twiddle_buffer: HipColumn) -> list:
    """domainLineTwiddlesFromTree (poly/utils.ts:78-100): the per-layer slices of a twiddle tree for a CircleDomain or
    LineDomain, largest layer first, as (word_offset, length) views into the device buffer — nothing is copied."""
    log = domain.halfCoset.log_size if isinstance(domain, CircleDomain) else domain.logSize()
    total = twiddle_buffer.len()
    if (1 << log) > total:
        raise ValueError("Not enough twiddles!")
    out = []
    for i in range(log):
        ln = 1 << i
        out.insert(0, (total - 2 * ln, ln))
    return out


domainLineTwiddlesFromTree = domain_line_twiddles_from_tree


class LineEvaluation:
    """LineEvaluation<HipBackend> (poly/line.ts:241-329): QM31 evaluations on a LineDomain, bit-reversed order."""

    def __init__(self, domain: LineDomain, values: SecureColumnByCoords):
        if values.len() != domain.size():
            raise ValueError("evaluation length does not match the domain size")
        self._domain, self.values = domain, values

    new = staticmethod(lambda domain, values: LineEvaluation(domain, values))

    @staticmethod
    def new_zero(domain: LineDomain): return LineEvaluation(domain, SecureColumnByCoords.zeros(domain.size()))
    def domain(self): return self._domain
    def len(self): return self.values.len()
