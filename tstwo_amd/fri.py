"""FriOps on the GPU (packages/core/src/fri.ts:93-192; backend/cpu/fri.ts:23-164)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .backend import HipColumn, SecureColumnByCoords, _vp
from .circle import CircleDomain, LineDomain, bit_reverse_index
from .fields import QM31, as_q4
from .poly import LineEvaluation, SecureEvaluation, TwiddleTree

FOLD_STEP = 1
CIRCLE_TO_LINE_FOLD_STEP = 1


def _explicit_inverse_twiddles(points_coord) -> HipColumn:
    """Tiny host-side fallback for domains that are not a doubling of a precomputed tree (and circle folds
    below log 3): the n/2 inverses domain.at(bitrev(2i)).{x,y}^-1 the reference computes per element."""
    return HipColumn(np.array([c.inverse().value for c in points_coord], dtype=np.uint32))


class HipFriOps:
    """fold_line / fold_circle_into_line / decompose with the reference's signatures and error texts."""

    @staticmethod
    def fold_line(eval_: LineEvaluation, alpha, twiddles: TwiddleTree | None = None) -> LineEvaluation:
        n = eval_.len()
        if n < 2:
            raise ValueError("fold_line: Evaluation too small, must have at least 2 elements.")
        domain = eval_.domain()
        k = domain.logSize()
        out = SecureColumnByCoords.uninitialized(n // 2)
        a = L.u32x(as_q4(alpha))
        if twiddles is not None and domain.coset().is_doubling_of(twiddles.rootCoset):
            L.call("tstwo_fri_fold_line", eval_.values.ptrs(), k, _vp(twiddles.itwiddles.ptr), twiddles.log_size, a, out.ptrs())
        else:
            inv = _explicit_inverse_twiddles([domain.at(bit_reverse_index(i << FOLD_STEP, k)) for i in range(n // 2)])
            L.call("tstwo_fri_fold_line_tw", eval_.values.ptrs(), k, _vp(inv.ptr), a, out.ptrs())
        return LineEvaluation(domain.double(), out)

    @staticmethod
    def fold_circle_into_line(dst: LineEvaluation, src: SecureEvaluation, alpha, twiddles: TwiddleTree | None = None) -> None:
        if (src.domain.size() >> CIRCLE_TO_LINE_FOLD_STEP) != dst.len():
            raise ValueError("fold_circle_into_line: Length mismatch between src and dst after considering fold step.")
        domain: CircleDomain = src.domain
        n = domain.log_size()
        a = L.u32x(as_q4(alpha))
        if twiddles is not None and n >= 3 and domain.halfCoset.is_doubling_of(twiddles.rootCoset):
            L.call("tstwo_fri_fold_circle_into_line", dst.values.ptrs(), dst.len(), src.values.ptrs(), n,
                   _vp(twiddles.itwiddles.ptr), twiddles.log_size, a)
        else:
            inv = _explicit_inverse_twiddles([domain.at(bit_reverse_index(i << CIRCLE_TO_LINE_FOLD_STEP, n)).y
                                              for i in range(dst.len())])
            L.call("tstwo_fri_fold_circle_into_line_tw", dst.values.ptrs(), dst.len(), src.values.ptrs(), n, _vp(inv.ptr), a)

    # ---- alpha in device memory (drawn by the device channel): same folds, no host round trip
    @staticmethod
    def can_fold_on_device(domain, twiddles: TwiddleTree | None) -> bool:
        if twiddles is None:
            return False
        if isinstance(domain, CircleDomain):
            return domain.log_size() >= 3 and domain.halfCoset.is_doubling_of(twiddles.rootCoset)
        return domain.logSize() >= 1 and domain.coset().is_doubling_of(twiddles.rootCoset)

    @staticmethod
    def fold_line_dev(eval_: LineEvaluation, alpha_ptr: int, twiddles: TwiddleTree) -> LineEvaluation:
        n = eval_.len()
        if n < 2:
            raise ValueError("fold_line: Evaluation too small, must have at least 2 elements.")
        domain = eval_.domain()
        out = SecureColumnByCoords.uninitialized(n // 2)
        L.call("tstwo_fri_fold_line_dev", eval_.values.ptrs(), domain.logSize(), _vp(twiddles.itwiddles.ptr), twiddles.log_size,
               C.c_void_p(alpha_ptr), out.ptrs())
        return LineEvaluation(domain.double(), out)

    @staticmethod
    def fold_circle_into_line_dev(dst: LineEvaluation, src: SecureEvaluation, alpha_ptr: int, twiddles: TwiddleTree) -> None:
        if (src.domain.size() >> CIRCLE_TO_LINE_FOLD_STEP) != dst.len():
            raise ValueError("fold_circle_into_line: Length mismatch between src and dst after considering fold step.")
        L.call("tstwo_fri_fold_circle_into_line_dev", dst.values.ptrs(), dst.len(), src.values.ptrs(), src.domain.log_size(),
               _vp(twiddles.itwiddles.ptr), twiddles.log_size, C.c_void_p(alpha_ptr))

    @staticmethod
    def decompose(eval_: SecureEvaluation):
        n = eval_.len()
        out = SecureColumnByCoords.uninitialized(n)
        lam = (C.c_uint32 * 4)()
        L.call("tstwo_fri_decompose", eval_.values.ptrs(), n, out.ptrs(), lam)
        return SecureEvaluation(eval_.domain, out), QM31.from_u32_unchecked(*lam)


# free-function aliases matching the reference exports (fri.ts:120,162; backend/cpu/fri.ts:23,61,133)
fold_line = HipFriOps.fold_line
fold_circle_into_line = HipFriOps.fold_circle_into_line
decompose = HipFriOps.decompose
foldLine, foldCircleIntoLine = fold_line, fold_circle_into_line
