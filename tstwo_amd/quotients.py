"""QuotientOps.accumulateQuotients and AccumulationOps on the GPU (packages/core/src/backend/cpu/quotients.ts,
constraints.ts:117-128, backend/cpu/accumulation.ts:38-63).  The per-batch constants are tiny and are
computed here on the host exactly as quotientConstants() does; the row loop runs on the device."""
from __future__ import annotations

from dataclasses import dataclass

from . import _lib as L
from .backend import SecureColumnByCoords
from .circle import CircleDomain, CirclePoint
from .fields import CM31, M31, QM31, as_q4
from .poly import HipCircleEvaluation, SecureEvaluation
from .semantics import ts_compat as _resolve


@dataclass
class ColumnSampleBatch:
    """quotients.ts:19-24."""
    point: CirclePoint                      # CirclePoint<QM31>
    columns_and_values: list                # [(column_index, QM31)]


def complexConjugateLineCoeffs(point: CirclePoint, value: QM31, alpha: QM31, ts_compat=None):
    """constraints.ts:117-128."""
    if point.y == point.y.complexConjugate(ts_compat):
        raise ValueError("Cannot evaluate a line with a single point")
    a = value.complexConjugate(ts_compat).sub(value)
    c = point.complexConjugate(ts_compat).y.sub(point.y)
    b = value.mul(c).sub(a.mul(point.y))
    return alpha.mul(a), alpha.mul(b), alpha.mul(c)


def quotientConstants(sample_batches, random_coeff: QM31, ts_compat=None):
    """columnLineCoeffs + batchRandomCoeffs (quotients.ts:124-152,183-191)."""
    line_coeffs, batch_coeffs = [], []
    for sb in sample_batches:
        alpha, lc = QM31.one(), []
        for _, v in sb.columns_and_values:
            alpha = alpha.mul(random_coeff)
            lc.append(complexConjugateLineCoeffs(sb.point, v, alpha, ts_compat))
        line_coeffs.append(lc)
        batch_coeffs.append(random_coeff.pow(len(sb.columns_and_values)))
    return line_coeffs, batch_coeffs


def marshal_quotient_args(domain: CircleDomain, columns, random_coeff: QM31, sample_batches, ts_compat=None):
    """Host-side part of accumulateQuotients: quotientConstants() + flattening into the C ABI's arrays.
    Returns (vals, args) where args is the argument tuple of tstwo_quotients_accumulate minus the output."""
    ts_compat = _resolve(ts_compat)
    line_coeffs, batch_coeffs = quotientConstants(sample_batches, random_coeff, ts_compat)
    off, cidx, abc, bco, prx, pry, pix, piy = [0], [], [], [], [], [], [], []
    zero = M31.zero()
    for sb, lc, bc in zip(sample_batches, line_coeffs, batch_coeffs):
        for (ci, _), (a, b, c) in zip(sb.columns_and_values, lc):
            cidx.append(ci)
            abc += [*a.tup(), *b.tup(), *c.tup()]
        off.append(len(cidx))
        bco += bc.tup()
        x, y = sb.point.x, sb.point.y
        if ts_compat:
            parts = (CM31(x.c0.real, zero), CM31(y.c0.real, zero), CM31(x.c0.imag, zero), CM31(y.c0.imag, zero))
        else:
            parts = (x.c0, y.c0, x.c1, y.c1)
        for lst, cm in zip((prx, pry, pix, piy), parts):
            lst += cm.tup()
    vals = [c.values if isinstance(c, HipCircleEvaluation) else c for c in columns]
    for v in vals:
        if v.len() != domain.size():
            raise ValueError("column length does not match the domain size")
    args = (domain.halfCoset.initial_index.value, domain.log_size(), L.ptr_array([v.ptr for v in vals]), len(vals),
            len(sample_batches), L.u32x(off), L.u32x(cidx), L.u32x(abc), L.u32x(bco), L.u32x(prx), L.u32x(pry), L.u32x(pix), L.u32x(piy))
    return vals, args


def accumulateQuotients(domain: CircleDomain, columns, random_coeff: QM31, sample_batches, _log_blowup_factor: int = 1,
                        ts_compat=None) -> SecureEvaluation:
    """accumulateQuotients (quotients.ts:52-75).  Default = Rust semantics (QM31 conjugation (c0,-c1); Pr/Pi = the
    c0/c1 parts of the sample point).  ts_compat=True reproduces the TS port's deviations (per-CM31 conjugation,
    qm31.ts:433-435; Pr/Pi taken from c0.real/c0.imag, quotients.ts:168-174) — see DESIGN.md "reference quirks"."""
    ts_compat = _resolve(ts_compat)
    out = SecureColumnByCoords.uninitialized(domain.size())
    if ts_compat:
        _vals, args = marshal_quotient_args(domain, columns, random_coeff, sample_batches, ts_compat)
        L.call("tstwo_quotients_accumulate", *args, out.ptrs())
        return SecureEvaluation(domain, out)
    # Rust semantics: the constants are computed inside the library from the samples (tstwo_quotients_accumulate_samples)
    vals = [c.values if isinstance(c, HipCircleEvaluation) else c for c in columns]
    for v in vals:
        if v.len() != domain.size():
            raise ValueError("column length does not match the domain size")
    off, cidx, points, values = [0], [], [], []
    for sb in sample_batches:
        points += [*sb.point.x.tup(), *sb.point.y.tup()]
        for ci, v in sb.columns_and_values:
            cidx.append(ci)
            values += v.tup()
        off.append(len(cidx))
    try:
        L.call("tstwo_quotients_accumulate_samples", domain.halfCoset.initial_index.value, domain.log_size(),
               L.ptr_array([v.ptr for v in vals]), len(vals), len(sample_batches), L.u32x(off), L.u32x(cidx), L.u32x(points),
               L.u32x(values), L.u32x(as_q4(random_coeff)), out.ptrs())
    except L.TstwoError as e:
        if "single point" in str(e):
            raise ValueError(str(e)) from None                   # complexConjugateLineCoeffs' own error (constraints.ts:120)
        raise
    return SecureEvaluation(domain, out)


def accumulate(column: SecureColumnByCoords, other: SecureColumnByCoords) -> None:
    """accumulation.ts:38-49."""
    if column.len() != other.len():
        raise ValueError("column length mismatch")
    L.call("tstwo_secure_accumulate", column.ptrs(), other.ptrs(), column.len())


def generate_secure_powers(felt: QM31, n_powers: int) -> list:
    """accumulation.ts:52-63 (a handful of scalars: host side)."""
    res, acc = [], QM31.one()
    for _ in range(n_powers):
        res.append(acc)
        acc = acc.mul(felt)
    return res
