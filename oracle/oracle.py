"""ctypes wrapper over the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from tstwo_amd/ (the product).  See
oracle/tstwo_oracle.h for what each function restates (reference file:line).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
P = 2147483647

ERRORS = {
    1: "0 has no inverse",
    2: "length is not power of two",
    3: "Not enough twiddles!",
    4: "fold_line: Evaluation too small, must have at least 2 elements.",
    5: "fold_circle_into_line: Length mismatch between src and dst after considering fold step.",
    6: "bad argument",
}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, f"oracle error {code}"))
        self.code = code


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (Makefile next to this file)."""
    srcs = [os.path.join(_HERE, f) for f in ("tstwo_oracle.c", "tstwo_oracle_mt.c", "tstwo_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


class CM31(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32)]


class QM31(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32), ("d", C.c_uint32)]

    def tup(self):
        return (self.a, self.b, self.c, self.d)


class Point(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]


class SPoint(C.Structure):
    _fields_ = [("x", QM31), ("y", QM31)]


class SampleBatch(C.Structure):
    _fields_ = [("point", SPoint), ("n_cols", C.c_size_t), ("col_idx", C.POINTER(C.c_uint32)),
                ("values", C.POINTER(QM31))]


_lib = None
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
P4 = u32p * 4


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u32, u64, i32, sz = C.c_uint32, C.c_uint64, C.c_int32, C.c_size_t
        sig = {
            "orc_m31_reduce": (u32, [u64]), "orc_m31_partial_reduce": (u32, [u32]),
            "orc_m31_from_i32": (u32, [i32]), "orc_m31_from_u32": (u32, [u32]),
            "orc_m31_add": (u32, [u32, u32]), "orc_m31_sub": (u32, [u32, u32]),
            "orc_m31_neg": (u32, [u32]), "orc_m31_mul": (u32, [u32, u32]),
            "orc_m31_pow2147483645": (u32, [u32]), "orc_m31_inverse": (C.c_int, [u32, u32p]),
            "orc_cm31_add": (CM31, [CM31, CM31]), "orc_cm31_sub": (CM31, [CM31, CM31]),
            "orc_cm31_neg": (CM31, [CM31]), "orc_cm31_mul": (CM31, [CM31, CM31]),
            "orc_cm31_inverse": (C.c_int, [CM31, C.POINTER(CM31)]),
            "orc_qm31_add": (QM31, [QM31, QM31]), "orc_qm31_sub": (QM31, [QM31, QM31]),
            "orc_qm31_neg": (QM31, [QM31]), "orc_qm31_mul": (QM31, [QM31, QM31]),
            "orc_qm31_mul_m31": (QM31, [QM31, u32]), "orc_qm31_mul_cm31": (QM31, [QM31, CM31]),
            "orc_qm31_inverse": (C.c_int, [QM31, C.POINTER(QM31)]),
            "orc_m31_batch_inverse": (C.c_int, [u32p, u32p, sz]),
            "orc_cm31_batch_inverse": (C.c_int, [C.POINTER(CM31), C.POINTER(CM31), sz]),
            "orc_qm31_batch_inverse_soa": (C.c_int, [P4, P4, sz]),
            "orc_m31_col_add": (None, [u32p, u32p, u32p, sz]), "orc_m31_col_sub": (None, [u32p, u32p, u32p, sz]),
            "orc_m31_col_mul": (None, [u32p, u32p, u32p, sz]), "orc_m31_col_neg": (None, [u32p, u32p, sz]),
            "orc_qm31_col_mul_soa": (None, [P4, P4, P4, sz]),
            "orc_bit_reverse_index": (u32, [u32, u32]), "orc_bit_reverse_u32": (C.c_int, [u32p, sz]),
            "orc_point_add": (Point, [Point, Point]), "orc_index_to_point": (Point, [u32]),
            "orc_subgroup_gen": (u32, [u32]), "orc_half_odds_initial": (u32, [u32]), "orc_odds_initial": (u32, [u32]),
            "orc_coset_at": (Point, [u32, u32, u32]), "orc_circle_domain_at": (Point, [u32, u32, u32]),
            "orc_precompute_twiddles": (C.c_int, [u32, u32, u32p, u32p]),
            "orc_cfft_evaluate": (C.c_int, [u32p, u32, u32, u32p, u32, C.c_int]),
            "orc_cfft_interpolate": (C.c_int, [u32p, u32, u32, u32p, u32, C.c_int]),
            "orc_eval_at_point": (QM31, [u32p, u32, SPoint]),
            "orc_fold_line": (C.c_int, [P4, u32, u32, QM31, P4]),
            "orc_fold_circle_into_line": (C.c_int, [P4, sz, P4, u32, u32, QM31]),
            "orc_decompose": (C.c_int, [P4, sz, P4, C.POINTER(QM31)]),
            "orc_blake2s": (None, [C.c_char_p, sz, u8p]),
            "orc_blake2s_compress": (None, [u32p, u32p, u32, u32, u32, u32, u32p]),
            "orc_hash_node": (None, [u8p, u8p, u32p, sz, u8p]),
            "orc_commit_on_layer": (None, [u32, u8p, C.POINTER(u32p), sz, u8p]),
            "orc_merkle_commit": (C.c_int, [C.POINTER(u32p), u32p, sz, u8p, u8p]),
            "orc_qm31_complex_conjugate": (QM31, [QM31]),
            "orc_line_coeffs": (None, [SPoint, QM31, QM31, C.POINTER(QM31)]),
            "orc_accumulate_quotients": (C.c_int, [u32, u32, C.POINTER(u32p), sz, QM31, C.POINTER(SampleBatch), sz, P4]),
            "orc_accumulate_quotients_consts": (C.c_int, [u32, u32, C.POINTER(u32p), sz, C.POINTER(sz), u32p,
                                                          C.POINTER(QM31), C.POINTER(QM31), C.POINTER(CM31),
                                                          C.POINTER(CM31), C.POINTER(CM31), C.POINTER(CM31), P4]),
            "orc_accumulate": (None, [P4, P4, sz]),
            "orc_generate_secure_powers": (None, [QM31, sz, C.POINTER(QM31)]),
            "orc_mt_cfft_evaluate": (C.c_int, [C.POINTER(u32p), sz, u32, u32, u32p, u32, C.c_uint]),
            "orc_mt_merkle_root": (C.c_int, [C.POINTER(u32p), sz, u32, C.c_uint, u8p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# ------------------------------------------------------------------ helpers
def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(u32p)


def _p8(a: np.ndarray):
    return a.ctypes.data_as(u8p)


def _p4(cols):
    return P4(*[_p(c) for c in cols])


def _chk(rc):
    if rc:
        raise OracleError(rc)


def q(t) -> QM31:
    return QM31(*[int(v) for v in t])


def soa_alloc(n):
    return [np.zeros(n, dtype=np.uint32) for _ in range(4)]


# ------------------------------------------------------------------ numpy-level API
def m31_batch_inverse(col):
    col = _u32(col)
    out = np.empty_like(col)
    _chk(lib().orc_m31_batch_inverse(_p(col), _p(out), col.size))
    return out


def qm31_batch_inverse(cols4):
    cols4 = [_u32(c) for c in cols4]
    out = soa_alloc(cols4[0].size)
    _chk(lib().orc_qm31_batch_inverse_soa(_p4(cols4), _p4(out), cols4[0].size))
    return out


def col_op(op, a, b=None):
    a = _u32(a)
    out = np.empty_like(a)
    if op == "neg":
        lib().orc_m31_col_neg(_p(a), _p(out), a.size)
    else:
        b = _u32(b)
        getattr(lib(), f"orc_m31_col_{op}")(_p(a), _p(b), _p(out), a.size)
    return out


def qm31_col_mul(a4, b4):
    a4 = [_u32(c) for c in a4]
    b4 = [_u32(c) for c in b4]
    out = soa_alloc(a4[0].size)
    lib().orc_qm31_col_mul_soa(_p4(a4), _p4(b4), _p4(out), a4[0].size)
    return out


def bit_reverse(v):
    v = _u32(v).copy()
    _chk(lib().orc_bit_reverse_u32(_p(v), v.size))
    return v


def precompute_twiddles(coset_initial, log_size, inverse=True):
    n = 1 << log_size
    buf = np.empty(n, dtype=np.uint32)
    ibuf = np.empty(n, dtype=np.uint32) if inverse else None
    _chk(lib().orc_precompute_twiddles(coset_initial, log_size, _p(buf), _p(ibuf) if inverse else None))
    return buf, ibuf


def cfft_evaluate(coeffs, log_size, half_initial, tw, tw_log, compat=False):
    v = _u32(coeffs).copy()
    tw = _u32(tw)
    _chk(lib().orc_cfft_evaluate(_p(v), log_size, half_initial, _p(tw), tw_log, int(compat)))
    return v


def cfft_interpolate(vals, log_size, half_initial, itw, tw_log, compat=False):
    v = _u32(vals).copy()
    itw = _u32(itw)
    _chk(lib().orc_cfft_interpolate(_p(v), log_size, half_initial, _p(itw), tw_log, int(compat)))
    return v


def eval_at_point(coeffs, log_size, px, py):
    c = _u32(coeffs)
    return lib().orc_eval_at_point(_p(c), log_size, SPoint(q(px), q(py))).tup()


def fold_line(in4, log_n, coset_initial, alpha):
    in4 = [_u32(c) for c in in4]
    out = soa_alloc((1 << log_n) // 2)
    _chk(lib().orc_fold_line(_p4(in4), log_n, coset_initial, q(alpha), _p4(out)))
    return out


def fold_circle_into_line(dst4, src4, log_n, half_initial, alpha):
    dst4 = [_u32(c).copy() for c in dst4]
    src4 = [_u32(c) for c in src4]
    _chk(lib().orc_fold_circle_into_line(_p4(dst4), dst4[0].size, _p4(src4), log_n, half_initial, q(alpha)))
    return dst4


def decompose(in4):
    in4 = [_u32(c) for c in in4]
    out = soa_alloc(in4[0].size)
    lam = QM31()
    _chk(lib().orc_decompose(_p4(in4), in4[0].size, _p4(out), C.byref(lam)))
    return out, lam.tup()


def blake2s(msg: bytes) -> bytes:
    out = np.zeros(32, dtype=np.uint8)
    lib().orc_blake2s(msg, len(msg), _p8(out))
    return out.tobytes()


def blake2s_compress(h, m, count_lo, count_hi, lastblock, lastnode):
    h = _u32(h)
    m = _u32(m)
    out = np.zeros(8, dtype=np.uint32)
    lib().orc_blake2s_compress(_p(h), _p(m), count_lo, count_hi, lastblock, lastnode, _p(out))
    return out


def hash_node(children, values) -> bytes:
    vals = _u32(values)
    out = np.zeros(32, dtype=np.uint8)
    if children is not None:
        l = np.frombuffer(children[0], dtype=np.uint8).copy()
        r = np.frombuffer(children[1], dtype=np.uint8).copy()
        lib().orc_hash_node(_p8(l), _p8(r), _p(vals), vals.size, _p8(out))
    else:
        lib().orc_hash_node(None, None, _p(vals), vals.size, _p8(out))
    return out.tobytes()


def _colptrs(cols):
    cols = [_u32(c) for c in cols]
    arr = (u32p * max(len(cols), 1))(*[_p(c) for c in cols])
    return cols, arr


def commit_on_layer(log_size, prev, cols):
    """prev: None or uint8 array of 2^(log+1)*32 bytes; returns uint8 array [2^log, 32]."""
    cols, arr = _colptrs(cols)
    out = np.zeros((1 << log_size, 32), dtype=np.uint8)
    pp = None
    if prev is not None:
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        pp = _p8(prev)
    lib().orc_commit_on_layer(log_size, pp, arr, len(cols), _p8(out))
    return out


def merkle_commit(cols, log_sizes=None):
    """Returns (layers root-first: list of [2^k,32] uint8 arrays, root bytes)."""
    cols = [_u32(c) for c in cols]
    if log_sizes is None:
        log_sizes = [int(c.size).bit_length() - 1 for c in cols]
    max_log = max(log_sizes) if cols else 0
    _, arr = _colptrs(cols)
    ls = _u32(log_sizes if cols else [0])
    flat = np.zeros(((2 << max_log) - 1, 32), dtype=np.uint8)
    root = np.zeros(32, dtype=np.uint8)
    _chk(lib().orc_merkle_commit(arr, _p(ls), len(cols), _p8(flat), _p8(root)))
    layers = [flat[(1 << k) - 1:(2 << k) - 1] for k in range(max_log + 1)]
    return layers, root.tobytes()


def accumulate_quotients(half_initial, log_size, cols, random_coeff, batches):
    """batches: list of (px, py, [(col_idx, value4), ...]) with QM31 4-tuples (Rust semantics)."""
    cols, arr = _colptrs(cols)
    keep = []
    sb = (SampleBatch * max(len(batches), 1))()
    for i, (px, py, cv) in enumerate(batches):
        idx = (C.c_uint32 * max(len(cv), 1))(*[c for c, _ in cv])
        vals = (QM31 * max(len(cv), 1))(*[q(v) for _, v in cv])
        keep += [idx, vals]
        sb[i] = SampleBatch(SPoint(q(px), q(py)), len(cv), C.cast(idx, u32p), C.cast(vals, C.POINTER(QM31)))
    out = soa_alloc(1 << log_size)
    _chk(lib().orc_accumulate_quotients(half_initial, log_size, arr, len(cols), q(random_coeff), sb, len(batches), _p4(out)))
    return out


def accumulate_quotients_consts(half_initial, log_size, cols, batch_off, col_idx, abc, batch_coeff, prx, pry, pix, piy):
    cols, arr = _colptrs(cols)
    nb = len(batch_coeff)
    off = (C.c_size_t * (nb + 1))(*batch_off)
    cidx = _u32(col_idx)
    abc_a = (QM31 * max(len(abc), 1))(*[q(v) for v in abc])
    bc = (QM31 * max(nb, 1))(*[q(v) for v in batch_coeff])
    mk = lambda xs: (CM31 * max(nb, 1))(*[CM31(int(a), int(b)) for a, b in xs])
    out = soa_alloc(1 << log_size)
    _chk(lib().orc_accumulate_quotients_consts(half_initial, log_size, arr, nb, off, _p(cidx), abc_a, bc,
                                               mk(prx), mk(pry), mk(pix), mk(piy), _p4(out)))
    return out


def accumulate(col4, other4):
    col4 = [_u32(c).copy() for c in col4]
    other4 = [_u32(c) for c in other4]
    lib().orc_accumulate(_p4(col4), _p4(other4), col4[0].size)
    return col4


def generate_secure_powers(felt, n):
    out = (QM31 * max(n, 1))()
    lib().orc_generate_secure_powers(q(felt), n, out)
    return [out[i].tup() for i in range(n)]


# ------------------------------------------------------------------ pthread drivers (tstwo_oracle_mt.c)
def mt_cfft_evaluate(cols, log_size, half_initial, tw, tw_log, threads):
    """In-place evaluate of every column (uint32 arrays, modified), one column per task on `threads` C threads."""
    cols = [c if (isinstance(c, np.ndarray) and c.dtype == np.uint32 and c.flags.c_contiguous) else _u32(c) for c in cols]
    arr = (u32p * max(len(cols), 1))(*[_p(c) for c in cols])
    tw = _u32(tw)
    _chk(lib().orc_mt_cfft_evaluate(arr, len(cols), log_size, half_initial, _p(tw), tw_log, threads))
    return cols


def mt_merkle_root(cols, log_size, threads) -> bytes:
    """Root of MerkleProver.commit over equal-length columns, leaf range sharded over `threads` C threads."""
    cols, arr = _colptrs(cols)
    root = np.zeros(32, dtype=np.uint8)
    _chk(lib().orc_mt_merkle_root(arr, len(cols), log_size, threads, _p8(root)))
    return root.tobytes()
