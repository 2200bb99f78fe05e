"""-m gpu parity tests: the HIP path, called through the C ABI (include/tstwo_hip.h), against the CPU
oracle on the same seeded inputs — bit-exact (all arithmetic is integer)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from conftest import P, column, rand_column
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

from tstwo_amd import _lib as L  # noqa: E402
from gpu_util import dev, dev_empty, host, p4, ptrs, vp  # noqa: E402

OL = orc.lib()


def half_odds(k):
    return OL.orc_half_odds_initial(k)


@pytest.fixture(scope="module", autouse=True)
def _init():
    L.init(0)
    yield
    L.sync()


# ------------------------------------------------------------------ field columns
@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 256, 1 << 12, (1 << 16) + 3, 1 << 20])
def test_m31_elementwise(n):
    a, b = rand_column(1, n), rand_column(2, n)
    # include the edge values 0 and P-1
    a[0], b[0] = 0, P - 1
    if n > 2:
        a[1], b[1] = P - 1, P - 1
    da, db, do = dev(a), dev(b), dev_empty(n)
    for op in ("add", "sub", "mul"):
        L.call(f"tstwo_m31_{op}", vp(da), vp(db), vp(do), n)
        assert (host(do, n) == orc.col_op(op, a, b)).all(), op
    L.call("tstwo_m31_neg", vp(da), vp(do), n)
    assert (host(do, n) == orc.col_op("neg", a)).all()


def test_m31_elementwise_unaligned():
    n = 1001
    a, b = rand_column(3, n + 1), rand_column(4, n + 1)
    da, db, do = dev(a), dev(b), dev_empty(n + 1)
    L.call("tstwo_m31_mul", vp(da, 4), vp(db, 4), vp(do, 4), n)
    assert (host(do, n + 1)[1:] == orc.col_op("mul", a[1:], b[1:])).all()


@pytest.mark.parametrize("n", [1, 4, 5, 17, 1000, 1 << 12, 1 << 20])
def test_m31_batch_inverse(n):
    a = rand_column(5, n, nonzero=True)
    da, do = dev(a), dev_empty(n)
    L.call("tstwo_m31_batch_inverse", vp(da), vp(do), n)
    assert (host(do, n) == orc.m31_batch_inverse(a)).all()


def test_batch_inverse_zero_is_an_error():
    a = rand_column(6, 4096, nonzero=True)
    a[1234] = 0
    da, do = dev(a), dev_empty(4096)
    with pytest.raises(L.TstwoError, match="0 has no inverse"):
        L.call("tstwo_m31_batch_inverse", vp(da), vp(do), 4096)
    a[1234] = 7   # the flag must have been cleared
    da.upload(a)
    L.call("tstwo_m31_batch_inverse", vp(da), vp(do), 4096)
    assert (host(do, 4096) == orc.m31_batch_inverse(a)).all()


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 1 << 14])
def test_qm31_batch_inverse_and_mul(n):
    a = [rand_column(10 + k, n, nonzero=True) for k in range(4)]
    b = [rand_column(20 + k, n) for k in range(4)]
    da, db = [dev(c) for c in a], [dev(c) for c in b]
    do = [dev_empty(n) for _ in range(4)]
    L.call("tstwo_qm31_batch_inverse", p4(da), p4(do), n)
    exp = orc.qm31_batch_inverse(a)
    for k in range(4):
        assert (host(do[k], n) == exp[k]).all()
    L.call("tstwo_qm31_mul", p4(da), p4(db), p4(do), n)
    exp = orc.qm31_col_mul(a, b)
    for k in range(4):
        assert (host(do[k], n) == exp[k]).all()
    L.call("tstwo_secure_accumulate", p4(da), p4(db), n)
    exp = orc.accumulate(a, b)
    for k in range(4):
        assert (host(da[k], n) == exp[k]).all()
    # zero element -> reference error text
    z = [c.copy() for c in a]
    for k in range(4):
        z[k][n // 2] = 0
    dz = [dev(c) for c in z]
    with pytest.raises(L.TstwoError, match="0 has no inverse"):
        L.call("tstwo_qm31_batch_inverse", p4(dz), p4(do), n)


def test_cm31_batch_inverse():
    n = 3000
    a = [rand_column(30 + k, n, nonzero=True) for k in range(2)]
    da, do = [dev(c) for c in a], [dev_empty(n) for _ in range(2)]
    L.call("tstwo_cm31_batch_inverse", L.P2(da[0].ptr, da[1].ptr), L.P2(do[0].ptr, do[1].ptr), n)
    got = [host(do[k], n) for k in range(2)]
    for i in range(0, n, 37):
        r = orc.CM31()
        assert OL.orc_cm31_inverse(orc.CM31(int(a[0][i]), int(a[1][i])), r) == 0
        assert (int(got[0][i]), int(got[1][i])) == (r.a, r.b)


# ------------------------------------------------------------------ bit reverse
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 8, 13, 20])
def test_bit_reverse(log_n):
    n = 1 << log_n
    cols = [rand_column(40 + c, n) for c in range(3)]
    d = [dev(c) for c in cols]
    L.call("tstwo_bit_reverse", ptrs(d), 3, n)
    for c in range(3):
        assert (host(d[c], n) == orc.bit_reverse(cols[c])).all()


@pytest.mark.parametrize("n", [0, 3, 6, 1000])
def test_bit_reverse_not_pow2(n):
    d = [dev(np.zeros(max(n, 1), dtype=np.uint32))]
    with pytest.raises(L.TstwoError, match="length is not power of two"):
        L.call("tstwo_bit_reverse", ptrs(d), 1, n)


# ------------------------------------------------------------------ twiddles
def build_twiddles(log, initial=None):
    initial = half_odds(log) if initial is None else initial
    n = 1 << log
    tw, itw = dev_empty(n), dev_empty(n)
    L.call("tstwo_twiddles_build", initial, log, vp(tw), vp(itw))
    return tw, itw


@pytest.mark.parametrize("log", [0, 1, 2, 3, 5, 8, 12, 16])
def test_twiddles(log, golden):
    tw, itw = build_twiddles(log)
    ebuf, eibuf = orc.precompute_twiddles(half_odds(log), log)
    assert (host(tw, 1 << log) == ebuf).all() and (host(itw, 1 << log) == eibuf).all()
    for e in golden["twiddles"]:
        if e["log"] == log:
            assert hashlib.blake2s(host(tw, 1 << log).tobytes()).hexdigest() == e["digest"]


def test_twiddles_other_cosets():
    for init, log in [(OL.orc_odds_initial(6), 6), (12345, 7)]:
        tw, _ = dev_empty(1 << log), None
        L.call("tstwo_twiddles_build", init, log, vp(tw), C.c_void_p(0))
        ebuf, _ = orc.precompute_twiddles(init, log, inverse=False)
        assert (host(tw, 1 << log) == ebuf).all()


# ------------------------------------------------------------------ CFFT
@pytest.mark.parametrize("n", list(range(1, 17)))
def test_cfft_vs_oracle(n):
    n_cols = 3
    tw_log = max(n - 1, 1)
    tw, itw = build_twiddles(tw_log)
    otw, oitw = orc.precompute_twiddles(half_odds(tw_log), tw_log)
    cols = [rand_column(100 * n + c, 1 << n) for c in range(n_cols)]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), n_cols, n, half_odds(n - 1), vp(tw), tw_log)
    evs = [host(x, 1 << n) for x in d]
    for c in range(n_cols):
        assert (evs[c] == orc.cfft_evaluate(cols[c], n, half_odds(n - 1), otw, tw_log)).all(), f"evaluate log {n} col {c}"
    L.call("tstwo_cfft_interpolate", ptrs(d), n_cols, n, half_odds(n - 1), vp(itw), tw_log)
    for c in range(n_cols):
        assert (host(d[c], 1 << n) == cols[c]).all(), f"interpolate log {n} col {c}"
    # interpolate of arbitrary values (not a round trip) against the oracle
    vals = rand_column(7 * n, 1 << n)
    dv = [dev(vals)]
    L.call("tstwo_cfft_interpolate", ptrs(dv), 1, n, half_odds(n - 1), vp(itw), tw_log)
    assert (host(dv[0], 1 << n) == orc.cfft_interpolate(vals, n, half_odds(n - 1), oitw, tw_log)).all()


@pytest.mark.parametrize("n", [14, 15, 16, 17, 18, 19, 20, 21])
def test_cfft_many_columns_default_tiles(n):
    """Enough columns that the transform takes the default tiles (2^13 contiguous + 2^14 strided, every K from 1 to 8) rather
    than the small tiles chosen for few columns; both directions against the oracle (threaded driver)."""
    n_cols = max(3, (2 * 256 + 1) >> (n - 14)) + 1
    tw, itw = build_twiddles(n - 1)
    otw, oitw = orc.precompute_twiddles(half_odds(n - 1), n - 1)
    cols = [rand_column(1000 * n + c, 1 << n) for c in range(n_cols)]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    exp = orc.mt_cfft_evaluate([c.copy() for c in cols], n, half_odds(n - 1), otw, n - 1, 16)
    for c in range(n_cols):
        assert (host(d[c], 1 << n) == exp[c]).all(), f"evaluate log {n} col {c} of {n_cols}"
    L.call("tstwo_cfft_interpolate", ptrs(d), n_cols, n, half_odds(n - 1), vp(itw), n - 1)
    for c in range(n_cols):
        assert (host(d[c], 1 << n) == cols[c]).all(), f"interpolate log {n} col {c} of {n_cols}"


def test_cfft_golden(golden):
    for e in golden["cfft"]:
        n = e["log"]
        tw_log = max(n - 1, 1)
        tw, _ = build_twiddles(tw_log)
        d = [dev(column(e["seed"], 1 << n))]
        L.call("tstwo_cfft_evaluate", ptrs(d), 1, n, e["half_initial"], vp(tw), tw_log)
        assert hashlib.blake2s(host(d[0], 1 << n).tobytes()).hexdigest() == e["eval_digest"], n


def test_cfft_interpolate_golden(golden):
    """interpolate known answers log 1..10 (inputs travel in the fixture; expected = seeded coefficients)."""
    from conftest import golden_interp_values
    for e in golden["cfft_interpolate"]:
        n = e["log"]
        tw_log = max(n - 1, 1)
        _, itw = build_twiddles(tw_log)
        vals = golden_interp_values(e)
        d = [dev(vals)]
        L.call("tstwo_cfft_interpolate", ptrs(d), 1, n, e["half_initial"], vp(itw), tw_log)
        got = host(d[0], 1 << n)
        assert hashlib.blake2s(got.tobytes()).hexdigest() == e["coeffs_digest"], n
        assert (got == column(e["coeffs_seed"], 1 << n)).all()
        # out-of-place variant: same answer, source untouched
        s, o = [dev(vals)], [dev_empty(1 << n)]
        L.call("tstwo_cfft_interpolate_to", ptrs(s), ptrs(o), 1, n, e["half_initial"], vp(itw), tw_log)
        assert (host(o[0], 1 << n) == got).all() and (host(s[0], 1 << n) == vals).all()


def test_cfft_bigger_tree_and_errors():
    n = 10
    tw, itw = build_twiddles(14)
    otw, _ = orc.precompute_twiddles(half_odds(n - 1), n - 1)
    col = rand_column(55, 1 << n)
    d = [dev(col)]
    L.call("tstwo_cfft_evaluate", ptrs(d), 1, n, half_odds(n - 1), vp(tw), 14)
    assert (host(d[0], 1 << n) == orc.cfft_evaluate(col, n, half_odds(n - 1), otw, n - 1)).all()
    small, _ = build_twiddles(3)
    with pytest.raises(L.TstwoError, match="Not enough twiddles"):
        L.call("tstwo_cfft_evaluate", ptrs(d), 1, n, half_odds(n - 1), vp(small), 3)


@pytest.mark.parametrize("n", [17, 18, 19, 20, 21, 22, 23, 24])
def test_cfft_large_properties(n):
    """BASELINE sizes: round trip, agreement with eval_at_point at sampled domain points (the reference's
    property test), linearity, and oracle equality on one column."""
    tw, itw = build_twiddles(n - 1)
    a, b = rand_column(n, 1 << n), rand_column(n + 1, 1 << n)
    s = orc.col_op("add", a, b)
    d = [dev(a), dev(b), dev(s)]
    L.call("tstwo_cfft_evaluate", ptrs(d), 3, n, half_odds(n - 1), vp(tw), n - 1)
    ea, eb, es = (host(x, 1 << n) for x in d)
    assert (orc.col_op("add", ea, eb) == es).all()                     # linearity
    rng = np.random.default_rng(n)
    for i in rng.integers(0, 1 << n, size=6 if n <= 22 else 2):
        p = OL.orc_circle_domain_at(half_odds(n - 1), n - 1, int(i))
        v = orc.eval_at_point(a, n, (p.x, 0, 0, 0), (p.y, 0, 0, 0))
        assert v == (int(ea[OL.orc_bit_reverse_index(int(i), n)]), 0, 0, 0)
    if n <= 20:
        otw, _ = orc.precompute_twiddles(half_odds(n - 1), n - 1, inverse=False)
        assert (ea == orc.cfft_evaluate(a, n, half_odds(n - 1), otw, n - 1)).all()
    L.call("tstwo_cfft_interpolate", ptrs(d), 3, n, half_odds(n - 1), vp(itw), n - 1)
    assert (host(d[0], 1 << n) == a).all() and (host(d[1], 1 << n) == b).all()


def test_poly_extend_and_eval_at_point(golden):
    for e in golden["eval_at_point"]:
        coeffs = column(e["seed"], 1 << e["log"])
        d = dev(coeffs)
        out = (C.c_uint32 * 4)()
        L.call("tstwo_eval_at_point", vp(d), e["log"], L.u32x(e["point"][0]), L.u32x(e["point"][1]), out)
        assert list(out) == e["value"], e["log"]
    for n in (7, 12, 17):
        coeffs = rand_column(n, 1 << n)
        d = dev(coeffs)
        px, py = golden["eval_at_point"][0]["point"]
        out = (C.c_uint32 * 4)()
        L.call("tstwo_eval_at_point", vp(d), n, L.u32x(px), L.u32x(py), out)
        assert tuple(out) == orc.eval_at_point(coeffs, n, px, py)
    src = rand_column(9, 1 << 5)
    ds, dd = dev(src), dev_empty(1 << 9)
    L.call("tstwo_poly_extend", vp(ds), 5, vp(dd), 9)
    got = host(dd, 1 << 9)
    assert (got[:32] == src).all() and not got[32:].any()
    with pytest.raises(L.TstwoError, match="log size too small"):
        L.call("tstwo_poly_extend", vp(dd), 9, vp(ds), 5)


# ------------------------------------------------------------------ FRI
ALPHA = (19283, 1, 2, 3)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 9, 14])
def test_fold_line(k):
    n = 1 << k
    cols = [rand_column(200 + 4 * k + j, n) for j in range(4)]
    _, itw = build_twiddles(max(k, 1) + 2)     # tree of half_odds(k+2); line domain = its doubling twice
    tw_log = max(k, 1) + 2
    coset_initial = (half_odds(tw_log) << (tw_log - k)) & 0x7FFFFFFF   # root.repeated_double(tw_log-k)
    d = [dev(c) for c in cols]
    o = [dev_empty(n // 2) for _ in range(4)]
    L.call("tstwo_fri_fold_line", p4(d), k, vp(itw), tw_log, L.u32x(ALPHA), p4(o))
    exp = orc.fold_line(cols, k, coset_initial, ALPHA)
    for j in range(4):
        assert (host(o[j], n // 2) == exp[j]).all()


def test_fold_line_golden_and_errors(golden):
    for e in golden["fold_line"]:
        k = e["log"]
        cols = [column(s, 1 << k) for s in e["seeds"]]
        _, itw = build_twiddles(k)              # LineDomain(half_odds(k)) == the tree's root coset
        d = [dev(c) for c in cols]
        o = [dev_empty(max((1 << k) // 2, 1)) for _ in range(4)]
        L.call("tstwo_fri_fold_line", p4(d), k, vp(itw), k, L.u32x(e["alpha"]), p4(o))
        got = [host(o[j], (1 << k) // 2) for j in range(4)]
        assert hashlib.blake2s(b"".join(g.tobytes() for g in got)).hexdigest() == e["out_digest"], k
    d = [dev(np.zeros(1, dtype=np.uint32)) for _ in range(4)]
    with pytest.raises(L.TstwoError, match="fold_line: Evaluation too small"):
        L.call("tstwo_fri_fold_line", p4(d), 0, vp(d[0]), 0, L.u32x(ALPHA), p4(d))


@pytest.mark.parametrize("n", [3, 4, 5, 10, 15])
def test_fold_circle_into_line(n):
    N = 1 << n
    src = [rand_column(300 + 4 * n + j, N) for j in range(4)]
    dst = [rand_column(400 + 4 * n + j, N // 2) for j in range(4)]
    _, itw = build_twiddles(n + 1)
    ds, dd = [dev(c) for c in src], [dev(c) for c in dst]
    L.call("tstwo_fri_fold_circle_into_line", p4(dd), N // 2, p4(ds), n, vp(itw), n + 1, L.u32x(ALPHA))
    # domain half coset = root(half_odds(n+1)).repeated_double(2) -> initial index * 4
    half_initial = (half_odds(n + 1) << 2) & 0x7FFFFFFF
    exp = orc.fold_circle_into_line(dst, src, n, half_initial, ALPHA)
    for j in range(4):
        assert (host(dd[j], N // 2) == exp[j]).all()


def test_fold_circle_golden_small_and_errors(golden):
    for e in golden["fold_circle"]:
        n = e["log"]
        src = [column(s, 1 << n) for s in e["src_seeds"]]
        dst = [column(s, 1 << (n - 1)) for s in e["dst_seeds"]]
        ds, dd = [dev(c) for c in src], [dev(c) for c in dst]
        if n >= 3:
            _, itw = build_twiddles(n - 1)
            L.call("tstwo_fri_fold_circle_into_line", p4(dd), 1 << (n - 1), p4(ds), n, vp(itw), n - 1, L.u32x(e["alpha"]))
        else:   # explicit twiddles: y^-1 of domain.at(bitrev(2i, n))
            inv_y = []
            for i in range(1 << (n - 1)):
                p = OL.orc_circle_domain_at(e["half_initial"], n - 1, OL.orc_bit_reverse_index(2 * i, n))
                inv_y.append(pow(p.y, P - 2, P))
            dt = dev(np.array(inv_y, dtype=np.uint32))
            L.call("tstwo_fri_fold_circle_into_line_tw", p4(dd), 1 << (n - 1), p4(ds), n, vp(dt), L.u32x(e["alpha"]))
        got = [host(dd[j], 1 << (n - 1)) for j in range(4)]
        assert hashlib.blake2s(b"".join(g.tobytes() for g in got)).hexdigest() == e["out_digest"], n
    with pytest.raises(L.TstwoError, match="fold_circle_into_line: Length mismatch"):
        L.call("tstwo_fri_fold_circle_into_line", p4(dd), 3, p4(ds), 4, vp(dd[0]), 4, L.u32x(ALPHA))


@pytest.mark.parametrize("n", [1, 2, 8, 1000 * 0 + 1024, 1 << 16])
def test_decompose(n):
    cols = [rand_column(500 + k, n) for k in range(4)]
    d, o = [dev(c) for c in cols], [dev_empty(n) for _ in range(4)]
    lam = (C.c_uint32 * 4)()
    L.call("tstwo_fri_decompose", p4(d), n, p4(o), lam)
    exp, elam = orc.decompose(cols)
    assert tuple(lam) == elam
    for k in range(4):
        assert (host(o[k], n) == exp[k]).all()


# ------------------------------------------------------------------ Merkle
def merkle_commit(cols, log_sizes):
    max_log = max(log_sizes) if cols else 0
    layers = L.DeviceBuffer(32 * ((2 << max_log) - 1))
    d = [dev(c) for c in cols]
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", ptrs(d), L.u32x(log_sizes), len(cols), vp(layers), root)
    flat = layers.download(np.uint8).reshape(-1, 32)
    return [flat[(1 << k) - 1:(2 << k) - 1] for k in range(max_log + 1)], bytes(root)


def test_qm31_mul_edge_grid():
    """qm31_mul is six 64-bit multiply-add chains with one reduction each (m31.cuh): every coordinate of both operands runs
    over the values where a lazy accumulator or the final conditional subtracts could go wrong (0, 1, 2, 2^30, P-2, P-1)."""
    P = 2**31 - 1
    edge = np.array([0, 1, 2, 1 << 30, P - 2, P - 1], dtype=np.uint32)
    g = np.stack(np.meshgrid(edge, edge, edge, edge, indexing="ij"), axis=-1).reshape(-1, 4)      # 1296 QM31 values
    xi, yi = np.meshgrid(np.arange(len(g)), np.arange(len(g)), indexing="ij")
    x, y = g[xi.ravel()], g[yi.ravel()]                                                            # 1 679 616 pairs
    n = len(x)
    a = [np.ascontiguousarray(x[:, k]) for k in range(4)]
    b = [np.ascontiguousarray(y[:, k]) for k in range(4)]
    da, db = [dev(c) for c in a], [dev(c) for c in b]
    do = [dev_empty(n) for _ in range(4)]
    L.call("tstwo_qm31_mul", p4(da), p4(db), p4(do), n)
    exp = orc.qm31_col_mul(a, b)
    for k in range(4):
        assert (host(do[k], n) == exp[k]).all()


_SUBTREE_SCRIPT = r"""
import hashlib, sys
sys.path[:0] = [{root!r}, {tests!r}]
from test_gpu_capi import rand_column, merkle_commit
log = 19
cols = [rand_column(900 + c, 1 << log) for c in range(4)]
layers, root = merkle_commit(cols, [log] * 4)
print(bytes(root).hex(), hashlib.blake2s(b"".join(l.tobytes() for l in layers)).hexdigest())
"""


@pytest.mark.parametrize("levels", ["0", "3", "4", "2-lane-stride"])
def test_merkle_subtree_levels_agree(levels):
    """TSTWO_MERKLE_SUBTREE (experiments build only; read once per process): every setting (layer per launch, 3 and 4 layers per
    in-lane subtree; 2 is the default the other tests run) must give the oracle's tree."""
    import os
    import subprocess
    import sys
    log = 19
    cols = [rand_column(900 + c, 1 << log) for c in range(4)]
    olayers, oroot = orc.merkle_commit(cols, [log] * 4)
    want = bytes(oroot).hex() + " " + hashlib.blake2s(b"".join(l.tobytes() for l in olayers)).hexdigest()
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    script = _SUBTREE_SCRIPT.format(root=os.path.dirname(tests_dir), tests=tests_dir)
    env = dict(os.environ, TSTWO_MERKLE_SUBTREE=levels, TSTWO_HIP_LIB=L.LIB_EXP_PATH)
    if levels == "2-lane-stride":         # round 3's two-level kernel (lane-strided accesses) instead of the coalesced k_merkle_subtree2c
        env.update(TSTWO_MERKLE_SUBTREE="2", TSTWO_MERKLE_SUBTREE_LANE_STRIDE="1")
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == want


def test_merkle_golden(golden):
    for e in golden["merkle"]:
        cols = [column(e["seed_base"] + i, 1 << lg) for i, lg in enumerate(e["log_sizes"])]
        layers, root = merkle_commit(cols, e["log_sizes"])
        assert root.hex() == e["root"], e["name"]
        assert hashlib.blake2s(b"".join(l.tobytes() for l in layers)).hexdigest() == e["layers_digest"], e["name"]
    e = golden["merkle_lcg"]
    layers, root = merkle_commit([np.array(c, dtype=np.uint32) for c in e["cols"]], e["log_sizes"])
    assert root.hex() == e["root"]
    assert [[bytes(h).hex() for h in l] for l in layers] == e["layers"]


@pytest.mark.parametrize("n_cols,log", [(1, 10), (4, 12), (15, 9), (16, 9), (17, 9), (32, 11), (48, 8), (256, 6), (300, 5), (600, 4),
                                        # 4 equal columns: the fused leaf + quad-level kernels and their size boundaries (9 | 10, 16 | 17)
                                        (4, 1), (4, 2), (4, 3), (4, 6), (4, 9), (4, 10), (4, 11), (4, 16), (4, 17)])
def test_merkle_vs_oracle(n_cols, log):
    cols = [rand_column(600 + c, 1 << log) for c in range(n_cols)]
    layers, root = merkle_commit(cols, [log] * n_cols)
    olayers, oroot = orc.merkle_commit(cols, [log] * n_cols)
    assert root == oroot
    for a, b in zip(layers, olayers):
        assert (a == b).all()


def test_merkle_commit_layer_with_prev_and_columns():
    log = 7
    big = [rand_column(700 + c, 1 << (log + 1)) for c in range(3)]
    small = [rand_column(710 + c, 1 << log) for c in range(20)]
    dprev = dev_empty(8 << (log + 1))
    dbig, dsmall = [dev(c) for c in big], [dev(c) for c in small]
    L.call("tstwo_merkle_commit_layer", log + 1, C.c_void_p(0), ptrs(dbig), 3, vp(dprev))
    out = dev_empty(8 << log)
    L.call("tstwo_merkle_commit_layer", log, vp(dprev), ptrs(dsmall), 20, vp(out))
    prev = orc.commit_on_layer(log + 1, None, big)
    assert (host(dprev, 32 << (log + 1), np.uint8).reshape(-1, 32) == prev).all()
    assert (host(out, 32 << log, np.uint8).reshape(-1, 32) == orc.commit_on_layer(log, prev, small)).all()


def test_merkle_large_property():
    """log 20, 4 columns: root equals the oracle's, and a checksum of all layers matches."""
    log = 20
    cols = [rand_column(800 + c, 1 << log) for c in range(4)]
    layers, root = merkle_commit(cols, [log] * 4)
    olayers, oroot = orc.merkle_commit(cols, [log] * 4)
    assert root == oroot
    assert hashlib.blake2s(b"".join(l.tobytes() for l in layers)).digest() == hashlib.blake2s(b"".join(l.tobytes() for l in olayers)).digest()


# ------------------------------------------------------------------ quotients
def gpu_quotients(half_initial, log, cols, random_coeff, batches):
    """batches: [(px, py, [(col, value4)])] in Rust semantics; constants come from the oracle's line-coeff helper."""
    off, cidx, abc, bcoef, prx, pry, pix, piy = [0], [], [], [], [], [], [], []
    for px, py, cv in batches:
        alpha = (1, 0, 0, 0)
        for ci, v in cv:
            alpha = OL.orc_qm31_mul(orc.q(alpha), orc.q(random_coeff)).tup()
            out = (orc.QM31 * 3)()
            OL.orc_line_coeffs(orc.SPoint(orc.q(px), orc.q(py)), orc.q(v), orc.q(alpha), out)
            for t in out:
                abc += list(t.tup())
            cidx.append(ci)
        off.append(len(cidx))
        bcoef += list(alpha)
        prx += px[:2]; pry += py[:2]; pix += px[2:]; piy += py[2:]
    d = [dev(c) for c in cols]
    o = [dev_empty(1 << log) for _ in range(4)]
    L.call("tstwo_quotients_accumulate", half_initial, log, ptrs(d), len(cols), len(batches), L.u32x(off), L.u32x(cidx),
           L.u32x(abc), L.u32x(bcoef), L.u32x(prx), L.u32x(pry), L.u32x(pix), L.u32x(piy), p4(o))
    return [host(x, 1 << log) for x in o]


def test_quotients_golden(golden):
    e = golden["quotients"][1]
    n = e["log"]
    cols = [column(s, 1 << n) for s in e["col_seeds"]]
    batches = [(b["point"][0], b["point"][1], [(ci, v) for ci, v in b["cols"]]) for b in e["batches"]]
    got = gpu_quotients(e["half_initial"], n, cols, e["random_coeff"], batches)
    assert hashlib.blake2s(b"".join(g.tobytes() for g in got)).hexdigest() == e["out_digest"]
    e0 = golden["quotients"][0]
    n = e0["log"]
    coeffs = column(e0["coeffs_seed"], 1 << e0["poly_log"])
    ext = np.concatenate([coeffs, np.zeros((1 << n) - coeffs.size, dtype=np.uint32)])
    otw, _ = orc.precompute_twiddles(half_odds(n - 1), n - 1, inverse=False)
    ev = orc.cfft_evaluate(ext, n, e0["half_initial"], otw, n - 1)
    got = gpu_quotients(e0["half_initial"], n, [ev], e0["random_coeff"], [(e0["point"][0], e0["point"][1], [(0, e0["value"])])])
    assert [g.tolist() for g in got] == e0["out"]


@pytest.mark.parametrize("log", [1, 2, 3, 6, 12, 16])
def test_quotients_vs_oracle(log, golden):
    px, py = golden["eval_at_point"][0]["point"]
    n_cols = 5
    cols = [rand_column(900 + log * 8 + c, 1 << log) for c in range(n_cols)]
    vals = [tuple(int(x) for x in rand_column(950 + j, 4)) for j in range(6)]
    py2 = OL.orc_qm31_mul(orc.q(py), orc.q(py)).tup()
    batches = [(px, py, [(0, vals[0]), (3, vals[1]), (4, vals[2])]), (py, py2, [(1, vals[3])]), (px, py2, [(2, vals[4]), (0, vals[5])])]
    got = gpu_quotients(half_odds(log - 1), log, cols, (1, 2, 3, 4), batches)
    exp = orc.accumulate_quotients(half_odds(log - 1), log, cols, (1, 2, 3, 4), batches)
    for k in range(4):
        assert (got[k] == exp[k]).all()


@pytest.mark.parametrize("log_poly,log_size", [(11, 13), (12, 13), (13, 15), (14, 15), (16, 18), (17, 18), (18, 20), (20, 22), (21, 22),
                                               (22, 24), (10, 13), (13, 13), (5, 7), (9, 12)])
def test_cfft_evaluate_extended_matches_extend_then_evaluate(log_poly, log_size):
    """tstwo_cfft_evaluate_extended == tstwo_poly_extend + tstwo_cfft_evaluate bit for bit (fused 1- and 2-bit extensions on
    the tiled path, fallback elsewhere), and == the oracle at the sizes it finishes quickly."""
    n_cols = 3 if log_size <= 20 else 2
    half = 1 << (31 - (log_size + 1))
    tw = L.DeviceBuffer(4 << (log_size - 1))
    L.call("tstwo_twiddles_build", half, log_size - 1, vp(tw), vp(None))
    polys = [rand_column(21000 + 7 * log_size + c, 1 << log_poly) for c in range(n_cols)]
    src = [dev(p) for p in polys]
    ref = [L.DeviceBuffer(4 << log_size) for _ in polys]
    for s_, r in zip(src, ref):
        L.call("tstwo_poly_extend", vp(s_), log_poly, vp(r), log_size)
    L.call("tstwo_cfft_evaluate", L.ptr_array([r.ptr for r in ref]), n_cols, log_size, half, vp(tw), log_size - 1)
    out = [L.DeviceBuffer(4 << log_size) for _ in polys]
    for o in out:
        L.call("tstwo_zero", vp(o), 4 << log_size)
    L.call("tstwo_cfft_evaluate_extended", L.ptr_array([s_.ptr for s_ in src]), log_poly, L.ptr_array([o.ptr for o in out]), n_cols,
           log_size, half, vp(tw), log_size - 1)
    for o, r, s_, p in zip(out, ref, src, polys):
        assert (o.download() == r.download()).all()
        assert (s_.download() == p).all()                       # the polynomial is read-only
    if log_size <= 18:
        otw, _ = orc.precompute_twiddles(half, log_size - 1, inverse=False)
        ext = np.concatenate([polys[0], np.zeros((1 << log_size) - polys[0].size, dtype=np.uint32)])
        assert (out[0].download() == orc.cfft_evaluate(ext, log_size, half, otw, log_size - 1)).all()


@pytest.mark.parametrize("log_poly,log_size,n_cols", [(20, 22, 8), (21, 22, 8), (21, 23, 4), (22, 23, 4), (20, 22, 5), (19, 21, 16), (20, 21, 9)])
def test_cfft_wide_plans_on_the_2_15_tile_fused_and_out_of_place(log_poly, log_size, n_cols):
    """From 4 columns (n = 22) / 2 columns (n = 23) on, the default plan puts the strided pass on the 2^15-word tile (13 + 9 /
    13 + 10): the fused extension (k_cfft_a<false, K, EXT, 15>) against extend + evaluate, that evaluation against the oracle on
    one column, and the out-of-place interpolate back to the zero-padded coefficients."""
    half = 1 << (31 - (log_size + 1))
    tw, itw = L.DeviceBuffer(4 << (log_size - 1)), L.DeviceBuffer(4 << (log_size - 1))
    L.call("tstwo_twiddles_build", half, log_size - 1, vp(tw), vp(itw))
    polys = [rand_column(31000 + 11 * log_size + c, 1 << log_poly) for c in range(n_cols)]
    src = [dev(p_) for p_ in polys]
    ref = [L.DeviceBuffer(4 << log_size) for _ in polys]
    for s_, r in zip(src, ref):
        L.call("tstwo_poly_extend", vp(s_), log_poly, vp(r), log_size)
    L.call("tstwo_cfft_evaluate", L.ptr_array([r.ptr for r in ref]), n_cols, log_size, half, vp(tw), log_size - 1)
    out = [L.DeviceBuffer(4 << log_size) for _ in polys]
    L.call("tstwo_cfft_evaluate_extended", L.ptr_array([s_.ptr for s_ in src]), log_poly, L.ptr_array([o.ptr for o in out]), n_cols,
           log_size, half, vp(tw), log_size - 1)
    for o, r in zip(out, ref):
        assert (o.download() == r.download()).all()
    otw, _ = orc.precompute_twiddles(half, log_size - 1, inverse=False)
    padded = np.zeros(1 << log_size, dtype=np.uint32)
    padded[:1 << log_poly] = polys[n_cols - 1]
    assert (out[n_cols - 1].download() == orc.cfft_evaluate(padded, log_size, half, otw, log_size - 1)).all()
    back = [L.DeviceBuffer(4 << log_size) for _ in polys]
    L.call("tstwo_cfft_interpolate_to", L.ptr_array([o.ptr for o in out]), L.ptr_array([b.ptr for b in back]), n_cols, log_size, half,
           vp(itw), log_size - 1)
    for b, p_ in zip(back, polys):
        got = b.download()
        assert (got[:1 << log_poly] == p_).all() and not got[1 << log_poly:].any()
    n_passes = C.c_uint32(0)
    L.call("tstwo_cfft_plan_passes", log_size, n_cols, C.byref(n_passes))
    assert n_passes.value == 2


def test_cfft_evaluate_extended_errors():
    tw = L.DeviceBuffer(4 << 12)
    L.call("tstwo_twiddles_build", 1 << (31 - 14), 12, vp(tw), vp(None))
    a, b = L.DeviceBuffer(4 << 13), L.DeviceBuffer(4 << 12)
    with pytest.raises(L.TstwoError, match="log size too small"):
        L.call("tstwo_cfft_evaluate_extended", L.ptr_array([a.ptr]), 13, L.ptr_array([b.ptr]), 1, 12, 1 << 18, vp(tw), 12)
    c = L.DeviceBuffer(4 << 15)
    with pytest.raises(L.TstwoError, match="Not enough twiddles!"):
        L.call("tstwo_cfft_evaluate_extended", L.ptr_array([a.ptr]), 13, L.ptr_array([c.ptr]), 1, 15, 1 << 15, vp(tw), 12)


@pytest.mark.parametrize("n", [3, 8, 12, 13, 14, 16, 19, 22, 23, 24])
def test_cfft_interpolate_to_matches_in_place(n):
    """tstwo_cfft_interpolate_to (source untouched, copy folded into the first pass) == copy + tstwo_cfft_interpolate."""
    n_cols = 3 if n <= 20 else 2
    half = 1 << (31 - (n + 1))
    tw, itw = L.DeviceBuffer(4 << (n - 1)), L.DeviceBuffer(4 << (n - 1))
    L.call("tstwo_twiddles_build", half, n - 1, vp(tw), vp(itw))
    evals = [rand_column(22000 + 5 * n + c, 1 << n) for c in range(n_cols)]
    src = [dev(e) for e in evals]
    ref = [dev(e) for e in evals]
    L.call("tstwo_cfft_interpolate", ptrs(ref), n_cols, n, half, vp(itw), n - 1)
    dst = [L.DeviceBuffer(4 << n) for _ in evals]
    L.call("tstwo_cfft_interpolate_to", ptrs(src), ptrs(dst), n_cols, n, half, vp(itw), n - 1)
    for d, r, s_, e in zip(dst, ref, src, evals):
        assert (d.download() == r.download()).all()
        assert (s_.download() == e).all()
    # and evaluating the coefficients gives the evaluations back
    L.call("tstwo_cfft_evaluate", ptrs(dst), n_cols, n, half, vp(tw), n - 1)
    assert (dst[0].download() == evals[0]).all()


@pytest.mark.parametrize("n", [23, 24])
def test_cfft_two_pass_plans_above_log22_vs_oracle(n):
    """n = 23 (14 + 9 layers on the 2^14-word bottom tile) and n = 24 (14 + 10: the strided pass on the 2^15-word tile, two virtual
    lanes per lane, scalar-base addressing) are TWO passes over memory: evaluate, interpolate, the out-of-place interpolate and the
    fused extension (n - 2 -> n, n - 1 -> n) against the oracle, every word."""
    half = half_odds(n - 1)
    tw, itw = build_twiddles(n - 1)
    otw, oitw = orc.precompute_twiddles(half, n - 1)
    cols = [rand_column(1000 * n + c, 1 << n) for c in range(2)]
    want = [orc.cfft_evaluate(c, n, half, otw, n - 1) for c in cols]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), 2, n, half, vp(tw), n - 1)
    for c in range(2):
        assert (host(d[c], 1 << n) == want[c]).all(), f"evaluate col {c}"
    # out-of-place interpolation of the evaluations gives the coefficients back and leaves the source untouched
    dst = [L.DeviceBuffer(4 << n) for _ in cols]
    L.call("tstwo_cfft_interpolate_to", ptrs(d), ptrs(dst), 2, n, half, vp(itw), n - 1)
    for c in range(2):
        assert (dst[c].download() == cols[c]).all() and (host(d[c], 1 << n) == want[c]).all()
    # in-place interpolation of arbitrary values against the oracle
    vals = rand_column(1000 * n + 100, 1 << n)
    dv = [dev(vals)]
    L.call("tstwo_cfft_interpolate", ptrs(dv), 1, n, half, vp(itw), n - 1)
    assert (host(dv[0], 1 << n) == orc.cfft_interpolate(vals, n, half, oitw, n - 1)).all()
    # fused extension: polynomials of log n - 2 / n - 1 evaluated on the log-n domain == zero-padded coefficients transformed
    for log_poly in (n - 2, n - 1):
        poly = rand_column(1000 * n + 200 + log_poly, 1 << log_poly)
        padded = np.zeros(1 << n, dtype=np.uint32)
        padded[:1 << log_poly] = poly
        out, src = [L.DeviceBuffer(4 << n)], [dev(poly)]
        L.call("tstwo_cfft_evaluate_extended", ptrs(src), log_poly, ptrs(out), 1, n, half, vp(tw), n - 1)
        assert (out[0].download() == orc.cfft_evaluate(padded, n, half, otw, n - 1)).all(), f"extended {log_poly} -> {n}"


@pytest.mark.parametrize("n", [25, 26, 27, 28, 29, 30])
def test_cfft_maximum_sizes(n):
    """The largest transforms the tiled path plans, up to the reference's MAX_CIRCLE_DOMAIN_LOG_SIZE = 30
    (poly/circle/domain.ts:4; 3 passes from log 23, a 4 GiB column at log 30): the evaluation agrees with eval_at_point at
    sampled domain points, interpolate inverts it, and (log 29, 30) the transform is linear."""
    tw, itw = build_twiddles(n - 1)
    a = rand_column(n, 1 << n)
    d = [dev(a)]
    coeffs = dev(a)
    L.call("tstwo_cfft_evaluate", ptrs(d), 1, n, half_odds(n - 1), vp(tw), n - 1)
    rng = np.random.default_rng(n)
    for i in rng.integers(0, 1 << n, size=3):
        p = OL.orc_circle_domain_at(half_odds(n - 1), n - 1, int(i))
        out = (C.c_uint32 * 4)()
        L.call("tstwo_eval_at_point", vp(coeffs), n, L.u32x((p.x, 0, 0, 0)), L.u32x((p.y, 0, 0, 0)), out)
        got = d[0].download(np.uint32, 1, 4 * OL.orc_bit_reverse_index(int(i), n))
        assert tuple(out) == (int(got[0]), 0, 0, 0)
    if n == 26:                                                        # one point through the CPU oracle as well
        i = int(rng.integers(0, 1 << n))
        p = OL.orc_circle_domain_at(half_odds(n - 1), n - 1, i)
        v = orc.eval_at_point(a, n, (p.x, 0, 0, 0), (p.y, 0, 0, 0))
        assert v == (int(d[0].download(np.uint32, 1, 4 * OL.orc_bit_reverse_index(i, n))[0]), 0, 0, 0)
    if n >= 29:      # linearity: evaluate(b) + evaluate(a) == evaluate(a + b), all three on the device
        b = dev(rand_column(n + 100, 1 << n))
        s_ = dev_empty(1 << n)
        L.call("tstwo_m31_add", vp(coeffs), vp(b), vp(s_), 1 << n)
        L.call("tstwo_cfft_evaluate", ptrs([b, s_]), 2, n, half_odds(n - 1), vp(tw), n - 1)
        L.call("tstwo_m31_add", vp(d[0]), vp(b), vp(b), 1 << n)
        assert (host(b, 1 << n) == host(s_, 1 << n)).all()
        b.free()
        s_.free()
    L.call("tstwo_cfft_interpolate", ptrs(d), 1, n, half_odds(n - 1), vp(itw), n - 1)
    back = host(d[0], 1 << n)
    assert (back == a).all()


def test_cfft_rejects_sizes_above_30():
    d = dev(rand_column(1, 16))
    with pytest.raises(L.TstwoError, match="exceeds MAX_CIRCLE_DOMAIN_LOG_SIZE"):
        L.call("tstwo_cfft_evaluate", ptrs([d]), 1, 31, 1, vp(d), 30)


def test_null_pointers_are_errors_not_faults():
    """Every entry point that takes device pointers answers a null with TSTWO_ERR_BAD_ARG before any kernel is launched."""
    d = dev(rand_column(1, 64))
    four = L.p4([d.ptr] * 4)
    bad4 = L.p4([d.ptr, d.ptr, 0, d.ptr])
    a = L.u32x([1, 0, 0, 0])
    null_err = pytest.raises(L.TstwoError, match="null")
    with null_err:
        L.call("tstwo_cfft_evaluate", L.ptr_array([d.ptr, 0]), 2, 5, 1 << 25, vp(d), 4)
    with null_err:
        L.call("tstwo_cfft_interpolate_to", L.ptr_array([0]), L.ptr_array([d.ptr]), 1, 5, 1 << 25, vp(d), 4)
    with null_err:
        L.call("tstwo_bit_reverse", L.ptr_array([0]), 1, 64)
    with null_err:
        L.call("tstwo_m31_batch_inverse", vp(None), vp(d), 64)
    with null_err:
        L.call("tstwo_qm31_mul", four, bad4, four, 64)
    with null_err:
        L.call("tstwo_secure_accumulate", bad4, four, 64)
    with null_err:
        L.call("tstwo_fri_fold_line", bad4, 6, vp(d), 6, a, four)
    with null_err:
        L.call("tstwo_fri_fold_line", four, 6, vp(None), 6, a, four)
    with null_err:
        L.call("tstwo_fri_fold_circle_into_line", four, 32, bad4, 6, vp(d), 6, a)
    with null_err:
        L.call("tstwo_merkle_commit", L.ptr_array([d.ptr, 0]), L.u32x([6, 6]), 2, vp(d), None)
    with null_err:
        L.call("tstwo_merkle_commit", L.ptr_array([d.ptr]), L.u32x([6]), 1, vp(None), None)
    with null_err:
        L.call("tstwo_eval_at_point", vp(None), 6, a, a, (C.c_uint32 * 4)())
    with null_err:
        L.call("tstwo_poly_extend", vp(d), 6, vp(None), 8)
    L.sync()


@pytest.mark.parametrize("n,n_cols", [(1, 200), (2, 130), (6, 300), (12, 150), (13, 70), (15, 97)])
def test_many_columns_one_launch(n, n_cols):
    """More than 64 columns go through a device-resident pointer table (one launch per pass instead of one per 64 columns):
    evaluate, interpolate, the out-of-place / extended variants and bit-reverse agree with the oracle column by column."""
    tw, itw = build_twiddles(n - 1)
    half = half_odds(n - 1)
    cols = [rand_column(40000 + 3 * n + c, 1 << n) for c in range(n_cols)]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), n_cols, n, half, vp(tw), max(n - 1, 0))
    otw, oitw = orc.precompute_twiddles(half, n - 1)
    check = sorted(set([0, 1, 63, 64, 65, n_cols - 1]))
    ev = {c: host(d[c], 1 << n) for c in check}
    for c in check:
        assert (ev[c] == orc.cfft_evaluate(cols[c], n, half, otw, max(n - 1, 0))).all(), c
    # out-of-place interpolate of all columns gives the inputs back; sources untouched
    out = [L.DeviceBuffer(4 << n) for _ in cols]
    L.call("tstwo_cfft_interpolate_to", ptrs(d), ptrs(out), n_cols, n, half, vp(itw), max(n - 1, 0))
    for c in range(n_cols):
        assert (host(out[c], 1 << n) == cols[c]).all(), c
    for c in check:
        assert (host(d[c], 1 << n) == ev[c]).all()
    # in-place interpolate as well
    L.call("tstwo_cfft_interpolate", ptrs(d), n_cols, n, half, vp(itw), max(n - 1, 0))
    for c in check:
        assert (host(d[c], 1 << n) == cols[c]).all()
    # bit-reverse of all columns
    L.call("tstwo_bit_reverse", ptrs(d), n_cols, 1 << n)
    for c in check:
        assert (host(d[c], 1 << n) == orc.bit_reverse(cols[c])).all()


def test_many_columns_evaluate_extended():
    n_poly, n, n_cols = 12, 14, 80
    tw, _ = build_twiddles(n - 1)
    polys = [rand_column(41000 + c, 1 << n_poly) for c in range(n_cols)]
    src = [dev(p) for p in polys]
    out = [L.DeviceBuffer(4 << n) for _ in polys]
    L.call("tstwo_cfft_evaluate_extended", ptrs(src), n_poly, ptrs(out), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    otw, _ = orc.precompute_twiddles(half_odds(n - 1), n - 1, inverse=False)
    for c in (0, 63, 64, 79):
        ext = np.concatenate([polys[c], np.zeros((1 << n) - (1 << n_poly), dtype=np.uint32)])
        assert (host(out[c], 1 << n) == orc.cfft_evaluate(ext, n, half_odds(n - 1), otw, n - 1)).all()


@pytest.mark.parametrize("log", [1, 3, 9, 14])
def test_quotients_from_samples_vs_oracle(log, golden):
    """tstwo_quotients_accumulate_samples (constants computed inside the library) == the oracle's per-row reference loop."""
    px, py = golden["eval_at_point"][0]["point"]
    n_cols = 6
    cols = [rand_column(45000 + log * 8 + c, 1 << log) for c in range(n_cols)]
    vals = [tuple(int(x) for x in rand_column(45500 + j, 4)) for j in range(7)]
    py2 = OL.orc_qm31_mul(orc.q(py), orc.q(py)).tup()
    batches = [(px, py, [(0, vals[0]), (3, vals[1]), (5, vals[2])]), (py, py2, [(1, vals[3])]), (px, py2, [(2, vals[4]), (0, vals[5]), (4, vals[6])])]
    d = [dev(c) for c in cols]
    out = [L.DeviceBuffer(max(4 << log, 16)) for _ in range(4)]
    off, cidx, points, values = [0], [], [], []
    for bx, by, cv in batches:
        points += [*bx, *by]
        for ci, v in cv:
            cidx.append(ci)
            values += list(v)
        off.append(len(cidx))
    L.call("tstwo_quotients_accumulate_samples", half_odds(log - 1), log, ptrs(d), n_cols, len(batches), L.u32x(off), L.u32x(cidx),
           L.u32x(points), L.u32x(values), L.u32x((1, 2, 3, 4)), p4(out))
    exp = orc.accumulate_quotients(half_odds(log - 1), log, cols, (1, 2, 3, 4), batches)
    for k in range(4):
        assert (host(out[k], 1 << log) == exp[k]).all()
    # a sample point whose y is its own conjugate (base-field y) cannot define a line
    with pytest.raises(L.TstwoError, match="Cannot evaluate a line with a single point"):
        L.call("tstwo_quotients_accumulate_samples", half_odds(log - 1), log, ptrs(d), n_cols, 1, L.u32x([0, 1]), L.u32x([0]),
               L.u32x([1, 2, 3, 4, 5, 6, 0, 0]), L.u32x([1, 1, 1, 1]), L.u32x((1, 2, 3, 4)), p4(out))


def test_quotient_batch_offsets_are_validated():
    """batch_off sizes host buffers inside the library: it must start at 0 and never decrease (TSTWO_ERR_BAD_ARG otherwise,
    nothing is launched)."""
    n = 6
    cols = [dev(rand_column(70 + c, 1 << n)) for c in range(2)]
    out = [dev_empty(1 << n) for _ in range(4)]
    pts = L.u32x([1, 0, 478637715, 513582971, 992285211, 649143431, 740191619, 1186584352] * 2)
    vals = L.u32x([7, 8, 9, 10] * 4)
    for bad, what in (([1, 2, 3], r"batch_off\[0\] must be 0"), ([0, 3, 2], "non-decreasing")):
        with pytest.raises(L.TstwoError, match=what):
            L.call("tstwo_quotients_accumulate_samples", half_odds(n - 1), n, ptrs(cols), 2, 2, L.u32x(bad), L.u32x([0, 1, 0, 1]), pts, vals,
                   L.u32x([1, 2, 3, 4]), p4(out))
    L.call("tstwo_quotients_accumulate_samples", half_odds(n - 1), n, ptrs(cols), 2, 2, L.u32x([0, 1, 2]), L.u32x([0, 1]), pts, vals,
           L.u32x([1, 2, 3, 4]), p4(out))


def test_allocator_modes_and_threads():
    """tstwo_set_alloc_mode rejects unknown modes; tstwo_malloc / tstwo_free are safe from several threads at once (the pool is
    behind a mutex: a finaliser thread may release blocks while another thread allocates)."""
    import threading
    with pytest.raises(L.TstwoError, match="unknown mode"):
        L.call("tstwo_set_alloc_mode", 7)
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            held = []
            for _ in range(400):
                if held and rng.random() < 0.5:
                    held.pop(int(rng.integers(0, len(held)))).free()
                else:
                    held.append(L.DeviceBuffer(int(rng.choice([16, 4096, 5000, 65536, 1 << 20]))))
            ptrs_ = [b.ptr for b in held]
            assert len(set(ptrs_)) == len(ptrs_)
            for b in held:
                b.free()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    L.call("tstwo_set_alloc_mode", L.ALLOC_POOL)


def test_async_alloc_mode_is_refused_by_default():
    """HIP's stream-ordered pool returns wrong data on ROCm 7.2 / gfx950 (profiles/r02_hipmallocasync_fault.txt): the mode must be
    refused unless TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC=1 opts in."""
    import os
    if os.environ.get("TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC"):
        pytest.skip("the opt-in is set")
    with pytest.raises(L.TstwoError, match="TSTWO_ALLOC_ASYNC refused"):
        L.call("tstwo_set_alloc_mode", L.ALLOC_ASYNC)
    with pytest.raises(L.TstwoError, match="TSTWO_ALLOC_ASYNC refused"):
        L.call("tstwo_set_alloc_mode", L.ALLOC_ASYNC | L.ALLOC_POISON)
    b = L.DeviceBuffer(4096)          # the allocator still works in its previous mode
    b.free()


def test_host_array_upload_during_graph_capture_is_refused():
    """A transform of more than 64 columns uploads its pointer table from a host array; recorded into a graph, the copy would read
    a recycled staging slot at replay.  The call must fail and leave the table cache untouched: the same call works eagerly
    afterwards and gives the oracle's evaluations."""
    evs_host = [rand_column(990 + i, 1 << 13) for i in range(70)]
    evs = [dev(e) for e in evs_host]
    tw = dev_empty(1 << 12)
    L.call("tstwo_twiddles_build", half_odds(12), 12, vp(tw), vp(None))
    L.sync()
    L.call("tstwo_graph_begin_capture")
    try:
        with pytest.raises(L.TstwoError, match="host-array upload during graph capture"):
            L.call("tstwo_cfft_evaluate", ptrs(evs), len(evs), 13, half_odds(12), vp(tw), 12)
    finally:
        h = C.c_void_p()
        try:
            L.call("tstwo_graph_end_capture", C.byref(h))
        except L.TstwoError:
            pass
        if h.value:
            L.call("tstwo_graph_destroy", h)
    otw = orc.precompute_twiddles(half_odds(12), 12)[0]
    L.call("tstwo_cfft_evaluate", ptrs(evs), len(evs), 13, half_odds(12), vp(tw), 12)
    for i in (0, 1, 69):
        assert (host(evs[i], 1 << 13) == orc.cfft_evaluate(evs_host[i], 13, half_odds(12), otw, 12)).all()


def test_device_pointer_table_is_refused_during_capture_on_a_cache_hit_too():
    """The pointer table of a > 64-column call lives in a slot of device memory that the NEXT such call rewrites.  Round 3 refused
    only the upload (a cache miss); on a hit — the same call made eagerly just before — the launch was recorded against the slot
    with no refusal, and a later call would have changed what the graph reads at replay.  Eager call, then the same call under
    capture: error; and the eager call still works afterwards."""
    evs_host = [rand_column(1990 + i, 1 << 13) for i in range(70)]
    evs = [dev(e) for e in evs_host]
    tw, itw = dev_empty(1 << 12), dev_empty(1 << 12)
    L.call("tstwo_twiddles_build", half_odds(12), 12, vp(tw), vp(itw))
    table = ptrs(evs)
    L.call("tstwo_cfft_evaluate", table, len(evs), 13, half_odds(12), vp(tw), 12)        # fills the slot: the next call is a hit
    L.sync()
    L.call("tstwo_graph_begin_capture")
    try:
        with pytest.raises(L.TstwoError, match="during graph capture"):
            L.call("tstwo_cfft_evaluate", table, len(evs), 13, half_odds(12), vp(tw), 12)
    finally:
        h = C.c_void_p()
        try:
            L.call("tstwo_graph_end_capture", C.byref(h))
        except L.TstwoError:
            pass
        if h.value:
            L.call("tstwo_graph_destroy", h)
    L.call("tstwo_cfft_interpolate", table, len(evs), 13, half_odds(12), vp(itw), 12)
    for i in (0, 33, 69):
        assert (host(evs[i], 1 << 13) == evs_host[i]).all()


_KNOB_SCRIPT = r"""
import sys
sys.path[:0] = [{root!r}, {tests!r}]
import numpy as np
from test_gpu_capi import rand_column, dev, dev_empty, host, ptrs, half_odds, vp, L
n = 15
cols = [dev(rand_column(4200 + c, 1 << n)) for c in range(3)]
tw = dev_empty(1 << (n - 1))
L.call("tstwo_twiddles_build", half_odds(n - 1), n - 1, vp(tw), vp(None))
L.call("tstwo_cfft_evaluate", ptrs(cols), 3, n, half_odds(n - 1), vp(tw), n - 1)
import hashlib
print(L.version(), hashlib.blake2s(b"".join(host(c, 1 << n).tobytes() for c in cols)).hexdigest())
"""


def test_shipped_library_ignores_the_experiment_knobs():
    """TSTWO_CFFT_GENERIC=4 SKIPS the bottom pass — in the experiments build.  The shipped library must not let its caller's
    environment change a result: with the variable set it still gives the oracle's evaluations, while the experiments build,
    given the same variable, demonstrably does not (so the test would notice if the switch stopped meaning anything)."""
    import hashlib
    import os
    import subprocess
    import sys
    n = 15
    otw = orc.precompute_twiddles(half_odds(n - 1), n - 1)[0]
    want = hashlib.blake2s(b"".join(orc.cfft_evaluate(rand_column(4200 + c, 1 << n), n, half_odds(n - 1), otw, n - 1).tobytes()
                                    for c in range(3))).hexdigest()
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    script = _KNOB_SCRIPT.format(root=os.path.dirname(tests_dir), tests=tests_dir)
    knobs = dict(TSTWO_CFFT_GENERIC="4", TSTWO_CFFT_KB="12", TSTWO_MERKLE_GENERIC="1", TSTWO_CFFT_ROUNDS="3")
    env = {k: v for k, v in os.environ.items() if k != "TSTWO_HIP_LIB"}
    out = subprocess.run([sys.executable, "-c", script], env=dict(env, **knobs), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    ver, digest = out.stdout.strip().splitlines()[-1].rsplit(" ", 1)
    assert "experiments" not in ver and digest == want
    out = subprocess.run([sys.executable, "-c", script], env=dict(env, TSTWO_HIP_LIB=L.LIB_EXP_PATH, **knobs), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    ver, digest = out.stdout.strip().splitlines()[-1].rsplit(" ", 1)
    assert "experiments" in ver and digest != want
    out = subprocess.run([sys.executable, "-c", script], env=dict(env, TSTWO_HIP_LIB=L.LIB_EXP_PATH, TSTWO_CFFT_KB="12"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].rsplit(" ", 1)[1] == want          # a plan-changing knob alone: same results
    out = subprocess.run([sys.executable, "-c", script], env=dict(env, TSTWO_HIP_LIB=L.LIB_EXP_PATH, TSTWO_CFFT_B8="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].rsplit(" ", 1)[1] == want          # the 8-words-per-lane bottom pass (measured, not shipped): same results


@pytest.mark.parametrize("shape", [(4, 32, 17), (8, 32, 18), (3, 16, 19), (2, 64, 17), (5, 48, 17), (2, 32, 12), (1, 32, 17), (3, 20, 17)], ids=str)
def test_merkle_commit_many_equals_commits_one_by_one(shape):
    """tstwo_merkle_commit_many: every layer of every tree byte for byte what tstwo_merkle_commit writes for that tree alone —
    shapes the shared launches serve (16 / 32 / 48 / 64 columns of one log size >= 17, up to 8 trees) and shapes that fall back
    to the tree-by-tree loop (small trees, other column counts, one tree); one root per tree against the oracle."""
    n_trees, n_cols, n = shape
    rng = np.random.default_rng(n_trees * 1000 + n_cols * 10 + n)
    base = [dev(rng.integers(0, P, size=1 << n, dtype=np.uint32)) for _ in range(min(n_cols + n_trees, 40))]
    trees = [[base[(t * 7 + k) % len(base)] for k in range(n_cols)] for t in range(n_trees)]       # overlapping column sets: inputs are read only
    nbytes = 32 * ((2 << n) - 1)
    many = [dev_empty(nbytes // 4) for _ in range(n_trees)]
    reqs = (L.CommitRequest * n_trees)()
    keep = []
    for t in range(n_trees):
        cp, lg = ptrs(trees[t]), L.u32x([n] * n_cols)
        keep += [cp, lg]
        reqs[t] = L.CommitRequest(cp, lg, n_cols, many[t].ptr)
    roots = (C.c_uint8 * (32 * n_trees))()
    L.call("tstwo_merkle_commit_many", reqs, n_trees, roots)
    single = dev_empty(nbytes // 4)
    for t in range(n_trees):
        root = (C.c_uint8 * 32)()
        L.call("tstwo_merkle_commit", ptrs(trees[t]), L.u32x([n] * n_cols), n_cols, vp(single), root)
        assert bytes(roots[32 * t:32 * t + 32]) == bytes(root), f"tree {t}"
        assert (host(many[t], nbytes // 4) == host(single, nbytes // 4)).all(), f"tree {t}: layers differ"
    cols0 = [host(c, 1 << n) for c in trees[0]]
    assert bytes(roots[:32]) == orc.mt_merkle_root(cols0, n, 8)


def test_merkle_commit_many_mixed_shapes_and_errors():
    n = 10
    a = [dev(rand_column(5000 + i, 1 << n)) for i in range(3)]
    b = [dev(rand_column(5100 + i, 1 << (n - 2))) for i in range(2)]
    la, lb = dev_empty(8 * ((2 << n) - 1)), dev_empty(8 * ((2 << n) - 1))
    reqs = (L.CommitRequest * 2)()
    pa, ga, pb, gb = ptrs(a), L.u32x([n] * 3), ptrs(a + b), L.u32x([n] * 3 + [n - 2] * 2)
    reqs[0] = L.CommitRequest(pa, ga, 3, la.ptr)
    reqs[1] = L.CommitRequest(pb, gb, 5, lb.ptr)
    roots = (C.c_uint8 * 64)()
    L.call("tstwo_merkle_commit_many", reqs, 2, roots)
    assert bytes(roots[:32]) == orc.merkle_commit([host(c, 1 << n) for c in a], [n] * 3)[1]
    assert bytes(roots[32:]) == orc.merkle_commit([host(c, 1 << n) for c in a] + [host(c, 1 << (n - 2)) for c in b], [n] * 3 + [n - 2] * 2)[1]
    L.call("tstwo_merkle_commit_many", reqs, 0, None)                 # nothing to do
    with pytest.raises(L.TstwoError, match="null request table"):
        L.call("tstwo_merkle_commit_many", None, 2, roots)


@pytest.mark.parametrize("sizes", [[10], [10, 16, 16, 16, 16, 16], [1, 0, 3, 70000, 5], list(range(1, 40)), [40000, 40000, 7]], ids=str)
def test_download_many_equals_downloads_one_by_one(sizes):
    """tstwo_download_many: the pieces back to back, byte-identical to one tstwo_download per piece — through the mapped result
    page (up to 256 KiB, more than 16 pieces = several packing launches) and through the piece-by-piece path above it."""
    rng = np.random.default_rng(sum(sizes))
    hosts = [rng.integers(0, 2**32, size=max(w, 1), dtype=np.uint32) for w in sizes]
    bufs = [dev(h) for h in hosts]
    got = L.download_many([(b.ptr, w) for b, w in zip(bufs, sizes)])
    assert len(got) == len(sizes)
    for g, h, w in zip(got, hosts, sizes):
        assert np.array_equal(g, h[:w])
    # a slice in the middle of a buffer (word-aligned offset)
    if sizes[0] >= 10:
        (mid,) = L.download_many([(bufs[0].ptr + 12, 5)])
        assert np.array_equal(mid, hosts[0][3:8])


def test_download_many_errors():
    b = dev(np.arange(8, dtype=np.uint32))
    out = np.zeros(8, dtype=np.uint32)
    srcs = (C.c_void_p * 1)(b.ptr)
    with pytest.raises(L.TstwoError, match="whole, 4-byte aligned words"):
        L.call("tstwo_download_many", srcs, (C.c_size_t * 1)(6), 1, out.ctypes.data_as(C.c_void_p))
    with pytest.raises(L.TstwoError, match="whole, 4-byte aligned words"):
        L.call("tstwo_download_many", (C.c_void_p * 1)(b.ptr + 2), (C.c_size_t * 1)(8), 1, out.ctypes.data_as(C.c_void_p))
    with pytest.raises(L.TstwoError, match="null argument"):
        L.call("tstwo_download_many", srcs, (C.c_size_t * 1)(8), 1, C.c_void_p(0))
    L.call("tstwo_download_many", srcs, (C.c_size_t * 1)(0), 1, out.ctypes.data_as(C.c_void_p))       # nothing to fetch: fine
    assert len(L.download_many([])) == 0


@pytest.mark.parametrize("log,n_cols", [(3, 2), (9, 5), (10, 4), (12, 9), (14, 33)])
def test_quotients_two_batches_over_one_column_list(log, n_cols, golden):
    """Every column opened at two points (two sample batches over the same column list): the kernel that loads the column words
    once for both batches (k_quotients8_multi<2>) against the oracle's per-row reference loop — and the same input with the second
    batch's columns in another order (same union list: still one sweep)."""
    px, py = golden["eval_at_point"][0]["point"]
    py2 = OL.orc_qm31_mul(orc.q(py), orc.q(py)).tup()
    cols = [rand_column(9100 + 7 * log + c, 1 << log) for c in range(n_cols)]
    vals = [tuple(int(x) for x in rand_column(9500 + 3 * log + j, 4)) for j in range(2 * n_cols)]
    d = [dev(c) for c in cols]
    for second in (list(range(n_cols)), list(range(n_cols))[::-1]):
        batches = [(px, py, [(c, vals[c]) for c in range(n_cols)]), (py, py2, [(c, vals[n_cols + i]) for i, c in enumerate(second)])]
        off, cidx, points, values = [0], [], [], []
        for bx, by, cv in batches:
            points += [*bx, *by]
            for ci, v in cv:
                cidx.append(ci)
                values += list(v)
            off.append(len(cidx))
        out = [L.DeviceBuffer(max(4 << log, 16)) for _ in range(4)]
        L.call("tstwo_quotients_accumulate_samples", half_odds(log - 1), log, ptrs(d), n_cols, 2, L.u32x(off), L.u32x(cidx),
               L.u32x(points), L.u32x(values), L.u32x((5, 6, 7, 8)), p4(out))
        exp = orc.accumulate_quotients(half_odds(log - 1), log, cols, (5, 6, 7, 8), batches)
        for k in range(4):
            assert (host(out[k], 1 << log) == exp[k]).all(), (log, n_cols, second[:3], k)


@pytest.mark.parametrize("log,n_cols,k", [(10, 5, 3), (12, 32, 3), (9, 4, 4), (13, 9, 4), (11, 6, 5), (12, 3, 7), (5, 4, 3)])
def test_quotients_k_batches_over_one_column_list(log, n_cols, k, golden):
    """Every column opened at k points (k sample batches over the SAME column list): sweeps of 3 or 2 batches that load the column
    words once per sweep and continue from the rows the previous sweep wrote (k_quotients8_multi<NB, ACCUM>), against the oracle's
    per-row reference loop; and the same input with ONE batch's columns in another order (served from the union list, by column
    identity), which must give the oracle's rows as well."""
    px, py = golden["eval_at_point"][0]["point"]
    pts = [(px, py)]
    for _ in range(k - 1):                                   # further points on the QM31 circle: repeated doubling
        x, y = pts[-1]
        x2 = OL.orc_qm31_mul(orc.q(x), orc.q(x)).tup()
        xy = OL.orc_qm31_mul(orc.q(x), orc.q(y)).tup()
        P_ = 2147483647
        pts.append((tuple((2 * a - (1 if i == 0 else 0)) % P_ for i, a in enumerate(x2)), tuple((2 * a) % P_ for a in xy)))
    cols = [rand_column(9700 + 7 * log + c, 1 << log) for c in range(n_cols)]
    vals = [tuple(int(x) for x in rand_column(9900 + 3 * log + j, 4)) for j in range(k * n_cols)]
    d = [dev(c) for c in cols]
    for permuted in (None, k - 1, 0):
        batches = []
        for b in range(k):
            order = list(range(n_cols))[::-1] if permuted == b else list(range(n_cols))
            batches.append((pts[b][0], pts[b][1], [(c, vals[b * n_cols + i]) for i, c in enumerate(order)]))
        off, cidx, points, values = [0], [], [], []
        for bx, by, cv in batches:
            points += [*bx, *by]
            for ci, v in cv:
                cidx.append(ci)
                values += list(v)
            off.append(len(cidx))
        out = [L.DeviceBuffer(max(4 << log, 16)) for _ in range(4)]
        for o in out:
            L.call("tstwo_zero", vp(o), max(4 << log, 16))
        L.call("tstwo_quotients_accumulate_samples", half_odds(log - 1), log, ptrs(d), n_cols, k, L.u32x(off), L.u32x(cidx),
               L.u32x(points), L.u32x(values), L.u32x((5, 6, 7, 8)), p4(out))
        exp = orc.accumulate_quotients(half_odds(log - 1), log, cols, (5, 6, 7, 8), batches)
        for c in range(4):
            assert (host(out[c], 1 << log) == exp[c]).all(), (log, n_cols, k, permuted, c)


@pytest.mark.parametrize("log,n_cols", [(10, 8), (12, 33), (6, 5)])
def test_quotients_batches_over_overlapping_column_lists(log, n_cols, golden):
    """Batches whose column lists overlap without being equal — every column at the first point, every second one also at a
    second point, a few (one of them listed twice) at a third — are served from the UNION list with zero coefficients where a
    batch does not sample a column; rows must equal the oracle's per-row reference loop.  Disjoint lists (below the 1.4 entries per
    union column the library asks for) take the per-batch kernel and must give the oracle's rows too."""
    px, py = golden["eval_at_point"][0]["point"]
    py2 = OL.orc_qm31_mul(orc.q(py), orc.q(py)).tup()
    pts = [(px, py), (py, py2), (py2, px)]
    cols = [rand_column(9300 + 7 * log + c, 1 << log) for c in range(n_cols)]
    d = [dev(c) for c in cols]
    val = lambda j: tuple(int(x) for x in rand_column(9400 + 3 * log + j, 4))
    lists = {
        "overlap": [list(range(n_cols)), list(range(0, n_cols, 2)), [n_cols - 1, 1, 1, 0]],
        "disjoint": [list(range(0, n_cols // 2)), list(range(n_cols // 2, n_cols))],
    }
    for name, ls in lists.items():
        batches, j = [], 0
        for b, cl in enumerate(ls):
            cv = []
            for c in cl:
                cv.append((c, val(j)))
                j += 1
            batches.append((pts[b][0], pts[b][1], cv))
        off, cidx, points, values = [0], [], [], []
        for bx, by, cv in batches:
            points += [*bx, *by]
            for ci, v in cv:
                cidx.append(ci)
                values += list(v)
            off.append(len(cidx))
        out = [L.DeviceBuffer(max(4 << log, 16)) for _ in range(4)]
        L.call("tstwo_quotients_accumulate_samples", half_odds(log - 1), log, ptrs(d), n_cols, len(batches), L.u32x(off), L.u32x(cidx),
               L.u32x(points), L.u32x(values), L.u32x((5, 6, 7, 8)), p4(out))
        exp = orc.accumulate_quotients(half_odds(log - 1), log, cols, (5, 6, 7, 8), batches)
        for k in range(4):
            assert (host(out[k], 1 << log) == exp[k]).all(), (name, log, n_cols, k)


def test_upload_async_from_registered_and_library_pinned_memory():
    """tstwo_host_register / tstwo_host_alloc + tstwo_upload_async / _fence / _wait (the createBaseFieldColumn boundary,
    backend/index.ts:20-31): copies issued on the copy stream land before main-stream work enqueued behind the fence, never
    overtake main-stream work enqueued before them, and a pipelined upload-under-transform gives the oracle's evaluations."""
    n = 16
    N = 1 << n
    tw = dev_empty(1 << (n - 1))
    L.call("tstwo_twiddles_build", half_odds(n - 1), n - 1, vp(tw), vp(None))
    otw = orc.precompute_twiddles(half_odds(n - 1), n - 1, inverse=False)[0]
    cols = [rand_column(5600 + c, N) for c in range(6)]
    # (a) registered caller memory, (b) library-pinned memory, (c) pageable memory: all three must arrive
    reg = cols[0].copy()
    L.host_register(reg)
    pin = L.PinnedArray(N)
    pin.array[:] = cols[1]
    d = [dev_empty(N) for _ in range(3)]
    d[0].upload_async(reg)
    d[1].upload_async(pin.array)
    d[2].upload_async(cols[2])
    L.upload_fence()
    L.call("tstwo_cfft_evaluate", ptrs(d), 3, n, half_odds(n - 1), vp(tw), n - 1)      # behind the fence: sees the uploaded words
    L.upload_wait()
    for k in range(3):
        assert (host(d[k], N) == orc.cfft_evaluate(cols[k], n, half_odds(n - 1), otw, n - 1)).all()
    # a copy does not overtake work enqueued before it: transform in place, THEN overwrite with fresh words
    d2 = dev(cols[3])
    L.call("tstwo_cfft_evaluate", ptrs([d2]), 1, n, half_odds(n - 1), vp(tw), n - 1)
    pin.array[:] = cols[4]
    d2.upload_async(pin.array)
    L.sync()                                                                           # tstwo_sync also waits for unfenced copies
    assert (host(d2, N) == cols[4]).all()
    # pipeline: column k+1 travels while column k is transformed
    pins = [L.PinnedArray(N) for _ in range(4)]
    for k in range(4):
        pins[k].array[:] = cols[k + 2]
    dd = [dev_empty(N) for _ in range(4)]
    dd[0].upload_async(pins[0].array)
    for k in range(4):
        L.upload_fence()
        if k + 1 < 4:
            dd[k + 1].upload_async(pins[k + 1].array)
        L.call("tstwo_cfft_evaluate", ptrs([dd[k]]), 1, n, half_odds(n - 1), vp(tw), n - 1)
    L.sync()
    for k in range(4):
        assert (host(dd[k], N) == orc.cfft_evaluate(cols[k + 2], n, half_odds(n - 1), otw, n - 1)).all()
    L.host_unregister(reg)
    for p_ in pins + [pin]:
        p_.free()
    with pytest.raises(L.TstwoError):
        L.call("tstwo_host_register", vp(None), 16)
