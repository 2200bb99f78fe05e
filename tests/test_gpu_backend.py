"""-m gpu tests of the host-side mirror (HipBackend & friends), written to read like the reference's own tests:
packages/core/test/backend/backend.test.ts, test/backend/cpu/circle.test.ts, test/poly/circleEvaluation.test.ts,
test/fri.test.ts, test/backend/cpu/fri.test.ts, test/vcs/prover.test.ts, test/backend/cpu/quotients.test.ts."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import P, column, rand_column
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import tstwo_amd as T  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

OL = orc.lib()


@pytest.fixture(scope="module")
def backend():
    L.init(0)
    return T.HipBackend()


# ---------------------------------------------------------------- backend.test.ts
def test_backend_name_and_columns(backend):
    assert backend.name == "HipBackend"
    col = backend.createBaseFieldColumn([T.M31(1), T.M31(2), T.M31(3), T.M31(4)])
    assert col.len() == 4 and not col.isEmpty()
    assert col.at(2) == T.M31(3)
    col.set(1, T.M31(42))
    assert [m.value for m in col.toCpu()] == [1, 42, 3, 4]
    with pytest.raises(IndexError, match="out of bounds"):
        col.at(4)
    with pytest.raises(IndexError, match="out of bounds"):
        col.set(-1, T.M31(0))
    z = T.HipColumn.zeros(5)
    assert [m.value for m in z.toCpu()] == [0] * 5
    assert T.HipColumn.uninitialized(7).len() == 7
    assert T.HipColumn([]).isEmpty()
    s = backend.createSecureFieldColumn([T.QM31.from_u32_unchecked(1, 2, 3, 4), T.QM31.from_u32_unchecked(5, 6, 7, 8)])
    assert s.at(1) == T.QM31.from_u32_unchecked(5, 6, 7, 8)
    s.set(0, T.QM31.from_u32_unchecked(9, 9, 9, 9))
    assert [q.tup() for q in s.to_vec()] == [(9, 9, 9, 9), (5, 6, 7, 8)]
    # value semantics: constructing from an array copies it (cpu/index.ts:89)
    a = np.array([1, 2, 3, 4], dtype=np.uint32)
    c = T.HipColumn(a)
    a[0] = 99
    assert c.at(0) == T.M31(1)


def test_bit_reverse_column(backend):
    col = backend.createBaseFieldColumn(list(range(8)))
    backend.bitReverseColumn(col)
    assert [m.value for m in col.toCpu()] == [0, 4, 2, 6, 1, 5, 3, 7]
    for bad in (3, 6):
        with pytest.raises(L.TstwoError, match="length is not power of two"):
            backend.bitReverseColumn(backend.createBaseFieldColumn(list(range(bad))))
    with pytest.raises(L.TstwoError, match="length is not power of two"):
        backend.bitReverseColumn(T.HipColumn([]))
    sec = T.SecureColumnByCoords.from_numpy([np.arange(4, dtype=np.uint32) + 10 * k for k in range(4)])
    backend.bitReverseColumn(sec)
    assert sec.to_numpy()[1].tolist() == [10, 12, 11, 13]


def test_field_column_ops_and_batch_inverse(backend):
    n = 1 << 20                                             # BASELINE config 1 size
    a, b = rand_column(1, n), rand_column(2, n, nonzero=True)
    da, db = T.HipColumn(a), T.HipColumn(b)
    assert (backend.add(da, db).to_numpy() == orc.col_op("add", a, b)).all()
    assert (backend.mul(da, db).to_numpy() == orc.col_op("mul", a, b)).all()
    assert (backend.sub(da, db).to_numpy() == orc.col_op("sub", a, b)).all()
    assert (backend.neg(da).to_numpy() == orc.col_op("neg", a)).all()
    inv = backend.batchInverse(db)
    assert (inv.to_numpy() == orc.m31_batch_inverse(b)).all()
    assert (backend.mul(inv, db).to_numpy() == 1).all()
    with pytest.raises(L.TstwoError, match="0 has no inverse"):
        backend.batchInverse(T.HipColumn([1, 2, 0, 4]))


# ---------------------------------------------------------------- circle.test.ts / circleEvaluation.test.ts
@pytest.mark.parametrize("log_size", [1, 2, 3])
def test_evaluate_matches_eval_at_point(log_size):
    """circle.test.ts:52-97 (sizes 2/4/8) — with the TRUE ordering (no log-3 swap): domain.at(i) <-> bitrev index."""
    coeffs = [T.M31(i + 1) for i in range(1 << log_size)]
    poly = T.HipCirclePoly(coeffs)
    domain = T.CanonicCoset(log_size).circleDomain()
    ev = poly.evaluate(domain).bitReverse().toCpu()
    for i in range(domain.size()):
        p = domain.at(i)
        pt = T.CirclePoint(T.QM31.from_(p.x), T.QM31.from_(p.y))
        assert poly.evalAtPoint(pt) == T.QM31.from_(ev[i]), (log_size, i)


def test_log3_compat_swap():
    coeffs = rand_column(3, 8)
    domain = T.CanonicCoset(3).circleDomain()
    true = T.HipCirclePoly(coeffs).evaluate(domain).values.to_numpy()
    T.HipCirclePoly.compatLog3Swap = True
    try:
        compat = T.HipCirclePoly(coeffs).evaluate(domain)
        v = compat.values.to_numpy()
        assert v[5] == true[7] and v[7] == true[5] and (np.delete(v, [5, 7]) == np.delete(true, [5, 7])).all()
        assert (compat.interpolate().coeffs.to_numpy() == coeffs).all()
    finally:
        T.HipCirclePoly.compatLog3Swap = False


@pytest.mark.parametrize("log_size", [1, 2, 3, 4, 5, 6, 10, 14])
def test_interpolate_evaluate_roundtrip(log_size):
    """circle.test.ts:236-256."""
    coeffs = rand_column(log_size, 1 << log_size)
    domain = T.CanonicCoset(log_size).circleDomain()
    ev = T.HipCirclePoly(coeffs).evaluate(domain)
    assert (ev.interpolate().coeffs.to_numpy() == coeffs).all()
    tw = T.precompute_twiddles(T.Coset.half_odds(log_size + 2))          # a bigger tree serves the domain
    ev2 = T.HipCirclePoly(coeffs).evaluateWithTwiddles(domain, tw)
    assert (ev2.values.to_numpy() == ev.values.to_numpy()).all()
    assert (ev2.interpolateWithTwiddles(tw).coeffs.to_numpy() == coeffs).all()


def test_evaluate_on_extended_domain_and_errors():
    coeffs = rand_column(9, 1 << 5)
    poly = T.HipCirclePoly(coeffs)
    big = T.CanonicCoset(8).circleDomain()
    ev = poly.evaluate(big)                                            # blowup 3: zero-extension fused by the wrapper
    ext = np.concatenate([coeffs, np.zeros((1 << 8) - 32, dtype=np.uint32)])
    otw, _ = orc.precompute_twiddles(big.halfCoset.initial_index.value, 7, inverse=False)
    assert (ev.values.to_numpy() == orc.cfft_evaluate(ext, 8, big.halfCoset.initial_index.value, otw, 7)).all()
    assert (poly.extend(7).coeffs.to_numpy()[:32] == coeffs).all() and poly.extend(7).logSize() == 7
    with pytest.raises(ValueError, match="log size too small"):
        poly.extend(3)
    with pytest.raises(ValueError, match="log size too small"):
        poly.evaluate(T.CanonicCoset(3).circleDomain())
    with pytest.raises(ValueError, match="twiddle tree mismatch"):
        poly.evaluateWithTwiddles(big, T.precompute_twiddles(T.Coset.half_odds(4)))
    with pytest.raises(ValueError, match="twiddle tree mismatch"):
        poly.evaluateWithTwiddles(big, T.precompute_twiddles(T.Coset.half_odds(9).shift(T.CirclePointIndex(12345))))
    with pytest.raises(L.TstwoError, match="0 has no inverse"):     # a subgroup coset contains x = 0 (reference throws too)
        T.precompute_twiddles(T.Coset.subgroup(9))


def test_batched_poly_ops_and_secure_evaluation():
    n = 9
    domain = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    cols = [rand_column(20 + c, 1 << n) for c in range(5)]
    evs = T.evaluate_polynomials([T.HipCirclePoly(c) for c in cols], domain, tw)
    otw, _ = orc.precompute_twiddles(domain.halfCoset.initial_index.value, n - 1, inverse=False)
    for c, e in zip(cols, evs):
        assert (e.values.to_numpy() == orc.cfft_evaluate(c, n, domain.halfCoset.initial_index.value, otw, n - 1)).all()
    back = T.interpolate_columns(evs, tw)
    for c, p in zip(cols, back):
        assert (p.coeffs.to_numpy() == c).all()
    sec = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs[:4]]))
    for c, p in zip(cols[:4], sec.interpolateWithTwiddles(tw)):
        assert (p.coeffs.to_numpy() == c).all()


# ---------------------------------------------------------------- fri.test.ts / backend/cpu/fri.test.ts
def test_fold_line_is_2_fe_plus_alpha_fo():
    """fri.test.ts:206-254: folding the evaluations of f on D equals 2*(f_e + alpha*f_o) on D.double()."""
    log = 7
    alpha = T.QM31.from_u32_unchecked(2, 1, 0, 0)   # BaseField-like alpha keeps the check in M31 x QM31
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    domain = T.LineDomain(T.Coset.half_odds(log))
    # f(x) = sum c_k x^k evaluated with python ints; even/odd parts in pi(x) = 2x^2 - 1
    rng = np.random.default_rng(5)
    even = [int(v) for v in rng.integers(0, P, size=4)]
    odd = [int(v) for v in rng.integers(0, P, size=4)]

    def poly(cs, x):
        acc = 0
        for c in reversed(cs):
            acc = (acc * x + c) % P
        return acc

    def f(x):
        px = (2 * x * x - 1) % P
        return (poly(even, px) + x * poly(odd, px)) % P

    vals_nat = [f(domain.at(i).value) for i in range(domain.size())]
    vals = [vals_nat[T.bit_reverse_index(i, log)] for i in range(domain.size())]
    ev = T.LineEvaluation(domain, T.SecureColumnByCoords.from_([(v, 0, 0, 0) for v in vals]))
    tw = T.precompute_twiddles(T.Coset.half_odds(log))
    folded = T.fold_line(ev, alpha, tw)
    assert folded.domain() == domain.double() and folded.len() == domain.size() // 2
    got = folded.values.to_vec()
    d2 = domain.double()
    for i in range(d2.size()):
        x = d2.at(T.bit_reverse_index(i, log - 1)).value
        exp = T.QM31.from_(T.M31(poly(even, x))).add(alpha.mulM31(T.M31(poly(odd, x)))).double()
        assert got[i] == exp
    # without a tree (or with a non-matching one) the per-element path gives the same values
    assert T.fold_line(ev, alpha).values.to_numpy()[0].tolist() == folded.values.to_numpy()[0].tolist()
    with pytest.raises(ValueError, match="fold_line: Evaluation too small"):
        T.fold_line(T.LineEvaluation(T.LineDomain(T.Coset.half_odds(0)), T.SecureColumnByCoords.zeros(1)), alpha)


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_fold_circle_into_line_vs_oracle(n):
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    domain = T.CanonicCoset(n).circleDomain()
    src_np = [rand_column(30 + k, 1 << n) for k in range(4)]
    dst_np = [rand_column(40 + k, 1 << (n - 1)) for k in range(4)]
    src = T.SecureEvaluation(domain, T.SecureColumnByCoords.from_numpy(src_np))
    dst = T.LineEvaluation(T.LineDomain(domain.halfCoset), T.SecureColumnByCoords.from_numpy(dst_np))
    tw = T.precompute_twiddles(T.Coset.half_odds(max(n - 1, 1) + 1))
    T.fold_circle_into_line(dst, src, alpha, tw)
    exp = orc.fold_circle_into_line(dst_np, src_np, n, domain.halfCoset.initial_index.value, alpha.tup())
    for g, e in zip(dst.values.to_numpy(), exp):
        assert (g == e).all()
    with pytest.raises(ValueError, match="fold_circle_into_line: Length mismatch"):
        T.fold_circle_into_line(T.LineEvaluation.new_zero(T.LineDomain(T.Coset.half_odds(n + 1))), src, alpha)


def test_decompose_reconstruction():
    """backend/cpu/fri.test.ts:44-72,124-184."""
    n = 6
    domain = T.CanonicCoset(n).circleDomain()
    cols = [rand_column(50 + k, 1 << n) for k in range(4)]
    g, lam = T.decompose(T.SecureEvaluation(domain, T.SecureColumnByCoords.from_numpy(cols)))
    exp, elam = orc.decompose(cols)
    assert lam.tup() == elam
    for a, b in zip(g.values.to_numpy(), exp):
        assert (a == b).all()


# ---------------------------------------------------------------- vcs/prover.test.ts, backend/cpu/blake2.test.ts
def test_merkle_prover_commit_and_layers(golden):
    e = golden["merkle_lcg"]                      # prepareMerkle data (vcs/test_utils.ts:47-144)
    cols = [T.HipColumn(np.array(c, dtype=np.uint32)) for c in e["cols"]]
    tree = T.MerkleProver.commit(cols)
    assert tree.root().hex() == e["root"]
    assert [[h.hex() for h in layer.toCpu()] for layer in tree.layers] == e["layers"]
    assert len(tree.layers[0]) == 1 and tree.layers[0].at(0) == tree.root()
    empty = T.MerkleProver.commit([])
    assert empty.root().hex() == golden["blake2s_kat"][""]
    # commitOnLayer == hashNode per node (backend/cpu/blake2.test.ts:55-176)
    vals = [rand_column(60 + c, 8) for c in range(3)]
    layer = T.HipMerkleOps.commitOnLayer(3, None, [T.HipColumn(v) for v in vals])
    for i, h in enumerate(layer.toCpu()):
        assert h == orc.hash_node(None, [v[i] for v in vals])
    up = T.HipMerkleOps.commitOnLayer(2, layer, [T.HipColumn(v[:4]) for v in vals[:1]])
    hs = layer.toCpu()
    for i, h in enumerate(up.toCpu()):
        assert h == orc.hash_node((hs[2 * i], hs[2 * i + 1]), [vals[0][i]])


# ---------------------------------------------------------------- quotients
def test_quotients_are_low_degree():
    """pcs/quotients.ts:179-201 (Rust test_quotients_are_low_degree): log 7 poly, blowup 1, coeff qm31(1,2,3,4)."""
    LOG_SIZE, LOG_BLOWUP = 7, 1
    coeffs = rand_column(70, 1 << LOG_SIZE)
    poly = T.HipCirclePoly(coeffs)
    eval_domain = T.CanonicCoset(LOG_SIZE + 1).circleDomain()
    ev = poly.evaluate(eval_domain)
    point = T.SECURE_FIELD_CIRCLE_GEN
    value = poly.evalAtPoint(point)
    coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
    quot_domain = T.CanonicCoset(LOG_SIZE + LOG_BLOWUP).circleDomain()
    q = T.accumulateQuotients(quot_domain, [ev], coeff, [T.ColumnSampleBatch(point, [(0, value)])], LOG_BLOWUP)
    tw = T.precompute_twiddles(quot_domain.halfCoset)
    for p in q.interpolateWithTwiddles(tw):
        c = p.coeffs.to_numpy()
        assert not c[1 << LOG_SIZE:].any() and c[: 1 << LOG_SIZE].any()
    exp = orc.accumulate_quotients(quot_domain.halfCoset.initial_index.value, LOG_SIZE + 1, [ev.values.to_numpy()], coeff.tup(),
                                   [(point.x.tup(), point.y.tup(), [(0, value.tup())])])
    for a, b in zip(q.values.to_numpy(), exp):
        assert (a == b).all()


def test_quotients_ts_compat_variant():
    """The TS port's deviations (per-CM31 conjugation; Pr/Pi from c0.real/c0.imag) as an opt-in: checked against the
    oracle's generic row loop fed with the same host constants."""
    from tstwo_amd.quotients import quotientConstants
    n = 6
    domain = T.CanonicCoset(n).circleDomain()
    cols = [rand_column(80 + c, 1 << n) for c in range(2)]
    point = T.SECURE_FIELD_CIRCLE_GEN
    vals = [T.QM31.from_u32_unchecked(7, 8, 9, 10), T.QM31.from_u32_unchecked(11, 12, 13, 14)]
    coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
    batches = [T.ColumnSampleBatch(point, [(0, vals[0]), (1, vals[1])])]
    got = T.accumulateQuotients(domain, [T.HipColumn(c) for c in cols], coeff, batches, 1, ts_compat=True)
    lc, bc = quotientConstants(batches, coeff, ts_compat=True)
    abc = [t.tup() for trip in lc[0] for t in trip]
    x, y = point.x, point.y
    exp = orc.accumulate_quotients_consts(domain.halfCoset.initial_index.value, n, cols, [0, 2], [0, 1], abc, [bc[0].tup()],
                                          [(x.c0.real.value, 0)], [(y.c0.real.value, 0)], [(x.c0.imag.value, 0)], [(y.c0.imag.value, 0)])
    for a, b in zip(got.values.to_numpy(), exp):
        assert (a == b).all()


def test_accumulate(backend):
    a = T.SecureColumnByCoords.from_numpy([rand_column(90 + k, 100) for k in range(4)])
    b = T.SecureColumnByCoords.from_numpy([rand_column(95 + k, 100) for k in range(4)])
    exp = orc.accumulate(a.to_numpy(), b.to_numpy())
    T.accumulate(a, b)
    for x, y in zip(a.to_numpy(), exp):
        assert (x == y).all()
    with pytest.raises(ValueError, match="column length mismatch"):
        T.accumulate(a, T.SecureColumnByCoords.zeros(3))


# ---------------------------------------------------------------- vcs/blake2_merkle.test.ts, vcs/prover.test.ts (decommit / verify)
def _prepare_merkle(golden):
    """prepareMerkle (vcs/test_utils.ts:47-144): LCG seed 0, 10 columns of log 3..4, 3 queries per log size."""
    e = golden["merkle_lcg"]
    s = [0]

    def nxt():
        s[0] = (1664525 * s[0] + 1013904223) % 2**32
        return s[0]

    log_sizes = [3 + nxt() % 2 for _ in range(10)]
    cols_np = [[nxt() % (1 << 30) for _ in range(1 << lg)] for lg in log_sizes]
    assert log_sizes == e["log_sizes"] and cols_np == e["cols"]
    queries = {}
    for lg in (4, 3):
        queries[lg] = sorted(set(nxt() % (1 << lg) for _ in range(3)))
    cols = [T.HipColumn(np.array(c, dtype=np.uint32)) for c in cols_np]
    tree = T.MerkleProver.commit(cols)
    values, dec = tree.decommit(queries, cols)
    verifier = T.MerkleVerifier(T.Blake2sMerkleHasher(), tree.root(), log_sizes)
    return queries, dec, values, verifier, cols_np, log_sizes


def test_merkle_decommit_verify_roundtrip(golden):
    queries, dec, values, verifier, cols_np, log_sizes = _prepare_merkle(golden)
    verifier.verify(queries, values, dec)                                  # test_merkle_success
    # queried values are exactly the queried rows of the columns of that size, largest layer first
    exp = []
    order = sorted(range(10), key=lambda i: -log_sizes[i])
    for lg in (4, 3):
        for q in queries[lg]:
            exp += [cols_np[i][q] for i in order if log_sizes[i] == lg]
    assert [v.value for v in values] == exp


def test_merkle_verify_failures(golden):
    """vcs/blake2_merkle.test.ts:30-100: the verifier's error cases."""
    import copy
    queries, dec, values, verifier, *_ = _prepare_merkle(golden)
    d = copy.deepcopy(dec); d.hashWitness[4] = bytes(32)
    with pytest.raises(ValueError, match="Root mismatch."):
        verifier.verify(queries, values, d)
    v = list(values); v[6] = T.M31(0)
    with pytest.raises(ValueError, match="Root mismatch."):
        verifier.verify(queries, v, dec)
    d = copy.deepcopy(dec); d.hashWitness.pop()
    with pytest.raises(ValueError, match="Witness is too short"):
        verifier.verify(queries, values, d)
    d = copy.deepcopy(dec); d.hashWitness.append(bytes(32))
    with pytest.raises(ValueError, match="Witness is too long."):
        verifier.verify(queries, values, d)
    d = copy.deepcopy(dec); d.columnWitness.append(T.M31(0))
    with pytest.raises(ValueError, match="Witness is too long."):
        verifier.verify(queries, values, d)
    with pytest.raises(ValueError, match="too many Queried values"):
        verifier.verify(queries, list(values) + [T.M31(0)], dec)
    with pytest.raises(ValueError, match="too few queried values"):
        verifier.verify(queries, list(values)[:-1], dec)


def test_merkle_decommit_large():
    """log 16, 6 columns of two sizes, 40 random queries: decommit from device layers verifies against the root."""
    rng = np.random.default_rng(7)
    log_sizes = [16, 16, 16, 14, 14, 16]
    cols = [T.HipColumn(rand_column(300 + i, 1 << lg)) for i, lg in enumerate(log_sizes)]
    tree = T.MerkleProver.commit(cols)
    queries = {16: sorted(set(int(x) for x in rng.integers(0, 1 << 16, 40))), 14: sorted(set(int(x) for x in rng.integers(0, 1 << 14, 40)))}
    values, dec = tree.decommit(queries, cols)
    T.MerkleVerifier(T.Blake2sMerkleHasher(), tree.root(), log_sizes).verify(queries, values, dec)
    assert len(dec.hashWitness) > 0 and all(len(h) == 32 for h in dec.hashWitness)
    # the in-library walk (tstwo_merkle_decommit) == the reference's walk on the host mirror
    v2, d2 = tree._decommit_walk(queries, cols)
    assert [v.value for v in values] == [v.value for v in v2]
    assert dec.hashWitness == d2.hashWitness and [v.value for v in dec.columnWitness] == [v.value for v in d2.columnWitness]


def test_merkle_decommit_capi_capacity_and_errors():
    """tstwo_merkle_decommit: too-small buffers return the required counts; out-of-layer queries are rejected."""
    cols = [T.HipColumn(rand_column(350 + i, 1 << 6)) for i in range(3)]
    tree = T.MerkleProver.commit(cols)
    q = (C.c_uint64 * 2)(3, 40)
    qp = (C.POINTER(C.c_uint64) * 1)(C.cast(q, C.POINTER(C.c_uint64)))
    nq = (C.c_size_t * 1)(2)
    n_q, n_h, n_w = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    args = lambda: (C.c_void_p(tree._buf.ptr), 6, L.ptr_array([c.ptr for c in cols]), L.u32x([6, 6, 6]), 3, L.u32x([6]), qp, nq, 1,
                    None, C.byref(n_q), None, C.byref(n_h), None, C.byref(n_w))
    with pytest.raises(L.TstwoError, match="output buffer too small"):
        L.call("tstwo_merkle_decommit", *args())
    assert (n_q.value, n_w.value) == (6, 0) and n_h.value == len(tree.decommit({6: [3, 40]}, cols)[1].hashWitness)
    q[1] = 64
    with pytest.raises(L.TstwoError, match="outside its layer"):
        L.call("tstwo_merkle_decommit", *args())


# ---------------------------------------------------------------- fri.test.ts: FriProver.commit (real Merkle + channel wiring)
def _secure_low_degree_eval(log_deg, log_blowup, seed):
    """A SecureEvaluation of degree < 2^log_deg on the canonic domain of log size log_deg + log_blowup."""
    domain = T.CanonicCoset(log_deg + log_blowup).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(rand_column(seed + k, 1 << log_deg)) for k in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    return T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs])), tw


def test_fri_commit_low_degree_and_transcript():
    """fri.test.ts (commit phase): a low-degree column folds down to a last layer within the degree bound; the
    transcript (roots mixed, alphas drawn) and every layer are reproduced with the CPU oracle."""
    LOG_DEG, BLOW = 8, 2
    cfg = T.FriConfig(2, BLOW, 3)
    col, tw = _secure_low_degree_eval(LOG_DEG, BLOW, 1000)
    ch = T.Blake2sChannel()
    prover = T.FriProver.commit(ch, cfg, [col], tw)
    assert len(prover.last_layer_poly) == 1 << cfg.log_last_layer_degree_bound
    n = LOG_DEG + BLOW
    assert len(prover.inner_layers) == (n - 1) - (cfg.log_last_layer_degree_bound + BLOW)
    # replay with the oracle
    ch2 = T.Blake2sChannel()
    src = col.values.to_numpy()
    _, root = orc.merkle_commit(src, [n] * 4)
    assert prover.first_layer.merkle_tree.root() == root
    ch2.mix_root(root)
    alpha = ch2.draw_felt()
    half = col.domain.halfCoset.initial_index.value
    cur = orc.fold_circle_into_line([np.zeros(1 << (n - 1), dtype=np.uint32)] * 4, src, n, half, alpha.tup())
    k, coset_init = n - 1, half
    for layer in prover.inner_layers:
        for a, b in zip(layer.evaluation.values.to_numpy(), cur):
            assert (a == b).all()
        _, r = orc.merkle_commit(cur, [k] * 4)
        assert layer.merkle_tree.root() == r
        ch2.mix_root(r)
        alpha = ch2.draw_felt()
        cur = orc.fold_line(cur, k, coset_init, alpha.tup())
        coset_init = (coset_init * 2) & 0x7FFFFFFF
        k -= 1
    ch2.mix_felts(prover.last_layer_poly.coeffs)
    assert ch.digest() == ch2.digest()


def test_fri_commit_rejects_high_degree_and_bad_inputs():
    cfg = T.FriConfig(2, 2, 3)
    col, tw = _secure_low_degree_eval(9, 1, 2000)       # degree 2^9 on a log-10 domain: blowup 1 < config's 2
    with pytest.raises(ValueError, match="invalid degree"):
        T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw)
    with pytest.raises(ValueError, match="no columns"):
        T.FriProver.commit(T.Blake2sChannel(), cfg, [], tw)
    good, tw2 = _secure_low_degree_eval(6, 2, 3000)
    with pytest.raises(ValueError, match="column sizes not decreasing"):
        T.FriProver.commit(T.Blake2sChannel(), cfg, [good, good], tw2)


def test_fri_commit_two_columns_mixed_sizes():
    """Two circle columns (log 10 and log 8): the smaller one is folded in when the line layer reaches its size."""
    cfg = T.FriConfig(1, 2, 3)
    big, tw = _secure_low_degree_eval(8, 2, 4000)
    small, _ = _secure_low_degree_eval(6, 2, 5000)
    prover = T.FriProver.commit(T.Blake2sChannel(), cfg, [big, small], tw)
    assert len(prover.last_layer_poly) == 2
    assert len(prover.first_layer.merkle_tree.layers) == 11


# ---------------------------------------------------------------- fri.test.ts: decommit + FriVerifier (Rust test_fri_* family)
def _fri_prove(cols, tw, cfg):
    ch = T.Blake2sChannel()
    prover = T.FriProver.commit(ch, cfg, cols, tw)
    proof, positions = prover.decommit(ch)
    return prover, proof, positions


def _query_evals(cols, positions):
    return [c.values.gather(positions[c.domain.logSize()]) for c in cols]


def _fri_verify(cfg, proof, bounds, query_evals, expect_positions=None):
    ch = T.Blake2sChannel()
    v = T.FriVerifier.commit(ch, cfg, proof, [T.CirclePolyDegreeBound(b) for b in bounds])
    pos = v.sample_query_positions(ch)
    if expect_positions is not None:
        assert pos == expect_positions                   # verifier's transcript reproduces the prover's queries
    v.decommit(query_evals)
    return v


def test_fri_prove_verify_single_column():
    """Rust fri.rs valid_proof_passes_verification: commit + decommit on the GPU, verify on the host."""
    cfg = T.FriConfig(2, 2, 12)
    col, tw = _secure_low_degree_eval(8, 2, 6000)
    _, proof, positions = _fri_prove([col], tw, cfg)
    assert len(proof.inner_layers) == 9 - 4
    _fri_verify(cfg, proof, [8], _query_evals([col], positions), positions)


def test_fri_prove_verify_ts_compatible_transcript():
    """tstwo_amd.set_semantics("ts"): the channel keeps the TS port's draw_felt queue (channel/blake2.ts:177-184).  Prover
    (GPU folds and trees, host transcript: the device channel implements Rust's draw only, and says so) and verifier must
    agree with each other, and disagree with the Rust-mode transcript of the same column."""
    import warnings
    cfg = T.FriConfig(2, 2, 12)
    col, tw = _secure_low_degree_eval(8, 2, 6100)
    _, rust_proof, rust_positions = _fri_prove([col], tw, cfg)
    T.set_semantics("ts")
    try:
        assert T.Blake2sChannel().ts_compat and T.get_semantics() == "ts"
        ch = T.Blake2sChannel()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            prover = T.FriProver.commit(ch, cfg, [col], tw)
        assert any("ts-compatible channel" in str(x.message) for x in w)
        proof, positions = prover.decommit(ch)
        _fri_verify(cfg, proof, [8], _query_evals([col], positions), positions)     # _fri_verify builds its channel under "ts" too
        assert positions != rust_positions or proof.last_layer_poly.coeffs != rust_proof.last_layer_poly.coeffs
        # a Rust-mode verifier draws other challenges from the same proof: it must not accept it
        with pytest.raises(Exception):
            v = T.FriVerifier.commit(T.Blake2sChannel(ts_compat=False), cfg, proof, [T.CirclePolyDegreeBound(8)])
            v.decommit(_query_evals([col], v.sample_query_positions(T.Blake2sChannel(ts_compat=False))))
    finally:
        T.set_semantics("rust")
    assert not T.Blake2sChannel().ts_compat


def test_fri_prove_verify_mixed_degree_columns():
    """Rust fri.rs valid_mixed_degree_proof_passes_verification: columns of log degree 9, 7, 5 under one first-layer tree."""
    cfg = T.FriConfig(1, 2, 10)
    cols, tw = [], None
    for i, lg in enumerate([9, 7, 5]):
        c, t = _secure_low_degree_eval(lg, 2, 7000 + 10 * i)
        cols.append(c)
        tw = tw or t
    _, proof, positions = _fri_prove(cols, tw, cfg)
    assert sorted(positions) == [7, 9, 11]
    _fri_verify(cfg, proof, [9, 7, 5], _query_evals(cols, positions), positions)


def test_fri_verifier_rejects_tampering():
    """Rust fri.rs proof_with_*_fails_verification cases, with the reference's FriVerificationError texts."""
    import copy
    cfg = T.FriConfig(2, 2, 8)
    col, tw = _secure_low_degree_eval(8, 2, 8000)
    _, proof, positions = _fri_prove([col], tw, cfg)
    evals = _query_evals([col], positions)
    # removed inner layer
    bad = copy.deepcopy(proof)
    bad.inner_layers.pop()
    with pytest.raises(T.FriVerificationError, match="invalid number of FRI layers"):
        _fri_verify(cfg, bad, [8], evals)
    # added inner layer
    bad = copy.deepcopy(proof)
    bad.inner_layers.append(copy.deepcopy(bad.inner_layers[-1]))
    with pytest.raises(T.FriVerificationError, match="invalid number of FRI layers"):
        _fri_verify(cfg, bad, [8], evals)
    # an evaluation of the first layer changed: the transcript is unchanged, the Merkle check of layer 0 fails
    bad_evals = [list(evals[0])]
    bad_evals[0][0] = bad_evals[0][0].add(T.QM31.one())
    with pytest.raises(T.FriVerificationError, match="do not resolve to their commitment in the first layer"):
        _fri_verify(cfg, proof, [8], bad_evals)
    # inner-layer witness value changed
    bad = copy.deepcopy(proof)
    assert bad.inner_layers[1].fri_witness
    bad.inner_layers[1].fri_witness[0] = bad.inner_layers[1].fri_witness[0].add(T.QM31.one())
    with pytest.raises(T.FriVerificationError, match="do not resolve to their commitment in inner layer 1"):
        _fri_verify(cfg, bad, [8], evals)
    # inner-layer witness too short
    bad = copy.deepcopy(proof)
    bad.inner_layers[0].fri_witness.pop()
    with pytest.raises(T.FriVerificationError, match="evaluations are invalid in inner layer 0"):
        _fri_verify(cfg, bad, [8], evals)
    # last layer polynomial of too high degree
    bad = copy.deepcopy(proof)
    bad.last_layer_poly = T.LinePoly(list(bad.last_layer_poly.coeffs) * 2)
    with pytest.raises(T.FriVerificationError, match="degree of last layer is invalid"):
        _fri_verify(cfg, bad, [8], evals)
    # last layer polynomial changed (Rust proof_with_invalid_last_layer_fails_verification); changes the transcript too, so
    # replay the prover's queries directly
    bad = copy.deepcopy(proof)
    bad.last_layer_poly = T.LinePoly([bad.last_layer_poly.coeffs[0].add(T.QM31.one())] + list(bad.last_layer_poly.coeffs[1:]))
    ch = T.Blake2sChannel()
    v = T.FriVerifier.commit(ch, cfg, bad, [T.CirclePolyDegreeBound(8)])
    with pytest.raises(T.FriVerificationError, match="evaluations in the last layer are invalid"):
        v.decommit_on_queries(T.Queries(positions[10], 10), evals)


def test_fri_constant_last_layer_and_invalid_query_domain():
    """fri.test.ts valid_proof_with_constant_last_layer_passes_verification (query [5], last layer bound 0) and
    decommit_queries_on_invalid_domain_fails_verification."""
    LOG_DEG, BLOW = 3, 2
    col, tw = _secure_low_degree_eval(LOG_DEG, BLOW, 8500)
    queries = T.Queries.from_positions([5], LOG_DEG + BLOW)
    cfg = T.FriConfig(0, BLOW, len(queries))
    prover = T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw)
    assert len(prover.last_layer_poly) == 1
    proof = prover.decommit_on_queries(queries)
    v = T.FriVerifier.commit(T.Blake2sChannel(), cfg, proof, [T.CirclePolyDegreeBound(LOG_DEG)])
    v.decommit_on_queries(queries, [col.values.gather(queries.positions)])
    with pytest.raises(ValueError, match="Domain size mismatch"):
        v.decommit_on_queries(T.Queries.from_positions([2], LOG_DEG + BLOW - 1), [col.values.gather([2])])


def test_fri_prove_verify_log16():
    """A larger instance (degree 2^14, blowup 2, 20 queries): 11 inner layers stay on the GPU, proof verifies."""
    cfg = T.FriConfig(3, 2, 20)
    col, tw = _secure_low_degree_eval(14, 2, 9000)
    _, proof, positions = _fri_prove([col], tw, cfg)
    _fri_verify(cfg, proof, [14], _query_evals([col], positions), positions)


def test_queries_generate_and_fold():
    """queries.test.ts: sorted unique positions below the domain size; fold halves and de-duplicates."""
    ch = T.Blake2sChannel()
    q = T.Queries.generate(ch, 31, 100)
    assert len(q) == 100 and list(q.positions) == sorted(set(q.positions)) and max(q.positions) < (1 << 31)
    q = T.Queries.generate(T.Blake2sChannel(), 6, 20)
    f = q.fold(2)
    assert f.log_domain_size == 4 and f.positions == sorted({p >> 2 for p in q.positions})
    with pytest.raises(TypeError, match="sorted in ascending order"):
        T.Queries([3, 1], 4)


# ---------------------------------------------------------------- pcs/prover.ts (Rust comment): commitment tree over HipBackend
def test_commitment_scheme_prover_roundtrip():
    """TreeBuilder.extend_evals -> interpolate; commit -> evaluate on the blown-up domain + Merkle + mix_root; then a
    decommit of queried rows verifies against the root.  Checked against the oracle end to end."""
    LOG, BLOW = 8, 2
    tw = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW + 1).circleDomain().halfCoset)      # one tree serves every size
    scheme = T.CommitmentSchemeProver(BLOW, tw)
    trace_domain = T.CanonicCoset(LOG).circleDomain()
    cols_np = [rand_column(6000 + c, 1 << LOG) for c in range(5)]
    small_np = [rand_column(6100 + c, 1 << (LOG - 2)) for c in range(2)]
    tb = scheme.tree_builder()
    span = tb.extend_evals([T.HipCircleEvaluation(trace_domain, c) for c in cols_np])
    assert span == (0, 0, 5)
    tb.extend_evals([T.HipCircleEvaluation(T.CanonicCoset(LOG - 2).circleDomain(), c) for c in small_np])
    ch = T.Blake2sChannel()
    tb.commit(ch)
    tree = scheme.trees[0]
    # oracle replay: interpolate on the trace domain, evaluate on the extended domain, commit
    def ext_eval(vals, log):
        h = T.CanonicCoset(log).circleDomain().halfCoset.initial_index.value
        otw, oitw = orc.precompute_twiddles(h, log - 1)
        coeffs = orc.cfft_interpolate(vals, log, h, oitw, log - 1)
        big = log + BLOW
        hb = T.CanonicCoset(big).circleDomain().halfCoset.initial_index.value
        otwb, _ = orc.precompute_twiddles(hb, big - 1, inverse=False)
        ext = np.concatenate([coeffs, np.zeros((1 << big) - coeffs.size, dtype=np.uint32)])
        return orc.cfft_evaluate(ext, big, hb, otwb, big - 1)
    exp = [ext_eval(c, LOG) for c in cols_np] + [ext_eval(c, LOG - 2) for c in small_np]
    for ev, e in zip(tree.evaluations, exp):
        assert (ev.values.to_numpy() == e).all()
    _, oroot = orc.merkle_commit(exp, [LOG + BLOW] * 5 + [LOG - 2 + BLOW] * 2)
    assert scheme.roots() == [oroot]
    ch2 = T.Blake2sChannel(); ch2.mix_root(oroot)
    assert ch.digest() == ch2.digest()
    queries = {LOG + BLOW: [3, 77, 500], LOG - 2 + BLOW: [0, 9]}
    values, dec = tree.decommit(queries)
    T.MerkleVerifier(T.Blake2sMerkleHasher(), oroot, [LOG + BLOW] * 5 + [LOG - 2 + BLOW] * 2).verify(queries, values, dec)


def test_commitment_scheme_commit_many_equals_commit_per_tree():
    """CommitmentSchemeProver.commit_many (the trees of one phase: one batched evaluation, tstwo_merkle_commit_many, mix_root per
    tree in order) == one commit() per tree: same evaluations, roots and channel state, for equally shaped trees (shared launches)
    and for trees of different shapes and mixed column sizes (tree-by-tree fallback inside the library)."""
    BLOW = 1
    tw = T.precompute_twiddles(T.CanonicCoset(18 + BLOW).circleDomain().halfCoset)
    for shapes in ([[17] * 16, [17] * 16, [17] * 16], [[12, 12, 10], [9], [12, 11, 11, 11, 8]]):
        polys = [[T.HipCirclePoly(T.HipColumn(rand_column(15000 + 100 * ti + c, 1 << lg))) for c, lg in enumerate(logs)] for ti, logs in enumerate(shapes)]
        a, b = T.CommitmentSchemeProver(BLOW, tw), T.CommitmentSchemeProver(BLOW, tw)
        cha, chb = T.Blake2sChannel(), T.Blake2sChannel()
        a.commit_many(polys, cha)
        for ps in polys:
            b.commit(ps, chb)
        assert a.roots() == b.roots() and cha.digest() == chb.digest()
        for ta, tb_ in zip(a.trees, b.trees):
            for ea, eb in zip(ta.evaluations, tb_.evaluations):
                assert ea.domain.log_size() == eb.domain.log_size() and (ea.values.to_numpy() == eb.values.to_numpy()).all()
        # and against the oracle for the first tree
        logs = shapes[0]
        exp = []
        for p_, lg in zip(polys[0], logs):
            big = lg + BLOW
            hb = T.CanonicCoset(big).circleDomain().halfCoset.initial_index.value
            otw, _ = orc.precompute_twiddles(hb, big - 1, inverse=False)
            ext = np.concatenate([p_.coeffs.to_numpy(), np.zeros((1 << big) - (1 << lg), dtype=np.uint32)])
            exp.append(orc.cfft_evaluate(ext, big, hb, otw, big - 1))
        _, oroot = orc.merkle_commit(exp, [lg + BLOW for lg in logs])
        assert a.roots()[0] == oroot


def _pcs_setup(config, col_logs_per_tree, seed=11000):
    """Commit trees of random trace columns (evaluations on canonic trace domains) the way a stwo prover does."""
    blow = config.fri_config.log_blowup_factor
    max_log = max(lg for t in col_logs_per_tree for lg in t)
    tw = T.precompute_twiddles(T.CanonicCoset(max_log + blow).circleDomain().halfCoset)
    scheme = T.CommitmentSchemeProver(config, tw)
    ch = T.Blake2sChannel()
    config.mix_into(ch)
    for ti, logs in enumerate(col_logs_per_tree):
        tb = scheme.tree_builder()
        tb.extend_evals([T.HipCircleEvaluation(T.CanonicCoset(lg).circleDomain(), rand_column(seed + 100 * ti + c, 1 << lg))
                         for c, lg in enumerate(logs)])
        tb.commit(ch)
    return scheme, ch


def _pcs_verifier(config, col_logs_per_tree, roots):
    v = T.CommitmentSchemeVerifier(config)
    ch = T.Blake2sChannel()
    config.mix_into(ch)
    for logs, root in zip(col_logs_per_tree, roots):
        v.commit(root, logs, ch)
    return v, ch


def test_pcs_prove_values_and_verify():
    """pcs/prover.ts + pcs/verifier.ts (Rust text): commit two trees of mixed sizes, open every column at an out-of-domain
    point drawn from the channel (one column at two points), prove on the GPU, verify on the host."""
    config = T.PcsConfig(pow_bits=10, fri_config=T.FriConfig(2, 2, 8))
    logs = [[10, 10, 8], [10, 9]]
    scheme, ch = _pcs_setup(config, logs)
    point = T.CirclePoint.get_random_point(ch)
    shifted = point.add(T.SECURE_FIELD_CIRCLE_GEN)
    sampled_points = [[[point], [point, shifted], [point]], [[point], [point]]]
    proof = scheme.prove_values(sampled_points, ch)
    assert proof.commitments == scheme.roots() and len(proof.sampled_values[0][1]) == 2
    # sampled values are the polynomials' values at the point (eval_at_point parity is covered by the C-ABI tests)
    verifier, vch = _pcs_verifier(config, logs, proof.commitments)
    vpoint = T.CirclePoint.get_random_point(vch)
    assert vpoint.x.tup() == point.x.tup()
    vshift = vpoint.add(T.SECURE_FIELD_CIRCLE_GEN)
    verifier.verify_values([[[vpoint], [vpoint, vshift], [vpoint]], [[vpoint], [vpoint]]], proof, vch)
    assert ch.digest() == vch.digest()                       # both transcripts end in the same state


def test_pcs_verify_rejects_bad_proofs():
    import copy
    config = T.PcsConfig(pow_bits=6, fri_config=T.FriConfig(1, 1, 6))
    logs = [[7, 6]]
    scheme, ch = _pcs_setup(config, logs, seed=12000)
    point = T.CirclePoint.get_random_point(ch)
    pts = [[[point], [point]]]
    proof = scheme.prove_values(pts, ch)

    def verify(pr, cfg=config):
        v, vch = _pcs_verifier(cfg, logs, pr.commitments)
        T.CirclePoint.get_random_point(vch)
        v.verify_values(pts, pr, vch)
    verify(proof)
    # a queried trace value changed -> the tree's Merkle decommitment fails
    bad = copy.deepcopy(proof)
    bad.queried_values[0][0] = bad.queried_values[0][0].add(T.M31.one())
    with pytest.raises(T.VerificationError, match="Merkle verification failed"):
        verify(bad)
    # a sampled value changed -> different transcript; the proof cannot verify
    bad = copy.deepcopy(proof)
    bad.sampled_values[0][0][0] = bad.sampled_values[0][0][0].add(T.QM31.one())
    with pytest.raises((T.VerificationError, T.FriVerificationError)):
        verify(bad)
    # wrong nonce -> proof of work (or, with probability 2^-6, a later failure)
    bad = copy.deepcopy(proof)
    bad.proof_of_work += 1
    with pytest.raises((T.VerificationError, T.FriVerificationError)):
        verify(bad)
    # claiming a sampled value the polynomial does not take: the prover itself cannot produce a low-degree quotient
    scheme2, ch2 = _pcs_setup(config, logs, seed=12000)
    p2 = T.CirclePoint.get_random_point(ch2)
    orig = T.HipCirclePoly.eval_at_point_batch
    try:
        T.HipCirclePoly.eval_at_point_batch = staticmethod(lambda polys, pt: [v.add(T.QM31.one()) for v in orig(polys, pt)])
        with pytest.raises(ValueError, match="invalid degree"):
            scheme2.prove_values([[[p2], [p2]]], ch2)
    finally:
        T.HipCirclePoly.eval_at_point_batch = staticmethod(orig)


def test_eval_at_point_batch_matches_single():
    """tstwo_eval_at_point_batch == tstwo_eval_at_point per polynomial == the oracle (sizes straddling the 2^5 fold chunks,
    more than 64 columns so the batch is split)."""
    pt = T.SECURE_FIELD_CIRCLE_GEN.add(T.SECURE_FIELD_CIRCLE_GEN)
    for lg, n in ((1, 3), (4, 2), (5, 5), (6, 3), (11, 70), (16, 4)):
        cols = [rand_column(14000 + 10 * lg + i, 1 << lg) for i in range(n)]
        polys = [T.HipCirclePoly(c) for c in cols]
        got = T.HipCirclePoly.eval_at_point_batch(polys, pt)
        assert [g.tup() for g in got] == [p.evalAtPoint(pt).tup() for p in polys]
        assert got[0].tup() == tuple(orc.eval_at_point(cols[0], lg, pt.x.tup(), pt.y.tup()))
    with pytest.raises(ValueError, match="one size"):
        T.HipCirclePoly.eval_at_point_batch([T.HipCirclePoly(rand_column(1, 4)), T.HipCirclePoly(rand_column(2, 8))], pt)


def test_fri_answers_match_device_quotients():
    """fri_answers (host row quotients at the queries) == the device accumulate_quotients column gathered at the queries."""
    LOG, BLOW = 8, 1
    tw = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW).circleDomain().halfCoset)
    polys = [T.HipCirclePoly(rand_column(13000 + c, 1 << LOG)) for c in range(3)]
    domain = T.CanonicCoset(LOG + BLOW).circleDomain()
    evals = T.evaluate_polynomials(polys, domain, tw)
    pt = T.SECURE_FIELD_CIRCLE_GEN
    samples = [[T.PointSample(pt, p.evalAtPoint(pt))] for p in polys]
    coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
    quot = T.compute_fri_quotients(evals, samples, coeff, BLOW)[0]
    queries = [0, 5, 100, 511]
    got = quot.values.gather(queries)
    vals = [int(e.values.to_numpy()[q]) for q in queries for e in evals]
    ans = T.fri_answers([[LOG + BLOW] * 3], [samples], coeff, {LOG + BLOW: queries}, [[T.M31(v) for v in vals]], [{LOG + BLOW: 3}])
    assert [a.tup() for a in ans[0]] == [g.tup() for g in got]


# ---------------------------------------------------------------- proof_of_work / backend/cpu/grind.ts
@pytest.mark.parametrize("pow_bits", [0, 1, 8, 14, 20])
def test_grind_matches_sequential_reference_loop(pow_bits):
    ch = T.Blake2sChannel()
    ch.mix_u64(0x1234 + pow_bits)
    nonce = T.grind(ch, pow_bits)
    c = ch.clone(); c.mix_u64(nonce)
    assert c.trailing_zeros() >= pow_bits
    if pow_bits <= 14:                       # the reference's loop (grind.ts:31-42) returns the first such nonce
        n = 0
        while True:
            c = ch.clone(); c.mix_u64(n)
            if c.trailing_zeros() >= pow_bits:
                break
            n += 1
        assert nonce == n
    else:                                    # minimality at 2^20 scale: no smaller nonce in a sampled window passes
        for n in range(max(0, nonce - 2000), nonce):
            c = ch.clone(); c.mix_u64(n)
            assert c.trailing_zeros() < pow_bits


def test_line_interpolate_with_and_without_tree():
    """lineIfft x^-1 from the inverse twiddle tree == the reference's per-element domain.at(i).inverse()."""
    k = 5
    tw = T.precompute_twiddles(T.Coset.half_odds(8))
    coset = T.Coset.half_odds(8).repeated_double(3)
    ev = T.LineEvaluation(T.LineDomain(coset), T.SecureColumnByCoords.from_numpy([rand_column(7000 + c, 1 << k) for c in range(4)]))
    a = T.line_interpolate(ev, tw)
    b = T.line_interpolate(ev, None)
    assert [x.tup() for x in a] == [x.tup() for x in b]


@pytest.mark.parametrize("k", [0, 1, 2, 3, 7, 9, 12])
def test_line_interpolate_on_device_equals_reference_formulation(k):
    """tstwo_line_interpolate (one workgroup, x^-1 from the tree) == LineEvaluation.interpolate as the reference computes it
    (per-element domain.at(i).inverse(), poly/line.ts:312-390), and evaluating the resulting LinePoly at domain points gives the
    evaluation back (the property of poly/line.test.ts)."""
    from tstwo_amd.fri_prover import line_interpolate_device, line_interpolate_words
    root = T.Coset.half_odds(13)
    tw = T.precompute_twiddles(root)
    coset = root.repeated_double(13 - k)
    cols = [rand_column(7100 + 10 * k + c, 1 << k) for c in range(4)]
    ev = T.LineEvaluation(T.LineDomain(coset), T.SecureColumnByCoords.from_numpy(cols))
    buf = line_interpolate_device(ev, tw)
    assert buf is not None
    got = buf.download(count=4 << k).reshape(4, 1 << k)
    want = line_interpolate_words(ev, None)                       # the reference's per-element inverses
    assert np.array_equal(got, want)
    if 1 <= k <= 7:
        poly = T.LinePoly([T.QM31.from_u32_unchecked(*row) for row in got.T.tolist()])
        from tstwo_amd.circle import bit_reverse_index
        for i in (0, 1, (1 << k) - 1):
            x = T.QM31.from_(ev.domain().at(bit_reverse_index(i, k)))
            assert poly.eval_at_point(x).tup() == tuple(int(c[i]) for c in cols)


def test_line_interpolate_capi_errors():
    tw = T.precompute_twiddles(T.Coset.half_odds(6))
    cols = T.SecureColumnByCoords.from_numpy([rand_column(7300 + c, 1 << 13) for c in range(4)])
    out = L.DeviceBuffer(16 << 13)
    o4 = L.p4([out.ptr + (4 << 13) * c for c in range(4)])
    with pytest.raises(L.TstwoError, match="at most 2\\^12 values"):
        L.call("tstwo_line_interpolate", cols.ptrs(), 13, L.vp(tw.itwiddles.buf.ptr), tw.log_size, o4)
    with pytest.raises(L.TstwoError, match="Not enough twiddles!"):
        L.call("tstwo_line_interpolate", cols.ptrs(), 8, L.vp(tw.itwiddles.buf.ptr), tw.log_size, o4)
    with pytest.raises(L.TstwoError, match="null"):
        L.call("tstwo_line_interpolate", cols.ptrs(), 4, L.vp(0), tw.log_size, o4)


def test_fri_commit_device_last_layer_equals_host_last_layer(monkeypatch):
    """The last layer interpolated on the device (one read-back with the channel state) and on the host (TSTWO_FRI_HOST_LAST_LAYER,
    and the per-piece read-backs of TSTWO_FRI_SEPARATE_READBACKS) give the same last-layer polynomial and channel."""
    LOGD, BLOW = 9, 2
    domain = T.CanonicCoset(LOGD + BLOW).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(rand_column(7400 + c, 1 << LOGD)) for c in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    col = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs]))
    results = []
    for env in (None, "TSTWO_FRI_HOST_LAST_LAYER", "TSTWO_FRI_SEPARATE_READBACKS"):
        for bound in (0, 3, 5):
            if env:
                monkeypatch.setenv(env, "1")
            ch = T.Blake2sChannel()
            fp = T.FriProver.commit(ch, T.FriConfig(bound, BLOW, 10), [col], tw)
            if env:
                monkeypatch.delenv(env)
            results.append((env, bound, [c.tup() for c in fp.last_layer_poly.coeffs], ch.digest_bytes() if hasattr(ch, "digest_bytes") else ch._digest))
    base = {b: r for e, b, *r in results if e is None}
    for e, b, *r in results:
        assert r == base[b], (e, b)


# ---------------------------------------------------------------- poly/circle/secure_poly.ts, poly.ts:56-73, poly/utils.ts:78-100
def test_secure_circle_poly_roundtrip_and_eval():
    LOG = 7
    domain = T.CanonicCoset(LOG + 1).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    coeffs = [rand_column(16000 + k, 1 << LOG) for k in range(4)]
    sp = T.SecureCirclePoly([T.HipCirclePoly(c) for c in coeffs])
    assert sp.logSize() == LOG and len(sp.intoCoordinatePolys()) == 4
    ev = sp.evaluateWithTwiddles(domain, tw)
    assert isinstance(ev, T.SecureEvaluation) and ev.len() == 1 << (LOG + 1)
    back = ev.interpolateWithTwiddles(tw)
    assert isinstance(back, T.SecureCirclePoly)
    for c, p in zip(coeffs, back):
        got = p.coeffs.to_numpy()
        assert (got[:1 << LOG] == c).all() and not got[1 << LOG:].any()
        assert p.isInFftSpace(LOG) and p.isInFriSpace(LOG) and not p.isInFftSpace(LOG - 1)
    pt = T.SECURE_FIELD_CIRCLE_GEN
    cols = sp.evalColumnsAtPoint(pt)
    assert [c.tup() for c in cols] == [tuple(orc.eval_at_point(c, LOG, pt.x.tup(), pt.y.tup())) for c in coeffs]
    assert sp.evalAtPoint(pt).tup() == T.QM31.from_partial_evals(cols).tup()
    assert sp.evalAtPoint(pt, ts_compat=True).tup() == cols[0].tup()        # the TS port's coordinate-0 behaviour
    # a value of the secure evaluation is the polynomial at that domain point
    i = 37
    p_i = domain.at(T.bit_reverse_index(i, LOG + 1))
    as_q = T.CirclePoint(T.QM31.from_(p_i.x), T.QM31.from_(p_i.y))
    assert ev.values.at(i).tup() == sp.evalAtPoint(as_q).tup()
    with pytest.raises(ValueError, match="4 coordinate"):
        T.SecureCirclePoly([T.HipCirclePoly(coeffs[0])])


def test_domain_line_twiddles_from_tree_kat():
    """test/poly/domainLineTwiddles.test.ts:7-13: buffer [0..7], domain of log 3 -> [[4,5,6,7]... slices from the end."""
    buf = T.HipColumn(np.arange(8, dtype=np.uint32))
    dom = T.LineDomain(T.Coset.half_odds(3))
    views = T.domain_line_twiddles_from_tree(dom, buf)
    host = buf.to_numpy()
    assert [list(host[o:o + n]) for o, n in views] == [[0, 1, 2, 3], [4, 5], [6]]
    with pytest.raises(ValueError, match="Not enough twiddles!"):
        T.domain_line_twiddles_from_tree(T.LineDomain(T.Coset.half_odds(4)), buf)


# ---------------------------------------------------------------- row sharding of FRI layers / one big tree
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_sharded_fold_and_commit_match_single_gpu(world):
    """Virtual ranks in one process: every rank folds and commits only its contiguous rows; concatenated folds and the
    combined subtree roots equal the whole-layer results (the exchange itself is covered by the gloo test)."""
    from tstwo_amd import distributed as D
    n = 12
    domain = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    src_np = [rand_column(15000 + k, 1 << n) for k in range(4)]
    dst_np = [rand_column(15100 + k, 1 << (n - 1)) for k in range(4)]
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    src = T.SecureEvaluation(domain, T.SecureColumnByCoords.from_numpy(src_np))
    dst = T.LineEvaluation(T.LineDomain(domain.halfCoset), T.SecureColumnByCoords.from_numpy(dst_np))
    T.fold_circle_into_line(dst, src, alpha, tw)
    line = T.fold_line(dst, alpha, tw)
    whole_circle, whole_line = dst.values.to_numpy(), line.values.to_numpy()
    whole_root = T.MerkleProver.commit(dst.values.columns).root()
    got_circle, got_line, subroots = [], [], []
    for rank in range(world):
        s, c = D.shard_rows(1 << (n - 1), world, rank)
        d = T.SecureColumnByCoords.from_numpy([x[s:s + c] for x in dst_np])
        D.fold_circle_into_line_rows(d, T.SecureColumnByCoords.from_numpy([x[2 * s:2 * (s + c)] for x in src_np]), n, rank, world, alpha, tw)
        got_circle.append(d.to_numpy())
        subroots.append(T.MerkleProver.commit(d.columns).root())
        # next layer: this rank's rows of the line layer (2^(n-1) rows) fold to its rows of the layer below
        s2, c2 = D.shard_rows(1 << (n - 2), world, rank)
        shard = T.SecureColumnByCoords.from_numpy([x[2 * s2:2 * (s2 + c2)] for x in whole_circle])
        got_line.append(D.fold_line_rows(shard, n - 1, rank, world, alpha, tw).to_numpy())
    for k in range(4):
        assert (np.concatenate([g[k] for g in got_circle]) == whole_circle[k]).all()
        assert (np.concatenate([g[k] for g in got_line]) == whole_line[k]).all()
    assert D.combine_subtree_roots(subroots) == whole_root


def test_row_shard_rejects_misaligned():
    from tstwo_amd import distributed as D
    n = 6
    tw = T.precompute_twiddles(T.CanonicCoset(n).circleDomain().halfCoset)
    sh = T.SecureColumnByCoords.zeros(12)
    with pytest.raises(L.TstwoError, match="4-aligned"):
        L.call("tstwo_fri_fold_line_rows", sh.ptrs(), n - 1, 2, 6, C.c_void_p(tw.itwiddles.ptr), tw.log_size,
               L.u32x([1, 0, 0, 0]), T.SecureColumnByCoords.zeros(6).ptrs())


def test_interpolate_columns_mixed_domains():
    """poly/circle/ops.ts:73-82: interpolateColumns interpolates each column on its own domain.  Regression: columns of
    different sizes in one call used to be transformed with the first column's size (writing past the smaller buffers)."""
    tw = T.precompute_twiddles(T.CanonicCoset(9).circleDomain().halfCoset)
    logs = [8, 6, 9, 6, 8]
    cols = [rand_column(17000 + i, 1 << lg) for i, lg in enumerate(logs)]
    evs = [T.HipCircleEvaluation(T.CanonicCoset(lg).circleDomain(), c) for lg, c in zip(logs, cols)]
    polys = T.interpolate_columns(evs, tw)
    assert [p.logSize() for p in polys] == logs
    for p, e, c, lg in zip(polys, evs, cols, logs):
        assert (e.values.to_numpy() == c).all()                                  # evaluations survive
        h = T.CanonicCoset(lg).circleDomain().halfCoset.initial_index.value
        _, oitw = orc.precompute_twiddles(h, lg - 1)
        assert (p.coeffs.to_numpy() == orc.cfft_interpolate(c, lg, h, oitw, lg - 1)).all()


# ---------------------------------------------------------------- BASELINE.json full sizes (configs 3 and 4): spot checks
def _device_rand_secure(seed, n):
    return T.SecureColumnByCoords.from_numpy([rand_column(seed + k, n) for k in range(4)])


def test_config4_full_size_fold_and_merkle():
    """Config 4 at its real size (log 24 secure column): fold_circle_into_line and fold_line agree with the host
    SparseEvaluation fold (itself checked against the oracle on CPU) at random rows; the Merkle tree over the 4 coordinate
    columns answers random queries with a decommitment the host verifier (hashlib) accepts."""
    from tstwo_amd.fri_verifier import SparseEvaluation
    n = 24
    domain = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    src = T.SecureEvaluation(domain, _device_rand_secure(18000, 1 << n))
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    dst = T.LineEvaluation.new_zero(T.LineDomain(domain.halfCoset))
    T.fold_circle_into_line(dst, src, alpha, tw)
    rng = np.random.default_rng(24)
    rows = sorted(int(r) for r in rng.integers(0, 1 << (n - 1), size=12))
    pairs = src.values.gather([2 * r + k for r in rows for k in (0, 1)])
    got = dst.values.gather(rows)
    for j, r in enumerate(rows):
        s = SparseEvaluation([[pairs[2 * j], pairs[2 * j + 1]]], [T.bit_reverse_index(2 * r, n)])
        assert s.fold_circle(alpha, domain)[0].tup() == got[j].tup()
    line = T.fold_line(dst, alpha, tw)
    rows2 = sorted(int(r) for r in rng.integers(0, 1 << (n - 2), size=12))
    pairs2 = dst.values.gather([2 * r + k for r in rows2 for k in (0, 1)])
    got2 = line.values.gather(rows2)
    for j, r in enumerate(rows2):
        s = SparseEvaluation([[pairs2[2 * j], pairs2[2 * j + 1]]], [T.bit_reverse_index(2 * r, n - 1)])
        assert s.fold_line(alpha, dst.domain())[0].tup() == got2[j].tup()
    tree = T.MerkleProver.commit(src.values.columns)
    queries = {n: sorted(set(int(q) for q in rng.integers(0, 1 << n, size=16)))}
    values, dec = tree.decommit(queries, src.values.columns)
    assert len(dec.hashWitness) > 16 * 10
    T.MerkleVerifier(T.Blake2sMerkleHasher, tree.root(), [n] * 4).verify(queries, values, dec)


def test_config3_full_size_quotients_and_inverse():
    """Config 3 at its real size (4 columns, log 22): the device quotient column equals the host row quotients at random
    rows; QM31 batch inverse times its input is one everywhere."""
    n = 22
    domain = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(rand_column(19000 + c, 1 << (n - 1))) for c in range(4)]
    evals = T.evaluate_polynomials(polys, domain, tw)
    pt = T.SECURE_FIELD_CIRCLE_GEN
    samples = [[T.PointSample(pt, v)] for v in T.HipCirclePoly.eval_at_point_batch(polys, pt)]
    coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
    quot = T.compute_fri_quotients(evals, samples, coeff, 1)[0]
    rng = np.random.default_rng(22)
    queries = sorted(set(int(q) for q in rng.integers(0, 1 << n, size=10)))
    vals = [v for q in queries for v in (e.values.at(q) for e in evals)]
    ans = T.fri_answers([[n] * 4], [samples], coeff, {n: queries}, [vals], [{n: 4}])
    assert [a.tup() for a in ans[0]] == [g.tup() for g in quot.values.gather(queries)]
    # the quotient of degree-2^21 polynomials sampled at their true values is itself of low degree
    qp = quot.interpolateWithTwiddles(tw)
    assert all(p.isInFriSpace(n - 1) for p in qp)
    sec = _device_rand_secure(19500, 1 << n)
    inv = T.HipBackend().batchInverse(sec)
    prod = T.HipBackend().secureMul(sec, inv).to_numpy()
    assert (prod[0] == 1).all() and not prod[1].any() and not prod[2].any() and not prod[3].any()


def test_simd_twiddle_dbls_export():
    """backend/simd/fft/index.ts:161-203: layer l = doubled x of the first half of coset.repeated_double(l), bit-reversed;
    the inverse variant holds doubled inverses."""
    coset = T.Coset.half_odds(6)
    tree = T.precompute_twiddles(coset)
    dbls, idbls = T.get_twiddle_dbls(tree), T.get_twiddle_dbls(tree, inverse=True)
    assert [len(d) for d in dbls] == [32, 16, 8, 4, 2, 1]
    cur = coset
    for l, (d, di) in enumerate(zip(dbls, idbls)):
        half = cur.size() // 2
        xs = [cur.at(i).x for i in range(half)]
        lg = half.bit_length() - 1
        want = [(xs[T.bit_reverse_index(i, lg)].value * 2) & 0xFFFFFFFF for i in range(half)]
        assert list(map(int, d)) == want
        assert list(map(int, di)) == [(xs[T.bit_reverse_index(i, lg)].inverse().value * 2) & 0xFFFFFFFF for i in range(half)]
        cur = cur.double()


# ---------------------------------------------------------------- device-resident Blake2sChannel
def test_device_channel_matches_host_channel():
    """tstwo_channel_mix_root_draw_felt == Blake2sChannel.mix_root + draw_felt (hashlib), including the counters."""
    host = T.Blake2sChannel()
    host.mix_u64(12345)
    ref = host.clone()
    dch = T.DeviceChannel(host)
    roots = [bytes((7 * i + k) & 0xFF for k in range(32)) for i in range(5)]
    felts = L.DeviceBuffer(16 * 8)
    want = []
    for i, r in enumerate(roots):
        rb = L.DeviceBuffer(32)
        rb.upload(np.frombuffer(r, dtype=np.uint8))
        dch.mix_root_draw_felt(rb.ptr, felts.ptr + 16 * i)
        ref.mix_root(r)
        want.append(ref.draw_felt().tup())
    dch.mix_root_draw_felt(None, felts.ptr + 16 * 5)            # draw only
    want.append(ref.draw_felt().tup())
    got = felts.download(count=24).reshape(6, 4)
    assert [tuple(int(x) for x in row) for row in got] == want
    dch.sync_to_host()
    assert (host.digest(), host.n_challenges, host.n_sent) == (ref.digest(), ref.n_challenges, ref.n_sent)
    with pytest.raises(ValueError, match="Rust draw semantics"):
        T.DeviceChannel(T.Blake2sChannel(ts_compat=True))


@pytest.mark.parametrize("logs", [[10], [11, 9, 7]])
def test_fri_commit_device_transcript_equals_host_transcript(logs):
    """FriProver.commit with the device channel (no read-back per layer) and with the host channel produce the same roots,
    the same last layer and leave the channel in the same state; decommit + verify work on the device-committed prover."""
    cfg = T.FriConfig(2, 2, 6)
    cols, tw = [], None
    for i, lg in enumerate(logs):
        c, t = _secure_low_degree_eval(lg - 2, 2, 23000 + 10 * i)
        cols.append(c)
        tw = tw or t
    ch_d, ch_h = T.Blake2sChannel(), T.Blake2sChannel()
    pd = T.FriProver.commit(ch_d, cfg, cols, tw, device_channel=True)
    ph = T.FriProver.commit(ch_h, cfg, cols, tw, device_channel=False)
    assert pd.first_layer.merkle_tree.root() == ph.first_layer.merkle_tree.root()
    assert [l.merkle_tree.root() for l in pd.inner_layers] == [l.merkle_tree.root() for l in ph.inner_layers]
    assert [c.tup() for c in pd.last_layer_poly.coeffs] == [c.tup() for c in ph.last_layer_poly.coeffs]
    assert (ch_d.digest(), ch_d.n_challenges, ch_d.n_sent) == (ch_h.digest(), ch_h.n_challenges, ch_h.n_sent)
    proof, positions = pd.decommit(ch_d)
    _fri_verify(cfg, proof, [lg - 2 for lg in logs], _query_evals(cols, positions), positions)


_SHARDED_FRI_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["TSTWO_ROOT"])
import numpy as np
import torch.distributed as dist
import tstwo_amd as T
from tstwo_amd.fri_sharded import fri_commit_row_sharded
from tstwo_amd.distributed import shard_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
T._lib.init(0)                                   # both ranks share the one GPU of the test box
LOGD, BLOW = 9, 2
n = LOGD + BLOW
domain = T.CanonicCoset(n).circleDomain()
tw = T.precompute_twiddles(domain.halfCoset)
polys = [T.HipCirclePoly(np.random.default_rng(31000 + k).integers(0, T.P, size=1 << LOGD, dtype=np.uint32)) for k in range(4)]
evs = T.evaluate_polynomials(polys, domain, tw)
full = [e.values.to_numpy() for e in evs]
s, c = shard_rows(1 << n, world, rank)
shard = T.SecureColumnByCoords.from_numpy([f[s:s + c] for f in full])
cfg = T.FriConfig(0, BLOW, 5)          # last layer of 4 rows: the tail of the commit must switch to replicated
ch = T.Blake2sChannel()
layers, last = fri_commit_row_sharded(ch, cfg, shard, n, rank, world, tw)
# single-GPU reference on the whole column
ch1 = T.Blake2sChannel()
col = T.SecureEvaluation(domain, T.SecureColumnByCoords.from_numpy(full))
p1 = T.FriProver.commit(ch1, cfg, [col], tw)
want_roots = [p1.first_layer.merkle_tree.root()] + [l.merkle_tree.root() for l in p1.inner_layers]
assert [l.root for l in layers] == want_roots, (rank, len(layers), len(want_roots))
assert [x.tup() for x in last.coeffs] == [x.tup() for x in p1.last_layer_poly.coeffs]
assert ch.digest() == ch1.digest()
assert (world < 4 or any(l.replicated for l in layers)) and not layers[0].replicated and len(layers[0].subtree_roots) == world
# this rank's rows of every sharded layer equal the corresponding rows of the single-GPU layer
for l, ref in zip(layers[1:], p1.inner_layers):
    if not l.replicated:
        rs, rc = shard_rows(1 << l.log_size, world, rank)
        for a, b in zip(l.shard.to_numpy(), ref.evaluation.values.to_numpy()):
            assert (a == b[rs:rs + rc]).all()
# decommitment of the row-sharded first-layer tree == the single-GPU decommitment
from tstwo_amd.distributed import decommit_rows_sharded
queries = {n: sorted(set(int(q) for q in np.random.default_rng(7).integers(0, 1 << n, size=9)))}
qv, dec = decommit_rows_sharded(layers[0].subtree, layers[0].subtree_roots, shard.columns, [n] * 4, queries, rank, world)
qv1, dec1 = p1.first_layer.merkle_tree.decommit(queries, col.values.columns)
assert [v.value for v in qv] == [v.value for v in qv1]
assert dec.hashWitness == dec1.hashWitness and [v.value for v in dec.columnWitness] == [v.value for v in dec1.columnWitness]
T.MerkleVerifier(T.Blake2sMerkleHasher, layers[0].root, [n] * 4).verify(queries, qv, dec)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 4])
def test_fri_commit_row_sharded_matches_single_gpu(tmp_path, world):
    """SURVEY 8(e) / north_star: FRI layers row-sharded across ranks (here `world` processes sharing the box's one GPU, gloo
    for the root all-gather) give the single-GPU transcript: same roots, same last layer, same channel state."""
    import subprocess
    import sys
    script = tmp_path / "worker_fri.py"
    script.write_text(_SHARDED_FRI_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSTWO_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29540 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"rank {r} ok" in o


def test_bit_reverse_back_and_coset_sub_evaluation():
    """evaluation.ts:122-136 (bitReverse / bitReverseBack are inverse permutations) and :182-196 (CosetSubEvaluation,
    test/poly/cosetSubEvaluation.test.ts: wrapping strided view)."""
    n = 6
    vals = rand_column(33000, 1 << n)
    ev = T.HipCircleEvaluation(T.CanonicCoset(n).circleDomain(), vals)
    br = ev.bitReverse()
    assert (br.values.to_numpy() == vals[[T.bit_reverse_index(i, n) for i in range(1 << n)]]).all()
    assert (br.bitReverseBack().values.to_numpy() == vals).all() and (ev.deref().to_numpy() == vals).all()
    sub = ev.coset_sub_evaluation(5, 3)
    assert sub.at(0).value == vals[5] and sub.get(30).value == vals[(5 + 90) & 63]
    assert [m.value for m in sub.gather(range(40))] == [int(vals[(5 + 3 * i) & 63]) for i in range(40)]


def test_pcs_trees_of_different_heights():
    """Regression (found by tests/fuzz_parity.py): every tree is handed the query positions of all column sizes; a tree lower
    than the tallest one must ignore the sizes it does not have (pcs/prover.ts Rust text :137-141)."""
    config = T.PcsConfig(pow_bits=3, fri_config=T.FriConfig(1, 1, 5))
    logs = [[9, 7], [5, 6]]                       # the second tree is lower than the first
    scheme, ch = _pcs_setup(config, logs, seed=34000)
    point = T.CirclePoint.get_random_point(ch)
    pts = [[[point], [point]], [[point], [point]]]
    proof = scheme.prove_values(pts, ch)
    verifier, vch = _pcs_verifier(config, logs, proof.commitments)
    T.CirclePoint.get_random_point(vch)
    verifier.verify_values(pts, proof, vch)


def test_fri_commit_plan_graph_replay_equals_eager():
    """hipGraph capture of the device-transcript commit loop (FriCommitPlan): replaying the graph on new data written into the
    same input buffers gives the eager commit's roots, last layer and channel state — twice, with different data."""
    LOGD, BLOW = 8, 2
    cfg = T.FriConfig(1, BLOW, 5)
    col, tw = _secure_low_degree_eval(LOGD, BLOW, 35000)
    plan = T.FriCommitPlan(cfg, [col], tw)
    for seed in (35100, 35200):
        fresh, _ = _secure_low_degree_eval(LOGD, BLOW, seed)
        for dst, src in zip(col.values.columns, fresh.values.columns):          # new evaluations into the plan's input buffers
            L.call("tstwo_copy", C.c_void_p(dst.ptr), C.c_void_p(src.ptr), 4 * dst.len())
        ch_g, ch_e = T.Blake2sChannel(), T.Blake2sChannel()
        ch_g.mix_u64(seed); ch_e.mix_u64(seed)
        pg = plan.run(ch_g)
        pe = T.FriProver.commit(ch_e, cfg, [fresh], tw)
        assert pg.first_layer.merkle_tree.root() == pe.first_layer.merkle_tree.root()
        assert [l.merkle_tree.root() for l in pg.inner_layers] == [l.merkle_tree.root() for l in pe.inner_layers]
        assert [c.tup() for c in pg.last_layer_poly.coeffs] == [c.tup() for c in pe.last_layer_poly.coeffs]
        assert ch_g.digest() == ch_e.digest()
        proof, positions = pg.decommit(ch_g)
        _fri_verify_with_channel = T.FriVerifier.commit
        vch = T.Blake2sChannel(); vch.mix_u64(seed)
        v = T.FriVerifier.commit(vch, cfg, proof, [T.CirclePolyDegreeBound(LOGD)])
        assert v.sample_query_positions(vch) == positions
        v.decommit(_query_evals([col], positions))


def test_merkle_decommit_many_equals_individual_calls():
    """tstwo_merkle_decommit_many (one round trip for several trees) == tstwo_merkle_decommit per tree."""
    rng = np.random.default_rng(36000)
    reqs = []
    for t in range(5):
        logs = sorted([int(rng.integers(1, 11)) for _ in range(int(rng.integers(1, 6)))], reverse=True)
        cols = [T.HipColumn(rand_column(36000 + 10 * t + i, 1 << lg)) for i, lg in enumerate(logs)]
        tree = T.MerkleProver.commit(cols)
        queries = {lg: sorted(set(int(x) for x in rng.integers(0, 1 << lg, size=4))) for lg in set(logs)}
        queries[12] = [5]                              # a size this tree does not have: ignored
        reqs.append((tree, queries, cols))
    many = T.MerkleProver.decommit_many(reqs)
    for (tree, queries, cols), (qv, dec) in zip(reqs, many):
        qv1, dec1 = tree.decommit(queries, cols)
        assert [v.value for v in qv] == [v.value for v in qv1]
        assert dec.hashWitness == dec1.hashWitness and [v.value for v in dec.columnWitness] == [v.value for v in dec1.columnWitness]
        T.MerkleVerifier(T.Blake2sMerkleHasher, tree.root(), [c.len().bit_length() - 1 for c in cols]).verify(queries, qv, dec)
    assert T.MerkleProver.decommit_many([]) == []


def test_c_abi_collective_world_of_one():
    """include/tstwo_hip.h "multi-GPU": the RCCL communicator behind the C ABI.  This pool gives one GPU, so the world has
    one rank (RCCL initialises, the all-gather returns the own root); N > 1 over xGMI is covered by bench.py --gpus N
    only.  Also: without a communicator the gathers are device copies, and the async form orders against the stream."""
    import ctypes as C
    from tstwo_amd.distributed import HipComm
    cols = [T.HipColumn(rand_column(900 + c, 1 << 10)) for c in range(5)]
    tree = T.MerkleProver.commit(cols)
    root = tree.root()
    out = L.DeviceBuffer(32)
    L.call("tstwo_allgather_roots", C.c_void_p(tree.root_ptr()), C.c_void_p(out.ptr))         # no communicator: copy
    assert out.download(np.uint8, 32).tobytes() == root
    comm = HipComm(0, 1, lambda uid: uid)
    try:
        r, w = C.c_int(-1), C.c_int(-1)
        L.call("tstwo_comm_info", C.byref(r), C.byref(w))
        assert (r.value, w.value) == (0, 1)
        assert comm.allgather_roots(tree.root_ptr()) == [root]
        out2 = L.DeviceBuffer(32)
        out2.zero()
        L.call("tstwo_allgather_async", C.c_void_p(tree.root_ptr()), C.c_void_p(out2.ptr), 32)
        L.call("tstwo_comm_wait")
        assert out2.download(np.uint8, 32).tobytes() == root
        with pytest.raises(L.TstwoError, match="already initialised"):
            HipComm(0, 1, lambda uid: uid)
    finally:
        comm.close()


# ---------------------------------------------------------------- tstwo_fri_decommit: the whole FRI opening in one library call
@pytest.mark.parametrize("shape", [([8], 2, 5), ([10, 8], 2, 17), ([9, 7, 5], 1, 40), ([12], 3, 3)], ids=str)
def test_fri_decommit_in_library_equals_host_walk(shape):
    """FriProver.decommit_on_queries (ONE tstwo_fri_decommit call: fri.ts:346-384 position logic + witness evaluations + every
    tree's Merkle decommitment + the roots) returns byte for byte what the round-2 path returns (positions planned in Python,
    a gather and tstwo_merkle_decommit_many) — and the verifier accepts it."""
    log_degs, blow, n_queries = shape
    cfg = T.FriConfig(1, blow, n_queries)
    cols, tw = [], None
    for i, ld in enumerate(log_degs):
        c, t = _secure_low_degree_eval(ld, blow, 7000 + 13 * i)
        cols.append(c)
        tw = tw or t
    prover = T.FriProver.commit(T.Blake2sChannel(), cfg, cols, tw)
    rng = np.random.default_rng(sum(log_degs) + n_queries)
    max_log = max(log_degs) + blow
    for trial in range(3):
        pos = sorted(set(int(x) for x in rng.integers(0, 1 << max_log, size=n_queries)))
        if trial == 2:
            pos = [0, 1, (1 << max_log) - 1]                       # coset mates and the last row
        queries = T.Queries(pos, max_log)
        a, b = prover.decommit_on_queries(queries), prover.decommit_on_queries_host_walk(queries)
        for la, lb in zip([a.first_layer] + a.inner_layers, [b.first_layer] + b.inner_layers):
            assert [w.tup() for w in la.fri_witness] == [w.tup() for w in lb.fri_witness]
            assert la.decommitment.hashWitness == lb.decommitment.hashWitness
            assert [v.value for v in la.decommitment.columnWitness] == [v.value for v in lb.decommitment.columnWitness]
            assert la.commitment == lb.commitment
        assert len(a.inner_layers) == len(b.inner_layers)


def test_fri_decommit_capi_errors():
    col, tw = _secure_low_degree_eval(6, 2, 7100)
    prover = T.FriProver.commit(T.Blake2sChannel(), T.FriConfig(1, 2, 4), [col], tw)
    with pytest.raises(L.TstwoError, match="ascending and distinct"):
        prover.decommit_on_queries(_RawQueries([5, 3], 8))
    with pytest.raises(L.TstwoError, match="outside the domain"):
        prover.decommit_on_queries(_RawQueries([1 << 8], 8))
    assert prover.decommit_on_queries(T.Queries([], 8)).first_layer.fri_witness == []


class _RawQueries:
    """positions handed to the library unchecked (T.Queries validates them itself)"""

    def __init__(self, positions, log_domain_size):
        self.positions, self.log_domain_size = positions, log_domain_size


def _fri_commit_digest(prover, ch):
    """Everything a commit produced, hashed: channel state, every tree's root, every layer's evaluation, the last-layer polynomial."""
    import hashlib
    h = hashlib.blake2s()
    h.update(ch.digest())
    h.update(prover.first_layer.merkle_tree.root())
    for la in prover.inner_layers:
        h.update(la.merkle_tree.root())
        for c in la.evaluation.values.to_numpy():
            h.update(c.tobytes())
    for c in prover.last_layer_poly.coeffs:
        h.update(np.array(c.tup(), dtype="<u4").tobytes())
    return h.hexdigest()


_FRI_BIG_CASES = {"single19": ([17], 2, (3, 2, 12)), "mixed19": ([17, 15, 12], 2, (4, 2, 10)), "single18b1": ([17], 1, (2, 1, 8))}
_FRI_BIG_SCRIPT = r"""
import sys
sys.path[:0] = [{root!r}, {tests!r}]
import tstwo_amd as T
from tstwo_amd import _lib as L
from test_gpu_backend import _FRI_BIG_CASES, _fri_commit_digest, _secure_low_degree_eval
for name, (log_degs, blow, cfg) in _FRI_BIG_CASES.items():
    cols = [_secure_low_degree_eval(ld, blow, 8300 + ld)[0] for ld in log_degs]
    tw = _secure_low_degree_eval(log_degs[0], blow, 8300 + log_degs[0])[1]
    ch = T.Blake2sChannel()
    print(L.version().replace(" ", "_"), name, _fri_commit_digest(T.FriProver.commit(ch, T.FriConfig(*cfg), cols, tw), ch))
"""


def test_fri_commit_layers_big_layers_match_host_loop_and_verify(monkeypatch):
    """The prover-critical branches of tstwo_fri_commit_layers above 2^16 rows — fold fused into k_merkle_leaf4<true> through
    commit_layer (grid-strided, deferred digest stores), k_fold_circle2 with a capped grid, commit_upper_levels starting with
    1024-lane workgroups, k_channel_mix_draw when the hook is left set — ran only in timing tools.  Circle log 18-19, one column
    and mixed sizes (a column entering at a line layer of 2^16 and one at 2^13 rows): the library loop against round 2's
    per-layer host loop (every root, every evaluation, last-layer coefficients, channel state), then decommit + verify; and the
    same commits in the experiments build with the fusion / the single-launch tail switched off (the unfused and spare-tree
    branches), which must give the same bytes."""
    import os
    import subprocess
    import sys
    want = {}
    for name, (log_degs, blow, cfgt) in _FRI_BIG_CASES.items():
        cfg = T.FriConfig(*cfgt)
        cols = [_secure_low_degree_eval(ld, blow, 8300 + ld)[0] for ld in log_degs]
        tw = _secure_low_degree_eval(log_degs[0], blow, 8300 + log_degs[0])[1]
        ch_a, ch_b = T.Blake2sChannel(), T.Blake2sChannel()
        a = T.FriProver.commit(ch_a, cfg, cols, tw)
        monkeypatch.setenv("TSTWO_FRI_COMMIT_HOST_LOOP", "1")
        b = T.FriProver.commit(ch_b, cfg, cols, tw)
        monkeypatch.delenv("TSTWO_FRI_COMMIT_HOST_LOOP")
        assert len(a.inner_layers) == len(b.inner_layers) == (log_degs[0] + blow - 1) - (cfgt[0] + blow)
        assert a.first_layer.merkle_tree.root() == b.first_layer.merkle_tree.root()
        for la, lb in zip(a.inner_layers, b.inner_layers):
            assert la.merkle_tree.root() == lb.merkle_tree.root()
            for ca, cb in zip(la.evaluation.values.to_numpy(), lb.evaluation.values.to_numpy()):
                assert (ca == cb).all()
        assert [c.tup() for c in a.last_layer_poly.coeffs] == [c.tup() for c in b.last_layer_poly.coeffs]
        assert ch_a.digest() == ch_b.digest()
        want[name] = _fri_commit_digest(a, ch_a)
        assert want[name] == _fri_commit_digest(b, ch_b)
        proof, positions = a.decommit(ch_a)
        _fri_verify(cfg, proof, log_degs, _query_evals(cols, positions), positions)
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    script = _FRI_BIG_SCRIPT.format(root=os.path.dirname(tests_dir), tests=tests_dir)
    for knobs in ({"TSTWO_FRI_NO_FOLD_FUSION": "1"}, {"TSTWO_FRI_NO_TAIL": "1"}, {"TSTWO_FRI_NO_FOLD_FUSION": "1", "TSTWO_FRI_NO_TAIL": "1"}):
        out = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, TSTWO_HIP_LIB=L.LIB_EXP_PATH, **knobs), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        got = {}
        for line in out.stdout.strip().splitlines()[-len(want):]:
            ver, name, digest = line.split()
            assert "experiments" in ver
            got[name] = digest
        assert got == want, knobs


def test_fri_commit_layers_capi_matches_host_loop_and_reports_errors(monkeypatch):
    """tstwo_fri_commit_layers (the whole commit loop in one call) against the per-layer calls of round 2 (same device
    transcript): identical trees, evaluations, last-layer polynomial and channel state, for one and for mixed-size columns; bad
    arguments come back as errors and leak nothing."""
    for log_degs in ([9], [10, 8, 7]):
        cfg = T.FriConfig(2, 2, 5)
        cols = [_secure_low_degree_eval(ld, 2, 8100 + ld)[0] for ld in log_degs]
        tw = _secure_low_degree_eval(log_degs[0], 2, 8100 + log_degs[0])[1]
        ch_a, ch_b = T.Blake2sChannel(), T.Blake2sChannel()
        a = T.FriProver.commit(ch_a, cfg, cols, tw)
        monkeypatch.setenv("TSTWO_FRI_COMMIT_HOST_LOOP", "1")
        b = T.FriProver.commit(ch_b, cfg, cols, tw)
        monkeypatch.delenv("TSTWO_FRI_COMMIT_HOST_LOOP")
        assert ch_a.digest() == ch_b.digest()
        assert a.first_layer.merkle_tree.root() == b.first_layer.merkle_tree.root()
        assert len(a.inner_layers) == len(b.inner_layers) > 0
        for la, lb in zip(a.inner_layers, b.inner_layers):
            assert la.merkle_tree.root() == lb.merkle_tree.root()
            assert la.evaluation.domain().logSize() == lb.evaluation.domain().logSize()
            for ca, cb in zip(la.evaluation.values.to_numpy(), lb.evaluation.values.to_numpy()):
                assert (ca == cb).all()
        assert [c.tup() for c in a.last_layer_poly.coeffs] == [c.tup() for c in b.last_layer_poly.coeffs]
    # errors through the C ABI
    col, tw = _secure_low_degree_eval(6, 2, 8200)
    ptrs4 = L.ptr_array([c.ptr for c in col.values.columns] * 2)
    chan, alphas = L.DeviceBuffer(64), L.DeviceBuffer(16 * 16)
    outs, n_out, first = (L.FriLayerOut * 16)(), C.c_size_t(0), L.vp()
    args = lambda logs, n, last, cap: (ptrs4, L.u32x(logs), n, C.c_void_p(tw.itwiddles.ptr), tw.log_size, last, C.c_void_p(chan.ptr),
                                       C.c_void_p(alphas.ptr), 16, C.byref(first), outs, cap, C.byref(n_out))
    with pytest.raises(L.TstwoError, match="column sizes not decreasing"):
        L.call("tstwo_fri_commit_layers", *args([8, 8], 2, 3, 16))
    with pytest.raises(L.TstwoError, match="capacity too small"):
        L.call("tstwo_fri_commit_layers", *args([8], 1, 3, 2))
    with pytest.raises(L.TstwoError, match="no columns"):
        L.call("tstwo_fri_commit_layers", *args([8], 0, 3, 16))
    assert n_out.value == 0 and not first.value
