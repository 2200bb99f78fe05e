#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep.  tests/test_gpu_fuzz.py runs a fixed-seed slice of it in the -m gpu suite; longer sweeps:
python tests/fuzz_parity.py --seconds 120 [--big] on a GPU box.
Random shapes for the CFFT entry points (in place, out of place, fused extension, many columns), Merkle trees of mixed sizes,
folds, batch inverses, bit reversal and quotients; every result must equal the CPU oracle bit for bit."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as orc  # noqa: E402  (test infrastructure)
from tstwo_amd import _lib as L  # noqa: E402
import tstwo_amd as T  # noqa: E402
from gpu_util import dev, host, p4, ptrs, vp  # noqa: E402

rng = np.random.default_rng(0)         # re-seeded by run()
P = L.P
half_odds = lambda lg: 1 << (31 - (lg + 2))
_tw_cache = {}


def twiddles(n):
    if n not in _tw_cache:
        tw, itw = L.DeviceBuffer(max(4 << max(n - 1, 0), 16)), L.DeviceBuffer(max(4 << max(n - 1, 0), 16))
        L.call("tstwo_twiddles_build", half_odds(n - 1), n - 1, vp(tw), vp(itw))
        _tw_cache[n] = (tw, itw, *orc.precompute_twiddles(half_odds(n - 1), n - 1))
    return _tw_cache[n]


def rcol(n):
    return rng.integers(0, P, size=n, dtype=np.uint32)


def case_cfft():
    n = int(rng.integers(1, 17))
    n_cols = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 65, 100]))
    if n >= 14:
        n_cols = min(n_cols, 8)
    tw, itw, otw, oitw = twiddles(n)
    cols = [rcol(1 << n) for _ in range(n_cols)]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    pick = rng.choice(n_cols, size=min(3, n_cols), replace=False)
    ev = {}
    for c in pick:
        ev[c] = host(d[c], 1 << n)
        assert (ev[c] == orc.cfft_evaluate(cols[c], n, half_odds(n - 1), otw, n - 1)).all(), ("evaluate", n, n_cols, c)
    out = [L.DeviceBuffer(4 << n) for _ in cols]
    L.call("tstwo_cfft_interpolate_to", ptrs(d), ptrs(out), n_cols, n, half_odds(n - 1), vp(itw), n - 1)
    for c in range(n_cols):
        assert (host(out[c], 1 << n) == cols[c]).all(), ("interpolate_to", n, n_cols, c)
    L.call("tstwo_cfft_interpolate", ptrs(d), n_cols, n, half_odds(n - 1), vp(itw), n - 1)
    for c in pick:
        assert (host(d[c], 1 << n) == cols[c]).all(), ("interpolate", n, n_cols, c)
    return f"cfft n={n} cols={n_cols}"


def case_extended():
    n = int(rng.integers(3, 17))
    ext = int(rng.integers(0, 4))
    n_poly = max(n - ext, 1)
    n_cols = int(rng.choice([1, 3, 9, 70]))
    if n >= 14:
        n_cols = min(n_cols, 4)
    tw, _, otw, _ = twiddles(n)
    polys = [rcol(1 << n_poly) for _ in range(n_cols)]
    src = [dev(p) for p in polys]
    out = [L.DeviceBuffer(4 << n) for _ in polys]
    L.call("tstwo_cfft_evaluate_extended", ptrs(src), n_poly, ptrs(out), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    c = int(rng.integers(0, n_cols))
    e = np.concatenate([polys[c], np.zeros((1 << n) - (1 << n_poly), dtype=np.uint32)])
    assert (host(out[c], 1 << n) == orc.cfft_evaluate(e, n, half_odds(n - 1), otw, n - 1)).all(), ("extended", n_poly, n, n_cols)
    return f"extended {n_poly}->{n} cols={n_cols}"


def case_merkle():
    k = int(rng.integers(1, 5))
    logs = []
    for _ in range(k):
        logs += [int(rng.integers(0, 13))] * int(rng.choice([1, 2, 4, 7, 16, 20, 33]))
    cols = [rcol(1 << lg) for lg in logs]
    d = [dev(c) for c in cols]
    mx = max(logs)
    layers = L.DeviceBuffer(32 * ((2 << mx) - 1))
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", ptrs(d), L.u32x(logs), len(logs), vp(layers), root)
    olayers, oroot = orc.merkle_commit(cols, logs)
    assert bytes(root) == oroot, ("merkle root", logs)
    got = layers.download(np.uint8).reshape(-1, 32)
    for lg in range(mx + 1):
        assert (got[(1 << lg) - 1:(2 << lg) - 1] == olayers[lg]).all(), ("merkle layer", lg, logs)
    return f"merkle {len(logs)} cols, max log {mx}"


def case_fold():
    n = int(rng.integers(3, 15))
    src = [rcol(1 << n) for _ in range(4)]
    dst = [rcol(1 << (n - 1)) for _ in range(4)]
    alpha = tuple(int(x) for x in rng.integers(0, P, size=4))
    _, itw, _, _ = twiddles(n)
    ds, dd = [dev(c) for c in src], [dev(c) for c in dst]
    L.call("tstwo_fri_fold_circle_into_line", p4(dd), 1 << (n - 1), p4(ds), n, vp(itw), n - 1, L.u32x(alpha))
    exp = orc.fold_circle_into_line(dst, src, n, half_odds(n - 1), alpha)
    for a, b in zip(dd, exp):
        assert (host(a, 1 << (n - 1)) == b).all(), ("fold_circle", n)
    out = [L.DeviceBuffer(max(4 << (n - 2), 16)) for _ in range(4)]
    L.call("tstwo_fri_fold_line", p4(dd), n - 1, vp(itw), n - 1, L.u32x(alpha), p4(out))
    exp2 = orc.fold_line(exp, n - 1, half_odds(n - 1), alpha)
    for a, b in zip(out, exp2):
        assert (host(a, 1 << (n - 2)) == b).all(), ("fold_line", n)
    return f"folds n={n}"


def case_fields():
    n = int(rng.integers(1, 100000))
    a = rng.integers(1, P, size=n, dtype=np.uint32)
    da, do = dev(a), L.DeviceBuffer(4 * n + 16)
    L.call("tstwo_m31_batch_inverse", vp(da), vp(do), n)
    assert (host(do, n) == orc.m31_batch_inverse(a)).all(), ("m31 inverse", n)
    lg = int(rng.integers(1, 17))
    b = rcol(1 << lg)
    db = dev(b)
    L.call("tstwo_bit_reverse", ptrs([db]), 1, 1 << lg)
    assert (host(db, 1 << lg) == orc.bit_reverse(b)).all(), ("bit reverse", lg)
    return f"fields n={n} bitrev log={lg}"


def _low_degree_secure(log_deg, blow):
    domain = T.CanonicCoset(log_deg + blow).circleDomain()
    polys = [T.HipCirclePoly(rcol(1 << log_deg)) for _ in range(4)]
    return domain, polys


def case_fri():
    """commit (device transcript) == commit (host transcript); decommit; host verifier accepts."""
    blow = int(rng.integers(1, 4))
    n_cols = int(rng.integers(1, 4))
    degs = sorted(rng.choice(np.arange(3, 12), size=n_cols, replace=False).tolist(), reverse=True)
    last = int(rng.integers(0, min(degs) - 1)) if min(degs) > 1 else 0
    cfg = T.FriConfig(last, blow, int(rng.integers(1, 12)))
    tw = T.precompute_twiddles(T.CanonicCoset(degs[0] + blow).circleDomain().halfCoset)
    cols = []
    for dg in degs:
        domain, polys = _low_degree_secure(dg, blow)
        evs = T.evaluate_polynomials(polys, domain, tw)
        cols.append(T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs])))
    ch_d, ch_h = T.Blake2sChannel(), T.Blake2sChannel()
    pd = T.FriProver.commit(ch_d, cfg, cols, tw, device_channel=True)
    ph = T.FriProver.commit(ch_h, cfg, cols, tw, device_channel=False)
    assert ch_d.digest() == ch_h.digest(), ("fri transcript", degs, blow, last)
    assert [l.merkle_tree.root() for l in pd.inner_layers] == [l.merkle_tree.root() for l in ph.inner_layers]
    proof, positions = pd.decommit(ch_d)
    vch = T.Blake2sChannel()
    v = T.FriVerifier.commit(vch, cfg, proof, [T.CirclePolyDegreeBound(dg) for dg in degs])
    assert v.sample_query_positions(vch) == positions
    v.decommit([c.values.gather(positions[c.domain.logSize()]) for c in cols])
    return f"fri degs={degs} blow={blow} last={last}"


def case_pcs():
    blow = int(rng.integers(1, 3))
    config = T.PcsConfig(pow_bits=int(rng.integers(0, 10)), fri_config=T.FriConfig(int(rng.integers(0, 3)), blow, int(rng.integers(1, 8))))
    trees = [[int(rng.integers(4, 11)) for _ in range(int(rng.integers(1, 5)))] for _ in range(int(rng.integers(1, 3)))]
    mx = max(lg for t in trees for lg in t)
    tw = T.precompute_twiddles(T.CanonicCoset(mx + blow).circleDomain().halfCoset)
    scheme = T.CommitmentSchemeProver(config, tw)
    ch = T.Blake2sChannel()
    config.mix_into(ch)
    for logs in trees:
        tb = scheme.tree_builder()
        tb.extend_evals([T.HipCircleEvaluation(T.CanonicCoset(lg).circleDomain(), rcol(1 << lg)) for lg in logs])
        tb.commit(ch)
    pt = T.CirclePoint.get_random_point(ch)
    pts = [[[pt] for _ in logs] for logs in trees]
    proof = scheme.prove_values(pts, ch)
    ver = T.CommitmentSchemeVerifier(config)
    vch = T.Blake2sChannel()
    config.mix_into(vch)
    for logs, root in zip(trees, proof.commitments):
        ver.commit(root, logs, vch)
    vpt = T.CirclePoint.get_random_point(vch)
    ver.verify_values([[[vpt] for _ in logs] for logs in trees], proof, vch)
    return f"pcs trees={trees} blow={blow}"


def _rq():
    return tuple(int(x) for x in rng.integers(0, P, size=4))


def _rand_secure_point():
    """a random point of the QM31 circle: k * SECURE_FIELD_CIRCLE_GEN for a random k"""
    k = int(rng.integers(1, 1 << 30))
    return T.SECURE_FIELD_CIRCLE_GEN.mul(k, T.QM31.one(), T.QM31.zero())


def case_quotients():
    """accumulateQuotients through the host mirror (quotientConstants + kernel) == the oracle's per-row reference formulation."""
    n = int(rng.integers(1, 13))
    n_cols = int(rng.integers(1, 9))
    n_batches = int(rng.integers(1, 4))
    batches_o, batches_t = [], []
    shared = rng.choice(n_cols, size=int(rng.integers(1, n_cols + 1)), replace=False) if rng.integers(0, 3) == 0 else None
    if shared is not None and rng.integers(0, 2) == 0:
        n_batches = int(rng.integers(2, 10))          # round 4: 3+ batches over one list take the row-pair / multi-batch sweeps (n >= 9: k_quotients_rp)
        n = max(n, int(rng.integers(3, 13)))
    cols = [rcol(1 << n) for _ in range(n_cols)]
    for _ in range(n_batches):
        pt = _rand_secure_point()
        # one time in three every batch covers the same columns in the same order (each column opened at several points: with two
        # batches that is the shared-load kernel k_quotients8_multi<2>; three and more: k_quotients_rp)
        chosen = shared if shared is not None else rng.choice(n_cols, size=int(rng.integers(1, n_cols + 1)), replace=False)
        cv = [(int(c), _rq()) for c in chosen]
        batches_o.append((pt.x.tup(), pt.y.tup(), cv))
        batches_t.append(T.ColumnSampleBatch(pt, [(c, T.QM31.from_u32_unchecked(*v)) for c, v in cv]))
    coeff = _rq()
    domain = T.CanonicCoset(n).circleDomain()
    got = T.accumulateQuotients(domain, [T.HipColumn(c) for c in cols], T.QM31.from_u32_unchecked(*coeff), batches_t).values.to_numpy()
    exp = orc.accumulate_quotients(half_odds(n - 1), n, cols, coeff, batches_o)
    for k in range(4):
        assert (got[k] == exp[k]).all(), ("quotients", n, n_cols, n_batches)
    return f"quotients n={n} cols={n_cols} batches={n_batches}"


def case_eval_decommit_qm31():
    n = int(rng.integers(1, 15))
    n_cols = int(rng.choice([1, 2, 5, 70]))
    cols = [rcol(1 << n) for _ in range(n_cols)]
    pt = _rand_secure_point()
    got = T.HipCirclePoly.eval_at_point_batch([T.HipCirclePoly(c) for c in cols], pt)
    c = int(rng.integers(0, n_cols))
    assert got[c].tup() == tuple(orc.eval_at_point(cols[c], n, pt.x.tup(), pt.y.tup())), ("eval_at_point", n, n_cols)
    # Merkle decommit: in-library walk == host walk, verifier accepts
    logs = sorted([int(rng.integers(0, 11)) for _ in range(int(rng.integers(1, 7)))], reverse=True)
    hc = [T.HipColumn(rcol(1 << lg)) for lg in logs]
    tree = T.MerkleProver.commit(hc)
    queries = {lg: sorted(set(int(x) for x in rng.integers(0, 1 << lg, size=int(rng.integers(1, 6))))) for lg in set(logs) if rng.random() < 0.8}
    if not queries:
        queries = {logs[0]: [0]}
    v1, d1 = tree.decommit(queries, hc)
    v2, d2 = tree._decommit_walk(queries, hc)
    assert [v.value for v in v1] == [v.value for v in v2] and d1.hashWitness == d2.hashWitness
    assert [v.value for v in d1.columnWitness] == [v.value for v in d2.columnWitness]
    T.MerkleVerifier(T.Blake2sMerkleHasher, tree.root(), logs).verify(queries, v1, d1)
    # QM31 column mul / batch inverse vs oracle
    m = int(rng.integers(1, 5000))
    a4 = [rng.integers(1, P, size=m, dtype=np.uint32) for _ in range(4)]
    b4 = [rcol(m) for _ in range(4)]
    A, B = T.SecureColumnByCoords.from_numpy(a4), T.SecureColumnByCoords.from_numpy(b4)
    be = T.HipBackend()
    for g, e in zip(be.secureMul(A, B).to_numpy(), orc.qm31_col_mul(a4, b4)):
        assert (g == e).all(), ("qm31 mul", m)
    for g, e in zip(be.batchInverse(A).to_numpy(), orc.qm31_batch_inverse(a4)):
        assert (g == e).all(), ("qm31 inverse", m)
    return f"eval/decommit/qm31 n={n} cols={n_cols} tree={logs}"


def case_rows_sharded():
    """virtual ranks: row-sharded folds and subtree roots == whole-layer results"""
    from tstwo_amd import distributed as D
    n = int(rng.integers(6, 13))
    world = int(rng.choice([2, 4, 8]))
    if (1 << (n - 2)) // world < 4:
        world = 2
    domain = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    src_np, dst_np = [rcol(1 << n) for _ in range(4)], [rcol(1 << (n - 1)) for _ in range(4)]
    alpha = T.QM31.from_u32_unchecked(*_rq())
    src = T.SecureEvaluation(domain, T.SecureColumnByCoords.from_numpy(src_np))
    dst = T.LineEvaluation(T.LineDomain(domain.halfCoset), T.SecureColumnByCoords.from_numpy(dst_np))
    T.fold_circle_into_line(dst, src, alpha, tw)
    whole = dst.values.to_numpy()
    root = T.MerkleProver.commit(dst.values.columns).root()
    parts, subroots = [], []
    for rank in range(world):
        s_, c_ = D.shard_rows(1 << (n - 1), world, rank)
        d = T.SecureColumnByCoords.from_numpy([x[s_:s_ + c_] for x in dst_np])
        D.fold_circle_into_line_rows(d, T.SecureColumnByCoords.from_numpy([x[2 * s_:2 * (s_ + c_)] for x in src_np]), n, rank, world, alpha, tw)
        parts.append(d.to_numpy())
        subroots.append(T.MerkleProver.commit(d.columns).root())
    for k in range(4):
        assert (np.concatenate([p[k] for p in parts]) == whole[k]).all(), ("rows fold", n, world)
    assert D.combine_subtree_roots(subroots) == root, ("rows root", n, world)
    return f"rows n={n} world={world}"


def case_cfft_big():
    """tiled path: random log 13..22 and column counts; oracle on one column up to log 18, round trips and the fused
    extension against extend + evaluate (GPU vs GPU) beyond."""
    n = int(rng.integers(13, 23))
    max_cols = max(1, min(40, (1 << 25) >> n))
    n_cols = int(rng.integers(1, max_cols + 1))
    tw, itw, otw, oitw = twiddles(n)
    cols = [rcol(1 << n) for _ in range(n_cols)]
    d = [dev(c) for c in cols]
    L.call("tstwo_cfft_evaluate", ptrs(d), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    c = int(rng.integers(0, n_cols))
    if n <= 18:
        assert (host(d[c], 1 << n) == orc.cfft_evaluate(cols[c], n, half_odds(n - 1), otw, n - 1)).all(), ("big evaluate", n, n_cols, c)
    out = [L.DeviceBuffer(4 << n) for _ in cols]
    L.call("tstwo_cfft_interpolate_to", ptrs(d), ptrs(out), n_cols, n, half_odds(n - 1), vp(itw), n - 1)
    for k in range(n_cols):
        assert (host(out[k], 1 << n) == cols[k]).all(), ("big roundtrip", n, n_cols, k)
    ext = int(rng.integers(1, 3))
    polys = [dev(col[:1 << (n - ext)]) for col in cols]
    L.call("tstwo_cfft_evaluate_extended", ptrs(polys), n - ext, ptrs(out), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    ref = [L.DeviceBuffer(4 << n) for _ in cols]
    for p_, r in zip(polys, ref):
        L.call("tstwo_poly_extend", vp(p_), n - ext, vp(r), n)
    L.call("tstwo_cfft_evaluate", ptrs(ref), n_cols, n, half_odds(n - 1), vp(tw), n - 1)
    for k in range(n_cols):
        assert (host(out[k], 1 << n) == host(ref[k], 1 << n)).all(), ("big extended", n, ext, n_cols, k)
    return f"cfft_big n={n} cols={n_cols} ext={ext}"


def case_merkle_big():
    logs = []
    for _ in range(int(rng.integers(1, 4))):
        logs += [int(rng.integers(10, 19))] * int(rng.choice([1, 4, 16, 20, 32]))
    while sum(1 << lg for lg in logs) > (1 << 24):
        logs.pop()
    cols = [rcol(1 << lg) for lg in logs]
    d = [dev(c) for c in cols]
    mx = max(logs)
    layers = L.DeviceBuffer(32 * ((2 << mx) - 1))
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", ptrs(d), L.u32x(logs), len(logs), vp(layers), root)
    assert bytes(root) == orc.merkle_commit(cols, logs)[1], ("merkle big", logs)
    return f"merkle_big {len(logs)} cols max log {mx}"


N_GENERATORS, N_GENERATORS_BIG = 10, 2        # round-robin: that many cases = every generator ran once


def run(seconds=60.0, seed=0, big=False, max_cases=None, verbose=True):
    """Round-robin over the case generators until `seconds` have passed or `max_cases` ran; returns the number of cases."""
    global rng
    rng = np.random.default_rng(seed)
    _tw_cache.clear()
    L.init(0)
    cases = [case_cfft, case_extended, case_merkle, case_fold, case_fields, case_fri, case_pcs, case_quotients, case_eval_decommit_qm31,
             case_rows_sharded]
    assert len(cases) == N_GENERATORS
    if big:
        cases = [case_cfft_big, case_merkle_big]
        assert len(cases) == N_GENERATORS_BIG
    t0, done = time.time(), 0
    while time.time() - t0 < seconds and (max_cases is None or done < max_cases):
        msg = cases[done % len(cases)]()
        done += 1
        if verbose and done % (7 if big else 501) == 0:
            print(f"[{time.time() - t0:6.1f}s] {done} cases ok (last: {msg})", flush=True)
    if verbose:
        print(f"fuzz ok: {done} cases in {time.time() - t0:.1f}s, seed {seed}")
    return done


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", action="store_true", help="only the large-size cases (tiled CFFT path, 13 <= log <= 22)")
    args = ap.parse_args()
    run(args.seconds, args.seed, args.big)
