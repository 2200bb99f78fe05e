#!/usr/bin/env python3
"""Where FriProver.decommit and CommitmentTreeProver.decommit spend their time: library calls against the Python around them."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tstwo_amd as T
from tstwo_amd import _lib as L
from tstwo_amd.pcs import compute_fri_quotients, PointSample
L.init(0)
rng = np.random.default_rng(0)
LOG, BLOW, NC = 20, 2, 32
tw = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW).circleDomain().halfCoset)
polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << LOG, dtype=np.uint32))) for _ in range(NC)]
cfg = T.PcsConfig(pow_bits=20, fri_config=T.FriConfig(5, BLOW, 40))
calls = {}
orig = L.call


def timed_call(name, *a):
    t0 = time.perf_counter()
    try:
        return orig(name, *a)
    finally:
        calls[name] = calls.get(name, 0.0) + time.perf_counter() - t0


def once():
    ch = T.Blake2sChannel(); cfg.mix_into(ch)
    scheme = T.CommitmentSchemeProver(cfg, tw)
    scheme.commit(polys, ch)
    pt = T.CirclePoint.get_random_point(ch)
    vals = T.HipCirclePoly.eval_at_point_batch(polys, pt)
    ch.mix_felts(vals)
    samples = [[PointSample(pt, v)] for v in vals]
    q = compute_fri_quotients(scheme.trees[0].evaluations, samples, ch.draw_felt(), BLOW)
    fp = T.FriProver.commit(ch, cfg.fri_config, q, tw)
    nonce = T.grind(ch, cfg.pow_bits); ch.mix_u64(nonce)
    L.sync()
    calls.clear()
    L.call = timed_call
    t0 = time.perf_counter()
    proof, pos = fp.decommit(ch)
    t1 = time.perf_counter()
    lib_fri = dict(calls); calls.clear()
    r = [tr.decommit(pos) for tr in scheme.trees]
    t2 = time.perf_counter()
    lib_tree = dict(calls)
    L.call = orig
    return t1 - t0, lib_fri, t2 - t1, lib_tree


once(); once()
for _ in range(3):
    a, la, b, lb = once()
    print(f"fri decommit {a * 1e3:.3f} ms, library calls {({k: round(v * 1e3, 3) for k, v in la.items()})}")
    print(f"tree decommit {b * 1e3:.3f} ms, library calls {({k: round(v * 1e3, 3) for k, v in lb.items()})}")
if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile, pstats
    ch = T.Blake2sChannel(); cfg.mix_into(ch)
    scheme = T.CommitmentSchemeProver(cfg, tw)
    scheme.commit(polys, ch)
    pt = T.CirclePoint.get_random_point(ch)
    vals = T.HipCirclePoly.eval_at_point_batch(polys, pt)
    ch.mix_felts(vals)
    samples = [[PointSample(pt, v)] for v in vals]
    q = compute_fri_quotients(scheme.trees[0].evaluations, samples, ch.draw_felt(), BLOW)
    fp = T.FriProver.commit(ch, cfg.fri_config, q, tw)
    L.sync()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        c2 = ch.clone()
        proof, pos = fp.decommit(c2)
        r = [tr.decommit(pos) for tr in scheme.trees]
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
