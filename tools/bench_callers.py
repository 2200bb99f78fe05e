#!/usr/bin/env python3
"""Times the callers on top of the hot path (SURVEY §8f rows built in this repo) on one GPU:
FriProver.commit (config-4-like: log 24 secure column), CommitmentTreeProver (32 columns, log 20 -> 22), eval_at_point,
decompose, grind.  Wall-clock around each call (they end with a host read-back of a root / value)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tstwo_amd as T  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
rng = np.random.default_rng(0)


def wall(fn, reps=3):
    fn()
    L.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    L.sync()
    return (time.perf_counter() - t0) / reps, r


def secure_low_degree(log_deg, blow):
    domain = T.CanonicCoset(log_deg + blow).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << log_deg, dtype=np.uint32))) for _ in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    return T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs])), tw


out = []
col, tw = secure_low_degree(22, 2)
cfg = T.FriConfig(5, 2, 20)
dt, prover = wall(lambda: T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw))
out.append({"caller": "FriProver.commit log 24 (fold_circle_into_line + 17 x (Merkle commit, fold_line) + last layer)", "ms": dt * 1e3,
            "inner_layers": len(prover.inner_layers)})

LOG, BLOW, NC = 20, 2, 32
tw2 = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW).circleDomain().halfCoset)
polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << LOG, dtype=np.uint32))) for _ in range(NC)]
dt, tree = wall(lambda: T.CommitmentTreeProver.new(polys, BLOW, T.Blake2sChannel(), tw2))
out.append({"caller": f"CommitmentTreeProver.new {NC} polys log {LOG} -> evaluate log {LOG + BLOW} + Merkle + mix_root (incl. extend/alloc)", "ms": dt * 1e3})

p22 = T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << 22, dtype=np.uint32)))
dt, _ = wall(lambda: p22.evalAtPoint(T.SECURE_FIELD_CIRCLE_GEN), reps=5)
out.append({"caller": "eval_at_point log 22", "ms": dt * 1e3})
dt, _ = wall(lambda: T.decompose(col), reps=5)
out.append({"caller": "decompose log 24", "ms": dt * 1e3})
ch = T.Blake2sChannel(); ch.mix_u64(7)
dt, nonce = wall(lambda: T.grind(ch, 26), reps=2)
out.append({"caller": "grind pow_bits 26", "ms": dt * 1e3, "nonce": nonce, "hashes_per_s": (nonce + 1) / dt})
for o in out:
    print(json.dumps(o), flush=True)
