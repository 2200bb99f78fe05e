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


def wall(fn, reps=3, spin_s=0.15):
    """Steady-state wall time per call: the call itself runs for `spin_s` first (the part needs 40-75 ms of load before its
    clocks settle, DESIGN 4.1 "Clocks" — three calls of a 3 ms caller sit inside that ramp and read 20-25 % high), then at least
    `reps` calls and 30 ms are timed."""
    fn()
    L.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < spin_s:
        fn()
    L.sync()
    t0 = time.perf_counter()
    n = 0
    while n < reps or time.perf_counter() - t0 < 0.03:
        r = fn()
        n += 1
    L.sync()
    return (time.perf_counter() - t0) / n, r


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
# whole PCS opening proof: 32 polys log 20, blowup 2 (extended log 22), one OODS point per column, 40 queries, 20 PoW bits
pcs_cfg = T.PcsConfig(pow_bits=20, fri_config=T.FriConfig(5, BLOW, 40))


def pcs_prove():
    ch = T.Blake2sChannel()
    pcs_cfg.mix_into(ch)
    scheme = T.CommitmentSchemeProver(pcs_cfg, tw2)
    t0 = time.perf_counter()
    scheme.commit(polys, ch)
    t1 = time.perf_counter()
    pt = T.CirclePoint.get_random_point(ch)
    proof = scheme.prove_values([[[pt]] * NC], ch)
    t2 = time.perf_counter()
    return (t1 - t0, t2 - t1, proof, pt)


pcs_prove()
c_ms, p_ms, proof, pt = pcs_prove()
t0 = time.perf_counter()
vch = T.Blake2sChannel()
pcs_cfg.mix_into(vch)
ver = T.CommitmentSchemeVerifier(pcs_cfg)
ver.commit(proof.commitments[0], [LOG] * NC, vch)
vpt = T.CirclePoint.get_random_point(vch)
ver.verify_values([[[vpt]] * NC], proof, vch)
v_ms = (time.perf_counter() - t0) * 1e3
out.append({"caller": f"CommitmentSchemeProver: commit {NC} polys log {LOG} (blowup {BLOW}) then prove_values (eval_at_point x{NC}, quotients, "
                      "FRI commit, grind 20 bits, FRI + tree decommit, 40 queries)", "commit_ms": c_ms * 1e3, "prove_values_ms": p_ms * 1e3,
            "host_verify_ms": v_ms})
for o in out:
    print(json.dumps(o), flush=True)
