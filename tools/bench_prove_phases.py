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
def once():
    t = {}
    ch = T.Blake2sChannel(); cfg.mix_into(ch)
    scheme = T.CommitmentSchemeProver(cfg, tw)
    L.sync(); t0 = time.perf_counter()
    scheme.commit(polys, ch); L.sync(); t['commit'] = time.perf_counter() - t0
    pt = T.CirclePoint.get_random_point(ch)
    t0 = time.perf_counter()
    vals = T.HipCirclePoly.eval_at_point_batch(polys, pt); t['eval_at_point_batch'] = time.perf_counter() - t0
    ch.mix_felts(vals)
    t0 = time.perf_counter()
    samples = [[PointSample(pt, v)] for v in vals]
    q = compute_fri_quotients(scheme.trees[0].evaluations, samples, ch.draw_felt(), BLOW); L.sync(); t['quotients'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    fp = T.FriProver.commit(ch, cfg.fri_config, q, tw); L.sync(); t['fri_commit'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    nonce = T.grind(ch, cfg.pow_bits); ch.mix_u64(nonce); t['grind'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    proof, pos = fp.decommit(ch); t['fri_decommit'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = [tr.decommit(pos) for tr in scheme.trees]; t['tree_decommit'] = time.perf_counter() - t0
    return t
once(); once()
acc = {}
for _ in range(5):
    for k, v in once().items():
        acc[k] = acc.get(k, 0) + v / 5
print({k: round(v * 1e3, 3) for k, v in acc.items()}, "total ms", round(sum(acc.values()) * 1e3, 2))
