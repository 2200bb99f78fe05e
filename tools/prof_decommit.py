import os, sys, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tstwo_amd as T
from tstwo_amd import _lib as L
L.init(0)
rng = np.random.default_rng(0)
LOGD, BLOW = 20, 2
domain = T.CanonicCoset(LOGD + BLOW).circleDomain()
tw = T.precompute_twiddles(domain.halfCoset)
polys = [T.HipCirclePoly(rng.integers(0, T.P, size=1 << LOGD, dtype=np.uint32)) for _ in range(4)]
evs = T.evaluate_polynomials(polys, domain, tw)
col = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs]))
cfg = T.FriConfig(5, BLOW, 40)
ch = T.Blake2sChannel()
fp = T.FriProver.commit(ch, cfg, [col], tw)
fp.decommit(ch.clone())
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    fp.decommit(ch.clone())
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
