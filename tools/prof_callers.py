#!/usr/bin/env python3
"""cProfile of the host side of FriProver.commit (log 24) and CommitmentSchemeProver.prove_values: where the time between
kernel launches goes."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tstwo_amd as T  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
rng = np.random.default_rng(0)
which = sys.argv[1] if len(sys.argv) > 1 else "fri"
if which == "fri":
    LOGD, BLOW = 22, 2
    domain = T.CanonicCoset(LOGD + BLOW).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << LOGD, dtype=np.uint32))) for _ in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    col = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs]))
    cfg = T.FriConfig(5, BLOW, 20)
    fn = lambda: T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw)
else:
    LOG, BLOW, NC = 20, 2, 32
    tw = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW).circleDomain().halfCoset)
    polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << LOG, dtype=np.uint32))) for _ in range(NC)]
    cfg = T.PcsConfig(pow_bits=20, fri_config=T.FriConfig(5, BLOW, 40))

    def fn():
        ch = T.Blake2sChannel()
        scheme = T.CommitmentSchemeProver(cfg, tw)
        scheme.commit(polys, ch)
        pt = T.CirclePoint.get_random_point(ch)
        return scheme.prove_values([[[pt]] * NC], ch)
fn(); fn()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    fn()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
