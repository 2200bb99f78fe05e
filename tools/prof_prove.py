#!/usr/bin/env python3
"""cProfile of CommitmentSchemeProver.prove_values (32 polys log 20, blowup 2, 40 queries): host time by function."""
import cProfile, pstats, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tstwo_amd as T
from tstwo_amd import _lib as L
L.init(0)
rng = np.random.default_rng(0)
LOG, BLOW, NC = 20, 2, 32
tw = T.precompute_twiddles(T.CanonicCoset(LOG + BLOW).circleDomain().halfCoset)
polys = [T.HipCirclePoly(T.HipColumn(rng.integers(0, T.P, size=1 << LOG, dtype=np.uint32))) for _ in range(NC)]
cfg = T.PcsConfig(pow_bits=20, fri_config=T.FriConfig(5, BLOW, 40))
prepared = []
for _ in range(24):
    ch = T.Blake2sChannel()
    scheme = T.CommitmentSchemeProver(cfg, tw)
    scheme.commit(polys, ch)
    prepared.append((scheme, ch, T.CirclePoint.get_random_point(ch)))
L.sync()
for scheme, ch, pt in prepared[:4]:
    scheme.prove_values([[[pt]] * NC], ch)
pr = cProfile.Profile()
pr.enable()
for scheme, ch, pt in prepared[4:]:
    scheme.prove_values([[[pt]] * NC], ch)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(30)
