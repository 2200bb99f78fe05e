#!/usr/bin/env python3
"""Target for `rocprofv3 --kernel-trace`: CommitmentSchemeProver commit + prove_values of 32 polynomials of log 20 (blowup 2),
repeated; tools/trace_fri_commit.py --timeline prints the kernel sequence of the last proof."""
import os, sys, time
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


def fn():
    ch = T.Blake2sChannel()
    scheme = T.CommitmentSchemeProver(cfg, tw)
    scheme.commit(polys, ch)
    L.sync()
    t0 = time.perf_counter()
    pt = T.CirclePoint.get_random_point(ch)
    scheme.prove_values([[[pt]] * NC], ch)
    return time.perf_counter() - t0


for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    fn()
print(f"prove_values: {min(fn() for _ in range(10)) * 1e3:.3f} ms (best of 10)", flush=True)
