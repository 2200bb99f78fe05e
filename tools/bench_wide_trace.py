#!/usr/bin/env python3
"""A wide mixed-size trace through TreeBuilder.extend_evals (interpolate) + commit (extend/evaluate + Merkle + mix_root)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tstwo_amd as T  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
rng = np.random.default_rng(0)
BLOW = 1
shape = [(18, 100), (16, 100), (14, 100)]
mx = max(lg for lg, _ in shape)
tw = T.precompute_twiddles(T.CanonicCoset(mx + BLOW).circleDomain().halfCoset)
evals = [T.HipCircleEvaluation(T.CanonicCoset(lg).circleDomain(), T.HipColumn(rng.integers(0, T.P, size=1 << lg, dtype=np.uint32)))
         for lg, cnt in shape for _ in range(cnt)]
words = sum((1 << lg) * cnt for lg, cnt in shape)


def once():
    scheme = T.CommitmentSchemeProver(T.PcsConfig(fri_config=T.FriConfig(0, BLOW, 3)), tw)
    ch = T.Blake2sChannel()
    L.sync(); t0 = time.perf_counter()
    tb = scheme.tree_builder()
    tb.extend_evals(evals)
    L.sync(); t1 = time.perf_counter()
    tb.commit(ch)
    L.sync(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1


once(); once()
a = np.mean([once() for _ in range(5)], axis=0)
print(f"{sum(c for _, c in shape)} columns {shape}, {words * 4 / 1e6:.0f} MB of trace: interpolate {a[0] * 1e3:.3f} ms, "
      f"extend+evaluate+Merkle+mix_root {a[1] * 1e3:.3f} ms")
