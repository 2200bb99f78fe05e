import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tstwo_amd as T
from tstwo_amd import _lib as L
from tstwo_amd.quotients import marshal_quotient_args
from tstwo_amd.pcs import column_sample_batches, PointSample
L.init(0)
rng = np.random.default_rng(0)
n, NC = 22, 32
domain = T.CanonicCoset(n).circleDomain()
cols = [T.HipColumn(rng.integers(0, T.P, size=1 << n, dtype=np.uint32)) for _ in range(NC)]
pt = T.SECURE_FIELD_CIRCLE_GEN
NPTS = int(os.environ.get("NPTS", "1"))          # sample points per column (= batches)
pts = [pt]
for _ in range(NPTS - 1):
    pts.append(pts[-1].add(T.SECURE_FIELD_CIRCLE_GEN))
samples = [[PointSample(p_, T.QM31.from_u32_unchecked(*map(int, rng.integers(0, T.P, size=4)))) for p_ in pts] for _ in range(NC)]
coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
for _ in range(3):
    t0 = time.perf_counter(); batches = column_sample_batches(samples); t1 = time.perf_counter()
    vals, args = marshal_quotient_args(domain, cols, coeff, batches); t2 = time.perf_counter()
    out = T.SecureColumnByCoords.uninitialized(domain.size()); L.sync(); t3 = time.perf_counter()
    L.call("tstwo_quotients_accumulate", *args, out.ptrs()); L.sync(); t4 = time.perf_counter()
    print(f"batches {1e3*(t1-t0):.3f} ms, marshal {1e3*(t2-t1):.3f} ms, alloc {1e3*(t3-t2):.3f}, C call {1e3*(t4-t3):.3f} ms")
