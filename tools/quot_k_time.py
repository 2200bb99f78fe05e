#!/usr/bin/env python3
"""Times accumulate_quotients of C columns x 2^n opened at k points each (k sample batches over ONE column list) through
tstwo_quotients_accumulate_samples_async (HIP events); prints one line per k.
    python tools/quot_k_time.py [--cols 32] [--log 22] [--kmax 5]      (TSTWO_HIP_LIB = experiments build + TSTWO_QUOT_NO_TRIPLE / NO_PAIR for A/B)"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402
from tstwo_amd.backend import HipBackend  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=32)
ap.add_argument("--log", type=int, default=22)
ap.add_argument("--kmax", type=int, default=5)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
rng = np.random.default_rng(3)
bufs = []
for _ in range(a.cols):
    b = L.DeviceBuffer(4 * N)
    b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
    bufs.append(b)
out = [L.DeviceBuffer(4 * N) for _ in range(4)]
half = HipBackend.canonic_half_coset_initial(n)
P = 2**31 - 1
import tstwo_amd as T
pt = T.SECURE_FIELD_CIRCLE_GEN
pts = [pt]
for _ in range(a.kmax):
    pts.append(pts[-1].add(T.SECURE_FIELD_CIRCLE_GEN))
for k in range(1, a.kmax + 1):
    off, cidx, points, values = [0], [], [], []
    for b in range(k):
        points += [*pts[b].x.tup(), *pts[b].y.tup()]
        for c in range(a.cols):
            cidx.append(c)
            values += [int(v) for v in rng.integers(0, P, size=4)]
        off.append(len(cidx))
    args = (half, n, L.ptr_array([b.ptr for b in bufs]), a.cols, k, L.u32x(off), L.u32x(cidx), L.u32x(points), L.u32x(values), L.u32x((5, 6, 7, 8)),
            L.ptr_array([o.ptr for o in out]))
    for _ in range(5):
        L.call("tstwo_quotients_accumulate_samples_async", *args)
    for _ in range(int(60.0 / 0.2)):          # clocks
        L.call("tstwo_quotients_accumulate_samples_async", *args)
    e0, e1 = L.Event(), L.Event()
    e0.record()
    for _ in range(a.reps):
        L.call("tstwo_quotients_accumulate_samples_async", *args)
    e1.record()
    ms = e0.elapsed_ms(e1) / a.reps
    byt = (4.0 * a.cols + 16.0) * N
    print(f"{os.environ.get('TSTWO_HIP_LIB', 'shipped').split('/')[-1]} NO_TRIPLE={os.environ.get('TSTWO_QUOT_NO_TRIPLE')} NO_PAIR={os.environ.get('TSTWO_QUOT_NO_PAIR')} "
          f"{a.cols} cols x 2^{n}, k = {k}: {ms * 1e3:.1f} us  ({byt / ms / 1e6:.0f} GB/s of one column sweep)", flush=True)
    L.call("tstwo_check_zero_flag")
