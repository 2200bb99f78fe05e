#!/usr/bin/env python3
"""Times tstwo_merkle_commit_layer (the leaf layer alone) of C columns x 2^n; one line.  TSTWO_HIP_LIB selects the build.
    python tools/leaf_layer_time.py [--cols 4] [--log 24] [--reps 100]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=4)
ap.add_argument("--log", type=int, default=24)
ap.add_argument("--reps", type=int, default=100)
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
rng = np.random.default_rng(1)
bufs = []
for _ in range(a.cols):
    b = L.DeviceBuffer(4 * N)
    b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
    bufs.append(b)
ptrs = L.ptr_array([b.ptr for b in bufs])
out = L.DeviceBuffer(32 * N)
for _ in range(30):
    L.call("tstwo_merkle_commit_layer", n, None, ptrs, a.cols, C.c_void_p(out.ptr))
ts = []
for _ in range(a.reps):
    e0, e1 = L.Event(), L.Event()
    e0.record()
    L.call("tstwo_merkle_commit_layer", n, None, ptrs, a.cols, C.c_void_p(out.ptr))
    e1.record()
    ts.append(e0.elapsed_ms(e1))
tag = os.path.basename(os.environ.get("TSTWO_HIP_LIB", "default")) + " cap=" + os.environ.get("TSTWO_MERKLE_CAP", "-")
print(f"{tag}: leaf layer {a.cols} x 2^{n}: avg {sum(ts) / len(ts) * 1e3:.1f} us  min {min(ts) * 1e3:.1f} us  "
      f"({N / (sum(ts) / len(ts) * 1e-3) / 1e9 * max(1, -(-a.cols // 16)):.1f} G compressions/s)", flush=True)
