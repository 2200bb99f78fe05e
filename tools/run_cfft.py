#!/usr/bin/env python3
"""Runs only the CFFT (evaluate or interpolate) of `cols` columns x 2^log a few times — a clean target for rocprofv3."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log", type=int, default=22)
ap.add_argument("--cols", type=int, default=32)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--inverse", action="store_true")
ap.add_argument("--merkle", action="store_true")
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
rng = np.random.default_rng(0)
bufs = []
for c in range(a.cols):
    b = L.DeviceBuffer(4 * N)
    b.upload(rng.integers(0, L.P, size=N, dtype=np.uint32))
    bufs.append(b)
ptrs = L.ptr_array([b.ptr for b in bufs])
half = 1 << (31 - (n + 1))
tw, itw = L.DeviceBuffer(2 * N), L.DeviceBuffer(2 * N)
L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(tw.ptr), C.c_void_p(itw.ptr))
layers = L.DeviceBuffer(32 * ((2 << n) - 1)) if a.merkle else None
L.sync()
e0, e1 = L.Event(), L.Event()
for r in range(a.reps + 1):
    if r == 1:
        e0.record()
    if a.merkle:
        L.call("tstwo_merkle_commit", ptrs, L.u32x([n] * a.cols), a.cols, C.c_void_p(layers.ptr), None)
    elif a.inverse:
        L.call("tstwo_cfft_interpolate", ptrs, a.cols, n, half, C.c_void_p(itw.ptr), n - 1)
    else:
        L.call("tstwo_cfft_evaluate", ptrs, a.cols, n, half, C.c_void_p(tw.ptr), n - 1)
e1.record()
L.sync()
ms = e0.elapsed_ms(e1) / a.reps
print(f"log={n} cols={a.cols} {'merkle' if a.merkle else 'interpolate' if a.inverse else 'evaluate'}: {ms*1e3:.1f} us/call, "
      f"{a.cols*n*(N//2)/ms/1e9:.1f} Gbutterflies/s, algo {8.0*N*a.cols/ms/1e6:.0f} GB/s")
