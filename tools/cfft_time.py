#!/usr/bin/env python3
"""Times tstwo_cfft_evaluate / interpolate of C columns x 2^n (HIP events on the library's stream); prints one line.
    python tools/cfft_time.py [--cols 32] [--log 22] [--reps 30] [--inv]      (TSTWO_HIP_LIB selects an experimental build)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402
from tstwo_amd.backend import HipBackend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=32)
ap.add_argument("--log", type=int, default=22)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--inv", action="store_true")
ap.add_argument("--split", type=int, default=1, help="transform the columns in SPLIT groups, one call per group (Infinity Cache reuse between the passes)")
ap.add_argument("--series", type=int, default=0, help="also print the mean of every SERIES consecutive repetitions (clock behaviour over time)")
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
half = HipBackend.canonic_half_coset_initial(n)
rng = np.random.default_rng(1)
bufs = []
for _ in range(a.cols):
    b = L.DeviceBuffer(4 * N)
    b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
    bufs.append(b)
tw, itw = L.DeviceBuffer(4 * (N // 2)), L.DeviceBuffer(4 * (N // 2))
L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(tw.ptr), C.c_void_p(itw.ptr))
ptrs = L.ptr_array([b.ptr for b in bufs])
gsz = a.cols // a.split
groups = [L.ptr_array([b.ptr for b in bufs[g * gsz:(g + 1) * gsz]]) for g in range(a.split)]
name = "tstwo_cfft_interpolate" if a.inv else "tstwo_cfft_evaluate"
t = itw if a.inv else tw
for _ in range(3):
    L.call(name, ptrs, a.cols, n, half, C.c_void_p(t.ptr), n - 1)
best, tot = 1e9, 0.0
series = []
for _ in range(a.reps):
    e0, e1 = L.Event(), L.Event()
    e0.record()
    if a.split == 1:
        L.call(name, ptrs, a.cols, n, half, C.c_void_p(t.ptr), n - 1)
    else:
        for gp in groups:
            L.call(name, gp, gsz, n, half, C.c_void_p(t.ptr), n - 1)
    e1.record()
    ms = e0.elapsed_ms(e1)
    best, tot = min(best, ms), tot + ms
    series.append(ms)
print(f"{os.environ.get('TSTWO_HIP_LIB', 'default')} {name} {a.cols} x 2^{n}: avg {tot / a.reps * 1e3:.1f} us  min {best * 1e3:.1f} us", flush=True)
if a.series:
    print("  series (us):", " ".join(f"{sum(series[i:i + a.series]) / len(series[i:i + a.series]) * 1e3:.0f}" for i in range(0, len(series), a.series)), flush=True)
