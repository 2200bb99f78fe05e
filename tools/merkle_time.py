#!/usr/bin/env python3
"""Times tstwo_merkle_commit of C columns x 2^n alone, back to back (HIP events on the library's stream); prints one line.
    python tools/merkle_time.py [--cols 32] [--log 22] [--reps 200] [--series 50]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=32)
ap.add_argument("--log", type=int, default=22)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--series", type=int, default=0)
ap.add_argument("--skew", type=int, default=0, help="place the columns in ONE buffer, column c at c * (4N + SKEW) bytes (channel-hotspot experiment)")
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
rng = np.random.default_rng(1)
bufs = []
if a.skew:
    big = L.DeviceBuffer(a.cols * (4 * N + a.skew))
    data = rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32)
    col_ptrs = [big.ptr + c * (4 * N + a.skew) for c in range(a.cols)]
    for cp in col_ptrs:
        L.call("tstwo_upload", C.c_void_p(cp), data.ctypes.data_as(C.c_void_p), 4 * N)
    ptrs = L.ptr_array(col_ptrs)
else:
    for _ in range(a.cols):
        b = L.DeviceBuffer(4 * N)
        b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
        bufs.append(b)
    ptrs = L.ptr_array([b.ptr for b in bufs])
print("column addresses mod 2^20:", [hex(p & 0xfffff) for p in (col_ptrs if a.skew else [b.ptr for b in bufs])[:4]], flush=True)
layers = L.DeviceBuffer(32 * ((2 << n) - 1))
logs = L.u32x([n] * a.cols)
for _ in range(3):
    L.call("tstwo_merkle_commit", ptrs, logs, a.cols, C.c_void_p(layers.ptr), None)
series, best = [], 1e9
for _ in range(a.reps):
    e0, e1 = L.Event(), L.Event()
    e0.record()
    L.call("tstwo_merkle_commit", ptrs, logs, a.cols, C.c_void_p(layers.ptr), None)
    e1.record()
    ms = e0.elapsed_ms(e1)
    series.append(ms)
    best = min(best, ms)
print(f"tstwo_merkle_commit {a.cols} x 2^{n}: avg {sum(series) / len(series) * 1e3:.1f} us  min {best * 1e3:.1f} us", flush=True)
if a.series:
    print("  series (us):", " ".join(f"{sum(series[i:i + a.series]) / len(series[i:i + a.series]) * 1e3:.0f}" for i in range(0, len(series), a.series)), flush=True)
