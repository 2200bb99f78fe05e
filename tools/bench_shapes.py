#!/usr/bin/env python3
"""Merkle commit and CFFT across trace shapes (wide / mixed-size), to spot launch- or latency-bound regimes."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
rng = np.random.default_rng(0)


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = L.Event(), L.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    L.sync()
    return e0.elapsed_ms(e1) / reps * 1e3


for shape in ([(16, 256)], [(14, 1024)], [(18, 100), (16, 100), (14, 100)], [(20, 20), (12, 500)], [(22, 32)]):
    bufs, logs = [], []
    for lg, cnt in shape:
        for _ in range(cnt):
            b = L.DeviceBuffer(4 << lg)
            b.upload(rng.integers(0, L.P, size=1 << lg, dtype=np.uint32))
            bufs.append(b)
            logs.append(lg)
    ptrs = L.ptr_array([b.ptr for b in bufs])
    mx = max(logs)
    layers = L.DeviceBuffer(32 * ((2 << mx) - 1))
    us = timed(lambda: L.call("tstwo_merkle_commit", ptrs, L.u32x(logs), len(logs), C.c_void_p(layers.ptr), None))
    words = sum(1 << lg for lg in logs)
    compress = sum((1 << lg) for lg in set(logs)) * 0 + sum(-(-sum(1 for l2 in logs if l2 == lg) // 16) * (1 << lg) for lg in set(logs)) + (1 << mx)
    print(json.dumps({"shape": shape, "merkle_us": round(us, 1), "column_words": words, "approx_compressions": compress,
                      "G_compress_per_s": round(compress / us / 1e3, 1)}))
    del bufs
    L.call("tstwo_trim")
