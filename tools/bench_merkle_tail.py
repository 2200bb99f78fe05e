#!/usr/bin/env python3
"""A/B timing of the Merkle upper-tree schemes: TSTWO_MERKLE_UP_LOG / TSTWO_MERKLE_UP_ONELANE are read once per process,
so this script times one setting; run it once per setting."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
rng = np.random.default_rng(0)
res = {"UP_LOG": os.environ.get("TSTWO_MERKLE_UP_LOG"), "ONELANE": os.environ.get("TSTWO_MERKLE_UP_ONELANE")}
for n, cols in ((22, 32), (24, 4), (16, 4), (12, 4)):
    N = 1 << n
    bufs = []
    for c in range(cols):
        x = L.DeviceBuffer(4 * N)
        x.upload(rng.integers(0, L.P, size=N, dtype=np.uint32))
        bufs.append(x)
    ptrs = L.ptr_array([x.ptr for x in bufs])
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    root = (C.c_uint8 * 32)()
    for _ in range(3):
        L.call("tstwo_merkle_commit", ptrs, L.u32x([n] * cols), cols, C.c_void_p(layers.ptr), root)
    e0, e1 = L.Event(), L.Event()
    reps = 20
    e0.record()
    for _ in range(reps):
        L.call("tstwo_merkle_commit", ptrs, L.u32x([n] * cols), cols, C.c_void_p(layers.ptr), None)
    e1.record()
    L.sync()
    res[f"C{cols}_log{n}_us"] = round(e0.elapsed_ms(e1) / reps * 1e3, 1)
    res[f"C{cols}_log{n}_root"] = bytes(root).hex()[:16]
print(json.dumps(res))
