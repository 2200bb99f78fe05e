#!/usr/bin/env python3
"""Digest of evaluate / interpolate outputs of seeded columns under the CURRENT library and environment: equal digests across
plans (TSTWO_HIP_LIB = experiments build + TSTWO_CFFT_KB / KA / LOGTA) mean equal results; the shipped plan is oracle-checked by tests/.
    python tools/plan_digest.py --log 22 [--cols 3]"""
import argparse, ctypes as C, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402
from tstwo_amd.backend import HipBackend  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--log", type=int, default=22)
ap.add_argument("--cols", type=int, default=3)
a = ap.parse_args()
L.init(0)
n, N = a.log, 1 << a.log
half = HipBackend.canonic_half_coset_initial(n)
rng = np.random.default_rng(n)
bufs = []
for _ in range(a.cols):
    b = L.DeviceBuffer(4 * N)
    b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
    bufs.append(b)
tw, itw = L.DeviceBuffer(4 * (N // 2)), L.DeviceBuffer(4 * (N // 2))
L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(tw.ptr), C.c_void_p(itw.ptr))
ptrs = L.ptr_array([b.ptr for b in bufs])
L.call("tstwo_cfft_evaluate", ptrs, a.cols, n, half, C.c_void_p(tw.ptr), n - 1)
h1 = hashlib.blake2s(b"".join(b.download().tobytes() for b in bufs)).hexdigest()[:16]
for b in bufs:
    b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32))
L.call("tstwo_cfft_interpolate", ptrs, a.cols, n, half, C.c_void_p(itw.ptr), n - 1)
h2 = hashlib.blake2s(b"".join(b.download().tobytes() for b in bufs)).hexdigest()[:16]
print(f"log {n} evaluate {h1} interpolate {h2}  [{L.version()}; KB={os.environ.get('TSTWO_CFFT_KB')} KA={os.environ.get('TSTWO_CFFT_KA')} LOGTA={os.environ.get('TSTWO_CFFT_LOGTA')}]", flush=True)
