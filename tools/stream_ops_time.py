#!/usr/bin/env python3
"""Streaming column operations at 2^24 words: microseconds and TB/s of the bytes each one must move (an audit for kernels that
sit far below the 5-6 TB/s the part gives to plain kernels of the same shape)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tstwo_amd import _lib as L
L.init(0)
rng = np.random.default_rng(0)
n = 1 << 24
vp = C.c_void_p


def buf(words=n, nonzero=False):
    b = L.DeviceBuffer(4 * words)
    b.upload(rng.integers(1 if nonzero else 0, L.P, size=words, dtype=np.uint32))
    return b


def timed(label, nbytes, fn, reps=60):
    for _ in range(150):
        fn()
    L.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    L.sync(); dt = (time.perf_counter() - t0) / reps
    print(f"{label:44s} {dt * 1e6:8.1f} us  {nbytes / dt / 1e12:5.2f} TB/s", flush=True)


a, b, o = buf(), buf(), buf()
timed("m31_add 2^24", 12 * n, lambda: L.call("tstwo_m31_add", vp(a.ptr), vp(b.ptr), vp(o.ptr), n))
timed("m31_mul 2^24", 12 * n, lambda: L.call("tstwo_m31_mul", vp(a.ptr), vp(b.ptr), vp(o.ptr), n))
timed("m31_neg 2^24", 8 * n, lambda: L.call("tstwo_m31_neg", vp(a.ptr), vp(o.ptr), n))
nz = buf(nonzero=True)
timed("m31_batch_inverse_async 2^24", 8 * n, lambda: L.call("tstwo_m31_batch_inverse_async", vp(nz.ptr), vp(o.ptr), n))
m = 1 << 22
qa = [buf(m, True) for _ in range(4)]; qb = [buf(m, True) for _ in range(4)]; qo = [buf(m) for _ in range(4)]
timed("qm31_mul 2^22", 48 * m, lambda: L.call("tstwo_qm31_mul", L.p4([x.ptr for x in qa]), L.p4([x.ptr for x in qb]), L.p4([x.ptr for x in qo]), m))
timed("qm31_batch_inverse_async 2^22", 32 * m, lambda: L.call("tstwo_qm31_batch_inverse_async", L.p4([x.ptr for x in qa]), L.p4([x.ptr for x in qo]), m))
cols = L.ptr_array([a.ptr])
timed("bit_reverse 2^24 (1 column)", 8 * n, lambda: L.call("tstwo_bit_reverse", cols, 1, n))
tw, itw = L.DeviceBuffer(4 << 23), L.DeviceBuffer(4 << 23)
timed("twiddles_build log 23 (tw only)", 4 << 23, lambda: L.call("tstwo_twiddles_build", 1 << (31 - 25), 23, vp(tw.ptr), vp(0)), reps=20)
timed("twiddles_build log 23 (tw + itw)", 12 << 23, lambda: L.call("tstwo_twiddles_build", 1 << (31 - 25), 23, vp(tw.ptr), vp(itw.ptr)), reps=20)
