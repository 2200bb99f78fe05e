#!/usr/bin/env python3
"""Debug aid: compares the specialised CFFT pass kernels with the generic kernel and the CPU oracle.
TSTWO_CFFT_GENERIC bit 0: generic bottom pass, bit 1: generic strided passes, bit 2: skip the bottom pass."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 1 << n
rng = np.random.default_rng(0)
a = rng.integers(0, L.P, size=N, dtype=np.uint32)
half = 1 << (31 - (n + 1))
tw, itw = L.DeviceBuffer(2 * N), L.DeviceBuffer(2 * N)
L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(tw.ptr), C.c_void_p(itw.ptr))


def run(mode, inverse=False, data=a):
    os.environ["TSTWO_CFFT_GENERIC"] = str(mode)
    b = L.DeviceBuffer(4 * N)
    b.upload(data)
    fn = "tstwo_cfft_interpolate" if inverse else "tstwo_cfft_evaluate"
    L.call(fn, L.ptr_array([b.ptr]), 1, n, half, C.c_void_p((itw if inverse else tw).ptr), n - 1)
    L.sync()
    return b.download(np.uint32, N)


otw, oitw = orc.precompute_twiddles(half, n - 1)
full = orc.cfft_evaluate(a, n, half, otw, n - 1)
for mode, name in [(3, "generic/generic"), (2, "fast bottom, generic strided"), (1, "generic bottom, fast strided"), (0, "fast/fast")]:
    got = run(mode)
    print(f"evaluate  {name:32s}: {'OK' if (got == full).all() else 'MISMATCH %d' % int((got != full).sum())}")
    back = run(mode, inverse=True, data=full)
    print(f"interpolate {name:30s}: {'OK' if (back == a).all() else 'MISMATCH %d' % int((back != a).sum())}")

# multi-column (columns-per-workgroup loop): 5 columns, forced cpw values
ncol = 5
cols_h = [rng.integers(0, L.P, size=N, dtype=np.uint32) for _ in range(ncol)]
exp = [orc.cfft_evaluate(c, n, half, otw, n - 1) for c in cols_h]
for cpw in (1, 2, 4, 8):
    os.environ["TSTWO_CFFT_CPW"] = str(cpw)
    for mode, name in [(2, "fast bottom only"), (1, "fast strided only"), (0, "fast/fast")]:
        os.environ["TSTWO_CFFT_GENERIC"] = str(mode)
        bufs = []
        for c in cols_h:
            b = L.DeviceBuffer(4 * N); b.upload(c); bufs.append(b)
        L.call("tstwo_cfft_evaluate", L.ptr_array([b.ptr for b in bufs]), ncol, n, half, C.c_void_p(tw.ptr), n - 1)
        L.sync()
        bad = [i for i, b in enumerate(bufs) if not (b.download(np.uint32, N) == exp[i]).all()]
        L.call("tstwo_cfft_interpolate", L.ptr_array([b.ptr for b in bufs]), ncol, n, half, C.c_void_p(itw.ptr), n - 1)
        L.sync()
        badi = [i for i, b in enumerate(bufs) if not (b.download(np.uint32, N) == cols_h[i]).all()]
        print(f"cpw={cpw} {name:18s}: evaluate bad cols {bad}  roundtrip bad cols {badi}")
