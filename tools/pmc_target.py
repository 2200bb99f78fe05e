#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: two calibration kernels with known byte counts (16-byte and 4-byte lane
accesses), then the bench step (C x 2^n evaluate + one Merkle commit per 32 columns) twice, then eval_at_point.

    pmc_target.py [--cols C] [--log-size n] [--cfft-only]

--cfft-only (bench.py's own child passes): calibration + the evaluate call only.  The transform is data-oblivious, so the
columns are device copies of one random column (no 4 GiB of host data to generate and upload)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=256)
ap.add_argument("--log-size", type=int, default=22)
ap.add_argument("--cfft-only", action="store_true")
args = ap.parse_args()

L.init(0)
vp = lambda p: C.c_void_p(p)
ncal = 1 << 26
a, b, o = L.DeviceBuffer(4 * ncal + 16), L.DeviceBuffer(4 * ncal + 16), L.DeviceBuffer(4 * ncal + 16)
a.zero(); b.zero(); o.zero()
L.call("tstwo_m31_add", vp(a.ptr), vp(b.ptr), vp(o.ptr), ncal)                    # k_m31_binop_vec4: 16 B per lane
L.call("tstwo_m31_add", vp(a.ptr + 4), vp(b.ptr + 4), vp(o.ptr + 4), ncal)        # k_m31_binop_scalar: 4 B per lane
n, cols = args.log_size, args.cols
N = 1 << n
rng = np.random.default_rng(0)
bufs = [L.DeviceBuffer(4 * N) for _ in range(cols)]
bufs[0].upload(rng.integers(0, L.P, size=N, dtype=np.uint32))
for x in bufs[1:]:
    L.call("tstwo_copy", vp(x.ptr), vp(bufs[0].ptr), 4 * N)
ptrs = L.ptr_array([x.ptr for x in bufs])
half = 1 << (31 - (n + 1))
tw = L.DeviceBuffer(2 * N)
L.call("tstwo_twiddles_build", half, n - 1, vp(tw.ptr), vp(0))
L.sync()
if args.cfft_only:
    for _ in range(2):
        L.call("tstwo_cfft_evaluate", ptrs, cols, n, half, vp(tw.ptr), n - 1)
    L.sync()
    print("done")
    sys.exit(0)
tree = 32 if cols % 32 == 0 else cols
layers = L.DeviceBuffer(32 * ((2 << n) - 1))
for _ in range(2):
    L.call("tstwo_cfft_evaluate", ptrs, cols, n, half, vp(tw.ptr), n - 1)
    for t in range(0, cols, tree):
        L.call("tstwo_merkle_commit", L.ptr_array([x.ptr for x in bufs[t:t + tree]]), L.u32x([n] * tree), tree, vp(layers.ptr), None)
# BASELINE config 3 / 4 kernels: quotients (C = 4, log 22, one sample batch), QM31 batch inverse (log 22), fold_circle_into_line
# (log 24), Merkle commit of 4 columns of 2^24 (the FRI first layer)
import tstwo_amd as T  # noqa: E402
from tstwo_amd.pcs import PointSample, column_sample_batches  # noqa: E402
from tstwo_amd.quotients import marshal_quotient_args  # noqa: E402
n3 = 22
dom3 = T.CanonicCoset(n3).circleDomain()
cols3 = [T.HipColumn(rng.integers(0, L.P, size=1 << n3, dtype=np.uint32)) for _ in range(4)]
pt = T.SECURE_FIELD_CIRCLE_GEN
samples = [[PointSample(pt, T.QM31.from_u32_unchecked(*map(int, rng.integers(1, L.P, size=4))))] for _ in range(4)]
_keep, qargs = marshal_quotient_args(dom3, cols3, T.QM31.from_u32_unchecked(1, 2, 3, 4), column_sample_batches(samples))
qout = T.SecureColumnByCoords.uninitialized(1 << n3)
sec = T.SecureColumnByCoords.from_numpy([rng.integers(1, L.P, size=1 << n3, dtype=np.uint32) for _ in range(4)])
for _ in range(2):
    L.call("tstwo_quotients_accumulate_async", *qargs, qout.ptrs())
    L.call("tstwo_qm31_batch_inverse_async", sec.ptrs(), qout.ptrs(), 1 << n3)
L.call("tstwo_check_zero_flag")
n4 = 24
dom4 = T.CanonicCoset(n4).circleDomain()
tw4 = T.precompute_twiddles(dom4.halfCoset)
src4 = T.SecureEvaluation(dom4, T.SecureColumnByCoords.from_numpy([rng.integers(0, L.P, size=1 << n4, dtype=np.uint32) for _ in range(4)]))
dst4 = T.LineEvaluation.new_zero(T.LineDomain(dom4.halfCoset))
lay4 = L.DeviceBuffer(32 * ((2 << n4) - 1))
for _ in range(2):
    L.call("tstwo_fri_fold_circle_into_line", dst4.values.ptrs(), 1 << (n4 - 1), src4.values.ptrs(), n4, vp(tw4.itwiddles.ptr), n4 - 1, L.u32x([19283, 1, 2, 3]))
    L.call("tstwo_merkle_commit", L.ptr_array([c.ptr for c in src4.values.columns]), L.u32x([n4] * 4), 4, vp(lay4.ptr), None)
# PolyOps.eval_at_point: one column log 24 (64 MiB read once)
from tstwo_amd.circle import SECURE_FIELD_CIRCLE_GEN as G  # noqa: E402
px, py, o4 = L.u32x(G.x.tup()), L.u32x(G.y.tup()), L.u32x([0] * 4)
big = L.DeviceBuffer(4 << 24)
big.upload(rng.integers(0, L.P, size=1 << 24, dtype=np.uint32))
L.call("tstwo_eval_at_point", vp(big.ptr), 24, px, py, o4)
L.sync()
print("done")
