#!/usr/bin/env python3
"""Experiment: the config-5 step (evaluate 256 columns + 8 trees) with the Merkle commit of column group g on a second stream
beside the CFFT of group g+1 — the transform is HBM-bound, the hashing VALU-bound.  Prints ms per step for the shipped form
(one evaluate call + one commit_many on one stream) and for groups of 32 / 64 / 128 columns on two streams."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tstwo_amd import _lib as L
from tstwo_amd.backend import HipBackend

hip = C.CDLL("libamdhip64.so")
L.init(0)
n, NC, TC = 22, 256, 32
N = 1 << n
rng = np.random.default_rng(1)
cols = []
for c in range(NC):
    b = L.DeviceBuffer(4 * N); b.upload(rng.integers(0, L.P, size=N, dtype=np.uint32)); cols.append(b)
backend = HipBackend()
half_initial = backend.canonic_half_coset_initial(n)
tw = L.DeviceBuffer(4 * (N // 2))
L.call("tstwo_twiddles_build", half_initial, n - 1, C.c_void_p(tw.ptr), C.c_void_p(0))
n_trees = NC // TC
layers = [L.DeviceBuffer(32 * ((2 << n) - 1)) for _ in range(n_trees)]
log_sizes = L.u32x([n] * TC)
col_ptrs = L.ptr_array([b.ptr for b in cols])
tree_ptrs = [L.ptr_array([b.ptr for b in cols[t * TC:(t + 1) * TC]]) for t in range(n_trees)]
reqs = (L.CommitRequest * n_trees)()
for t in range(n_trees):
    reqs[t] = L.CommitRequest(tree_ptrs[t], log_sizes, TC, layers[t].ptr)


def shipped():
    L.call("tstwo_cfft_evaluate", col_ptrs, NC, n, half_initial, C.c_void_p(tw.ptr), n - 1)
    L.call("tstwo_merkle_commit_many", reqs, n_trees, None)


s = [C.c_void_p(), C.c_void_p()]
for k in range(2):
    assert hip.hipStreamCreateWithFlags(C.byref(s[k]), 1) == 0          # non-blocking
ev_c = [C.c_void_p() for _ in range(n_trees)]
ev_m = [C.c_void_p() for _ in range(n_trees)]
for e in ev_c + ev_m:
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0              # disable timing


def grouped(gt):
    """gt trees (gt*32 columns) per group; CFFT on stream 0, commits on stream 1."""
    for g in range(0, n_trees, gt):
        L.call("tstwo_set_stream", s[0])
        hip.hipStreamWaitEvent(s[0], ev_m[g], 0)                          # last step's commit of this group has read the columns
        sub = L.ptr_array([b.ptr for b in cols[g * TC:(g + gt) * TC]])
        L.call("tstwo_cfft_evaluate", sub, gt * TC, n, half_initial, C.c_void_p(tw.ptr), n - 1)
        hip.hipEventRecord(ev_c[g], s[0])
        L.call("tstwo_set_stream", s[1])
        hip.hipStreamWaitEvent(s[1], ev_c[g], 0)
        if gt == 1:
            L.call("tstwo_merkle_commit", tree_ptrs[g], log_sizes, TC, C.c_void_p(layers[g].ptr), None)
        else:
            sub_reqs = (L.CommitRequest * gt)(*[reqs[g + k] for k in range(gt)])
            L.call("tstwo_merkle_commit_many", sub_reqs, gt, None)
        hip.hipEventRecord(ev_m[g], s[1])


def sync_all():
    hip.hipStreamSynchronize(s[0]); hip.hipStreamSynchronize(s[1]); L.call("tstwo_set_stream", None); L.sync()


def timeit(fn, label, steps=30):
    for _ in range(40):
        fn()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync_all()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", flush=True)


root_ref = None
timeit(shipped, "one stream, evaluate(256) + commit_many(8)")
ref = [l.download(np.uint8, 32).tobytes() for l in layers]
for gt in (1, 2, 4):
    for e in ev_m:
        hip.hipEventRecord(e, s[1])
    timeit(lambda: grouped(gt), f"two streams, groups of {gt * TC} columns")
L.call("tstwo_set_stream", None)
print("done")
