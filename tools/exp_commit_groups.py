#!/usr/bin/env python3
"""8 trees of 32 columns x 2^22 committed as 8 / 4 / 2 / 1 tstwo_merkle_commit_many calls (1 / 2 / 4 / 8 trees per call): does a
smaller group keep a layer's digests in the 256 MB memory-side cache for the launch that reads them?"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tstwo_amd import _lib as L
L.init(0)
n, NC, TC = 22, 256, 32
N = 1 << n
rng = np.random.default_rng(1)
cols = []
for c in range(NC):
    b = L.DeviceBuffer(4 * N); b.upload(rng.integers(0, L.P, size=N, dtype=np.uint32)); cols.append(b)
n_trees = NC // TC
layers = [L.DeviceBuffer(32 * ((2 << n) - 1)) for _ in range(n_trees)]
log_sizes = L.u32x([n] * TC)
tree_ptrs = [L.ptr_array([b.ptr for b in cols[t * TC:(t + 1) * TC]]) for t in range(n_trees)]
reqs = (L.CommitRequest * n_trees)()
for t in range(n_trees):
    reqs[t] = L.CommitRequest(tree_ptrs[t], log_sizes, TC, layers[t].ptr)
for gt in (8, 4, 2, 1, 8):
    subs = [(L.CommitRequest * gt)(*[reqs[g + k] for k in range(gt)]) for g in range(0, n_trees, gt)]

    def step():
        for sreq in subs:
            L.call("tstwo_merkle_commit_many", sreq, gt, None)
    for _ in range(60):
        step()
    L.sync()
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    L.sync()
    print(f"{gt} trees per call: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms for the 8 trees", flush=True)
