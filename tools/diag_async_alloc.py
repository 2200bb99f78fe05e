#!/usr/bin/env python3
"""Diagnostic for the round-1 "different Merkle roots run to run with hipMallocAsync" report.

Runs the 32 x 2^22 evaluate + commit sequence of tests/test_gpu_config5.py under TSTWO_ALLOC_ASYNC several times and
checks, stage by stage, (a) that no two LIVE blocks handed out by hipMallocAsync overlap, (b) each uploaded column read
back, (c) the twiddles, (d) every evaluation word against a pool-mode run of the same input, (e) every layer of the tree.
Prints one line per finding.  Usage: python tools/diag_async_alloc.py [passes]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import splitmix_column  # noqa: E402
from tstwo_amd import _lib as L  # noqa: E402

n, NC = 22, 32
N = 1 << n
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L.init(0)
from tstwo_amd.backend import HipBackend  # noqa: E402
half = HipBackend.canonic_half_coset_initial(n)
coeffs = [splitmix_column(100 + c, N) for c in range(NC)]


def overlaps(blocks):
    iv = sorted((b.ptr, b.ptr + b.nbytes, name) for name, b in blocks)
    bad = []
    for (a0, a1, an), (b0, b1, bn) in zip(iv, iv[1:]):
        if b0 < a1:
            bad.append((an, hex(a0), hex(a1), bn, hex(b0), hex(b1)))
    return bad


def run(tag, verify=None, order=None, exact=True, lazy=False):
    """exact=True: the call sequence of tests/test_gpu_config5.py (no host synchronisation between the twiddle build, the
    transform, the layers allocation and the commit); verification happens afterwards."""
    order = order or list(range(NC))
    blocks, bufs = [], []
    for i in order:
        b = L.DeviceBuffer(4 * N)
        b.upload(coeffs[i])
        bufs.append(b)
        blocks.append((f"col{i}", b))
    tw = L.DeviceBuffer(4 * (N // 2))
    blocks.append(("tw", tw))
    L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(tw.ptr), C.c_void_p(0))
    ptrs = L.ptr_array([b.ptr for b in bufs])
    if not exact:
        L.sync()
    L.call("tstwo_cfft_evaluate", ptrs, NC, n, half, C.c_void_p(tw.ptr), n - 1)
    if not exact:
        L.sync()
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    blocks.append(("layers", layers))
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", ptrs, L.u32x([n] * NC), NC, C.c_void_p(layers.ptr), root)
    for x in overlaps(blocks):
        print(f"[{tag}] OVERLAP of live blocks: {x}", flush=True)
    if verify is not None and lazy:
        want = verify["root"] if order == list(range(NC)) else verify.get("root_rev")
        if want is None or bytes(root) == want:
            print(f"[{tag}] root OK (no downloads)", flush=True)
            for _, b in blocks:
                b.free()
            return {"root": bytes(root)}
        print(f"[{tag}] ROOT DIFFERS -> post-mortem", flush=True)
    twh = tw.download(np.uint32, N // 2)
    evs = [b.download(np.uint32, N) for b in bufs]
    lay = layers.download(np.uint8, 32 * ((2 << n) - 1))
    res = {"tw": twh, "evs": dict(zip(order, evs)), "layers": lay, "root": bytes(root)}
    if verify is not None:
        if not (twh == verify["tw"]).all():
            d = np.flatnonzero(twh != verify["tw"])
            print(f"[{tag}] twiddles differ: {d.size} words, range [{d[0]}, {d[-1]}]", flush=True)
        for i in order:
            a, b_ = res["evs"][i], verify["evs"][i]
            if not (a == b_).all():
                d = np.flatnonzero(a != b_)
                same_as_coeffs = int((a == coeffs[i]).sum())
                print(f"[{tag}] evaluation of col{i} (slot {order.index(i)}, ptr {hex(bufs[order.index(i)].ptr)}) differs: {d.size} words, range [{d[0]}, {d[-1]}], "
                      f"first {d[:6].tolist()}; words equal to the uploaded coefficients: {same_as_coeffs}", flush=True)
        if order == list(range(NC)):
            if not (lay == verify["layers"]).all():
                d = np.unique(np.flatnonzero(lay != verify["layers"]) // 32)
                lv = np.floor(np.log2(d + 1)).astype(int)
                print(f"[{tag}] layers differ: {d.size} digests, layers (log size) {sorted(set(lv.tolist()))}", flush=True)
        print(f"[{tag}] root {'OK' if order != list(range(NC)) or res['root'] == verify['root'] else 'DIFFERS'}  ptrs col0 {hex(bufs[0].ptr)} tw {hex(tw.ptr)} layers {hex(layers.ptr)}", flush=True)
    for _, b in blocks:
        b.free()
    return res


L.call("tstwo_set_alloc_mode", L.ALLOC_POOL)
ref = run("pool-ref")
REV = list(range(NC))[::-1]
ref["root_rev"] = run("pool-rev", order=REV)["root"]
ref2 = run("pool-2", verify=ref)
seq = os.environ.get("DIAG_SEQ", "poolpoison,direct,async,asyncpoison").split(",")
MODES = {"async": L.ALLOC_ASYNC, "direct": L.ALLOC_DIRECT, "pool": L.ALLOC_POOL, "poolpoison": L.ALLOC_POOL | L.ALLOC_POISON,
         "asyncpoison": L.ALLOC_ASYNC | L.ALLOC_POISON}
for name in seq:
    L.call("tstwo_set_alloc_mode", MODES[name])
    for k in range(passes):
        run(f"{name}-{k}", verify=ref, order=list(range(NC)) if k % 2 == 0 else REV, lazy=k > 0)
print("diag done", flush=True)
