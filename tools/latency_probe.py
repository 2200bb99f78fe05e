import ctypes as C, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tstwo_amd import _lib as L
L.init(0)
vp=lambda p: C.c_void_p(p)
a=L.DeviceBuffer(4<<20); b=L.DeviceBuffer(4<<20)
a.upload(np.random.default_rng(0).integers(1, L.P, size=1<<20, dtype=np.uint32))
def t(fn, reps=200):
    fn(); L.sync()
    t0=time.perf_counter()
    for _ in range(reps): fn()
    L.sync()
    return (time.perf_counter()-t0)/reps*1e6
print("m31_add 2^10 (async)      us:", round(t(lambda: L.call("tstwo_m31_add", vp(a.ptr), vp(a.ptr), vp(b.ptr), 1<<10)),1))
print("m31_add 2^10 + sync       us:", round(t(lambda: (L.call("tstwo_m31_add", vp(a.ptr), vp(a.ptr), vp(b.ptr), 1<<10), L.sync())),1))
print("batch_inverse 2^10        us:", round(t(lambda: L.call("tstwo_m31_batch_inverse", vp(a.ptr), vp(b.ptr), 1<<10)),1))
print("batch_inverse 2^20        us:", round(t(lambda: L.call("tstwo_m31_batch_inverse", vp(a.ptr), vp(b.ptr), 1<<20)),1))
out=np.empty(8,dtype=np.uint32)
print("download 32 B             us:", round(t(lambda: L.call("tstwo_download", out.ctypes.data_as(C.c_void_p), vp(a.ptr), 32)),1))
root=(C.c_uint8*32)()
cols=L.ptr_array([a.ptr]*4)
lay=L.DeviceBuffer(32*((2<<10)-1))
print("merkle_commit log10 root  us:", round(t(lambda: L.call("tstwo_merkle_commit", cols, L.u32x([10]*4), 4, vp(lay.ptr), root)),1))
print("merkle_commit log10 async us:", round(t(lambda: L.call("tstwo_merkle_commit", cols, L.u32x([10]*4), 4, vp(lay.ptr), None)),1))
