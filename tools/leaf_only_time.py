import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tstwo_amd import _lib as L
L.init(0)
n, N, cols = 22, 1 << 22, 32
rng = np.random.default_rng(1)
bufs = []
for _ in range(cols):
    b = L.DeviceBuffer(4 * N); b.upload(rng.integers(0, 2**31 - 1, size=N, dtype=np.uint32)); bufs.append(b)
ptrs = L.ptr_array([b.ptr for b in bufs])
out = L.DeviceBuffer(32 * N)
for _ in range(100):
    L.call("tstwo_merkle_commit_layer", n, None, ptrs, cols, C.c_void_p(out.ptr))
ts = []
for _ in range(300):
    e0, e1 = L.Event(), L.Event()
    e0.record(); L.call("tstwo_merkle_commit_layer", n, None, ptrs, cols, C.c_void_p(out.ptr)); e1.record()
    ts.append(e0.elapsed_ms(e1))
print(f"leaf layer alone: avg {sum(ts)/len(ts)*1e3:.1f} us  min {min(ts)*1e3:.1f} us")
