#!/usr/bin/env python3
"""Host -> device rates of the hand-over paths: pageable tstwo_upload, registered / library-pinned memory through tstwo_upload_async
(columns of 2^22 words, and one 1 GiB piece).    python tools/h2d_rate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstwo_amd import _lib as L  # noqa: E402
L.init(0)
N, C_ = 1 << 22, 64
dev = [L.DeviceBuffer(4 * N) for _ in range(C_)]
host = [np.full(N, 7 + c, dtype=np.uint32) for c in range(C_)]
def rate(label, fn, nbytes):
    fn(); L.sync()
    t0 = time.perf_counter(); fn(); L.sync(); dt = time.perf_counter() - t0
    print(f"{label}: {nbytes / dt / 1e9:.1f} GB/s ({dt * 1e3:.1f} ms)", flush=True)
rate("pageable tstwo_upload, 64 x 16 MiB", lambda: [d.upload(h) for d, h in zip(dev, host)], 4.0 * N * C_)
t0 = time.perf_counter()
for h in host: L.host_register(h)
print(f"tstwo_host_register of 1 GiB in 64 pieces: {(time.perf_counter() - t0) * 1e3:.1f} ms")
rate("registered tstwo_upload_async, 64 x 16 MiB", lambda: [d.upload_async(h) for d, h in zip(dev, host)], 4.0 * N * C_)
rate("registered tstwo_upload (synchronous), 64 x 16 MiB", lambda: [d.upload(h) for d, h in zip(dev, host)], 4.0 * N * C_)
for h in host: L.host_unregister(h)
pins = [L.PinnedArray(N) for _ in range(C_)]
for p, h in zip(pins, host): p.array[:] = h
rate("library-pinned tstwo_upload_async, 64 x 16 MiB", lambda: [d.upload_async(p.array) for d, p in zip(dev, pins)], 4.0 * N * C_)
big_d, big_h = L.DeviceBuffer(1 << 30), L.PinnedArray(1 << 28)
rate("library-pinned tstwo_upload_async, one 1 GiB piece", lambda: big_d.upload_async(big_h.array), float(1 << 30))
half = 1 << 27
rate("library-pinned tstwo_upload_async, 2 x 512 MiB (two copy streams)", lambda: (big_d.upload_async(big_h.array[:half]), big_d.upload_async(big_h.array[half:], 4 * half)), float(1 << 30))
