"""Helpers for the -m gpu parity tests: thin numpy <-> device plumbing over the C ABI."""
import ctypes as C

import numpy as np

from tstwo_amd import _lib as L


def dev(arr) -> L.DeviceBuffer:
    arr = np.ascontiguousarray(arr)
    b = L.DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        b.upload(arr)
    return b


def dev_empty(n_words) -> L.DeviceBuffer:
    return L.DeviceBuffer(max(4 * n_words, 16))


def host(buf: L.DeviceBuffer, n, dtype=np.uint32):
    return buf.download(dtype, n)


def ptrs(bufs):
    return L.ptr_array([b.ptr for b in bufs])


def p4(bufs):
    return L.p4([b.ptr for b in bufs])


def vp(buf, offset=0):
    return C.c_void_p((buf.ptr if buf is not None else 0) + offset) if buf is not None else C.c_void_p(0)
