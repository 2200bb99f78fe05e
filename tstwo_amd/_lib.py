"""ctypes binding of libtstwo_hip.so — the same C ABI (include/tstwo_hip.h) a bun:ffi HipBackend binds.

There is NO CPU fallback: loading fails loudly if the shared library is missing, and every entry
point returns an error if no GPU is present.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSTWO_HIP_LIB") or os.path.join(HERE, "libtstwo_hip.so")
# The experiments build (python -m tstwo_amd.build --experiments): the same library with DESIGN §8's TSTWO_* tuning / A-B switches
# read from the environment.  Never loaded by default — tools/ and the tests that pin a non-default branch put it in TSTWO_HIP_LIB.
LIB_EXP_PATH = os.path.join(HERE, "libtstwo_hip_exp.so")
P = 2147483647


class TstwoError(RuntimeError):
    """Raised with the reference's own error text (e.g. "0 has no inverse")."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
P4 = vp * 4
P2 = vp * 2

class DecommitRequest(C.Structure):
    """tstwo_decommit_request (include/tstwo_hip.h)."""
    _fields_ = [("layers", vp), ("max_log", C.c_uint32), ("cols", C.POINTER(vp)), ("col_log_sizes", u32p), ("n_cols", C.c_size_t),
                ("query_logs", u32p), ("queries", C.POINTER(C.POINTER(C.c_uint64))), ("n_queries", C.POINTER(C.c_size_t)),
                ("n_query_sets", C.c_size_t)]


class FriLayer(C.Structure):
    """tstwo_fri_layer (include/tstwo_hip.h)."""
    _fields_ = [("layers", vp), ("max_log", C.c_uint32), ("cols", C.POINTER(vp)), ("eval_logs", u32p), ("n_evals", C.c_size_t)]


class CommitRequest(C.Structure):
    """tstwo_commit_request (include/tstwo_hip.h)."""
    _fields_ = [("cols", C.POINTER(vp)), ("log_sizes", u32p), ("n_cols", C.c_size_t), ("layers", vp)]


class FriLayerOut(C.Structure):
    """tstwo_fri_layer_out (include/tstwo_hip.h)."""
    _fields_ = [("log_size", C.c_uint32), ("cols", vp * 4), ("layers", vp)]


_SIGS = {
    "tstwo_init": [C.c_int],
    "tstwo_shutdown": [],
    "tstwo_device_count": [C.POINTER(C.c_int)],
    "tstwo_device_name": [C.c_char_p, C.c_size_t],
    "tstwo_set_stream": [vp],
    "tstwo_sync": [],
    "tstwo_malloc": [C.POINTER(vp), C.c_size_t],
    "tstwo_free": [vp],
    "tstwo_trim": [],
    "tstwo_set_alloc_mode": [C.c_int],
    "tstwo_upload": [vp, vp, C.c_size_t],
    "tstwo_host_register": [vp, C.c_size_t],
    "tstwo_host_unregister": [vp],
    "tstwo_host_alloc": [C.POINTER(vp), C.c_size_t],
    "tstwo_host_free": [vp],
    "tstwo_upload_async": [vp, vp, C.c_size_t],
    "tstwo_upload_fence": [],
    "tstwo_upload_wait": [],
    "tstwo_download": [vp, vp, C.c_size_t],
    "tstwo_download_many": [C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, vp],
    "tstwo_copy": [vp, vp, C.c_size_t],
    "tstwo_zero": [vp, C.c_size_t],
    "tstwo_graph_begin_capture": [],
    "tstwo_graph_end_capture": [C.POINTER(vp)],
    "tstwo_graph_launch": [vp],
    "tstwo_graph_destroy": [vp],
    "tstwo_event_create": [C.POINTER(vp)],
    "tstwo_event_record": [vp],
    "tstwo_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
    "tstwo_event_destroy": [vp],
    "tstwo_comm_unique_id": [u8p],
    "tstwo_comm_init": [C.c_int, C.c_int, u8p],
    "tstwo_comm_destroy": [],
    "tstwo_comm_info": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "tstwo_allgather_roots": [vp, vp],
    "tstwo_allgather": [vp, vp, C.c_size_t],
    "tstwo_allgather_async": [vp, vp, C.c_size_t],
    "tstwo_comm_wait": [],
    "tstwo_m31_add": [vp, vp, vp, C.c_size_t],
    "tstwo_m31_sub": [vp, vp, vp, C.c_size_t],
    "tstwo_m31_mul": [vp, vp, vp, C.c_size_t],
    "tstwo_m31_neg": [vp, vp, C.c_size_t],
    "tstwo_m31_batch_inverse": [vp, vp, C.c_size_t],
    "tstwo_cm31_batch_inverse": [P2, P2, C.c_size_t],
    "tstwo_qm31_batch_inverse": [P4, P4, C.c_size_t],
    "tstwo_m31_batch_inverse_async": [vp, vp, C.c_size_t],
    "tstwo_cm31_batch_inverse_async": [P2, P2, C.c_size_t],
    "tstwo_qm31_batch_inverse_async": [P4, P4, C.c_size_t],
    "tstwo_check_zero_flag": [],
    "tstwo_qm31_mul": [P4, P4, P4, C.c_size_t],
    "tstwo_secure_accumulate": [P4, P4, C.c_size_t],
    "tstwo_bit_reverse": [C.POINTER(vp), C.c_size_t, C.c_size_t],
    "tstwo_twiddles_build": [C.c_uint32, C.c_uint32, vp, vp],
    "tstwo_cfft_evaluate": [C.POINTER(vp), C.c_size_t, C.c_uint32, C.c_uint32, vp, C.c_uint32],
    "tstwo_cfft_interpolate": [C.POINTER(vp), C.c_size_t, C.c_uint32, C.c_uint32, vp, C.c_uint32],
    "tstwo_cfft_interpolate_to": [C.POINTER(vp), C.POINTER(vp), C.c_size_t, C.c_uint32, C.c_uint32, vp, C.c_uint32],
    "tstwo_cfft_evaluate_extended": [C.POINTER(vp), C.c_uint32, C.POINTER(vp), C.c_size_t, C.c_uint32, C.c_uint32, vp, C.c_uint32],
    "tstwo_cfft_plan_passes": [C.c_uint32, C.c_size_t, u32p],
    "tstwo_poly_extend": [vp, C.c_uint32, vp, C.c_uint32],
    "tstwo_eval_at_point": [vp, C.c_uint32, u32p, u32p, u32p],
    "tstwo_eval_at_point_batch": [C.POINTER(vp), C.c_size_t, C.c_uint32, u32p, u32p, u32p],
    "tstwo_line_interpolate": [P4, C.c_uint32, vp, C.c_uint32, P4],
    "tstwo_fri_fold_line": [P4, C.c_uint32, vp, C.c_uint32, u32p, P4],
    "tstwo_fri_fold_circle_into_line": [P4, C.c_size_t, P4, C.c_uint32, vp, C.c_uint32, u32p],
    "tstwo_fri_fold_line_tw": [P4, C.c_uint32, vp, u32p, P4],
    "tstwo_fri_fold_circle_into_line_tw": [P4, C.c_size_t, P4, C.c_uint32, vp, u32p],
    "tstwo_fri_fold_line_dev": [P4, C.c_uint32, vp, C.c_uint32, vp, P4],
    "tstwo_fri_fold_circle_into_line_dev": [P4, C.c_size_t, P4, C.c_uint32, vp, C.c_uint32, vp],
    "tstwo_channel_mix_root_draw_felt": [vp, vp, vp],
    "tstwo_fri_fold_line_rows": [P4, C.c_uint32, C.c_size_t, C.c_size_t, vp, C.c_uint32, u32p, P4],
    "tstwo_fri_fold_circle_into_line_rows": [P4, P4, C.c_uint32, C.c_size_t, C.c_size_t, vp, C.c_uint32, u32p],
    "tstwo_fri_decompose": [P4, C.c_size_t, P4, u32p],
    "tstwo_merkle_commit_layer": [C.c_uint32, vp, C.POINTER(vp), C.c_size_t, vp],
    "tstwo_merkle_commit": [C.POINTER(vp), u32p, C.c_size_t, vp, u8p],
    "tstwo_merkle_commit_many": [C.POINTER(CommitRequest), C.c_size_t, u8p],
    "tstwo_grind_blake2s": [u8p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)],
    "tstwo_merkle_decommit": [vp, C.c_uint32, C.POINTER(vp), u32p, C.c_size_t, u32p, C.POINTER(C.POINTER(C.c_uint64)),
                              C.POINTER(C.c_size_t), C.c_size_t, u32p, C.POINTER(C.c_size_t), u8p, C.POINTER(C.c_size_t),
                              u32p, C.POINTER(C.c_size_t)],
    "tstwo_merkle_decommit_many": [C.POINTER(DecommitRequest), C.c_size_t, u32p, u8p, u32p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "tstwo_fri_commit_layers": [C.POINTER(vp), u32p, C.c_size_t, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_size_t, C.POINTER(vp),
                                C.POINTER(FriLayerOut), C.c_size_t, C.POINTER(C.c_size_t)],
    "tstwo_fri_decommit": [C.POINTER(FriLayer), C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u8p, u32p,
                           u8p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "tstwo_gather_words": [C.POINTER(vp), C.POINTER(C.c_uint64), C.c_uint32, C.c_size_t, u32p],
    "tstwo_quotients_accumulate_samples": [C.c_uint32, C.c_uint32, C.POINTER(vp), C.c_size_t, C.c_size_t, u32p, u32p, u32p, u32p, u32p, P4],
    "tstwo_quotients_accumulate": [C.c_uint32, C.c_uint32, C.POINTER(vp), C.c_size_t, C.c_size_t, u32p, u32p, u32p,
                                   u32p, u32p, u32p, u32p, u32p, P4],
    "tstwo_quotients_accumulate_samples_async": [C.c_uint32, C.c_uint32, C.POINTER(vp), C.c_size_t, C.c_size_t, u32p, u32p, u32p, u32p, u32p, P4],
    "tstwo_quotients_accumulate_async": [C.c_uint32, C.c_uint32, C.POINTER(vp), C.c_size_t, C.c_size_t, u32p, u32p, u32p,
                                         u32p, u32p, u32p, u32p, u32p, P4],
}
ALLOC_POOL, ALLOC_DIRECT, ALLOC_ASYNC, ALLOC_POISON = 0, 1, 2, 0x10
# c_void_p arguments above are DEVICE addresses, except these (host memory of any element type)
HOST_VOID_ARGS = {"tstwo_upload": {1}, "tstwo_download": {0}, "tstwo_download_many": {3}, "tstwo_host_register": {0}, "tstwo_host_unregister": {0},
                  "tstwo_host_free": {0}, "tstwo_upload_async": {1}}
# every symbol include/tstwo_hip.h declares (tests check the library exports all of them)
EXPORTS = sorted(list(_SIGS) + ["tstwo_last_error", "tstwo_version", "tstwo_merkle_layers_bytes"])

_lib = None


def lib() -> C.CDLL:
    """dlopen the in-tree HIP library.  Raises if it was not built (python -m tstwo_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TstwoError(-1, f"{LIB_PATH} is missing: build it with `python -m tstwo_amd.build` "
                                 "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = args
        L.tstwo_last_error.restype = C.c_char_p
        L.tstwo_last_error.argtypes = []
        L.tstwo_version.restype = C.c_char_p
        L.tstwo_version.argtypes = []
        L.tstwo_merkle_layers_bytes.restype = C.c_size_t
        L.tstwo_merkle_layers_bytes.argtypes = [C.c_uint32]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc:
        raise TstwoError(rc, lib().tstwo_last_error().decode("utf-8", "replace"))


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args))


_initialised = False


def init(device: int | None = None) -> None:
    """Select the GPU (LOCAL_RANK by default, one process per GPU) and create the stream."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        n = device_count()
        if n:
            device %= n
    call("tstwo_init", device)
    _initialised = True


def ensure_init() -> None:
    if not _initialised:
        init()


def version() -> str:
    """tstwo_version(): "... +experiments" for the experiments build."""
    return lib().tstwo_version().decode()


def device_count() -> int:
    n = C.c_int(0)
    lib().tstwo_device_count(C.byref(n))
    return n.value


def device_name() -> str:
    ensure_init()
    buf = C.create_string_buffer(256)
    call("tstwo_device_name", buf, 256)
    return buf.value.decode()


def sync() -> None:
    ensure_init()
    call("tstwo_sync")


def host_register(arr: np.ndarray) -> None:
    """Page-lock the memory of a host array the caller owns (tstwo_host_register): uploads from it are one DMA at link rate."""
    ensure_init()
    call("tstwo_host_register", arr.ctypes.data_as(vp), arr.nbytes)


def host_unregister(arr: np.ndarray) -> None:
    call("tstwo_host_unregister", arr.ctypes.data_as(vp))


def upload_fence() -> None:
    call("tstwo_upload_fence")


def upload_wait() -> None:
    call("tstwo_upload_wait")


class PinnedArray:
    """A numpy array over page-locked host memory from the library (tstwo_host_alloc / tstwo_host_free)."""

    def __init__(self, count: int, dtype=np.uint32):
        ensure_init()
        dt = np.dtype(dtype)
        p = vp()
        call("tstwo_host_alloc", C.byref(p), count * dt.itemsize)
        self._ptr = p.value
        buf = (C.c_uint8 * (count * dt.itemsize)).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=dt, count=count)

    def free(self) -> None:
        if self._ptr:
            self.array = None
            call("tstwo_host_free", vp(self._ptr))
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:          # noqa: BLE001 — interpreter shutdown
            pass


class DeviceBuffer:
    """An owned device allocation (tstwo_malloc / tstwo_free)."""

    __slots__ = ("ptr", "nbytes")

    def __init__(self, nbytes: int):
        ensure_init()
        p = vp()
        call("tstwo_malloc", C.byref(p), nbytes)
        self.ptr = p.value
        self.nbytes = nbytes

    @staticmethod
    def adopt(ptr: int, nbytes: int) -> "DeviceBuffer":
        """Take ownership of a block the library allocated with tstwo_malloc and handed out (tstwo_fri_commit_layers)."""
        b = DeviceBuffer.__new__(DeviceBuffer)
        b.ptr, b.nbytes = ptr, nbytes
        return b

    def free(self) -> None:
        if self.ptr:
            call("tstwo_free", vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.tstwo_free(vp(self.ptr))
                self.ptr = None
        except Exception:
            pass

    # -- host transfer
    def upload(self, arr: np.ndarray, offset: int = 0) -> None:
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        call("tstwo_upload", vp(self.ptr + offset), arr.ctypes.data_as(vp), arr.nbytes)

    def upload_async(self, arr: np.ndarray, offset: int = 0) -> None:
        """tstwo_upload_async: the copy runs on the library's copy stream beside the kernels; `arr` (best: page-locked — PinnedArray
        or host_register) must stay alive and unchanged until upload_wait() / sync()."""
        assert arr.flags["C_CONTIGUOUS"] and offset + arr.nbytes <= self.nbytes
        call("tstwo_upload_async", vp(self.ptr + offset), arr.ctypes.data_as(vp), arr.nbytes)

    def download(self, dtype=np.uint32, count: int | None = None, offset: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dt.itemsize
        out = np.empty(count, dtype=dt)
        call("tstwo_download", out.ctypes.data_as(vp), vp(self.ptr + offset), out.nbytes)
        return out

    def zero(self) -> None:
        call("tstwo_zero", vp(self.ptr), self.nbytes)


def download_many(pieces) -> list:
    """Several small device buffers in one round trip (tstwo_download_many): pieces = [(device pointer, n_words)];
    returns one uint32 array per piece."""
    n = len(pieces)
    if n == 0:
        return []
    srcs = (vp * n)(*[vp(int(p)) for p, _ in pieces])
    sizes = (C.c_size_t * n)(*[4 * int(w) for _, w in pieces])
    out = np.empty(sum(int(w) for _, w in pieces), dtype=np.uint32)
    call("tstwo_download_many", srcs, sizes, n, out.ctypes.data_as(vp))
    res, off = [], 0
    for _, w in pieces:
        res.append(out[off:off + int(w)])
        off += int(w)
    return res


def ptr_array(ptrs) -> C.Array:
    ptrs = list(ptrs)
    return (vp * max(len(ptrs), 1))(*ptrs)


def p4(ptrs):
    return P4(*ptrs)


def u32x(vals):
    return (C.c_uint32 * max(len(vals), 1))(*[int(v) for v in vals])


class Event:
    def __init__(self):
        ensure_init()
        e = vp()
        call("tstwo_event_create", C.byref(e))
        self.h = e

    def record(self):
        call("tstwo_event_record", self.h)
        return self

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        call("tstwo_event_elapsed_ms", self.h, stop.h, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if _lib is not None and self.h:
                _lib.tstwo_event_destroy(self.h)
        except Exception:
            pass
