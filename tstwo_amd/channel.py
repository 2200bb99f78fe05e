"""Blake2sChannel — host-side mirror of the reference's Fiat-Shamir transcript (packages/core/src/channel/blake2.ts:25-224,
vcs/blake2_merkle.ts:28-31).  O(1) hashes per protocol round: it stays on the host and only *receives* Merkle roots from
the GPU path.  hashlib.blake2s = BLAKE2s-256 unkeyed (what @noble/hashes computes for the reference)."""
from __future__ import annotations

import hashlib

from .fields import M31, P, QM31

BLAKE_BYTES_PER_HASH = 32
FELTS_PER_HASH = 8
SECURE_EXTENSION_DEGREE = 4


class Blake2sChannel:
    """ts_compat: the TS port keeps the 4 unused base felts of a draw_felt() in a queue that survives later mix_*() calls
    (blake2.ts:177-184), so every second challenge does not depend on what was mixed since — Rust draws 8 fresh base felts
    per draw_felt and drops 4.  Default = Rust (sound Fiat-Shamir); ts_compat=True reproduces the TS queue."""

    def __init__(self, ts_compat=None):
        self._digest = bytes(32)              # Blake2sHash default: all zeros (blake2.ts:42-49)
        self.n_challenges = 0
        self.n_sent = 0
        from .semantics import ts_compat as _resolve
        self.ts_compat = _resolve(ts_compat)        # None -> tstwo_amd.set_semantics() (default "rust")
        self._base_queue = []

    create = classmethod(lambda cls, ts_compat=None: cls(ts_compat))

    def digest(self) -> bytes:
        return self._digest

    def _update_digest(self, d: bytes) -> None:   # blake2.ts:76-79: inc_challenges resets n_sent
        self._digest = d
        self.n_challenges += 1
        self.n_sent = 0

    def clone(self) -> "Blake2sChannel":
        c = Blake2sChannel(self.ts_compat)
        c._digest, c.n_challenges, c.n_sent, c._base_queue = self._digest, self.n_challenges, self.n_sent, list(self._base_queue)
        return c

    def trailing_zeros(self) -> int:              # blake2.ts:96-111: first 16 bytes as a little-endian u128
        v = int.from_bytes(self._digest[:16], "little")
        return 128 if v == 0 else (v & -v).bit_length() - 1

    # ---- mixing
    def mix_root(self, root: bytes) -> None:       # Blake2sMerkleChannel.mix_root (vcs/blake2_merkle.ts:28-31)
        self._update_digest(hashlib.blake2s(self._digest + root).digest())

    def mix_felts(self, felts, _le_bytes: bytes | None = None) -> None:            # blake2.ts:113-118 (QM31.into_slice: 4 LE u32 each)
        h = hashlib.blake2s(self._digest)
        if _le_bytes is not None:                  # the caller already holds the felts' into_slice bytes (16 per felt)
            assert len(_le_bytes) == 16 * len(felts)
            h.update(_le_bytes)
        else:
            for f in felts:
                for v in f.tup():
                    h.update(int(v).to_bytes(4, "little"))
        self._update_digest(h.digest())

    def mix_u32s(self, data) -> None:              # blake2.ts:120-136
        h = hashlib.blake2s(self._digest)
        for w in data:
            if not (0 <= int(w) < 2**32):
                raise TypeError(f"Invalid u32 value: {w}")
            h.update(int(w).to_bytes(4, "little"))
        self._update_digest(h.digest())

    def mix_u64(self, value: int) -> None:         # blake2.ts:138-148
        if not (0 <= int(value) < 2**64):
            raise TypeError(f"Invalid u64 value: {value}")
        self.mix_u32s([int(value) & 0xFFFFFFFF, int(value) >> 32])

    # ---- drawing
    def draw_random_bytes(self) -> bytes:          # blake2.ts:211-223: H(digest || LE32(n_sent) padded to 32 bytes)
        counter = self.n_sent.to_bytes(4, "little") + bytes(BLAKE_BYTES_PER_HASH - 4)
        self.n_sent += 1
        return hashlib.blake2s(self._digest + counter).digest()

    def _draw_base_felts(self):                    # blake2.ts:158-175: retry until all 8 words < 2P
        while True:
            b = self.draw_random_bytes()
            u32s = [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(FELTS_PER_HASH)]
            if all(x < 2 * P for x in u32s):
                return [M31.reduce(x) for x in u32s]

    def draw_felt(self) -> QM31:                   # Rust Blake2sChannel::draw_felt; TS variant blake2.ts:177-184
        if not self.ts_compat:
            return QM31.from_u32_unchecked(*[m.value for m in self._draw_base_felts()[:SECURE_EXTENSION_DEGREE]])
        while len(self._base_queue) < SECURE_EXTENSION_DEGREE:
            self._base_queue += self._draw_base_felts()
        a = self._base_queue[:4]
        del self._base_queue[:4]
        return QM31.from_u32_unchecked(*[m.value for m in a])

    def draw_felts(self, n: int):                  # blake2.ts:186-208 (= Rust): a queue local to the call, leftovers dropped
        out, queue = [], []
        for _ in range(n):
            while len(queue) < SECURE_EXTENSION_DEGREE:
                queue += self._draw_base_felts()
            out.append(QM31.from_u32_unchecked(*[m.value for m in queue[:4]]))
            del queue[:4]
        return out


class DeviceChannel:
    """The channel's state (digest, n_challenges, n_sent) mirrored in 40 bytes of device memory, so that a launch sequence
    (FRI commit: tree -> mix_root -> draw_felt -> fold -> tree ...) never waits for the host.  Rust draw semantics only."""

    def __init__(self, host: Blake2sChannel, buf=None):
        from . import _lib as L
        self.buf = buf if buf is not None else L.DeviceBuffer(64)
        self.load(host)

    def load(self, host: Blake2sChannel) -> None:
        """(Re)start from the host channel's state (one 64-byte upload)."""
        import numpy as np
        if host.ts_compat:
            raise ValueError("the device channel implements the Rust draw semantics only")
        self.host = host
        st = np.zeros(16, dtype=np.uint32)
        st[:8] = np.frombuffer(host.digest(), dtype="<u4")
        st[8], st[9] = host.n_challenges, host.n_sent
        self.buf.upload(st)

    def mix_root_draw_felt(self, root_ptr: int | None, felt_ptr: int | None) -> None:
        """mix_root of the 32 bytes at device address root_ptr, then draw_felt into the 4 words at felt_ptr (either may be None)."""
        import ctypes as C

        from . import _lib as L
        L.call("tstwo_channel_mix_root_draw_felt", C.c_void_p(self.buf.ptr), C.c_void_p(root_ptr or 0), C.c_void_p(felt_ptr or 0))

    def sync_to_host(self, st=None) -> Blake2sChannel:
        """The host channel continues from the device state (st: the 10 state words when the caller already fetched them)."""
        if st is None:
            st = self.buf.download(count=10)
        self.host._digest = st[:8].astype("<u4").tobytes()
        self.host.n_challenges, self.host.n_sent = int(st[8]), int(st[9])
        return self.host


class HipGrindOps:
    """GrindOps (backend/cpu/grind.ts:31-42): the nonce search runs on the GPU (embarrassingly parallel Blake2s), the
    sequential reference loop returns the same first nonce."""

    @staticmethod
    def grind(channel: Blake2sChannel, pow_bits: int) -> int:
        import ctypes as C

        import numpy as np

        from . import _lib as L
        L.ensure_init()
        d = np.frombuffer(channel.digest(), dtype=np.uint8).copy()
        out = C.c_uint64(0)
        L.call("tstwo_grind_blake2s", d.ctypes.data_as(L.u8p), pow_bits, 0, C.byref(out))
        return out.value


def grind(channel: Blake2sChannel, pow_bits: int) -> int:
    return HipGrindOps.grind(channel, pow_bits)
