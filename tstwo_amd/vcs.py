"""MerkleOps / MerkleProver.commit with Blake2s on the GPU (packages/core/src/vcs/ops.ts:16-26,
vcs/blake2_merkle.ts:9-24, vcs/prover.ts:13-30,111-113).  Layers stay resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .backend import HipColumn, _vp


class DeviceHashLayer:
    """A layer of 32-byte Blake2s digests in device memory (the GPU twin of Blake2sHash[])."""

    def __init__(self, buf: L.DeviceBuffer, n: int, offset: int = 0):
        self.buf, self.n, self.offset = buf, n, offset

    @property
    def ptr(self) -> int:
        return self.buf.ptr + self.offset

    def __len__(self): return self.n
    def to_numpy(self) -> np.ndarray: return self.buf.download(np.uint8, 32 * self.n, self.offset).reshape(self.n, 32)
    def toCpu(self) -> list: return [bytes(r) for r in self.to_numpy()]
    def at(self, i: int) -> bytes: return bytes(self.buf.download(np.uint8, 32, self.offset + 32 * i))


class HipMerkleOps:
    """MerkleOps<Blake2sHash>.commitOnLayer (vcs/ops.ts:21-25) following hashNode semantics (children AND the
    layer's column values in one message — SURVEY.md App. B-2)."""

    @staticmethod
    def commitOnLayer(logSize: int, prevLayer: DeviceHashLayer | None, columns) -> DeviceHashLayer:
        n = 1 << logSize
        for c in columns:
            if c.len() != n:
                raise ValueError("column length does not match the layer size")
        if prevLayer is not None and len(prevLayer) != 2 * n:
            raise ValueError("previous layer must have twice the nodes")
        out = DeviceHashLayer(L.DeviceBuffer(32 * n), n)
        L.call("tstwo_merkle_commit_layer", logSize, _vp(prevLayer.ptr if prevLayer is not None else 0),
               L.ptr_array([c.ptr for c in columns]), len(columns), _vp(out.ptr))
        return out


class MerkleProver:
    """MerkleProver.commit / root (vcs/prover.ts:13-30,111-113): layers[0] = [root], all layers retained on device."""

    def __init__(self, layers: list, buf: L.DeviceBuffer, root: bytes):
        self.layers, self._buf, self._root = layers, buf, root

    @staticmethod
    def commit(columns, ops=None) -> "MerkleProver":
        """columns: HipColumn list of power-of-two lengths (mixed sizes allowed; order kept within a size)."""
        log_sizes = []
        for c in columns:
            n = c.len()
            if n == 0 or n & (n - 1):
                raise ValueError("column length is not a power of two")
            log_sizes.append(n.bit_length() - 1)
        max_log = max(log_sizes) if columns else 0
        buf = L.DeviceBuffer(32 * ((2 << max_log) - 1))
        root = (C.c_uint8 * 32)()
        L.call("tstwo_merkle_commit", L.ptr_array([c.ptr for c in columns]), L.u32x(log_sizes), len(columns), _vp(buf.ptr), root)
        layers = [DeviceHashLayer(buf, 1 << k, 32 * ((1 << k) - 1)) for k in range(max_log + 1)]
        return MerkleProver(layers, buf, bytes(root))

    def root(self) -> bytes:
        return self._root
