"""MerkleOps / MerkleProver.commit with Blake2s on the GPU (packages/core/src/vcs/ops.ts:16-26,
vcs/blake2_merkle.ts:9-24, vcs/prover.ts:13-30,111-113).  Layers stay resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

import hashlib
from dataclasses import dataclass, field

from . import _lib as L
from .backend import HipColumn, _vp
from .fields import M31


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


class _LazyList:
    """A list whose elements are built when they are looked at.  Sub-classes give `_n()` and `_make(i)`; the first mutation
    (a test tampering with a proof, a caller appending) turns it into an ordinary list of built elements."""
    __slots__ = ("_items",)

    def _all(self):
        if self._items is None:
            self._items = [self._make(i) for i in range(self._n())]
        return self._items

    def __len__(self): return self._n() if self._items is None else len(self._items)
    def __getitem__(self, i):
        if self._items is not None:
            return self._items[i]
        if isinstance(i, slice):
            return [self._make(k) for k in range(*i.indices(self._n()))]
        n = self._n()
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("list index out of range")
        return self._make(i)
    def __iter__(self): return iter(self._items) if self._items is not None else (self._make(i) for i in range(self._n()))
    def __setitem__(self, i, v): self._all()[i] = v
    def append(self, v): self._all().append(v)
    def pop(self, i=-1): return self._all().pop(i)
    def __eq__(self, o): return list(self) == list(o)
    def __add__(self, o): return list(self) + list(o)
    def __radd__(self, o): return list(o) + list(self)
    def __repr__(self): return f"{type(self).__name__}({list(self)!r})"
    def __deepcopy__(self, memo):
        import copy
        c = copy.copy(self)
        if c._items is not None:
            c._items = list(c._items)
        return c


class M31Values(_LazyList):
    """M31 over the u32 words that came back from the device (queried values, column witness).  Elements become M31 objects when
    they are looked at (the verifier, a serialiser), not when the proof is assembled: the 1280 queried values of a 40-query,
    32-column opening were a quarter of a millisecond of host objects otherwise."""
    __slots__ = ("words",)

    def __init__(self, words):
        self.words, self._items = words, None                  # list of ints

    def _n(self): return len(self.words)
    def _make(self, i): return M31(self.words[i])


class HashSlices(_LazyList):
    """32-byte digests over one bytes object (the hash witness as it came back from the device)."""
    __slots__ = ("raw", "n")

    def __init__(self, raw: bytes, n: int):
        self.raw, self.n, self._items = raw, n, None

    def _n(self): return self.n
    def _make(self, i): return self.raw[32 * i:32 * i + 32]


class TreeLayers(_LazyList):
    """The layers of a tree committed in one buffer (tstwo_merkle_commit layout: layer k = 2^k digests at byte 32 (2^k - 1)),
    layers[0] = [root]: a DeviceHashLayer view is built when a layer is looked at (a FRI proof of a log-22 column commits 15
    trees of 10-20 layers each; building every view up front was 0.2 ms of host objects per proof)."""
    __slots__ = ("buf", "max_log")

    def __init__(self, buf, max_log: int):
        self.buf, self.max_log, self._items = buf, max_log, None

    def _n(self): return self.max_log + 1
    def _make(self, k): return DeviceHashLayer(self.buf, 1 << k, 32 * ((1 << k) - 1))


class MerkleProver:
    """MerkleProver.commit / root (vcs/prover.ts:13-30,111-113): layers[0] = [root], all layers retained on device."""

    def __init__(self, layers: list, buf: L.DeviceBuffer, root: bytes):
        self.layers, self._buf, self._root = layers, buf, root

    @staticmethod
    def commit(columns, ops=None, sync_root: bool = True) -> "MerkleProver":
        """columns: HipColumn list of power-of-two lengths (mixed sizes allowed; order kept within a size).
        sync_root=False: nothing is read back (root() fetches the 32 bytes on first use) — for callers that keep going on the device."""
        log_sizes = []
        for c in columns:
            n = c.len()
            if n == 0 or n & (n - 1):
                raise ValueError("column length is not a power of two")
            log_sizes.append(n.bit_length() - 1)
        max_log = max(log_sizes) if columns else 0
        buf = L.DeviceBuffer(32 * ((2 << max_log) - 1))
        root = (C.c_uint8 * 32)() if sync_root else None
        L.call("tstwo_merkle_commit", L.ptr_array([c.ptr for c in columns]), L.u32x(log_sizes), len(columns), _vp(buf.ptr), root)
        layers = TreeLayers(buf, max_log)
        return MerkleProver(layers, buf, bytes(root) if sync_root else None)

    @staticmethod
    def commit_many(column_sets, sync_root: bool = True) -> list:
        """MerkleProver.commit of several trees in one launch sequence (tstwo_merkle_commit_many: a TreeVec committed together,
        pcs/prover.ts:62-64).  Same trees as commit() one by one; equally shaped trees share their launches."""
        column_sets = [list(cs) for cs in column_sets]
        reqs = (L.CommitRequest * max(len(column_sets), 1))()
        keep, bufs, max_logs = [], [], []
        for r, cols in enumerate(column_sets):
            logs = []
            for c in cols:
                n = c.len()
                if n == 0 or n & (n - 1):
                    raise ValueError("column length is not a power of two")
                logs.append(n.bit_length() - 1)
            max_log = max(logs) if cols else 0
            buf = L.DeviceBuffer(32 * ((2 << max_log) - 1))
            colp, lg = L.ptr_array([c.ptr for c in cols]), L.u32x(logs)
            keep += [colp, lg]
            reqs[r] = L.CommitRequest(colp, lg, len(cols), buf.ptr)
            bufs.append(buf)
            max_logs.append(max_log)
        roots = (C.c_uint8 * (32 * max(len(column_sets), 1)))() if sync_root else None
        L.call("tstwo_merkle_commit_many", reqs, len(column_sets), roots)
        out = []
        for r, (buf, max_log) in enumerate(zip(bufs, max_logs)):
            layers = TreeLayers(buf, max_log)
            out.append(MerkleProver(layers, buf, bytes(roots[32 * r:32 * r + 32]) if sync_root else None))
        return out

    def root(self) -> bytes:
        if self._root is None:                      # committed asynchronously (sync_root=False): fetch the 32 bytes now
            self._root = self._buf.download(np.uint8, 32).tobytes()
        return self._root

    def root_ptr(self) -> int:
        """Device address of the root (byte 0 of the layers buffer) for consumers that stay on the device."""
        return self._buf.ptr

    def decommit(self, queriesPerLogSize: dict, columns, want_queried: bool = True) -> tuple:
        """MerkleProver.decommit (vcs/prover.ts:32-109): returns (queried_values, MerkleDecommitment).
        The walk over the layers and the two device gathers run inside the library (tstwo_merkle_decommit);
        `_decommit_walk` below is the same walk on the host mirror, kept for cross-checking."""
        cols = list(columns)
        max_log = len(self.layers) - 1
        # like the reference, only the entries for layers this tree has are looked at (a commitment scheme hands every tree the
        # query positions of ALL column sizes, pcs/prover.ts Rust text :137-141)
        sets = [(lg, list(q)) for lg, q in queriesPerLogSize.items() if q and 0 <= lg <= max_log]
        total_q = sum(len(q) for _, q in sets)
        cap_v = max(1, total_q * max(1, len(cols)))
        cap_h = max(1, 2 * total_q * (max_log + 1))
        qarrs = [(C.c_uint64 * max(len(q), 1))(*q) for _, q in sets]
        qptrs = (C.POINTER(C.c_uint64) * max(len(sets), 1))(*[C.cast(a, C.POINTER(C.c_uint64)) for a in qarrs])
        nq = (C.c_size_t * max(len(sets), 1))(*[len(q) for _, q in sets])
        queried = np.empty(cap_v, dtype=np.uint32)
        colwit = np.empty(cap_v, dtype=np.uint32)
        hashes = np.empty(32 * cap_h, dtype=np.uint8)
        n_q, n_h, n_w = C.c_size_t(cap_v), C.c_size_t(cap_h), C.c_size_t(cap_v)
        L.call("tstwo_merkle_decommit", _vp(self._buf.ptr), max_log, L.ptr_array([c.ptr for c in cols]),
               L.u32x([c.len().bit_length() - 1 for c in cols]), len(cols), L.u32x([lg for lg, _ in sets]), qptrs, nq, len(sets),
               queried.ctypes.data_as(L.u32p), C.byref(n_q), hashes.ctypes.data_as(L.u8p), C.byref(n_h),
               colwit.ctypes.data_as(L.u32p), C.byref(n_w))
        hb = hashes.tobytes()
        dec = MerkleDecommitment(HashSlices(hb, n_h.value), M31Values(colwit[:n_w.value].tolist()))
        # (FRI layers already hold their queried evaluations: want_queried=False skips the queried values)
        return (M31Values(queried[:n_q.value].tolist()) if want_queried else None), dec

    @staticmethod
    def decommit_many(requests, want_queried: bool = True) -> list:
        """decommit() of several trees in ONE round trip (tstwo_merkle_decommit_many): `requests` = [(tree, queriesPerLogSize,
        columns)]; returns [(queried_values or None, MerkleDecommitment)] in request order."""
        requests = list(requests)
        if not requests:
            return []
        keep, reqs = [], (L.DecommitRequest * len(requests))()
        cap_v = cap_h = 1
        for r, (tree, qpl, columns) in enumerate(requests):
            cols = list(columns)
            max_log = len(tree.layers) - 1
            sets = [(lg, list(q)) for lg, q in qpl.items() if q and 0 <= lg <= max_log]
            total_q = sum(len(q) for _, q in sets)
            cap_v += total_q * max(1, len(cols))
            cap_h += 2 * total_q * (max_log + 1)
            qarrs = [(C.c_uint64 * max(len(q), 1))(*q) for _, q in sets]
            qptrs = (C.POINTER(C.c_uint64) * max(len(sets), 1))(*[C.cast(a, C.POINTER(C.c_uint64)) for a in qarrs])
            nq = (C.c_size_t * max(len(sets), 1))(*[len(q) for _, q in sets])
            colp = L.ptr_array([c.ptr for c in cols])
            logs = L.u32x([c.len().bit_length() - 1 for c in cols])
            qlogs = L.u32x([lg for lg, _ in sets])
            keep += [qarrs, qptrs, nq, colp, logs, qlogs]
            reqs[r] = L.DecommitRequest(tree._buf.ptr, max_log, colp, logs, len(cols), qlogs, qptrs, nq, len(sets))
        queried = np.empty(cap_v, dtype=np.uint32)
        colwit = np.empty(cap_v, dtype=np.uint32)
        hashes = np.empty(32 * cap_h, dtype=np.uint8)
        counts = (C.c_size_t * (3 * len(requests)))()
        totals = (C.c_size_t * 3)(cap_v, cap_h, cap_v)
        L.call("tstwo_merkle_decommit_many", reqs, len(requests), queried.ctypes.data_as(L.u32p), hashes.ctypes.data_as(L.u8p),
               colwit.ctypes.data_as(L.u32p), counts, totals)
        hb = hashes.tobytes()
        out, q0, h0, w0 = [], 0, 0, 0
        ql, wl = queried.tolist() if want_queried else None, colwit[:totals[2]].tolist()
        for r in range(len(requests)):
            nq_, nh_, nw_ = counts[3 * r], counts[3 * r + 1], counts[3 * r + 2]
            dec = MerkleDecommitment(HashSlices(hb[32 * h0:32 * (h0 + nh_)], nh_), M31Values(wl[w0:w0 + nw_]))
            out.append((M31Values(ql[q0:q0 + nq_]) if want_queried else None, dec))
            q0, h0, w0 = q0 + nq_, h0 + nh_, w0 + nw_
        return out

    def _decommit_walk(self, queriesPerLogSize: dict, columns) -> tuple:
        """The reference's walk (vcs/prover.ts:32-109) planned on the host (decommit_requests) + two tstwo_gather_words calls;
        kept to cross-check the in-library tstwo_merkle_decommit."""
        cols = list(columns)
        hreq, qreq, wreq = decommit_requests(len(self.layers) - 1, [c.len().bit_length() - 1 for c in cols], queriesPerLogSize)
        hashes = _gather([(self.layers[lg].ptr, node) for lg, node in hreq], 8)
        vals = _gather([(cols[c].ptr, node) for c, node in qreq + wreq], 1)
        nq = len(qreq)
        dec = MerkleDecommitment([hashes[8 * i:8 * i + 8].tobytes() for i in range(len(hreq))], [M31(int(v)) for v in vals[nq:]])
        return [M31(int(v)) for v in vals[:nq]], dec


def decommit_requests(max_log: int, col_log_sizes, queriesPerLogSize: dict) -> tuple:
    """The walk of MerkleProver.decommit (vcs/prover.ts:32-109) as pure index logic: returns
    (hash_req [(layer_log, node)], queried_req [(column_index, node)], witness_req [(column_index, node)]) in the order the
    decommitment lists them.  Columns keep the caller's order within a size (stable sort by size, like the reference)."""
    order = sorted(range(len(col_log_sizes)), key=lambda i: -col_log_sizes[i])
    col_i = 0
    hash_req, queried_req, witness_req = [], [], []
    last_nodes = []
    for log in range(max_log, -1, -1):
        layer_cols = []
        while col_i < len(order) and col_log_sizes[order[col_i]] == log:
            layer_cols.append(order[col_i])
            col_i += 1
        cur_nodes = []
        parents = _Peekable(last_nodes)
        direct = _Peekable(queriesPerLogSize.get(log) or [])
        while True:
            node = next_decommitment_node(parents, direct)
            if node is None:
                break
            if log < max_log:
                for k in (2 * node, 2 * node + 1):
                    if parents.peek() == k:
                        parents.next()
                    else:
                        hash_req.append((log + 1, k))
            reqs = [(c, node) for c in layer_cols]
            if direct.peek() == node:
                direct.next()
                queried_req += reqs
            else:
                witness_req += reqs
            cur_nodes.append(node)
        last_nodes = cur_nodes
    return hash_req, queried_req, witness_req


def _gather(reqs, words: int) -> np.ndarray:
    out = np.empty(len(reqs) * words, dtype=np.uint32)
    if reqs:
        srcs = L.ptr_array([p for p, _ in reqs])
        idx = (C.c_uint64 * len(reqs))(*[i for _, i in reqs])
        L.call("tstwo_gather_words", srcs, idx, words, len(reqs), out.ctypes.data_as(L.u32p))
    return out


class _Peekable:
    """vcs/utils.ts:8-32."""

    def __init__(self, it):
        self._it = iter(it)
        self._buf = []

    def peek(self):
        if not self._buf:
            try:
                self._buf.append(next(self._it))
            except StopIteration:
                return None
        return self._buf[0]

    def next(self):
        v = self.peek()
        if self._buf:
            self._buf.pop(0)
        return v


def next_decommitment_node(prev_queries: _Peekable, layer_queries: _Peekable):
    """vcs/utils.ts:40-57."""
    cands = []
    p = prev_queries.peek()
    if p is not None:
        cands.append(p // 2)
    q = layer_queries.peek()
    if q is not None:
        cands.append(q)
    return min(cands) if cands else None


@dataclass
class MerkleDecommitment:
    """vcs/verifier.ts:5-8."""
    hashWitness: list = field(default_factory=list)      # 32-byte digests
    columnWitness: list = field(default_factory=list)    # M31


class Blake2sMerkleHasher:
    """hashNode (vcs/blake2_merkle.ts:9-24) on the host — used by the verifier only (a handful of nodes)."""

    @staticmethod
    def hashNode(children, column_values) -> bytes:
        h = hashlib.blake2s()
        if children is not None:
            h.update(children[0])
            h.update(children[1])
        for v in column_values:
            h.update(int(v.value if isinstance(v, M31) else v).to_bytes(4, "little"))
        return h.digest()


class MerkleVerifier:
    """MerkleVerifier (vcs/verifier.ts:15-147), same error texts."""

    def __init__(self, hasher, root: bytes, column_log_sizes):
        self.hasher, self.root, self.columnLogSizes = hasher, root, list(column_log_sizes)
        self.nColumnsPerLogSize = {}
        for lg in self.columnLogSizes:
            self.nColumnsPerLogSize[lg] = self.nColumnsPerLogSize.get(lg, 0) + 1

    def verify(self, queriesPerLogSize: dict, queriedValues, decommitment: MerkleDecommitment) -> None:
        if not self.columnLogSizes:
            return
        max_log = max(self.columnLogSizes)
        qi = hi = ci = 0
        last = None
        for log in range(max_log, -1, -1):
            n_cols = self.nColumnsPerLogSize.get(log, 0)
            total = []
            prev_q = _Peekable([q for q, _ in (last or [])])
            prev_h = _Peekable(last) if last is not None else None
            layer_q = _Peekable(queriesPerLogSize.get(log) or [])
            while True:
                node = next_decommitment_node(prev_q, layer_q)
                if node is None:
                    break
                while prev_q.peek() is not None and prev_q.peek() // 2 == node:
                    prev_q.next()
                node_hashes = None
                if prev_h is not None:
                    pair = []
                    for k in (2 * node, 2 * node + 1):
                        pk = prev_h.peek()
                        if pk is not None and pk[0] == k:
                            pair.append(prev_h.next()[1])
                        else:
                            if hi >= len(decommitment.hashWitness):
                                raise ValueError("Witness is too short")
                            pair.append(decommitment.hashWitness[hi])
                            hi += 1
                    node_hashes = tuple(pair)
                from_queried = layer_q.peek() == node
                if from_queried:
                    layer_q.next()
                vals = []
                for _ in range(n_cols):
                    if from_queried:
                        if qi >= len(queriedValues):
                            raise ValueError("too few queried values")
                        vals.append(queriedValues[qi])
                        qi += 1
                    else:
                        if ci >= len(decommitment.columnWitness):
                            raise ValueError("Witness is too short")
                        vals.append(decommitment.columnWitness[ci])
                        ci += 1
                total.append((node, self.hasher.hashNode(node_hashes, vals)))
            last = total
        if hi != len(decommitment.hashWitness):
            raise ValueError("Witness is too long.")
        if qi != len(queriedValues):
            raise ValueError("too many Queried values")
        if ci != len(decommitment.columnWitness):
            raise ValueError("Witness is too long.")
        if not last:            # no query reached the root (the reference dereferences lastLayerHashes[0] and throws, verifier.ts:128)
            raise ValueError("no queried node: nothing to verify")
        if last[0][1] != self.root:
            raise ValueError("Root mismatch.")
