"""Multi-GPU layer: one process per GPU, column sharding, and the single exchange on the path — an
all-gather of 32-byte Merkle roots.  torch.distributed is plumbing only: backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  No column data ever crosses GPUs."""
from __future__ import annotations

import numpy as np

from .backend import shard_columns  # re-export


class HipComm:
    """The library's own RCCL communicator (include/tstwo_hip.h "multi-GPU"): what a Bun host binds.  One per process.

    `exchange_id(id_or_None) -> id`: any host-side broadcast of 128 bytes from rank 0 (a file, a socket, torch.distributed
    on gloo, an environment variable set by the launcher)."""

    def __init__(self, rank: int, world: int, exchange_id):
        import ctypes as C
        from . import _lib as L
        L.ensure_init()
        mine = None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            L.call("tstwo_comm_unique_id", buf)
            mine = bytes(buf)
        uid = exchange_id(mine)
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        L.call("tstwo_comm_init", rank, world, (C.c_uint8 * 128).from_buffer_copy(bytes(uid)))
        self.rank, self.world = rank, world

    def allgather_roots(self, layers_or_root_ptr: int) -> list:
        """All-gather of the 32 bytes at a device address (byte 0 of a tstwo_merkle_commit layers buffer = the root):
        returns the world's roots in rank order, as bytes."""
        import ctypes as C
        from . import _lib as L
        out = L.DeviceBuffer(32 * self.world)
        L.call("tstwo_allgather_roots", C.c_void_p(layers_or_root_ptr), C.c_void_p(out.ptr))
        flat = out.download(np.uint8, 32 * self.world).tobytes()
        return [flat[32 * r:32 * r + 32] for r in range(self.world)]

    def close(self):
        from . import _lib as L
        L.call("tstwo_comm_destroy")


def allgather_roots(root: bytes, group=None, device=None) -> list:
    """Every rank contributes its 32-byte root; returns the world's roots in rank order (stwo's TreeVec order,
    pcs/prover.ts:62-64,227-228 comment: each root is then mixed into the channel in order)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return [root]
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    mine = torch.tensor(np.frombuffer(root, dtype=np.uint8).copy(), dtype=torch.uint8, device=dev)
    out = torch.empty(32 * world, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    flat = out.cpu().numpy().tobytes()
    return [flat[32 * r:32 * r + 32] for r in range(world)]


def commit_sharded(columns, rank: int, world: int, commit_fn, group=None):
    """Commit this rank's shard of `columns` (list indexed by global column id; only this rank's entries are used)
    with `commit_fn(list_of_columns) -> object with .root()`, then all-gather the roots.
    Returns (local_tree, [root_0, ..., root_{world-1}])."""
    mine = [columns[i] for i in shard_columns(len(columns), world, rank)]
    tree = commit_fn(mine)
    return tree, allgather_roots(tree.root(), group=group)


# ---------------------------------------------------------------- row sharding of one big tree / FRI layer
def shard_rows(n_rows: int, world: int, rank: int) -> tuple:
    """Contiguous row shard [start, start + count) of a power-of-two layer; world must be a power of two <= n_rows / 4."""
    if world & (world - 1) or n_rows % world or (n_rows // world) % 4:
        raise ValueError("row sharding needs a power-of-two world size and shards of a multiple of 4 rows")
    per = n_rows // world
    return rank * per, per


def combine_subtree_roots(roots) -> bytes:
    """Top log2(world) levels of a row-sharded Merkle tree: rank r's tree over its contiguous leaf range is the
    subtree rooted at node r of level log2(world); the levels above hold no column values, so each node is
    H(left || right) (hashNode, vcs/blake2_merkle.ts:9-24).  Computed redundantly on every rank (world - 1 hashes)."""
    import hashlib
    level = list(roots)
    if len(level) & (len(level) - 1):
        raise ValueError("number of subtree roots must be a power of two")
    while len(level) > 1:
        level = [hashlib.blake2s(level[2 * i] + level[2 * i + 1]).digest() for i in range(len(level) // 2)]
    return level[0]


def commit_rows_sharded(shard_columns_, commit_fn, group=None):
    """Commit this rank's ROW shard of every column (each shard a power-of-two number of rows) and combine:
    returns (local_subtree, [subtree_root_0..], root) with root bit-identical to the single-GPU MerkleProver.commit
    over the whole columns.  The only exchange is the all-gather of world x 32 bytes."""
    tree = commit_fn(shard_columns_)
    roots = allgather_roots(tree.root(), group=group)
    return tree, roots, combine_subtree_roots(roots)


def fold_line_rows(shard, log_n: int, rank: int, world: int, alpha, twiddles):
    """fold_line (fri.ts:120-152) of this rank's row shard of a line layer of 2^log_n rows on a domain that is a
    doubling of the tree's root: `shard` holds input rows [2*start, 2*(start+count)); returns output rows
    [start, start+count) as a SecureColumnByCoords.  No exchange: output i needs inputs 2i, 2i+1 only."""
    from . import _lib as L
    from .backend import SecureColumnByCoords, _vp
    from .fields import as_q4
    start, count = shard_rows(1 << (log_n - 1), world, rank)
    if shard.len() != 2 * count:
        raise ValueError("shard length does not match its row range")
    out = SecureColumnByCoords.uninitialized(count)
    L.call("tstwo_fri_fold_line_rows", shard.ptrs(), log_n, start, count, _vp(twiddles.itwiddles.ptr), twiddles.log_size,
           L.u32x(as_q4(alpha)), out.ptrs())
    return out


def fold_circle_into_line_rows(dst_shard, src_shard, log_n: int, rank: int, world: int, alpha, twiddles) -> None:
    """fold_circle_into_line (fri.ts:162-192) on row shards: src_shard = rows [2*start, 2*(start+count)) of a circle
    evaluation of 2^log_n rows, dst_shard = rows [start, start+count) of the line layer, updated in place."""
    from . import _lib as L
    from .backend import _vp
    from .fields import as_q4
    start, count = shard_rows(1 << (log_n - 1), world, rank)
    if src_shard.len() != 2 * count or dst_shard.len() != count:
        raise ValueError("fold_circle_into_line: Length mismatch between src and dst after considering fold step.")
    L.call("tstwo_fri_fold_circle_into_line_rows", dst_shard.ptrs(), src_shard.ptrs(), log_n, start, count,
           _vp(twiddles.itwiddles.ptr), twiddles.log_size, L.u32x(as_q4(alpha)))


def decommit_rows_sharded(subtree, subtree_roots, shard_columns_, full_col_log_sizes, queriesPerLogSize: dict, rank: int, world: int,
                          group=None):
    """MerkleProver.decommit for a ROW-SHARDED tree (commit_rows_sharded): every rank plans the same global walk
    (vcs.decommit_requests), serves the requests that fall into its subtree / its rows with two device gathers, and the
    answers are exchanged with one all-gather of a few hundred bytes.  Levels above the subtree roots are recomputed from
    the gathered roots.  Returns (queried_values, MerkleDecommitment) identical to the single-GPU decommit."""
    import hashlib

    import torch.distributed as dist

    from .fields import M31
    from .vcs import MerkleDecommitment, _gather, decommit_requests
    log_w = world.bit_length() - 1
    max_log = max(full_col_log_sizes) if full_col_log_sizes else 0
    hash_req, queried_req, witness_req = decommit_requests(max_log, list(full_col_log_sizes), queriesPerLogSize)
    # top of the tree (levels 0..log_w) from the subtree roots, known to every rank
    top = {log_w: list(subtree_roots)}
    for lv in range(log_w - 1, -1, -1):
        top[lv] = [hashlib.blake2s(top[lv + 1][2 * i] + top[lv + 1][2 * i + 1]).digest() for i in range(1 << lv)]
    mine_h, mine_v = {}, {}
    dev_h, dev_v = [], []
    for j, (lg, node) in enumerate(hash_req):
        if lg <= log_w:
            continue                                           # served from `top` by everybody
        local_log = lg - log_w
        if node >> local_log == rank:
            dev_h.append((j, subtree.layers[local_log].ptr, node & ((1 << local_log) - 1)))
    vals_req = [("q", j, c, node) for j, (c, node) in enumerate(queried_req)] + [("w", j, c, node) for j, (c, node) in enumerate(witness_req)]
    for kind, j, c, node in vals_req:
        local_log = full_col_log_sizes[c] - log_w
        if local_log < 0:
            raise ValueError("row sharding needs every column to have at least `world` rows")
        if node >> local_log == rank:
            dev_v.append(((kind, j), shard_columns_[c].ptr, node & ((1 << local_log) - 1)))
    hw = _gather([(p, i) for _, p, i in dev_h], 8)
    for k, (j, _, _) in enumerate(dev_h):
        mine_h[j] = hw[8 * k:8 * k + 8].tobytes()
    vw = _gather([(p, i) for _, p, i in dev_v], 1)
    for k, (key, _, _) in enumerate(dev_v):
        mine_v[key] = int(vw[k])
    if world > 1 and dist.is_available() and dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine_h, mine_v), group=group)
    else:
        gathered = [(mine_h, mine_v)]
    all_h, all_v = {}, {}
    for h, v in gathered:
        all_h.update(h)
        all_v.update(v)
    hashes = [top[lg][node] if lg <= log_w else all_h[j] for j, (lg, node) in enumerate(hash_req)]
    queried = [M31(all_v[("q", j)]) for j in range(len(queried_req))]
    colwit = [M31(all_v[("w", j)]) for j in range(len(witness_req))]
    return queried, MerkleDecommitment(hashes, colwit)
