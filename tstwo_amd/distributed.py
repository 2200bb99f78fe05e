"""Multi-GPU layer: one process per GPU, column sharding, and the single exchange on the path — an
all-gather of 32-byte Merkle roots.  torch.distributed is plumbing only: backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  No column data ever crosses GPUs."""
from __future__ import annotations

import numpy as np

from .backend import shard_columns  # re-export


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
