"""FRI commit phase with every layer ROW-SHARDED across the GPUs of a node (BASELINE north_star: "per-layer
FRI folds shard across the 8 GPUs ... RCCL all-gather only at the Merkle-root boundary").

Rank g of W holds the contiguous rows [g*N/W, (g+1)*N/W) of the (bit-reversed) secure column.  Output row i of a fold needs
input rows 2i, 2i+1 only, so folds need no exchange; a layer's Merkle tree is W subtrees whose roots are all-gathered
(W x 32 bytes) and combined on every rank.  All ranks run the same Blake2sChannel on the same roots, so they draw the same
alphas.  When a layer has fewer than 8 rows per rank the ranks all-gather the (tiny) layer and finish replicated.
The transcript — roots, alphas, last-layer polynomial — is bit-identical to the single-GPU FriProver.commit."""
from __future__ import annotations

import numpy as np

from .backend import SecureColumnByCoords
from .circle import CanonicCoset, Coset, LineDomain, bit_reverse_index
from .distributed import allgather_roots, combine_subtree_roots, fold_circle_into_line_rows, fold_line_rows, shard_rows
from .fri import CIRCLE_TO_LINE_FOLD_STEP, HipFriOps
from .fri_prover import FriConfig, LinePoly, line_interpolate
from .poly import LineEvaluation, TwiddleTree
from .vcs import MerkleProver

MIN_ROWS_PER_RANK = 8          # a fold shard must keep >= 4 output rows (4-aligned twiddle slices)


class ShardedFriLayer:
    """One committed layer: this rank's rows, its subtree, the W subtree roots and the combined root."""

    def __init__(self, log_size, shard, subtree, subtree_roots, root, replicated):
        self.log_size, self.shard, self.subtree = log_size, shard, subtree
        self.subtree_roots, self.root, self.replicated = subtree_roots, root, replicated


def _allgather_secure(shard: SecureColumnByCoords, group=None) -> SecureColumnByCoords:
    """Every rank's rows of a small layer, concatenated in rank order (replicated continuation)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.from_numpy(np.stack(shard.to_numpy()).astype(np.int64)).to(dev)         # (4, rows)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    full = torch.cat([o.cpu() for o in out], dim=1).numpy().astype(np.uint32)
    return SecureColumnByCoords.from_numpy([np.ascontiguousarray(full[k]) for k in range(4)])


def fri_commit_row_sharded(channel, config: FriConfig, src_shard: SecureColumnByCoords, log_size: int, rank: int, world: int,
                           twiddles: TwiddleTree, group=None):
    """Commit ONE circle column of 2^log_size rows on the canonic domain, of which this rank holds `src_shard`.
    Returns (layers: list[ShardedFriLayer], last_layer_poly: LinePoly).  layers[0] is the circle column itself."""
    n = log_size
    rows0, cnt0 = shard_rows(1 << n, world, rank)
    if src_shard.len() != cnt0:
        raise ValueError("shard length does not match its row range")
    domain = CanonicCoset(n).circleDomain()
    if not HipFriOps.can_fold_on_device(domain, twiddles):
        raise ValueError("twiddle tree mismatch")

    def commit_layer(shard: SecureColumnByCoords, lg: int, replicated: bool) -> ShardedFriLayer:
        tree = MerkleProver.commit(shard.columns)
        if replicated:
            roots, root = [tree.root()], tree.root()
        else:
            roots = allgather_roots(tree.root(), group=group)
            root = combine_subtree_roots(roots)
        channel.mix_root(root)
        return ShardedFriLayer(lg, shard, tree, roots, root, replicated)

    layers = [commit_layer(src_shard, n, False)]
    alpha = channel.draw_felt()
    # circle -> first line layer (2^(n-1) rows), still sharded
    lg = n - 1
    _, cnt = shard_rows(1 << lg, world, rank)
    cur = SecureColumnByCoords.zeros(cnt)
    fold_circle_into_line_rows(cur, src_shard, n, rank, world, alpha, twiddles)
    line_domain = LineDomain(Coset.half_odds(lg))
    replicated = False
    while (1 << lg) > config.last_layer_domain_size():
        layers.append(commit_layer(cur, lg, replicated))
        alpha = channel.draw_felt()
        if not replicated and (1 << (lg - 1)) // world < MIN_ROWS_PER_RANK // 2:
            # too small to keep sharded: every rank takes the whole layer (a few hundred bytes) and goes on alone
            cur = _allgather_secure(cur, group) if world > 1 else cur
            replicated = True
        if replicated:
            cur = HipFriOps.fold_line(LineEvaluation(line_domain, cur), alpha, twiddles).values
        else:
            cur = fold_line_rows(cur, lg, rank, world, alpha, twiddles)
        lg -= 1
        line_domain = line_domain.double()
    if not replicated and world > 1:
        cur = _allgather_secure(cur, group)
    # last layer (fri.ts:718-754), identical on every rank
    coeffs_br = line_interpolate(LineEvaluation(line_domain, cur), twiddles)
    k = len(coeffs_br).bit_length() - 1
    ordered = [coeffs_br[bit_reverse_index(i, k)] for i in range(len(coeffs_br))]
    bound = 1 << config.log_last_layer_degree_bound
    if any(c.tup() != (0, 0, 0, 0) for c in ordered[bound:]):
        raise ValueError("invalid degree")
    last = LinePoly.from_ordered_coefficients(ordered[:bound])
    channel.mix_felts(last.coeffs)
    return layers, last
