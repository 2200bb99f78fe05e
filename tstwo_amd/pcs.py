"""Commitment-tree prover over HipBackend: the caller on the other side of the hot path.

The reference only carries the Rust text of this layer as a comment (packages/core/src/pcs/prover.ts:26-252);
this follows it: TreeBuilder.extend_evals -> interpolate_columns; CommitmentTreeProver.new ->
evaluate_polynomials(log_blowup_factor) -> MerkleProver.commit -> MC::mix_root.  Everything between interpolation and
the 32-byte root stays in HBM."""
from __future__ import annotations

from .circle import CanonicCoset
from .poly import HipCirclePoly, TwiddleTree, evaluate_polynomials, interpolate_columns
from .vcs import MerkleProver


class CommitmentTreeProver:
    """pcs/prover.ts:209-252 (Rust comment)."""

    def __init__(self, polynomials, evaluations, commitment: MerkleProver):
        self.polynomials, self.evaluations, self.commitment = polynomials, evaluations, commitment

    @staticmethod
    def new(polynomials, log_blowup_factor: int, channel, twiddles: TwiddleTree) -> "CommitmentTreeProver":
        # "Extension": each poly is evaluated on the canonic domain of log size (poly log size + blowup); polys of one
        # size share a batched launch sequence (PolyOps.evaluatePolynomials)
        by_size = {}
        for i, p in enumerate(polynomials):
            by_size.setdefault(p.logSize(), []).append(i)
        evaluations = [None] * len(polynomials)
        for log, idxs in by_size.items():
            domain = CanonicCoset(log + log_blowup_factor).circleDomain()
            for i, ev in zip(idxs, evaluate_polynomials([polynomials[i] for i in idxs], domain, twiddles)):
                evaluations[i] = ev
        # "Merkle"
        tree = MerkleProver.commit([ev.values for ev in evaluations])
        channel.mix_root(tree.root())
        return CommitmentTreeProver(list(polynomials), evaluations, tree)

    def decommit(self, queries: dict):
        return self.commitment.decommit(queries, [ev.values for ev in self.evaluations])


class TreeBuilder:
    """pcs/prover.ts:170-207 (Rust comment)."""

    def __init__(self, scheme: "CommitmentSchemeProver"):
        self.scheme, self.polys = scheme, []

    def extend_evals(self, columns) -> tuple:
        start = len(self.polys)
        self.polys += interpolate_columns(list(columns), self.scheme.twiddles)     # "Interpolation for commitment"
        return (len(self.scheme.trees), start, len(self.polys))

    def extend_polys(self, polys) -> tuple:
        start = len(self.polys)
        self.polys += list(polys)
        return (len(self.scheme.trees), start, len(self.polys))

    def commit(self, channel) -> None:
        self.scheme.commit(self.polys, channel)


class CommitmentSchemeProver:
    """pcs/prover.ts:26-80 (Rust comment): a list of commitment trees sharing one twiddle tree."""

    def __init__(self, log_blowup_factor: int, twiddles: TwiddleTree):
        self.log_blowup_factor, self.twiddles, self.trees = log_blowup_factor, twiddles, []

    def tree_builder(self) -> TreeBuilder:
        return TreeBuilder(self)

    def commit(self, polynomials, channel) -> None:
        self.trees.append(CommitmentTreeProver.new(polynomials, self.log_blowup_factor, channel, self.twiddles))

    def roots(self) -> list:
        return [t.commitment.root() for t in self.trees]
