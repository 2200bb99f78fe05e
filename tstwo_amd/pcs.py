"""Commitment-tree prover over HipBackend: the caller on the other side of the hot path.

The reference only carries the Rust text of this layer as a comment (packages/core/src/pcs/prover.ts:26-252);
this follows it: TreeBuilder.extend_evals -> interpolate_columns; CommitmentTreeProver.new ->
evaluate_polynomials(log_blowup_factor) -> MerkleProver.commit -> MC::mix_root.  Everything between interpolation and
the 32-byte root stays in HBM."""
from __future__ import annotations

from dataclasses import dataclass, field

from .channel import grind
from .circle import CanonicCoset, CirclePoint
from .fields import QM31
from .fri_prover import FriConfig, FriProof, FriProver
from .poly import HipCirclePoly, TwiddleTree, evaluate_polynomials, interpolate_columns
from .quotients import ColumnSampleBatch, accumulateQuotients
from .vcs import MerkleProver


@dataclass
class PcsConfig:
    """pcs/index.ts (Rust mod.rs text): proof-of-work bits + FRI parameters; Default = (5, FriConfig(0, 1, 3))."""
    pow_bits: int = 5
    fri_config: FriConfig = field(default_factory=lambda: FriConfig(0, 1, 3))

    def security_bits(self) -> int:
        return self.pow_bits + self.fri_config.security_bits()

    def mix_into(self, channel) -> None:
        channel.mix_u64(self.pow_bits)
        self.fri_config.mixInto(channel)


@dataclass
class PointSample:
    """pcs/quotients.ts (Rust text :77-80)."""
    point: CirclePoint
    value: QM31


def column_sample_batches(samples) -> list:
    """ColumnSampleBatch::new_vec (pcs/quotients.ts Rust text :48-75): group the per-column samples by point, keeping
    first-seen order (IndexMap)."""
    grouped = {}
    for column_index, col_samples in enumerate(samples):
        for s in col_samples:
            key = (s.point.x.tup(), s.point.y.tup())
            grouped.setdefault(key, (s.point, []))[1].append((column_index, s.value))
    return [ColumnSampleBatch(point, cv) for point, cv in grouped.values()]


def compute_fri_quotients(columns, samples, random_coeff: QM31, log_blowup_factor: int) -> list:
    """compute_fri_quotients (pcs/quotients.ts Rust text :82-109): columns grouped by log size (descending, stable), one
    accumulate_quotients launch per size on the canonic domain of that size."""
    order = sorted(range(len(columns)), key=lambda i: -columns[i].domain.logSize())
    out, i = [], 0
    while i < len(order):
        log_size = columns[order[i]].domain.logSize()
        grp = []
        while i < len(order) and columns[order[i]].domain.logSize() == log_size:
            grp.append(order[i])
            i += 1
        domain = CanonicCoset(log_size).circleDomain()
        batches = column_sample_batches([samples[j] for j in grp])
        out.append(accumulateQuotients(domain, [columns[j] for j in grp], random_coeff, batches, log_blowup_factor))
    return out


@dataclass
class CommitmentSchemeProof:
    """pcs/prover.ts Rust text :158-167.  TreeVec = list indexed by tree; ColumnVec = list indexed by column."""
    config: PcsConfig
    commitments: list
    sampled_values: list
    decommitments: list
    queried_values: list
    proof_of_work: int
    fri_proof: FriProof


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

    @staticmethod
    def new_many(polynomial_sets, log_blowup_factor: int, channel, twiddles: TwiddleTree) -> list:
        """Several trees committed in ONE protocol phase (stwo's TreeVec: pcs/prover.ts:62-64 pushes a CommitmentTreeProver per
        tree, each mixing its root, :227-228) — the same transcript as new() tree by tree, because nothing is drawn between the
        commits of a phase: the roots are mixed in tree order once they exist.  Every polynomial of every tree goes through one
        batched evaluation per size, and the trees through one tstwo_merkle_commit_many launch sequence (equally shaped trees
        share their launches: BASELINE config 5's 8 trees of 32 columns)."""
        polynomial_sets = [list(ps) for ps in polynomial_sets]
        flat = [p for ps in polynomial_sets for p in ps]
        by_size = {}
        for i, p in enumerate(flat):
            by_size.setdefault(p.logSize(), []).append(i)
        evaluations = [None] * len(flat)
        for log, idxs in by_size.items():
            domain = CanonicCoset(log + log_blowup_factor).circleDomain()
            for i, ev in zip(idxs, evaluate_polynomials([flat[i] for i in idxs], domain, twiddles)):
                evaluations[i] = ev
        per_tree, k = [], 0
        for ps in polynomial_sets:
            per_tree.append(evaluations[k:k + len(ps)])
            k += len(ps)
        trees = MerkleProver.commit_many([[ev.values for ev in evs] for evs in per_tree])
        for t in trees:
            channel.mix_root(t.root())
        return [CommitmentTreeProver(ps, evs, t) for ps, evs, t in zip(polynomial_sets, per_tree, trees)]

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

    def __init__(self, config, twiddles: TwiddleTree):
        """config: a PcsConfig (as in Rust) or, for commitment-only use, just the log blowup factor."""
        if isinstance(config, int):
            config = PcsConfig(fri_config=FriConfig(0, config, 3))
        self.config, self.twiddles, self.trees = config, twiddles, []
        self.log_blowup_factor = config.fri_config.log_blowup_factor

    def tree_builder(self) -> TreeBuilder:
        return TreeBuilder(self)

    def commit(self, polynomials, channel) -> None:
        self.trees.append(CommitmentTreeProver.new(polynomials, self.log_blowup_factor, channel, self.twiddles))

    def commit_many(self, polynomial_sets, channel) -> None:
        """The trees of one phase committed together (CommitmentTreeProver.new_many): same trees, roots and transcript as one
        commit() per set."""
        self.trees += CommitmentTreeProver.new_many(polynomial_sets, self.log_blowup_factor, channel, self.twiddles)

    def roots(self) -> list:
        return [t.commitment.root() for t in self.trees]

    def polynomials(self) -> list:
        return [list(t.polynomials) for t in self.trees]

    def evaluations(self) -> list:
        return [list(t.evaluations) for t in self.trees]

    def prove_values(self, sampled_points, channel) -> CommitmentSchemeProof:
        """prove_values (pcs/prover.ts Rust text :82-155).  sampled_points[tree][column] = list of CirclePoint<QM31>.
        Out-of-domain evaluation, quotients, FRI commit, grinding and every decommitment read device-resident data;
        only sampled values, roots, witnesses and queried values reach the host."""
        # "Evaluate columns out of domain"
        # (all polynomials of one size sampled at the same point share one batched launch sequence)
        groups = {}
        for ti, (tree, tree_pts) in enumerate(zip(self.trees, sampled_points)):
            if len(tree_pts) != len(tree.polynomials):
                raise ValueError("sampled_points does not match the committed columns")
            for ci, (poly, pts) in enumerate(zip(tree.polynomials, tree_pts)):
                for pi, pt in enumerate(pts):
                    key = (poly.logSize(), pt.x.tup(), pt.y.tup())
                    groups.setdefault(key, (pt, []))[1].append((ti, ci, pi, poly))
        values = {}
        for pt, members in groups.values():
            for (ti, ci, pi, _), v in zip(members, HipCirclePoly.eval_at_point_batch([m[3] for m in members], pt)):
                values[(ti, ci, pi)] = v
        samples = [[[PointSample(pt, values[(ti, ci, pi)]) for pi, pt in enumerate(pts)] for ci, pts in enumerate(tree_pts)]
                   for ti, tree_pts in enumerate(sampled_points)]
        sampled_values = [[[s.value for s in col] for col in tree] for tree in samples]
        channel.mix_felts([v for tree in sampled_values for col in tree for v in col])
        # OODS quotients over every tree's evaluations, flattened
        columns = [ev for t in self.trees for ev in t.evaluations]
        flat_samples = [col for tree in samples for col in tree]
        quotients = compute_fri_quotients(columns, flat_samples, channel.draw_felt(), self.log_blowup_factor)
        # FRI commitment phase on the quotients
        fri_prover = FriProver.commit(channel, self.config.fri_config, quotients, self.twiddles)
        # proof of work
        proof_of_work = grind(channel, self.config.pow_bits)
        channel.mix_u64(proof_of_work)
        # FRI decommitment phase, then the trace trees on the same queries
        fri_proof, query_positions = fri_prover.decommit(channel)
        results = MerkleProver.decommit_many([(t.commitment, query_positions, [ev.values for ev in t.evaluations]) for t in self.trees])
        return CommitmentSchemeProof(self.config, self.roots(), sampled_values, [d for _, d in results],
                                     [v for v, _ in results], proof_of_work, fri_proof)
