"""FRI verifier (host) — the consumer of FriProver.decommit's proof.

The reference's TS `FriVerifier` keeps placeholders for the layer checks (fri.ts:562-600,933-940) and a last-layer
domain one size too small (`fri.ts:854,872` subtract 1; its own prover builds the first line layer on
`Coset.half_odds(log_size - 1)`, fri.ts:687-688, i.e. `layer_bound + log_blowup`).  This follows the Rust algorithm those
placeholders stand for, with the error texts of `FriVerificationError` (fri.ts:249-257).  Everything here is a few
hundred field operations per query — host work, no device calls."""
from __future__ import annotations

import hashlib

from .circle import CanonicCoset, CircleDomain, Coset, LineDomain, bit_reverse_index
from .fields import QM31
from .fri import CIRCLE_TO_LINE_FOLD_STEP
from .fri_prover import FOLD_STEP, FriConfig, FriProof
from .queries import Queries, get_query_positions_by_log_size
from .vcs import Blake2sMerkleHasher, MerkleVerifier

SECURE_EXTENSION_DEGREE = 4


class FriVerificationError(Exception):
    InvalidNumFriLayers = "proof contains an invalid number of FRI layers"
    FirstLayerEvaluationsInvalid = "evaluations are invalid in the first layer"
    FirstLayerCommitmentInvalid = "queries do not resolve to their commitment in the first layer"
    InnerLayerCommitmentInvalid = "queries do not resolve to their commitment in inner layer"
    InnerLayerEvaluationsInvalid = "evaluations are invalid in inner layer"
    LastLayerDegreeInvalid = "degree of last layer is invalid"
    LastLayerEvaluationsInvalid = "evaluations in the last layer are invalid"


class InsufficientWitnessError(Exception):
    def __init__(self):
        super().__init__("Insufficient witness data")


class CirclePolyDegreeBound:
    """fri.ts:197-219."""
    def __init__(self, log_degree_bound: int): self.log_degree_bound = log_degree_bound
    new = staticmethod(lambda b: CirclePolyDegreeBound(b))
    def fold_to_line(self): return LinePolyDegreeBound(self.log_degree_bound - CIRCLE_TO_LINE_FOLD_STEP)
    foldToLine = fold_to_line


class LinePolyDegreeBound:
    """fri.ts:224-244."""
    def __init__(self, log_degree_bound: int): self.log_degree_bound = log_degree_bound
    new = staticmethod(lambda b: LinePolyDegreeBound(b))
    def fold(self, n_folds: int):
        return None if self.log_degree_bound < n_folds else LinePolyDegreeBound(self.log_degree_bound - n_folds)


def _ibutterfly(v0: QM31, v1: QM31, itw) -> tuple:
    return v0.add(v1), v0.sub(v1).mulM31(itw)                                  # fft.ts:25-30


class SparseEvaluation:
    """fri.ts:283-332: per queried folding coset, its 2 evaluations and the (natural) domain index it starts at."""

    def __init__(self, subset_evals, subset_domain_initial_indexes):
        if not all(len(e) == 1 << FOLD_STEP for e in subset_evals):
            raise ValueError("All subset evaluations must have length equal to 2^FOLD_STEP")
        if len(subset_evals) != len(subset_domain_initial_indexes):
            raise ValueError("Number of subset evaluations must match number of domain indexes")
        self.subset_evals, self.subset_domain_initial_indexes = subset_evals, subset_domain_initial_indexes

    def fold_line(self, alpha: QM31, source_domain: LineDomain) -> list:
        """fold_line (fri.ts:120-152) of each 2-element subset on LineDomain(Coset(index_at(initial), 1))."""
        out = []
        for (v0, v1), idx in zip(self.subset_evals, self.subset_domain_initial_indexes):
            x = source_domain.coset().index_at(idx).to_point().x
            f0, f1 = _ibutterfly(v0, v1, x.inverse())
            out.append(f0.add(alpha.mul(f1)))
        return out

    def fold_circle(self, alpha: QM31, source_domain: CircleDomain) -> list:
        """fold_circle_into_line (fri.ts:162-192) of each 2-element subset into a zero buffer."""
        out = []
        for (v0, v1), idx in zip(self.subset_evals, self.subset_domain_initial_indexes):
            p = source_domain.indexAt(idx).to_point()
            f0, f1 = _ibutterfly(v0, v1, p.y.inverse())
            out.append(alpha.mul(f1).add(f0))
        return out

    foldLine, foldCircle = fold_line, fold_circle


def compute_decommitment_positions_and_rebuild_evals(queries: Queries, query_evals, witness_evals, fold_step: int):
    """fri.ts:389-448 (witness_evals: an iterator, advanced in place)."""
    decommitment_positions, subset_evals, initials = [], [], []
    qs = list(queries.positions)
    qi = i = 0
    while i < len(qs):
        coset = qs[i] >> fold_step
        start = coset << fold_step
        end = start + (1 << fold_step)
        decommitment_positions += range(start, end)
        subset = []
        while i < len(qs) and (qs[i] >> fold_step) == coset:
            subset.append(qs[i])
            i += 1
        ev, k = [], 0
        for position in range(start, end):
            if k < len(subset) and subset[k] == position:
                ev.append(query_evals[qi])
                qi += 1
                k += 1
            else:
                try:
                    ev.append(next(witness_evals))
                except StopIteration:
                    raise InsufficientWitnessError() from None
        subset_evals.append(ev)
        initials.append(bit_reverse_index(start, queries.log_domain_size))
    return decommitment_positions, SparseEvaluation(subset_evals, initials)


computeDecommitmentPositionsAndRebuildEvals = compute_decommitment_positions_and_rebuild_evals


def accumulate_line(layer_query_evals: list, column_query_evals: list, folding_alpha: QM31) -> None:
    """fri.ts:453-462."""
    a2 = folding_alpha.mul(folding_alpha)
    for i in range(len(layer_query_evals)):
        layer_query_evals[i] = layer_query_evals[i].mul(a2).add(column_query_evals[i])


def _flatten_m31(sparse: SparseEvaluation) -> list:
    from .fields import M31
    return [M31(w) for ev in sparse.subset_evals for q in ev for w in q.to_m31_array()]


class FriFirstLayerVerifier:
    def __init__(self, column_bounds, column_commitment_domains, folding_alpha, proof):
        self.column_bounds, self.column_commitment_domains = column_bounds, column_commitment_domains
        self.folding_alpha, self.proof = folding_alpha, proof

    def verify(self, queries: Queries, query_evals_by_column) -> list:
        max_log = self.column_commitment_domains[0].logSize()
        assert queries.log_domain_size == max_log
        if len(query_evals_by_column) != len(self.column_commitment_domains):
            raise FriVerificationError(FriVerificationError.FirstLayerEvaluationsInvalid)
        witness = iter(self.proof.fri_witness)
        positions_by_log, sparse_by_column, values = {}, [], []
        for domain, evals in zip(self.column_commitment_domains, query_evals_by_column):
            cq = queries.fold(queries.log_domain_size - domain.logSize())
            try:
                pos, sparse = compute_decommitment_positions_and_rebuild_evals(cq, evals, witness, CIRCLE_TO_LINE_FOLD_STEP)
            except (InsufficientWitnessError, IndexError):
                raise FriVerificationError(FriVerificationError.FirstLayerEvaluationsInvalid) from None
            positions_by_log[domain.logSize()] = pos
            values += _flatten_m31(sparse)
            sparse_by_column.append(sparse)
        if next(witness, None) is not None:                                     # proof holds too many evaluations
            raise FriVerificationError(FriVerificationError.FirstLayerEvaluationsInvalid)
        mv = MerkleVerifier(Blake2sMerkleHasher, self.proof.commitment,
                            [d.logSize() for d in self.column_commitment_domains for _ in range(SECURE_EXTENSION_DEGREE)])
        try:
            mv.verify(positions_by_log, values, self.proof.decommitment)
        except ValueError as e:
            raise FriVerificationError(f"{FriVerificationError.FirstLayerCommitmentInvalid}: {e}") from None
        return sparse_by_column


class FriInnerLayerVerifier:
    def __init__(self, degree_bound, domain: LineDomain, folding_alpha, layer_index: int, proof):
        self.degree_bound, self.domain, self.folding_alpha = degree_bound, domain, folding_alpha
        self.layer_index, self.proof = layer_index, proof

    def verify_and_fold(self, queries: Queries, evals_at_queries) -> tuple:
        assert queries.log_domain_size == self.domain.logSize()
        witness = iter(self.proof.fri_witness)
        try:
            pos, sparse = compute_decommitment_positions_and_rebuild_evals(queries, evals_at_queries, witness, FOLD_STEP)
        except (InsufficientWitnessError, IndexError):
            raise FriVerificationError(f"{FriVerificationError.InnerLayerEvaluationsInvalid} {self.layer_index}") from None
        if next(witness, None) is not None:
            raise FriVerificationError(f"{FriVerificationError.InnerLayerEvaluationsInvalid} {self.layer_index}")
        mv = MerkleVerifier(Blake2sMerkleHasher, self.proof.commitment, [self.domain.logSize()] * SECURE_EXTENSION_DEGREE)
        try:
            mv.verify({self.domain.logSize(): pos}, _flatten_m31(sparse), self.proof.decommitment)
        except ValueError as e:
            raise FriVerificationError(f"{FriVerificationError.InnerLayerCommitmentInvalid} {self.layer_index}: {e}") from None
        return queries.fold(FOLD_STEP), sparse.fold_line(self.folding_alpha, self.domain)

    verifyAndFold = verify_and_fold


class FriVerifier:
    """FriVerifier (fri.ts:791-979), errors raised as FriVerificationError instead of returned."""

    def __init__(self, config, first_layer, inner_layers, last_layer_domain, last_layer_poly):
        self.config, self.first_layer, self.inner_layers = config, first_layer, inner_layers
        self.last_layer_domain, self.last_layer_poly = last_layer_domain, last_layer_poly
        self.queries = None

    @staticmethod
    def commit(channel, config: FriConfig, proof: FriProof, column_bounds) -> "FriVerifier":
        for a, b in zip(column_bounds, column_bounds[1:]):
            if a.log_degree_bound < b.log_degree_bound:
                raise FriVerificationError(FriVerificationError.InvalidNumFriLayers)
        channel.mix_root(proof.first_layer.commitment)
        max_bound = column_bounds[0]
        domains = [CanonicCoset(b.log_degree_bound + config.log_blowup_factor).circle_domain() for b in column_bounds]
        first = FriFirstLayerVerifier(column_bounds, domains, channel.draw_felt(), proof.first_layer)
        inner = []
        layer_bound = max_bound.fold_to_line()
        layer_domain = LineDomain(Coset.half_odds(layer_bound.log_degree_bound + config.log_blowup_factor))
        for i, lp in enumerate(proof.inner_layers):
            channel.mix_root(lp.commitment)
            inner.append(FriInnerLayerVerifier(layer_bound, layer_domain, channel.draw_felt(), i, lp))
            layer_bound = layer_bound.fold(FOLD_STEP)
            if layer_bound is None:
                raise FriVerificationError(FriVerificationError.InvalidNumFriLayers)
            layer_domain = layer_domain.double()
        if layer_bound.log_degree_bound != config.log_last_layer_degree_bound:
            raise FriVerificationError(FriVerificationError.InvalidNumFriLayers)
        if proof.last_layer_poly.len() > (1 << config.log_last_layer_degree_bound):
            raise FriVerificationError(FriVerificationError.LastLayerDegreeInvalid)
        channel.mix_felts(proof.last_layer_poly.coeffs)
        return FriVerifier(config, first, inner, layer_domain, proof.last_layer_poly)

    def sample_query_positions(self, channel) -> dict:
        """fri.ts:969-978."""
        log_sizes = {d.logSize() for d in self.first_layer.column_commitment_domains}
        self.queries = Queries.generate(channel, max(log_sizes), self.config.n_queries)
        return get_query_positions_by_log_size(self.queries, log_sizes)

    sampleQueryPositions = sample_query_positions

    def decommit(self, first_layer_query_evals) -> None:
        if self.queries is None:
            raise RuntimeError("queries not sampled")
        self.decommit_on_queries(self.queries, first_layer_query_evals)

    def decommit_on_queries(self, queries: Queries, first_layer_query_evals) -> None:
        expected = self.first_layer.column_commitment_domains[0].logSize()
        if queries.log_domain_size != expected:
            raise ValueError(f"Domain size mismatch: expected {expected}, got {queries.log_domain_size}")
        sparse = self.first_layer.verify(queries, first_layer_query_evals)
        lq, le = self._decommit_inner_layers(queries.fold(CIRCLE_TO_LINE_FOLD_STEP), sparse)
        self._decommit_last_layer(lq, le)

    decommitOnQueries = decommit_on_queries

    def _decommit_inner_layers(self, queries: Queries, first_layer_sparse_evals) -> tuple:
        layer_queries = queries
        layer_evals = [QM31.zero() for _ in range(len(queries))]
        sparse_it = iter(first_layer_sparse_evals)
        cols = list(zip(self.first_layer.column_bounds, self.first_layer.column_commitment_domains))
        ci = 0
        prev_alpha = self.first_layer.folding_alpha
        for layer in self.inner_layers:
            # circle columns committed in the first layer whose folded size is this layer's: fold with the previous alpha
            while ci < len(cols) and cols[ci][0].fold_to_line().log_degree_bound == layer.degree_bound.log_degree_bound:
                folded = next(sparse_it).fold_circle(prev_alpha, cols[ci][1])
                accumulate_line(layer_evals, folded, prev_alpha)
                ci += 1
            layer_queries, layer_evals = layer.verify_and_fold(layer_queries, layer_evals)
            prev_alpha = layer.folding_alpha
        if not self.inner_layers:
            # no inner layer: the single circle column folds straight into the last layer
            while ci < len(cols):
                folded = next(sparse_it).fold_circle(prev_alpha, cols[ci][1])
                accumulate_line(layer_evals, folded, prev_alpha)
                ci += 1
        assert ci == len(cols) and next(sparse_it, None) is None
        return layer_queries, layer_evals

    def _decommit_last_layer(self, queries: Queries, query_evals) -> None:
        d = self.last_layer_domain
        for q, ev in zip(queries.positions, query_evals):
            x = d.at(bit_reverse_index(q, d.logSize()))
            if not ev.equals(self.last_layer_poly.eval_at_point(QM31.from_(x))):
                raise FriVerificationError(FriVerificationError.LastLayerEvaluationsInvalid)
