"""FRI commit phase on the GPU — the caller on top of FriOps + MerkleOps.

Mirrors `FriProver.commit` (packages/core/src/fri.ts:644-754) with the wiring the Rust stwo prover has and the TS
port still mocks (`fri.ts:497,530,693,704`): every layer is committed with a real Blake2s Merkle tree over its 4
coordinate columns (device resident), the root is mixed into the channel, and the folding alpha is drawn from it.
Layers never leave HBM between fold and commit; only 32-byte roots and the (tiny) last layer reach the host."""
from __future__ import annotations

from .backend import SecureColumnByCoords
from .circle import Coset, LineDomain, bit_reverse_index
from .fields import M31, QM31
from .fri import CIRCLE_TO_LINE_FOLD_STEP, HipFriOps
from .poly import LineEvaluation, SecureEvaluation, TwiddleTree
from .vcs import MerkleProver


class FriConfig:
    """fri.ts:28-88."""

    def __init__(self, log_last_layer_degree_bound: int, log_blowup_factor: int, n_queries: int):
        if not (0 <= log_last_layer_degree_bound <= 10):
            raise ValueError("log_last_layer_degree_bound must be between 0 and 10")
        if not (1 <= log_blowup_factor <= 16):
            raise ValueError("log_blowup_factor must be between 1 and 16")
        self.log_last_layer_degree_bound = log_last_layer_degree_bound
        self.log_blowup_factor = log_blowup_factor
        self.n_queries = n_queries

    def last_layer_domain_size(self) -> int:
        return 1 << (self.log_last_layer_degree_bound + self.log_blowup_factor)

    def security_bits(self) -> int:
        return self.log_blowup_factor * self.n_queries

    def mixInto(self, channel) -> None:
        channel.mix_u64(self.log_blowup_factor)
        channel.mix_u64(self.n_queries)
        channel.mix_u64(self.log_last_layer_degree_bound)


def line_interpolate(evaluation: LineEvaluation, twiddles: TwiddleTree | None = None) -> list:
    """LineEvaluation.interpolate + lineIfft (poly/line.ts:312-390) on the host: the last FRI layer has at most
    2^(log_last_layer_degree_bound + log_blowup_factor) elements.  Returns bit-reversed-order coefficients.
    The x^-1 of each level are read from the tail of the inverse twiddle tree when the domain is a doubling of its
    root (level of log size k = 2^(k-1) entries, bit-reversed); otherwise they are computed per element like the reference."""
    vals = evaluation.values.to_vec()
    n = len(vals)
    log_n = n.bit_length() - 1
    vals = [vals[bit_reverse_index(i, log_n)] for i in range(n)]
    domain = evaluation.domain()
    tail = None
    if twiddles is not None and log_n >= 1 and domain.coset().is_doubling_of(twiddles.rootCoset):
        L_ = twiddles.itwiddles.len()
        tail = twiddles.itwiddles.buf.download(count=n, offset=4 * (L_ - n))        # last n entries of the tree
    while domain.size() > 1:
        size, half = domain.size(), domain.size() // 2
        k = domain.logSize()
        if tail is not None:
            seg = tail[n - size:n - half]                                            # tree[L - 2^k : L - 2^(k-1)]
            inv = [M31(int(seg[bit_reverse_index(i, k - 1)])) for i in range(half)]
        else:
            inv = [domain.at(i).inverse() for i in range(half)]
        for start in range(0, n, size):
            for i in range(half):
                a, b = vals[start + i], vals[start + i + half]
                vals[start + i], vals[start + i + half] = a.add(b), a.sub(b).mulM31(inv[i])    # ibutterfly (fft.ts:25-30)
        domain = domain.double()
    len_inv = M31.from_(n).inverse()
    return [v.mulM31(len_inv) for v in vals]


class FriLayer:
    def __init__(self, evaluation, merkle_tree: MerkleProver):
        self.evaluation, self.merkle_tree = evaluation, merkle_tree


class FriProver:
    def __init__(self, config, first_layer, inner_layers, last_layer_coeffs):
        self.config, self.first_layer, self.inner_layers, self.last_layer_poly = config, first_layer, inner_layers, last_layer_coeffs

    @staticmethod
    def commit(channel, config: FriConfig, columns, twiddles: TwiddleTree) -> "FriProver":
        """columns: SecureEvaluation list, canonic domains, strictly decreasing sizes (fri.ts:644-674)."""
        if not columns:
            raise ValueError("no columns")
        if not all(c.domain.isCanonic() for c in columns):
            raise ValueError("not canonic")
        for a, b in zip(columns, columns[1:]):
            if a.domain.size() <= b.domain.size():
                raise ValueError("column sizes not decreasing")
        # first layer: one tree over every column's coordinate columns (Rust FriFirstLayerProver::new), root -> channel
        coord_cols = [cc for c in columns for cc in c.values.columns]
        first_tree = MerkleProver.commit(coord_cols)
        channel.mix_root(first_tree.root())
        first_layer = FriLayer(columns, first_tree)

        folded = lambda v: v.domain.size() >> CIRCLE_TO_LINE_FOLD_STEP
        first_log = (folded(columns[0])).bit_length() - 1
        layer_eval = LineEvaluation.new_zero(LineDomain(Coset.half_odds(first_log)))
        it = iter(columns)
        alpha = channel.draw_felt()
        HipFriOps.fold_circle_into_line(layer_eval, next(it), alpha, twiddles)
        nxt = next(it, None)
        inner = []
        while layer_eval.len() > config.last_layer_domain_size():
            tree = MerkleProver.commit(layer_eval.values.columns)           # FriInnerLayerProver::new
            channel.mix_root(tree.root())
            alpha = channel.draw_felt()
            layer = FriLayer(layer_eval, tree)
            layer_eval = HipFriOps.fold_line(layer_eval, alpha, twiddles)
            if nxt is not None and folded(nxt) == layer_eval.len():
                HipFriOps.fold_circle_into_line(layer_eval, nxt, alpha, twiddles)
                nxt = next(it, None)
            inner.append(layer)
        # last layer (fri.ts:718-754)
        if layer_eval.len() != config.last_layer_domain_size():
            raise ValueError("last layer domain size mismatch")
        coeffs_br = line_interpolate(layer_eval, twiddles)
        log_n = len(coeffs_br).bit_length() - 1
        ordered = [coeffs_br[bit_reverse_index(i, log_n)] for i in range(len(coeffs_br))]   # intoOrderedCoefficients
        bound = 1 << config.log_last_layer_degree_bound
        if any(c.tup() != (0, 0, 0, 0) for c in ordered[bound:]):
            raise ValueError("invalid degree")
        last = ordered[:bound]
        channel.mix_felts(last)
        return FriProver(config, first_layer, inner, last)
