"""FRI commit phase on the GPU — the caller on top of FriOps + MerkleOps.

Mirrors `FriProver.commit` (packages/core/src/fri.ts:644-754) with the wiring the Rust stwo prover has and the TS
port still mocks (`fri.ts:497,530,693,704`): every layer is committed with a real Blake2s Merkle tree over its 4
coordinate columns (device resident), the root is mixed into the channel, and the folding alpha is drawn from it.
Layers never leave HBM between fold and commit; only 32-byte roots and the (tiny) last layer reach the host."""
from __future__ import annotations

import os
from dataclasses import dataclass, field

from . import _lib as L
from .backend import HipColumn, SecureColumnByCoords
from .channel import DeviceChannel
from .circle import Coset, LineDomain, bit_reverse_index, bit_reverse_perm
from .queries import Queries, get_query_positions_by_log_size
from .fields import M31, P, QM31
from .fri import CIRCLE_TO_LINE_FOLD_STEP, HipFriOps
from .poly import LineEvaluation, SecureEvaluation, TwiddleTree
from .vcs import DeviceHashLayer, HashSlices, M31Values, MerkleDecommitment, MerkleProver, TreeLayers

FOLD_STEP = 1


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


def _line_interpolate_uses_tree(evaluation: LineEvaluation, twiddles) -> bool:
    n = evaluation.len()
    return twiddles is not None and n >= 2 and evaluation.domain().coset().is_doubling_of(twiddles.rootCoset)


def line_interpolate(evaluation: LineEvaluation, twiddles: TwiddleTree | None = None, prefetched=None) -> list:
    """LineEvaluation.interpolate (poly/line.ts:312-329): the LinePoly coefficients, bit-reversed order, as QM31."""
    vals = line_interpolate_words(evaluation, twiddles, prefetched)
    return [QM31.from_u32_unchecked(*row) for row in vals.T.tolist()]


def line_interpolate_device(evaluation: LineEvaluation, twiddles: TwiddleTree, out: "L.DeviceBuffer | None" = None) -> "L.DeviceBuffer | None":
    """The same on the device (tstwo_line_interpolate: one workgroup, at most 2^12 values on a doubling of the twiddle tree's
    root): enqueues the kernel and returns the buffer of the four coefficient columns (4 x n words, coordinate-major) for the
    caller to fetch together with whatever else it reads back; None when the layer does not qualify."""
    n = evaluation.len()
    if twiddles is None or n > (1 << 12) or not all(isinstance(c, HipColumn) for c in evaluation.values.columns):
        return None
    if n >= 2 and not _line_interpolate_uses_tree(evaluation, twiddles):
        return None
    if out is None:
        out = L.DeviceBuffer(16 * n)
    L.call("tstwo_line_interpolate", evaluation.values.ptrs(), n.bit_length() - 1, L.vp(twiddles.itwiddles.buf.ptr),
           twiddles.log_size, L.p4([out.ptr + 4 * n * k for k in range(4)]))
    return out


def line_interpolate_words(evaluation: LineEvaluation, twiddles: TwiddleTree | None = None, prefetched=None):
    """LineEvaluation.interpolate + lineIfft (poly/line.ts:312-390) on the host: the last FRI layer has at most
    2^(log_last_layer_degree_bound + log_blowup_factor) elements.  Returns the bit-reversed-order coefficients as a (4, n) uint32 array.
    Vectorised over the layer with numpy u64 (4 coordinate rows; M31 ops are coordinate-wise because every twiddle is
    in the base field).  The x^-1 of each level are read from the tail of the inverse twiddle tree when the domain is a
    doubling of its root (level of log size k = 2^(k-1) entries, bit-reversed); otherwise they are computed per element
    like the reference."""
    import numpy as np
    P_ = np.uint64(P)
    n = evaluation.len()
    log_n = n.bit_length() - 1
    br = bit_reverse_perm(log_n)
    # prefetched = (the 4 coordinate columns, the last n entries of the inverse twiddle tree or None): the caller fetched them
    # together with other small results in one round trip (FriProver.commit)
    cols_host = prefetched[0] if prefetched is not None else evaluation.values.to_numpy()
    vals = np.stack([c.astype(np.uint64) for c in cols_host])[:, br]                          # (4, n), natural order
    domain = evaluation.domain()
    tail = None
    if _line_interpolate_uses_tree(evaluation, twiddles):
        if prefetched is not None:
            tail = prefetched[1]
        else:
            L_ = twiddles.itwiddles.len()
            tail = twiddles.itwiddles.buf.download(count=n, offset=4 * (L_ - n))              # last n entries of the tree
    while domain.size() > 1:
        size, half = domain.size(), domain.size() // 2
        k = domain.logSize()
        if tail is not None:
            seg = tail[n - size:n - half]                                                     # tree[L - 2^k : L - 2^(k-1)]
            inv = seg[bit_reverse_perm(k - 1)].astype(np.uint64)
        else:
            inv = np.array([domain.at(i).inverse().value for i in range(half)], dtype=np.uint64)
        v = vals.reshape(4, n // size, 2, half)
        a, b = v[:, :, 0, :], v[:, :, 1, :]
        s_, d_ = (a + b) % P_, ((a + P_ - b) % P_) * inv % P_                                  # ibutterfly (fft.ts:25-30)
        vals = np.stack([s_, d_], axis=2).reshape(4, n)
        domain = domain.double()
    len_inv = np.uint64(M31.from_(n).inverse().value)
    return (vals * len_inv % P_).astype(np.uint32)                                             # (4, n) coefficient words


class LinePoly:
    """LinePoly (poly/line.ts:127-236): coefficients of the x-basis in bit-reversed order, QM31, on the host."""

    def __init__(self, coeffs):
        coeffs = list(coeffs)
        if not coeffs or len(coeffs) & (len(coeffs) - 1):
            raise ValueError("coeffs length must be power of two")
        self.coeffs = coeffs
        self._log_size = len(coeffs).bit_length() - 1

    new = staticmethod(lambda coeffs: LinePoly(coeffs))

    def len(self): return 1 << self._log_size
    __len__ = len

    def eval_at_point(self, x: QM31) -> QM31:
        """line.ts:145-153 + fold (poly/utils.ts:36-59): doublings [x, pi(x), pi^2(x), ...], pi(x) = 2x^2 - 1."""
        doublings, cur = [], x
        for _ in range(self._log_size):
            doublings.append(cur)
            cur = cur.square().double().sub(QM31.one())

        def fold(values, factors):
            if len(values) == 1:
                return values[0]
            h = len(values) // 2
            return fold(values[:h], factors[1:]).add(fold(values[h:], factors[1:]).mul(factors[0]))
        return fold(self.coeffs, doublings)

    evalAtPoint = eval_at_point

    def into_ordered_coefficients(self) -> list:
        return [self.coeffs[bit_reverse_index(i, self._log_size)] for i in range(len(self.coeffs))]

    @staticmethod
    def from_ordered_coefficients(coeffs) -> "LinePoly":
        coeffs = list(coeffs)
        lg = len(coeffs).bit_length() - 1
        return LinePoly([coeffs[bit_reverse_index(i, lg)] for i in range(len(coeffs))])

    intoOrderedCoefficients, fromOrderedCoefficients = into_ordered_coefficients, from_ordered_coefficients


class QM31Rows:
    """A read-only sequence of QM31 over a (k, 4) uint32 array: the witness evaluations of a FRI layer as they came back from
    the device.  Elements become QM31 objects when they are looked at (the verifier, a serialiser), not when the proof is
    assembled — 600 witness evaluations of a 40-query proof cost 7 host objects each otherwise."""
    __slots__ = ("rows",)

    def __init__(self, rows):
        self.rows = rows

    def __len__(self): return len(self.rows)
    def __getitem__(self, i):
        if isinstance(i, slice):
            return [QM31.from_u32_unchecked(*r) for r in self.rows[i]]
        return QM31.from_u32_unchecked(*self.rows[i])
    def __iter__(self): return (QM31.from_u32_unchecked(*r) for r in self.rows)
    def __setitem__(self, i, q): self.rows[i] = list(q.to_m31_array_values() if hasattr(q, "to_m31_array_values") else (*q.c0.tup(), *q.c1.tup()))
    def pop(self, i=-1): return QM31.from_u32_unchecked(*self.rows.pop(i))
    def append(self, q): self.rows.append([*q.c0.tup(), *q.c1.tup()])
    def __eq__(self, o): return list(self) == list(o)
    def __add__(self, o): return list(self) + list(o)
    def __radd__(self, o): return list(o) + list(self)
    def __repr__(self): return f"QM31Rows({list(self)!r})"


@dataclass
class FriLayerProof:
    """fri.ts:262-269."""
    fri_witness: list = field(default_factory=list)          # QM31 the verifier cannot deduce
    decommitment: MerkleDecommitment = field(default_factory=MerkleDecommitment)
    commitment: bytes = b""


@dataclass
class FriProof:
    """fri.ts:274-278."""
    first_layer: FriLayerProof
    inner_layers: list
    last_layer_poly: LinePoly


def _decommitment_positions(query_positions, fold_step: int) -> tuple:
    """The index part of fri.ts:346-384: (decommitment positions, positions whose evaluation the verifier cannot compute)."""
    decommitment_positions, witness_positions = [], []
    qs = list(query_positions)
    i = 0
    while i < len(qs):
        coset = qs[i] >> fold_step
        start = coset << fold_step
        subset = set()
        while i < len(qs) and (qs[i] >> fold_step) == coset:
            subset.add(qs[i])
            i += 1
        for position in range(start, start + (1 << fold_step)):
            decommitment_positions.append(position)
            if position not in subset:
                witness_positions.append(position)
    return decommitment_positions, witness_positions


def compute_decommitment_positions_and_witness_evals(column: SecureColumnByCoords, query_positions, fold_step: int):
    """fri.ts:346-384.  Same walk; the witness values are fetched from the device column with ONE gather."""
    decommitment_positions, witness_positions = [], []
    qs = list(query_positions)
    i = 0
    while i < len(qs):
        coset = qs[i] >> fold_step
        start = coset << fold_step
        subset = []
        while i < len(qs) and (qs[i] >> fold_step) == coset:
            subset.append(qs[i])
            i += 1
        k = 0
        for position in range(start, start + (1 << fold_step)):
            decommitment_positions.append(position)
            if k < len(subset) and subset[k] == position:            # the verifier can calculate this one
                k += 1
                continue
            witness_positions.append(position)
    return decommitment_positions, column.gather(witness_positions)


computeDecommitmentPositionsAndWitnessEvals = compute_decommitment_positions_and_witness_evals


class FriFirstLayerProver:
    """Rust FriFirstLayerProver (TS mock at fri.ts:485-517): all circle columns under one Merkle tree."""

    def __init__(self, columns, merkle_tree: MerkleProver):
        self.columns, self.merkle_tree = list(columns), merkle_tree
        self.evaluation = self.columns                                   # FriLayer-compatible view

    def column_log_sizes(self): return {c.domain.logSize() for c in self.columns}
    def max_column_log_size(self): return max(self.column_log_sizes())
    columnLogSizes, maxColumnLogSize = column_log_sizes, max_column_log_size

    def decommit(self, queries: Queries) -> FriLayerProof:
        max_log = queries.log_domain_size
        assert max_log == self.max_column_log_size()
        fri_witness, positions_by_log = [], {}
        for column in self.columns:
            lg = column.domain.logSize()
            cq = queries.fold(max_log - lg)
            pos, wit = compute_decommitment_positions_and_witness_evals(column.values, cq.positions, CIRCLE_TO_LINE_FOLD_STEP)
            positions_by_log[lg] = pos
            fri_witness += wit
        _, dec = self.merkle_tree.decommit(positions_by_log, [cc for c in self.columns for cc in c.values.columns], want_queried=False)
        return FriLayerProof(fri_witness, dec, self.merkle_tree.root())


class FriInnerLayerProver:
    """Rust FriInnerLayerProver (TS mock at fri.ts:519-542): one line evaluation, its 4 coordinate columns committed."""

    def __init__(self, evaluation: LineEvaluation, merkle_tree: MerkleProver):
        self.evaluation, self.merkle_tree = evaluation, merkle_tree

    def decommit(self, queries: Queries) -> FriLayerProof:
        pos, wit = compute_decommitment_positions_and_witness_evals(self.evaluation.values, queries.positions, FOLD_STEP)
        _, dec = self.merkle_tree.decommit({self.evaluation.domain().logSize(): pos}, self.evaluation.values.columns, want_queried=False)
        return FriLayerProof(wit, dec, self.merkle_tree.root())


FriLayer = FriInnerLayerProver


class _HostTranscript:
    """mix_root / draw_felt on the host channel: one 32-byte root read-back per layer."""
    sync_root = True

    def __init__(self, channel):
        self.channel = channel

    def mix_and_draw(self, tree):
        self.channel.mix_root(tree.root())
        return self.channel.draw_felt()

    fold_line = staticmethod(HipFriOps.fold_line)
    fold_circle = staticmethod(HipFriOps.fold_circle_into_line)


class _DeviceTranscript:
    """mix_root / draw_felt by the device channel: alpha k lands in slot k of `alphas`; nothing is read back."""
    sync_root = False

    def __init__(self, dch: DeviceChannel, alphas):
        self.dch, self.alphas, self.k = dch, alphas, 0

    def mix_and_draw(self, tree) -> int:
        ptr = self.alphas.ptr + 16 * self.k
        self.k += 1
        self.dch.mix_root_draw_felt(tree.root_ptr(), ptr)
        return ptr

    fold_line = staticmethod(HipFriOps.fold_line_dev)
    fold_circle = staticmethod(HipFriOps.fold_circle_into_line_dev)


class FriProver:
    def __init__(self, config, first_layer, inner_layers, last_layer_coeffs):
        self.config, self.first_layer, self.inner_layers, self.last_layer_poly = config, first_layer, inner_layers, last_layer_coeffs

    @staticmethod
    def commit(channel, config: FriConfig, columns, twiddles: TwiddleTree, device_channel: bool = True) -> "FriProver":
        """columns: SecureEvaluation list, canonic domains, strictly decreasing sizes (fri.ts:644-674)."""
        if not columns:
            raise ValueError("no columns")
        if not all(c.domain.isCanonic() for c in columns):
            raise ValueError("not canonic")
        for a, b in zip(columns, columns[1:]):
            if a.domain.size() <= b.domain.size():
                raise ValueError("column sizes not decreasing")
        # Device transcript: when every fold can take its twiddles from the tree and the channel has Rust semantics, the whole
        # commit loop is one launch sequence — roots are mixed and alphas drawn by the device channel, nothing is read back
        # until the last layer.  Otherwise: the host channel, one 32-byte read-back per layer.
        on_device = (device_channel and not getattr(channel, "ts_compat", False) and hasattr(channel, "_digest")
                     and FriProver._device_capable(columns, twiddles))
        if device_channel and getattr(channel, "ts_compat", False):
            # the device channel implements Rust's draw_felt only: say so instead of silently taking the slower path
            import warnings
            warnings.warn("FriProver.commit: a ts-compatible channel (draw_felt queue, channel/blake2.ts:177-184) keeps the "
                          "transcript on the host: one 32-byte root read-back per FRI layer instead of the device channel",
                          RuntimeWarning, stacklevel=2)
        if on_device:
            dch = DeviceChannel(channel)
            alphas = L.DeviceBuffer(16 * (columns[0].domain.logSize() + 2))
            # the whole layer loop is ONE library call (tstwo_fri_commit_layers); FriCommitPlan captures the per-layer calls of
            # _commit_layers instead (a capture cannot allocate)
            if os.environ.get("TSTWO_FRI_COMMIT_HOST_LOOP"):            # A/B timing: round 2's loop, ~10 C-ABI calls per layer
                first_layer, inner, layer_eval = FriProver._commit_layers(config, columns, twiddles, _DeviceTranscript(dch, alphas))
            else:
                first_layer, inner, layer_eval = FriProver._commit_layers_in_library(config, columns, twiddles, dch, alphas)
            prefetched = FriProver._fetch_end_of_commit(dch, layer_eval, twiddles)
        else:
            first_layer, inner, layer_eval = FriProver._commit_layers(config, columns, twiddles, _HostTranscript(channel))
            prefetched = None
        last = FriProver._commit_last_layer(channel, config, layer_eval, twiddles, prefetched)
        return FriProver(config, first_layer, inner, last)

    @staticmethod
    def _fetch_end_of_commit(dch: DeviceChannel, layer_eval: LineEvaluation, twiddles: TwiddleTree, coeff_buf=None):
        """What the host needs to finish a device-transcript commit — the channel state and the last layer's polynomial — in ONE
        round trip (six separate read-backs were 0.17 ms of a 0.76 ms commit).  The interpolation itself
        (LineEvaluation.interpolate) runs on the device when the layer fits one workgroup (coeff_buf: already enqueued by the
        caller); otherwise its inputs (coordinate columns, x^-1 slice of the tree) come back.  Updates the host channel; returns
        what _commit_last_layer takes as `prefetched`."""
        n_last = layer_eval.len()
        separate = bool(os.environ.get("TSTWO_FRI_SEPARATE_READBACKS"))   # A/B timing: round 3's first form, one tstwo_download per piece
        if coeff_buf is None and not separate and not os.environ.get("TSTWO_FRI_HOST_LAST_LAYER"):
            coeff_buf = line_interpolate_device(layer_eval, twiddles)
        pieces = [(dch.buf.ptr, 10)]
        uses_tree = _line_interpolate_uses_tree(layer_eval, twiddles)
        if coeff_buf is not None:
            pieces.append((coeff_buf.ptr, 4 * n_last))
        else:
            pieces += [(c.buf.ptr, n_last) for c in layer_eval.values.columns]
            if uses_tree:
                pieces.append((twiddles.itwiddles.buf.ptr + 4 * (twiddles.itwiddles.len() - n_last), n_last))
        got = [L.download_many([pc])[0] for pc in pieces] if separate else L.download_many(pieces)
        dch.sync_to_host(got[0])                                         # the host channel continues from the device state
        if coeff_buf is not None:
            return {"coeffs": got[1].reshape(4, n_last)}
        return (got[1:5], got[5] if uses_tree else None)

    @staticmethod
    def _device_capable(columns, twiddles) -> bool:
        first_log = (columns[0].domain.size() >> CIRCLE_TO_LINE_FOLD_STEP).bit_length() - 1
        return (all(HipFriOps.can_fold_on_device(c.domain, twiddles) for c in columns)
                and HipFriOps.can_fold_on_device(LineDomain(Coset.half_odds(first_log)), twiddles))

    @staticmethod
    def _commit_layers_in_library(config: FriConfig, columns, twiddles: TwiddleTree, dch: DeviceChannel, alphas) -> tuple:
        """commitInnerLayers (fri.ts:676-716) through tstwo_fri_commit_layers: first-layer tree, then per layer mix root / draw
        alpha / fold / commit on the device, in one call; the buffers it returns are adopted by the host objects below."""
        import ctypes as C
        logs = [c.domain.logSize() for c in columns]
        first_log = logs[0] - CIRCLE_TO_LINE_FOLD_STEP
        last_log = config.last_layer_domain_size().bit_length() - 1
        if first_log < last_log:
            raise ValueError("last layer domain size mismatch")          # what commitLastLayer (fri.ts:718-754) says for this config
        cap = first_log - last_log + 1
        outs = (L.FriLayerOut * cap)()
        n_out, first = C.c_size_t(0), L.vp()
        L.call("tstwo_fri_commit_layers", L.ptr_array([cc.ptr for c in columns for cc in c.values.columns]), L.u32x(logs), len(columns),
               C.c_void_p(twiddles.itwiddles.ptr), twiddles.log_size, last_log, C.c_void_p(dch.buf.ptr), C.c_void_p(alphas.ptr),
               alphas.nbytes // 16, C.byref(first), outs, cap, C.byref(n_out))
        # every block the library handed out is adopted HERE, before any other object is built: an exception further down must not
        # leak the blocks not yet wrapped
        first_buf = L.DeviceBuffer.adopt(first.value, 32 * ((2 << logs[0]) - 1))
        adopted = []
        for i in range(n_out.value):
            o = outs[i]
            n = 1 << o.log_size
            cols = [L.DeviceBuffer.adopt(o.cols[k], 4 * n) for k in range(4)]
            tree = L.DeviceBuffer.adopt(o.layers, 32 * ((2 << o.log_size) - 1)) if o.layers else None
            adopted.append((o.log_size, cols, tree))

        def tree_of(buf, max_log):
            return MerkleProver(TreeLayers(buf, max_log), buf, None)

        def eval_of(entry, domain):
            n = 1 << entry[0]
            return LineEvaluation(domain, SecureColumnByCoords([HipColumn(_buf=b, _len=n) for b in entry[1]]))
        first_layer = FriFirstLayerProver(columns, tree_of(first_buf, logs[0]))
        domain = LineDomain(Coset.half_odds(first_log))
        inner = []
        for entry in adopted[:-1]:
            inner.append(FriInnerLayerProver(eval_of(entry, domain), tree_of(entry[2], entry[0])))
            domain = domain.double()
        return first_layer, inner, eval_of(adopted[-1], domain)

    @staticmethod
    def _commit_layers(config: FriConfig, columns, twiddles: TwiddleTree, transcript) -> tuple:
        """commitInnerLayers (fri.ts:676-716) with the Merkle / channel wiring of Rust.  `transcript` mixes a tree's root and
        draws alpha (host or device).  With the device transcript this only ENQUEUES work (capturable into a hipGraph)."""
        folded = lambda v: v.domain.size() >> CIRCLE_TO_LINE_FOLD_STEP
        first_log = (folded(columns[0])).bit_length() - 1
        # first layer: one tree over every column's coordinate columns (Rust FriFirstLayerProver::new), root -> channel
        coord_cols = [cc for c in columns for cc in c.values.columns]
        first_tree = MerkleProver.commit(coord_cols, sync_root=transcript.sync_root)
        alpha = transcript.mix_and_draw(first_tree)
        first_layer = FriFirstLayerProver(columns, first_tree)
        layer_eval = LineEvaluation.new_zero(LineDomain(Coset.half_odds(first_log)))
        it = iter(columns)
        transcript.fold_circle(layer_eval, next(it), alpha, twiddles)
        nxt = next(it, None)
        inner = []
        while layer_eval.len() > config.last_layer_domain_size():
            tree = MerkleProver.commit(layer_eval.values.columns, sync_root=transcript.sync_root)      # FriInnerLayerProver::new
            alpha = transcript.mix_and_draw(tree)
            layer = FriInnerLayerProver(layer_eval, tree)
            layer_eval = transcript.fold_line(layer_eval, alpha, twiddles)
            if nxt is not None and folded(nxt) == layer_eval.len():
                transcript.fold_circle(layer_eval, nxt, alpha, twiddles)
                nxt = next(it, None)
            inner.append(layer)
        if nxt is not None:
            raise ValueError("not all columns were consumed")                # Rust: assert!(columns.is_empty())
        return first_layer, inner, layer_eval

    @staticmethod
    def _commit_last_layer(channel, config: FriConfig, layer_eval: LineEvaluation, twiddles: TwiddleTree, prefetched=None) -> LinePoly:
        """commitLastLayer (fri.ts:718-754)."""
        if layer_eval.len() != config.last_layer_domain_size():
            raise ValueError("last layer domain size mismatch")
        import numpy as np
        if isinstance(prefetched, dict):
            cw = prefetched["coeffs"]                                        # interpolated on the device (line_interpolate_device)
        else:
            cw = line_interpolate_words(layer_eval, twiddles, prefetched)    # (4, n) words, bit-reversed coefficient order
        log_n = cw.shape[1].bit_length() - 1
        ordered = cw[:, bit_reverse_perm(log_n)]                             # intoOrderedCoefficients
        bound = 1 << config.log_last_layer_degree_bound
        if ordered[:, bound:].any():
            raise ValueError("invalid degree")
        # LinePoly.from_ordered_coefficients(ordered[:bound]): bit-reversed order over the bound
        words = np.ascontiguousarray(ordered[:, :bound][:, bit_reverse_perm(config.log_last_layer_degree_bound)].T, dtype="<u4")
        last = LinePoly([QM31.from_u32_unchecked(*row) for row in words.tolist()])
        channel.mix_felts(last.coeffs, _le_bytes=words.tobytes())   # Rust: channel.mix_felts(&last_layer_poly) = its bit-reversed coefficient slice
        return last

    def decommit(self, channel) -> tuple:
        """fri.ts:759-766: draws the queries, returns (FriProof, query positions by column log size)."""
        max_log = self.first_layer.max_column_log_size()
        queries = Queries.generate(channel, max_log, self.config.n_queries)
        by_log = get_query_positions_by_log_size(queries, self.first_layer.column_log_sizes())
        return self.decommit_on_queries(queries), by_log

    def decommit_on_queries(self, queries: Queries) -> FriProof:
        """fri.ts:768-785 in ONE library call (tstwo_fri_decommit): the position logic of fri.ts:346-384 for every layer, the
        witness evaluations and every tree's decommitment come back from one gather round trip."""
        import ctypes as C

        import numpy as np
        max_log = queries.log_domain_size
        assert max_log == self.first_layer.max_column_log_size()
        layers = [self.first_layer] + list(self.inner_layers)
        n = len(layers)
        descs = (L.FriLayer * n)()
        keep = []
        total_evals = 0
        for r, layer in enumerate(layers):
            evs = layer.columns if r == 0 else [layer.evaluation]
            colp = L.ptr_array([cc.ptr for e in evs for cc in e.values.columns])
            logs = L.u32x([(e.domain.logSize() if r == 0 else e.domain().logSize()) for e in evs])
            keep += [colp, logs]
            tree = layer.merkle_tree
            descs[r] = L.FriLayer(tree._buf.ptr, len(tree.layers) - 1, colp, logs, len(evs))
            total_evals += len(evs)
        nq = len(queries.positions)
        qarr = (C.c_uint64 * max(nq, 1))(*queries.positions)
        # capacities: a query touches one coset of 2^step positions per evaluation and layer; a position asks for at most one sibling
        # hash per level.  Should an estimate fall short the library reports the exact sizes and the call is repeated with them.
        cap_e = max(1, 2 * nq * total_evals)
        cap_h = max(1, sum(2 * nq * (len(l.columns) if r == 0 else 1) * len(l.merkle_tree.layers) for r, l in enumerate(layers)))
        cap_w = max(1, 8 * nq * total_evals)
        roots = np.empty(32 * n, dtype=np.uint8)
        counts = (C.c_size_t * (3 * n))()
        for attempt in range(2):
            evals = np.empty(4 * cap_e, dtype=np.uint32)
            hashes = np.empty(32 * cap_h, dtype=np.uint8)
            colwit = np.empty(cap_w, dtype=np.uint32)
            totals = (C.c_size_t * 3)(cap_e, cap_h, cap_w)
            try:
                L.call("tstwo_fri_decommit", descs, n, qarr, nq, max_log, CIRCLE_TO_LINE_FOLD_STEP, FOLD_STEP, evals.ctypes.data_as(L.u32p),
                       hashes.ctypes.data_as(L.u8p), colwit.ctypes.data_as(L.u32p), roots.ctypes.data_as(L.u8p), counts, totals)
                break
            except L.TstwoError as e:
                if attempt or "output buffer too small" not in str(e):
                    raise
                cap_e, cap_h, cap_w = max(1, totals[0]), max(1, totals[1]), max(1, totals[2])
        hb, rb = hashes.tobytes(), roots.tobytes()
        ev = evals[:4 * totals[0]].reshape(-1, 4).tolist()
        wl = colwit[:totals[2]].tolist()
        proofs, e0, h0, w0 = [], 0, 0, 0
        for r, layer in enumerate(layers):
            ne, nh, nw = counts[3 * r], counts[3 * r + 1], counts[3 * r + 2]
            dec = MerkleDecommitment(HashSlices(hb[32 * h0:32 * (h0 + nh)], nh), M31Values(wl[w0:w0 + nw]))
            layer.merkle_tree._root = rb[32 * r:32 * r + 32]          # (what root() would read back: 32 bytes per tree)
            proofs.append(FriLayerProof(QM31Rows(ev[e0:e0 + ne]), dec, layer.merkle_tree._root))
            e0, h0, w0 = e0 + ne, h0 + nh, w0 + nw
        return FriProof(proofs[0], proofs[1:], self.last_layer_poly)

    def decommit_on_queries_host_walk(self, queries: Queries) -> FriProof:
        """The round-2 path, kept to cross-check the in-library one: positions planned here (fri.ts:346-384), ONE gather for all
        witness evaluations and ONE tstwo_merkle_decommit_many call for all trees."""
        plans = []                      # (tree, positions_by_log, merkle columns, [(SecureColumnByCoords, witness positions)])
        max_log = queries.log_domain_size
        assert max_log == self.first_layer.max_column_log_size()
        by_log, wit = {}, []
        for column in self.first_layer.columns:
            lg = column.domain.logSize()
            pos, wpos = _decommitment_positions(queries.fold(max_log - lg).positions, CIRCLE_TO_LINE_FOLD_STEP)
            by_log[lg] = pos
            wit.append((column.values, wpos))
        plans.append((self.first_layer.merkle_tree, by_log, [cc for c in self.first_layer.columns for cc in c.values.columns], wit))
        layer_queries = queries.fold(CIRCLE_TO_LINE_FOLD_STEP)
        for layer in self.inner_layers:
            pos, wpos = _decommitment_positions(layer_queries.positions, FOLD_STEP)
            plans.append((layer.merkle_tree, {layer.evaluation.domain().logSize(): pos}, layer.evaluation.values.columns,
                          [(layer.evaluation.values, wpos)]))
            layer_queries = layer_queries.fold(FOLD_STEP)
        # all witness evaluations with one device gather
        flat = [(vals, p) for _, _, _, w in plans for vals, wp in w for p in wp]
        k = len(flat)
        witness = []
        if k:
            import ctypes as C

            import numpy as np
            srcs = (L.vp * (4 * k))(*[vals.columns[c].ptr for c in range(4) for vals, _ in flat])
            idx = (C.c_uint64 * (4 * k))(*([p for _, p in flat] * 4))
            out = np.empty(4 * k, dtype=np.uint32)
            L.call("tstwo_gather_words", srcs, idx, 1, 4 * k, out.ctypes.data_as(L.u32p))
            witness = [QM31.from_u32_unchecked(*r) for r in out.reshape(4, k).T.tolist()]
        decs = MerkleProver.decommit_many([(t, q, cols) for t, q, cols, _ in plans], want_queried=False)
        proofs, w0 = [], 0
        for (tree, _, _, w), (_, dec) in zip(plans, decs):
            n_w = sum(len(wp) for _, wp in w)
            proofs.append(FriLayerProof(witness[w0:w0 + n_w], dec, tree.root()))
            w0 += n_w
        return FriProof(proofs[0], proofs[1:], self.last_layer_poly)

    decommitOnQueries = decommit_on_queries


class FriCommitPlan:
    """A FRI commit of FIXED shape captured once into a hipGraph (tstwo_graph_*): the 4-6 small launches per layer of the
    device-transcript commit loop are launch-bound below ~2^16 rows, so replaying them as one graph removes the host from
    the loop entirely.  The plan owns every buffer the sequence touches; `columns` are the INPUT buffers — write new
    evaluations into them (same shape) and call run() again.  The FriProver returned by run() views the plan's buffers and
    is valid until the next run()."""

    def __init__(self, config: FriConfig, columns, twiddles: TwiddleTree):
        import ctypes as C
        if not columns or not all(c.domain.isCanonic() for c in columns):
            raise ValueError("no columns" if not columns else "not canonic")
        if not FriProver._device_capable(columns, twiddles):
            raise ValueError("twiddle tree mismatch")
        self.config, self.columns, self.twiddles = config, list(columns), twiddles
        self.chan = L.DeviceBuffer(64)
        self.alphas = L.DeviceBuffer(16 * (columns[0].domain.logSize() + 2))
        self._dch = None
        # one eager run: fills the allocator's free lists with exactly the blocks the sequence needs and performs the
        # one-time kernel set-up, so that the captured run issues nothing but launches
        from .channel import Blake2sChannel
        self._dch = DeviceChannel(Blake2sChannel(), buf=self.chan)
        FriProver._commit_layers(config, self.columns, twiddles, _DeviceTranscript(self._dch, self.alphas))
        L.sync()
        n_last = config.last_layer_domain_size()
        self._coeff_buf = L.DeviceBuffer(16 * n_last) if n_last <= (1 << 12) else None      # (a capture cannot allocate)
        L.call("tstwo_graph_begin_capture")
        try:
            self.first_layer, self.inner, self.last_eval = FriProver._commit_layers(
                config, self.columns, twiddles, _DeviceTranscript(self._dch, self.alphas))
            if self._coeff_buf is not None:                       # the last layer's interpolation is part of the graph
                self._coeff_buf = line_interpolate_device(self.last_eval, twiddles, out=self._coeff_buf)
        finally:
            h = C.c_void_p()
            L.call("tstwo_graph_end_capture", C.byref(h))
        self._exec = h

    def run(self, channel) -> FriProver:
        self._dch.load(channel)                                   # host channel state -> device (64 bytes)
        L.call("tstwo_graph_launch", self._exec)
        if self._coeff_buf is not None:
            prefetched = FriProver._fetch_end_of_commit(self._dch, self.last_eval, self.twiddles, coeff_buf=self._coeff_buf)
        else:
            self._dch.sync_to_host()
            prefetched = None
        for layer in [self.first_layer] + self.inner:             # roots are re-read lazily from the freshly written layers
            layer.merkle_tree._root = None
        last = FriProver._commit_last_layer(channel, self.config, self.last_eval, self.twiddles, prefetched)
        return FriProver(self.config, self.first_layer, self.inner, last)

    def __del__(self):
        try:
            if getattr(self, "_exec", None):
                L.call("tstwo_graph_destroy", self._exec)
        except Exception:
            pass

