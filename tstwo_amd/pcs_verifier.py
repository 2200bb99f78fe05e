"""CommitmentSchemeVerifier (host) — consumes CommitmentSchemeProver.prove_values' proof.

Follows the Rust text the reference carries in packages/core/src/pcs/verifier.ts:1-125 and pcs/quotients.ts:111-167
(fri_answers).  The row quotients are a few field operations per query and run on the host with the same constants
(`quotientConstants`) the device kernel receives."""
from __future__ import annotations

from .circle import CanonicCoset, bit_reverse_index
from .fields import CM31, M31, QM31
from .fri_verifier import CirclePolyDegreeBound, FriVerifier
from .pcs import CommitmentSchemeProof, PcsConfig, PointSample, column_sample_batches
from .quotients import quotientConstants
from .vcs import Blake2sMerkleHasher, MerkleVerifier


class VerificationError(Exception):
    ProofOfWork = "Proof of work verification failed."


def denominator_inverses(sample_batches, domain_point) -> list:
    """backend/cpu/quotients.ts:160-178.  Rust semantics: Pr = c0, Pi = c1 of the sample point; under
    tstwo_amd.set_semantics("ts") the TS port's reading (c0.real / c0.imag as real CM31s), like the prover side."""
    from .semantics import ts_compat
    out = []
    for sb in sample_batches:
        if ts_compat():
            z = M31.zero()
            prx, pry = CM31(sb.point.x.c0.real, z), CM31(sb.point.y.c0.real, z)
            pix, piy = CM31(sb.point.x.c0.imag, z), CM31(sb.point.y.c0.imag, z)
        else:
            prx, pry, pix, piy = sb.point.x.c0, sb.point.y.c0, sb.point.x.c1, sb.point.y.c1
        px, py = CM31(domain_point.x, M31.zero()), CM31(domain_point.y, M31.zero())
        out.append(prx.sub(px).mul(piy).sub(pry.sub(py).mul(pix)).inverse())
    return out


def accumulate_row_quotients(sample_batches, queried_values_at_row, constants, domain_point) -> QM31:
    """backend/cpu/quotients.ts:80-116."""
    line_coeffs, batch_coeffs = constants
    acc = QM31.zero()
    for sb, lc, bc, dinv in zip(sample_batches, line_coeffs, batch_coeffs, denominator_inverses(sample_batches, domain_point)):
        num = QM31.zero()
        for (ci, _), (a, b, c) in zip(sb.columns_and_values, lc):
            value = c.mulM31(queried_values_at_row[ci])
            num = num.add(value.sub(a.mulM31(domain_point.y).add(b)))
        acc = acc.mul(bc).add(num.mul_cm31(dinv))
    return acc


def fri_answers_for_log_size(log_size, samples, random_coeff, query_positions, queried_values_iters, n_columns) -> list:
    """pcs/quotients.ts Rust text :136-167.  queried_values_iters: one iterator per tree (advanced in place);
    n_columns: per tree, how many of its columns have this log size."""
    batches = column_sample_batches(samples)
    constants = quotientConstants(batches, random_coeff)
    domain = CanonicCoset(log_size).circleDomain()
    out = []
    for q in query_positions:
        p = domain.at(bit_reverse_index(q, log_size))
        row = []
        for it, n in zip(queried_values_iters, n_columns):
            for _ in range(n):
                try:
                    row.append(next(it))
                except StopIteration:
                    raise VerificationError("too few queried values") from None
        out.append(accumulate_row_quotients(batches, row, constants, p))
    return out


def fri_answers(column_log_sizes, samples, random_coeff, query_positions_per_log_size, queried_values, n_columns_per_log_size) -> list:
    """pcs/quotients.ts Rust text :111-134: per log size (descending), the quotient value at each query."""
    its = [iter(v) for v in queried_values]
    flat = [(lg, s) for tree_lg, tree_s in zip(column_log_sizes, samples) for lg, s in zip(tree_lg, tree_s)]
    flat.sort(key=lambda t: -t[0])                                  # stable
    out, i = [], 0
    while i < len(flat):
        lg = flat[i][0]
        grp = []
        while i < len(flat) and flat[i][0] == lg:
            grp.append(flat[i][1])
            i += 1
        out.append(fri_answers_for_log_size(lg, grp, random_coeff, query_positions_per_log_size[lg], its,
                                            [m.get(lg, 0) for m in n_columns_per_log_size]))
    return out


class CommitmentSchemeVerifier:
    """pcs/verifier.ts Rust text :19-124."""

    def __init__(self, config: PcsConfig):
        self.config, self.trees = config, []

    def column_log_sizes(self) -> list:
        return [list(t.columnLogSizes) for t in self.trees]

    def commit(self, commitment: bytes, log_sizes, channel) -> None:
        channel.mix_root(commitment)
        ext = [lg + self.config.fri_config.log_blowup_factor for lg in log_sizes]
        self.trees.append(MerkleVerifier(Blake2sMerkleHasher, commitment, ext))

    def verify_values(self, sampled_points, proof: CommitmentSchemeProof, channel) -> None:
        channel.mix_felts([v for tree in proof.sampled_values for col in tree for v in col])
        random_coeff = channel.draw_felt()
        blow = self.config.fri_config.log_blowup_factor
        sizes = sorted({lg for tree in self.column_log_sizes() for lg in tree}, reverse=True)
        bounds = [CirclePolyDegreeBound(lg - blow) for lg in sizes]
        fri_verifier = FriVerifier.commit(channel, self.config.fri_config, proof.fri_proof, bounds)
        channel.mix_u64(proof.proof_of_work)
        if channel.trailing_zeros() < self.config.pow_bits:
            raise VerificationError(VerificationError.ProofOfWork)
        query_positions = fri_verifier.sample_query_positions(channel)
        if not (len(self.trees) == len(proof.decommitments) == len(proof.queried_values)):
            raise VerificationError("proof does not match the number of commitment trees")
        for tree, dec, vals in zip(self.trees, proof.decommitments, proof.queried_values):
            try:
                tree.verify(query_positions, vals, dec)
            except ValueError as e:
                raise VerificationError(f"Merkle verification failed: {e}") from None
        samples = [[[PointSample(pt, v) for pt, v in zip(pts, vals)] for pts, vals in zip(tree_pts, tree_vals)]
                   for tree_pts, tree_vals in zip(sampled_points, proof.sampled_values)]
        answers = fri_answers(self.column_log_sizes(), samples, random_coeff, query_positions, proof.queried_values,
                              [t.nColumnsPerLogSize for t in self.trees])
        fri_verifier.decommit(answers)
