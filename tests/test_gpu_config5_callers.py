"""-m gpu: the callers either side of the hot path at BASELINE config 5's size (pcs/prover.ts:26-252 Rust text):
the 256-column x 2^22 trace committed as 8 trees of 32 polynomials on the blown-up domain (log 23) in ONE phase
(CommitmentSchemeProver.commit_many -> batched extend + evaluate, tstwo_merkle_commit_many, mix_root per tree) — every root
against the C oracle at full size —, the transcript equal to one commit() per tree, and an opening proof over it that the host
verifier accepts."""
import os
import sys

import numpy as np
import pytest

from conftest import rand_column
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tstwo_amd as T  # noqa: E402
from bench import cpu_quota_cores, host_cores, splitmix_columns  # noqa: E402

pytestmark = pytest.mark.gpu
N_LOG, BLOW, TOTAL, TREE = 22, 1, 256, 32


def _threads():
    q = cpu_quota_cores()
    return max(1, min(host_cores(), int(2 * q) if q else host_cores()))


def test_config5_sized_commit_many_roots_match_the_oracle_and_one_commit_per_tree():
    n, big = N_LOG, N_LOG + BLOW
    coeffs = splitmix_columns([100 + c for c in range(TOTAL)], 1 << n)        # the trace's polynomials (bench.py's seeds)
    tw = T.precompute_twiddles(T.CanonicCoset(big).circleDomain().halfCoset)
    scheme = T.CommitmentSchemeProver(T.PcsConfig(fri_config=T.FriConfig(0, BLOW, 3)), tw)
    polys = [T.HipCirclePoly(T.HipColumn(c)) for c in coeffs]
    sets = [polys[k:k + TREE] for k in range(0, TOTAL, TREE)]
    ch = T.Blake2sChannel()
    scheme.commit_many(sets, ch)
    roots = scheme.roots()
    assert len(roots) == TOTAL // TREE == 8
    # oracle, one tree (1 GiB of evaluations) at a time: zero-extend, evaluate on the log-23 domain, leaf-sharded root
    half = orc.lib().orc_half_odds_initial(big - 1)
    otw, _ = orc.precompute_twiddles(half, big - 1, inverse=False)
    th = _threads()
    och = T.Blake2sChannel()
    for t in range(TOTAL // TREE):
        ext = []
        for c in coeffs[t * TREE:(t + 1) * TREE]:
            e = np.zeros(1 << big, dtype=np.uint32)
            e[:1 << n] = c
            ext.append(e)
        orc.mt_cfft_evaluate(ext, big, half, otw, big - 1, th)
        if t in (0, 5):                 # spot check of the evaluations themselves
            for k in (0, TREE - 1):
                assert (scheme.trees[t].evaluations[k].values.to_numpy() == ext[k]).all()
        assert roots[t] == orc.mt_merkle_root(ext, big, th), f"tree {t}"
        och.mix_root(roots[t])
    assert ch.digest() == och.digest()                     # mix_root in TreeVec order (pcs/prover.ts:227-228)
    # one commit() per tree: the same trees and the same transcript (two trees are enough: the launches are shared per shape)
    scheme1 = T.CommitmentSchemeProver(T.PcsConfig(fri_config=T.FriConfig(0, BLOW, 3)), tw)
    ch1, ch2 = T.Blake2sChannel(), T.Blake2sChannel()
    for s in sets[:2]:
        scheme1.commit(s, ch1)
    ch2.mix_root(roots[0]); ch2.mix_root(roots[1])
    assert scheme1.roots() == roots[:2] and ch1.digest() == ch2.digest()


def test_config5_sized_extend_evals_interpolates_like_the_oracle():
    """TreeBuilder.extend_evals at full width: 256 evaluations of 2^22 through one out-of-place interpolation; four of the columns
    against the oracle, all of them by evaluating back."""
    n = N_LOG
    vals = [rand_column(7100 + c, 1 << n) for c in range(TOTAL)]
    tw = T.precompute_twiddles(T.CanonicCoset(n).circleDomain().halfCoset)
    dom = T.CanonicCoset(n).circleDomain()
    scheme = T.CommitmentSchemeProver(T.PcsConfig(fri_config=T.FriConfig(0, BLOW, 3)), tw)
    tb = scheme.tree_builder()
    evs = [T.HipCircleEvaluation(dom, v) for v in vals]
    assert tb.extend_evals(evs) == (0, 0, TOTAL)
    half = dom.halfCoset.initial_index.value
    _, oitw = orc.precompute_twiddles(half, n - 1)
    for k in (0, 63, 64, 255):
        assert (tb.polys[k].coeffs.to_numpy() == orc.cfft_interpolate(vals[k], n, half, oitw, n - 1)).all()
        assert (evs[k].values.to_numpy() == vals[k]).all()          # value semantics: the evaluation survives
    back = T.evaluate_polynomials(tb.polys, dom, tw)
    for k in range(0, TOTAL, 17):
        assert (back[k].values.to_numpy() == vals[k]).all()


def test_config5_sized_opening_proof_verifies():
    """prove_values over the committed 256-column trace, every column opened at two points (two sample batches over one column
    list: the pair kernel at 256 columns x 2^23), FRI at log 23; the host verifier accepts it and ends in the prover's transcript state."""
    n = N_LOG
    config = T.PcsConfig(pow_bits=8, fri_config=T.FriConfig(0, BLOW, 6))
    tw = T.precompute_twiddles(T.CanonicCoset(n + BLOW).circleDomain().halfCoset)
    scheme = T.CommitmentSchemeProver(config, tw)
    ch = T.Blake2sChannel()
    config.mix_into(ch)
    polys = [T.HipCirclePoly(T.HipColumn(rand_column(7500 + c, 1 << n))) for c in range(TOTAL)]
    scheme.commit_many([polys[k:k + TREE] for k in range(0, TOTAL, TREE)], ch)
    point = T.CirclePoint.get_random_point(ch)
    shifted = point.add(T.SECURE_FIELD_CIRCLE_GEN)
    pts = [[[point, shifted] for _ in range(TREE)] for _ in range(TOTAL // TREE)]
    proof = scheme.prove_values(pts, ch)
    assert proof.commitments == scheme.roots()
    v = T.CommitmentSchemeVerifier(config)
    vch = T.Blake2sChannel()
    config.mix_into(vch)
    for root in proof.commitments:
        v.commit(root, [n] * TREE, vch)
    vpoint = T.CirclePoint.get_random_point(vch)
    vshift = vpoint.add(T.SECURE_FIELD_CIRCLE_GEN)
    v.verify_values([[[vpoint, vshift] for _ in range(TREE)] for _ in range(TOTAL // TREE)], proof, vch)
    assert ch.digest() == vch.digest()
