"""-m gpu: BASELINE config 5 as SURVEY.md §8(d) defines it — the FIXED 256-column x 2^22 trace — at its 1-, 2- and 4-GPU points
on one GPU, against the CPU oracle (16 C threads).

N = 1: all 256 columns evaluated by ONE tstwo_cfft_evaluate call, then one MerkleProver.commit per 32 columns = 8 trees (stwo's
TreeVec: one CommitmentTreeProver per tree, /root/reference/packages/core/src/pcs/prover.ts:62-64,209-237); every one of the
2^30 evaluation words and the 8 roots are compared.  N = 2 / 4 (128 / 64 columns per GPU): the first and the last rank's
shards go through the same calls; their roots must be the SAME 8 roots (the tree split does not depend on the GPU count).
Also: one single tree over all 256 columns (a 1 KiB leaf = 16 Blake2s blocks; the generic leaf kernel) against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

from tstwo_amd import _lib as L  # noqa: E402
from tstwo_amd.backend import shard_columns  # noqa: E402
from bench import splitmix_columns  # noqa: E402

N_LOG, TOTAL, TREE = 22, 256, 32
THREADS = max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def trace():
    n = N_LOG
    L.init(0)
    half = orc.lib().orc_half_odds_initial(n - 1)
    tw, _ = orc.precompute_twiddles(half, n - 1, inverse=False)
    coeffs = splitmix_columns([100 + c for c in range(TOTAL)], 1 << n)
    dev = []
    for c in coeffs:                                  # device copies of the coefficients, kept for the shard tests
        b = L.DeviceBuffer(4 << n)
        b.upload(c)
        dev.append(b)
    evals = orc.mt_cfft_evaluate(coeffs, n, half, tw, n - 1, THREADS)          # in place: `coeffs` now holds the evaluations
    roots = [orc.mt_merkle_root(evals[t:t + TREE], n, THREADS) for t in range(0, TOTAL, TREE)]
    assert len(set(roots)) == len(roots)
    dtw = L.DeviceBuffer(4 << (n - 1))
    L.call("tstwo_twiddles_build", half, n - 1, C.c_void_p(dtw.ptr), C.c_void_p(0))
    yield half, dev, evals, roots, dtw
    for b in dev:
        b.free()
    dtw.free()
    L.call("tstwo_trim")


def run_shard(half, dtw, dev_coeffs, cols):
    """Fresh copies of the shard's coefficient columns -> ONE evaluate call -> one commit per 32 columns."""
    n = N_LOG
    bufs = []
    for c in cols:
        b = L.DeviceBuffer(4 << n)
        L.call("tstwo_copy", C.c_void_p(b.ptr), C.c_void_p(dev_coeffs[c].ptr), 4 << n)
        bufs.append(b)
    L.call("tstwo_cfft_evaluate", L.ptr_array([b.ptr for b in bufs]), len(bufs), n, half, C.c_void_p(dtw.ptr), n - 1)
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    roots = []
    for t in range(0, len(bufs), TREE):
        root = (C.c_uint8 * 32)()
        L.call("tstwo_merkle_commit", L.ptr_array([b.ptr for b in bufs[t:t + TREE]]), L.u32x([n] * TREE), TREE, C.c_void_p(layers.ptr), root)
        roots.append(bytes(root))
    layers.free()
    return bufs, roots


def test_config5_trace_on_one_gpu_every_word_and_8_roots(trace):
    half, dev, evals, roots, dtw = trace
    bufs, got_roots = run_shard(half, dtw, dev, list(range(TOTAL)))
    assert got_roots == roots, [i for i, (a, b) in enumerate(zip(got_roots, roots)) if a != b]
    for i, (b, e) in enumerate(zip(bufs, evals)):
        got = b.download(np.uint32, 1 << N_LOG)
        assert (got == e).all(), f"evaluation column {i}: {int((got != e).sum())} words differ, first at {int(np.argmax(got != e))}"
    # the same 8 trees committed together (tstwo_merkle_commit_many: what bench.py's step calls)
    many = [L.DeviceBuffer(32 * ((2 << N_LOG) - 1)) for _ in range(TOTAL // TREE)]
    reqs = (L.CommitRequest * len(many))()
    keep = []
    for t in range(len(many)):
        cp, lg = L.ptr_array([b.ptr for b in bufs[t * TREE:(t + 1) * TREE]]), L.u32x([N_LOG] * TREE)
        keep += [cp, lg]
        reqs[t] = L.CommitRequest(cp, lg, TREE, many[t].ptr)
    roots_many = (C.c_uint8 * (32 * len(many)))()
    L.call("tstwo_merkle_commit_many", reqs, len(many), roots_many)
    assert [bytes(roots_many[32 * t:32 * t + 32]) for t in range(len(many))] == roots
    for m in many:
        m.free()
    # one tree over all 256 columns (leaf = 1 KiB): the shape a one-tree-per-GPU split would give at N = 1
    layers = L.DeviceBuffer(32 * ((2 << N_LOG) - 1))
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", L.ptr_array([b.ptr for b in bufs]), L.u32x([N_LOG] * TOTAL), TOTAL, C.c_void_p(layers.ptr), root)
    assert bytes(root) == orc.mt_merkle_root(evals, N_LOG, THREADS)
    layers.free()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("world", [2, 4])
def test_config5_shards_of_the_2_and_4_gpu_points_give_the_same_roots(trace, world):
    half, dev, evals, roots, dtw = trace
    for rank in (0, world - 1):
        cols = shard_columns(TOTAL, world, rank)
        assert len(cols) == TOTAL // world and len(cols) % TREE == 0
        bufs, got_roots = run_shard(half, dtw, dev, cols)
        first_tree = cols[0] // TREE
        assert got_roots == roots[first_tree:first_tree + len(cols) // TREE], f"world {world} rank {rank}"
        rng = np.random.default_rng(world * 10 + rank)
        for i in rng.choice(len(cols), size=4, replace=False):          # sampled columns, every word
            assert (bufs[i].download(np.uint32, 1 << N_LOG) == evals[cols[i]]).all(), f"world {world} rank {rank} column {cols[i]}"
        for b in bufs:
            b.free()
