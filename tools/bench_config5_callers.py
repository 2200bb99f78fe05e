#!/usr/bin/env python3
"""The callers either side of the hot path AT CONFIG 5's SIZE (pcs/prover.ts:26-252 Rust text), on one GPU:

  * commit: TreeBuilder.extend_evals of the 256-column x 2^22 trace ("Interpolation for commitment": out-of-place interpolate),
    then the trace committed as 8 trees of 32 polynomials — "Extension" to the blown-up domain (log 22 + blowup 1 = 23, fused
    extend + evaluate), 8 Merkle trees in one tstwo_merkle_commit_many sequence, mix_root of each;
  * prove_values on it: every column opened at two points (z and a second point: two sample batches over ONE column list — the
    pair kernel), out-of-domain evaluation, quotients on the log-23 domain, FRI commit, grind, FRI + tree decommitments.

    python tools/bench_config5_callers.py [--reps 3]        one JSON line per caller

bench.py imports run_config5_callers() and appends the records to `configs` of its JSON line.  Algorithmic bytes (what the
step must move if every array crossed HBM once): interpolate 8 N per column; extend + evaluate 4 N read + 8 N written per
column (2 N-word evaluations); Merkle per tree (4 C + 64) 2 N.  tests/test_gpu_config5_callers.py checks the 8 roots of this
commit against the C oracle at full size.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM = 8000.0


def build_trace(total_cols, n, seed0=100):
    from bench import splitmix_columns
    return splitmix_columns([seed0 + c for c in range(total_cols)], 1 << n)


def commit_trace(T, cols_dev, n, blowup, tree_cols, tw, channel):
    """extend_evals + commit of the trace as trees of tree_cols polynomials (one phase).  Returns the scheme."""
    scheme = T.CommitmentSchemeProver(T.PcsConfig(pow_bits=10, fri_config=T.FriConfig(0, blowup, 40)), tw)
    tb = scheme.tree_builder()
    dom = T.CanonicCoset(n).circleDomain()
    tb.extend_evals([T.HipCircleEvaluation(dom, c) for c in cols_dev])
    sets = [tb.polys[k:k + tree_cols] for k in range(0, len(tb.polys), tree_cols)]
    scheme.commit_many(sets, channel)
    return scheme


def run_config5_callers(reps=3, total_cols=256, n=22, blowup=1, tree_cols=32, emit_line=None):
    import tstwo_amd as T
    from tstwo_amd import _lib as L
    L.ensure_init()
    N = 1 << n
    tw = T.precompute_twiddles(T.CanonicCoset(n + blowup).circleDomain().halfCoset)
    cols_dev = [T.HipColumn(c) for c in build_trace(total_cols, n)]
    L.sync()
    records = []

    def emit(rec):
        records.append(rec)
        if emit_line:
            emit_line(rec)

    # ---- commit
    times = []
    scheme = None
    for _ in range(reps + 1):                # first repetition: warm-up (allocator, clocks)
        scheme = None
        L.sync()
        t0 = time.perf_counter()
        scheme = commit_trace(T, cols_dev, n, blowup, tree_cols, tw, T.Blake2sChannel())
        L.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = min(times[1:])
    n_trees = total_cols // tree_cols
    b_interp = 8.0 * N * total_cols
    b_eval = 12.0 * N * total_cols
    b_merkle = (4.0 * tree_cols + 64.0) * (N << blowup) * n_trees
    gbps = (b_interp + b_eval + b_merkle) / (ms * 1e-3) / 1e9
    emit({"config": 5, "kernel": f"CommitmentSchemeProver: extend_evals (interpolate {total_cols} x 2^{n}) + commit_many of {n_trees} trees of {tree_cols} "
                                 f"polynomials, blowup {blowup} (extend + evaluate to log {n + blowup}, {n_trees} Merkle trees in one sequence, mix_root) "
                                 f"(wall per call, host mirror included)",
          "ms": round(ms, 4), "ms_all": [round(t, 3) for t in times], "algorithmic_GB": round((b_interp + b_eval + b_merkle) / 1e9, 3),
          "algorithmic_GB_parts": {"interpolate": round(b_interp / 1e9, 3), "extend_evaluate": round(b_eval / 1e9, 3), "merkle": round(b_merkle / 1e9, 3)},
          "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM, 4), "elems_per_s": total_cols * N / (ms * 1e-3),
          "roots": [r.hex() for r in scheme.roots()]})

    # ---- prove_values: every column opened at two points
    z = T.SECURE_FIELD_CIRCLE_GEN
    z2 = z.add(z) if hasattr(z, "add") else z.double()
    pts = [[[z, z2] for _ in range(tree_cols)] for _ in range(n_trees)]
    times = []
    for _ in range(reps + 1):
        ch = T.Blake2sChannel()
        L.sync()
        t0 = time.perf_counter()
        proof = scheme.prove_values(pts, ch)
        L.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = min(times[1:])
    n_eval = N << blowup
    b_oods = 2 * 4.0 * N * total_cols                      # two points x every coefficient once
    b_quot = 4.0 * n_eval * total_cols + 16.0 * n_eval     # every evaluation word once + the quotient column written
    gbps = (b_oods + b_quot) / (ms * 1e-3) / 1e9
    emit({"config": 5, "kernel": f"CommitmentSchemeProver.prove_values on that commitment: {total_cols} columns opened at 2 points each (eval_at_point x {2 * total_cols}, "
                                 f"quotients over {total_cols} columns x 2^{n + blowup} with the pair kernel, FRI commit at log {n + blowup}, grind {scheme.config.pow_bits} bits, "
                                 f"{scheme.config.fri_config.n_queries} queries: FRI + {n_trees} tree decommitments) (wall per call, host mirror included)",
          "ms": round(ms, 4), "ms_all": [round(t, 3) for t in times],
          "algorithmic_GB": round((b_oods + b_quot) / 1e9, 3), "algorithmic_GB_parts": {"out_of_domain_sampling": round(b_oods / 1e9, 3), "quotients": round(b_quot / 1e9, 3)},
          "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM, 4),
          "fri_layers": len(proof.fri_proof.inner_layers), "queried_values": sum(len(v) for v in proof.queried_values)})
    return records


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--log", type=int, default=22)
    a = ap.parse_args()
    run_config5_callers(a.reps, a.cols, a.log, emit_line=lambda r: print(json.dumps(r), flush=True))
