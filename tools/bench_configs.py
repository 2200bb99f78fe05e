#!/usr/bin/env python3
"""Times BASELINE.json configs 1-4 on one GPU (HIP events on the library's stream): algorithmic bytes /
time vs the 8 TB/s HBM roofline, plus a bounded CPU-oracle timing beside each.

    python tools/bench_configs.py [--no-cpu] [--reps 10]      one JSON line per kernel

bench.py imports run_configs() and puts the same records into its JSON line ("configs"), so the driver-run bench carries
configs 1-4 as well as the headline config 5.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM = 8000.0


def run_configs(reps=10, no_cpu=False, emit_line=None):
    """Runs every config-1..4 kernel; returns the list of records (and calls emit_line(record) as each is ready)."""
    from bench import cpu_oracle, splitmix_column
    from tstwo_amd import _lib as L
    import tstwo_amd as T

    class A:            # the old script's `args`
        pass
    args = A()
    args.reps, args.no_cpu = reps, no_cpu
    L.ensure_init()
    records = []

    last_cold = [None]

    def timed(fn, reps=args.reps, warm=2):
        """Steady-state time per call (ms): the GPU idles during the CPU-oracle legs between the kernels, and the part needs
        40-75 ms of load before its clocks settle (DESIGN.md 4.1 "Clocks"), so after `warm` calls the first `reps` calls are
        timed as the COLD figure (emit() records it as ms_cold), the kernel then runs for ~80 ms untimed, and the steady figure is
        taken over >= 20 ms of back-to-back calls."""
        def measure(k):
            e0, e1 = L.Event(), L.Event()
            L.sync()
            e0.record()
            for _ in range(k):
                fn()
            e1.record()
            return e0.elapsed_ms(e1) / k
        for _ in range(warm):
            fn()
        cold = measure(reps)
        last_cold[0] = cold
        for _ in range(min(5000, int(80.0 / max(cold, 1e-3)) + 1)):
            fn()
        return measure(min(5000, max(reps, int(20.0 / max(cold, 1e-3)) + 1)))

    def emit(config, kernel, ms, algo_bytes, units, unit_name, cpu=None):
        gbps = algo_bytes / (ms * 1e-3) / 1e9
        out = {"config": config, "kernel": kernel, "ms": round(ms, 5), "ms_cold": round(last_cold[0], 5) if last_cold[0] else None,
               "algorithmic_GB": round(algo_bytes / 1e9, 4),
               "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM, 4), unit_name + "_per_s": units / (ms * 1e-3)}
        if cpu:
            out["cpu_oracle"] = cpu
        records.append(out)
        if emit_line:
            emit_line(out)

    def cpu_time(fn, units, unit_name, sample):
        if args.no_cpu:
            return None
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        return {unit_name + "_per_s": units / dt, "seconds": round(dt, 3), "cores": 1, "sample": sample}

    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    orc = None if args.no_cpu else cpu_oracle()   # cpu_baseline leg: the oracle is the timed CPU "port", never part of the GPU path

    # ---------------------------------------------------------------- config 1: M31 add / mul / batch_inverse on 2^20
    n1 = 1 << 20
    a_h, b_h = splitmix_column(1, n1), splitmix_column(2, n1)
    b_h[b_h == 0] = 1
    a, b, o = T.HipColumn(a_h), T.HipColumn(b_h), T.HipColumn.uninitialized(n1)
    for _ in range(2000):        # the GPU may have idled for seconds (bench.py's CPU leg): bring the clocks back up before timing
        L.call("tstwo_m31_mul", vp(a), vp(b), vp(o), n1)
    L.sync()
    for op in ("add", "mul"):
        ms = timed(lambda: L.call(f"tstwo_m31_{op}", vp(a), vp(b), vp(o), n1))
        emit(1, f"m31_{op} 2^20", ms, 12.0 * n1, n1, "elems", cpu_time(lambda: orc.col_op(op, a_h, b_h), n1, "elems", "full 2^20") if not args.no_cpu else None)
    ms = timed(lambda: L.call("tstwo_m31_batch_inverse", vp(b), vp(o), n1))
    emit(1, "m31_batch_inverse 2^20", ms, 8.0 * n1, n1, "elems", cpu_time(lambda: orc.m31_batch_inverse(b_h), n1, "elems", "full 2^20") if not args.no_cpu else None)
    ms = timed(lambda: L.call("tstwo_m31_batch_inverse_async", vp(b), vp(o), n1))
    L.call("tstwo_check_zero_flag")
    emit(1, "m31_batch_inverse_async 2^20 (zero check deferred to one tstwo_check_zero_flag per phase)", ms, 8.0 * n1, n1, "elems")

    # ---------------------------------------------------------------- config 2: interpolate + evaluate, 1 column, log 20
    n = 20
    N = 1 << n
    dom = T.CanonicCoset(n).circleDomain()
    tw = T.precompute_twiddles(dom.halfCoset)
    col_h = splitmix_column(3, N)
    col = T.HipColumn(col_h)
    ptr1 = L.ptr_array([col.ptr])
    half = dom.halfCoset.initial_index.value
    ms_e = timed(lambda: L.call("tstwo_cfft_evaluate", ptr1, 1, n, half, vp(tw.twiddles), n - 1))
    cold_e = last_cold[0]
    ms_i = timed(lambda: L.call("tstwo_cfft_interpolate", ptr1, 1, n, half, vp(tw.itwiddles), n - 1))
    cold_i = last_cold[0]
    cpu = None
    if not args.no_cpu:
        otw, _ = orc.precompute_twiddles(half, n - 1, inverse=False)
        cpu = cpu_time(lambda: orc.cfft_evaluate(col_h, n, half, otw, n - 1), n * (N // 2), "butterflies", "full column log 20")
    last_cold[0] = cold_e
    emit(2, "cfft_evaluate 1 col log 20", ms_e, 8.0 * N, n * (N // 2), "butterflies", cpu)
    last_cold[0] = cold_i
    emit(2, "cfft_interpolate 1 col log 20", ms_i, 8.0 * N, n * (N // 2), "butterflies")
    last_cold[0] = None                       # the eval_at_point records below are wall-clock loops of their own
    # PolyOps.eval_at_point (a14): one point, one column log 22 (call = kernels + 16-byte read-back); 32 columns log 20 at one point
    import time as _t
    n22 = 22
    c22 = T.HipColumn(splitmix_column(31, 1 << n22))
    pt = T.SECURE_FIELD_CIRCLE_GEN
    px, py, o4 = L.u32x(pt.x.tup()), L.u32x(pt.y.tup()), L.u32x([0] * 4)

    def wall(fn, reps=50):
        for _ in range(3):
            fn()
        L.sync()
        t0 = _t.perf_counter()
        for _ in range(reps):
            fn()
        return (_t.perf_counter() - t0) / reps * 1e3
    ms = wall(lambda: L.call("tstwo_eval_at_point", vp(c22), n22, px, py, o4))
    emit(2, "eval_at_point 1 col log 22 (wall per call incl. result read-back)", ms, 4.0 * (1 << n22), 1 << n22, "coeffs")
    c20 = [T.HipColumn(splitmix_column(40 + i, 1 << 20)) for i in range(32)]
    p20 = L.ptr_array([c.ptr for c in c20])
    o128 = L.u32x([0] * 128)
    ms = wall(lambda: L.call("tstwo_eval_at_point_batch", p20, 32, 20, px, py, o128))
    emit(2, "eval_at_point_batch 32 cols log 20 (wall per call incl. read-back)", ms, 4.0 * 32 * (1 << 20), 32 << 20, "coeffs")
    del c22, c20

    # ---------------------------------------------------------------- config 3: quotients (C=4, 1 batch) + QM31 batch inverse, log 22
    n = 22
    N = 1 << n
    dom = T.CanonicCoset(n).circleDomain()
    cols_h = [splitmix_column(4 + c, N) for c in range(4)]
    cols = [T.HipColumn(c) for c in cols_h]
    point = T.SECURE_FIELD_CIRCLE_GEN
    vals = [T.QM31.from_u32_unchecked(7 + c, 8, 9, 10) for c in range(4)]
    batches = [T.ColumnSampleBatch(point, [(c, vals[c]) for c in range(4)])]
    coeff = T.QM31.from_u32_unchecked(1, 2, 3, 4)
    from tstwo_amd.quotients import marshal_quotient_args
    _keep, qargs = marshal_quotient_args(dom, cols, coeff, batches)
    qout = T.SecureColumnByCoords.uninitialized(N)
    ms = timed(lambda: L.call("tstwo_quotients_accumulate", *qargs, qout.ptrs()), reps=5)
    cpu = None
    if not args.no_cpu:
        s = 18
        sub = [c[: 1 << s] for c in cols_h]
        cpu = cpu_time(lambda: orc.accumulate_quotients(T.CanonicCoset(s).circleDomain().halfCoset.initial_index.value, s, sub, coeff.tup(),
                                                        [(point.x.tup(), point.y.tup(), [(c, vals[c].tup()) for c in range(4)])]),
                       1 << s, "rows", "2^18 rows (same per-row work)")
    emit(3, "accumulate_quotients C=4 log 22 (C-ABI call: constant upload + kernel + error-flag readback)", ms, 32.0 * N, N, "rows", cpu)
    ms = timed(lambda: L.call("tstwo_quotients_accumulate_async", *qargs, qout.ptrs()), reps=args.reps)
    L.call("tstwo_check_zero_flag")
    emit(3, "accumulate_quotients_async C=4 log 22 (constant upload + kernel; zero check deferred to one tstwo_check_zero_flag per phase)", ms, 32.0 * N, N, "rows")
    sec_h = [splitmix_column(8 + c, N) for c in range(4)]
    for c in sec_h:
        c[c == 0] = 1
    sec = T.SecureColumnByCoords.from_numpy(sec_h)
    out4 = T.SecureColumnByCoords.uninitialized(N)
    ms = timed(lambda: L.call("tstwo_qm31_batch_inverse", sec.ptrs(), out4.ptrs(), N), reps=5)
    cpu = cpu_time(lambda: orc.qm31_batch_inverse([c[: 1 << 18] for c in sec_h]), 1 << 18, "elems", "2^18 elements") if not args.no_cpu else None
    emit(3, "qm31_batch_inverse log 22", ms, 32.0 * N, N, "elems", cpu)
    ms = timed(lambda: L.call("tstwo_qm31_batch_inverse_async", sec.ptrs(), out4.ptrs(), N), reps=args.reps)
    L.call("tstwo_check_zero_flag")
    emit(3, "qm31_batch_inverse_async log 22 (kernel only; zero check deferred)", ms, 32.0 * N, N, "elems")

    # ---------------------------------------------------------------- config 4: fold_circle_into_line + Merkle (C=4), log 24
    n = 24
    N = 1 << n
    dom = T.CanonicCoset(n).circleDomain()
    tw24 = T.precompute_twiddles(dom.halfCoset)
    src_h = [splitmix_column(9 + c, N) for c in range(4)]
    src = T.SecureEvaluation(dom, T.SecureColumnByCoords.from_numpy(src_h))
    dst = T.LineEvaluation.new_zero(T.LineDomain(dom.halfCoset))
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    a4 = L.u32x(alpha.tup())
    ms = timed(lambda: L.call("tstwo_fri_fold_circle_into_line", dst.values.ptrs(), N // 2, src.values.ptrs(), n, vp(tw24.itwiddles), n - 1, a4))
    cpu = None
    if not args.no_cpu:
        s = 18
        hs = T.CanonicCoset(s).circleDomain().halfCoset.initial_index.value
        cpu = cpu_time(lambda: orc.fold_circle_into_line([np.zeros(1 << (s - 1), dtype=np.uint32)] * 4, [c[: 1 << s] for c in src_h], s, hs, alpha.tup()),
                       1 << (s - 1), "rows", "2^17 output rows (reference formulation: per-row scalar mul + inverse)")
    emit(4, "fold_circle_into_line log 24", ms, 32.0 * N, N // 2, "rows", cpu)
    line = dst
    out_line = T.SecureColumnByCoords.uninitialized(N // 4)
    ms = timed(lambda: L.call("tstwo_fri_fold_line", line.values.ptrs(), n - 1, vp(tw24.itwiddles), n - 1, a4, out_line.ptrs()))
    emit(4, "fold_line log 23 -> 22", ms, 24.0 * (N // 2), N // 4, "rows")
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    cptrs = L.ptr_array([c.ptr for c in src.values.columns])
    ls = L.u32x([n] * 4)
    ms = timed(lambda: L.call("tstwo_merkle_commit", cptrs, ls, 4, vp(layers), None), reps=5)
    cpu = None
    if not args.no_cpu:
        s = 18
        cpu = cpu_time(lambda: orc.merkle_commit([c[: 1 << s] for c in src_h], [s] * 4), 2 * (1 << s), "compressions", "4 columns x 2^18")
    emit(4, "merkle_commit C=4 log 24", ms, 80.0 * N, 2 * N, "compressions", cpu)
    # bit reverse (ColumnOps)
    br = T.HipColumn(src_h[0])
    ms = timed(lambda: L.call("tstwo_bit_reverse", L.ptr_array([br.ptr]), 1, N))
    emit(4, "bit_reverse log 24", ms, 8.0 * N, N, "elems")

    return records


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    run_configs(a.reps, a.no_cpu, emit_line=lambda r: print(json.dumps(r), flush=True))
