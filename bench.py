#!/usr/bin/env python3
"""bench.py — the hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 5, column-sharded; weak scaling): every rank owns 32 trace columns
of 2^22 M31 words (256 columns at 8 GPUs).  One step = PolyOps.evaluate (Circle FFT) of the rank's
32 columns on CanonicCoset(22).circleDomain(), then MerkleProver.commit (Blake2s) over them, then
an all-gather of the ranks' 32-byte Merkle roots (RCCL; N > 1 only).  Inputs are synthetic
(SplitMix64 seeds 100+c) and resident in HBM before the timed region; twiddles are prebuilt.
The transform is data-oblivious, so each step re-evaluates the previous step's output in place
(uniform canonical M31 columns again) — no work is skipped or cached.
Order of a run: one step on the fresh input (its Merkle root is kept and compared with the CPU oracle's -> `root_match`),
--spinup untimed steps (default 80: the part needs ~50 ms of load before its clocks settle; `spinup_steps` in the line),
W - 1 more warm-up steps, barrier, K timed steps, barrier.

Rank 0 prints ONE JSON line:  value = (all ranks' columns * 2^22 elements * K) / max-over-ranks time.
`roofline` prices the dominant kernel (the CFFT pass kernel) against the 8 TB/s HBM roofline with the
algorithmic bytes of SURVEY.md §8(d); `cpu_baseline` times the CPU oracle on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_SIZE = 22
COLS_PER_GPU = 32
VALU_PER_BUTTERFLY = 11.6      # measured: (113.7M + 153.6M wave instr) * 64 / (32 cols * 22 layers * 2^21), profiles/r02_sq_counters.json
VALU_PEAK = 256 * 4 * 16 * 2.4e9          # nominal: one wave64 VALU instruction per SIMD per 4 cycles at 2.4 GHz
VALU_PEAK_MEASURED = 35.3e12              # what ONE VALU issue port sustains (one instruction per ~4.4 nominal cycles per SIMD):
                                          # tools/microbench2.hip, profiles/r02_microbench.json
VALU_PEAK_DUAL_MEASURED = 60.3e12         # the butterfly's 11 instructions issued in priority phases (second port takes the light
                                          # VOP2s): 2.61 nominal cycles each, tools/microbench3.hip, profiles/r02_microbench3.json
SPINUP_STEPS = 80                         # untimed steps before the W warm-up steps: the part needs ~40-75 ms of load before its
                                          # clocks settle (tools/cfft_time.py --series: 560 -> 487 us per transform), and W = 5
                                          # steps are 5 ms
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def splitmix_column(seed: int, n: int) -> np.ndarray:
    """Uniform M31 column from SplitMix64(seed) with rejection of values >= P (vectorised)."""
    P = 2147483647
    out = np.empty(n, dtype=np.uint32)
    filled = 0
    state = np.uint64(seed)
    gamma = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        while filled < n:
            m = n - filled + 64
            s = state + gamma * np.arange(1, m + 1, dtype=np.uint64)
            state = s[-1]
            z = s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            v = (z >> np.uint64(33)).astype(np.uint32)
            v = v[v < P][: n - filled]
            out[filled:filled + v.size] = v
            filled += v.size
    return out


def cpu_oracle():
    """The CPU oracle module (oracle/): reached ONLY from the cpu_baseline legs — this file's and tools/bench_configs.py's,
    which times BASELINE configs 1-4 the same way.  Never imported by the product (tstwo_amd/)."""
    from oracle import oracle as orc
    return orc


def cpu_baseline(sample_cols: int, threads: int = 0):
    """CPU oracle (oracle/, kind "port": -O3 scalar C) timed beside the GPU step on a bounded sample:
      * one thread: `sample_cols` of the 32 columns — CFFT evaluate + Merkle commit over them;
      * all host cores (reported as cpu_baseline.value): the FULL 32-column step on C threads (oracle/tstwo_oracle_mt.c:
        one column per task for the CFFT, the leaf range cut into contiguous shards for the Merkle tree, whose root equals
        the single-tree root) — `threads` = every core the process may run on unless given.
    Returns the record and the oracle's root of the step's input, which main() compares with the GPU's."""
    orc = cpu_oracle()
    n = LOG_SIZE
    half = orc.lib().orc_half_odds_initial(n - 1)
    tw, _ = orc.precompute_twiddles(half, n - 1, inverse=False)       # untimed, like the GPU side
    cols = [splitmix_column(100 + c, 1 << n) for c in range(sample_cols)]
    t0 = time.perf_counter()
    evs = [orc.cfft_evaluate(c, n, half, tw, n - 1) for c in cols]
    t1 = time.perf_counter()
    orc.merkle_commit(evs, [n] * sample_cols)
    t2 = time.perf_counter()
    single = {
        "value": sample_cols * (1 << n) / (t2 - t0), "unit": "elems/s", "cores": 1,
        "sample": f"{sample_cols} of {COLS_PER_GPU} columns x 2^{n}: CFFT evaluate ({t1 - t0:.2f} s) + Merkle commit over them ({t2 - t1:.2f} s)",
        "cfft_butterflies_per_s": sample_cols * n * (1 << (n - 1)) / (t1 - t0),
    }
    del evs
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if threads <= 0:
        threads = cores
    all_cols = cols + [splitmix_column(100 + c, 1 << n) for c in range(sample_cols, COLS_PER_GPU)]
    t3 = time.perf_counter()
    orc.mt_cfft_evaluate(all_cols, n, half, tw, n - 1, threads)       # in place
    t4 = time.perf_counter()
    root = orc.mt_merkle_root(all_cols, n, threads)
    t5 = time.perf_counter()
    return {
        "value": COLS_PER_GPU * (1 << n) / (t5 - t3),
        "unit": "elems/s",
        "cores": threads,
        "kind": "port",
        "sample": f"the full step ({COLS_PER_GPU} columns x 2^{n}) on {threads} C threads = every core this process may use "
                  f"(os.cpu_count() = {os.cpu_count()}): column-parallel CFFT ({t4 - t3:.2f} s) + leaf-sharded Merkle commit "
                  f"({t5 - t4:.2f} s); oracle built -O3",
        "cfft_butterflies_per_s": COLS_PER_GPU * n * (1 << (n - 1)) / (t4 - t3),
        "root": root.hex(),
        "single_thread": single,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup", type=int, default=SPINUP_STEPS, help="untimed clock spin-up steps before the warm-up steps")
    ap.add_argument("--cols", type=int, default=COLS_PER_GPU, help="columns per GPU")
    ap.add_argument("--log-size", type=int, default=LOG_SIZE)
    ap.add_argument("--cpu-cols", type=int, default=4, help="columns in the CPU-oracle sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 1-4 (the `configs` list of the JSON line)")
    ap.add_argument("--pmc-json", default=None, help="tools/pmc_summary.py output of a --pmc run of THIS command: fills roofline.traffic")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n, n_cols = args.log_size, args.cols
    N = 1 << n

    # torch.distributed is control plane only (rendezvous, barrier, max-reduce of the elapsed time, hand-over of the RCCL
    # unique id) on the gloo backend; the one collective on the data path — the all-gather of Merkle roots — is RCCL
    # over xGMI issued through the library's own C ABI (tstwo_comm_init / tstwo_allgather_async), exactly what a Bun
    # host would call.  TSTWO_BENCH_COLLECTIVE=gloo rehearses the N > 1 control flow on a one-GPU box (all ranks share
    # GPU 0, where RCCL refuses several ranks per device): roots then travel through host memory.
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)
    collective = os.environ.get("TSTWO_BENCH_COLLECTIVE", "rccl")
    use_dist = world > 1 or bool(os.environ.get("TSTWO_FORCE_DIST"))   # FORCE: rehearse the collective path at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from tstwo_amd import _lib as L
    from tstwo_amd.backend import HipBackend, shard_columns

    L.init(dev_index)
    if use_dist and collective == "rccl":
        ids = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            L.call("tstwo_comm_unique_id", buf)
            ids[0] = bytes(buf)
        dist.broadcast_object_list(ids, src=0)
        L.call("tstwo_comm_init", rank, world, (C.c_uint8 * 128).from_buffer_copy(ids[0]))
    backend = HipBackend()

    # ---- the rank's shard of the (world * n_cols) trace columns, resident in HBM
    my_cols = shard_columns(world * n_cols, world, rank)
    dev_cols = []
    for c in my_cols:
        b = L.DeviceBuffer(4 * N)
        b.upload(splitmix_column(100 + c, N))
        dev_cols.append(b)
    col_ptrs = L.ptr_array([b.ptr for b in dev_cols])
    half_initial = backend.canonic_half_coset_initial(n)
    tw = L.DeviceBuffer(4 * (N // 2))
    L.call("tstwo_twiddles_build", half_initial, n - 1, C.c_void_p(tw.ptr), C.c_void_p(0))
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    log_sizes = L.u32x([n] * n_cols)
    # two root slots; the all-gather of step k runs on the library's collective stream and overlaps the CFFT of step k+1
    # (tstwo_comm_wait at the next issue point makes the main stream wait for it on the device, never the host)
    root_slot = [L.DeviceBuffer(32) for _ in range(2)]
    roots_all = [L.DeviceBuffer(32 * world) for _ in range(2)]
    roots_host = [np.zeros(32 * world, dtype=np.uint8) for _ in range(2)]
    step_no = [0]
    L.sync()

    # HIP events on the library's stream, three per timed step, read after the timed region
    evs = [[L.Event() for _ in range(3)] for _ in range(args.steps)]

    def step(ev):
        if ev:
            ev[0].record()
        L.call("tstwo_cfft_evaluate", col_ptrs, n_cols, n, half_initial, C.c_void_p(tw.ptr), n - 1)
        if ev:
            ev[1].record()
        L.call("tstwo_merkle_commit", col_ptrs, log_sizes, n_cols, C.c_void_p(layers.ptr), None)
        if ev:
            ev[2].record()
        if use_dist:   # the only exchange on the path: 32-byte roots over RCCL/xGMI
            k = step_no[0] & 1
            step_no[0] += 1
            if collective == "rccl":
                L.call("tstwo_comm_wait")                  # the previous step's collective (it had this whole step to finish)
                L.call("tstwo_copy", C.c_void_p(root_slot[k].ptr), C.c_void_p(layers.ptr), 32)
                L.call("tstwo_allgather_async", C.c_void_p(root_slot[k].ptr), C.c_void_p(roots_all[k].ptr), 32)
            else:                                           # rehearsal: host round trip + gloo
                mine = torch.from_numpy(layers.download(np.uint8, 32).copy())
                out = torch.from_numpy(roots_host[k])
                dist.all_gather_into_tensor(out, mine)

    def drain():
        if collective == "rccl":
            L.call("tstwo_comm_wait")

    def barrier():
        if use_dist:
            drain()
        L.sync()
        if n_dev:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        L.sync()
        if n_dev:
            torch.cuda.synchronize()

    # Correctness of the measured workload: the first (untimed) step runs on the fresh synthetic input; its root is
    # compared below with the CPU oracle's root of the same 32 x 2^22 columns (cpu_baseline.root) -> "root_match".
    step(None)
    gpu_root_first = bytes(layers.download(np.uint8, 32).tobytes())
    for _ in range(args.spinup):          # clock spin-up (not counted as warm-up; see SPINUP_STEPS)
        step(None)
    for _ in range(max(args.warmup - 1, 0)):
        step(None)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])
    barrier()
    elapsed = time.perf_counter() - t0
    t_cfft = sum(e[0].elapsed_ms(e[1]) for e in evs)
    t_merkle = sum(e[1].elapsed_ms(e[2]) for e in evs)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's root must have arrived in rank order: compare the RCCL result with a gloo all-gather of the same roots
        last = (step_no[0] - 1) & 1
        got = roots_all[last].download(np.uint8, 32 * world) if collective == "rccl" else roots_host[last]
        mine = torch.from_numpy(layers.download(np.uint8, 32).copy())
        ref = torch.zeros(32 * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(ref, mine)
        assert bytes(got.tobytes()) == bytes(ref.numpy().tobytes()), "all-gathered roots differ from the ranks' own roots"

    if rank == 0:
        steps = args.steps
        total_elems = world * n_cols * N * steps
        cfft_ms = t_cfft / steps
        merkle_ms = t_merkle / steps
        # dominant kernel: the CFFT pass kernels (fast::k_cfft_a<INV,K>, fast::k_cfft_b<INV,LOGT>), (passes) launches per step over all columns.
        passes = 1 if n <= 13 else 1 + -(-(n - 13) // 9)
        algo_bytes_transform = 8.0 * N * n_cols                     # 8*N per column transform
        algo_bytes_launch = algo_bytes_transform / passes
        launch_ms = cfft_ms / passes
        achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9
        merkle_bytes = (4.0 * n_cols + 64.0) * N                    # 4*C*N read + 64*N written
        # HBM traffic is a PMC measurement of a separate rocprofv3 --pmc run of this same command; it enters the line only
        # when that run's summary is handed in (tools/refresh_profiles.sh does), never from a stale committed file.
        traffic, traffic_source = None, None
        if args.pmc_json and os.path.exists(args.pmc_json):
            pj = json.load(open(args.pmc_json))
            traffic = pj.get("hbm_bytes_per_launch")
            import hashlib
            lib_now = hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16]
            traffic_source = {"file": os.path.relpath(os.path.abspath(args.pmc_json), ROOT), "kernels": pj.get("cfft_kernels"),
                              "lib_sha16": pj.get("lib_sha16"), "same_build": pj.get("lib_sha16") == lib_now}
            if not traffic_source["same_build"]:
                traffic = None          # counters of another build say nothing about this one
        out = {
            "metric": "M31 CFFT elems/sec at log_size=22 (per step: CFFT evaluate + Blake2s Merkle commit + root all-gather)",
            "value": total_elems / elapsed,
            "unit": "elems/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "spinup_steps": args.spinup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 5 shard: {n_cols} columns x 2^{n} per GPU "
                                   f"({world * n_cols} columns total), CircleDomain of CanonicCoset({n}); "
                                   "evaluate + per-GPU Merkle tree" + (" + RCCL all-gather of roots (C ABI: tstwo_allgather_async)" if world > 1 else ""),
                       "log_size": n, "columns_per_gpu": n_cols, "parallelism": f"column-shard x{world}"},
            "cfft_ms": cfft_ms,
            "cfft_butterflies_per_s": n_cols * n * (N // 2) / (cfft_ms * 1e-3),
            "cfft_elems_per_s": n_cols * N / (cfft_ms * 1e-3),
            "merkle_ms": merkle_ms,
            "merkle_GBps": merkle_bytes / (merkle_ms * 1e-3) / 1e9,
            "merkle_frac_of_hbm_peak": merkle_bytes / (merkle_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "roofline": {"bound": "hbm", "kernel": "fast::k_cfft_a<false,9,0,14> + fast::k_cfft_b<false,13,false> (the two passes of one transform)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": passes, "avg_launch_ms": launch_ms,
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         # the lane-op rate is reported next to the HBM fraction.  11.3-12.2 VALU instructions
                         # per butterfly (profiles/r02_sq_counters.json); one issue port: nominal 39.3e12 lane-ops/s
                         # = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz, measured 35.3e12; with the second port taking the light
                         # VOP2s of another wave (priority phases) the same instruction mix peaks at 60.3e12.
                         "valu": {"instr_per_butterfly": VALU_PER_BUTTERFLY, "peak_lane_ops_per_s": VALU_PEAK,
                                  "achieved_lane_ops_per_s": VALU_PER_BUTTERFLY * n_cols * n * (N // 2) / (cfft_ms * 1e-3),
                                  "frac": VALU_PER_BUTTERFLY * n_cols * n * (N // 2) / (cfft_ms * 1e-3) / VALU_PEAK,
                                  "measured_peak_lane_ops_per_s": VALU_PEAK_MEASURED,
                                  "frac_of_measured_peak": VALU_PER_BUTTERFLY * n_cols * n * (N // 2) / (cfft_ms * 1e-3) / VALU_PEAK_MEASURED,
                                  "dual_issue_peak_lane_ops_per_s": VALU_PEAK_DUAL_MEASURED,
                                  "frac_of_dual_issue_peak": VALU_PER_BUTTERFLY * n_cols * n * (N // 2) / (cfft_ms * 1e-3) / VALU_PEAK_DUAL_MEASURED}},
            "device": L.device_name(),
        }
        root_ok = None
        if not args.no_cpu and args.cpu_cols > 0 and n == LOG_SIZE and n_cols == COLS_PER_GPU and world == 1:   # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_cols)
            root_ok = out["cpu_baseline"]["root"] == gpu_root_first.hex()
        else:
            out["cpu_baseline"] = None
        out["gpu_root"] = gpu_root_first.hex()
        out["root_match"] = root_ok            # GPU root of the first step == CPU oracle root of the same input (None: oracle leg not run)
        if world == 1 and not args.no_configs:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from bench_configs import run_configs
            del dev_cols[:]
            out["configs"] = run_configs(reps=10, no_cpu=args.no_cpu)      # BASELINE configs 1-4, same JSON line
        print(json.dumps(out), flush=True)
        if root_ok is False:
            raise SystemExit("bench.py: the GPU Merkle root of the 32 x 2^22 step differs from the CPU oracle's: the measured numbers are void")

    if use_dist:
        dist.barrier()
        if collective == "rccl":
            L.call("tstwo_comm_destroy")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
