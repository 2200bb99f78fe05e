#!/usr/bin/env python3
"""bench.py — the hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json config 5 as SURVEY.md §8(d) defines it: a FIXED trace of 256 columns x 2^22 M31 words, column-sharded
(strong scaling): rank g of N owns columns shard_columns(256, N, g) — all 256 at N = 1, 128 / 64 / 32 at N = 2 / 4 / 8 — and
commits them as 32-column Merkle trees, i.e. 8 trees in all whatever N is: the 8 roots (stwo's TreeVec, one
CommitmentTreeProver per tree, pcs/prover.ts:62-64,209-237) do not depend on the GPU count.  One step = PolyOps.evaluate
(Circle FFT) of the rank's columns on CanonicCoset(22).circleDomain() in ONE call, MerkleProver.commit (Blake2s) of the rank's
trees (one tstwo_merkle_commit_many call — a TreeVec committed together; the trees are byte for byte what tstwo_merkle_commit
builds one by one, and --commit-per-tree does exactly that),
then (N > 1) an all-gather of the ranks' 8/N roots of 32 bytes (RCCL, through the C ABI).  `--scaling weak` keeps round 2's
mode instead (32 columns = one tree per GPU, 32 N columns in all).  Inputs are synthetic (SplitMix64 seeds 100+c) and
resident in HBM before the timed region; twiddles are prebuilt.  The transform is data-oblivious, so each step re-evaluates
the previous step's output in place (uniform canonical M31 columns again) — no work is skipped or cached.

Order of a run (rank 0, N = 1: first a child process `rocprofv3 --pmc ... -- python tools/pmc_target.py` collects the HBM
counters of the same launches -> roofline.traffic): one step on the fresh input (its 8/N roots are kept and compared with the
CPU oracle's -> `root_match`), W - 1 more warm-up steps, barrier, K timed steps from idle clocks (`ms_per_step_cold`: what a
prover that commits a handful of trees sees), spin-up steps until the part has been under load for --spinup-ms (its clocks
need 40-75 ms to settle; `spinup_steps` in the line), barrier, K timed steps (`ms_per_step`, `value`), barrier.

Rank 0 prints ONE JSON line:  value = (all ranks' columns * 2^22 elements * K) / max-over-ranks time.
`roofline` prices the dominant kernel (the CFFT pass kernels) against the 8 TB/s HBM roofline with the
algorithmic bytes of SURVEY.md §8(d); `cpu_baseline` times the CPU oracle on the same step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_SIZE = 22
TOTAL_COLS = 256                # the fixed trace of BASELINE config 5
TREE_COLS = 32                  # columns per Merkle tree (8 trees = stwo's TreeVec of config 5)
VALU_PER_BUTTERFLY = 11.6      # measured: (113.7M + 153.6M wave instr) * 64 / (32 cols * 22 layers * 2^21), profiles/r02_sq_counters.json
VALU_PEAK = 256 * 4 * 16 * 2.4e9          # nominal: one wave64 VALU instruction per SIMD per 4 cycles at 2.4 GHz
VALU_PEAK_MEASURED = 35.3e12              # what ONE VALU issue port sustains (one instruction per ~4.4 nominal cycles per SIMD):
                                          # tools/microbench2.hip, profiles/r02_microbench.json
VALU_PEAK_DUAL_MEASURED = 60.3e12         # the butterfly's 11 instructions issued in priority phases (second port takes the light
                                          # VOP2s): 2.61 nominal cycles each, tools/microbench3.hip, profiles/r02_microbench3.json
SPINUP_MS = 200.0                         # load the part must have seen before the headline timed region: it needs ~40-75 ms
                                          # before its clocks settle (tools/cfft_time.py --series: 560 -> 487 us per transform)
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
            z = np.arange(1, m + 1, dtype=np.uint64)
            z *= gamma
            z += state
            state = z[-1].copy()
            t = z >> np.uint64(30)
            z ^= t
            z *= np.uint64(0xBF58476D1CE4E5B9)
            np.right_shift(z, np.uint64(27), out=t)
            z ^= t
            z *= np.uint64(0x94D049BB133111EB)
            np.right_shift(z, np.uint64(31), out=t)
            z ^= t
            z >>= np.uint64(33)
            v = z.astype(np.uint32)
            v = v[v < P][: n - filled]
            out[filled:filled + v.size] = v
            filled += v.size
    return out


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_quota_cores():
    """The CPU time this process's cgroup may use, in cores (cgroup v2 cpu.max, v1 cfs quota); None = unlimited / unknown.
    A GPU box shows every core of the host in the affinity mask and grants a fraction of them as quota."""
    for quota_f, period_f in (("/sys/fs/cgroup/cpu.max", None),
                              ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            txt = open(quota_f).read().split()
            if period_f is None:
                quota, period = txt[0], txt[1]
            else:
                quota, period = txt[0], open(period_f).read().split()[0]
            if quota in ("max", "-1"):
                return None
            return float(quota) / float(period)
        except (OSError, IndexError, ValueError):
            continue
    return None


def launch_ranks(n_gpus: int, argv: list) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process — which has not imported torch nor
    touched the GPU library, and never will — starts N fresh interpreters of this file, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, a free port on 127.0.0.1), lets rank 0's stdout through (the ONE JSON line),
    sends the other ranks' stdout to stderr, and returns non-zero if any rank does (the others are then stopped by PID)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = set(range(n_gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:
                    procs[o].terminate()
                deadline = time.time() + 20
                for o in live:
                    try:
                        procs[o].wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[o].kill()
        time.sleep(0.05)
    return rc


def splitmix_columns(seeds, n: int) -> list:
    """The same columns, generated on a few host threads (numpy releases the GIL): 256 columns x 2^22 take ~0.4 s each on one."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(16, host_cores()))) as ex:
        return list(ex.map(lambda sd: splitmix_column(sd, n), seeds))


def cpu_oracle():
    """The CPU oracle module (oracle/): reached ONLY from the cpu_baseline legs — this file's and tools/bench_configs.py's,
    which times BASELINE configs 1-4 the same way.  Never imported by the product (tstwo_amd/)."""
    from oracle import oracle as orc
    return orc


def cpu_baseline(cols_host: list, n: int, tree_cols: int, single_cols: int, threads: int = 0):
    """CPU oracle (oracle/, kind "port": -O3 scalar C) timed beside the GPU step:
      * one thread: `single_cols` of the columns — CFFT evaluate + Merkle commit over them;
      * a thread sweep on a bounded sample (CFFT of up to 32 of the columns, copies): 8, 16, ... up to every core in the
        affinity mask, plus the cgroup's CPU quota when it has one — a GPU box shows all of the host's cores in the mask and
        grants a fraction of them, so "every core" oversubscribes; the fastest count is the one the full step then runs on;
      * the FULL step of the trace — every column, every tree — on that many C threads (oracle/tstwo_oracle_mt.c: one column
        per task for the CFFT; per tree the leaf range cut into contiguous shards, whose root equals the single-tree root):
        reported as cpu_baseline.value with cores = the threads used, `affinity_cores` and `cpu_quota_cores` beside it.
    `cols_host` (the step's input, host copies) is transformed IN PLACE.  Returns the record; its "roots" are the oracle's
    roots of the step's trees, which main() compares with the GPU's."""
    orc = cpu_oracle()
    n_cols = len(cols_host)
    half = orc.lib().orc_half_odds_initial(n - 1)
    tw, _ = orc.precompute_twiddles(half, n - 1, inverse=False)       # untimed, like the GPU side
    t0 = time.perf_counter()
    evs = [orc.cfft_evaluate(c, n, half, tw, n - 1) for c in cols_host[:single_cols]]      # copies
    t1 = time.perf_counter()
    orc.merkle_commit(evs, [n] * single_cols)
    t2 = time.perf_counter()
    single = {
        "value": single_cols * (1 << n) / (t2 - t0), "unit": "elems/s", "cores": 1,
        "sample": f"{single_cols} of {n_cols} columns x 2^{n}: CFFT evaluate ({t1 - t0:.2f} s) + Merkle commit over them ({t2 - t1:.2f} s)",
        "cfft_butterflies_per_s": single_cols * n * (1 << (n - 1)) / (t1 - t0),
    }
    del evs
    cores = host_cores()
    quota = cpu_quota_cores()
    sweep = {}
    if threads <= 0:
        cand = {c for c in (8, 16, 32, 64, 128, 256, 512) if c < cores} | {cores}
        if quota:
            cand.add(max(1, min(cores, int(round(quota)))))
        sample = [c.copy() for c in cols_host[:min(32, n_cols)]]
        for t in sorted(cand):
            work = [c.copy() for c in sample]
            ts = time.perf_counter()
            orc.mt_cfft_evaluate(work, n, half, tw, n - 1, t)
            sweep[t] = len(work) * (1 << n) / (time.perf_counter() - ts)
        del sample, work
        threads = max(sweep, key=sweep.get)
    t3 = time.perf_counter()
    orc.mt_cfft_evaluate(cols_host, n, half, tw, n - 1, threads)       # in place
    t4 = time.perf_counter()
    roots = [orc.mt_merkle_root(cols_host[t:t + tree_cols], n, threads) for t in range(0, n_cols, tree_cols)]
    t5 = time.perf_counter()
    return {
        "value": n_cols * (1 << n) / (t5 - t3),
        "unit": "elems/s",
        "cores": threads,
        "affinity_cores": cores,
        "cpu_quota_cores": quota,
        "kind": "port",
        "sample": f"the full step ({n_cols} columns x 2^{n}, {len(roots)} trees of {tree_cols} columns) on {threads} C threads — the fastest "
                  f"of a sweep over {sorted(sweep) if sweep else [threads]} threads on a {min(32, n_cols)}-column CFFT sample (affinity mask {cores} "
                  f"cores, cgroup CPU quota {quota if quota else 'none'}): column-parallel CFFT ({t4 - t3:.2f} s) + leaf-sharded "
                  f"Merkle commits ({t5 - t4:.2f} s); oracle built -O3",
        "thread_sweep_cfft_elems_per_s": {str(k): v for k, v in sorted(sweep.items())},
        "cfft_butterflies_per_s": n_cols * n * (1 << (n - 1)) / (t4 - t3),
        "roots": [r.hex() for r in roots],
        "single_thread": single,
    }


def pmc_traffic(n: int, n_cols: int, timeout_s: float = 240.0):
    """HBM traffic of the CFFT pass kernels from the hardware counters, collected by THIS run: two fresh child processes
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- <python> tools/pmc_target.py --cfft-only` (separate passes: the TCC block cannot hold
    both; --pmc only, no trace domains; the interpreter itself after `--`, no shell / env / re-exec), summarised by
    tools/pmc_summary.py (the guide's gfx950 corrections, calibrated on kernels with known byte counts in the same run).
    Called before this process touches the GPU.  Returns (bytes per launch | None, source record)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, {"reason": "rocprofv3 not found"}
    if any(k.startswith("ROCPROF") or k.startswith("ROCPROFILER") for k in os.environ):
        return None, {"reason": "already running under a profiler"}
    tmp = tempfile.mkdtemp(prefix="tstwo_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    target = os.path.join(ROOT, "tools", "pmc_target.py")
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "p", "--",
                   sys.executable, target, "--cfft-only", "--cols", str(n_cols), "--log-size", str(n)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None, {"reason": f"rocprofv3 --pmc {ctr} exited with {r.returncode}: {r.stdout.decode(errors='replace')[-300:]}"}
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        pj = pmc_summary.summarise(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), n, n_cols)
    except subprocess.TimeoutExpired:
        return None, {"reason": f"rocprofv3 --pmc child exceeded {timeout_s:.0f} s"}
    except Exception as e:      # noqa: BLE001 — a counter leg that fails must not take the measurement down with it
        return None, {"reason": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    src = {"collector": "child rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_target.py --cfft-only (this run)",
           "kernels": pj.get("cfft_kernels"), "per_kernel": pj.get("cfft_per_kernel"), "lib_sha16": pj.get("lib_sha16"),
           "calibration_true_over_counter": pj.get("calibration_true_over_counter"),
           "algorithmic_bytes_per_launch": pj.get("algorithmic_bytes_per_launch"), "seconds": round(time.perf_counter() - t0, 1)}
    return pj.get("hbm_bytes_per_launch"), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): the fixed 256-column trace of config 5 sharded over the GPUs; weak: 32 columns per GPU")
    ap.add_argument("--total-cols", type=int, default=TOTAL_COLS, help="trace columns in all (strong scaling)")
    ap.add_argument("--cols", type=int, default=TREE_COLS, help="columns per GPU (weak scaling)")
    ap.add_argument("--tree-cols", type=int, default=TREE_COLS, help="columns per Merkle tree")
    ap.add_argument("--log-size", type=int, default=LOG_SIZE)
    ap.add_argument("--spinup-ms", type=float, default=SPINUP_MS, help="load (ms) the part must have seen before the headline timed region")
    ap.add_argument("--cpu-cols", type=int, default=4, help="columns in the one-thread CPU-oracle sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 1-4 (the `configs` list of the JSON line)")
    ap.add_argument("--commit-per-tree", action="store_true", help="one tstwo_merkle_commit call per tree instead of one tstwo_merkle_commit_many per step (A/B)")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the from-host legs (registered-memory upload rate, pipelined upload + step)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the child rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--pmc-json", default=None, help="tools/pmc_summary.py output of an earlier --pmc run of THIS build, instead of the child passes")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be the launcher (nothing above imported torch or the GPU library)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line for another N")
    n, tree_cols = args.log_size, args.tree_cols
    N = 1 << n
    total_cols = args.total_cols if args.scaling == "strong" else args.cols * world
    if total_cols % (world * tree_cols):
        raise SystemExit(f"bench.py: {total_cols} columns do not split into {tree_cols}-column trees over {world} GPUs")

    # ---- HBM counters of the launches this run times, from a child process, before this one touches the GPU (N = 1 only)
    # (N > 1: rank 0 collects them for ITS shard's launch shape before the rendezvous; the other ranks wait there and have not
    # touched a GPU either)
    traffic, traffic_source = None, {"reason": "not collected (--no-pmc or not rank 0)"}
    if rank == 0 and not args.no_pmc and not args.pmc_json:
        traffic, traffic_source = pmc_traffic(n, total_cols // world)

    # torch.distributed is control plane only (rendezvous, barrier, max-reduce of the elapsed time, hand-over of the RCCL
    # unique id) on the gloo backend; the one collective on the data path — the all-gather of Merkle roots — is RCCL
    # over xGMI issued through the library's own C ABI (tstwo_comm_init / tstwo_allgather_async), exactly what a Bun
    # host would call.  TSTWO_BENCH_COLLECTIVE=gloo rehearses the N > 1 control flow on a one-GPU box (all ranks share
    # GPU 0, where RCCL refuses several ranks per device): roots then travel through host memory.
    # Nothing below initialises the GPU before the rendezvous (torch.cuda.device_count() does not).
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)
    collective = os.environ.get("TSTWO_BENCH_COLLECTIVE", "rccl")
    use_dist = world > 1 or bool(os.environ.get("TSTWO_FORCE_DIST"))   # FORCE: rehearse the collective path at world size 1
    saved_stdout = None
    if use_dist:
        # gloo announces its connections on the C-level stdout ("[Gloo] Rank 0 is connected to ..."): route fd 1 to stderr until the
        # JSON line is printed, so that stdout carries that ONE line and nothing else
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))

    from tstwo_amd import _lib as L
    from tstwo_amd.backend import HipBackend, shard_columns

    L.init(dev_index)
    if use_dist and collective == "rccl":
        # one node by contract: RCCL's bootstrap sockets on the loopback interface (the container's other interfaces / host name
        # may not resolve; the data path between the GPUs is xGMI peer-to-peer either way)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        ids = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            L.call("tstwo_comm_unique_id", buf)
            ids[0] = bytes(buf)
        dist.broadcast_object_list(ids, src=0)
        L.call("tstwo_comm_init", rank, world, (C.c_uint8 * 128).from_buffer_copy(ids[0]))
    backend = HipBackend()

    # ---- the rank's shard of the trace columns, resident in HBM; trees = consecutive groups of tree_cols of them
    my_cols = shard_columns(total_cols, world, rank)
    n_cols = len(my_cols)
    n_trees = n_cols // tree_cols
    want_cpu = rank == 0 and not args.no_cpu          # the oracle leg: the WHOLE trace on rank 0's host cores, whatever N is
    cols_host = splitmix_columns([100 + c for c in my_cols], N)
    dev_cols = []
    L.sync()
    t_up = time.perf_counter()
    for h in cols_host:
        b = L.DeviceBuffer(4 * N)
        b.upload(h)
        dev_cols.append(b)
    L.sync()
    t_up = time.perf_counter() - t_up                      # host -> device hand-over of the rank's columns (tstwo_col_upload from pageable numpy arrays): reported, never in `value`
    want_host = rank == 0 and world == 1 and not args.no_host_boundary       # the from-host legs below need the host copies too
    if not want_cpu and not want_host:
        cols_host = None
    col_ptrs = L.ptr_array([b.ptr for b in dev_cols])
    tree_ptrs = [L.ptr_array([b.ptr for b in dev_cols[t * tree_cols:(t + 1) * tree_cols]]) for t in range(n_trees)]
    half_initial = backend.canonic_half_coset_initial(n)
    tw = L.DeviceBuffer(4 * (N // 2))
    L.call("tstwo_twiddles_build", half_initial, n - 1, C.c_void_p(tw.ptr), C.c_void_p(0))
    layers = [L.DeviceBuffer(32 * ((2 << n) - 1)) for _ in range(n_trees)]
    log_sizes = L.u32x([n] * tree_cols)
    # the rank's trees as ONE tstwo_merkle_commit_many request table (a TreeVec committed together: equally shaped trees share their
    # launches, so the latency-bound top of the trees runs side by side); every tree gets exactly what tstwo_merkle_commit writes
    commit_reqs = (L.CommitRequest * n_trees)()
    for t in range(n_trees):
        commit_reqs[t] = L.CommitRequest(tree_ptrs[t], log_sizes, tree_cols, layers[t].ptr)
    # two send/receive slots; the all-gather of step k runs on the library's collective stream and overlaps the CFFT of step k+1
    # (tstwo_comm_wait at the next issue point makes the main stream wait for it on the device, never the host)
    rec = 32 * n_trees                                   # this rank's record: its trees' roots in TreeVec order
    root_slot = [L.DeviceBuffer(rec) for _ in range(2)]
    roots_all = [L.DeviceBuffer(rec * world) for _ in range(2)]
    roots_host = [np.zeros(rec * world, dtype=np.uint8) for _ in range(2)]
    step_no = [0]
    L.sync()

    # HIP events on the library's stream, three per timed step, read after the timed region
    evs = [[L.Event() for _ in range(3)] for _ in range(args.steps)]

    def my_roots() -> bytes:
        return b"".join(bytes(l.download(np.uint8, 32).tobytes()) for l in layers)

    def step(ev):
        if ev:
            ev[0].record()
        L.call("tstwo_cfft_evaluate", col_ptrs, n_cols, n, half_initial, C.c_void_p(tw.ptr), n - 1)
        if ev:
            ev[1].record()
        if args.commit_per_tree:
            for t in range(n_trees):
                L.call("tstwo_merkle_commit", tree_ptrs[t], log_sizes, tree_cols, C.c_void_p(layers[t].ptr), None)
        else:
            L.call("tstwo_merkle_commit_many", commit_reqs, n_trees, None)
        if ev:
            ev[2].record()
        if use_dist:   # the only exchange on the path: 32-byte roots over RCCL/xGMI
            k = step_no[0] & 1
            step_no[0] += 1
            if collective == "rccl":
                L.call("tstwo_comm_wait")                  # the previous step's collective (it had this whole step to finish)
                for t in range(n_trees):
                    L.call("tstwo_copy", C.c_void_p(root_slot[k].ptr + 32 * t), C.c_void_p(layers[t].ptr), 32)
                L.call("tstwo_allgather_async", C.c_void_p(root_slot[k].ptr), C.c_void_p(roots_all[k].ptr), rec)
            else:                                           # rehearsal: host round trip + gloo
                mine = torch.from_numpy(np.frombuffer(my_roots(), dtype=np.uint8).copy())
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

    def timed(k, events=None):
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step(events[i] if events else None)
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Correctness of the measured workload: the first (untimed) step runs on the fresh synthetic input; its roots are
    # compared below with the CPU oracle's roots of the same columns (cpu_baseline.roots) -> "root_match".
    step(None)
    gpu_roots_first = my_roots()
    for _ in range(max(args.warmup - 1, 0)):
        step(None)
    cold = timed(args.steps)                       # K steps from idle clocks: reported as ms_per_step_cold
    # spin-up: as many further steps as the remaining load time needs at the cold step time (`cold` is already the maximum over
    # ranks, so every rank runs the same number of steps and of collectives)
    need_ms = args.spinup_ms - cold * 1e3
    spin = int(np.ceil(need_ms / (cold * 1e3 / args.steps))) if need_ms > 0 else 0
    for _ in range(spin):
        step(None)
    elapsed = timed(args.steps, evs)               # the headline: EXACTLY K steps between two barriers, max over ranks
    t_cfft = sum(e[0].elapsed_ms(e[1]) for e in evs)
    t_merkle = sum(e[1].elapsed_ms(e[2]) for e in evs)

    if use_dist:
        # every rank's roots must have arrived in rank order: compare the RCCL result with a gloo all-gather of the same roots
        last = (step_no[0] - 1) & 1
        got = roots_all[last].download(np.uint8, rec * world) if collective == "rccl" else roots_host[last]
        mine = torch.from_numpy(np.frombuffer(my_roots(), dtype=np.uint8).copy())
        ref = torch.zeros(rec * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(ref, mine)
        assert bytes(got.tobytes()) == bytes(ref.numpy().tobytes()), "all-gathered roots differ from the ranks' own roots"
        first = torch.from_numpy(np.frombuffer(gpu_roots_first, dtype=np.uint8).copy())
        allfirst = torch.zeros(rec * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(allfirst, first)
        tree_roots_first = bytes(allfirst.numpy().tobytes())
    else:
        tree_roots_first = gpu_roots_first

    # ---- the host boundary (never in `value`): what a caller pays who hands the trace over from HOST memory.
    #  (i) page-locked source (the caller's arrays registered with tstwo_host_register) -> tstwo_upload_async: one DMA per column;
    #  (ii) a whole step FROM HOST memory, pipelined: upload of tree k+1's columns on the copy stream under evaluate + commit of
    #       tree k on the main stream (tstwo_upload_async / tstwo_upload_fence) — against max(upload, compute) of the same run.
    host_legs = None
    if want_host:
        t0 = time.perf_counter()
        for h in cols_host:
            L.host_register(h)
        t_reg = time.perf_counter() - t0
        # (the FIRST copy out of freshly registered memory also pays for mapping its pages into the device: 37 GB/s on a box whose
        # steady rate is 55 — both are reported, the steady one is the rate of the hand-over)
        t_pin_first = None
        for attempt in range(2):
            L.sync()
            t0 = time.perf_counter()
            for h, b in zip(cols_host, dev_cols):
                b.upload_async(h)
            L.upload_wait()
            t_pin = time.perf_counter() - t0
            if t_pin_first is None:
                t_pin_first = t_pin

        def step_from_host():
            for c in range(tree_cols):
                dev_cols[c].upload_async(cols_host[c])
            for t in range(n_trees):
                L.upload_fence()                                         # tree t's columns have landed before its transform starts
                for c in range((t + 1) * tree_cols, min((t + 2) * tree_cols, n_cols)):
                    dev_cols[c].upload_async(cols_host[c])               # tree t + 1 travels under tree t's kernels
                L.call("tstwo_cfft_evaluate", tree_ptrs[t], tree_cols, n, half_initial, C.c_void_p(tw.ptr), n - 1)
                L.call("tstwo_merkle_commit", tree_ptrs[t], log_sizes, tree_cols, C.c_void_p(layers[t].ptr), None)
            L.sync()
        step_from_host()
        roots_from_host = my_roots()
        L.sync()
        t0 = time.perf_counter()
        step_from_host()
        t_pipe = time.perf_counter() - t0
        # the same per-tree step with the columns already resident: the compute side of the pipeline
        for h, b in zip(cols_host, dev_cols):
            b.upload_async(h)
        L.sync()
        t0 = time.perf_counter()
        for t in range(n_trees):
            L.call("tstwo_cfft_evaluate", tree_ptrs[t], tree_cols, n, half_initial, C.c_void_p(tw.ptr), n - 1)
            L.call("tstwo_merkle_commit", tree_ptrs[t], log_sizes, tree_cols, C.c_void_p(layers[t].ptr), None)
        L.sync()
        t_comp = time.perf_counter() - t0
        for h in cols_host:
            L.host_unregister(h)
        host_legs = {"register_seconds": t_reg, "h2d_seconds_registered": t_pin, "h2d_GBps_registered": 4.0 * N * n_cols / t_pin / 1e9,
                     "h2d_GBps_registered_first_pass": 4.0 * N * n_cols / t_pin_first / 1e9,
                     "pipelined_step_ms": t_pipe * 1e3, "upload_alone_ms": t_pin * 1e3, "compute_alone_ms": t_comp * 1e3,
                     "pipelined_over_max": t_pipe / max(t_pin, t_comp), "roots_match_resident_step": roots_from_host == gpu_roots_first,
                     "pcie_inclusive_elems_per_s_pipelined": n_cols * N / t_pipe,
                     "how": "tstwo_host_register on the numpy columns; tstwo_upload_async per column on the copy stream; tree k+1's columns "
                            "travel under evaluate + commit of tree k (tstwo_upload_fence in front of each tree)"}
        if not host_legs["roots_match_resident_step"]:
            raise SystemExit("bench.py: the step fed from host memory gave other Merkle roots than the resident step")

    if rank == 0:
        steps = args.steps
        total_elems = total_cols * N * steps
        cfft_ms = t_cfft / steps
        merkle_ms = t_merkle / steps
        # dominant kernel: the CFFT pass kernels (fast::k_cfft_a<INV,K>, fast::k_cfft_b<INV,LOGT>), (passes) launches per step over all columns.
        # (launch count from the planner itself: tstwo_cfft_plan_passes — 2 at n = 22 .. 24, 1 up to n = 14)
        n_passes = C.c_uint32(0)
        L.call("tstwo_cfft_plan_passes", n, n_cols, C.byref(n_passes))
        passes = int(n_passes.value)
        algo_bytes_transform = 8.0 * N * n_cols                     # 8*N per column transform
        algo_bytes_launch = algo_bytes_transform / passes
        launch_ms = cfft_ms / passes
        achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9
        merkle_bytes = (4.0 * tree_cols + 64.0) * N * n_trees       # per tree 4*C*N read + 64*N written
        import hashlib
        lib_now = hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16]
        if args.pmc_json and os.path.exists(args.pmc_json):        # counters of an earlier run of this build, handed in
            pj = json.load(open(args.pmc_json))
            traffic = pj.get("hbm_bytes_per_launch")
            traffic_source = {"file": os.path.relpath(os.path.abspath(args.pmc_json), ROOT), "kernels": pj.get("cfft_kernels"),
                              "lib_sha16": pj.get("lib_sha16")}
        if traffic is not None:
            traffic_source["same_build"] = traffic_source.get("lib_sha16") == lib_now
            if not traffic_source["same_build"] or traffic_source.get("algorithmic_bytes_per_launch", algo_bytes_launch) != algo_bytes_launch:
                traffic = None          # counters of another build or another launch shape say nothing about this one
        valu_rate = VALU_PER_BUTTERFLY * n_cols * n * (N // 2) / (cfft_ms * 1e-3)
        out = {
            "metric": "M31 CFFT elems/sec at log_size=22 (per step: CFFT evaluate + Blake2s Merkle commit + root all-gather)",
            "value": total_elems / elapsed,
            "unit": "elems/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "spinup_steps": spin,
            "cold_steps": steps,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_cold": cold / steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 5: {total_cols} columns x 2^{n}, column-shard x{world} "
                                   f"({n_cols} columns = {n_trees} trees of {tree_cols} per GPU; {total_cols // tree_cols} trees in all), "
                                   f"CircleDomain of CanonicCoset({n}); evaluate in one call + one Merkle tree per {tree_cols} columns"
                                   + (" + RCCL all-gather of the roots (C ABI: tstwo_allgather_async)" if world > 1 else ""),
                       "log_size": n, "total_columns": total_cols, "columns_per_gpu": n_cols, "trees_per_gpu": n_trees,
                       "columns_per_tree": tree_cols, "parallelism": f"column-shard x{world}"},
            "cfft_ms": cfft_ms,
            "cfft_butterflies_per_s": n_cols * n * (N // 2) / (cfft_ms * 1e-3),
            "cfft_elems_per_s": n_cols * N / (cfft_ms * 1e-3),
            "merkle_ms": merkle_ms,
            "merkle_GBps": merkle_bytes / (merkle_ms * 1e-3) / 1e9,
            "merkle_frac_of_hbm_peak": merkle_bytes / (merkle_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "roofline": {"bound": "hbm", "kernel": "fast::k_cfft_a<false,9,0,15> + fast::k_cfft_b<false,13,false> (the two passes of one transform)" if n == 22 else f"the {passes} CFFT pass kernel(s) of one transform", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "launches_per_step": passes, "avg_launch_ms": launch_ms,
                         "algorithmic_bytes_per_launch": algo_bytes_launch, "lib_sha16": lib_now,
                         # the lane-op rate is reported next to the HBM fraction.  11.3-12.2 VALU instructions
                         # per butterfly (profiles/r02_sq_counters.json); one issue port: nominal 39.3e12 lane-ops/s
                         # = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz, measured 35.3e12; with the second port taking the light
                         # VOP2s of another wave (priority phases) the same instruction mix peaks at 60.3e12.
                         "valu": {"instr_per_butterfly": VALU_PER_BUTTERFLY, "peak_lane_ops_per_s": VALU_PEAK,
                                  "achieved_lane_ops_per_s": valu_rate, "frac": valu_rate / VALU_PEAK,
                                  "measured_peak_lane_ops_per_s": VALU_PEAK_MEASURED,
                                  "frac_of_measured_peak": valu_rate / VALU_PEAK_MEASURED,
                                  "dual_issue_peak_lane_ops_per_s": VALU_PEAK_DUAL_MEASURED,
                                  "frac_of_dual_issue_peak": valu_rate / VALU_PEAK_DUAL_MEASURED}},
            # What a caller pays who hands the trace over from host memory on every step (the design does not: columns stay
            # resident from evaluate to the folds).  Measured on this run's own upload of the rank's columns; never `value`.
            "host_boundary": dict({"h2d_seconds": t_up, "h2d_bytes": 4.0 * N * n_cols, "h2d_GBps": 4.0 * N * n_cols / t_up / 1e9,
                                   "source": "pageable numpy arrays through tstwo_upload, one call per column",
                                   "pcie_inclusive_elems_per_s": n_cols * N / (t_up + elapsed / steps)}, **(host_legs or {})),
            "device": L.device_name(),
        }
        root_ok = None
        if want_cpu:
            # rank 0, every N: the oracle transforms and commits the WHOLE fixed trace (at N > 1 the other ranks' columns are
            # generated here from their seeds), so cpu_baseline is the N = 1 figure of this box and its roots check all the
            # first-step roots the ranks all-gathered — the TreeVec's roots do not depend on N (pcs/prover.ts:62-64,227-228)
            del dev_cols[:], layers[:]
            if world > 1:
                all_cols = [c for r in range(world) for c in shard_columns(total_cols, world, r)]
                mine = dict(zip(my_cols, cols_host))
                others = [c for c in all_cols if c not in mine]
                mine.update(zip(others, splitmix_columns([100 + c for c in others], N)))
                cols_host = [mine[c] for c in all_cols]
            out["cpu_baseline"] = cpu_baseline(cols_host, n, tree_cols, min(args.cpu_cols, n_cols))
            root_ok = "".join(out["cpu_baseline"]["roots"]) == tree_roots_first.hex()
            cols_host = None
        else:
            out["cpu_baseline"] = None
        out["gpu_roots"] = [tree_roots_first[32 * t:32 * t + 32].hex() for t in range(len(tree_roots_first) // 32)]
        out["root_match"] = root_ok            # GPU roots of the first step == CPU oracle roots of the same input (None: oracle leg not run)
        if world == 1 and not args.no_configs:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from bench_configs import run_configs
            del dev_cols[:], layers[:]
            out["configs"] = run_configs(reps=10, no_cpu=args.no_cpu)      # BASELINE configs 1-4, same JSON line
            # ... and the callers either side of the hot path at config 5's size (pcs/prover.ts:26-252): the trace committed as
            # 8 trees on the blown-up domain, and an opening proof over it (tools/bench_config5_callers.py)
            if n == LOG_SIZE and total_cols == TOTAL_COLS and args.scaling == "strong":
                from bench_config5_callers import run_config5_callers
                out["configs"] += run_config5_callers(reps=3, total_cols=total_cols, n=n, tree_cols=tree_cols)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
        if root_ok is False:
            raise SystemExit("bench.py: the GPU Merkle roots of the step differ from the CPU oracle's: the measured numbers are void")

    if use_dist:
        dist.barrier()
        if collective == "rccl":
            L.call("tstwo_comm_destroy")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
