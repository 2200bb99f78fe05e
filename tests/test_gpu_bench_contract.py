"""-m gpu: bench.py run as the driver runs it (a subprocess, one JSON line on stdout), at a reduced size: the contract's keys, the
root check against the oracle inside the run, and the N = 2 control flow rehearsed with two processes on the one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--log-size", "14", "--total-cols", "64", "--steps", "3", "--warmup", "1", "--spinup-ms", "0", "--no-pmc", "--no-configs", "--cpu-cols", "2"]


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout, nothing else
    return json.loads(lines[0])


def test_bench_line_contract_at_reduced_size():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "host_boundary"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "u32" and d["scaling"] == "strong" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(64 * (1 << 14) / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["traffic"] is None                                      # --no-pmc: no counters, and the line says so
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and len(c["roots"]) == 2
    assert d["root_match"] is True and d["gpu_roots"] == c["roots"]    # the GPU's roots of the first step == the oracle's
    hb = d["host_boundary"]
    assert hb["h2d_GBps"] > 0 and hb["h2d_GBps_registered"] > 0 and hb["roots_match_resident_step"] is True and hb["pipelined_step_ms"] > 0


def test_bench_two_ranks_rehearsed_on_one_gpu():
    env = dict(os.environ, TSTWO_BENCH_COLLECTIVE="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + ["--no-cpu"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["columns_per_gpu"] == 32 and d["config"]["parallelism"] == "column-shard x2"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--no-cpu"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    assert _line(one.stdout)["gpu_roots"] == d["gpu_roots"]         # the TreeVec's roots do not depend on the GPU count


def test_bench_gpus_2_starts_its_own_ranks_and_checks_every_root():
    """`python bench.py --gpus 2` with no launcher around it — the form the driver uses — must start two ranks itself, print
    n_gpus 2 and check the all-gathered first-step roots of BOTH ranks against the oracle (root_match), with a cpu_baseline."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TSTWO_BENCH_COLLECTIVE"] = "gloo"               # one GPU here: both ranks share it, roots travel through the host
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["columns_per_gpu"] == 32 and d["config"]["parallelism"] == "column-shard x2"
    assert d["root_match"] is True and len(d["gpu_roots"]) == 2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["roots"] == d["gpu_roots"] and c["cores"] >= 1
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--no-cpu"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    assert _line(one.stdout)["gpu_roots"] == d["gpu_roots"]         # the same trees whatever N is


def test_bench_rccl_collective_path_at_world_size_one():
    """TSTWO_FORCE_DIST=1: the N > 1 code path (RCCL communicator through the C ABI, tstwo_allgather_async on the collective stream,
    the gathered roots compared with the ranks' own) with a world of one on the one GPU — including the from-host legs, which run
    after that comparison and must not disturb it (round 4: they once re-committed fresh columns in front of it)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TSTWO_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["root_match"] is True and d["host_boundary"]["roots_match_resident_step"] is True
