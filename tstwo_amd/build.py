"""Builds libtstwo_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m tstwo_amd.build [--force] [--verbose] [--experiments]

--experiments builds the SECOND library, libtstwo_hip_exp.so (-DTSTWO_EXPERIMENTS): the same sources with the TSTWO_* tuning and
A/B switches of DESIGN.md §8 read from the environment (once).  The shipped library has them compiled out; tools/ and the few
tests that pin a non-default branch load the experiments build through TSTWO_HIP_LIB.

hipcc cross-compiles for gfx950 without a GPU, so this also runs in the GPU-less build container.
The shared library lands next to this file (git-ignored, but it travels to the GPU box with gpurun).
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libtstwo_hip.so")
OBJ_EXP = os.path.join(CSRC, "obj", "exp")
LIB_EXP = os.path.join(HERE, "libtstwo_hip_exp.so")
SOURCES = ["context.hip", "field_ops.hip", "cfft.hip", "fri.hip", "merkle.hip", "quotients.hip", "comm.hip"]
# every header under csrc/ (globbed: a new .cuh / .h marks all objects stale without having to be listed) + the public one
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))) + [os.path.join("..", "..", "include", "tstwo_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, verbose: bool, experiments: bool = False) -> str:
    obj = os.path.join(OBJ_EXP if experiments else OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _stale(obj, deps):
        cmd = [_hipcc(), *FLAGS, *(["-DTSTWO_EXPERIMENTS"] if experiments else []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    obj_dir, lib = (OBJ_EXP, LIB_EXP) if experiments else (OBJ, LIB)
    os.makedirs(obj_dir, exist_ok=True)
    if force:
        for f in os.listdir(obj_dir):
            if f.endswith(".o"):
                os.remove(os.path.join(obj_dir, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(7, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose, experiments), SOURCES))
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, experiments="--experiments" in sys.argv))
