#!/usr/bin/env python3
"""FRI commit across sizes: host transcript, device transcript, and hipGraph replay of the device-transcript loop."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tstwo_amd as T
from tstwo_amd import _lib as L
L.init(0)
rng = np.random.default_rng(0)
for logd in (10, 14, 18, 20, 22):
    blow = 2
    domain = T.CanonicCoset(logd + blow).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(rng.integers(0, T.P, size=1 << logd, dtype=np.uint32)) for _ in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    col = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs]))
    cfg = T.FriConfig(2, blow, 20)
    for dev, host_loop in ((True, False), (True, True), (False, False)):
        if host_loop:
            os.environ["TSTWO_FRI_COMMIT_HOST_LOOP"] = "1"
        else:
            os.environ.pop("TSTWO_FRI_COMMIT_HOST_LOOP", None)
        tw0 = time.perf_counter()
        while time.perf_counter() - tw0 < 0.2:          # 200 ms of this very loop first: the clocks settle (DESIGN 4.1 "Clocks")
            T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw, device_channel=dev)
        L.sync(); t0 = time.perf_counter()
        for _ in range(20):
            T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw, device_channel=dev)
        L.sync()
        tag = "device transcript, " + ("per-layer calls from the host" if host_loop else "tstwo_fri_commit_layers") if dev else "host transcript"
        print(f"log {logd + blow}: FriProver.commit ({tag}): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
    os.environ.pop("TSTWO_FRI_COMMIT_HOST_LOOP", None)
    plan = T.FriCommitPlan(cfg, [col], tw)
    tw0 = time.perf_counter()
    while time.perf_counter() - tw0 < 0.2:
        plan.run(T.Blake2sChannel())
    L.sync(); t0 = time.perf_counter()
    for _ in range(10):
        plan.run(T.Blake2sChannel())
    L.sync()
    print(f"log {logd + blow}: FriCommitPlan.run (hipGraph replay):      {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
