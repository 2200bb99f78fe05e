#!/usr/bin/env python3
"""Target for `rocprofv3 --kernel-trace`: FriProver.commit (tstwo_fri_commit_layers) of one size, repeated; and, with
--timeline DIR, the kernel sequence of the LAST commit in a finished trace: start offset, duration, gap to the previous kernel."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))


def timeline(root, n_last):
    import csv, glob
    rows = []
    for f in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n_last:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0.0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name:44s} grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']}")
        busy += (e - s) / 1e3
        prev_end = e
    print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy:.1f} us, gaps {(prev_end - t0) / 1e3 - busy:.1f} us over {len(rows)} launches")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
        timeline(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    import numpy as np
    import tstwo_amd as T
    from tstwo_amd import _lib as L
    logd = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    L.init(0)
    rng = np.random.default_rng(0)
    blow = 2
    domain = T.CanonicCoset(logd + blow).circleDomain()
    tw = T.precompute_twiddles(domain.halfCoset)
    polys = [T.HipCirclePoly(rng.integers(0, T.P, size=1 << logd, dtype=np.uint32)) for _ in range(4)]
    evs = T.evaluate_polynomials(polys, domain, tw)
    col = T.SecureEvaluation(domain, T.SecureColumnByCoords([e.values for e in evs]))
    cfg = T.FriConfig(2, blow, 20)
    for _ in range(reps):
        T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw, device_channel=True)
    L.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        T.FriProver.commit(T.Blake2sChannel(), cfg, [col], tw, device_channel=True)
    L.sync()
    print(f"log {logd + blow}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per commit", flush=True)
