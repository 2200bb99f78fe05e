#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic, applying the gfx950
corrections of MI355X_MICROARCH.md §HBM after calibrating them on kernels with known byte counts.

    pmc_summary.py FETCH_DIR WRITE_DIR [log_size cols] > out.json        (bench.py calls summarise() directly)"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict


def load(root, counter):
    out = defaultdict(list)
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                out[(name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return out


def summarise(fetch_dir, write_dir, log_size=22, cols=256):
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    ncal = (1 << 26) * 4.0
    res = {"unit": "bytes per launch", "kernels": {}}
    cal = {}
    for (name, grid), v in fetch.items():
        f = sum(v) / len(v) * 1024.0
        w = sum(write.get((name, grid), [0])) / max(len(write.get((name, grid), [1])), 1) * 1024.0
        if "binop_vec4" in name:
            cal["fetch_16B"] = 2 * ncal / f if f else None
            cal["write_16B"] = ncal / w if w else None
        if "binop_scalar" in name:
            cal["fetch_4B"] = 2 * ncal / f if f else None
            cal["write_4B"] = ncal / w if w else None
        res["kernels"][f"{name} grid={grid}"] = {"FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes_raw": w, "launches": len(v)}
    res["calibration_true_over_counter"] = cal
    k16 = cal.get("fetch_16B") or 2.0
    for k, d in res["kernels"].items():
        d["hbm_bytes_corrected"] = d["FETCH_SIZE_bytes_raw"] * k16 + d["WRITE_SIZE_bytes_raw"] * (cal.get("write_16B") or 1.0)
    cf = {k: d for k, d in res["kernels"].items() if "k_cfft" in k}
    if cf:
        res["hbm_bytes_per_launch"] = sum(d["hbm_bytes_corrected"] for d in cf.values()) / len(cf)   # avg over the transform's passes
        res["algorithmic_bytes_per_launch"] = 8.0 * (1 << log_size) * cols / len(cf)
        res["cfft_kernels"] = sorted(cf)
        res["cfft_per_kernel"] = {k: {"fetch": round(d["FETCH_SIZE_bytes_raw"] * k16), "write": round(d["WRITE_SIZE_bytes_raw"] * (cal.get("write_16B") or 1.0))}
                                  for k, d in cf.items()}
    # which build these counters belong to (bench.py puts this next to roofline.traffic)
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tstwo_amd", "libtstwo_hip.so")
    if os.path.exists(lib):
        res["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    return res


if __name__ == "__main__":
    extra = [int(x) for x in sys.argv[3:5]]
    print(json.dumps(summarise(sys.argv[1], sys.argv[2], *extra), indent=1))
