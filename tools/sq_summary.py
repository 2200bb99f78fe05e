#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc SQ counter passes (one directory per pass) into per-kernel averages and the derived
VALU-issue time: SQ_INSTS_VALU wave-instructions x 4 cycles (wave64 on a 16-lane SIMD) / (256 CUs x 4 SIMDs) / 2.4 GHz.
usage: sq_summary.py DIR [DIR ...] > out.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

N_SIMD, CLK = 256 * 4, 2.4e9


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for root in sys.argv[1:]:
        for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if name.startswith("__amd") or name.startswith("at::"):
                    continue
                acc[f"{name} grid={r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "per-dispatch averages; SQ_*_CYCLES counters are summed over waves/SEs as rocprofv3 reports them",
           "kernels": {}}
    for k, ctrs in acc.items():
        d = {c: sum(v) / len(v) for c, v in ctrs.items()}
        if "SQ_INSTS_VALU" in d:
            d["valu_issue_us_at_2.4GHz"] = d["SQ_INSTS_VALU"] * 4 / N_SIMD / CLK * 1e6
            if d.get("SQ_WAVES"):
                d["valu_instr_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        out["kernels"][k] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
