#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid, workgroup) count / avg / total in microseconds."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    files = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)
    d = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0][:48]
            key = (name, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Scratch_Size"])
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    tot = sum(sum(v) for v in d.values())
    print(f"{'kernel':48s} {'grid':>9s} {'wg':>4s} {'vgpr':>4s} {'scr':>4s} {'calls':>5s} {'avg_us':>9s} {'min_us':>9s} {'total_us':>10s} {'%':>5s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:48s} {k[1]:>9s} {k[2]:>4s} {k[3]:>4s} {k[4]:>4s} {len(v):5d} {sum(v)/len(v):9.1f} {min(v):9.1f} {sum(v):10.0f} {100*sum(v)/tot:5.1f}")


if __name__ == "__main__":
    main()
