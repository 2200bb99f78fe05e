"""Prints VGPR/SGPR/LDS/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os
import re
import subprocess
import sys

from .build import CSRC, FLAGS, SOURCES, _hipcc


def main():
    srcs = sys.argv[1:] or SOURCES
    for src in srcs:
        cmd = [_hipcc(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"]
        out = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = {}
        for line in out.splitlines():
            m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2).strip()
            if k == "Function Name":
                cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
            else:
                cur[k] = v
            if k.startswith("LDS Size"):
                print(f"{src:14s} {cur['name']:90s} vgpr={cur.get('VGPRs')} sgpr={cur.get('TotalSGPRs')} "
                      f"scratch={cur.get('ScratchSize [bytes/lane]')} occ={cur.get('Occupancy [waves/SIMD]')} lds={v}")


if __name__ == "__main__":
    main()
