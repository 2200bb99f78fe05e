#!/usr/bin/env python3
"""Prints, for kernels of an assembly listing (hipcc -S --cuda-device-only), the instruction stream as one character per
instruction: l = light VALU (dual-issue port 1 capable: VOP2 add/sub/xor/and/or/shift/mov without literal or SGPR operands),
H = heavy VALU, |n| = s_setprio n, d = LDS, g = global memory, B = s_barrier, w = s_waitcnt, X = scratch, . = other.
    python tools/isa_phases.py /tmp/cfft.s k_cfft_bILb0ELi13ELb0E [k_cfft_a...]"""
import re
import sys

LIGHT = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32"}
LIGHT |= {x + "_e32" for x in LIGHT}
s = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r"^(\S*" + re.escape(pat) + r"\S*):", s, flags=re.M):
        i = m.end()
        body = s[i:s.index("s_endpgm", i)]
        seq = []
        for line in body.split("\n"):
            line = line.split(";")[0].strip()
            if not line or line.startswith("."):
                continue
            parts = line.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            if op.startswith("v_"):
                light = op in LIGHT and not re.search(r"\bs\d+\b|\bs\[|0x[0-9a-f]{3,}|vcc|exec", args)
                seq.append("l" if light else "H")
            elif op == "s_setprio":
                seq.append("|" + args + "|")
            elif op.startswith("ds_"):
                seq.append("d")
            elif op.startswith(("global_", "buffer_", "flat_")):
                seq.append("g")
            elif op == "s_barrier":
                seq.append("B")
            elif op == "s_waitcnt":
                seq.append("w")
            elif op.startswith("scratch_"):
                seq.append("X")
            else:
                seq.append(".")
        t = "".join(seq)
        print(m.group(1)[:90])
        print(t)
        print("heavy", t.count("H"), "light", t.count("l"), "scratch", t.count("X"), "setprio", t.count("|") // 2)
