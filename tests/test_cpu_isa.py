"""ISA properties of the built hot kernels (CPU-only: the gfx950 code objects are taken out of the in-tree object files and
disassembled with llvm-objdump; nothing is executed).  These are the properties DESIGN.md 4/4.1/4.2 rest on:
  * the CFFT pass kernels and the Blake2s kernels issue their butterflies / G functions in priority phases (s_setprio),
    every light run is made of VOP2 add/sub/xor/shift only and no heavy VALU instruction sits inside one;
  * column data is accessed with global_* instructions (a flat_* access also counts on lgkmcnt and would make the LDS-only
    barriers wait for in-flight prefetches);
  * the four headline CFFT pass kernels (2^13 bottom pass, 9-layer strided pass, both directions) and the Blake2s leaf / inner
    kernels use no scratch memory at all (no scratch_load / scratch_store in their ISA)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tstwo_amd", "csrc", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
LIGHT = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32")


def _disasm(tu, tmp_path):
    obj = os.path.join(OBJ, tu + ".o")
    if not os.path.exists(obj):
        pytest.skip("library objects not built (python -m tstwo_amd.build)")
    if not (shutil.which("objcopy") and os.path.exists(os.path.join(LLVM, "clang-offload-bundler"))):
        pytest.skip("binutils / ROCm LLVM tools not available")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    kernels, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip() and not line.startswith("Disassembly"):
            ins = line.split("//")[0].strip()
            if ins:
                kernels[name].append(ins)
    return kernels


def _phases(ins_list):
    """[(priority, [valu instructions])] in program order, starting at the first s_setprio."""
    out, cur = [], None
    for ins in ins_list:
        op = ins.split()[0]
        if op == "s_setprio":
            cur = (int(ins.split()[1]), [])
            out.append(cur)
        elif cur is not None and op.startswith("v_"):
            cur[1].append(ins)
    return out


def _is_light(ins):
    op, args = ins.split(None, 1)
    op = op.replace("_e32", "")
    return op in LIGHT and not re.search(r"\bs\d+\b|\bs\[|0x[0-9a-f]{3,}|vcc|exec", args)


@pytest.mark.parametrize("tu,pattern,min_phases,min_light_share", [
    ("cfft", r"k_cfft_bILb0ELi13ELb0E", 40, 0.9), ("cfft", r"k_cfft_bILb1ELi13ELb0E", 40, 0.9),
    ("cfft", r"k_cfft_aILb0ELi9ELi0ELi14E", 30, 0.9), ("cfft", r"k_cfft_aILb1ELi9ELi0ELi14E", 30, 0.9),
    # round 4: the 10-layer strided pass on the 2^15-word tile (n = 24 in two passes), both directions and the fused extension
    ("cfft", r"k_cfft_aILb0ELi9ELi0ELi15E", 60, 0.9), ("cfft", r"k_cfft_aILb1ELi9ELi0ELi15E", 60, 0.9),       # the headline strided pass since round 4 (n = 22: 13 + 9)
    ("cfft", r"k_cfft_aILb0ELi10ELi0ELi15E", 60, 0.9), ("cfft", r"k_cfft_aILb1ELi10ELi0ELi15E", 60, 0.9), ("cfft", r"k_cfft_aILb0ELi10ELi2ELi15E", 50, 0.9),
    ("merkle", r"k_merkle_leaf_staticILi2E", 300, 0.9), ("merkle", r"k_merkle_innerE", 150, 0.9),
    # round 3: the 8-rows-per-lane field kernels on field8.cuh (quotients, QM31 batch inverse through the norms).  Their
    # priority-0 stretches also hold what cannot be phased — the one Fermat chain per 8 values and the tree products around it
    # (f8::inverse8), issued in program order at low priority on purpose — hence the lower share of light instructions there.
    ("quotients", r"k_quotients8ILb1ELb0E", 80, 0.7), ("quotients", r"k_quotients8ILb1ELb1E", 80, 0.7), ("quotients", r"k_quotients8_multiILi2ELb0E", 80, 0.65), ("field_ops", r"k_qm31_batch_inverse_normE", 60, 0.7),
])
def test_hot_kernels_are_phased_and_use_global_memory_instructions(tu, pattern, min_phases, min_light_share, tmp_path):
    kernels = _disasm(tu, tmp_path)
    names = [k for k in kernels if re.search(pattern, k)]
    assert len(names) == 1, names
    ins = kernels[names[0]]
    assert not [i for i in ins if i.startswith(("flat_load", "flat_store"))], "column data must not be accessed with flat_* instructions"
    assert not [i for i in ins if i.startswith("scratch_")], "the headline kernels must not spill (ScratchSize 0)"
    phases = _phases(ins)
    assert len(phases) >= min_phases, len(phases)
    light_runs = [p for p in phases if p[0] == 0 and len(p[1]) >= 4]
    heavy_runs = [p for p in phases if p[0] == 3]
    assert light_runs and heavy_runs
    # the light runs (priority 0) hold light VOP2 almost exclusively — the exceptions are a rematerialised modulus
    # (v_bfrev_b32 v, -2), the run that ends a layer sequence and is followed by address arithmetic, and the first round of a
    # compression, where the state still holds constants (literal operands)
    n_light = sum(sum(1 for i in p[1] if _is_light(i)) for p in light_runs)
    n_all = sum(len(p[1]) for p in light_runs)
    assert n_light >= min_light_share * n_all, (n_light, n_all)
    # and the heavy runs hold the multiplies / mins / rotates: no long stretch of light instructions at high priority
    for prio, run in heavy_runs:
        longest = cur = 0
        for i in run:
            cur = cur + 1 if _is_light(i) else 0
            longest = max(longest, cur)
        assert longest <= (12 if min_light_share >= 0.9 else 16), (names[0], longest)      # field8 kernels: + operand shuffles (v_mov) behind a run

