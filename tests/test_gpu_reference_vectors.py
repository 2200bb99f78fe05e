"""-m gpu: the reference's own Rust-generated field vectors (tests/golden/*-test-vectors.json = the reference's
test-vectors/*.json, replayed the way test-equivalence/fields/*.test.ts does) fed THROUGH THE C ABI to the HIP kernels.

tests/test_oracle_fields.py pins the CPU oracle with the same vectors; this file pins the device path directly, so the
"bit-exact against test-vectors/*.json" claim does not rest on oracle == GPU transitivity.

How each vector family maps onto include/tstwo_hip.h (columns are SoA, one vector per row):
  m31  add/sub/mul/neg            -> tstwo_m31_{add,sub,mul,neg}
  m31  inverse, pow2147483645     -> tstwo_m31_batch_inverse (x^(P-2) IS the inverse, fields/m31.ts:305-326)
  m31  into_slice                 -> the device column's byte image (tstwo_upload / tstwo_download of a HipColumn)
  cm31 add/sub/neg                -> tstwo_m31_* on the real and the imaginary coordinate column
  cm31 mul                        -> tstwo_qm31_mul with both second CM31 coordinates zero ((a + 0u)(b + 0u) = ab)
  cm31 inverse                    -> tstwo_cm31_batch_inverse
  qm31 add                        -> tstwo_secure_accumulate (col += other) and tstwo_m31_add per coordinate
  qm31 sub/neg                    -> tstwo_m31_{sub,neg} per coordinate column
  qm31 mul, mul_cm31              -> tstwo_qm31_mul (mul_cm31: the CM31 factor embedded as (c, 0))
  qm31 inverse                    -> tstwo_qm31_batch_inverse
  securecolumn set_and_at/len/is_empty/to_vec/from_iter -> SecureColumnByCoords over device columns (tstwo_amd.backend)
Vectors with no device entry point (scalar constructors from_i32 / from_u32 / reduce / partial_reduce, zero/one/is_zero)
stay host-side and are covered by tests/test_oracle_fields.py and tests/test_cpu_host.py.
"""
import numpy as np
import pytest

from conftest import load_vectors

pytestmark = pytest.mark.gpu

from tstwo_amd import _lib as L  # noqa: E402
from tstwo_amd.backend import HipColumn, SecureColumnByCoords  # noqa: E402
from gpu_util import dev, dev_empty, host, p4, vp  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _init():
    L.init(0)
    yield
    L.sync()


def u32(xs):
    return np.array(list(xs), dtype=np.uint32)


def run_binop(name, a, b=None):
    n = len(a)
    da, do = dev(u32(a)), dev_empty(n)
    if b is None:
        L.call(name, vp(da), vp(do), n)
    else:
        db = dev(u32(b))
        L.call(name, vp(da), vp(db), vp(do), n)
    return host(do, n).tolist()


def qm31_mul_cols(a_rows, b_rows):
    """rows of 4 words -> rows of 4 words through tstwo_qm31_mul on SoA device columns."""
    n = len(a_rows)
    da = [dev(u32(r[k] for r in a_rows)) for k in range(4)]
    db = [dev(u32(r[k] for r in b_rows)) for k in range(4)]
    do = [dev_empty(n) for _ in range(4)]
    L.call("tstwo_qm31_mul", p4(da), p4(db), p4(do), n)
    cols = [host(do[k], n) for k in range(4)]
    return [[int(cols[k][i]) for k in range(4)] for i in range(n)]


def test_m31_vectors_through_the_c_abi():
    vs = load_vectors("m31")
    seen = 0
    for op in ("add", "sub", "mul"):
        sel = [v for v in vs if v["operation"] == op]
        assert len(sel) == 100
        got = run_binop(f"tstwo_m31_{op}", [v["inputs"]["a"] for v in sel], [v["inputs"]["b"] for v in sel])
        assert got == [v["output"] for v in sel], op
        seen += len(sel)
    sel = [v for v in vs if v["operation"] == "neg"]
    assert run_binop("tstwo_m31_neg", [v["inputs"]["a"] for v in sel]) == [v["output"] for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] in ("inverse", "pow2147483645")]
    assert len(sel) == 12
    xs = u32(v["inputs"]["value"] for v in sel)
    di, do = dev(xs), dev_empty(len(sel))
    L.call("tstwo_m31_batch_inverse", vp(di), vp(do), len(sel))
    assert host(do, len(sel)).tolist() == [v["output"] for v in sel]
    L.call("tstwo_m31_batch_inverse_async", vp(di), vp(do), len(sel))
    L.call("tstwo_check_zero_flag")
    assert host(do, len(sel)).tolist() == [v["output"] for v in sel]
    seen += len(sel)
    for v in vs:
        if v["operation"] == "into_slice":          # M31.intoSlice (fields/m31.ts:272-284) == the device column's bytes
            col = HipColumn(u32(v["inputs"]["elements"]))
            assert list(col.buf.download(np.uint8, 4 * col.len())) == v["output"]
            seen += 1
    assert seen == 413


def test_cm31_vectors_through_the_c_abi():
    vs = load_vectors("cm31")
    seen = 0
    for op in ("add", "sub"):
        sel = [v for v in vs if v["operation"] == op]
        re = run_binop(f"tstwo_m31_{op}", [v["inputs"]["a_real"] for v in sel], [v["inputs"]["b_real"] for v in sel])
        im = run_binop(f"tstwo_m31_{op}", [v["inputs"]["a_imag"] for v in sel], [v["inputs"]["b_imag"] for v in sel])
        assert list(zip(re, im)) == [(v["output"]["real"], v["output"]["imag"]) for v in sel], op
        seen += len(sel)
    sel = [v for v in vs if v["operation"] == "neg"]
    re = run_binop("tstwo_m31_neg", [v["inputs"]["real"] for v in sel])
    im = run_binop("tstwo_m31_neg", [v["inputs"]["imag"] for v in sel])
    assert list(zip(re, im)) == [(v["output"]["real"], v["output"]["imag"]) for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] == "mul"]
    got = qm31_mul_cols([[v["inputs"]["a_real"], v["inputs"]["a_imag"], 0, 0] for v in sel],
                        [[v["inputs"]["b_real"], v["inputs"]["b_imag"], 0, 0] for v in sel])
    assert got == [[v["output"]["real"], v["output"]["imag"], 0, 0] for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] == "inverse"]
    n = len(sel)
    di = [dev(u32(v["inputs"][k] for v in sel)) for k in ("real", "imag")]
    do = [dev_empty(n), dev_empty(n)]
    L.call("tstwo_cm31_batch_inverse", L.P2(di[0].ptr, di[1].ptr), L.P2(do[0].ptr, do[1].ptr), n)
    assert list(zip(host(do[0], n).tolist(), host(do[1], n).tolist())) == [(v["output"]["real"], v["output"]["imag"]) for v in sel]
    seen += n
    for v in vs:
        if v["operation"] == "into_slice":
            flat = u32(x for e in v["inputs"]["elements"] for x in (e["real"], e["imag"]))
            b = dev(flat)
            assert list(b.download(np.uint8, flat.nbytes)) == v["output"]
            seen += 1
    assert seen == 206


def test_qm31_vectors_through_the_c_abi():
    vs = load_vectors("qm31")
    seen = 0
    for op in ("add", "sub"):
        sel = [v for v in vs if v["operation"] == op]
        cols = [run_binop(f"tstwo_m31_{op}", [v["inputs"]["a"][k] for v in sel], [v["inputs"]["b"][k] for v in sel]) for k in range(4)]
        assert [[cols[k][i] for k in range(4)] for i in range(len(sel))] == [v["output"] for v in sel], op
        seen += len(sel)
    # add once more through AccumulationOps.accumulate (col += other)
    sel = [v for v in vs if v["operation"] == "add"]
    da = [dev(u32(v["inputs"]["a"][k] for v in sel)) for k in range(4)]
    db = [dev(u32(v["inputs"]["b"][k] for v in sel)) for k in range(4)]
    L.call("tstwo_secure_accumulate", p4(da), p4(db), len(sel))
    assert [[int(host(da[k], len(sel))[i]) for k in range(4)] for i in range(len(sel))] == [v["output"] for v in sel]
    sel = [v for v in vs if v["operation"] == "neg"]
    cols = [run_binop("tstwo_m31_neg", [v["inputs"]["value"][k] for v in sel]) for k in range(4)]
    assert [[cols[k][i] for k in range(4)] for i in range(len(sel))] == [v["output"] for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] == "mul"]
    assert qm31_mul_cols([v["inputs"]["a"] for v in sel], [v["inputs"]["b"] for v in sel]) == [v["output"] for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] == "mul_cm31"]
    assert qm31_mul_cols([v["inputs"]["qm31"] for v in sel], [list(v["inputs"]["cm31"]) + [0, 0] for v in sel]) == [v["output"] for v in sel]
    seen += len(sel)
    sel = [v for v in vs if v["operation"] == "inverse"]
    n = len(sel)
    di = [dev(u32(v["inputs"]["value"][k] for v in sel)) for k in range(4)]
    do = [dev_empty(n) for _ in range(4)]
    L.call("tstwo_qm31_batch_inverse", p4(di), p4(do), n)
    assert [[int(host(do[k], n)[i]) for k in range(4)] for i in range(n)] == [v["output"] for v in sel]
    seen += n
    for v in vs:
        if v["operation"] == "into_slice":
            flat = u32(x for e in v["inputs"]["elements"] for x in e)
            assert list(dev(flat).download(np.uint8, flat.nbytes)) == v["output"]
            seen += 1
    assert seen == 125


def test_securecolumn_vectors_on_device_columns():
    """securecolumn-test-vectors.json (BASELINE config 3 names it) against SecureColumnByCoords = 4 device columns."""
    vs = load_vectors("securecolumn")
    col = SecureColumnByCoords.zeros(5)
    seen = 0
    for v in vs:
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op == "set_and_at":
            col.set(i["index"], i["value"])
            assert list(col.at(i["index"]).to_m31_array()) == out
        elif op == "len":
            assert SecureColumnByCoords.zeros(i["column_size"]).len() == out
        elif op == "is_empty":
            assert SecureColumnByCoords.zeros(i["column_size"]).isEmpty() == out
        elif op == "to_vec":
            c = SecureColumnByCoords.from_(i["column_values"])
            assert [[int(x[k]) for k in range(4)] for x in zip(*c.to_numpy())] == out
            # after the five set_and_at vectors the column built by set() holds the same rows
            assert [[int(x[k]) for k in range(4)] for x in zip(*col.to_numpy())] == out
        elif op == "from_iter":
            c = SecureColumnByCoords.from_(i["input_values"])
            assert [[int(cc.at(r).value) for cc in c.columns] for r in range(c.len())] == out
        else:
            raise AssertionError(op)
        seen += 1
    assert seen == 10
