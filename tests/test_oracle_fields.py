"""Pins the CPU oracle's field arithmetic against the reference's Rust-generated vectors
(test-vectors/*.json, replayed the way test-equivalence/fields/*.test.ts does) and the constants ported
from the Rust unit tests (packages/core/test/fields/{m31,cm31,qm31}.test.ts)."""
import numpy as np
import pytest

from conftest import P, load_vectors
from oracle import oracle as orc

L = orc.lib()


def cm(d):
    return orc.CM31(d["real"], d["imag"])


def test_m31_vectors():
    n = 0
    for v in load_vectors("m31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op in ("add", "sub", "mul"):
            assert getattr(L, f"orc_m31_{op}")(i["a"], i["b"]) == out
            if op == "mul":
                assert i["a"] * i["b"] == v["intermediates"]["product_u64"]
        elif op == "neg":
            assert L.orc_m31_neg(i["a"]) == out
        elif op == "from_u32_unchecked":
            assert i["value"] == out
        elif op == "from_i32":
            assert L.orc_m31_from_i32(i["value"]) == out
        elif op == "from_u32":
            assert L.orc_m31_from_u32(i["value"]) == out
        elif op == "partial_reduce":
            assert L.orc_m31_partial_reduce(i["value"]) == out
        elif op == "reduce":
            assert L.orc_m31_reduce(int(i["value"])) == out
        elif op == "inverse":
            r = orc.C.c_uint32()
            assert L.orc_m31_inverse(i["value"], r) == 0 and r.value == out
            assert L.orc_m31_mul(r.value, i["value"]) == 1
        elif op == "pow2147483645":
            assert L.orc_m31_pow2147483645(i["value"]) == out
        elif op == "into_slice":
            assert list(np.array(i["elements"], dtype="<u4").tobytes()) == out
        elif op in ("zero", "one", "is_zero", "complex_conjugate"):
            pass
        else:
            raise AssertionError(op)
        n += 1
    assert n == 452


def test_cm31_vectors():
    n = 0
    for v in load_vectors("cm31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op in ("add", "sub", "mul"):
            r = getattr(L, f"orc_cm31_{op}")(orc.CM31(i["a_real"], i["a_imag"]), orc.CM31(i["b_real"], i["b_imag"]))
            assert (r.a, r.b) == (out["real"], out["imag"])
        elif op == "neg":
            r = L.orc_cm31_neg(cm(i))
            assert (r.a, r.b) == (out["real"], out["imag"])
        elif op == "inverse":
            r = orc.CM31()
            assert L.orc_cm31_inverse(cm(i), r) == 0
            assert (r.a, r.b) == (out["real"], out["imag"])
        elif op == "into_slice":
            flat = [x for e in i["elements"] for x in (e["real"], e["imag"])]
            assert list(np.array(flat, dtype="<u4").tobytes()) == out
        n += 1
    assert n == 220


def test_qm31_vectors():
    n = 0
    for v in load_vectors("qm31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op in ("add", "sub", "mul"):
            assert list(getattr(L, f"orc_qm31_{op}")(orc.q(i["a"]), orc.q(i["b"])).tup()) == out
        elif op == "neg":
            assert list(L.orc_qm31_neg(orc.q(i["value"])).tup()) == out
        elif op == "inverse":
            r = orc.QM31()
            assert L.orc_qm31_inverse(orc.q(i["value"]), r) == 0
            assert list(r.tup()) == out
        elif op == "mul_cm31":
            assert list(L.orc_qm31_mul_cm31(orc.q(i["qm31"]), orc.CM31(*i["cm31"])).tup()) == out
        elif op == "into_slice":
            flat = [x for e in i["elements"] for x in e]
            assert list(np.array(flat, dtype="<u4").tobytes()) == out
        n += 1
    assert n == 135


def test_securecolumn_vectors():
    """SecureColumnByCoords = SoA of 4 M31 columns (fields/secure_columns.ts:124-217)."""
    for v in load_vectors("securecolumn"):
        if v["operation"] in ("to_vec", "from_iter"):
            vals = v["inputs"].get("column_values") or v["inputs"].get("input_values")
            cols = [np.array([q[k] for q in vals], dtype=np.uint32) for k in range(4)]
            back = [[int(cols[k][r]) for k in range(4)] for r in range(len(vals))]
            assert back == v["output"]


def test_rust_unit_constants(golden):
    k = golden["field_kat"]
    assert list(L.orc_qm31_mul(orc.q(k["qm31_mul"]["a"]), orc.q(k["qm31_mul"]["b"])).tup()) == k["qm31_mul"]["out"]
    r = L.orc_cm31_mul(orc.CM31(*k["cm31_mul"]["a"]), orc.CM31(*k["cm31_mul"]["b"]))
    assert [r.a, r.b] == k["cm31_mul"]["out"]
    # packages/core/test/fields/m31.test.ts:104-140
    assert L.orc_m31_from_i32(-1) == P - 1
    assert L.orc_m31_partial_reduce(2 * P - 19) == P - 19
    assert L.orc_m31_reduce(P * P - 19) == P - 19
    r = orc.C.c_uint32()
    assert L.orc_m31_inverse(0, r) == 1  # "0 has no inverse"
    assert L.orc_m31_from_u32(P) == 0    # App. B-6


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 8, 64, 1000, 4096])
def test_batch_inverse_equals_elementwise(n):
    """packages/core/test/fields/fields.test.ts:35-97,243-330."""
    rng = np.random.default_rng(n)
    col = rng.integers(1, P, size=n, dtype=np.uint32)
    out = orc.m31_batch_inverse(col)
    assert all(pow(int(a), P - 2, P) == int(b) for a, b in zip(col, out))
    cols4 = [rng.integers(1, P, size=n, dtype=np.uint32) for _ in range(4)]
    inv4 = orc.qm31_batch_inverse(cols4)
    prod = orc.qm31_col_mul(cols4, inv4)
    assert all((prod[0] == 1)) and all(not prod[k].any() for k in (1, 2, 3))


def test_batch_inverse_zero_throws():
    col = np.array([5, 0, 7, 9, 1, 2, 3, 4], dtype=np.uint32)
    with pytest.raises(orc.OracleError, match="0 has no inverse"):
        orc.m31_batch_inverse(col)
