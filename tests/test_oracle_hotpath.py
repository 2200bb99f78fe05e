"""Pins the CPU oracle's CFFT / FRI / Merkle / quotient restatement against (i) the independent big-int
fixtures of tests/golden/hotpath_golden.json, (ii) the absolute KATs held by the reference's tests and
(iii) ports of the reference's mathematical property tests."""
import hashlib

import numpy as np
import pytest

from conftest import P, column, golden_interp_values, rand_column
from oracle import oracle as orc

L = orc.lib()


def digest(cols):
    return hashlib.blake2s(b"".join(np.asarray(c, dtype="<u4").tobytes() for c in cols)).hexdigest()


def half_odds(k):
    return L.orc_half_odds_initial(k)


# ------------------------------------------------------------------ twiddles
def test_twiddle_trees(golden):
    for e in golden["twiddles"]:
        buf, ibuf = orc.precompute_twiddles(e["coset_initial"], e["log"])
        assert digest([buf]) == e["digest"] and digest([ibuf]) == e["idigest"]
        if "buf" in e:
            assert buf.tolist() == e["buf"] and ibuf.tolist() == e["ibuf"]
        assert buf[-1] == 1 and len(buf) == 1 << e["log"]


def test_slicing_kat(golden):
    """test/poly/domainLineTwiddles.test.ts:7-13: slices [[0..3],[4,5],[6]] of an 8-entry buffer.
    The oracle's CFFT consumes exactly these slices; check through a transform that only uses the rule:
    lineTw[j] = buf[L-2^(n-1-j) : L-2^(n-2-j)] for a line domain of log n-1 = 3."""
    k = golden["slicing_kat"]
    buf, n = k["buffer"], k["log_line_domain"] + 1
    L_ = len(buf)
    got = [buf[L_ - (1 << (n - 1 - j)): L_ - (1 << (n - 2 - j))] for j in range(n - 1)]
    assert got == k["expect"]


# ------------------------------------------------------------------ CFFT
def test_cfft_golden(golden):
    for e in golden["cfft"]:
        n = e["log"]
        coeffs = column(e["seed"], 1 << n)
        assert digest([coeffs]) == e["coeffs_digest"]
        tw_log = max(n - 1, 1)
        tw, itw = orc.precompute_twiddles(half_odds(tw_log), tw_log)
        ev = orc.cfft_evaluate(coeffs, n, e["half_initial"], tw, tw_log)
        assert digest([ev]) == e["eval_digest"], f"log {n}"
        if "eval" in e:
            assert ev.tolist() == e["eval"]
        back = orc.cfft_interpolate(ev, n, e["half_initial"], itw, tw_log)
        assert back.tolist() == coeffs.tolist()


def test_cfft_interpolate_golden(golden):
    """interpolate known answers log 1..10 (values of a seeded polynomial, stated by definition -> its coefficients)."""
    assert [e["log"] for e in golden["cfft_interpolate"]] == list(range(1, 11))
    assert [e["log"] for e in golden["cfft"]] == list(range(1, 11))
    for e in golden["cfft_interpolate"]:
        n = e["log"]
        vals = golden_interp_values(e)
        assert digest([vals]) == e["values_digest"]
        tw_log = max(n - 1, 1)
        _, itw = orc.precompute_twiddles(half_odds(tw_log), tw_log)
        co = orc.cfft_interpolate(vals, n, e["half_initial"], itw, tw_log)
        assert digest([co]) == e["coeffs_digest"], f"log {n}"
        assert co.tolist() == column(e["coeffs_seed"], 1 << n).tolist()
        if "coeffs" in e:
            assert co.tolist() == e["coeffs"]


@pytest.mark.parametrize("n", [3, 6, 10])
def test_cfft_larger_tree_serves_smaller_domain(n):
    """A twiddle tree of a bigger root coset serves smaller domains (slices are taken from the END,
    poly/utils.ts:89-98; is_doubling_of, circle.ts:264-266)."""
    coeffs = rand_column(n, 1 << n)
    tw_s, itw_s = orc.precompute_twiddles(half_odds(n - 1), n - 1)
    tw_b, itw_b = orc.precompute_twiddles(half_odds(n + 2), n + 2)
    a = orc.cfft_evaluate(coeffs, n, half_odds(n - 1), tw_s, n - 1)
    b = orc.cfft_evaluate(coeffs, n, half_odds(n - 1), tw_b, n + 2)
    assert (a == b).all()
    assert (orc.cfft_interpolate(a, n, half_odds(n - 1), itw_b, n + 2) == coeffs).all()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_evaluate_matches_eval_at_point(n):
    """test/backend/cpu/circle.test.ts:52-97: evaluate(poly) bit-reversed == eval_at_point at every domain point."""
    coeffs = rand_column(100 + n, 1 << n)
    tw_log = max(n - 1, 1)
    tw, _ = orc.precompute_twiddles(half_odds(tw_log), tw_log)
    ev = orc.cfft_evaluate(coeffs, n, half_odds(n - 1), tw, tw_log)
    for i in range(1 << n):
        p = L.orc_circle_domain_at(half_odds(n - 1), n - 1, i)
        v = orc.eval_at_point(coeffs, n, (p.x, 0, 0, 0), (p.y, 0, 0, 0))
        assert v == (int(ev[L.orc_bit_reverse_index(i, n)]), 0, 0, 0)


def test_log3_compat_swap():
    """App. B-1: the reference swaps outputs 5<->7 at log 3 (backend/cpu/circle.ts:123-131,145-151)."""
    coeffs = rand_column(3, 8)
    tw, itw = orc.precompute_twiddles(half_odds(2), 2)
    true = orc.cfft_evaluate(coeffs, 3, half_odds(2), tw, 2)
    compat = orc.cfft_evaluate(coeffs, 3, half_odds(2), tw, 2, compat=True)
    sw = true.copy(); sw[5], sw[7] = true[7], true[5]
    assert (compat == sw).all()
    assert (orc.cfft_interpolate(compat, 3, half_odds(2), itw, 2, compat=True) == coeffs).all()


def test_eval_at_point_golden(golden):
    for e in golden["eval_at_point"]:
        coeffs = column(e["seed"], 1 << e["log"])
        assert list(orc.eval_at_point(coeffs, e["log"], *e["point"])) == e["value"]


def test_not_enough_twiddles():
    tw, _ = orc.precompute_twiddles(half_odds(2), 2)
    with pytest.raises(orc.OracleError, match="Not enough twiddles"):
        orc.cfft_evaluate(np.zeros(64, dtype=np.uint32), 6, half_odds(5), tw, 2)


# ------------------------------------------------------------------ bit reverse
def test_bit_reverse():
    """test/backend/backend.test.ts:60-90."""
    assert orc.bit_reverse(np.arange(8)).tolist() == [0, 4, 2, 6, 1, 5, 3, 7]
    assert orc.bit_reverse(np.array([7])).tolist() == [7]
    for bad in (0, 3, 6):
        with pytest.raises(orc.OracleError, match="length is not power of two"):
            orc.bit_reverse(np.zeros(bad, dtype=np.uint32))


# ------------------------------------------------------------------ FRI
def test_fold_line_golden(golden):
    for e in golden["fold_line"]:
        cols = [column(s, 1 << e["log"]) for s in e["seeds"]]
        out = orc.fold_line(cols, e["log"], e["coset_initial"], e["alpha"])
        assert digest(out) == e["out_digest"], e["log"]
        if "out" in e:
            assert [c.tolist() for c in out] == e["out"]


def test_fold_circle_golden(golden):
    for e in golden["fold_circle"]:
        src = [column(s, 1 << e["log"]) for s in e["src_seeds"]]
        dst = [column(s, 1 << (e["log"] - 1)) for s in e["dst_seeds"]]
        out = orc.fold_circle_into_line(dst, src, e["log"], e["half_initial"], e["alpha"])
        assert digest(out) == e["out_digest"], e["log"]
        if "out" in e:
            assert [c.tolist() for c in out] == e["out"]


def test_fold_errors():
    z = [np.zeros(1, dtype=np.uint32)] * 4
    with pytest.raises(orc.OracleError, match="Evaluation too small"):
        orc.fold_line(z, 0, 0, (1, 0, 0, 0))
    with pytest.raises(orc.OracleError, match="Length mismatch"):
        orc.fold_circle_into_line([np.zeros(3, dtype=np.uint32)] * 4, [np.zeros(8, dtype=np.uint32)] * 4, 3, half_odds(2), (1, 0, 0, 0))


@pytest.mark.parametrize("n", [1, 2, 8, 64])
def test_decompose_reconstructs(n):
    """test/backend/cpu/fri.test.ts:44-72,124-184: g -/+ lambda halves reconstruct f; sums of g halves balance."""
    cols = [rand_column(40 + k, n) for k in range(4)]
    g, lam = orc.decompose(cols)
    half = n // 2
    for k in range(4):
        if n == 1:
            assert (int(g[k][0]) + lam[k]) % P == int(cols[k][0])
            continue
        assert ((g[k][:half].astype(np.uint64) + lam[k]) % P == cols[k][:half]).all()
        assert ((g[k][half:].astype(np.uint64) + P - lam[k]) % P == cols[k][half:]).all()
        assert int(g[k][:half].astype(np.uint64).sum() % P) == int(g[k][half:].astype(np.uint64).sum() % P)


# ------------------------------------------------------------------ Blake2s / Merkle
def test_blake2s_kats(golden):
    k = golden["blake2s_kat"]
    assert orc.blake2s(b"").hex() == k[""]
    assert orc.blake2s(b"a").hex() == k["a"]
    assert orc.blake2s(b"b").hex() == k["b"]
    assert orc.blake2s(orc.blake2s(b"a") + orc.blake2s(b"b")).hex() == k["H(a)||H(b)"]
    assert orc.blake2s_compress([0] * 8, [0] * 16, 0, 0, 0, 0).tolist() == golden["compress_zero_kat"]
    assert orc.blake2s(bytes(32) + (4).to_bytes(4, "little") + bytes(4)).hex() == golden["mix_u64_4_digest"]
    assert golden["mix_u64_4_digest"].startswith("af0e8a72") and golden["mix_u64_4_digest"].endswith("2ac5")
    for n in (1, 31, 32, 55, 63, 64, 65, 127, 128, 129, 1000):
        msg = bytes((i * 7 + n) & 0xFF for i in range(n))
        assert orc.blake2s(msg) == hashlib.blake2s(msg).digest()


def test_merkle_golden(golden):
    for e in golden["merkle"]:
        cols = [column(e["seed_base"] + i, 1 << lg) for i, lg in enumerate(e["log_sizes"])]
        layers, root = orc.merkle_commit(cols, e["log_sizes"])
        assert root.hex() == e["root"], e["name"]
        assert [len(l) for l in layers] == e["layer_sizes"]
        assert hashlib.blake2s(b"".join(l.tobytes() for l in layers)).hexdigest() == e["layers_digest"]


def test_merkle_lcg_case(golden):
    """prepareMerkle data of vcs/test_utils.ts:47-144 (LCG seed 0, 10 columns of log 3..4)."""
    e = golden["merkle_lcg"]
    layers, root = orc.merkle_commit([np.array(c, dtype=np.uint32) for c in e["cols"]], e["log_sizes"])
    assert root.hex() == e["root"]
    assert [[bytes(h).hex() for h in l] for l in layers] == e["layers"]


def test_commit_on_layer_matches_hash_node():
    """test/backend/cpu/blake2.test.ts:55-176 pattern: commitOnLayer == hashNode per node."""
    cols = [rand_column(7 + i, 8) for i in range(3)]
    l3 = orc.commit_on_layer(3, None, cols)
    for i in range(8):
        assert l3[i].tobytes() == orc.hash_node(None, [c[i] for c in cols])
    l2 = orc.commit_on_layer(2, l3, [c[:4] for c in cols[:2]])
    for i in range(4):
        assert l2[i].tobytes() == orc.hash_node((l3[2 * i].tobytes(), l3[2 * i + 1].tobytes()), [c[i] for c in cols[:2]])


# ------------------------------------------------------------------ quotients
def test_quotients_golden_and_low_degree(golden):
    e = golden["quotients"][0]
    n, npoly = e["log"], e["poly_log"]
    coeffs = column(e["coeffs_seed"], 1 << npoly)
    ext = np.concatenate([coeffs, np.zeros((1 << n) - (1 << npoly), dtype=np.uint32)])
    tw, itw = orc.precompute_twiddles(half_odds(n - 1), n - 1)
    ev = orc.cfft_evaluate(ext, n, e["half_initial"], tw, n - 1)
    assert digest([ev]) == e["col_digest"]
    assert list(orc.eval_at_point(coeffs, npoly, *e["point"])) == e["value"]
    out = orc.accumulate_quotients(e["half_initial"], n, [ev], e["random_coeff"], [(e["point"][0], e["point"][1], [(0, e["value"])])])
    assert [c.tolist() for c in out] == e["out"]
    # test_quotients_are_low_degree (pcs/quotients.ts:179-201 comment): each coordinate polynomial of the
    # quotient has degree < 2^poly_log, i.e. the upper coefficients of its interpolation vanish.
    for c in out:
        co = orc.cfft_interpolate(c, n, e["half_initial"], itw, n - 1)
        assert not co[1 << npoly:].any() and co[: 1 << npoly].any()


def test_quotients_two_batches(golden):
    e = golden["quotients"][1]
    n = e["log"]
    cols = [column(s, 1 << n) for s in e["col_seeds"]]
    batches = [(b["point"][0], b["point"][1], [(ci, v) for ci, v in b["cols"]]) for b in e["batches"]]
    out = orc.accumulate_quotients(e["half_initial"], n, cols, e["random_coeff"], batches)
    assert digest(out) == e["out_digest"]
    assert [[int(out[k][r]) for k in range(4)] for r in range(4)] == e["out_head"]


def test_accumulate_and_powers():
    a = [rand_column(60 + k, 16) for k in range(4)]
    b = [rand_column(70 + k, 16) for k in range(4)]
    s = orc.accumulate(a, b)
    for k in range(4):
        assert ((a[k].astype(np.uint64) + b[k]) % P == s[k]).all()
    pw = orc.generate_secure_powers((1, 2, 3, 4), 4)
    assert pw[0] == (1, 0, 0, 0) and pw[1] == (1, 2, 3, 4)
    assert pw[2] == L.orc_qm31_mul(orc.q(pw[1]), orc.q(pw[1])).tup()
    assert orc.generate_secure_powers((1, 2, 3, 4), 0) == []
