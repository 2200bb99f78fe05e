#!/usr/bin/env python3
"""Independent big-int model of the hot path -> tests/golden/hotpath_golden.json.

The reference (TypeScript on Bun) cannot run in the build image and holds no golden vectors for
CFFT / FRI folds / quotients / Merkle layers.  This script states those
results from their *mathematical definitions* with plain Python integers and the stdlib
hashlib.blake2s — deliberately NOT the layer-loop / twiddle-tree formulation used by oracle/ and by
the HIP kernels — so that agreement between the three is meaningful:

  evaluate(coeffs)[bitrev(i)]  := poly(domain.at(i))      (definition used by the reference's own
                                                           property test, test/backend/cpu/circle.test.ts:52-97)
  fold_line / fold_circle_into_line := formulas of fri.ts:120-192 with explicit domain points
  merkle node := BLAKE2s-256(left || right || LE32(values...))   (vcs/blake2_merkle.ts:9-24)
  quotient row := formula of backend/cpu/quotients.ts:80-116 with Rust conjugation semantics

Run:  python3 tests/golden/gen_golden.py   (deterministic; rewrites the JSON next to it)
It imports nothing from oracle/, tstwo_amd/ or /root/reference.
"""
import hashlib
import json
import os

P = 2**31 - 1
GEN = (2, 1268011823)
HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- deterministic inputs
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    def m31(self, nonzero=False):
        while True:
            v = self.next() >> 33
            if v < P and not (nonzero and v == 0):
                return v


def column(seed, n, nonzero=False):
    r = SplitMix64(seed)
    return [r.m31(nonzero) for _ in range(n)]


# ---------------------------------------------------------------- fields
def inv(a):
    assert a % P != 0
    return pow(a, P - 2, P)


def cadd(x, y): return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
def csub(x, y): return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
def cmul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
def cinv(x):
    n = inv((x[0] * x[0] + x[1] * x[1]) % P)
    return (x[0] * n % P, (-x[1]) * n % P)


R = (2, 1)
def qadd(x, y): return tuple((a + b) % P for a, b in zip(x, y))
def qsub(x, y): return tuple((a - b) % P for a, b in zip(x, y))
def qmul(x, y):
    a0, a1, b0, b1 = x[:2], x[2:], y[:2], y[2:]
    c0 = cadd(cmul(a0, b0), cmul(R, cmul(a1, b1)))
    c1 = cadd(cmul(a0, b1), cmul(a1, b0))
    return c0 + c1
def qmulm(x, m): return tuple(a * m % P for a in x)
def qmulc(x, c): return cmul(x[:2], c) + cmul(x[2:], c)
def qconj(x): return (x[0], x[1], (-x[2]) % P, (-x[3]) % P)     # Rust ComplexConjugate for QM31
def qfrom(m): return (m % P, 0, 0, 0)
def qinv(x):
    b2 = cmul(x[2:], x[2:])
    ib2 = ((-b2[1]) % P, b2[0])
    denom = csub(cmul(x[:2], x[:2]), cadd(cadd(b2, b2), ib2))
    di = cinv(denom)
    c1 = cmul(x[2:], di)
    return cmul(x[:2], di) + ((-c1[0]) % P, (-c1[1]) % P)


# ---------------------------------------------------------------- circle group
def padd(p, q): return ((p[0] * q[0] - p[1] * q[1]) % P, (p[0] * q[1] + p[1] * q[0]) % P)
def point(idx):
    idx %= 2**31
    res, cur = (1, 0), GEN
    while idx:
        if idx & 1:
            res = padd(res, cur)
        cur = padd(cur, cur)
        idx >>= 1
    return res


def subgroup_gen(k): return (1 << (31 - k)) % 2**31
def half_odds(k): return (subgroup_gen(k + 2), k)          # (initial index, log size)
def coset_at(c, i): return point(c[0] + i * subgroup_gen(c[1]))
def coset_double(c): return ((2 * c[0]) % 2**31, c[1] - 1)
def domain_at(half, i):
    h = 1 << half[1]
    if i < h:
        return point(half[0] + i * subgroup_gen(half[1]))
    return point(-(half[0] + (i - h) * subgroup_gen(half[1])))


def bitrev(i, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def bitrev_list(v):
    n = len(v).bit_length() - 1
    return [v[bitrev(i, n)] for i in range(len(v))]


# ---------------------------------------------------------------- definitions
def twiddle_tree(c):
    buf = []
    while c[1] > 0:
        xs = [coset_at(c, i)[0] for i in range((1 << c[1]) // 2)]
        buf += bitrev_list(xs)
        c = coset_double(c)
    return buf + [1]


def poly_eval_m31(coeffs, p):
    """circle-poly basis: coefficient k multiplies y^{k0} x^{k1} pi(x)^{k2} ... (backend/cpu/circle.ts:52-69)."""
    n = len(coeffs).bit_length() - 1
    if n == 0:
        return coeffs[0]
    factors = [p[1]]
    x = p[0]
    for _ in range(1, n):
        factors.append(x)
        x = (2 * x * x - 1) % P
    total = 0
    for k, c in enumerate(coeffs):
        if c == 0:
            continue
        t = c
        for b in range(n):
            if (k >> b) & 1:
                t = t * factors[b] % P
        total += t
    return total % P


def poly_eval_qm31(coeffs, px, py):
    n = len(coeffs).bit_length() - 1
    if n == 0:
        return qfrom(coeffs[0])
    factors = [py]
    x = px
    for _ in range(1, n):
        factors.append(x)
        x = qsub(qmulm(qmul(x, x), 2), qfrom(1))
    total = (0, 0, 0, 0)
    for k, c in enumerate(coeffs):
        t = qfrom(c)
        for b in range(n):
            if (k >> b) & 1:
                t = qmul(t, factors[b])
        total = qadd(total, t)
    return total


def basis_at(n, p):
    """[b_k(p) for k < 2^n], b_k = y^{k0} x^{k1} pi(x)^{k2} ...: the same monomials as poly_eval_m31, expanded once per point."""
    factors = [p[1]]
    x = p[0]
    for _ in range(1, n):
        factors.append(x)
        x = (2 * x * x - 1) % P
    b = [1]
    for f in factors:
        b = b + [v * f % P for v in b]
    return b


def evaluate(coeffs, n):
    half = half_odds(n - 1)
    if n <= 8:
        return [poly_eval_m31(coeffs, domain_at(half, bitrev(i, n))) for i in range(1 << n)]
    out = []
    for i in range(1 << n):          # log 9, 10: sum_k c_k b_k(p) with the basis expanded once per point
        b = basis_at(n, domain_at(half, bitrev(i, n)))
        out.append(sum(c * v for c, v in zip(coeffs, b)) % P)
    return out


def fold_line(vals, k, coset, alpha):
    out = []
    for i in range(len(vals) // 2):
        x = coset_at(coset, bitrev(2 * i, k))[0]
        a, b = vals[2 * i], vals[2 * i + 1]
        f0, f1 = qadd(a, b), qmulm(qsub(a, b), inv(x))
        out.append(qadd(f0, qmul(alpha, f1)))
    return out


def fold_circle_into_line(dst, src, n, half, alpha):
    a2 = qmul(alpha, alpha)
    out = []
    for i in range(len(dst)):
        p = domain_at(half, bitrev(2 * i, n))
        a, b = src[2 * i], src[2 * i + 1]
        f0, f1 = qadd(a, b), qmulm(qsub(a, b), inv(p[1]))
        out.append(qadd(qmul(dst[i], a2), qadd(qmul(alpha, f1), f0)))
    return out


def b2(msg): return hashlib.blake2s(msg).digest()
def le32(vals): return b"".join(int(v).to_bytes(4, "little") for v in vals)


def merkle_layers(cols):
    """MerkleProver.commit (vcs/prover.ts:13-30): layers root-first."""
    if not cols:
        return [[b2(b"")]]
    logs = [len(c).bit_length() - 1 for c in cols]
    prev, layers = None, []
    for lg in range(max(logs), -1, -1):
        lc = [c for c, l in zip(cols, logs) if l == lg]
        layer = []
        for i in range(1 << lg):
            msg = (prev[2 * i] + prev[2 * i + 1]) if prev is not None else b""
            layer.append(b2(msg + le32([c[i] for c in lc])))
        layers.append(layer)
        prev = layer
    return layers[::-1]


def quotients(cols, n, half, random_coeff, batches):
    """backend/cpu/quotients.ts:52-191, Rust semantics for conj and Pr/Pi."""
    consts = []
    for (px, py, cv) in batches:
        alpha, lc = qfrom(1), []
        for (_, v) in cv:
            alpha = qmul(alpha, random_coeff)
            a = qsub(qconj(v), v)
            c = qsub(qconj(py), py)
            b = qsub(qmul(v, c), qmul(a, py))
            lc.append((qmul(alpha, a), qmul(alpha, b), qmul(alpha, c)))
        consts.append((lc, alpha))          # alpha == random_coeff ** len(cv)
    out = []
    for row in range(1 << n):
        p = domain_at(half, bitrev(row, n))
        acc = (0, 0, 0, 0)
        for (px, py, cv), (lc, bc) in zip(batches, consts):
            prx, pry, pix, piy = px[:2], py[:2], px[2:], py[2:]
            den = csub(cmul(csub(prx, (p[0], 0)), piy), cmul(csub(pry, (p[1], 0)), pix))
            num = (0, 0, 0, 0)
            for (ci, _), (a, b, c) in zip(cv, lc):
                num = qadd(num, qsub(qmul(qfrom(cols[ci][row]), c), qadd(qmul(a, qfrom(p[1])), b)))
            acc = qadd(qmul(acc, bc), qmulc(num, cinv(den)))
        out.append(acc)
    return out


def soa(qs): return [[v[k] for v in qs] for k in range(4)]
def digest_u32(cols): return hashlib.blake2s(b"".join(le32(c) for c in cols)).hexdigest()

SECURE_GEN = ((1, 0, 478637715, 513582971), (992285211, 649143431, 740191619, 1186584352))  # circle.ts:143-146


def main():
    g = {"_comment": "generated by tests/golden/gen_golden.py (independent big-int model); do not edit"}

    # twiddle trees of half_odds(m), m = 1..8  (full arrays to log 5, digests beyond)
    g["twiddles"] = []
    for m in range(1, 9):
        c = half_odds(m)
        buf = twiddle_tree(c)
        ibuf = [inv(x) for x in buf]
        e = {"coset_initial": c[0], "log": m, "digest": digest_u32([buf]), "idigest": digest_u32([ibuf])}
        if m <= 5:
            e["buf"], e["ibuf"] = buf, ibuf
        g["twiddles"].append(e)
    g["slicing_kat"] = {"buffer": list(range(8)), "log_line_domain": 3, "expect": [[0, 1, 2, 3], [4, 5], [6]]}

    # CFFT by definition, log 1..10: evaluate(coeffs) and, as its own known-answer pair with other seeds,
    # interpolate: the values of a seeded polynomial on the domain (by definition) are the input, its coefficients the
    # expected output (interpolate is the inverse map, backend/cpu/circle.ts:136-207).
    g["cfft"] = []
    for n in range(1, 11):
        coeffs = column(300 + n, 1 << n)
        ev = evaluate(coeffs, n)
        e = {"log": n, "half_initial": half_odds(n - 1)[0], "seed": 300 + n,
             "coeffs_digest": digest_u32([coeffs]), "eval_digest": digest_u32([ev])}
        if n <= 5:
            e["coeffs"], e["eval"] = coeffs, ev
        g["cfft"].append(e)
    g["cfft_interpolate"] = []
    for n in range(1, 11):
        coeffs = column(350 + n, 1 << n)
        vals = evaluate(coeffs, n)
        e = {"log": n, "half_initial": half_odds(n - 1)[0], "coeffs_seed": 350 + n,
             "values_digest": digest_u32([vals]), "coeffs_digest": digest_u32([coeffs])}
        if n <= 5:
            e["values"], e["coeffs"] = vals, coeffs
        elif n <= 10:
            e["values_b64"] = __import__("base64").b64encode(le32(vals)).decode()   # the input must travel: it is not seed-derived
        g["cfft_interpolate"].append(e)
    # eval_at_point at the secure-field generator
    g["eval_at_point"] = []
    for n in range(0, 7):
        coeffs = column(400 + n, 1 << n)
        g["eval_at_point"].append({"log": n, "seed": 400 + n, "point": [list(SECURE_GEN[0]), list(SECURE_GEN[1])],
                                   "value": list(poly_eval_qm31(coeffs, *SECURE_GEN))})

    # FRI folds, log 1..8 / 2..8
    alpha = (19283, 1, 2, 3)
    g["fold_line"], g["fold_circle"] = [], []
    for k in range(1, 9):
        cols = [column(500 + 4 * k + j, 1 << k) for j in range(4)]
        vals = list(zip(*cols))
        c = half_odds(k)
        out = fold_line(vals, k, c, alpha)
        e = {"log": k, "coset_initial": c[0], "alpha": list(alpha), "seeds": [500 + 4 * k + j for j in range(4)],
             "out_digest": digest_u32(soa(out))}
        if k <= 4:
            e["in"], e["out"] = [list(c_) for c_ in cols], soa(out)
        g["fold_line"].append(e)
    for n in range(2, 9):
        cols = [column(600 + 4 * n + j, 1 << n) for j in range(4)]
        dcols = [column(700 + 4 * n + j, 1 << (n - 1)) for j in range(4)]
        half = half_odds(n - 1)
        out = fold_circle_into_line(list(zip(*dcols)), list(zip(*cols)), n, half, alpha)
        e = {"log": n, "half_initial": half[0], "alpha": list(alpha),
             "src_seeds": [600 + 4 * n + j for j in range(4)], "dst_seeds": [700 + 4 * n + j for j in range(4)],
             "out_digest": digest_u32(soa(out))}
        if n <= 4:
            e["src"], e["dst"], e["out"] = [list(c_) for c_ in cols], [list(c_) for c_ in dcols], soa(out)
        g["fold_circle"].append(e)

    # Merkle: (C, log) cases of SURVEY §8c + the mixed-size LCG case of vcs/test_utils.ts:47-144
    g["merkle"] = []
    for name, shape in [("c1_log0", [0]), ("c3_log1", [1, 1, 1]), ("c20_log0", [0] * 20), ("empty", []),
                        ("c4_log5", [5] * 4), ("c16_log4", [4] * 16), ("c17_log3", [3] * 17),
                        ("mixed", [3, 5, 5, 2, 0, 3, 4, 5, 1, 4])]:
        cols = [column(800 + i, 1 << lg) for i, lg in enumerate(shape)]
        layers = merkle_layers(cols)
        g["merkle"].append({"name": name, "log_sizes": shape, "seed_base": 800, "root": layers[0][0].hex(),
                            "layers_digest": hashlib.blake2s(b"".join(b"".join(l) for l in layers)).hexdigest(),
                            "layer_sizes": [len(l) for l in layers]})
    # LCG case: a=1664525 c=1013904223 m=2^32 seed 0; 10 columns, log sizes from genRange(3,5), values genRange(0,2^30)
    s = [0]
    def nxt():
        s[0] = (1664525 * s[0] + 1013904223) % 2**32
        return s[0]
    logs = [3 + nxt() % 2 for _ in range(10)]
    cols = [[nxt() % (1 << 30) for _ in range(1 << lg)] for lg in logs]
    layers = merkle_layers(cols)
    g["merkle_lcg"] = {"log_sizes": logs, "cols": cols, "root": layers[0][0].hex(),
                       "layers": [[h.hex() for h in l] for l in layers]}

    # Blake2s absolute KATs held by the reference's tests (test/vcs/blake2_hash.test.ts:6-8,127,166)
    g["blake2s_kat"] = {"": "69217a3079908094e11121d042354a7c1f55b6482ca1a51e1b250dfd1ed0eef9",
                        "a": "4a0d129873403037c2cd9b9048203687f6233fb6738956e0349bd4320fec3e90",
                        "b": "04449e92c9a7657ef2d677b8ef9da46c088f13575ea887e4818fc455a2bca500",
                        "H(a)||H(b)": "2d12d4f7a2c2c9e02fc6300b0d23c772457aa5e30d1d69e7b589b8a48afe5425"}
    for k, v in [("", b""), ("a", b"a"), ("b", b"b")]:
        assert b2(v).hex() == g["blake2s_kat"][k]
    assert b2(b2(b"a") + b2(b"b")).hex() == g["blake2s_kat"]["H(a)||H(b)"]
    # compression KAT test/vcs/blake2s_ref.test.ts:81-92 and the Rust-side channel digest
    g["compress_zero_kat"] = [1848029226, 2795995149, 1371241353, 520215377, 125539373, 602280490, 2742896865, 1845544798]
    g["mix_u64_4_digest"] = b2(bytes(32) + (4).to_bytes(4, "little") + bytes(4)).hex()

    # Quotients: test_quotients_are_low_degree setup (pcs/quotients.ts:179-201 comment): log 7 poly, blowup 1
    n_poly, blow = 5, 1
    coeffs = column(900, 1 << n_poly)
    nq = n_poly + blow
    half = half_odds(nq - 1)
    ev = evaluate(coeffs + [0] * ((1 << nq) - (1 << n_poly)), nq)
    value = poly_eval_qm31(coeffs, *SECURE_GEN)
    coeff = (1, 2, 3, 4)
    qs = quotients([ev], nq, half, coeff, [(SECURE_GEN[0], SECURE_GEN[1], [(0, value)])])
    # second case: 3 columns, 2 batches (second batch at 2*SECURE_GEN), exercises batch_random_coeffs
    cols3 = [column(910 + j, 1 << nq) for j in range(3)]
    def spadd(p, q_):
        return (qsub(qmul(p[0], q_[0]), qmul(p[1], q_[1])), qadd(qmul(p[0], q_[1]), qmul(p[1], q_[0])))
    pt2 = spadd(SECURE_GEN, SECURE_GEN)
    vals = [(7, 8, 9, 10), (11, 12, 13, 14), (15, 16, 17, 18), (19, 20, 21, 22)]
    batches = [(SECURE_GEN[0], SECURE_GEN[1], [(0, vals[0]), (2, vals[1])]), (pt2[0], pt2[1], [(1, vals[2]), (0, vals[3])])]
    qs2 = quotients(cols3, nq, half, coeff, batches)
    g["quotients"] = [
        {"name": "low_degree", "log": nq, "half_initial": half[0], "coeffs_seed": 900, "poly_log": n_poly,
         "random_coeff": list(coeff), "point": [list(SECURE_GEN[0]), list(SECURE_GEN[1])], "value": list(value),
         "col_digest": digest_u32([ev]), "out": soa(qs)},
        {"name": "two_batches", "log": nq, "half_initial": half[0], "col_seeds": [910, 911, 912],
         "random_coeff": list(coeff),
         "batches": [{"point": [list(b[0]), list(b[1])], "cols": [[ci, list(v)] for ci, v in b[2]]} for b in batches],
         "out_digest": digest_u32(soa(qs2)), "out_head": [list(v) for v in qs2[:4]]},
    ]

    # field constants ported from the Rust unit tests (test/fields/qm31.test.ts:46-62, cm31.test.ts:100-125)
    g["field_kat"] = {"qm31_mul": {"a": [1, 2, 3, 4], "b": [4, 5, 6, 7], "out": [P - 71, 93, P - 16, 50]},
                      "cm31_mul": {"a": [1, 2], "b": [4, 5], "out": [P - 6, 13]}}
    assert list(qmul((1, 2, 3, 4), (4, 5, 6, 7))) == g["field_kat"]["qm31_mul"]["out"]
    assert list(cmul((1, 2), (4, 5))) == g["field_kat"]["cm31_mul"]["out"]

    with open(os.path.join(HERE, "hotpath_golden.json"), "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "hotpath_golden.json"), os.path.getsize(os.path.join(HERE, "hotpath_golden.json")), "bytes")


if __name__ == "__main__":
    main()
