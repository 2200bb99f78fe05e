/*
 * tstwo_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see tstwo_oracle.h).
 * Scalar, dependency-free restatement of the reference algorithms; every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference/packages/core/src).  Deliberately follows the reference's
 * *algorithm shape* (per-element domain.at()+inverse() in the folds, layer
 * loops in the CFFT) so that it is independent of the GPU kernels' twiddle-tree
 * formulation.
 */
#include "tstwo_oracle.h"
#include <stdlib.h>
#include <string.h>

#define P ORC_P
typedef uint32_t u32;
typedef uint64_t u64;

/* ------------------------------------------------------------------ M31 */
/* fields/m31.ts:89-101 */
u32 orc_m31_reduce(u64 x) { return (u32)((((((x >> 31) + x + 1) >> 31) + x)) & P); }
/* fields/m31.ts:60-64 */
u32 orc_m31_partial_reduce(u32 x) { return x >= P ? x - P : x; }
/* fields/m31.ts:252-267: negative v -> 2P - |v| then reduce */
u32 orc_m31_from_i32(int32_t v) {
    if (v < 0) {
        u64 a = (u64)(-(int64_t)v);
        return orc_m31_reduce(2ull * P - a);
    }
    return orc_m31_reduce((u64)v);
}
u32 orc_m31_from_u32(u32 v) { return orc_m31_reduce((u64)v); }
/* fields/m31.ts:147-149,161-163,126-131,168-173 */
u32 orc_m31_add(u32 a, u32 b) { return orc_m31_partial_reduce(a + b); }
u32 orc_m31_sub(u32 a, u32 b) { return orc_m31_partial_reduce(a + P - b); }
u32 orc_m31_neg(u32 a) { return orc_m31_partial_reduce(P - a); }
u32 orc_m31_mul(u32 a, u32 b) { return orc_m31_reduce((u64)a * (u64)b); }

static u32 sqn(u32 v, int n) {
    for (int i = 0; i < n; i++) v = orc_m31_mul(v, v);
    return v;
}
/* fields/m31.ts:305-326 — v^(2^31-3), 30 squarings + 7 multiplications */
u32 orc_m31_pow2147483645(u32 v) {
    u32 t0 = orc_m31_mul(sqn(v, 2), v);
    u32 t1 = orc_m31_mul(sqn(t0, 1), t0);
    u32 t2 = orc_m31_mul(sqn(t1, 3), t0);
    u32 t3 = orc_m31_mul(sqn(t2, 1), t0);
    u32 t4 = orc_m31_mul(sqn(t3, 8), t3);
    u32 t5 = orc_m31_mul(sqn(t4, 8), t3);
    return orc_m31_mul(sqn(t5, 7), t2);
}
/* fields/m31.ts:137-142 */
int orc_m31_inverse(u32 v, u32 *out) {
    if (v == 0) return ORC_ERR_ZERO_INVERSE;
    *out = orc_m31_pow2147483645(v);
    return ORC_OK;
}
static u32 minv(u32 v) { return orc_m31_pow2147483645(v); } /* caller guarantees v != 0 */

/* ------------------------------------------------------------------ CM31 (fields/cm31.ts) */
orc_cm31 orc_cm31_add(orc_cm31 x, orc_cm31 y) { return (orc_cm31){orc_m31_add(x.a, y.a), orc_m31_add(x.b, y.b)}; }
orc_cm31 orc_cm31_sub(orc_cm31 x, orc_cm31 y) { return (orc_cm31){orc_m31_sub(x.a, y.a), orc_m31_sub(x.b, y.b)}; }
orc_cm31 orc_cm31_neg(orc_cm31 x) { return (orc_cm31){orc_m31_neg(x.a), orc_m31_neg(x.b)}; }
/* cm31.ts:139-149: (ac - bd, ad + bc) */
orc_cm31 orc_cm31_mul(orc_cm31 x, orc_cm31 y) {
    return (orc_cm31){orc_m31_sub(orc_m31_mul(x.a, y.a), orc_m31_mul(x.b, y.b)),
                      orc_m31_add(orc_m31_mul(x.a, y.b), orc_m31_mul(x.b, y.a))};
}
static orc_cm31 cm31_mul_m31(orc_cm31 x, u32 m) { return (orc_cm31){orc_m31_mul(x.a, m), orc_m31_mul(x.b, m)}; }
/* cm31.ts:237-251: conj / (a^2 + b^2) */
int orc_cm31_inverse(orc_cm31 x, orc_cm31 *out) {
    u32 norm = orc_m31_add(orc_m31_mul(x.a, x.a), orc_m31_mul(x.b, x.b));
    if (norm == 0) return ORC_ERR_ZERO_INVERSE;
    u32 ni = minv(norm);
    *out = (orc_cm31){orc_m31_mul(x.a, ni), orc_m31_mul(orc_m31_neg(x.b), ni)};
    return ORC_OK;
}

/* ------------------------------------------------------------------ QM31 (fields/qm31.ts) */
static orc_cm31 q0(orc_qm31 x) { return (orc_cm31){x.a, x.b}; }
static orc_cm31 q1(orc_qm31 x) { return (orc_cm31){x.c, x.d}; }
static orc_qm31 qmake(orc_cm31 c0, orc_cm31 c1) { return (orc_qm31){c0.a, c0.b, c1.a, c1.b}; }
static const orc_cm31 QR = {2, 1}; /* qm31.ts:9  R = 2 + i */
orc_qm31 orc_qm31_add(orc_qm31 x, orc_qm31 y) { return qmake(orc_cm31_add(q0(x), q0(y)), orc_cm31_add(q1(x), q1(y))); }
orc_qm31 orc_qm31_sub(orc_qm31 x, orc_qm31 y) { return qmake(orc_cm31_sub(q0(x), q0(y)), orc_cm31_sub(q1(x), q1(y))); }
orc_qm31 orc_qm31_neg(orc_qm31 x) { return qmake(orc_cm31_neg(q0(x)), orc_cm31_neg(q1(x))); }
/* qm31.ts:223-233: (a0b0 + R a1b1, a0b1 + a1b0) */
orc_qm31 orc_qm31_mul(orc_qm31 x, orc_qm31 y) {
    orc_cm31 a0b0 = orc_cm31_mul(q0(x), q0(y));
    orc_cm31 a1b1 = orc_cm31_mul(q1(x), q1(y));
    orc_cm31 c0 = orc_cm31_add(a0b0, orc_cm31_mul(QR, a1b1));
    orc_cm31 c1 = orc_cm31_add(orc_cm31_mul(q0(x), q1(y)), orc_cm31_mul(q1(x), q0(y)));
    return qmake(c0, c1);
}
orc_qm31 orc_qm31_mul_m31(orc_qm31 x, u32 m) { return qmake(cm31_mul_m31(q0(x), m), cm31_mul_m31(q1(x), m)); }
/* qm31.ts:333-335 */
orc_qm31 orc_qm31_mul_cm31(orc_qm31 x, orc_cm31 m) { return qmake(orc_cm31_mul(q0(x), m), orc_cm31_mul(q1(x), m)); }
/* qm31.ts:282-305: b2=c1^2; ib2=(-b2.im,b2.re); denom=c0^2-(b2+b2+ib2); (c0*d^-1, -c1*d^-1) */
int orc_qm31_inverse(orc_qm31 x, orc_qm31 *out) {
    if ((x.a | x.b | x.c | x.d) == 0) return ORC_ERR_ZERO_INVERSE;
    orc_cm31 b2 = orc_cm31_mul(q1(x), q1(x));
    orc_cm31 ib2 = {orc_m31_neg(b2.b), b2.a};
    orc_cm31 denom = orc_cm31_sub(orc_cm31_mul(q0(x), q0(x)), orc_cm31_add(orc_cm31_add(b2, b2), ib2));
    orc_cm31 di;
    int rc = orc_cm31_inverse(denom, &di);
    if (rc) return rc;
    *out = qmake(orc_cm31_mul(q0(x), di), orc_cm31_neg(orc_cm31_mul(q1(x), di)));
    return ORC_OK;
}
static orc_qm31 qm31_from_m31(u32 v) { return (orc_qm31){v, 0, 0, 0}; }
static const orc_qm31 QZERO = {0, 0, 0, 0};
static const orc_qm31 QONE = {1, 0, 0, 0};

/* ------------------------------------------------------------------ batch inverse
 * fields/fields.ts:66-91 (classic) and :96-160 (WIDTH=4 interleave), written once per type with a
 * macro.  An input zero makes the single final inverse() fail -> ORC_ERR_ZERO_INVERSE. */
#define DEF_BATCH_INVERSE(NAME, T, MUL, INV, ONE)                                        \
    static int NAME##_classic(const T *col, T *dst, size_t n) {                          \
        if (n == 0) return ORC_OK;                                                        \
        dst[0] = col[0];                                                                  \
        for (size_t i = 1; i < n; i++) dst[i] = MUL(dst[i - 1], col[i]);                  \
        T cur;                                                                            \
        int rc = INV(dst[n - 1], &cur);                                                   \
        if (rc) return rc;                                                                \
        for (size_t i = n - 1; i > 0; i--) {                                              \
            dst[i] = MUL(dst[i - 1], cur);                                                \
            cur = MUL(cur, col[i]);                                                       \
        }                                                                                 \
        dst[0] = cur;                                                                     \
        return ORC_OK;                                                                    \
    }                                                                                     \
    static int NAME##_impl(const T *col, T *dst, size_t n) {                              \
        enum { W = 4 };                                                                   \
        if (n <= W || n % W != 0) return NAME##_classic(col, dst, n);                     \
        T cum[W], tail[W], tmp[W];                                                        \
        for (int i = 0; i < W; i++) cum[i] = ONE;                                         \
        for (size_t i = 0; i < n; i++) {                                                  \
            cum[i % W] = MUL(cum[i % W], col[i]);                                         \
            dst[i] = cum[i % W];                                                          \
        }                                                                                 \
        for (int i = 0; i < W; i++) tmp[i] = dst[n - W + i];                              \
        int rc = NAME##_classic(tmp, tail, W);                                            \
        if (rc) return rc;                                                                \
        for (size_t i = n - 1; i >= W; i--) {                                             \
            dst[i] = MUL(dst[i - W], tail[i % W]);                                        \
            tail[i % W] = MUL(tail[i % W], col[i]);                                       \
        }                                                                                 \
        for (int i = 0; i < W; i++) dst[i] = tail[i];                                     \
        return ORC_OK;                                                                    \
    }
static const u32 M_ONE = 1;
static const orc_cm31 C_ONE = {1, 0};
DEF_BATCH_INVERSE(bi_m31, u32, orc_m31_mul, orc_m31_inverse, M_ONE)
DEF_BATCH_INVERSE(bi_cm31, orc_cm31, orc_cm31_mul, orc_cm31_inverse, C_ONE)
DEF_BATCH_INVERSE(bi_qm31, orc_qm31, orc_qm31_mul, orc_qm31_inverse, QONE)

int orc_m31_batch_inverse(const u32 *col, u32 *dst, size_t n) {
    if (col == dst) { /* reference allocates a fresh dst; keep that aliasing-free contract */
        u32 *tmp = (u32 *)malloc((n ? n : 1) * sizeof(u32));
        int rc = bi_m31_impl(col, tmp, n);
        if (!rc) memcpy(dst, tmp, n * sizeof(u32));
        free(tmp);
        return rc;
    }
    return bi_m31_impl(col, dst, n);
}
int orc_cm31_batch_inverse(const orc_cm31 *col, orc_cm31 *dst, size_t n) { return bi_cm31_impl(col, dst, n); }
int orc_qm31_batch_inverse_soa(const u32 *const in[4], u32 *const out[4], size_t n) {
    orc_qm31 *a = (orc_qm31 *)malloc((n ? n : 1) * sizeof(orc_qm31));
    orc_qm31 *d = (orc_qm31 *)malloc((n ? n : 1) * sizeof(orc_qm31));
    for (size_t i = 0; i < n; i++) a[i] = (orc_qm31){in[0][i], in[1][i], in[2][i], in[3][i]};
    int rc = bi_qm31_impl(a, d, n);
    if (!rc)
        for (size_t i = 0; i < n; i++) {
            out[0][i] = d[i].a; out[1][i] = d[i].b; out[2][i] = d[i].c; out[3][i] = d[i].d;
        }
    free(a);
    free(d);
    return rc;
}

void orc_m31_col_add(const u32 *a, const u32 *b, u32 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = orc_m31_add(a[i], b[i]); }
void orc_m31_col_sub(const u32 *a, const u32 *b, u32 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = orc_m31_sub(a[i], b[i]); }
void orc_m31_col_mul(const u32 *a, const u32 *b, u32 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = orc_m31_mul(a[i], b[i]); }
void orc_m31_col_neg(const u32 *a, u32 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = orc_m31_neg(a[i]); }
void orc_qm31_col_mul_soa(const u32 *const a[4], const u32 *const b[4], u32 *const o[4], size_t n) {
    for (size_t i = 0; i < n; i++) {
        orc_qm31 r = orc_qm31_mul((orc_qm31){a[0][i], a[1][i], a[2][i], a[3][i]},
                                  (orc_qm31){b[0][i], b[1][i], b[2][i], b[3][i]});
        o[0][i] = r.a; o[1][i] = r.b; o[2][i] = r.c; o[3][i] = r.d;
    }
}

/* ------------------------------------------------------------------ bit reverse */
/* utils.ts:15-22 */
u32 orc_bit_reverse_index(u32 idx, u32 log_size) {
    u32 rev = 0;
    for (u32 i = 0; i < log_size; i++) {
        rev = (rev << 1) | (idx & 1);
        idx >>= 1;
    }
    return rev;
}
static u32 ilog2(size_t n) {
    u32 l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}
/* backend/cpu/index.ts:62-79 */
int orc_bit_reverse_u32(u32 *v, size_t n) {
    if (n == 0 || (n & (n - 1)) != 0) return ORC_ERR_NOT_POW2;
    u32 lg = ilog2(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = orc_bit_reverse_index((u32)i, lg);
        if (j > i) { u32 t = v[i]; v[i] = v[j]; v[j] = t; }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ circle group */
static const orc_point GEN = {2, 1268011823}; /* circle.ts:137 */
#define IDX_MASK 0x7fffffffu                  /* indices live mod 2^31, circle.ts:163-165 */
/* circle.ts:101-105 */
orc_point orc_point_add(orc_point p, orc_point q) {
    return (orc_point){orc_m31_sub(orc_m31_mul(p.x, q.x), orc_m31_mul(p.y, q.y)),
                       orc_m31_add(orc_m31_mul(p.x, q.y), orc_m31_mul(p.y, q.x))};
}
/* circle.ts:58-70,172-174: double-and-add scalar multiplication of the generator */
orc_point orc_index_to_point(u32 idx) {
    idx &= IDX_MASK;
    orc_point res = {1, 0}, cur = GEN;
    while (idx) {
        if (idx & 1) res = orc_point_add(res, cur);
        cur = orc_point_add(cur, cur);
        idx >>= 1;
    }
    return res;
}
u32 orc_subgroup_gen(u32 log_size) { return log_size == 0 ? 0u : (1u << (31 - log_size)); } /* 2^31 == 0 mod 2^31 */
u32 orc_half_odds_initial(u32 log_size) { return orc_subgroup_gen(log_size + 2); }
u32 orc_odds_initial(u32 log_size) { return orc_subgroup_gen(log_size + 1); }
static u32 coset_index_at(u32 initial, u32 log_size, u32 i) {
    return (initial + (u32)((u64)orc_subgroup_gen(log_size) * i)) & IDX_MASK;
}
orc_point orc_coset_at(u32 initial, u32 log_size, u32 i) { return orc_index_to_point(coset_index_at(initial, log_size, i)); }
/* poly/circle/domain.ts:64-88 */
orc_point orc_circle_domain_at(u32 half_initial, u32 half_log, u32 i) {
    u32 half = 1u << half_log;
    if (i < half) return orc_index_to_point(coset_index_at(half_initial, half_log, i));
    u32 idx = coset_index_at(half_initial, half_log, i - half);
    return orc_index_to_point((0x80000000u - idx) & IDX_MASK);
}

/* ------------------------------------------------------------------ twiddles */
/* backend/cpu/circle.ts:210-221 (slowPrecomputeTwiddles) + :223-239 (itwiddles = elementwise inverse) */
int orc_precompute_twiddles(u32 coset_initial, u32 log_size, u32 *buf, u32 *ibuf) {
    u32 init = coset_initial & IDX_MASK, lg = log_size;
    size_t off = 0;
    for (u32 lvl = 0; lvl < log_size; lvl++) {
        size_t half = ((size_t)1 << lg) / 2;
        orc_point cur = orc_index_to_point(init);
        orc_point step = orc_index_to_point(orc_subgroup_gen(lg));
        for (size_t k = 0; k < half; k++) { /* Coset.iter(): cur += step, circle.ts:293-310 */
            buf[off + k] = cur.x;
            cur = orc_point_add(cur, step);
        }
        orc_bit_reverse_u32(buf + off, half);
        off += half;
        init = (init * 2u) & IDX_MASK; /* Coset.double(), circle.ts:253-256 */
        lg -= 1;
    }
    buf[off] = 1;
    if (ibuf) {
        size_t n = (size_t)1 << log_size;
        for (size_t i = 0; i < n; i++) {
            if (buf[i] == 0) return ORC_ERR_ZERO_INVERSE;
            ibuf[i] = minv(buf[i]);
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ CFFT */
/* fft.ts:12-17 / :25-30 */
static void bf(u32 *v0, u32 *v1, u32 t) {
    u32 tmp = orc_m31_mul(*v1, t);
    u32 a = orc_m31_add(*v0, tmp), b = orc_m31_sub(*v0, tmp);
    *v0 = a; *v1 = b;
}
static void ibf(u32 *v0, u32 *v1, u32 t) {
    u32 a = orc_m31_add(*v0, *v1), b = orc_m31_mul(orc_m31_sub(*v0, *v1), t);
    *v0 = a; *v1 = b;
}
/* backend/cpu/circle.ts:243-257 */
static void layer_loop(u32 *v, u32 i, size_t h, u32 t, int inverse) {
    for (size_t l = 0; l < ((size_t)1 << i); l++) {
        size_t i0 = (h << (i + 1)) + l, i1 = i0 + ((size_t)1 << i);
        if (inverse) ibf(&v[i0], &v[i1], t); else bf(&v[i0], &v[i1], t);
    }
}
/* poly/utils.ts:78-100: lineTw[j] = buf[L - 2^(n-1-j) .. L - 2^(n-2-j)), j = 0..n-2 */
static const u32 *line_tw(const u32 *buf, size_t L, u32 n, u32 j, size_t *len) {
    *len = (size_t)1 << (n - 2 - j);
    return buf + L - 2 * (*len);
}
/* backend/cpu/circle.ts:270-278 */
static u32 *circle_tw(const u32 *first, size_t len) {
    u32 *res = (u32 *)malloc((len ? 2 * len : 1) * sizeof(u32));
    for (size_t i = 0; i + 1 < len; i += 2) {
        u32 x = first[i], y = first[i + 1];
        res[2 * i] = y; res[2 * i + 1] = orc_m31_neg(y); res[2 * i + 2] = orc_m31_neg(x); res[2 * i + 3] = x;
    }
    return res;
}
static void swap57(u32 *v) { u32 t = v[5]; v[5] = v[7]; v[7] = t; }

/* backend/cpu/circle.ts:84-134 */
int orc_cfft_evaluate(u32 *v, u32 n, u32 half_initial, const u32 *tw, u32 tw_log, int compat) {
    if (n == 0) return ORC_ERR_BAD_ARG;
    size_t L = (size_t)1 << tw_log;
    if (n == 1) { /* :93-98 */
        orc_point p = orc_index_to_point(half_initial);
        bf(&v[0], &v[1], p.y);
        return ORC_OK;
    }
    if (n == 2) { /* :99-110 */
        orc_point p = orc_index_to_point(half_initial);
        bf(&v[0], &v[2], p.x); bf(&v[1], &v[3], p.x);
        bf(&v[0], &v[1], p.y); bf(&v[2], &v[3], orc_m31_neg(p.y));
        return ORC_OK;
    }
    if (((size_t)1 << (n - 1)) > L) return ORC_ERR_NOT_ENOUGH_TWIDDLES;
    for (int j = (int)n - 2; j >= 0; j--) { /* :115-118 */
        size_t len; const u32 *lt = line_tw(tw, L, n, (u32)j, &len);
        for (size_t h = 0; h < len; h++) layer_loop(v, (u32)j + 1, h, lt[h], 0);
    }
    size_t len0; const u32 *l0 = line_tw(tw, L, n, 0, &len0);
    u32 *ct = circle_tw(l0, len0);
    for (size_t h = 0; h < 2 * len0; h++) layer_loop(v, 0, h, ct[h], 0); /* :121 */
    free(ct);
    if (compat && n == 3) swap57(v); /* :127-131 */
    return ORC_OK;
}
/* backend/cpu/circle.ts:136-207 */
int orc_cfft_interpolate(u32 *v, u32 n, u32 half_initial, const u32 *itw, u32 tw_log, int compat) {
    if (n == 0) return ORC_ERR_BAD_ARG;
    size_t L = (size_t)1 << tw_log, N = (size_t)1 << n;
    if (compat && n == 3) swap57(v); /* :145-151 */
    if (n == 1) { /* :153-163 — one shared inversion */
        orc_point p = orc_index_to_point(half_initial);
        u32 yn = orc_m31_mul(p.y, 2);
        if (yn == 0) return ORC_ERR_ZERO_INVERSE;
        u32 yni = minv(yn), yi = orc_m31_mul(yni, 2), ni = orc_m31_mul(yni, p.y);
        ibf(&v[0], &v[1], yi);
        v[0] = orc_m31_mul(v[0], ni); v[1] = orc_m31_mul(v[1], ni);
        return ORC_OK;
    }
    if (n == 2) { /* :164-185 */
        orc_point p = orc_index_to_point(half_initial);
        u32 xyn = orc_m31_mul(orc_m31_mul(p.x, p.y), 4);
        if (xyn == 0) return ORC_ERR_ZERO_INVERSE;
        u32 xyni = minv(xyn);
        u32 xi = orc_m31_mul(orc_m31_mul(xyni, p.y), 4), yi = orc_m31_mul(orc_m31_mul(xyni, p.x), 4);
        u32 ni = orc_m31_mul(orc_m31_mul(xyni, p.x), p.y);
        ibf(&v[0], &v[1], yi); ibf(&v[2], &v[3], orc_m31_neg(yi));
        ibf(&v[0], &v[2], xi); ibf(&v[1], &v[3], xi);
        for (int i = 0; i < 4; i++) v[i] = orc_m31_mul(v[i], ni);
        return ORC_OK;
    }
    if (((size_t)1 << (n - 1)) > L) return ORC_ERR_NOT_ENOUGH_TWIDDLES;
    size_t len0; const u32 *l0 = line_tw(itw, L, n, 0, &len0);
    u32 *ct = circle_tw(l0, len0);
    for (size_t h = 0; h < 2 * len0; h++) layer_loop(v, 0, h, ct[h], 1); /* :190-192 */
    free(ct);
    for (u32 j = 0; j + 2 <= n; j++) { /* :195-199 */
        size_t len; const u32 *lt = line_tw(itw, L, n, j, &len);
        for (size_t h = 0; h < len; h++) layer_loop(v, j + 1, h, lt[h], 1);
    }
    u32 inv = minv(orc_m31_reduce((u64)N)); /* :202-205 */
    for (size_t i = 0; i < N; i++) v[i] = orc_m31_mul(v[i], inv);
    return ORC_OK;
}

/* poly/utils.ts:36-59 with values lifted to QM31 (circle.ts:60) */
static orc_qm31 fold_rec(const u32 *vals, size_t n, const orc_qm31 *factors) {
    if (n == 1) return qm31_from_m31(vals[0]);
    orc_qm31 lo = fold_rec(vals, n / 2, factors + 1);
    orc_qm31 hi = fold_rec(vals + n / 2, n / 2, factors + 1);
    return orc_qm31_add(lo, orc_qm31_mul(hi, factors[0]));
}
/* backend/cpu/circle.ts:52-69 */
orc_qm31 orc_eval_at_point(const u32 *coeffs, u32 n, orc_spoint p) {
    if (n == 0) return qm31_from_m31(coeffs[0]);
    orc_qm31 maps[32];
    u32 cnt = 0;
    maps[cnt++] = p.y;
    orc_qm31 x = p.x;
    for (u32 i = 1; i < n; i++) {
        maps[cnt++] = x;
        orc_qm31 sx = orc_qm31_mul(x, x); /* circle.ts:37-40 double_x = 2x^2 - 1 */
        x = orc_qm31_sub(orc_qm31_add(sx, sx), QONE);
    }
    orc_qm31 rev[32];
    for (u32 i = 0; i < cnt; i++) rev[i] = maps[cnt - 1 - i];
    return fold_rec(coeffs, (size_t)1 << n, rev);
}

/* ------------------------------------------------------------------ FRI */
static orc_qm31 ld4(const u32 *const c[4], size_t i) { return (orc_qm31){c[0][i], c[1][i], c[2][i], c[3][i]}; }
static void st4(u32 *const c[4], size_t i, orc_qm31 v) { c[0][i] = v.a; c[1][i] = v.b; c[2][i] = v.c; c[3][i] = v.d; }
/* fft.ts:25-30 on QM31 with an M31 twiddle */
static void ibf_q(orc_qm31 *v0, orc_qm31 *v1, u32 t) {
    orc_qm31 a = orc_qm31_add(*v0, *v1), b = orc_qm31_mul_m31(orc_qm31_sub(*v0, *v1), t);
    *v0 = a; *v1 = b;
}
/* fri.ts:120-152 */
int orc_fold_line(const u32 *const in[4], u32 log_n, u32 coset_initial, orc_qm31 alpha, u32 *const out[4]) {
    size_t n = (size_t)1 << log_n;
    if (n < 2) return ORC_ERR_TOO_SMALL;
    for (size_t i = 0; i < n / 2; i++) {
        orc_qm31 f0 = ld4(in, 2 * i), f1 = ld4(in, 2 * i + 1);
        u32 x = orc_coset_at(coset_initial, log_n, orc_bit_reverse_index((u32)(i << 1), log_n)).x;
        if (x == 0) return ORC_ERR_ZERO_INVERSE;
        ibf_q(&f0, &f1, minv(x));
        st4(out, i, orc_qm31_add(f0, orc_qm31_mul(alpha, f1)));
    }
    return ORC_OK;
}
/* fri.ts:162-192 */
int orc_fold_circle_into_line(u32 *const dst[4], size_t dst_len, const u32 *const src[4], u32 log_n,
                              u32 half_initial, orc_qm31 alpha) {
    size_t n = (size_t)1 << log_n;
    if ((n >> 1) != dst_len) return ORC_ERR_LEN_MISMATCH;
    orc_qm31 alpha_sq = orc_qm31_mul(alpha, alpha);
    for (size_t i = 0; i < dst_len; i++) {
        orc_qm31 f0 = ld4(src, 2 * i), f1 = ld4(src, 2 * i + 1);
        orc_point p = orc_circle_domain_at(half_initial, log_n - 1, orc_bit_reverse_index((u32)(i << 1), log_n));
        if (p.y == 0) return ORC_ERR_ZERO_INVERSE;
        ibf_q(&f0, &f1, minv(p.y));
        orc_qm31 fp = orc_qm31_add(orc_qm31_mul(alpha, f1), f0);
        const u32 *const d[4] = {dst[0], dst[1], dst[2], dst[3]};
        st4(dst, i, orc_qm31_add(orc_qm31_mul(ld4(d, i), alpha_sq), fp));
    }
    return ORC_OK;
}
/* backend/cpu/fri.ts:97-164 */
int orc_decompose(const u32 *const in[4], size_t n, u32 *const out[4], orc_qm31 *lambda) {
    if (n == 0) return ORC_ERR_BAD_ARG;
    orc_qm31 lam;
    if (n == 1) {
        lam = orc_qm31_sub(QZERO, ld4(in, 0));
        st4(out, 0, orc_qm31_sub(ld4(in, 0), lam));
        *lambda = lam;
        return ORC_OK;
    }
    size_t half = n / 2;
    orc_qm31 a = QZERO, b = QZERO;
    for (size_t i = 0; i < half; i++) a = orc_qm31_add(a, ld4(in, i));
    for (size_t i = half; i < n; i++) b = orc_qm31_add(b, ld4(in, i));
    u32 nm = orc_m31_reduce((u64)n);
    if (nm == 0) return ORC_ERR_ZERO_INVERSE;
    lam = orc_qm31_mul_m31(orc_qm31_sub(a, b), minv(nm));
    for (size_t i = 0; i < half; i++) st4(out, i, orc_qm31_sub(ld4(in, i), lam));
    for (size_t i = half; i < n; i++) st4(out, i, orc_qm31_add(ld4(in, i), lam));
    *lambda = lam;
    return ORC_OK;
}

/* ------------------------------------------------------------------ Blake2s (RFC 7693) */
static const u32 B2_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                             0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u}; /* vcs/blake2s_ref.ts:4-7 */
static const uint8_t B2_SIGMA[10][16] = { /* vcs/blake2s_ref.ts:9-20 */
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static u32 rotr(u32 x, int r) { return (x >> r) | (x << (32 - r)); }
#define B2_G(a, b, c, d, x, y)      \
    do {                            \
        a = a + b + (x); d = rotr(d ^ a, 16); \
        c = c + d;       b = rotr(b ^ c, 12); \
        a = a + b + (y); d = rotr(d ^ a, 8);  \
        c = c + d;       b = rotr(b ^ c, 7);  \
    } while (0)
/* vcs/blake2s_ref.ts:176-230 */
void orc_blake2s_compress(const u32 h[8], const u32 m[16], u32 count_lo, u32 count_hi, u32 lastblock,
                          u32 lastnode, u32 out[8]) {
    u32 v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = B2_IV[i]; }
    v[12] ^= count_lo; v[13] ^= count_hi; v[14] ^= lastblock; v[15] ^= lastnode;
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2_SIGMA[r];
        B2_G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
        B2_G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
        B2_G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
        B2_G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
        B2_G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
        B2_G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
        B2_G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
        B2_G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) out[i] = h[i] ^ v[i] ^ v[8 + i];
}
typedef struct { u32 h[8]; uint8_t buf[64]; size_t buflen; u64 t; } b2s_state;
static void b2s_init(b2s_state *S) {
    for (int i = 0; i < 8; i++) S->h[i] = B2_IV[i];
    S->h[0] ^= 0x01010020u; /* digest 32, key 0, fanout 1, depth 1 (noble default, vcs/blake2_hash.ts:53) */
    S->buflen = 0; S->t = 0;
}
static void b2s_block(b2s_state *S, const uint8_t *blk, int last) {
    u32 m[16], o[8];
    for (int i = 0; i < 16; i++)
        m[i] = (u32)blk[4 * i] | ((u32)blk[4 * i + 1] << 8) | ((u32)blk[4 * i + 2] << 16) | ((u32)blk[4 * i + 3] << 24);
    orc_blake2s_compress(S->h, m, (u32)S->t, (u32)(S->t >> 32), last ? 0xFFFFFFFFu : 0, 0, o);
    memcpy(S->h, o, sizeof o);
}
static void b2s_update(b2s_state *S, const uint8_t *in, size_t len) {
    while (len > 0) {
        if (S->buflen == 64) { /* buffer full and more input follows -> not the last block */
            S->t += 64;
            b2s_block(S, S->buf, 0);
            S->buflen = 0;
        }
        size_t take = 64 - S->buflen;
        if (take > len) take = len;
        memcpy(S->buf + S->buflen, in, take);
        S->buflen += take; in += take; len -= take;
    }
}
static void b2s_final(b2s_state *S, uint8_t out[32]) {
    S->t += S->buflen;
    memset(S->buf + S->buflen, 0, 64 - S->buflen);
    b2s_block(S, S->buf, 1);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)S->h[i]; out[4 * i + 1] = (uint8_t)(S->h[i] >> 8);
        out[4 * i + 2] = (uint8_t)(S->h[i] >> 16); out[4 * i + 3] = (uint8_t)(S->h[i] >> 24);
    }
}
void orc_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]) {
    b2s_state S;
    b2s_init(&S);
    b2s_update(&S, msg, len);
    b2s_final(&S, out);
}
/* vcs/blake2_merkle.ts:9-24 */
void orc_hash_node(const uint8_t *left32, const uint8_t *right32, const u32 *values, size_t n_values, uint8_t out[32]) {
    b2s_state S;
    b2s_init(&S);
    if (left32 && right32) { b2s_update(&S, left32, 32); b2s_update(&S, right32, 32); }
    for (size_t i = 0; i < n_values; i++) {
        uint8_t le[4] = {(uint8_t)values[i], (uint8_t)(values[i] >> 8), (uint8_t)(values[i] >> 16), (uint8_t)(values[i] >> 24)};
        b2s_update(&S, le, 4);
    }
    b2s_final(&S, out);
}
/* vcs/test_utils.ts:17-43 (== vcs/ops.ts:16-26 contract) */
void orc_commit_on_layer(u32 log_size, const uint8_t *prev, const u32 *const *cols, size_t n_cols, uint8_t *out) {
    size_t n = (size_t)1 << log_size;
    u32 *row = (u32 *)malloc((n_cols ? n_cols : 1) * sizeof(u32));
    for (size_t i = 0; i < n; i++) {
        for (size_t c = 0; c < n_cols; c++) row[c] = cols[c][i];
        orc_hash_node(prev ? prev + 64 * i : NULL, prev ? prev + 64 * i + 32 : NULL, row, n_cols, out + 32 * i);
    }
    free(row);
}
/* vcs/prover.ts:13-30 — stable sort by length desc == keep input order within a size class */
int orc_merkle_commit(const u32 *const *cols, const u32 *log_sizes, size_t n_cols, uint8_t *layers_out, uint8_t root[32]) {
    if (n_cols == 0) { /* prover.ts:17-19: one hash of the empty message */
        orc_commit_on_layer(0, NULL, NULL, 0, layers_out);
        memcpy(root, layers_out, 32);
        return ORC_OK;
    }
    u32 max_log = 0;
    for (size_t c = 0; c < n_cols; c++) if (log_sizes[c] > max_log) max_log = log_sizes[c];
    const u32 **lc = (const u32 **)malloc(n_cols * sizeof(*lc));
    const uint8_t *prev = NULL;
    for (int lg = (int)max_log; lg >= 0; lg--) {
        size_t k = 0;
        for (size_t c = 0; c < n_cols; c++) if (log_sizes[c] == (u32)lg) lc[k++] = cols[c];
        uint8_t *dst = layers_out + 32 * (((size_t)1 << lg) - 1); /* layer k starts after 2^k - 1 hashes */
        orc_commit_on_layer((u32)lg, prev, lc, k, dst);
        prev = dst;
    }
    free(lc);
    memcpy(root, layers_out, 32);
    return ORC_OK;
}

/* ------------------------------------------------------------------ quotients */
/* Rust ComplexConjugate for QM31: (c0, -c1).  (The TS port conjugates each CM31 instead,
 * fields/qm31.ts:433-435 — DESIGN.md "reference quirks".) */
orc_qm31 orc_qm31_complex_conjugate(orc_qm31 v) { return (orc_qm31){v.a, v.b, orc_m31_neg(v.c), orc_m31_neg(v.d)}; }
/* constraints.ts:117-128 */
void orc_line_coeffs(orc_spoint point, orc_qm31 value, orc_qm31 alpha, orc_qm31 out[3]) {
    orc_qm31 a = orc_qm31_sub(orc_qm31_complex_conjugate(value), value);
    orc_qm31 c = orc_qm31_sub(orc_qm31_complex_conjugate(point.y), point.y);
    orc_qm31 b = orc_qm31_sub(orc_qm31_mul(value, c), orc_qm31_mul(a, point.y));
    out[0] = orc_qm31_mul(alpha, a); out[1] = orc_qm31_mul(alpha, b); out[2] = orc_qm31_mul(alpha, c);
}
/* backend/cpu/quotients.ts:52-116,160-178 with host-supplied constants */
int orc_accumulate_quotients_consts(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_batches,
                                    const size_t *off, const u32 *col_idx, const orc_qm31 *abc,
                                    const orc_qm31 *batch_coeff, const orc_cm31 *prx, const orc_cm31 *pry,
                                    const orc_cm31 *pix, const orc_cm31 *piy, u32 *const out[4]) {
    size_t N = (size_t)1 << log_size;
    orc_cm31 *den = (orc_cm31 *)malloc((n_batches ? n_batches : 1) * sizeof(orc_cm31));
    orc_cm31 *dinv = (orc_cm31 *)malloc((n_batches ? n_batches : 1) * sizeof(orc_cm31));
    int rc = ORC_OK;
    for (size_t row = 0; row < N && !rc; row++) {
        orc_point p = orc_circle_domain_at(half_initial, log_size - 1, orc_bit_reverse_index((u32)row, log_size));
        for (size_t b = 0; b < n_batches; b++) { /* quotients.ts:160-178 */
            orc_cm31 dx = orc_cm31_sub(prx[b], (orc_cm31){p.x, 0}), dy = orc_cm31_sub(pry[b], (orc_cm31){p.y, 0});
            den[b] = orc_cm31_sub(orc_cm31_mul(dx, piy[b]), orc_cm31_mul(dy, pix[b]));
        }
        rc = orc_cm31_batch_inverse(den, dinv, n_batches);
        if (rc) break;
        orc_qm31 acc = QZERO;
        for (size_t b = 0; b < n_batches; b++) { /* quotients.ts:89-113 */
            orc_qm31 num = QZERO;
            for (size_t j = off[b]; j < off[b + 1]; j++) {
                orc_qm31 value = orc_qm31_mul(qm31_from_m31(cols[col_idx[j]][row]), abc[3 * j + 2]);
                orc_qm31 lin = orc_qm31_add(orc_qm31_mul(abc[3 * j], qm31_from_m31(p.y)), abc[3 * j + 1]);
                num = orc_qm31_add(num, orc_qm31_sub(value, lin));
            }
            acc = orc_qm31_add(orc_qm31_mul(acc, batch_coeff[b]), orc_qm31_mul_cm31(num, dinv[b]));
        }
        st4(out, row, acc);
    }
    free(den); free(dinv);
    return rc;
}
/* backend/cpu/quotients.ts:124-152,183-191 (constants) then the row loop above */
int orc_accumulate_quotients(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols,
                             orc_qm31 random_coeff, const orc_sample_batch *batches, size_t n_batches,
                             u32 *const out[4]) {
    (void)n_cols;
    size_t total = 0;
    for (size_t b = 0; b < n_batches; b++) total += batches[b].n_cols;
    size_t *off = (size_t *)malloc((n_batches + 1) * sizeof(size_t));
    u32 *cidx = (u32 *)malloc((total ? total : 1) * sizeof(u32));
    orc_qm31 *abc = (orc_qm31 *)malloc((total ? total : 1) * 3 * sizeof(orc_qm31));
    orc_qm31 *bc = (orc_qm31 *)malloc((n_batches ? n_batches : 1) * sizeof(orc_qm31));
    size_t nb = n_batches ? n_batches : 1;
    orc_cm31 *prx = (orc_cm31 *)malloc(nb * 4 * sizeof(orc_cm31));
    orc_cm31 *pry = prx + nb, *pix = pry + nb, *piy = pix + nb;
    size_t k = 0;
    for (size_t b = 0; b < n_batches; b++) {
        off[b] = k;
        orc_qm31 alpha = QONE, coeff = QONE;
        for (size_t j = 0; j < batches[b].n_cols; j++, k++) {
            alpha = orc_qm31_mul(alpha, random_coeff); /* quotients.ts:129-131 */
            coeff = orc_qm31_mul(coeff, random_coeff); /* random_coeff.pow(n_cols), :151 */
            cidx[k] = batches[b].col_idx[j];
            orc_line_coeffs(batches[b].point, batches[b].values[j], alpha, &abc[3 * k]);
        }
        bc[b] = coeff;
        /* Rust pcs/quotients.rs denominator_inverses: Pr = point.{x,y}.0, Pi = point.{x,y}.1 */
        prx[b] = q0(batches[b].point.x); pry[b] = q0(batches[b].point.y);
        pix[b] = q1(batches[b].point.x); piy[b] = q1(batches[b].point.y);
    }
    off[n_batches] = k;
    int rc = orc_accumulate_quotients_consts(half_initial, log_size, cols, n_batches, off, cidx, abc, bc, prx, pry, pix, piy, out);
    free(off); free(cidx); free(abc); free(bc); free(prx);
    return rc;
}

/* ------------------------------------------------------------------ accumulation */
/* backend/cpu/accumulation.ts:38-49 */
void orc_accumulate(u32 *const col[4], const u32 *const other[4], size_t n) {
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) col[k][i] = orc_m31_add(col[k][i], other[k][i]);
}
/* backend/cpu/accumulation.ts:52-63 */
void orc_generate_secure_powers(orc_qm31 felt, size_t n, orc_qm31 *out) {
    orc_qm31 acc = QONE;
    for (size_t i = 0; i < n; i++) { out[i] = acc; acc = orc_qm31_mul(acc, felt); }
}
