// m31.cuh — Mersenne-31 / CM31 / QM31 arithmetic for gfx950 device code.
//
// Semantics follow the reference's fields (packages/core/src/fields/{m31,cm31,qm31}.ts): canonical
// representatives in [0, P), P = 2^31 - 1.  Every function here maps canonical inputs to canonical
// outputs, so stored results are bit-identical to the reference's without a final fix-up pass.
//
// Instruction notes (CDNA4): a 32x32->64 multiply is one v_mad_u64_u32 (quarter rate); the 31-bit
// fold is and/alignbit/add + a min-based conditional subtract; add/sub are add + sub + v_min_u32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define M31_P 2147483647u
#ifdef __HIPCC__
#include "phase.cuh"
#endif

__device__ __forceinline__ u32 m31_add(u32 a, u32 b) {
    u32 s = a + b;               // < 2P < 2^32
    return min(s, s - M31_P);    // s < P: s-P wraps high, min keeps s
}
__device__ __forceinline__ u32 m31_sub(u32 a, u32 b) {
    u32 d = a - b;               // wraps when a < b
    return min(d, d + M31_P);    // wrapped: d+P wraps back into [0,P)
}
__device__ __forceinline__ u32 m31_neg(u32 a) { return m31_sub(0u, a); }
__device__ __forceinline__ u32 m31_double(u32 a) { return m31_add(a, a); }
__device__ __forceinline__ u32 m31_reduce64(u64 p) {  // p < 2^62: fields/m31.ts:89-101 result
    u32 lo = (u32)p & M31_P;
    u32 hi = (u32)(p >> 31);
    u32 s = lo + hi;             // < 2P, so one conditional subtract canonicalises
    return min(s, s - M31_P);
}
__device__ __forceinline__ u32 m31_mul(u32 a, u32 b) { return m31_reduce64((u64)a * (u64)b); }
// any x < 2^64 -> canonical: x = t1 + 2^31 t2 + 2^63 t3 (31 + 32 + 1 bits) = t1 + (t2 & P) + (t2 >> 31) + 2 t3 (mod P); the form
// qm31_mul uses (3 heavy + 10 light instructions; no 64-bit adds)
__device__ __forceinline__ u32 m31_reduce_u64(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    const u32 t2 = __builtin_amdgcn_alignbit(hi, lo, 31), t3 = hi >> 31;
    u32 s = (lo & M31_P) + (t2 & M31_P);      // <= 2P
    s = min(s, s - M31_P);                    // <= P
    s = s + (t2 >> 31) + t3 + t3;             // <= P + 3
    return min(s, s - M31_P);
}

__device__ __forceinline__ u32 m31_sqr(u32 a) { return m31_mul(a, a); }
// a*b + c*d with one reduction (products < 2^62 each, sum < 2^63)
__device__ __forceinline__ u32 m31_mul_add_mul(u32 a, u32 b, u32 c, u32 d) {
    u64 p = (u64)a * b + (u64)c * d;             // < 2^63
    u64 f = (p & M31_P) + (p >> 31);             // < 2^31 + 2^32
    return m31_reduce64(f);
}
__device__ __forceinline__ u32 m31_sqn(u32 v, int n) {
    for (int i = 0; i < n; i++) v = m31_sqr(v);
    return v;
}
// v^(P-2): the reference's fixed 37-multiplication chain (fields/m31.ts:305-326)
__device__ __forceinline__ u32 m31_inv(u32 v) {
    u32 t0 = m31_mul(m31_sqn(v, 2), v);
    u32 t1 = m31_mul(m31_sqn(t0, 1), t0);
    u32 t2 = m31_mul(m31_sqn(t1, 3), t0);
    u32 t3 = m31_mul(m31_sqn(t2, 1), t0);
    u32 t4 = m31_mul(m31_sqn(t3, 8), t3);
    u32 t5 = m31_mul(m31_sqn(t4, 8), t3);
    return m31_mul(m31_sqn(t5, 7), t2);
}

// ------------------------------------------------------------------ CM31 = M31[i]/(i^2+1)
struct cm31 { u32 a, b; };
__device__ __forceinline__ cm31 cm31_add(cm31 x, cm31 y) { return {m31_add(x.a, y.a), m31_add(x.b, y.b)}; }
__device__ __forceinline__ cm31 cm31_sub(cm31 x, cm31 y) { return {m31_sub(x.a, y.a), m31_sub(x.b, y.b)}; }
__device__ __forceinline__ cm31 cm31_neg(cm31 x) { return {m31_neg(x.a), m31_neg(x.b)}; }
// (ac - bd, ad + bc), fields/cm31.ts:139-149
__device__ __forceinline__ cm31 cm31_mul(cm31 x, cm31 y) {
    u32 re = m31_sub(m31_mul(x.a, y.a), m31_mul(x.b, y.b));
    u32 im = m31_mul_add_mul(x.a, y.b, x.b, y.a);
    return {re, im};
}
__device__ __forceinline__ cm31 cm31_mul_m31(cm31 x, u32 m) { return {m31_mul(x.a, m), m31_mul(x.b, m)}; }
__device__ __forceinline__ cm31 cm31_sqr(cm31 x) { return cm31_mul(x, x); }
// conj / (a^2 + b^2), fields/cm31.ts:237-251 (caller guarantees x != 0)
__device__ __forceinline__ cm31 cm31_inv(cm31 x) {
    u32 ni = m31_inv(m31_mul_add_mul(x.a, x.a, x.b, x.b));
    return {m31_mul(x.a, ni), m31_mul(m31_neg(x.b), ni)};
}
__device__ __forceinline__ bool cm31_is_zero(cm31 x) { return (x.a | x.b) == 0; }

// ------------------------------------------------------------------ QM31 = CM31[u]/(u^2 - (2+i))
struct qm31 { u32 a, b, c, d; };
__device__ __forceinline__ cm31 q_c0(qm31 x) { return {x.a, x.b}; }
__device__ __forceinline__ cm31 q_c1(qm31 x) { return {x.c, x.d}; }
__device__ __forceinline__ qm31 q_make(cm31 c0, cm31 c1) { return {c0.a, c0.b, c1.a, c1.b}; }
__device__ __forceinline__ qm31 qm31_add(qm31 x, qm31 y) {
    return {m31_add(x.a, y.a), m31_add(x.b, y.b), m31_add(x.c, y.c), m31_add(x.d, y.d)};
}
__device__ __forceinline__ qm31 qm31_sub(qm31 x, qm31 y) {
    return {m31_sub(x.a, y.a), m31_sub(x.b, y.b), m31_sub(x.c, y.c), m31_sub(x.d, y.d)};
}
__device__ __forceinline__ qm31 qm31_neg(qm31 x) { return {m31_neg(x.a), m31_neg(x.b), m31_neg(x.c), m31_neg(x.d)}; }
// R * z with R = 2 + i: (2a - b, a + 2b)
__device__ __forceinline__ cm31 cm31_mul_R(cm31 z) {
    return {m31_sub(m31_double(z.a), z.b), m31_add(z.a, m31_double(z.b))};
}
// (a0b0 + R a1b1, a0b1 + a1b0), fields/qm31.ts:223-233 — the reference's formulation, kept as the readable statement of
// what qm31_mul below computes (and used by it nowhere)
__device__ __forceinline__ qm31 qm31_mul_ref(qm31 x, qm31 y) {
    cm31 a0 = q_c0(x), a1 = q_c1(x), b0 = q_c0(y), b1 = q_c1(y);
    cm31 c0 = cm31_add(cm31_mul(a0, b0), cm31_mul_R(cm31_mul(a1, b1)));
    cm31 c1 = cm31_add(cm31_mul(a0, b1), cm31_mul(a1, b0));
    return q_make(c0, c1);
}
// The same product written out over the four M31 coordinates, with every minus sign moved into an operand (P - y) and
// every factor 2 into an operand (2y < 2^32), so that each coordinate is a sum of products accumulated in 64 bits by
// v_mad_u64_u32 chains — at most 4 "units" of (P-1)P < 2^62 per accumulator, 4 (2^62 - 3 2^31 + 2) < 2^64:
//   r0 = [x0 y0 + x1 (P-y1) + x2 (2 y2)] + [x3 (2(P-y3)) + x2 (P-y3) + x3 (P-y2)]      (re of a0 b0 + (2+i) a1 b1)
//   r1 = [x0 y1 + x1 y0 + x2 y2 + x3 (P-y3)] + [x2 (2 y3) + x3 (2 y2)]                 (im of the same)
//   r2 =  x0 y2 + x1 (P-y3) + x2 y0 + x3 (P-y1)                                         (re of a0 b1 + a1 b0)
//   r3 =  x0 y3 + x1 y2 + x2 y1 + x3 y0                                                 (im)
// 20 multiply-adds and 6 reductions instead of 16 multiplies, 16 reductions and 14 modular additions (~132 -> ~105
// instructions), and the six accumulators are independent, so the whole product is issued in priority phases (phase.cuh):
// 20 x v_mad_u64_u32 | 12 light | 6 x v_alignbit | 30 light | 6 x v_min | 18 light | 6 x v_min.
// Reduction of a 64-bit x = t1 + 2^31 t2 + 2^63 t3 (t1 31 bits, t2 32 bits, t3 one bit): 2^31 = 1, 2^63 = 2 (mod P), and
// t2 = u + 2^31 w, so x = t1 + u + w + 2 t3: s = t1 + u <= 2P -> one conditional subtract (<= P), + w + 2 t3 <= P + 3 ->
// one more: canonical.  Canonical in, canonical out; bit-identical to qm31_mul_ref (tests: the reference's qm31 vectors
// through tstwo_qm31_mul, the fuzz suite).
__device__ __forceinline__ qm31 qm31_mul(qm31 x, qm31 y) {
    const u32 P = vgpr_P();
    u32 pre[6] = {P - y.b, P - y.c, P - y.d, y.c + y.c, y.d + y.d, 0u};     // ny1, ny2, ny3, dy2, dy3, dny3
    pre[5] = pre[2] + pre[2];
    phase<kPrioHeavy>(pre);
    const u32 ny1 = pre[0], ny2 = pre[1], ny3 = pre[2], dy2 = pre[3], dy3 = pre[4], dny3 = pre[5];
    u64 acc[6];
    acc[0] = (u64)x.a * y.a;  acc[1] = (u64)x.d * dny3; acc[2] = (u64)x.a * y.b; acc[3] = (u64)x.c * dy3;
    acc[4] = (u64)x.a * y.c;  acc[5] = (u64)x.a * y.d;
    acc[0] += (u64)x.b * ny1; acc[1] += (u64)x.c * ny3; acc[2] += (u64)x.b * y.a; acc[3] += (u64)x.d * dy2;
    acc[4] += (u64)x.b * ny3; acc[5] += (u64)x.b * y.c;
    acc[0] += (u64)x.c * dy2; acc[1] += (u64)x.d * ny2; acc[2] += (u64)x.c * y.c;
    acc[4] += (u64)x.c * y.a; acc[5] += (u64)x.c * y.b;
    acc[2] += (u64)x.d * ny3; acc[4] += (u64)x.d * ny1; acc[5] += (u64)x.d * y.a;
    phase<kPrioLight>(acc);
    u32 t1[6], t3[6], t2[6], s[6], d[6], w[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { t1[i] = (u32)acc[i] & P; t3[i] = (u32)(acc[i] >> 32) >> 31; }
    phase<kPrioHeavy>(t1, t3);
#pragma unroll
    for (int i = 0; i < 6; i++) t2[i] = __builtin_amdgcn_alignbit((u32)(acc[i] >> 32), (u32)acc[i], 31);
    phase<kPrioLight>(t2);
#pragma unroll
    for (int i = 0; i < 6; i++) { s[i] = t1[i] + (t2[i] & P); w[i] = (t2[i] >> 31) + t3[i]; d[i] = s[i] - P; }
    phase<kPrioHeavy>(d, w);
#pragma unroll
    for (int i = 0; i < 6; i++) s[i] = min(s[i], d[i]);
    phase<kPrioLight>(s);
#pragma unroll
    for (int i = 0; i < 6; i++) { s[i] = s[i] + w[i] + t3[i]; d[i] = s[i] - P; }
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int i = 0; i < 6; i++) s[i] = min(s[i], d[i]);
    phase<kPrioLight>(s);
    return {m31_add(s[0], s[1]), m31_add(s[2], s[3]), s[4], s[5]};
}
__device__ __forceinline__ qm31 qm31_mul_m31(qm31 x, u32 m) {
    return {m31_mul(x.a, m), m31_mul(x.b, m), m31_mul(x.c, m), m31_mul(x.d, m)};
}
__device__ __forceinline__ qm31 qm31_mul_cm31(qm31 x, cm31 m) { return q_make(cm31_mul(q_c0(x), m), cm31_mul(q_c1(x), m)); }
__device__ __forceinline__ qm31 qm31_from_m31(u32 v) { return {v, 0u, 0u, 0u}; }
__device__ __forceinline__ bool qm31_is_zero(qm31 x) { return (x.a | x.b | x.c | x.d) == 0; }
// fields/qm31.ts:282-305 (caller guarantees x != 0)
__device__ __forceinline__ qm31 qm31_inv(qm31 x) {
    cm31 b2 = cm31_sqr(q_c1(x));
    cm31 ib2 = {m31_neg(b2.b), b2.a};
    cm31 denom = cm31_sub(cm31_sqr(q_c0(x)), cm31_add(cm31_add(b2, b2), ib2));
    cm31 di = cm31_inv(denom);
    return q_make(cm31_mul(q_c0(x), di), cm31_neg(cm31_mul(q_c1(x), di)));
}

// ------------------------------------------------------------------ circle group over M31 (circle.ts)
struct cpoint { u32 x, y; };
__device__ __forceinline__ cpoint cpoint_add(cpoint p, cpoint q) {
    return {m31_sub(m31_mul(p.x, q.x), m31_mul(p.y, q.y)), m31_mul_add_mul(p.x, q.y, p.y, q.x)};
}
// idx * GEN by double-and-add over a 31-entry table of GEN*2^k (filled by the host at init)
__device__ __forceinline__ cpoint cpoint_from_index(u32 idx, const cpoint *__restrict__ gen_pow2) {
    cpoint r = {1u, 0u};
    idx &= 0x7fffffffu;
#pragma unroll 1
    for (int k = 0; idx != 0; k++, idx >>= 1)
        if (idx & 1u) r = cpoint_add(r, gen_pow2[k]);
    return r;
}

// the same from the windowed table (Context::gen_win: [w][k] = (k 2^(8w)) GEN): 4 lookups and 3 additions instead of one addition
// per set bit of idx
__device__ __forceinline__ cpoint cpoint_from_index_win(u32 idx, const cpoint *__restrict__ gen_win) {
    idx &= 0x7fffffffu;
    cpoint r = gen_win[idx & 255u];
    r = cpoint_add(r, gen_win[256u + ((idx >> 8) & 255u)]);
    r = cpoint_add(r, gen_win[512u + ((idx >> 16) & 255u)]);
    return cpoint_add(r, gen_win[768u + (idx >> 24)]);
}

// fft.ts:12-17 / :25-30
__device__ __forceinline__ void m31_butterfly(u32 &v0, u32 &v1, u32 t) {
    u32 tmp = m31_mul(v1, t);
    u32 a = m31_add(v0, tmp);
    v1 = m31_sub(v0, tmp);
    v0 = a;
}
__device__ __forceinline__ void m31_ibutterfly(u32 &v0, u32 &v1, u32 t) {
    u32 a = m31_add(v0, v1);
    v1 = m31_mul(m31_sub(v0, v1), t);
    v0 = a;
}
