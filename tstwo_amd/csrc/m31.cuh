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
// (a0b0 + R a1b1, a0b1 + a1b0), fields/qm31.ts:223-233
__device__ __forceinline__ qm31 qm31_mul(qm31 x, qm31 y) {
    cm31 a0 = q_c0(x), a1 = q_c1(x), b0 = q_c0(y), b1 = q_c1(y);
    cm31 c0 = cm31_add(cm31_mul(a0, b0), cm31_mul_R(cm31_mul(a1, b1)));
    cm31 c1 = cm31_add(cm31_mul(a0, b1), cm31_mul(a1, b0));
    return q_make(c0, c1);
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
