// field8.cuh — M31 arithmetic on 8 independent instances per lane, issued in priority phases (phase.cuh).
//
// The quotient, batch-inverse and fold kernels give every lane 8 rows / elements whose arithmetic is independent, so each
// operation is written opcode by opcode over the 8 instances: runs of 8 (or more) heavy instructions at kPrioHeavy
// (v_mad_u64_u32, v_alignbit, v_min: port 0 only), runs of light VOP2 at kPrioLight (add / sub / and / shifts on VGPR
// operands: either port, so they pair with another wave's heavy run).  The modulus lives in a VGPR (a literal operand makes
// an add heavy).  Every routine ends in a heavy run and closes it: the wave is back at kPrioLight on return.  Canonical in,
// canonical out, bit-identical to m31.cuh.
#pragma once
#include "m31.cuh"

namespace f8 {

// Ordering only: the 8 values pass through an empty volatile asm, so that everything computing them stays in front of the
// next phase boundary (volatile asms keep their order) and everything using them stays behind it.  No instruction.
template <class T>
__device__ __forceinline__ void pin(T (&a)[8]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}
__device__ __forceinline__ void done() { phase<kPrioLight>(); }
// A phase boundary that several arrays cross: each is pinned in front of the s_setprio (its producers stay before it) AND behind
// it (its consumers stay after it).  A value pinned in front only does not hold its consumers back: a multiply-add run whose
// operands were all pinned that way was found scheduled ahead of the s_setprio 3 that should have opened it.
template <int PRIO, class... A>
__device__ __forceinline__ void boundary(A &...arrs) {
    (pin(arrs), ...);
    phase<PRIO>();
    (pin(arrs), ...);
}

// r = a * b (products < 2^62): H 8 x mad, 8 x alignbit | L 8 x (and, add, sub) | H 8 x min
__device__ __forceinline__ void mul(u32 (&r)[8], u32 (&a)[8], u32 (&b)[8]) {
    const u32 P = vgpr_P();
    u64 p[8];
    u32 hi[8], s[8], d[8];
    pin(b);
    phase<kPrioHeavy>(a);
#pragma unroll
    for (int e = 0; e < 8; e++) p[e] = (u64)a[e] * (u64)b[e];
#pragma unroll
    for (int e = 0; e < 8; e++) hi[e] = __builtin_amdgcn_alignbit((u32)(p[e] >> 32), (u32)p[e], 31);
    pin(p);
    phase<kPrioLight>(hi);
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] = ((u32)p[e] & P) + hi[e]; d[e] = s[e] - P; }
    pin(s);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = min(s[e], d[e]);
    phase<kPrioLight>(r);          // the closing heavy run ends here: whatever follows (shuffles, the next routine) starts light
}
// r = a * b for a wave-uniform / per-lane scalar b
__device__ __forceinline__ void mul_s(u32 (&r)[8], u32 (&a)[8], u32 b) {
    u32 bb[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bb[e] = b;
    mul(r, a, bb);
}
// r = a + b, r = a - b: L 8 x (add, sub) | H 8 x min
__device__ __forceinline__ void add(u32 (&r)[8], u32 (&a)[8], u32 (&b)[8]) {
    const u32 P = vgpr_P();
    u32 s[8], d[8];
    pin(b);
    phase<kPrioLight>(a);
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] = a[e] + b[e]; d[e] = s[e] - P; }
    pin(s);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = min(s[e], d[e]);
    phase<kPrioLight>(r);          // the closing heavy run ends here: whatever follows (shuffles, the next routine) starts light
}
__device__ __forceinline__ void sub(u32 (&r)[8], u32 (&a)[8], u32 (&b)[8]) {
    const u32 P = vgpr_P();
    u32 s[8], d[8];
    pin(b);
    phase<kPrioLight>(a);
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] = a[e] - b[e]; d[e] = s[e] + P; }
    pin(s);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = min(s[e], d[e]);
    phase<kPrioLight>(r);          // the closing heavy run ends here: whatever follows (shuffles, the next routine) starts light
}
// r[e] = bit e of MASK ? a[e] - b[e] : a[e] + b[e] (the rows of a lane differ by signs: conjugate / antipodal domain points)
template <unsigned MASK>
__device__ __forceinline__ void addsub(u32 (&r)[8], u32 (&a)[8], u32 (&b)[8]) {
    const u32 P = vgpr_P();
    u32 s[8], d[8];
    pin(b);
    phase<kPrioLight>(a);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if ((MASK >> e) & 1u) { s[e] = a[e] - b[e]; d[e] = s[e] + P; }
        else { s[e] = a[e] + b[e]; d[e] = s[e] - P; }
    }
    pin(s);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = min(s[e], d[e]);
    phase<kPrioLight>(r);
}
// r = P - a (a canonical; a = 0 gives P, which is NOT canonical: only for operands of a multiplication, where P acts as 0)
__device__ __forceinline__ void neg_operand(u32 (&r)[8], const u32 (&a)[8]) {
    const u32 P = vgpr_P();
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = P - a[e];
}
// 64-bit accumulators (sums of at most 4 "units" of (P-1)P, m31.cuh: qm31_mul) -> canonical M31:
// x = t1 + 2^31 t2 + 2^63 t3 = t1 + (t2 & P) + (t2 >> 31) + 2 t3 (mod P).  TOP = false: the sums are known to be below 2^63
// (at most 2 units), t3 = 0.  Three-operand sums are kept apart on purpose (the compiler fuses a + b + c into a heavy v_add3).
// L 8 x (and [, shift, and]) | H 8 x alignbit | L 8 x (and, add, shift [, add], sub) | H 8 x min | L 8 x (add, sub) | H 8 x min
template <bool TOP = true>
__device__ __forceinline__ void reduce(u32 (&r)[8], u64 (&acc)[8]) {
    const u32 P = vgpr_P();
    u32 t1[8], t2[8], t3d[8], s[8], d[8], w[8];
    phase<kPrioLight>(acc);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        t1[e] = (u32)acc[e] & P;
        t3d[e] = TOP ? ((u32)(acc[e] >> 32) >> 30) & 2u : 0u;          // 2 x bit 63
    }
    pin(t1);
    if (TOP) pin(t3d);
    phase<kPrioHeavy>(acc);          // (the rotates read acc only: it has to pass through THIS boundary to stay behind it)
#pragma unroll
    for (int e = 0; e < 8; e++) t2[e] = __builtin_amdgcn_alignbit((u32)(acc[e] >> 32), (u32)acc[e], 31);
    phase<kPrioLight>(t2);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        s[e] = t1[e] + (t2[e] & P);
        w[e] = t2[e] >> 31;
        if (TOP) w[e] += t3d[e];
        d[e] = s[e] - P;
    }
    pin(s); pin(w);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = min(s[e], d[e]);
    phase<kPrioLight>(s);
#pragma unroll
    for (int e = 0; e < 8; e++) { s[e] = s[e] + w[e]; d[e] = s[e] - P; }
    pin(s);
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = min(s[e], d[e]);
    phase<kPrioLight>(r);          // the closing heavy run ends here: whatever follows (shuffles, the next routine) starts light
}
// acc (+)= a * b: a heavy run (call between phase<kPrioHeavy> and the reduce)
__device__ __forceinline__ void mad(u64 (&acc)[8], const u32 (&a)[8], const u32 (&b)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] += (u64)a[e] * (u64)b[e];
}
__device__ __forceinline__ void mul64(u64 (&acc)[8], const u32 (&a)[8], const u32 (&b)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = (u64)a[e] * (u64)b[e];
}

// Inverses of 8 nonzero values with ONE Fermat chain: products up a binary tree (4 + 2 + 1 independent multiplications), the
// root inverted (fields/m31.ts:305-326: 37 dependent multiplications, issued at low priority — a chain cannot pair with
// itself), inverses down the tree (2 + 4 + 8).  Same 21 multiplications as Montgomery's prefix chain, but in runs of
// independent ones.  The unique inverses: what batchInverse (fields/fields.ts:66-207) returns.  Leaves the wave at kPrioLight.
__device__ __forceinline__ void inverse8(u32 (&inv)[8], u32 (&x)[8]) {
    done();
    u32 p2[4], p4[2];
#pragma unroll
    for (int i = 0; i < 4; i++) p2[i] = m31_mul(x[2 * i], x[2 * i + 1]);
    p4[0] = m31_mul(p2[0], p2[1]);
    p4[1] = m31_mul(p2[2], p2[3]);
    const u32 top = m31_inv(m31_mul(p4[0], p4[1]));
    const u32 i4[2] = {m31_mul(top, p4[1]), m31_mul(top, p4[0])};
    u32 i2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) i2[i] = m31_mul(i4[i >> 1], p2[i ^ 1]);
    u32 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = i2[i >> 1]; b[i] = x[i ^ 1]; }
    mul(inv, a, b);
    done();
}

}  // namespace f8
