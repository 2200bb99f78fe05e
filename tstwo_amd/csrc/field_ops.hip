// field_ops.hip — column arithmetic, batch inversion, bit-reversal and twiddle-tree generation.
//
// All of these are HBM-streaming kernels: one coalesced 16-byte access per lane where alignment
// allows, grid capped at a few waves per SIMD and grid-strided (guide §6 G11/G13).  Algorithmic
// bytes per element: add/sub/mul 12 B, neg 8 B, m31 batch inverse 8 B,
// qm31 batch inverse 32 B, bit reverse 8 B.
#include "common.h"
#include "field8.cuh"

using namespace tstwo;

namespace {

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_NEG = 3 };

template <int OP>
__device__ __forceinline__ u32 apply_op(u32 a, u32 b) {
    if (OP == OP_ADD) return m31_add(a, b);
    if (OP == OP_SUB) return m31_sub(a, b);
    if (OP == OP_MUL) return m31_mul(a, b);
    return m31_neg(a);
}

// fields/m31.ts:147-173 applied per element; 4 elements (16 B) per lane per iteration
template <int OP>
__global__ void __launch_bounds__(256) k_m31_binop_vec4(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                        uint4 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint4 x = a[i];
        uint4 y = (OP == OP_NEG) ? x : b[i];
        uint4 r;
        r.x = apply_op<OP>(x.x, y.x);
        r.y = apply_op<OP>(x.y, y.y);
        r.z = apply_op<OP>(x.z, y.z);
        r.w = apply_op<OP>(x.w, y.w);
        out[i] = r;
    }
}
template <int OP>
__global__ void __launch_bounds__(256) k_m31_binop_scalar(const u32 *__restrict__ a, const u32 *__restrict__ b,
                                                          u32 *__restrict__ out, size_t begin, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = apply_op<OP>(a[i], (OP == OP_NEG) ? 0u : b[i]);
}

template <int OP>
int launch_binop(const u32 *a, const u32 *b, u32 *out, size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    if (!a || !out || (OP != OP_NEG && !b)) return set_error(TSTWO_ERR_BAD_ARG, "null column pointer");
    Context &c = ctx();
    bool aligned = (((uintptr_t)a | (uintptr_t)out | (uintptr_t)(OP == OP_NEG ? a : b)) & 15) == 0;
    size_t n4 = aligned ? n / 4 : 0;
    unsigned max_blocks = (unsigned)c.n_cus * 64;
    if (n4) {
        unsigned blocks = ceil_div(n4, 256);
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL(k_m31_binop_vec4<OP>, dim3(blocks), dim3(256), 0, c.stream, (const uint4 *)a,
                           (const uint4 *)b, (uint4 *)out, n4);
    }
    if (n4 * 4 < n) {
        size_t rest = n - n4 * 4;
        unsigned blocks = ceil_div(rest, 256);
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL(k_m31_binop_scalar<OP>, dim3(blocks), dim3(256), 0, c.stream, a, b, out, n4 * 4, n);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

// ---------------------------------------------------------------- QM31 SoA elementwise
__global__ void __launch_bounds__(256) k_qm31_mul(CSoa4 a, CSoa4 b, Soa4 o, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        qm31 x = {a.p[0][i], a.p[1][i], a.p[2][i], a.p[3][i]};
        qm31 y = {b.p[0][i], b.p[1][i], b.p[2][i], b.p[3][i]};
        qm31 r = qm31_mul(x, y);
        o.p[0][i] = r.a; o.p[1][i] = r.b; o.p[2][i] = r.c; o.p[3][i] = r.d;
    }
}
// backend/cpu/accumulation.ts:38-49: col[k][i] += other[k][i]
// VEC: 16-byte accesses (columns 16-byte aligned, n a multiple of 4)
template <bool VEC>
__global__ void __launch_bounds__(256) k_secure_accumulate(Soa4 col, CSoa4 other, size_t n) {
    u32 *c = col.p[blockIdx.y];
    const u32 *o = other.p[blockIdx.y];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
            const uint4 x = gload4(c + 4 * i), y = gload4(o + 4 * i);
            gstore4(c + 4 * i, make_uint4(m31_add(x.x, y.x), m31_add(x.y, y.y), m31_add(x.z, y.z), m31_add(x.w, y.w)));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c[i] = m31_add(c[i], o[i]);
    }
}

// ---------------------------------------------------------------- batch inverse
// Montgomery's trick per thread over K elements strided by the thread count (coalesced for every j),
// one Fermat chain (fields/m31.ts:305-326) per thread: 3(K-1)+37 multiplications per K elements.
// The result is the unique elementwise inverse, i.e. what fields/fields.ts:66-207 returns.
// A zero input raises the flag (the reference throws "0 has no inverse") and is treated as 1.
template <int K>
__global__ void __launch_bounds__(256) k_m31_batch_inverse(const u32 *__restrict__ in, u32 *__restrict__ out, size_t n,
                                                          size_t T, u32 *flag) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    u32 x[K], pre[K];
    bool zero = false;
#pragma unroll
    for (int j = 0; j < K; j++) {
        size_t i = t + (size_t)j * T;
        u32 v = i < n ? in[i] : 1u;
        if (v == 0) { zero = true; v = 1u; }
        x[j] = v;
        pre[j] = j == 0 ? v : m31_mul(pre[j - 1], v);
    }
    if (zero) raise_flag(flag);
    u32 cur = m31_inv(pre[K - 1]);
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        size_t i = t + (size_t)j * T;
        u32 r = j == 0 ? cur : m31_mul(pre[j - 1], cur);
        cur = m31_mul(cur, x[j]);
        if (i < n) out[i] = r;
    }
}

struct CSoa2 { const u32 *p[2]; };
struct Soa2 { u32 *p[2]; };
template <int K>
__global__ void __launch_bounds__(256) k_cm31_batch_inverse(CSoa2 in, Soa2 out, size_t n, size_t T, u32 *flag) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    cm31 x[K], pre[K];
    bool zero = false;
#pragma unroll
    for (int j = 0; j < K; j++) {
        size_t i = t + (size_t)j * T;
        cm31 v = {1u, 0u};
        if (i < n) v = {in.p[0][i], in.p[1][i]};
        if (cm31_is_zero(v)) { zero = true; v = {1u, 0u}; }
        x[j] = v;
        pre[j] = j == 0 ? v : cm31_mul(pre[j - 1], v);
    }
    if (zero) raise_flag(flag);
    cm31 cur = cm31_inv(pre[K - 1]);
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        size_t i = t + (size_t)j * T;
        cm31 r = j == 0 ? cur : cm31_mul(pre[j - 1], cur);
        cur = cm31_mul(cur, x[j]);
        if (i < n) { out.p[0][i] = r.a; out.p[1][i] = r.b; }
    }
}
// The same with 16-byte accesses: lane t owns the 4 consecutive elements 4t .. 4t+3 of M groups a stride of 4T apart, so every
// coordinate column is read and written with dwordx4 instructions (the 4-byte-per-lane form above moves 134 MB in 76 us =
// 1.8 TB/s whatever K is: it is bound by the number of memory instructions, not by the Montgomery chain).
template <int M>
__global__ void __launch_bounds__(256) k_qm31_batch_inverse_v4(CSoa4 in, Soa4 out, size_t T, u32 *flag) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    constexpr int K = 4 * M;
    qm31 x[K], pre[K];
    bool zero = false;
#pragma unroll
    for (int g = 0; g < M; g++) {
        const size_t i = 4 * (t + (size_t)g * T);
        const uint4 a = gload4(in.p[0] + i), b = gload4(in.p[1] + i);
        const uint4 c = gload4(in.p[2] + i), d = gload4(in.p[3] + i);
        x[4 * g + 0] = {a.x, b.x, c.x, d.x}; x[4 * g + 1] = {a.y, b.y, c.y, d.y};
        x[4 * g + 2] = {a.z, b.z, c.z, d.z}; x[4 * g + 3] = {a.w, b.w, c.w, d.w};
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (qm31_is_zero(x[j])) { zero = true; x[j] = {1u, 0u, 0u, 0u}; }
        pre[j] = j == 0 ? x[j] : qm31_mul(pre[j - 1], x[j]);
    }
    if (zero) raise_flag(flag);
    qm31 cur = qm31_inv(pre[K - 1]);
    qm31 r[K];
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        r[j] = j == 0 ? cur : qm31_mul(pre[j - 1], cur);
        cur = qm31_mul(cur, x[j]);
    }
#pragma unroll
    for (int g = 0; g < M; g++) {
        const size_t i = 4 * (t + (size_t)g * T);
        gstore4(out.p[0] + i, make_uint4(r[4 * g].a, r[4 * g + 1].a, r[4 * g + 2].a, r[4 * g + 3].a));
        gstore4(out.p[1] + i, make_uint4(r[4 * g].b, r[4 * g + 1].b, r[4 * g + 2].b, r[4 * g + 3].b));
        gstore4(out.p[2] + i, make_uint4(r[4 * g].c, r[4 * g + 1].c, r[4 * g + 2].c, r[4 * g + 3].c));
        gstore4(out.p[3] + i, make_uint4(r[4 * g].d, r[4 * g + 1].d, r[4 * g + 2].d, r[4 * g + 3].d));
    }
}

// QM31 batch inverse through the norms.  The inverse of x = c0 + c1 u (c0 = a + bi, c1 = c + di; fields/qm31.ts:282-305) is
//   x^-1 = (c0 - c1 u) / D,  D = c0^2 - (2 + i) c1^2 in CM31,   D^-1 = conj(D) / n,  n = D.re^2 + D.im^2 in M31,
// so the only inversion is of the M31 norm n — and batchInverse (fields/fields.ts:66-207) returns the unique elementwise
// inverse whatever the schedule.  A lane owns 8 elements (two runs of 4 consecutive ones: 16-byte accesses); their 8 norms
// share one Fermat chain (f8::inverse8).  Per element: 9 + 2 + 8 multiply-adds, 10 lazy reductions and 5 multiplications
// (~210 VALU instructions) against 3 QM31 products (~320) in Montgomery's trick over QM31 values, and every step is the same
// operation on 8 independent elements, issued in priority phases (field8.cuh).
//   D.re = a a + b (P-b) + c 2(d-c)  +  d 2d          D.im = a 2b + c 2(P-d)  +  c 2(P-d) + c (P-c) + d d
//   out  = (a ir + b (P-ii),  a ii + b ir,  c (P-ir) + d ii,  c (P-ii) + d (P-ir)),   (ir, ii) = (D.re, P - D.im) / n
// (sums of at most 4 units of (P-1)P per 64-bit accumulator, doubled operands count twice: m31.cuh, qm31_mul).
__global__ void __launch_bounds__(256) k_qm31_batch_inverse_norm(CSoa4 in, Soa4 out, size_t T, u32 *flag) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const u32 P = vgpr_P();
    u32 a[8], b[8], c[8], d[8];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const size_t i = 4 * (t + (size_t)g * T);
        const uint4 va = gload4(in.p[0] + i), vb = gload4(in.p[1] + i), vc = gload4(in.p[2] + i), vd = gload4(in.p[3] + i);
        a[4 * g] = va.x; a[4 * g + 1] = va.y; a[4 * g + 2] = va.z; a[4 * g + 3] = va.w;
        b[4 * g] = vb.x; b[4 * g + 1] = vb.y; b[4 * g + 2] = vb.z; b[4 * g + 3] = vb.w;
        c[4 * g] = vc.x; c[4 * g + 1] = vc.y; c[4 * g + 2] = vc.z; c[4 * g + 3] = vc.w;
        d[4 * g] = vd.x; d[4 * g + 1] = vd.y; d[4 * g + 2] = vd.z; d[4 * g + 3] = vd.w;
    }
    bool zero = false;
#pragma unroll
    for (int e = 0; e < 8; e++)
        if ((a[e] | b[e] | c[e] | d[e]) == 0) { zero = true; a[e] = 1u; }
    if (zero) raise_flag(flag);
    // operands: dmc2 = 2 (d - c mod P), d2 = 2d, b2 = 2b, nb = P - b, nc = P - c, nd2 = 2 (P - d)
    u32 dmc[8], dmc2[8], d2[8], b2[8], nb[8], nc[8], nd2[8];
    f8::sub(dmc, d, c);
    f8::done();
#pragma unroll
    for (int e = 0; e < 8; e++) {
        dmc2[e] = dmc[e] + dmc[e]; d2[e] = d[e] + d[e]; b2[e] = b[e] + b[e];
        nb[e] = P - b[e]; nc[e] = P - c[e]; nd2[e] = (P - d[e]) + (P - d[e]);
    }
    u64 re0[8], re1[8], im0[8], im1[8];
    f8::boundary<kPrioHeavy>(a, b, c, d, dmc2, d2, b2, nb, nc, nd2);      // every operand of the multiply-add run crosses its boundary
    f8::mul64(re0, a, a); f8::mad(re0, b, nb); f8::mad(re0, c, dmc2);
    f8::mul64(re1, d, d2);
    f8::mul64(im0, a, b2); f8::mad(im0, c, nd2);
    f8::mul64(im1, c, nd2); f8::mad(im1, c, nc); f8::mad(im1, d, d);
    f8::pin(re1); f8::pin(im0); f8::pin(im1);
    u32 r0[8], r1[8], i0[8], i1[8], dr[8], di[8];
    f8::reduce(r0, re0); f8::reduce<false>(r1, re1); f8::reduce(i0, im0); f8::reduce(i1, im1);
    f8::add(dr, r0, r1);
    f8::add(di, i0, i1);
    // norms and their inverses
    u64 nn[8];
    u32 n[8], ninv[8];
    f8::boundary<kPrioHeavy>(dr, di);
    f8::mul64(nn, dr, dr); f8::mad(nn, di, di);
    f8::reduce<false>(n, nn);
    f8::inverse8(ninv, n);
    // (ir, ii) = (dr, P - di) / n, then the four output coordinates
    u32 ir[8], ii[8], ndi[8], nir[8], nii[8];
    f8::neg_operand(ndi, di);
    f8::mul(ir, dr, ninv);
    f8::mul(ii, ndi, ninv);
    f8::done();
    f8::neg_operand(nir, ir);
    f8::neg_operand(nii, ii);
    u64 oa[8], ob[8], oc[8], od[8];
    f8::boundary<kPrioHeavy>(a, b, c, d, ir, ii, nir, nii);
    f8::mul64(oa, a, ir); f8::mad(oa, b, nii);
    f8::mul64(ob, a, ii); f8::mad(ob, b, ir);
    f8::mul64(oc, c, nir); f8::mad(oc, d, ii);
    f8::mul64(od, c, nii); f8::mad(od, d, nir);
    f8::pin(ob); f8::pin(oc); f8::pin(od);
    u32 xa[8], xb[8], xc[8], xd[8];
    f8::reduce<false>(xa, oa); f8::reduce<false>(xb, ob); f8::reduce<false>(xc, oc); f8::reduce<false>(xd, od);
    f8::pin(xa); f8::pin(xb); f8::pin(xc);
    phase<kPrioLight>(xd);
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const size_t i = 4 * (t + (size_t)g * T);
        gstore4(out.p[0] + i, make_uint4(xa[4 * g], xa[4 * g + 1], xa[4 * g + 2], xa[4 * g + 3]));
        gstore4(out.p[1] + i, make_uint4(xb[4 * g], xb[4 * g + 1], xb[4 * g + 2], xb[4 * g + 3]));
        gstore4(out.p[2] + i, make_uint4(xc[4 * g], xc[4 * g + 1], xc[4 * g + 2], xc[4 * g + 3]));
        gstore4(out.p[3] + i, make_uint4(xd[4 * g], xd[4 * g + 1], xd[4 * g + 2], xd[4 * g + 3]));
    }
}

template <int K>
__global__ void __launch_bounds__(256) k_qm31_batch_inverse(CSoa4 in, Soa4 out, size_t n, size_t T, u32 *flag) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    qm31 x[K], pre[K];
    bool zero = false;
#pragma unroll
    for (int j = 0; j < K; j++) {
        size_t i = t + (size_t)j * T;
        qm31 v = {1u, 0u, 0u, 0u};
        if (i < n) v = {in.p[0][i], in.p[1][i], in.p[2][i], in.p[3][i]};
        if (qm31_is_zero(v)) { zero = true; v = {1u, 0u, 0u, 0u}; }
        x[j] = v;
        pre[j] = j == 0 ? v : qm31_mul(pre[j - 1], v);
    }
    if (zero) raise_flag(flag);
    qm31 cur = qm31_inv(pre[K - 1]);
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        size_t i = t + (size_t)j * T;
        qm31 r = j == 0 ? cur : qm31_mul(pre[j - 1], cur);
        cur = qm31_mul(cur, x[j]);
        if (i < n) { out.p[0][i] = r.a; out.p[1][i] = r.b; out.p[2][i] = r.c; out.p[3][i] = r.d; }
    }
}

int finish_inverse() {
    u32 flag = 0;
    int rc = read_and_clear_flag(&flag);
    if (rc) return rc;
    if (flag) return set_error(TSTWO_ERR_ZERO_INVERSE, "0 has no inverse");
    return TSTWO_OK;
}

// ---------------------------------------------------------------- bit reverse (backend/cpu/index.ts:62-79)
// In-place swap of i <-> bitrev(i) for i < bitrev(i); one column per blockIdx.y.
__global__ void __launch_bounds__(256) k_bit_reverse(ColPtrs cols, u32 log_n) {
    u32 *v = colp_u(cols, blockIdx.y);
    size_t n = (size_t)1 << log_n;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u32 j = __brev((u32)i) >> (32 - log_n);
        if (j > i) {
            u32 a = gload1(v + i), b = gload1(v + j);
            gstore1(v + i, b);
            gstore1(v + j, a);
        }
    }
}

// Tiled variant for log_n >= 12: index i = (a | m | b) with 6-bit a (top) and b (bottom) maps to
// (rev b | rev m | rev a), so the 64x64-word tile with middle bits m trades places (transposed and bit-reversed
// within) with tile rev(m).  Both tiles are staged in LDS; every global access is a 16-byte lane access on
// 256-byte runs, instead of the scattered 4-byte swaps of k_bit_reverse.
__global__ void __launch_bounds__(256) k_bit_reverse_tiled(ColPtrs cols, u32 log_n) {
    constexpr int T = 6, S = 1 << T, STRIDE = S + 1;
    __shared__ u32 lds[2][S * STRIDE];
    u32 *__restrict__ v = colp_u(cols, blockIdx.y);
    const u32 mid_bits = log_n - 2 * T;
    const u32 m = blockIdx.x;
    const u32 rm = mid_bits ? (__brev(m) >> (32 - mid_bits)) : 0u;
    if (rm < m) return;                                  // the pair is handled by the block of the smaller index
    const u32 row_shift = log_n - T;
    for (int which = 0; which < (rm == m ? 1 : 2); which++) {
        const u32 mm = which ? rm : m;
#pragma unroll
        for (int it = 0; it < S * S / (256 * 4); it++) {
            const u32 idx4 = threadIdx.x + it * 256;
            const u32 a = idx4 / (S / 4), b4 = (idx4 % (S / 4)) * 4;
            const uint4 x = gload4(v + (((size_t)a << row_shift) | ((size_t)mm << T) | b4));
            u32 *p = &lds[which][a * STRIDE + b4];
            p[0] = x.x; p[1] = x.y; p[2] = x.z; p[3] = x.w;
        }
    }
    __syncthreads();
    for (int which = 0; which < (rm == m ? 1 : 2); which++) {
        const u32 dst = which ? m : rm;                  // tile m's words land in tile rev(m) and vice versa
        const u32 *L = lds[which];
#pragma unroll
        for (int it = 0; it < S * S / (256 * 4); it++) {
            const u32 idx4 = threadIdx.x + it * 256;
            const u32 r = idx4 / (S / 4), c4 = (idx4 % (S / 4)) * 4;
            const u32 rr = __brev(r) >> (32 - T);
            uint4 x;
            x.x = L[(__brev(c4 + 0) >> (32 - T)) * STRIDE + rr];
            x.y = L[(__brev(c4 + 1) >> (32 - T)) * STRIDE + rr];
            x.z = L[(__brev(c4 + 2) >> (32 - T)) * STRIDE + rr];
            x.w = L[(__brev(c4 + 3) >> (32 - T)) * STRIDE + rr];
            gstore4(v + (((size_t)r << row_shift) | ((size_t)dst << T) | c4), x);
        }
    }
}

// ---------------------------------------------------------------- twiddle tree (backend/cpu/circle.ts:210-221)
// Entry e of the tree of coset (init, m): level lvl holds the x coordinates of the first half of
// coset.repeated_double(lvl), bit-reversed; the last entry is 1.
__global__ void __launch_bounds__(256) k_twiddles(u32 init, u32 m, u32 *__restrict__ tw, const cpoint *__restrict__ gen_pow2) {
    size_t total = (size_t)1 << m;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    if (e == total - 1) { tw[e] = 1u; return; }
    u32 r = (u32)(total - e);                    // 2 <= r <= 2^m
    u32 lg = 32 - __clz(r - 1);                  // ceil(log2 r) = log size of this level's coset
    u32 lvl = m - lg;
    u32 off = (u32)(total - ((size_t)1 << lg));
    u32 j = (u32)e - off;                        // < 2^(lg-1)
    u32 k = lg > 1 ? (__brev(j) >> (32 - (lg - 1))) : 0u;
    u32 init_l = (init << lvl) & 0x7fffffffu;    // Coset.double(): initial*2, circle.ts:253-256
    u32 idx = (init_l + (k << (31 - lg))) & 0x7fffffffu;
    tw[e] = cpoint_from_index_win(idx, gen_pow2).x;      // gen_pow2 = Context::gen_win
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_extend(const u32 *__restrict__ src, size_t n_src, u32 *__restrict__ dst, size_t n_dst) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {              // 16-byte accesses: both lengths multiples of 4, both buffers 16-byte aligned
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst / 4; i += stride)
            gstore4(dst + 4 * i, 4 * i < n_src ? gload4(src + 4 * i) : make_uint4(0u, 0u, 0u, 0u));
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += stride) dst[i] = i < n_src ? src[i] : 0u;
    }
}

unsigned capped_blocks(size_t work_items, unsigned threads) {
    unsigned blocks = ceil_div(work_items, threads);
    unsigned cap = (unsigned)ctx().n_cus * 64;        // (fri.hip: streaming kernels of this shape are 5-6 % faster at 32+ workgroups per CU than at 8)
    if (blocks > cap) blocks = cap;
    return blocks ? blocks : 1;
}

}  // namespace

extern "C" {

int tstwo_m31_add(const u32 *a, const u32 *b, u32 *out, size_t n) { return launch_binop<OP_ADD>(a, b, out, n); }
int tstwo_m31_sub(const u32 *a, const u32 *b, u32 *out, size_t n) { return launch_binop<OP_SUB>(a, b, out, n); }
int tstwo_m31_mul(const u32 *a, const u32 *b, u32 *out, size_t n) { return launch_binop<OP_MUL>(a, b, out, n); }
int tstwo_m31_neg(const u32 *a, u32 *out, size_t n) { return launch_binop<OP_NEG>(a, nullptr, out, n); }

int tstwo_check_zero_flag(void) {
    TSTWO_REQUIRE_READY();
    return finish_inverse();
}

int tstwo_m31_batch_inverse_async(const u32 *in, u32 *out, size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    TSTWO_REQUIRE_PTRS(in, out);
    // elements per lane: enough lanes to fill the chip first, then amortise the 37-multiplication Fermat chain
    if (n >= ((size_t)1 << 24)) {
        constexpr int K = 16;
        size_t T = (n + K - 1) / K;
        hipLaunchKernelGGL(k_m31_batch_inverse<K>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, in, out, n, T, ctx().flag);
    } else {
        constexpr int K = 4;
        size_t T = (n + K - 1) / K;
        hipLaunchKernelGGL(k_m31_batch_inverse<K>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, in, out, n, T, ctx().flag);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
int tstwo_m31_batch_inverse(const u32 *in, u32 *out, size_t n) {
    int rc = tstwo_m31_batch_inverse_async(in, out, n);
    return rc || n == 0 ? rc : finish_inverse();
}
int tstwo_cm31_batch_inverse_async(const u32 *const in[2], u32 *const out[2], size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    TSTWO_REQUIRE_TABLE(in, 2); TSTWO_REQUIRE_TABLE(out, 2);
    constexpr int K = 8;
    size_t T = (n + K - 1) / K;
    CSoa2 i2 = {{in[0], in[1]}};
    Soa2 o2 = {{out[0], out[1]}};
    hipLaunchKernelGGL(k_cm31_batch_inverse<K>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i2, o2, n, T, ctx().flag);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
int tstwo_cm31_batch_inverse(const u32 *const in[2], u32 *const out[2], size_t n) {
    int rc = tstwo_cm31_batch_inverse_async(in, out, n);
    return rc || n == 0 ? rc : finish_inverse();
}
int tstwo_qm31_batch_inverse_async(const u32 *const in[4], u32 *const out[4], size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4);
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    bool aligned = true;
    for (int k = 0; k < 4; k++) aligned = aligned && ((((uintptr_t)in[k]) | ((uintptr_t)out[k])) & 15) == 0;
    const int kq = knobs().qinv_k;      // experiments: 4-byte form, K per lane
    const bool mont = knobs().qinv_montgomery;      // A/B timing: Montgomery's trick over QM31 values
    if (kq == 0 && aligned && n % 8 == 0 && n >= 8 && !mont) {  // 16-byte accesses, 8 elements per lane, one M31 inversion per 8
        size_t T = n / 8;
        hipLaunchKernelGGL(k_qm31_batch_inverse_norm, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i4, o4, T, ctx().flag);
    } else if (kq == 0 && aligned && n % 8 == 0 && n >= 8) {
        size_t T = n / 8;
        hipLaunchKernelGGL(k_qm31_batch_inverse_v4<2>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i4, o4, T, ctx().flag);
    } else if (kq == 0 && aligned && n % 4 == 0 && n >= 4) {
        size_t T = n / 4;
        hipLaunchKernelGGL(k_qm31_batch_inverse_v4<1>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i4, o4, T, ctx().flag);
    } else if (kq == 4) {
        size_t T = (n + 3) / 4;
        hipLaunchKernelGGL(k_qm31_batch_inverse<4>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i4, o4, n, T, ctx().flag);
    } else {
        size_t T = (n + 7) / 8;
        hipLaunchKernelGGL(k_qm31_batch_inverse<8>, dim3(ceil_div(T, 256)), dim3(256), 0, ctx().stream, i4, o4, n, T, ctx().flag);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
int tstwo_qm31_batch_inverse(const u32 *const in[4], u32 *const out[4], size_t n) {
    int rc = tstwo_qm31_batch_inverse_async(in, out, n);
    return rc || n == 0 ? rc : finish_inverse();
}

int tstwo_qm31_mul(const u32 *const a[4], const u32 *const b[4], u32 *const out[4], size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    TSTWO_REQUIRE_TABLE(a, 4); TSTWO_REQUIRE_TABLE(b, 4); TSTWO_REQUIRE_TABLE(out, 4);
    CSoa4 a4 = {{a[0], a[1], a[2], a[3]}}, b4 = {{b[0], b[1], b[2], b[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    hipLaunchKernelGGL(k_qm31_mul, dim3(capped_blocks(n, 256)), dim3(256), 0, ctx().stream, a4, b4, o4, n);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_secure_accumulate(u32 *const col[4], const u32 *const other[4], size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return TSTWO_OK;
    TSTWO_REQUIRE_TABLE(col, 4); TSTWO_REQUIRE_TABLE(other, 4);
    Soa4 c4 = {{col[0], col[1], col[2], col[3]}};
    CSoa4 o4 = {{other[0], other[1], other[2], other[3]}};
    bool vec = n % 4 == 0;
    for (int k = 0; k < 4; k++) vec = vec && ((((uintptr_t)col[k]) | ((uintptr_t)other[k])) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_secure_accumulate<true>, dim3(capped_blocks(n / 4, 256), 4), dim3(256), 0, ctx().stream, c4, o4, n);
    else hipLaunchKernelGGL(k_secure_accumulate<false>, dim3(capped_blocks(n, 256), 4), dim3(256), 0, ctx().stream, c4, o4, n);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_bit_reverse(u32 *const *cols, size_t n_cols, size_t n) {
    TSTWO_REQUIRE_READY();
    if (n == 0 || (n & (n - 1)) != 0) return set_error(TSTWO_ERR_NOT_POW2, "length is not power of two");
    u32 log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    if (log_n == 0 || n_cols == 0) return TSTWO_OK;
    TSTWO_REQUIRE_TABLE(cols, n_cols);
    const size_t kChunk = 32768;                       // gridDim.y limit; one launch for up to 32768 columns
    for (size_t base = 0; base < n_cols; base += kChunk) {
        const size_t cnt = n_cols - base < kChunk ? n_cols - base : kChunk;
        ColPtrs cp;
        int rc_tab = fill_col_table(cp, cols + base, cnt, 0);
        if (rc_tab) return rc_tab;
        bool aligned = true;
        for (size_t i = 0; i < cnt; i++) aligned = aligned && ((((uintptr_t)cols[base + i]) & 15) == 0);
        if (log_n >= 12 && log_n <= 40 && aligned) {
            hipLaunchKernelGGL(k_bit_reverse_tiled, dim3(1u << (log_n - 12), (unsigned)cnt), dim3(256), 0, ctx().stream, cp, log_n);
        } else {
            unsigned blocks = capped_blocks(n, 256);
            hipLaunchKernelGGL(k_bit_reverse, dim3(blocks, (unsigned)cnt), dim3(256), 0, ctx().stream, cp, log_n);
        }
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_twiddles_build(u32 coset_initial, u32 log_size, u32 *tw, u32 *itw) {
    TSTWO_REQUIRE_READY();
    if (log_size > 30 || !tw) return set_error(TSTWO_ERR_BAD_ARG, "tstwo_twiddles_build: bad log_size or null buffer");
    size_t n = (size_t)1 << log_size;
    hipLaunchKernelGGL(k_twiddles, dim3(ceil_div(n, 256)), dim3(256), 0, ctx().stream, coset_initial & 0x7fffffffu,
                       log_size, tw, ctx().gen_win);
    TSTWO_LAUNCH_CHECK();
    if (itw) return tstwo_m31_batch_inverse(tw, itw, n);   // backend/cpu/circle.ts:223-239 ("0 has no inverse" for cosets through x = 0)
    return TSTWO_OK;
}

int tstwo_poly_extend(const u32 *src, u32 log_src, u32 *dst, u32 log_dst) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_PTRS(src, dst);
    if (log_dst > 31) return set_error(TSTWO_ERR_BAD_ARG, "extend: log size out of range");
    if (log_dst < log_src) return set_error(TSTWO_ERR_LOG_SIZE, "log size too small");
    size_t ns = (size_t)1 << log_src, nd = (size_t)1 << log_dst;
    if (log_src >= 2 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0)
        hipLaunchKernelGGL(k_extend<true>, dim3(capped_blocks(nd / 4, 256)), dim3(256), 0, ctx().stream, src, ns, dst, nd);
    else
        hipLaunchKernelGGL(k_extend<false>, dim3(capped_blocks(nd, 256)), dim3(256), 0, ctx().stream, src, ns, dst, nd);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

}  // extern "C"
