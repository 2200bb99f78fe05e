// fri.hip — FriOps (fold_line, fold_circle_into_line, decompose) and PolyOps.eval_at_point.
//
// A FRI fold is one inverse-CFFT layer followed by f0 + alpha*f1: the per-output
// twiddle domain.at(bitrev(2i)).x^-1 (resp. .y^-1) that the reference recomputes per element with a
// scalar multiplication and a Fermat inverse (fri.ts:138-141,180-183) is a slice of the inverse
// twiddle tree.  One lane per output row on SoA QM31 (4 coalesced 8-byte loads, 4 coalesced stores).
// Algorithmic bytes per output row: fold_line 48 (32 in + 16 out), fold_circle_into_line 64
// (32 src + 16 dst in + 16 dst out).
#include <string.h>
#include <vector>

#include "common.h"
#include "host_field.h"

using namespace tstwo;

namespace {

__device__ __forceinline__ qm31 load_pair_fold(const CSoa4 &in, size_t i, u32 t, qm31 *f0_out) {
    // (f0, f1) = ibutterfly(in[2i], in[2i+1], t) per coordinate (fft.ts:25-30)
    uint2 a = gload2(in.p[0] + 2 * i);
    uint2 b = gload2(in.p[1] + 2 * i);
    uint2 c = gload2(in.p[2] + 2 * i);
    uint2 d = gload2(in.p[3] + 2 * i);
    *f0_out = {m31_add(a.x, a.y), m31_add(b.x, b.y), m31_add(c.x, c.y), m31_add(d.x, d.y)};
    qm31 diff = {m31_sub(a.x, a.y), m31_sub(b.x, b.y), m31_sub(c.x, c.y), m31_sub(d.x, d.y)};
    return qm31_mul_m31(diff, t);
}

// fri.ts:120-152.  inv_x[i] = domain.at(bitrev(2i)).x^-1.
__global__ void __launch_bounds__(256) k_fold_line(CSoa4 in, Soa4 out, size_t n_out, const u32 *__restrict__ inv_x, qm31 alpha,
                                                  const qm31 *__restrict__ alpha_dev) {
    if (alpha_dev) alpha = *alpha_dev;          // alpha drawn by the device channel (uniform load)
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        qm31 f0;
        qm31 f1 = load_pair_fold(in, i, inv_x[i], &f0);
        qm31 r = qm31_add(f0, qm31_mul(alpha, f1));
        out.p[0][i] = r.a; out.p[1][i] = r.b; out.p[2][i] = r.c; out.p[3][i] = r.d;
    }
}

// fri.ts:162-192.  Twiddle = circle-layer inverse twiddle: either explicit inv_y[i], or derived from the
// layer-1 slice of the inverse tree: +-seg1[(i>>1)^1], negative iff (i ^ (i>>1)) & 1.
// ACCUM = false: dst is written, not updated (dst = alpha f1 + f0) — the first fold of a FRI commit, whose line evaluation starts
// at zero (fri.ts:687-693): no zero fill of dst, no read of it.
template <bool FROM_TREE, bool ACCUM = true>
__global__ void __launch_bounds__(256) k_fold_circle(Soa4 dst, CSoa4 src, size_t n_out, const u32 *__restrict__ twp,
                                                    qm31 alpha, qm31 alpha_sq, const qm31 *__restrict__ alpha_dev) {
    if (alpha_dev) { alpha = *alpha_dev; if (ACCUM) alpha_sq = qm31_mul(alpha, alpha); }
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        u32 t;
        if (FROM_TREE) {
            t = twp[(i >> 1) ^ 1];
            if ((i ^ (i >> 1)) & 1) t = m31_neg(t);
        } else {
            t = twp[i];
        }
        qm31 f0;
        qm31 f1 = load_pair_fold(src, i, t, &f0);
        qm31 fp = qm31_add(qm31_mul(alpha, f1), f0);
        qm31 r = fp;
        if (ACCUM) {
            qm31 cur = {dst.p[0][i], dst.p[1][i], dst.p[2][i], dst.p[3][i]};
            r = qm31_add(qm31_mul(cur, alpha_sq), fp);
        }
        dst.p[0][i] = r.a; dst.p[1][i] = r.b; dst.p[2][i] = r.c; dst.p[3][i] = r.d;
    }
}

// Two consecutive output rows per lane: 16-byte loads of the source, 8-byte accesses of dst (log 24: 94.2 against 96.7 us for the
// one-row form; needs 16-byte aligned source columns and 8-byte aligned dst columns, tree twiddles).  TSTWO_FOLD1 = the one-row form.
template <bool ACCUM>
__global__ void __launch_bounds__(256) k_fold_circle2(Soa4 dst, CSoa4 src, size_t n_out, const u32 *__restrict__ twp,
                                                     qm31 alpha, qm31 alpha_sq, const qm31 *__restrict__ alpha_dev) {
    if (alpha_dev) { alpha = *alpha_dev; if (ACCUM) alpha_sq = qm31_mul(alpha, alpha); }
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out / 2; j += stride) {
        // rows i = 2j, 2j + 1 share the twiddle twp[j ^ 1] up to sign: negative iff (i ^ (i >> 1)) & 1
        const u32 tw = twp[j ^ 1];
        const u32 t0 = (j & 1) ? m31_neg(tw) : tw, t1 = (j & 1) ? tw : m31_neg(tw);
        const uint4 a = gload4(src.p[0] + 4 * j), b = gload4(src.p[1] + 4 * j), c = gload4(src.p[2] + 4 * j), d = gload4(src.p[3] + 4 * j);
        uint2 cur[4];
        if (ACCUM) {
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = gload2(dst.p[k] + 2 * j);
        }
        qm31 r[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const u32 ax = h ? a.z : a.x, ay = h ? a.w : a.y, bx = h ? b.z : b.x, by = h ? b.w : b.y;
            const u32 cx = h ? c.z : c.x, cy = h ? c.w : c.y, dx = h ? d.z : d.x, dy = h ? d.w : d.y;
            const qm31 f0 = {m31_add(ax, ay), m31_add(bx, by), m31_add(cx, cy), m31_add(dx, dy)};
            const qm31 f1 = qm31_mul_m31({m31_sub(ax, ay), m31_sub(bx, by), m31_sub(cx, cy), m31_sub(dx, dy)}, h ? t1 : t0);
            r[h] = qm31_add(qm31_mul(alpha, f1), f0);
            if (ACCUM) {
                const qm31 cu = {h ? cur[0].y : cur[0].x, h ? cur[1].y : cur[1].x, h ? cur[2].y : cur[2].x, h ? cur[3].y : cur[3].x};
                r[h] = qm31_add(qm31_mul(cu, alpha_sq), r[h]);
            }
        }
        *(TSTWO_GLOBAL u32x2_t *)(dst.p[0] + 2 * j) = u32x2_t{r[0].a, r[1].a};
        *(TSTWO_GLOBAL u32x2_t *)(dst.p[1] + 2 * j) = u32x2_t{r[0].b, r[1].b};
        *(TSTWO_GLOBAL u32x2_t *)(dst.p[2] + 2 * j) = u32x2_t{r[0].c, r[1].c};
        *(TSTWO_GLOBAL u32x2_t *)(dst.p[3] + 2 * j) = u32x2_t{r[0].d, r[1].d};
    }
}

// (A form with a lane owning 4 consecutive output rows — every access 16 bytes per lane instead of 8 / 4 — was built and
// measured in round 3: fold_circle_into_line log 24 103.9 against 106.7 us, fold_line log 23 37.9 against 35.3 us: the
// folds already move 5.0 - 5.9 TB/s and the 120 VGPRs of the 4-row form cost as much occupancy as the wider accesses
// gain.  Removed; gpurun_out/r03b/f1.log.)
// LineEvaluation.interpolate (poly/line.ts:312-329) with lineIfft (line.ts:354-390) for a layer that fits one workgroup's LDS — the
// last FRI layer (2^7 rows by default): bit reversal on the way in, log_n levels of ibutterflies with x^-1 =
// domain.at(i)^-1 taken from the inverse twiddle tree (level of coset size 2^k: itw_end - 2^k + bitrev(i, k - 1)), the 1/n
// scaling, coefficients out in the reference's bit-reversed order.  One workgroup, coordinate c of a value handled like a
// column (every twiddle is in the base field).
__global__ void __launch_bounds__(256) k_line_interpolate(CSoa4 in, Soa4 out, u32 log_n, const u32 *__restrict__ itw_end, u32 n_inv) {
    extern __shared__ u32 lsh[];                 // [4][n]
    const u32 n = 1u << log_n, t = threadIdx.x;
    for (u32 i = t; i < 4 * n; i += 256) {
        const u32 c = i >> log_n, j = i & (n - 1);
        const u32 nat = log_n ? __brev(j) >> (32 - log_n) : 0u;
        lsh[(c << log_n) + nat] = gload1(in.p[c] + j);
    }
    __syncthreads();
    for (u32 k = log_n; k >= 1; k--) {                              // chunks of 2^k values
        const u32 half = 1u << (k - 1);
        for (u32 w = t; w < 2 * n; w += 256) {                      // 4 coordinates x n/2 butterflies
            const u32 c = w >> (log_n - 1), b = w & ((n >> 1) - 1);
            const u32 i = b & (half - 1), chunk = b >> (k - 1);
            const u32 l = (c << log_n) + (chunk << k) + i, r = l + half;
            const u32 br = k > 1 ? __brev(i) >> (32 - (k - 1)) : 0u;
            const u32 x_inv = itw_end[(int)br - (int)(1u << k)];
            const u32 a = lsh[l], bb = lsh[r];
            lsh[l] = m31_add(a, bb);                                  // ibutterfly (fft.ts:25-30)
            lsh[r] = m31_mul(m31_sub(a, bb), x_inv);
        }
        __syncthreads();
    }
    for (u32 i = t; i < 4 * n; i += 256) gstore1(out.p[i >> log_n] + (i & (n - 1)), m31_mul(lsh[i], n_inv));
}

// backend/cpu/fri.ts:97-123: sums of the two halves of each coordinate column (exact in u64).
// VEC: 16-byte loads (columns 16-byte aligned, halves a multiple of 4 words).
template <bool VEC>
__global__ void __launch_bounds__(256) k_half_sums(CSoa4 in, size_t n, unsigned long long *sums /* [4][2] */) {
    __shared__ unsigned long long sh[256 / 64];
    const u32 coord = blockIdx.y, halfsel = blockIdx.z;
    const size_t half = n / 2;
    const u32 *p = in.p[coord] + (halfsel ? half : 0);
    const size_t cnt = halfsel ? n - half : half;
    unsigned long long acc = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt / 4; i += stride) {
            const uint4 v = gload4(p + 4 * i);
            acc += (unsigned long long)v.x + v.y + ((unsigned long long)v.z + v.w);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) acc += p[i];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = sh[0] + sh[1] + sh[2] + sh[3];
        atomicAdd(&sums[coord * 2 + halfsel], tot);
    }
}
// backend/cpu/fri.ts:133-164: g = f - lambda on the first half, f + lambda on the second (n == 1: f - lambda)
template <bool VEC>
__global__ void __launch_bounds__(256) k_decompose_apply(CSoa4 in, Soa4 out, size_t n, qm31 lambda) {
    const u32 coord = blockIdx.y;
    const u32 lam = coord == 0 ? lambda.a : coord == 1 ? lambda.b : coord == 2 ? lambda.c : lambda.d;
    const size_t half = n / 2;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {              // n >= 8: the 4 words of a vector lie in one half
        const u32 nlam = m31_neg(lam);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
            const uint4 v = gload4(in.p[coord] + 4 * i);
            const u32 add = 4 * i < half ? nlam : lam;          // f - lambda = f + (-lambda)
            gstore4(out.p[coord] + 4 * i, make_uint4(m31_add(v.x, add), m31_add(v.y, add), m31_add(v.z, add), m31_add(v.w, add)));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            u32 v = in.p[coord][i];
            out.p[coord][i] = (i < half || n == 1) ? m31_sub(v, lam) : m31_add(v, lam);
        }
    }
}

// ---------------------------------------------------------------- eval_at_point
// PolyOps.eval_at_point (backend/cpu/circle.ts:52-69) = fold(coeffs, [y, x, pi(x), pi^2(x), ...] reversed)
// (poly/utils.ts:36-59).  Unrolled, the fold is the multilinear form
//     value = sum_i coeffs[i] * prod_{s : bit s of i is set} fac[s],      fac = [y, x, pi(x), pi^2(x), ...],
// and the product splits over any partition of the index bits.  One pass over the coefficients (kernel E1), then one tiny
// kernel per further 12 index bits (E2):
//   E1  a workgroup of 256 lanes owns 4096 consecutive coefficients.  Lane t reads four 16-byte vectors, a KiB apart per
//       wave (fully coalesced): coefficient (r, t, j) = base + 1024 r + 4 t + j.  Bits {0,1} (j) and {10,11} (r) are the same
//       for every lane, so their 16 factor products W[r][j] arrive as kernel arguments (SGPRs) and the lane's 16 terms are
//       16 x 4 v_mad_u64_u32 (M31 x QM31 = 4 multiplications, accumulated lazily in 64 bits).  Bits 2..9 (t) give a
//       per-lane factor A[t & 15] * B[t >> 4] from two 16-entry tables that 32 lanes build in LDS while the loads are in
//       flight.  After that the workgroup's partial is a plain sum over lanes (DPP-free shuffles + one LDS hop).
//   E2  folds up to 4096 QM31 partials per workgroup the same way (QM31 x QM31 terms), bits 8..11 through a 16-entry
//       argument table, bits 0..7 through the lane factor.
// log 22: 1024 workgroups + one; the result is read back with one 16-byte copy.  Algorithmic bytes: 4 per coefficient.
struct EvalW { qm31 w[16]; };        // products over the 4 "uniform" bits of a level (entry 0 = 1)
struct EvalF { qm31 f[8]; };         // factors of the 8 lane bits of a level (A: f[0..3], B: f[4..7])

__device__ __forceinline__ u32 red64(u64 x) {          // x < 2^64: canonical x mod P
    u64 f = (x & M31_P) + (x >> 31);                     // < 2^31 + 2^33
    return m31_reduce64(f);
}
// lane factor tables: entry e < 16 = prod_{i<4, bit i of e} f[i]; entry 16 + e = the same over f[4..7]
__device__ __forceinline__ void eval_build_tables(qm31 *tab, const EvalF &ff) {
    const u32 t = threadIdx.x;
    if (t < 32) {
        const u32 e = t & 15;
        const int o = t < 16 ? 0 : 4;
        qm31 v = {1u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const qm31 f = (o == 0) ? ff.f[i] : ff.f[4 + i];
            const qm31 p = qm31_mul(v, f);
            if ((e >> i) & 1) v = p;
        }
        tab[t] = v;
    }
}
// sum of one QM31 per lane over the workgroup (256 lanes); the result is valid in lane 0
__device__ __forceinline__ qm31 eval_wg_sum(qm31 v, qm31 *scratch /* >= 4 entries */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        qm31 o = {(u32)__shfl_down((int)v.a, off, 64), (u32)__shfl_down((int)v.b, off, 64), (u32)__shfl_down((int)v.c, off, 64),
                  (u32)__shfl_down((int)v.d, off, 64)};
        v = qm31_add(v, o);
    }
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) v = qm31_add(qm31_add(scratch[0], scratch[1]), qm31_add(scratch[2], scratch[3]));
    return v;
}

// Fold of up to 4096 QM31 partials by one workgroup (R = 16 per lane; fewer when `left` < 4096): entry i lives at
// in[i * elem_stride].  The result is valid in lane 0.
__device__ __forceinline__ qm31 eval_fold_partials(const qm31 *__restrict__ in, size_t elem_stride, size_t left, const EvalW &W, const EvalF &F,
                                                   qm31 *tab) {
    const u32 t = threadIdx.x;
    eval_build_tables(tab, F);
    __syncthreads();
    const qm31 ft = qm31_mul(tab[t & 15], tab[16 + (t >> 4)]);
    qm31 acc = {0u, 0u, 0u, 0u};
#pragma unroll 4
    for (int r = 0; r < 16; r++) {
        const size_t i = (size_t)r * 256 + t;
        if (i < left) acc = qm31_add(acc, qm31_mul(in[i * elem_stride], W.w[r]));
    }
    const qm31 v = qm31_mul(acc, ft);
    return eval_wg_sum(v, tab + 32);
}

// E1: coefficients -> one partial per chunk of 4096 * G coefficients.  grid = (chunks, columns).  G = 4 (64 coefficients
// per lane: the per-lane fixed work — lane factor, workgroup sum — is paid once per 64 instead of once per 16) when that still
// leaves >= 2 workgroups per CU, else G = 1.  Group g of a chunk (index bits 12, 13) is folded with the uniform factors H[g].
// n_coeffs < 4096 or unaligned columns take the guarded scalar loads (coefficients beyond the polynomial count as zero).
struct EvalH { qm31 h[4]; };         // h[g] = prod_{bit of g} fac[12 + bit]  (h[0] = 1)
// (A one-launch variant — the workgroup that arrives last on an agent-scope counter folds the partials itself, hand-off by the
// CDNA guide's release / acquire recipe — was built and measured: 28.7 us instead of 25.5 for one column of 2^22, 93 instead
// of 50 for 32 columns of 2^20: a release fence (L2 write-back) in every one of the 256 .. 2048 workgroups costs more than
// the second launch it saves.  Removed.)
template <bool FAST, int G>
__global__ void __launch_bounds__(256) k_eval_coeffs(ColPtrs cols, size_t n_coeffs, EvalW W, EvalF F, EvalH H, qm31 *__restrict__ partial_out,
                                                    size_t out_stride) {
    __shared__ qm31 tab[32 + 4];
    const u32 t = threadIdx.x;
    const u32 *__restrict__ c = colp_u(cols, blockIdx.y);
    const size_t base = (size_t)blockIdx.x * (4096 * G) + 4 * t;
    auto load_group = [&](uint4 (&x)[4], int g) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t i = base + 4096 * (size_t)g + 1024 * (size_t)r;
            if (FAST) {
                x[r] = gload4(c + i);
            } else {
                x[r].x = i + 0 < n_coeffs ? gload1(c + i + 0) : 0u; x[r].y = i + 1 < n_coeffs ? gload1(c + i + 1) : 0u;
                x[r].z = i + 2 < n_coeffs ? gload1(c + i + 2) : 0u; x[r].w = i + 3 < n_coeffs ? gload1(c + i + 3) : 0u;
            }
        }
    };
    uint4 x[4];
    load_group(x, 0);
    eval_build_tables(tab, F);               // overlaps the loads above
    __syncthreads();
    const qm31 ft = qm31_mul(tab[t & 15], tab[16 + (t >> 4)]);
    qm31 total = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < G; g++) {
        uint4 y[4];
        if (g + 1 < G) load_group(y, g + 1);       // next group's loads fly while this one is multiplied
        u32 ua = 0, ub = 0, uc = 0, ud = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u32 v[4] = {x[r].x, x[r].y, x[r].z, x[r].w};
            u64 a = ua, b = ub, cc = uc, d = ud;     // 4 products < 2^62 each + carry-in < 2^31: no overflow
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const qm31 w = W.w[4 * r + j];
                a += (u64)v[j] * w.a; b += (u64)v[j] * w.b; cc += (u64)v[j] * w.c; d += (u64)v[j] * w.d;
            }
            ua = red64(a); ub = red64(b); uc = red64(cc); ud = red64(d);
        }
        const qm31 u = {ua, ub, uc, ud};
        total = g == 0 ? u : qm31_add(total, qm31_mul(u, H.h[g]));
        if (g + 1 < G) {
#pragma unroll
            for (int r = 0; r < 4; r++) x[r] = y[r];
        }
    }
    qm31 v = qm31_mul(total, ft);
    v = eval_wg_sum(v, tab + 32);
    if (t == 0) partial_out[(size_t)blockIdx.y * out_stride + blockIdx.x] = v;
}

// E2: QM31 partials -> one partial per 4096 of them.  grid = (groups, columns).
__global__ void __launch_bounds__(256) k_eval_partials(const qm31 *__restrict__ partial_in, size_t in_stride, size_t m_in, EvalW W, EvalF F,
                                                      qm31 *__restrict__ partial_out, size_t out_stride) {
    __shared__ qm31 tab[32 + 4];
    const qm31 *__restrict__ in = partial_in + (size_t)blockIdx.y * in_stride + (size_t)blockIdx.x * 4096;
    const size_t left = m_in - (size_t)blockIdx.x * 4096;      // entries of this group (>= 1)
    const qm31 v = eval_fold_partials(in, 1, left, W, F, tab);
    if (threadIdx.x == 0) partial_out[(size_t)blockIdx.y * out_stride + blockIdx.x] = v;
}

unsigned capped_blocks(size_t work_items, unsigned threads) {
    unsigned blocks = ceil_div(work_items, threads);
    const unsigned mult = (unsigned)knobs().fold_cap;     // workgroups per CU before lanes grid-stride (8: fold_circle log 24 102 us, fold_line log 23 33.5 us; 32-1024: 97 / 32.5 us)
    unsigned cap = (unsigned)ctx().n_cus * mult;
    if (blocks > cap) blocks = cap;
    return blocks ? blocks : 1;
}
qm31 to_q(const u32 a[4]) { return {a[0], a[1], a[2], a[3]}; }
qm31 to_q(host::Q a) { return {a.v[0], a.v[1], a.v[2], a.v[3]}; }
host::Q to_hq(const u32 a[4]) { host::Q q; for (int i = 0; i < 4; i++) q.v[i] = a[i]; return q; }

static void launch_fold_line(const CSoa4 &i4, const Soa4 &o4, size_t n_out, const u32 *inv_x, qm31 alpha, const qm31 *alpha_dev) {
    // (a two-rows-per-lane form with 16-byte loads, as in k_fold_circle2, measured the same 6.2-6.4 TB/s: not kept)
    hipLaunchKernelGGL(k_fold_line, dim3(capped_blocks(n_out, 256)), dim3(256), 0, ctx().stream, i4, o4, n_out, inv_x, alpha, alpha_dev);
}
static void launch_fold_circle(bool from_tree, const Soa4 &d4, const CSoa4 &s4, size_t n_out, const u32 *twp, qm31 a, qm31 a2, const qm31 *alpha_dev) {
    const bool fold1 = knobs().fold1;
    bool two = from_tree && !fold1 && n_out >= 4;
    for (int k = 0; k < 4; k++) two = two && (((uintptr_t)s4.p[k]) & 15) == 0 && (((uintptr_t)d4.p[k]) & 7) == 0;
    if (two)
        hipLaunchKernelGGL(k_fold_circle2<true>, dim3(capped_blocks(n_out / 2, 256)), dim3(256), 0, ctx().stream, d4, s4, n_out, twp, a, a2, alpha_dev);
    else if (from_tree)
        hipLaunchKernelGGL(k_fold_circle<true>, dim3(capped_blocks(n_out, 256)), dim3(256), 0, ctx().stream, d4, s4, n_out, twp, a, a2, alpha_dev);
    else
        hipLaunchKernelGGL(k_fold_circle<false>, dim3(capped_blocks(n_out, 256)), dim3(256), 0, ctx().stream, d4, s4, n_out, twp, a, a2, alpha_dev);
}

}  // namespace

extern "C" {

int tstwo_fri_fold_line_tw(const u32 *const in[4], u32 log_n, const u32 *inv_x, const u32 alpha[4], u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4); TSTWO_REQUIRE_PTRS(inv_x, alpha);
    if (log_n == 0) return set_error(TSTWO_ERR_TOO_SMALL, "fold_line: Evaluation too small, must have at least 2 elements.");
    if (log_n > 31) return set_error(TSTWO_ERR_BAD_ARG, "fold_line: log size out of range");
    size_t n_out = (size_t)1 << (log_n - 1);
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    launch_fold_line(i4, o4, n_out, inv_x, to_q(alpha), nullptr);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_line_interpolate(const u32 *const in[4], u32 log_n, const u32 *itw, u32 tw_log, u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4);
    if (!itw) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer");
    if (log_n > 12) return set_error(TSTWO_ERR_BAD_ARG, "line_interpolate: at most 2^12 values (one workgroup); larger layers are interpolated by the caller");
    if (tw_log > 31 || log_n > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    const u32 n_inv = host::inv((u32)1 << log_n);
    hipLaunchKernelGGL(k_line_interpolate, dim3(1), dim3(256), (size_t)16 << log_n, ctx().stream, i4, o4, log_n, itw + ((size_t)1 << tw_log), n_inv);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_fri_fold_line(const u32 *const in[4], u32 log_n, const u32 *itw, u32 tw_log, const u32 alpha[4], u32 *const out[4]) {
    if (!itw) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer");
    if (log_n == 0) return set_error(TSTWO_ERR_TOO_SMALL, "fold_line: Evaluation too small, must have at least 2 elements.");
    if (tw_log > 31 || log_n > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    // level of the tree whose coset has log size log_n: 2^(log_n-1) entries starting 2^log_n before the end
    const u32 *seg = itw + ((size_t)1 << tw_log) - ((size_t)1 << log_n);
    return tstwo_fri_fold_line_tw(in, log_n, seg, alpha, out);
}

static int fold_circle_common(bool from_tree, u32 *const dst[4], size_t dst_len, const u32 *const src[4], u32 log_n,
                              const u32 *twp, const u32 alpha[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_TABLE(dst, 4); TSTWO_REQUIRE_TABLE(src, 4); TSTWO_REQUIRE_PTRS(twp, alpha);
    if (log_n == 0 || log_n > 31 || (((size_t)1 << log_n) >> 1) != dst_len)
        return set_error(TSTWO_ERR_LEN_MISMATCH, "fold_circle_into_line: Length mismatch between src and dst after considering fold step.");
    host::Q a = to_hq(alpha);
    host::Q a2 = host::qmul(a, a);
    Soa4 d4 = {{dst[0], dst[1], dst[2], dst[3]}};
    CSoa4 s4 = {{src[0], src[1], src[2], src[3]}};
    launch_fold_circle(from_tree, d4, s4, dst_len, twp, to_q(a), to_q(a2), nullptr);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_fri_fold_circle_into_line_tw(u32 *const dst[4], size_t dst_len, const u32 *const src[4], u32 log_n,
                                       const u32 *inv_y, const u32 alpha[4]) {
    return fold_circle_common(false, dst, dst_len, src, log_n, inv_y, alpha);
}

int tstwo_fri_fold_circle_into_line(u32 *const dst[4], size_t dst_len, const u32 *const src[4], u32 log_n,
                                    const u32 *itw, u32 tw_log, const u32 alpha[4]) {
    if (!itw) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer");
    if (log_n == 0 || log_n > 31 || (((size_t)1 << log_n) >> 1) != dst_len)
        return set_error(TSTWO_ERR_LEN_MISMATCH, "fold_circle_into_line: Length mismatch between src and dst after considering fold step.");
    if (log_n < 3) return set_error(TSTWO_ERR_BAD_ARG, "fold_circle_into_line: log_n < 3 needs explicit twiddles (tstwo_fri_fold_circle_into_line_tw)");
    if (tw_log > 31 || log_n - 1 > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    const u32 *seg1 = itw + ((size_t)1 << tw_log) - ((size_t)1 << (log_n - 1));   // layer-1 slice, 2^(log_n-2) entries
    return fold_circle_common(true, dst, dst_len, src, log_n, seg1, alpha);
}

// ---- alpha in device memory (written by tstwo_channel_mix_root_draw_felt on the same stream): the FRI commit loop then
// needs no host round trip between a layer's Merkle tree and the next fold.
int tstwo_fri_fold_line_dev(const u32 *const in[4], u32 log_n, const u32 *itw, u32 tw_log, const u32 *alpha_dev, u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    if (log_n == 0) return set_error(TSTWO_ERR_TOO_SMALL, "fold_line: Evaluation too small, must have at least 2 elements.");
    if (tw_log > 31 || log_n > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4); TSTWO_REQUIRE_PTRS(itw, alpha_dev);
    if (((uintptr_t)alpha_dev) & 15) return set_error(TSTWO_ERR_BAD_ARG, "fold: alpha must be 16-byte aligned");
    const size_t n_out = (size_t)1 << (log_n - 1);
    const u32 *seg = itw + ((size_t)1 << tw_log) - ((size_t)1 << log_n);
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    launch_fold_line(i4, o4, n_out, seg, qm31{0, 0, 0, 0}, (const qm31 *)alpha_dev);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_fri_fold_circle_into_line_dev(u32 *const dst[4], size_t dst_len, const u32 *const src[4], u32 log_n, const u32 *itw, u32 tw_log,
                                        const u32 *alpha_dev) {
    TSTWO_REQUIRE_READY();
    if (log_n == 0 || log_n > 31 || (((size_t)1 << log_n) >> 1) != dst_len)
        return set_error(TSTWO_ERR_LEN_MISMATCH, "fold_circle_into_line: Length mismatch between src and dst after considering fold step.");
    if (log_n < 3) return set_error(TSTWO_ERR_BAD_ARG, "fold_circle_into_line: log_n < 3 needs explicit twiddles (tstwo_fri_fold_circle_into_line_tw)");
    if (tw_log > 31 || log_n - 1 > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    TSTWO_REQUIRE_TABLE(dst, 4); TSTWO_REQUIRE_TABLE(src, 4); TSTWO_REQUIRE_PTRS(itw, alpha_dev);
    if (((uintptr_t)alpha_dev) & 15) return set_error(TSTWO_ERR_BAD_ARG, "fold: alpha must be 16-byte aligned");
    const u32 *seg1 = itw + ((size_t)1 << tw_log) - ((size_t)1 << (log_n - 1));
    Soa4 d4 = {{dst[0], dst[1], dst[2], dst[3]}};
    CSoa4 s4 = {{src[0], src[1], src[2], src[3]}};
    launch_fold_circle(true, d4, s4, dst_len, seg1, qm31{0, 0, 0, 0}, qm31{0, 0, 0, 0}, (const qm31 *)alpha_dev);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

// ---- row shards (contiguous row sharding of FRI layers, no exchange): the pointers address this
// shard's rows only; log_n is the WHOLE layer's size; the shard produces output rows [row_offset, row_offset+n_rows).
static int check_shard(const char *fn, u32 log_n, size_t row_offset, size_t n_rows) {
    size_t n_out = (size_t)1 << (log_n - 1);
    if (n_rows == 0 || row_offset + n_rows > n_out || (row_offset & 3) || ((n_rows & 3) && n_rows != n_out)) {
        char buf[160];
        snprintf(buf, sizeof buf, "%s: shard rows [%zu, +%zu) must be 4-aligned and lie inside the %zu output rows", fn, row_offset, n_rows, n_out);
        return set_error(TSTWO_ERR_BAD_ARG, buf);
    }
    return TSTWO_OK;
}

int tstwo_fri_fold_line_rows(const u32 *const in[4], u32 log_n, size_t row_offset, size_t n_rows, const u32 *itw, u32 tw_log,
                             const u32 alpha[4], u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4); TSTWO_REQUIRE_PTRS(itw, alpha);
    if (log_n == 0) return set_error(TSTWO_ERR_TOO_SMALL, "fold_line: Evaluation too small, must have at least 2 elements.");
    if (tw_log > 31 || log_n > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    int rc = check_shard("fold_line_rows", log_n, row_offset, n_rows);
    if (rc) return rc;
    const u32 *seg = itw + ((size_t)1 << tw_log) - ((size_t)1 << log_n) + row_offset;
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    launch_fold_line(i4, o4, n_rows, seg, to_q(alpha), nullptr);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_fri_fold_circle_into_line_rows(u32 *const dst[4], const u32 *const src[4], u32 log_n, size_t row_offset, size_t n_rows,
                                         const u32 *itw, u32 tw_log, const u32 alpha[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_TABLE(dst, 4); TSTWO_REQUIRE_TABLE(src, 4); TSTWO_REQUIRE_PTRS(itw, alpha);
    if (log_n < 3 || log_n > 31) return set_error(TSTWO_ERR_BAD_ARG, "fold_circle_into_line_rows: log_n must be in [3, 31]");
    if (tw_log > 31 || log_n - 1 > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    int rc = check_shard("fold_circle_into_line_rows", log_n, row_offset, n_rows);
    if (rc) return rc;
    // the kernel reads seg1[(i>>1)^1] with sign (i ^ (i>>1)) & 1: both are invariant under a 4-aligned shift of i
    const u32 *seg1 = itw + ((size_t)1 << tw_log) - ((size_t)1 << (log_n - 1)) + (row_offset >> 1);
    host::Q a = to_hq(alpha);
    host::Q a2 = host::qmul(a, a);
    Soa4 d4 = {{dst[0], dst[1], dst[2], dst[3]}};
    CSoa4 s4 = {{src[0], src[1], src[2], src[3]}};
    launch_fold_circle(true, d4, s4, n_rows, seg1, to_q(a), to_q(a2), nullptr);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

// ---- FriProver.commit's layer loop in ONE call (commitInnerLayers, fri.ts:676-716, with the Merkle / channel wiring of the Rust
// text): first-layer tree over every circle column's coordinate columns, then per layer: mix the root and draw alpha on the
// device channel, fold, commit the folded evaluation.  The host-side loop did the same through ~10 C-ABI calls and a dozen
// host objects per layer (40-50 us of host time each, more than the kernels of a layer below 2^16 rows take); here a layer
// costs its launches only.  Everything is enqueued on the library's stream; nothing is read back.
int tstwo_fri_commit_layers(const u32 *const *circle_cols, const u32 *col_logs, size_t n_columns, const u32 *itw, u32 tw_log,
                            u32 log_last_layer_size, u32 *chan, u32 *alphas, size_t alphas_cap, uint8_t **first_tree,
                            tstwo_fri_layer_out *out, size_t out_cap, size_t *n_out) {
    TSTWO_REQUIRE_READY();
    if (!n_columns) return set_error(TSTWO_ERR_BAD_ARG, "no columns");
    if (!circle_cols || !col_logs || !first_tree || !out || !n_out) return set_error(TSTWO_ERR_BAD_ARG, "fri commit: null argument");
    TSTWO_REQUIRE_TABLE(circle_cols, 4 * n_columns);
    TSTWO_REQUIRE_PTRS(itw, chan, alphas);
    if (((uintptr_t)alphas) & 15) return set_error(TSTWO_ERR_BAD_ARG, "fold: alpha must be 16-byte aligned");
    for (size_t i = 0; i < n_columns; i++) {
        if (col_logs[i] < 3 || col_logs[i] > 31) return set_error(TSTWO_ERR_BAD_ARG, "fri commit: circle evaluations of log size 3..31");
        if (i && col_logs[i - 1] <= col_logs[i]) return set_error(TSTWO_ERR_BAD_ARG, "column sizes not decreasing");
    }
    const u32 first_log = col_logs[0] - 1;          // CIRCLE_TO_LINE_FOLD_STEP = 1
    if (log_last_layer_size > first_log) return set_error(TSTWO_ERR_BAD_ARG, "fri commit: last layer larger than the first line layer");
    const size_t n_inner = first_log - log_last_layer_size;
    if (out_cap < n_inner + 1 || alphas_cap < n_inner + 1) return set_error(TSTWO_ERR_BAD_ARG, "fri commit: output / alpha capacity too small");
    *n_out = 0;
    *first_tree = nullptr;
    std::vector<void *> owned;                       // everything allocated here, released again if a step fails
    auto fail = [&](int rc) { for (void *p : owned) (void)tstwo_free(p); *n_out = 0; *first_tree = nullptr; return rc; };
    auto alloc = [&](void **p, size_t bytes) { int rc = tstwo_malloc(p, bytes); if (!rc) owned.push_back(*p); return rc; };
    auto alloc_eval = [&](u32 *cols[4], u32 lg) {
        for (int k = 0; k < 4; k++) { int rc = alloc((void **)&cols[k], sizeof(u32) << lg); if (rc) return rc; }
        return (int)TSTWO_OK;
    };
    int rc;
    // first layer: one tree over every column's coordinate columns (Rust FriFirstLayerProver::new), root -> channel -> alpha_0
    {
        std::vector<u32> logs(4 * n_columns);
        for (size_t i = 0; i < n_columns; i++) for (int k = 0; k < 4; k++) logs[4 * i + k] = col_logs[i];
        void *t = nullptr;
        if ((rc = alloc(&t, tstwo_merkle_layers_bytes(col_logs[0])))) return fail(rc);
        if ((rc = merkle_commit_then_channel(circle_cols, logs.data(), 4 * n_columns, (uint8_t *)t, chan, alphas))) return fail(rc);
        *first_tree = (uint8_t *)t;
    }
    u32 *alpha = alphas;
    u32 *cur[4];
    u32 cur_log = first_log;
    if ((rc = alloc_eval(cur, cur_log))) return fail(rc);
    size_t nxt = 0;
    {   // the first fold lands in a line evaluation that starts at zero (fri.ts:687-693): written, not accumulated — bit-identical
        // to zero-filling it and folding into it (0 * alpha^2 + x = x), without the fill and the read of the zeros
        if (tw_log > 31 || col_logs[0] - 1 > tw_log) return fail(set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!"));
        const u32 *seg1 = itw + ((size_t)1 << tw_log) - ((size_t)1 << (col_logs[0] - 1));
        Soa4 d4 = {{cur[0], cur[1], cur[2], cur[3]}};
        CSoa4 s4 = {{circle_cols[0], circle_cols[1], circle_cols[2], circle_cols[3]}};
        bool two = cur_log >= 2 && !knobs().fold1;
        for (int k = 0; k < 4; k++) two = two && (((uintptr_t)s4.p[k]) & 15) == 0 && (((uintptr_t)d4.p[k]) & 7) == 0;
        if (two)
            hipLaunchKernelGGL(k_fold_circle2<false>, dim3(capped_blocks((size_t)1 << (cur_log - 1), 256)), dim3(256), 0, ctx().stream, d4, s4,
                               (size_t)1 << cur_log, seg1, qm31{0, 0, 0, 0}, qm31{0, 0, 0, 0}, (const qm31 *)alpha);
        else
            hipLaunchKernelGGL((k_fold_circle<true, false>), dim3(capped_blocks((size_t)1 << cur_log, 256)), dim3(256), 0, ctx().stream, d4, s4,
                               (size_t)1 << cur_log, seg1, qm31{0, 0, 0, 0}, qm31{0, 0, 0, 0}, (const qm31 *)alpha);
        if (hipGetLastError() != hipSuccess) return fail(set_error(TSTWO_ERR_HIP, "fri commit: fold launch failed"));
        nxt = 1;
    }
    auto fold_circle_in = [&]() {      // the circle column whose folded size is the current line layer joins it (same alpha)
        const u32 *const src[4] = {circle_cols[4 * nxt], circle_cols[4 * nxt + 1], circle_cols[4 * nxt + 2], circle_cols[4 * nxt + 3]};
        int r = tstwo_fri_fold_circle_into_line_dev(cur, (size_t)1 << cur_log, src, col_logs[nxt], itw, tw_log, alpha);
        nxt++;
        return r;
    };
    size_t n = 0;
    const bool no_tail = knobs().fri_no_tail;      // A/B timing: per-layer launches down to the last layer
    uint8_t *cur_tree = nullptr;         // set: `cur` is already committed into it (its leaves were hashed by the fold that produced it)
                                         // and alpha (n + 1) is drawn
    uint8_t *spare_tree = nullptr;       // a tree buffer of the current size allocated for a fusion that an override refused
    const u32 *tail_pre[4] = {nullptr, nullptr, nullptr, nullptr};      // set: `cur` is still to be computed — the tail launch folds it
    const u32 *tail_pre_alpha = nullptr;                                // from this evaluation with this alpha (one launch fewer)
    while (cur_log > log_last_layer_size) {
        if (!cur_tree && !no_tail && cur_log <= 9 && nxt == n_columns) {
            // every remaining layer fits one workgroup's LDS: ONE launch does tree / mix / draw / fold for all of them (k_fri_tail)
            const u32 nl = cur_log - log_last_layer_size;
            u32 *ev[11][4];
            uint8_t *trees[10];
            for (int k = 0; k < 4; k++) ev[0][k] = cur[k];
            for (u32 i = 0; i < nl; i++) {
                void *t = i == 0 ? (void *)spare_tree : nullptr;
                if (!t && (rc = alloc(&t, tstwo_merkle_layers_bytes(cur_log - i)))) return fail(rc);
                trees[i] = (uint8_t *)t;
                if ((rc = alloc_eval(ev[i + 1], cur_log - i - 1))) return fail(rc);
            }
            spare_tree = nullptr;
            if ((rc = launch_fri_tail(ev, trees, nl, cur_log, itw, tw_log, chan, alphas + 4 * (n + 1), tail_pre_alpha ? tail_pre : nullptr, tail_pre_alpha)))
                return fail(rc);
            for (u32 i = 0; i < nl; i++) {
                out[n].log_size = cur_log - i;
                for (int k = 0; k < 4; k++) out[n].cols[k] = ev[i][k];
                out[n].layers = trees[i];
                n++;
            }
            for (int k = 0; k < 4; k++) cur[k] = ev[nl][k];
            cur_log = log_last_layer_size;
            break;
        }
        tstwo_fri_layer_out &o = out[n];
        o.log_size = cur_log;
        for (int k = 0; k < 4; k++) o.cols[k] = cur[k];
        alpha = alphas + 4 * (n + 1);
        if (cur_tree) {
            o.layers = cur_tree;
            cur_tree = nullptr;
        } else {
            void *t = spare_tree;
            spare_tree = nullptr;
            if (!t && (rc = alloc(&t, tstwo_merkle_layers_bytes(cur_log)))) return fail(rc);
            o.layers = (uint8_t *)t;
            const u32 lg4[4] = {cur_log, cur_log, cur_log, cur_log};
            if ((rc = merkle_commit_then_channel(cur, lg4, 4, o.layers, chan, alpha))) return fail(rc);      // FriInnerLayerProver::new + mix / draw
        }
        // fold this layer.  When the folded evaluation is committed as it stands (no circle column joins it, the tail does not take
        // it), the fold runs inside the leaf launch of ITS tree: the folded row is that tree's leaf message (merkle_commit4_folded)
        const u32 next_log = cur_log - 1;
        u32 *folded[4];
        if ((rc = alloc_eval(folded, next_log))) return fail(rc);
        const bool joins = nxt < n_columns && col_logs[nxt] - 1 == next_log;
        const bool tail_next = !no_tail && next_log <= 9 && nxt + (joins ? 1 : 0) == n_columns;
        bool fused = false;
        if (next_log > log_last_layer_size && !joins && !tail_next) {
            if (tw_log > 31 || cur_log > tw_log) return fail(set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!"));
            void *t = nullptr;
            if ((rc = alloc(&t, tstwo_merkle_layers_bytes(next_log)))) return fail(rc);
            const u32 *seg = itw + ((size_t)1 << tw_log) - ((size_t)1 << cur_log);
            rc = merkle_commit4_folded(cur, next_log, seg, alpha, folded, (uint8_t *)t, chan, alphas + 4 * (n + 2));
            if (rc == -1) spare_tree = (uint8_t *)t;          // an environment override keeps 4-column trees off the leaf4 kernels
            else if (rc) return fail(rc);
            else { cur_tree = (uint8_t *)t; fused = true; }
        }
        if (!fused && tail_next && !joins && next_log > log_last_layer_size) {
            for (int k = 0; k < 4; k++) tail_pre[k] = cur[k];          // the tail launch folds this layer on its way in
            tail_pre_alpha = alpha;
        } else if (!fused && (rc = tstwo_fri_fold_line_dev(cur, cur_log, itw, tw_log, alpha, folded))) return fail(rc);
        for (int k = 0; k < 4; k++) cur[k] = folded[k];
        cur_log = next_log;
        n++;
        if (joins)
            if ((rc = fold_circle_in())) return fail(rc);
    }
    if (nxt != n_columns) return fail(set_error(TSTWO_ERR_BAD_ARG, "not all columns were consumed"));      // Rust: assert!(columns.is_empty())
    out[n].log_size = cur_log;
    for (int k = 0; k < 4; k++) out[n].cols[k] = cur[k];
    out[n].layers = nullptr;                         // the last layer is interpolated, not committed (fri.ts:718-754)
    *n_out = n + 1;
    return TSTWO_OK;
}

int tstwo_fri_decompose(const u32 *const in[4], size_t n, u32 *const out[4], u32 lambda[4]) {
    TSTWO_REQUIRE_READY();
    if (n == 0) return set_error(TSTWO_ERR_BAD_ARG, "decompose: empty evaluation");
    TSTWO_REQUIRE_TABLE(in, 4); TSTWO_REQUIRE_TABLE(out, 4); TSTWO_REQUIRE_PTRS(lambda);
    Context &c = ctx();
    int rc = ensure_scratch(64);
    if (rc) return rc;
    unsigned long long *sums = (unsigned long long *)c.scratch;
    TSTWO_HIP(hipMemsetAsync(sums, 0, 8 * sizeof(unsigned long long), c.stream));
    CSoa4 i4 = {{in[0], in[1], in[2], in[3]}};
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    bool vec = n >= 8 && n % 8 == 0;              // both halves whole 16-byte vectors
    for (int k = 0; k < 4; k++) vec = vec && ((((uintptr_t)in[k]) | ((uintptr_t)out[k])) & 15) == 0;
    unsigned blocks = ceil_div(n, 256 * 16);
    if (blocks > 256) blocks = 256;           // (x 4 coordinates x 2 halves; every workgroup ends in one 64-bit atomic on one of 8 words: 4096 of them per word cost 0.25 ms)
    if (blocks == 0) blocks = 1;
    if (vec) hipLaunchKernelGGL(k_half_sums<true>, dim3(blocks, 4, 2), dim3(256), 0, c.stream, i4, n, sums);
    else hipLaunchKernelGGL(k_half_sums<false>, dim3(blocks, 4, 2), dim3(256), 0, c.stream, i4, n, sums);
    TSTWO_LAUNCH_CHECK();
    unsigned long long h[8];
    { int rc2 = small_d2h(h, sums, sizeof(h)); if (rc2) return rc2; }
    // lambda = (a_sum - b_sum) / n  (n == 1: first half empty -> lambda = -f[0])
    u32 n_inv = host::inv((u32)(n % host::P));
    qm31 lam;
    u32 l[4];
    for (int k = 0; k < 4; k++) {
        u32 a = (u32)(h[2 * k] % host::P), b = (u32)(h[2 * k + 1] % host::P);
        l[k] = host::mul(host::sub(a, b), n_inv);
    }
    lam = {l[0], l[1], l[2], l[3]};
    if (vec) hipLaunchKernelGGL(k_decompose_apply<true>, dim3(capped_blocks(n / 4, 256), 4), dim3(256), 0, c.stream, i4, o4, n, lam);
    else hipLaunchKernelGGL(k_decompose_apply<false>, dim3(capped_blocks(n, 256), 4), dim3(256), 0, c.stream, i4, o4, n, lam);
    TSTWO_LAUNCH_CHECK();
    for (int k = 0; k < 4; k++) lambda[k] = l[k];
    return TSTWO_OK;
}

// eval_at_point of n_cols polynomials of one size at one point: kernels E1/E2 above, one read-back of 16 bytes per column.
static int eval_at_point_impl(const u32 *const *coeffs, size_t n_cols, u32 log_size, const u32 px[4], const u32 py[4], u32 *out) {
    Context &c = ctx();
    if (log_size == 0) {   // circle.ts:53-59: the constant polynomial
        for (size_t i = 0; i < n_cols; i++) {
            u32 v;
            { int rc2 = small_d2h(&v, coeffs[i], 4); if (rc2) return rc2; }
            out[4 * i] = v; out[4 * i + 1] = out[4 * i + 2] = out[4 * i + 3] = 0;
        }
        return TSTWO_OK;
    }
    // fac[s] multiplies every coefficient whose index has bit s set: y, x, pi(x), ... (circle.ts:61-67 before the reverse)
    const host::Q one = {{1, 0, 0, 0}}, zero = {{0, 0, 0, 0}};
    host::Q fac[44];
    fac[0] = to_hq(py);
    {
        host::Q x = to_hq(px);
        for (u32 i = 1; i < log_size; i++) {
            fac[i] = x;
            host::Q sx = host::qmul(x, x);
            x = host::qsub(host::qadd(sx, sx), one);   // circle.ts:37-40
        }
        for (u32 i = log_size; i < 44; i++) fac[i] = zero;   // bits the polynomial does not have: those coefficients are zero
    }
    auto level_tables = [&](const int (&wbits)[4], u32 lane_bit0, EvalW &W, EvalF &F) {
        for (int e = 0; e < 16; e++) {
            host::Q v = one;
            for (int i = 0; i < 4; i++)
                if ((e >> i) & 1) v = host::qmul(v, fac[wbits[i]]);
            W.w[e] = to_q(v);
        }
        for (int i = 0; i < 8; i++) F.f[i] = to_q(fac[lane_bit0 + i]);
    };
    const size_t n_coeffs = (size_t)1 << log_size;
    // 64 coefficients per lane (G = 4) when the grid still has a workgroup per CU, else 16 (G = 1)
    const u32 glog = (log_size >= 14 && (((size_t)1 << (log_size - 14)) * n_cols >= (size_t)c.n_cus)) ? 2u : 0u;
    const u32 chunk_log = 12 + glog;
    const size_t chunks = log_size > chunk_log ? (size_t)1 << (log_size - chunk_log) : 1;
    const size_t groups1 = chunks > 4096 ? chunks / 4096 : 1;
    bool aligned = true;
    for (size_t i = 0; i < n_cols; i++) aligned = aligned && ((((uintptr_t)coeffs[i]) & 15) == 0);
    const bool fast = log_size >= chunk_log && aligned;
    const size_t kChunkCols = 32768;                       // gridDim.y limit
    for (size_t col0 = 0; col0 < n_cols; col0 += kChunkCols) {
        const size_t g = n_cols - col0 < kChunkCols ? n_cols - col0 : kChunkCols;
        int rc = ensure_scratch((g * (chunks + groups1) + 8) * sizeof(qm31));
        if (rc) return rc;
        qm31 *bufA = (qm31 *)c.scratch, *bufB = bufA + g * chunks;
        // the last level stores its g results straight into the page-locked host buffer (device-visible): the call then ends
        // with one stream synchronisation instead of a copy + synchronisation
        qm31 *host_dst = (c.pinned && g * sizeof(qm31) <= kPinnedBytes) ? (qm31 *)c.pinned : nullptr;
        ColPtrs cp;
        rc = fill_col_table(cp, coeffs + col0, g, 0);
        if (rc) return rc;
        EvalW W;
        EvalF F;
        qm31 *src = bufA, *dst = bufB;
        size_t m_in = chunks;
        u32 bit0 = chunk_log;
        {   // E1: bits 0,1 (j) and 10,11 (r) through W[4 r + j]; bits 2..9 are the lane bits; bits 12,13 the groups of a chunk
            const int wb[4] = {0, 1, 10, 11};
            level_tables(wb, 2, W, F);
            EvalH H;
            H.h[0] = to_q(one); H.h[1] = to_q(fac[12]); H.h[2] = to_q(fac[13]); H.h[3] = to_q(host::qmul(fac[12], fac[13]));
            const size_t stride = chunks == 1 ? 1 : chunks;
            qm31 *o = (chunks == 1 && host_dst) ? host_dst : bufA;
            const dim3 grid((unsigned)chunks, (unsigned)g);
            if (glog) {
                if (fast) hipLaunchKernelGGL((k_eval_coeffs<true, 4>), grid, dim3(256), 0, c.stream, cp, n_coeffs, W, F, H, o, stride);
                else hipLaunchKernelGGL((k_eval_coeffs<false, 4>), grid, dim3(256), 0, c.stream, cp, n_coeffs, W, F, H, o, stride);
            } else {
                if (fast) hipLaunchKernelGGL((k_eval_coeffs<true, 1>), grid, dim3(256), 0, c.stream, cp, n_coeffs, W, F, H, o, stride);
                else hipLaunchKernelGGL((k_eval_coeffs<false, 1>), grid, dim3(256), 0, c.stream, cp, n_coeffs, W, F, H, o, stride);
            }
        }
        while (m_in > 1) {   // E2: 12 more bits per level (lane bits bit0..bit0+7, W over bit0+8..bit0+11)
            const int wb[4] = {(int)bit0 + 8, (int)bit0 + 9, (int)bit0 + 10, (int)bit0 + 11};
            level_tables(wb, bit0, W, F);
            const size_t groups = m_in > 4096 ? m_in / 4096 : 1;
            hipLaunchKernelGGL(k_eval_partials, dim3((unsigned)groups, (unsigned)g), dim3(256), 0, c.stream, (const qm31 *)src, m_in, m_in, W, F,
                               (groups == 1 && host_dst) ? host_dst : dst, groups);
            qm31 *tmp = src; src = dst; dst = tmp;
            m_in = groups;
            bit0 += 12;
        }
        TSTWO_LAUNCH_CHECK();
        // one QM31 per column, contiguous (the last level has stride 1)
        if (host_dst) {
            if (int rcw = wait_stream()) return rcw;
            memcpy(out + 4 * col0, host_dst, g * sizeof(qm31));
        } else {
            rc = small_d2h(out + 4 * col0, src, g * sizeof(qm31));
            if (rc) return rc;
        }
    }
    return TSTWO_OK;
}

int tstwo_eval_at_point(const u32 *coeffs, u32 log_size, const u32 px[4], const u32 py[4], u32 out[4]) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_PTRS(coeffs, px, py, out);
    if (log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "eval_at_point: log size out of range");
    return eval_at_point_impl(&coeffs, 1, log_size, px, py, out);
}

// eval_at_point of n_cols polynomials of one size at one point (CommitmentSchemeProver.prove_values samples every
// column of a tree at the same out-of-domain point, pcs/prover.ts Rust text :93-110): one launch sequence and one
// read-back for all of them.  out = 4 words per column.
int tstwo_eval_at_point_batch(const u32 *const *coeffs, size_t n_cols, u32 log_size, const u32 px[4], const u32 py[4], u32 *out) {
    TSTWO_REQUIRE_READY();
    if (n_cols == 0) return TSTWO_OK;
    if (!coeffs || !out) return set_error(TSTWO_ERR_BAD_ARG, "eval_at_point: null argument");
    TSTWO_REQUIRE_TABLE(coeffs, n_cols); TSTWO_REQUIRE_PTRS(px, py);
    if (log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "eval_at_point: log size out of range");
    return eval_at_point_impl(coeffs, n_cols, log_size, px, py, out);
}

}  // extern "C"
