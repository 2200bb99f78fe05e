// cfft.hip — Circle FFT (PolyOps.evaluate / PolyOps.interpolate) for gfx950.
//
// What it computes (reference: backend/cpu/circle.ts:84-207,243-278, poly/utils.ts:78-100):
//   layer i pairs v[idx] with v[idx + 2^i]; its twiddle depends only on h = idx >> (i+1):
//     line layers i >= 1 : tw(i,h) = tree[L - 2^(n-i) + h]           (tree = twiddle buffer, L = |tree|)
//     circle layer i = 0 : tw(0,h) = +-tw(1, (h>>1)^1), negative iff (h ^ (h>>1)) & 1   ([y,-y,-x,x])
//   evaluate  : layers n-1 .. 0 with butterfly  (v0 + v1 t, v0 - v1 t)
//   interpolate: layers 0 .. n-1 with ibutterfly (v0 + v1, (v0 - v1) t) on the inverse tree, then * 2^-n.
//
// How it maps to the MI355X: the n layers are cut into passes over HBM.  A pass owns a contiguous
// range of layers [lo, lo+k) and a workgroup owns a *tile*: every index whose bits outside
// [lo, lo+k) u [0, c) are fixed, i.e. 2^k "rows" of 2^c contiguous words.  The tile is staged in LDS
// (coalesced 16-byte global accesses; rows are >= 128 B so strided passes still move whole lines),
// the k layers run as register radix-32/16 stages (G = 5/4 layers per LDS round trip, every
// butterfly pair owned by exactly one lane), and the tile is written back in place.  log 22 = two
// passes: layers 21..13 on [2^9 rows x 32 words] tiles, then layers 12..0 on contiguous 2^13 tiles.
// LDS index e is padded as e + (e >> 5): unit-stride runs and the stride-32 pattern of the bottom
// radix-32 stage are both bank-conflict free (32 banks for 4-byte accesses).
// Twiddles are read straight from the tree (L2 resident; a tile's twiddles are shared by every
// column, and blocks of the same tile are launched adjacently).
//
// Algorithmic bytes: 8 per element per transform (column read once + written once); real HBM/MALL
// traffic: 8 per element per pass (+ <= 2 for twiddles).  DESIGN.md §CFFT has the roofline numbers.
#include <unordered_map>
#include <unordered_set>

#include "common.h"
#include "host_field.h"
#include <stdlib.h>

using namespace tstwo;

namespace {

constexpr u32 kMaxLogTileB = 13;   // contiguous (bottom) tile: 2^13 words = 32 KiB + pad, 256 lanes
constexpr u32 kLogTileA = 14;      // strided tile: 2^k rows x 2^(14-k) words = 64 KiB + pad, 512 lanes
constexpr u32 kMaxLogSize = 30;    // MAX_CIRCLE_DOMAIN_LOG_SIZE (poly/circle/domain.ts:4): a 4 GiB column; word offsets stay below 2^30
constexpr u32 kMaxKA = 9;          // at most 9 layers per strided pass (rows of >= 32 words = 128 B)
constexpr int kThreadsB = 256, kThreadsA = 512;
constexpr int kMaxV4 = 8;          // 16-byte vectors per lane per tile (32 words)

struct PassParams {
    u32 n;        // log size of the column
    u32 lo;       // lowest layer of this pass
    u32 k;        // layers [lo, lo+k)
    u32 c;        // log2(words per row); 0 for the bottom pass (lo == 0, one contiguous row)
    u32 logt;     // c + k = log2(tile words)
    u32 scale;    // interpolate's 2^-n, applied by the last pass; 0 = no scaling
    u32 cols_per_wg;
    const u32 *tw_end;  // tree + L
};

__device__ __forceinline__ u32 phys(u32 e) { return e + (e >> 5); }

// x * t for a twiddle stored doubled (t2 = 2t < 2^32): the 64-bit product's high word is
// floor(x t / 2^31) and its low word >> 1 is (x t) mod 2^31, so the Mersenne fold needs no
// and/alignbit (stwo's SIMD backend keeps "dbl" twiddles for the same reason).
__device__ __forceinline__ u32 m31_mul_dbl(u32 x, u32 t2) {
    u64 p = (u64)x * (u64)t2;
    u32 s = (u32)(p >> 32) + ((u32)p >> 1);
    return min(s, s - M31_P);
}
__device__ __forceinline__ void bf_dbl(u32 &v0, u32 &v1, u32 t2) {
    u32 m = m31_mul_dbl(v1, t2);
    u32 a = m31_add(v0, m);
    v1 = m31_sub(v0, m);
    v0 = a;
}
__device__ __forceinline__ void ibf_dbl(u32 &v0, u32 &v1, u32 t2) {
    u32 a = m31_add(v0, v1);
    v1 = m31_mul_dbl(m31_sub(v0, v1), t2);
    v0 = a;
}

// ---- butterflies in priority phases (the issue model and TSTWO_PHASE: common.h)
// N independent butterflies (x[i], y[i]) with doubled twiddles t2[i]; leaves the wave at kPrioHeavy (the next layer starts
// heavy as well) — the caller drops to kPrioLight after its last layer.
template <bool INV, int N>
__device__ __forceinline__ void bf_layer(u32 (&x)[N], u32 (&y)[N], const u32 (&t2)[N]) {
    static_assert(N == 8 || N == 4, "phase() pins 4, 8 or 16 values");
    const u32 P = vgpr_P();
    u32 s[N], d[N], u[N], w[N], u2[N], w2[N];
    u64 p[N];
    if (!INV) {
        phase<kPrioHeavy>(y);
#pragma unroll
        for (int i = 0; i < N; i++) p[i] = (u64)y[i] * (u64)t2[i];
        phase<kPrioLight>(p);
#pragma unroll
        for (int i = 0; i < N; i++) { s[i] = (u32)(p[i] >> 32) + ((u32)p[i] >> 1); d[i] = s[i] - P; }
        phase<kPrioHeavy>(d);
#pragma unroll
        for (int i = 0; i < N; i++) s[i] = min(s[i], d[i]);
        phase<kPrioLight>(s);
#pragma unroll
        for (int i = 0; i < N; i++) { u[i] = x[i] + s[i]; w[i] = x[i] - s[i]; u2[i] = u[i] - P; w2[i] = w[i] + P; }
        phase<kPrioHeavy>(u2, w2);
#pragma unroll
        for (int i = 0; i < N; i++) { x[i] = min(u[i], u2[i]); y[i] = min(w[i], w2[i]); }
    } else {
        phase<kPrioLight>(x, y);
#pragma unroll
        for (int i = 0; i < N; i++) { u[i] = x[i] + y[i]; w[i] = x[i] - y[i]; u2[i] = u[i] - P; w2[i] = w[i] + P; }
        phase<kPrioHeavy>(u2, w2);
#pragma unroll
        for (int i = 0; i < N; i++) { x[i] = min(u[i], u2[i]); p[i] = (u64)min(w[i], w2[i]) * (u64)t2[i]; }
        phase<kPrioLight>(p);
#pragma unroll
        for (int i = 0; i < N; i++) { s[i] = (u32)(p[i] >> 32) + ((u32)p[i] >> 1); d[i] = s[i] - P; }
        phase<kPrioHeavy>(d);
#pragma unroll
        for (int i = 0; i < N; i++) y[i] = min(s[i], d[i]);
    }
}

// v[i] *= t (t2 = 2t, wave-uniform): the 2^-n scale of the last inverse pass, in the same phases (leaves kPrioHeavy)
__device__ __forceinline__ void mul8_dbl(u32 (&v)[8], u32 t2) {
    const u32 P = vgpr_P();
    u32 s[8], d[8];
    u64 p[8];
    phase<kPrioHeavy>(v);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = (u64)v[i] * (u64)t2;
    phase<kPrioLight>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) { s[i] = (u32)(p[i] >> 32) + ((u32)p[i] >> 1); d[i] = s[i] - P; }
    phase<kPrioHeavy>(d);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = min(s[i], d[i]);
}

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits vmcnt(0), which would
// serialise the in-flight prefetch loads / tile stores behind every stage (guide §5 "Pipelining across
// barriers"); global memory is never shared between lanes inside these kernels, so LDS ordering suffices.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Ordering between LDS accesses of ONE wave (a stage whose exchange stays inside the wave): the LDS executes a wave's
// instructions in issue order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void lds_wave_fence() { asm volatile("" ::: "memory"); }

}  // namespace
#include "cfft_fast.cuh"
namespace {

// One register stage: G consecutive layers on LDS bits [q, q+G).  Twiddles come from the tile's LDS
// heap `twl` (doubled values): layer bit b lives at twl[2^(logt-1-b) + (e >> (b+1))].
// CIRCLE: bit 0 of this stage is the circle layer (only when q == 0 and lo == 0; needs G >= 3).
template <int G, bool INV, bool CIRCLE, int THREADS>
__device__ __forceinline__ void run_stage(u32 *lds, const u32 *twl, const u32 q, const u32 logt) {
    const u32 ngroups = 1u << (logt - G);
    for (u32 gid = threadIdx.x; gid < ngroups; gid += THREADS) {
        const u32 low = gid & ((1u << q) - 1u), high = gid >> q;
        const u32 e0 = (high << (q + G)) | low;
        u32 v[1 << G];
#pragma unroll
        for (int m = 0; m < (1 << G); m++) v[m] = lds[phys(e0 + ((u32)m << q))];
#pragma unroll
        for (int step = 0; step < G; step++) {
            const int l = INV ? step : (G - 1 - step);
            const u32 b = q + (u32)l;                       // LDS bit of this layer
            if (CIRCLE && l == 0) {
                // tw(0,h) = +-tw(1,(h>>1)^1), negative iff (h ^ (h>>1)) & 1   (backend/cpu/circle.ts:270-278)
                const u32 *t1 = twl + (1u << (logt - 2)) + (high << (G - 2));
#pragma unroll
                for (int j = 0; j < (1 << (G - 1)); j++) {
                    u32 t2 = t1[(j >> 1) ^ 1];
                    if ((j ^ (j >> 1)) & 1) t2 = 0xFFFFFFFEu - t2;      // 2(P - t)
                    if (INV) ibf_dbl(v[2 * j], v[2 * j + 1], t2);
                    else bf_dbl(v[2 * j], v[2 * j + 1], t2);
                }
            } else {
                const u32 *tl = twl + (1u << (logt - 1 - b)) + (high << (G - 1 - l));
#pragma unroll
                for (int j = 0; j < (1 << (G - 1 - l)); j++) {
                    const u32 t2 = tl[j];
#pragma unroll
                    for (int r = 0; r < (1 << l); r++) {
                        const int m0 = (j << (l + 1)) | r;
                        if (INV) ibf_dbl(v[m0], v[m0 + (1 << l)], t2);
                        else bf_dbl(v[m0], v[m0 + (1 << l)], t2);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << G); m++) lds[phys(e0 + ((u32)m << q))] = v[m];
    }
}

template <bool INV, bool CIRCLE, int THREADS>
__device__ __forceinline__ void dispatch_stage(int g, u32 *lds, const u32 *twl, u32 q, u32 logt) {
    switch (g) {
        case 5: run_stage<5, INV, CIRCLE, THREADS>(lds, twl, q, logt); break;
        case 4: run_stage<4, INV, CIRCLE, THREADS>(lds, twl, q, logt); break;
        case 3: run_stage<3, INV, CIRCLE, THREADS>(lds, twl, q, logt); break;
        case 2: if (!CIRCLE) run_stage<2, INV, false, THREADS>(lds, twl, q, logt); break;
        case 1: if (!CIRCLE) run_stage<1, INV, false, THREADS>(lds, twl, q, logt); break;
        default: break;
    }
}

// Stage schedule for the layer bits [c, logt) of a tile: the bottom pass (lo == 0) spends min(5, k)
// layers on a conflict-free radix-32 stage at q = 0, everything else is cut into balanced stages of <= 5
// layers (q >= 5 there, so lanes read unit-stride runs).  Returned low -> high.
__device__ __forceinline__ int plan_stages(const PassParams &pp, int *g_out) {
    int cnt = 0;
    u32 rem = pp.k;
    if (pp.lo == 0) {
        u32 g0 = rem < 5 ? rem : 5;
        g_out[cnt++] = (int)g0;
        rem -= g0;
    }
    if (rem) {
        u32 stages = (rem + 4) / 5;
        u32 base = rem / stages, extra = rem % stages;
        for (u32 s = 0; s < stages; s++) g_out[cnt++] = (int)(base + (s < extra ? 1 : 0));
    }
    return cnt;
}

// One workgroup = one tile position x `cols_per_wg` columns.  The tile's twiddles are staged once
// in LDS and reused for every column; the next column's tile is prefetched into registers while
// the current one is transformed, so HBM latency overlaps the butterflies inside the workgroup.
template <bool INV, int THREADS>
// (512 lanes = the strided passes of the generic route, which only the experiments build's TSTWO_CFFT_GENERIC reaches: asked for 2
// waves per SIMD it keeps its radix-32 stage in registers; at 4 it spilled 120-132 bytes per lane at 128 VGPRs)
__global__ void __launch_bounds__(THREADS, (THREADS == 512 ? 2 : 3)) k_cfft_pass(ColPtrs cols, u32 n_cols, PassParams pp) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 tile_words = 1u << pp.logt;
    u32 *twl = lds + ((tile_words + (tile_words >> 5) + 3u) & ~3u);     // twiddle heap behind the padded tile

    // column group is the fastest-varying block coordinate: blocks sharing a tile (hence twiddles) are adjacent
    const u32 groups = (n_cols + pp.cols_per_wg - 1) / pp.cols_per_wg;
    const u32 cgroup = blockIdx.x % groups;
    const u32 tile = blockIdx.x / groups;
    const u32 col0 = cgroup * pp.cols_per_wg;
    const u32 col1 = min(col0 + pp.cols_per_wg, n_cols);

    const u32 mid_bits = pp.lo - pp.c;                     // 0 for the bottom pass
    const u32 mid = tile & ((1u << mid_bits) - 1u);
    const u32 hi = tile >> mid_bits;
    const size_t base = ((size_t)hi << (pp.lo + pp.k)) | ((size_t)mid << pp.c);
    const u32 cmask = (1u << pp.c) - 1u;

    // ---- prefetch the first column's tile (16 bytes per lane per access)
    uint4 pf[kMaxV4];
    {
        const u32 *__restrict__ data = colp_u(cols, col0);
#pragma unroll
        for (int it = 0; it < kMaxV4; it++) {
            const u32 e = 4u * (threadIdx.x + (u32)it * THREADS);
            if (e < tile_words) pf[it] = gload4(data + base + ((size_t)(e >> pp.c) << pp.lo) + (e & cmask));
        }
    }
    // ---- twiddle heap: level lv (2^lv entries at twl[2^lv ..]) holds layer bit b = logt-1-lv, i.e. global
    //      layer i = lo + b - c, entries tree[L - 2^(n-i) + (hi << lv) + hl]; stored doubled.
    {
        const u32 skip = (pp.lo == 0) ? 1u : 0u;            // the circle layer has no entries of its own
        const u32 heap = 1u << (pp.k - skip);
        for (u32 idx = threadIdx.x + 1; idx < heap; idx += THREADS) {
            const u32 lv = 31u - (u32)__clz(idx);
            const u32 b = pp.logt - 1u - lv;
            const u32 i = pp.lo + b - pp.c;
            const u32 t = pp.tw_end[-(ptrdiff_t)(1u << (pp.n - i)) + (ptrdiff_t)((hi << lv) + (idx - (1u << lv)))];
            twl[idx] = t + t;
        }
    }

    int gs[8];
    const int n_stages = plan_stages(pp, gs);

    for (u32 col = col0; col < col1; col++) {
        // ---- registers -> LDS
#pragma unroll
        for (int it = 0; it < kMaxV4; it++) {
            const u32 e = 4u * (threadIdx.x + (u32)it * THREADS);
            if (e < tile_words) {
                const u32 p = phys(e);
                lds[p] = pf[it].x; lds[p + 1] = pf[it].y; lds[p + 2] = pf[it].z; lds[p + 3] = pf[it].w;
            }
        }
        __syncthreads();
        // ---- prefetch the next column while this one is transformed
        if (col + 1 < col1) {
            const u32 *__restrict__ next = colp_u(cols, col + 1);
#pragma unroll
            for (int it = 0; it < kMaxV4; it++) {
                const u32 e = 4u * (threadIdx.x + (u32)it * THREADS);
                if (e < tile_words) pf[it] = gload4(next + base + ((size_t)(e >> pp.c) << pp.lo) + (e & cmask));
            }
        }
        if (!INV) {
            u32 q = pp.logt;
            for (int s = n_stages - 1; s >= 0; s--) {
                q -= (u32)gs[s];
                if (pp.lo == 0 && q == 0) dispatch_stage<false, true, THREADS>(gs[s], lds, twl, q, pp.logt);
                else dispatch_stage<false, false, THREADS>(gs[s], lds, twl, q, pp.logt);
                __syncthreads();
            }
        } else {
            u32 q = pp.c;
            for (int s = 0; s < n_stages; s++) {
                if (pp.lo == 0 && q == 0) dispatch_stage<true, true, THREADS>(gs[s], lds, twl, q, pp.logt);
                else dispatch_stage<true, false, THREADS>(gs[s], lds, twl, q, pp.logt);
                q += (u32)gs[s];
                __syncthreads();
            }
        }
        // ---- LDS -> global (fused 2^-n scaling on interpolate's last pass)
        u32 *__restrict__ data = colp_u(cols, col);
#pragma unroll
        for (int it = 0; it < kMaxV4; it++) {
            const u32 e = 4u * (threadIdx.x + (u32)it * THREADS);
            if (e < tile_words) {
                const u32 p = phys(e);
                uint4 x = make_uint4(lds[p], lds[p + 1], lds[p + 2], lds[p + 3]);
                if (INV && pp.scale) {
                    x.x = m31_mul(x.x, pp.scale); x.y = m31_mul(x.y, pp.scale);
                    x.z = m31_mul(x.z, pp.scale); x.w = m31_mul(x.w, pp.scale);
                }
                gstore4(data + base + ((size_t)(e >> pp.c) << pp.lo) + (e & cmask), x);
            }
        }
        __syncthreads();
    }
}

// log_size 1 and 2 (backend/cpu/circle.ts:93-110,153-185): twiddles are coordinates of the half
// coset's initial point, supplied by the host.  tx/ty (and the scale) are already inverted for INV.
template <bool INV>
__global__ void k_cfft_small(ColPtrs cols, u32 n_cols, u32 n, u32 tx, u32 ty, u32 scale) {
    u32 col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    u32 *v = colp(cols, col);
    if (n == 1) {
        u32 v0 = gload1(v), v1 = gload1(v + 1);
        if (!INV) m31_butterfly(v0, v1, ty);
        else { m31_ibutterfly(v0, v1, ty); v0 = m31_mul(v0, scale); v1 = m31_mul(v1, scale); }
        gstore1(v, v0); gstore1(v + 1, v1);
    } else {
        u32 v0 = gload1(v), v1 = gload1(v + 1), v2 = gload1(v + 2), v3 = gload1(v + 3);
        if (!INV) {
            m31_butterfly(v0, v2, tx); m31_butterfly(v1, v3, tx);
            m31_butterfly(v0, v1, ty); m31_butterfly(v2, v3, m31_neg(ty));
        } else {
            m31_ibutterfly(v0, v1, ty); m31_ibutterfly(v2, v3, m31_neg(ty));
            m31_ibutterfly(v0, v2, tx); m31_ibutterfly(v1, v3, tx);
            v0 = m31_mul(v0, scale); v1 = m31_mul(v1, scale); v2 = m31_mul(v2, scale); v3 = m31_mul(v3, scale);
        }
        gstore1(v, v0); gstore1(v + 1, v1); gstore1(v + 2, v2); gstore1(v + 3, v3);
    }
}

struct Pass { u32 lo, k, c; };
// passes low -> high.  kb = layers of the bottom pass (contiguous tile), the rest is cut into strided passes of <= ka_max.
int plan_passes(u32 n, Pass *out, u32 kb = kMaxLogTileB, u32 ka_max = kMaxKA, u32 logta = kLogTileA) {
    int cnt = 0;
    if (n <= kb) {
        out[cnt++] = {0, n, 0};
        return cnt;
    }
    out[cnt++] = {0, kb, 0};
    u32 rem = n - kb, lo = kb;
    u32 np = (rem + ka_max - 1) / ka_max;
    u32 base = rem / np, extra = rem % np;
    for (u32 s = 0; s < np; s++) {
        u32 k = base + (s < extra ? 1 : 0);
        u32 c = logta - k;
        if (c > lo) c = lo;
        out[cnt++] = {lo, k, c};
        lo += k;
    }
    return cnt;
}

// The default plan of a many-column transform (evaluate / interpolate / interpolate_to / evaluate_extended share it):
//   n <= 13, n = 14      ONE pass on the contiguous tile (2^14 words at n = 14: 256 columns 16.4 against 21.2 us for 13 + 1);
//   15 <= n <= 20        13 bottom layers + one strided pass on 2^14-word tiles (n = 19, 20 on the 2^15 tile measured equal or 1 % slower);
//   n = 21               13 + 8 on the 2^15-word tile (512-byte rows) from two workgroups per CU on: 512 x 2^21 3.60 against 3.64 ms,
//                        interpolate 3.78 against 3.85 (128 columns: 0.930 / 0.976 against 0.938 / 0.989);
//   n = 22, 23           13 + 9 / 13 + 10 with the strided pass on the 2^15-word tile (k_cfft_a<., K, ., 15>: two virtual lanes per
//                        lane, 256- / 128-byte rows) when the launch still has two workgroups per CU — round 4, same box, 256 x 2^22:
//                        3.775 against 3.805 ms forward, 3.985 against 4.027 inverse; 128 x 2^23: 3.815 against 3.895 (14 + 9 on
//                        2^14-word tiles, round 3's two-pass plan, which stays for few columns) and 4.02 against 4.08;
//   n = 24               14 + 10, the strided pass on the 2^15-word tile: TWO passes instead of 13 + 6 + 5 (round 4: 32 x 2^24
//                        2.09 against 2.66 ms, 64 columns 4.08 against 5.29);
//   n >= 25              13 bottom layers + two strided passes (a 2^14-word bottom tile changes nothing there).
// 12 + 10 at n = 22 (a layer moved from the issue-bound bottom pass to the HBM-bound strided one) measured 3.86 against 3.78.
struct PlanShape { u32 kb, ka_max, logta; };
inline PlanShape default_shape(u32 n, size_t n_cols) {
    const bool wide = n >= 15 && (((size_t)1 << (n - 15)) * n_cols) >= (size_t)2 * (size_t)ctx().n_cus;     // the 2^15 tile halves the workgroup count
    if (n == 24) return {14u, 10u, 15u};
    if (n == 23) return wide ? PlanShape{13u, 10u, 15u} : PlanShape{14u, kMaxKA, kLogTileA};
    if ((n == 22 || n == 21) && wide) return {13u, 9u, 15u};
    if (n == 14) return {14u, kMaxKA, kLogTileA};
    return {kMaxLogTileB, kMaxKA, kLogTileA};
}
inline int plan_default(u32 n, size_t n_cols, Pass *out) {
    const PlanShape sh = default_shape(n, n_cols);
    return plan_passes(n, out, sh.kb, sh.ka_max, sh.logta);
}

template <bool INV, int THREADS>
int launch_pass_t(u32 *const *cols, size_t n_cols, const PassParams &pp0) {
    Context &c = ctx();
    PassParams pp = pp0;
    const size_t tile_words = (size_t)1 << pp.logt;
    const size_t heap_words = (size_t)1 << (pp.k - (pp.lo == 0 ? 1 : 0));
    const size_t lds_bytes = (((tile_words + (tile_words >> 5) + 3) & ~(size_t)3) + heap_words + 4) * sizeof(u32);
    static bool lds_attr_set = false;   // tiles above 64 KiB need the opt-in (160 KiB LDS per CU on gfx950)
    if (!lds_attr_set) {
        TSTWO_HIP(hipFuncSetAttribute((const void *)k_cfft_pass<INV, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_attr_set = true;
    }
    const size_t tiles = (size_t)1 << (pp.n - pp.logt);
    {   // every column in one launch (more than 64: pointer table in device memory)
        const size_t cnt = n_cols;
        ColPtrs cp;
        int rc_tab = fill_col_table(cp, cols, n_cols, 0);
        if (rc_tab) return rc_tab;
        // columns per workgroup: amortise the twiddle staging, but keep >= ~6 workgroups per CU in the grid
        u32 cpw = 4;
        while (cpw > 1 && tiles * ((cnt + cpw - 1) / cpw) < (size_t)c.n_cus * 6) cpw >>= 1;
        if (cpw > cnt) cpw = (u32)cnt;
        pp.cols_per_wg = cpw;
        size_t blocks = tiles * ((cnt + cpw - 1) / cpw);
        if (blocks > 0x7fffffffu) return set_error(TSTWO_ERR_BAD_ARG, "cfft: grid too large");
        hipLaunchKernelGGL((k_cfft_pass<INV, THREADS>), dim3((unsigned)blocks), dim3(THREADS), lds_bytes, c.stream, cp, (u32)cnt, pp);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

template <bool INV>
int launch_pass(u32 *const *cols, size_t n_cols, u32 n, const Pass &ps, const u32 *tw, u32 tw_log, u32 scale) {
    PassParams pp;
    pp.n = n; pp.lo = ps.lo; pp.k = ps.k; pp.c = ps.c; pp.logt = ps.c + ps.k; pp.scale = scale; pp.cols_per_wg = 1;
    pp.tw_end = tw + ((size_t)1 << tw_log);
    for (size_t i = 0; i < n_cols; i++)
        if (((uintptr_t)cols[i]) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: columns must be 16-byte aligned");
    if (ps.lo == 0) return launch_pass_t<INV, kThreadsB>(cols, n_cols, pp);
    return launch_pass_t<INV, kThreadsA>(cols, n_cols, pp);
}

// Grid of a pass kernel: its (tile, column) work items are dealt in equal contiguous shares to as many workgroups as the
// chip holds at once (CUs x resident workgroups per CU for this kernel's registers / LDS), so the whole pass is ONE round of
// workgroups with no tail round and each workgroup stages a tile's twiddles once for all of its columns of that tile.
// (32 x 2^22: 2048 workgroups of 8 columns on 768 resident slots made 2.67 rounds, i.e. 11 % of the slots idle in the last
// one, and paid the twiddle / first-load prologue 2048 times instead of 768.)  TSTWO_CFFT_ROUNDS=k asks for k x that many
// workgroups (dynamic balance against the extra prologues); fewer items than slots: one item per workgroup.
unsigned plan_grid(const void *kernel, int threads, size_t lds_bytes, size_t total_items) {
    static std::unordered_map<const void *, int> resident;        // workgroups per CU, per kernel
    auto it = resident.find(kernel);
    if (it == resident.end()) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        it = resident.emplace(kernel, per_cu).first;
    }
    size_t slots = (size_t)ctx().n_cus * (size_t)it->second;
    slots *= (size_t)knobs().cfft_rounds;
    const int cap = knobs().cfft_maxwg;      // experiments
    if (cap && slots > (size_t)cap) slots = (size_t)cap;
    return (unsigned)(total_items < slots ? total_items : slots);
}

// Tiles above 64 KiB of LDS need a per-kernel opt-in (160 KiB per CU on gfx950); done once per kernel, not per launch.
int allow_big_lds(const void *kernel) {
    static std::unordered_set<const void *> done;
    if (done.insert(kernel).second)
        TSTWO_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return TSTWO_OK;
}

template <typename KernelT, typename... Args>
int launch_fast_kernel(KernelT kernel, int threads, size_t lds_bytes, size_t tiles, u32 *const *cols, size_t n_cols, Args... args) {
    // kernel signature: (ColPtrs cols, NoSrc, n_cols, total_items, args...)
    Context &c = ctx();
    lds_bytes += (size_t)knobs().cfft_lds_pad;     // experiments: extra (unused) dynamic LDS per workgroup lowers the resident workgroups per CU
    { int rc_attr = allow_big_lds((const void *)kernel); if (rc_attr) return rc_attr; }
    if (knobs().cfft_trace) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, (const void *)kernel);
        fprintf(stderr, "[cfft] kernel %p threads %d lds %zu: getattr=%d maxDynamicSharedSizeBytes=%d sharedSizeBytes=%zu numRegs=%d maxThreadsPerBlock=%d\n",
                (const void *)kernel, threads, lds_bytes, (int)e, fa.maxDynamicSharedSizeBytes, fa.sharedSizeBytes, fa.numRegs, fa.maxThreadsPerBlock);
    }
    {
        const size_t cnt = n_cols;
        ColPtrs cp;
        int rc_tab = fill_col_table(cp, cols, n_cols, 0);
        if (rc_tab) return rc_tab;
        const size_t items = tiles * cnt;
        if (items > 0xffffffffu) return set_error(TSTWO_ERR_BAD_ARG, "cfft: too many (tile, column) work items");
        const unsigned blocks = plan_grid((const void *)kernel, threads, lds_bytes, items);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, c.stream, cp, fast::NoSrc{}, (u32)cnt, (u32)items, args...);
    }
    TSTWO_LAUNCH_CHECK();
    if (knobs().cfft_sync) TSTWO_HIP(hipStreamSynchronize(c.stream));
    return TSTWO_OK;
}

template <bool INV, int K, int LOGTA = 14, int V = (LOGTA == 15 ? 2 : 1)>
int launch_a(u32 *const *cols, size_t n_cols, u32 n, u32 lo, const u32 *tw_end, u32 scale) {
    const size_t tiles = (size_t)1 << (n - LOGTA);
    return launch_fast_kernel(fast::k_cfft_a<INV, K, 0, LOGTA, V>, (1 << (LOGTA - 4)) / V,
                              ((size_t)(1 << LOGTA) + (1 << (LOGTA - 5)) + ((size_t)1 << K)) * sizeof(u32), tiles, cols, n_cols, n, lo, tw_end, scale);
}

// First forward pass of an evaluation whose input is a smaller polynomial (log size n - EXT) in its own buffers.
template <int K, int EXT, int LOGTA = 14>
int launch_a_ext(u32 *const *cols, const u32 *const *src, size_t n_cols, u32 n, u32 lo, const u32 *tw_end) {
    Context &c = ctx();
    const size_t tiles = (size_t)1 << (n - LOGTA);
    const size_t lds_bytes = ((size_t)(1 << LOGTA) + (1 << (LOGTA - 5)) + ((size_t)1 << K)) * sizeof(u32);
    auto kernel = fast::k_cfft_a<false, K, EXT, LOGTA>;
    { int rc_attr = allow_big_lds((const void *)kernel); if (rc_attr) return rc_attr; }
    {
        const size_t cnt = n_cols;
        ColPtrs cp, sp;
        int rc_tab = fill_col_table(cp, cols, n_cols, 0);
        if (!rc_tab) rc_tab = fill_col_table(sp, src, n_cols, 1);
        if (rc_tab) return rc_tab;
        const size_t items = tiles * cnt;
        if (items > 0xffffffffu) return set_error(TSTWO_ERR_BAD_ARG, "cfft: too many (tile, column) work items");
        const unsigned blocks = plan_grid((const void *)kernel, 1024, lds_bytes, items);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_bytes, c.stream, cp, sp, (u32)cnt, (u32)items, n, lo, tw_end, 0u);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
template <int EXT>
int launch_a_ext_k(u32 k, u32 logta, u32 *const *cols, const u32 *const *src, size_t n_cols, u32 n, u32 lo, const u32 *tw_end) {
    if (logta == 15) {
        if (k == 10) return launch_a_ext<10, EXT, 15>(cols, src, n_cols, n, lo, tw_end);
        if (k == 9) return launch_a_ext<9, EXT, 15>(cols, src, n_cols, n, lo, tw_end);
        if (k == 8) return launch_a_ext<8, EXT, 15>(cols, src, n_cols, n, lo, tw_end);
        return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
    }
    switch (k) {
        case 2: return launch_a_ext<2, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 3: return launch_a_ext<3, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 4: return launch_a_ext<4, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 5: return launch_a_ext<5, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 6: return launch_a_ext<6, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 7: return launch_a_ext<7, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 8: return launch_a_ext<8, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 9: return launch_a_ext<9, EXT>(cols, src, n_cols, n, lo, tw_end);
        case 10: return launch_a_ext<10, EXT>(cols, src, n_cols, n, lo, tw_end);
        default: return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
    }
}

template <bool INV>
int launch_fast(u32 *const *cols, size_t n_cols, u32 n, const Pass &ps, const u32 *tw_end, u32 scale) {
    if (ps.lo == 0) {
        const size_t tiles = (size_t)1 << (n - ps.k);
        const size_t lds = (((size_t)1 << ps.k) + ((size_t)1 << (ps.k - 5)) + ((size_t)1 << (ps.k - 4))) * sizeof(u32);
        switch (ps.k) {
            case 14: return launch_fast_kernel(fast::k_cfft_b<INV, 14>, 1024, lds, tiles, cols, n_cols, n, tw_end, scale);
            case 13:
#ifdef TSTWO_EXPERIMENTS      // TSTWO_CFFT_B8=1: the 8-words-per-lane bottom pass (1024 lanes, 8 waves per SIMD)
                if (knobs().cfft_b8)
                    return launch_fast_kernel(fast::k_cfft_b8<INV, 13>, 1024, (((size_t)1 << 13) + ((size_t)1 << 8) + ((size_t)1 << 10)) * sizeof(u32), tiles, cols,
                                              n_cols, n, tw_end, scale);
#endif
                return launch_fast_kernel(fast::k_cfft_b<INV, 13>, 512, lds, tiles, cols, n_cols, n, tw_end, scale);
            case 12: return launch_fast_kernel(fast::k_cfft_b<INV, 12>, 256, lds, tiles, cols, n_cols, n, tw_end, scale);
            case 11: return launch_fast_kernel(fast::k_cfft_b<INV, 11>, 128, lds, tiles, cols, n_cols, n, tw_end, scale);
            default: return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported bottom pass");
        }
    }
    const u32 logta = ps.c + ps.k;
    if (logta == 12) {          // small strided tiles (few columns): 256 lanes, rows of 2^(12-K) >= 16 words
        switch (ps.k) {
            case 1: return launch_a<INV, 1, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 2: return launch_a<INV, 2, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 3: return launch_a<INV, 3, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 4: return launch_a<INV, 4, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 5: return launch_a<INV, 5, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 6: return launch_a<INV, 6, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 7: return launch_a<INV, 7, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 8: return launch_a<INV, 8, 12>(cols, n_cols, n, ps.lo, tw_end, scale);
            default: return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
        }
    }
    if (logta == 13) {
        if (ps.k == 9) return launch_a<INV, 9, 13>(cols, n_cols, n, ps.lo, tw_end, scale);
        return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
    }
    if (logta == 15) {          // the 128 KiB tile (two virtual lanes per lane): 10 layers on 128-byte rows, 9 on 256-byte rows
        switch (ps.k) {
            case 8: return launch_a<INV, 8, 15>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 9: return launch_a<INV, 9, 15>(cols, n_cols, n, ps.lo, tw_end, scale);
            case 10: return launch_a<INV, 10, 15>(cols, n_cols, n, ps.lo, tw_end, scale);
            default: return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
        }
    }
    if (logta != 14) return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
#ifdef TSTWO_EXPERIMENTS      // TSTWO_CFFT_AV=2: the 2^14 tile on 512 lanes x 32 words, two workgroups per CU (measured slower: 3.86 against 3.75 ms)
    if (knobs().cfft_av == 2 && (ps.k == 8 || ps.k == 9))
        return ps.k == 9 ? launch_a<INV, 9, 14, 2>(cols, n_cols, n, ps.lo, tw_end, scale) : launch_a<INV, 8, 14, 2>(cols, n_cols, n, ps.lo, tw_end, scale);
#endif
    switch (ps.k) {
        case 1: return launch_a<INV, 1>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 2: return launch_a<INV, 2>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 3: return launch_a<INV, 3>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 4: return launch_a<INV, 4>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 5: return launch_a<INV, 5>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 6: return launch_a<INV, 6>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 7: return launch_a<INV, 7>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 8: return launch_a<INV, 8>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 9: return launch_a<INV, 9>(cols, n_cols, n, ps.lo, tw_end, scale);
        case 10: return launch_a<INV, 10>(cols, n_cols, n, ps.lo, tw_end, scale);
        default: return set_error(TSTWO_ERR_BAD_ARG, "cfft: unsupported pass shape");
    }
}

// bottom tile, strided-pass layer limit and strided tile of an in-place transform of n_cols columns of 2^n words (n >= 13)
void choose_plan(u32 n, size_t n_cols, u32 &kb, u32 &ka_max, u32 &logta) {
    Context &c = ctx();
    const PlanShape shape = default_shape(n, n_cols);
    kb = shape.kb; ka_max = shape.ka_max;
    const Knobs &kn = knobs();
    if (kn.cfft_kb) kb = (u32)kn.cfft_kb;           // experiments build only: bottom-pass size, strided-pass limit
    if (kn.cfft_ka) ka_max = (u32)kn.cfft_ka;
    logta = shape.logta;
    if (n > kb && n >= kLogTileA && !kn.cfft_kb && !kn.cfft_ka) {     // n = 13 (and 14) is one bottom pass whatever the column count
        // Few columns: the default tiles (2^13 contiguous, 2^14 strided) give 2^(n-13) x cols and 2^(n-14) x cols workgroups;
        // below ~2 per CU pick the split with the most workgroups in its emptier pass (ties: the larger tiles).
        // n = 20, one column: 12 + 8 layers on 2^12-word tiles = 256 + 256 workgroups instead of 128 + 64.
        const size_t want = (size_t)2 * (size_t)c.n_cus;
        if ((((size_t)1 << (n - kLogTileA)) * n_cols) < want && n - 13 <= ka_max) {
            size_t best = 0;
            for (u32 tb = 13; tb >= 11; tb--) {
                const u32 k = n - tb;
                if (k < 1 || k > ka_max) continue;        // one strided pass
                const u32 ta = k <= 8 ? 12u : (k == 9 ? 13u : 14u);
                if (ta > n) continue;
                size_t wa = ((size_t)1 << (n - ta)) * n_cols, wb = ((size_t)1 << (n - tb)) * n_cols;
                size_t score = wa < wb ? wa : wb;
                if (score > want) score = want;
                if (score > best) { best = score; kb = tb; logta = ta; }
            }
        }
    }
    if (kn.cfft_logta) logta = (u32)kn.cfft_logta;   // experiments
}

template <bool INV>
int cfft(u32 *const *cols, size_t n_cols, u32 n, u32 half_initial, const u32 *tw, u32 tw_log) {
    TSTWO_REQUIRE_READY();
    if (n == 0 || n > 31) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size out of range");
    if (n > kMaxLogSize) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size > 30 exceeds MAX_CIRCLE_DOMAIN_LOG_SIZE (poly/circle/domain.ts:4)");
    if (n_cols == 0) return TSTWO_OK;
    if (!cols) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null column table");
    Context &c = ctx();
    TSTWO_REQUIRE_TABLE(cols, n_cols);
    const u32 N = 1u << n;
    const u32 n_inv = host::inv(N % M31_P);
    if (n <= 2) {
        u32 px, py;
        host::point(half_initial, &px, &py);
        u32 tx = px, ty = py, scale = 1;
        if (INV) {
            if (py == 0 || (n == 2 && px == 0)) return set_error(TSTWO_ERR_ZERO_INVERSE, "0 has no inverse");
            ty = host::inv(py);
            tx = n == 2 ? host::inv(px) : 0;
            scale = n_inv;
        }
        {
            ColPtrs cp;
            int rc_tab = fill_col_table(cp, cols, n_cols, 0);
            if (rc_tab) return rc_tab;
            hipLaunchKernelGGL(k_cfft_small<INV>, dim3((unsigned)((n_cols + 63) / 64)), dim3(64), 0, c.stream, cp, (u32)n_cols, n, tx, ty, scale);
        }
        TSTWO_LAUNCH_CHECK();
        return TSTWO_OK;
    }
    if (!tw) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null twiddle buffer");
    if (tw_log > 31 || ((size_t)1 << (n - 1)) > ((size_t)1 << tw_log)) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    Pass passes[8];
    u32 kb, ka_max, logta;
    choose_plan(n, n_cols, kb, ka_max, logta);
    const Knobs &kn = knobs();
    int np = n >= kMaxLogTileB ? plan_passes(n, passes, kb, ka_max, logta) : plan_passes(n, passes);
    for (size_t i = 0; i < n_cols; i++)
        if (((uintptr_t)cols[i]) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: columns must be 16-byte aligned");
    if (((uintptr_t)tw) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: twiddle buffer must be 16-byte aligned");
    const u32 *tw_end = tw + ((size_t)1 << tw_log);
    const bool fast_path = n >= kMaxLogTileB && n <= kMaxLogSize;
    const int dbg_generic = kn.cfft_generic;      // experiments build only: 1 = generic kernel for the bottom pass, 2 = for strided passes, 4 = skip the bottom pass
    // All columns go through a pass in one launch.  (Measured on MI355X, 32 x 2^22: running the passes back to back
    // on Infinity-Cache-sized column groups is slower — 688 us ungrouped vs 724/771/879 us for groups of 16/8/4 —
    // the extra launch tails cost more than MALL residency of the intermediate returns.  TSTWO_CFFT_GROUP=k re-enables it.)
    size_t group = n_cols;
    if (np > 1 && kn.cfft_group > 0 && (size_t)kn.cfft_group < n_cols) group = (size_t)kn.cfft_group;
    for (size_t g0 = 0; g0 < n_cols; g0 += group) {
        const size_t gc = n_cols - g0 < group ? n_cols - g0 : group;
        u32 *const *gcols = cols + g0;
        if (!INV) {
            for (int s = np - 1; s >= 0; s--) {
                if ((dbg_generic & 4) && passes[s].lo == 0) continue;
                const bool f = fast_path && !(dbg_generic & (passes[s].lo == 0 ? 1 : 2));
                int rc = f ? launch_fast<false>(gcols, gc, n, passes[s], tw_end, 0)
                           : launch_pass<false>(gcols, gc, n, passes[s], tw, tw_log, 0);
                if (rc) return rc;
            }
        } else {
            for (int s = 0; s < np; s++) {
                const u32 sc = s == np - 1 ? n_inv : 0;
                const bool f = fast_path && !(dbg_generic & (passes[s].lo == 0 ? 1 : 2));
                int rc = f ? launch_fast<true>(gcols, gc, n, passes[s], tw_end, sc)
                           : launch_pass<true>(gcols, gc, n, passes[s], tw, tw_log, sc);
                if (rc) return rc;
            }
        }
    }
    return TSTWO_OK;
}

}  // namespace

extern "C" {

int tstwo_cfft_evaluate(u32 *const *cols, size_t n_cols, u32 log_size, u32 half_initial, const u32 *tw, u32 tw_log) {
    return cfft<false>(cols, n_cols, log_size, half_initial, tw, tw_log);
}
int tstwo_cfft_interpolate(u32 *const *cols, size_t n_cols, u32 log_size, u32 half_initial, const u32 *itw, u32 tw_log) {
    return cfft<true>(cols, n_cols, log_size, half_initial, itw, tw_log);
}

int tstwo_cfft_plan_passes(u32 log_size, size_t n_cols, u32 *n_passes) {
    TSTWO_REQUIRE_READY();          // (the planner asks the context for the CU count)
    if (!n_passes) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null output");
    if (log_size == 0 || log_size > kMaxLogSize) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size out of range");
    if (log_size < kMaxLogTileB) { *n_passes = 1; return TSTWO_OK; }
    Pass passes[8];
    u32 kb, ka_max, logta;
    choose_plan(log_size, n_cols, kb, ka_max, logta);
    *n_passes = (u32)plan_passes(log_size, passes, kb, ka_max, logta);
    return TSTWO_OK;
}

// Out-of-place interpolation: src[i] (evaluations, left untouched) -> dst[i] (coefficients).  On the tiled path the first
// (bottom) pass reads src and writes dst, so the clone the value-semantics API needs costs no extra pass over HBM.
int tstwo_cfft_interpolate_to(const u32 *const *src, u32 *const *dst, size_t n_cols, u32 log_size, u32 half_initial, const u32 *itw, u32 tw_log) {
    TSTWO_REQUIRE_READY();
    if (n_cols == 0) return TSTWO_OK;
    if (!src || !dst) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null column table");
    TSTWO_REQUIRE_TABLE(src, n_cols); TSTWO_REQUIRE_TABLE(dst, n_cols);
    if (log_size == 0 || log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size out of range");
    const Knobs &kn = knobs();
    const bool tiled = log_size >= kMaxLogTileB && log_size <= kMaxLogSize && !kn.cfft_generic && !kn.cfft_kb && !kn.cfft_ka && !kn.cfft_no_oop;
    if (tiled) {
        if (!itw) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null twiddle buffer");
        if (tw_log > 31 || ((size_t)1 << (log_size - 1)) > ((size_t)1 << tw_log)) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
        for (size_t i = 0; i < n_cols; i++)
            if ((((uintptr_t)dst[i]) & 15) || (((uintptr_t)src[i]) & 15)) return set_error(TSTWO_ERR_BAD_ARG, "cfft: columns must be 16-byte aligned");
        if (((uintptr_t)itw) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: twiddle buffer must be 16-byte aligned");
        Context &c = ctx();
        Pass passes[8];
        const int np = plan_default(log_size, n_cols, passes);
        const u32 kb = passes[0].k;
        const u32 *tw_end = itw + ((size_t)1 << tw_log);
        const u32 n_inv = host::inv((1u << log_size) % M31_P);
        // bottom pass, out of place
        {
            const size_t tiles = (size_t)1 << (log_size - kb);
            const size_t lds = (((size_t)1 << kb) + ((size_t)1 << (kb - 5)) + ((size_t)1 << (kb - 4))) * sizeof(u32);
            const unsigned threads = 1u << (kb - 4);
            auto kernel = kb == 14 ? fast::k_cfft_b<true, 14, true> : fast::k_cfft_b<true, 13, true>;
            { int rc_attr = allow_big_lds((const void *)kernel); if (rc_attr) return rc_attr; }
            {
                const size_t cnt = n_cols;
                ColPtrs cp, sp;
                int rc_tab = fill_col_table(cp, dst, n_cols, 0);
                if (!rc_tab) rc_tab = fill_col_table(sp, src, n_cols, 1);
                if (rc_tab) return rc_tab;
                const size_t items = tiles * cnt;
                if (items > 0xffffffffu) return set_error(TSTWO_ERR_BAD_ARG, "cfft: too many (tile, column) work items");
                const unsigned blocks = plan_grid((const void *)kernel, (int)threads, lds, items);
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, c.stream, cp, sp, (u32)cnt, (u32)items, log_size, tw_end,
                                   np == 1 ? n_inv : 0u);
            }
            TSTWO_LAUNCH_CHECK();
        }
        for (int s = 1; s < np; s++) {
            int rc = launch_fast<true>(dst, n_cols, log_size, passes[s], tw_end, s == np - 1 ? n_inv : 0u);
            if (rc) return rc;
        }
        return TSTWO_OK;
    }
    for (size_t i = 0; i < n_cols; i++) {
        int rc = tstwo_copy(dst[i], src[i], (size_t)4 << log_size);
        if (rc) return rc;
    }
    return cfft<true>(dst, n_cols, log_size, half_initial, itw, tw_log);
}

// CirclePoly.extend + evaluate (backend/cpu/circle.ts:71-134) in one call: polys[i] holds 2^log_poly coefficients,
// out[i] receives the 2^log_size evaluations.  When the extension is by 1 or 2 bits and the transform takes the tiled
// path, the zero padding is never materialised: the first pass reads the small polynomial and skips the replicating
// layers.  Otherwise: tstwo_poly_extend into out[i], then the in-place transform.
int tstwo_cfft_evaluate_extended(const u32 *const *polys, u32 log_poly, u32 *const *out, size_t n_cols, u32 log_size, u32 half_initial,
                                 const u32 *tw, u32 tw_log) {
    TSTWO_REQUIRE_READY();
    if (n_cols == 0) return TSTWO_OK;
    if (!polys || !out) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null column table");
    TSTWO_REQUIRE_TABLE(polys, n_cols); TSTWO_REQUIRE_TABLE(out, n_cols);
    if (log_size < log_poly) return set_error(TSTWO_ERR_LOG_SIZE, "log size too small");
    if (log_size == 0 || log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size out of range");
    const u32 ext = log_size - log_poly;
    Pass passes[8];
    int np = 0;
    const Knobs &kn = knobs();
    const bool tiled = log_size >= kMaxLogTileB && log_size <= kMaxLogSize && !kn.cfft_generic && !kn.cfft_kb && !kn.cfft_ka && !kn.cfft_no_fused_extend;
    if (tiled) np = log_size == 14 ? plan_passes(log_size, passes, kMaxLogTileB) : plan_default(log_size, n_cols, passes);   // (n = 14: 13 + 1 keeps the fused extension)
    if (tiled && np >= 2 && (ext == 1 || ext == 2) && passes[np - 1].k >= 2) {
        if (!tw) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null twiddle buffer");
        if (tw_log > 31 || ((size_t)1 << (log_size - 1)) > ((size_t)1 << tw_log)) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
        for (size_t i = 0; i < n_cols; i++)
            if ((((uintptr_t)out[i]) & 15) || (((uintptr_t)polys[i]) & 15)) return set_error(TSTWO_ERR_BAD_ARG, "cfft: columns must be 16-byte aligned");
        if (((uintptr_t)tw) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: twiddle buffer must be 16-byte aligned");
        const u32 *tw_end = tw + ((size_t)1 << tw_log);
        const Pass &top = passes[np - 1];
        int rc = ext == 1 ? launch_a_ext_k<1>(top.k, top.c + top.k, out, polys, n_cols, log_size, top.lo, tw_end)
                          : launch_a_ext_k<2>(top.k, top.c + top.k, out, polys, n_cols, log_size, top.lo, tw_end);
        if (rc) return rc;
        for (int s = np - 2; s >= 0; s--) {
            rc = launch_fast<false>(out, n_cols, log_size, passes[s], tw_end, 0);
            if (rc) return rc;
        }
        return TSTWO_OK;
    }
    for (size_t i = 0; i < n_cols; i++) {
        int rc = tstwo_poly_extend(polys[i], log_poly, out[i], log_size);
        if (rc) return rc;
    }
    return cfft<false>(out, n_cols, log_size, half_initial, tw, tw_log);
}

}  // extern "C"
