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
#include "common.h"
#include "host_field.h"

using namespace tstwo;

namespace {

constexpr int kThreads = 256;
constexpr u32 kMaxLogTileB = 13;   // contiguous (bottom) tile: 2^13 words = 32 KiB + pad
constexpr u32 kLogTileA = 14;      // strided tile: 2^k rows x 2^(14-k) words = 64 KiB + pad
constexpr u32 kMaxKA = 9;          // at most 9 layers per strided pass (rows of >= 32 words = 128 B)

struct PassParams {
    u32 n;        // log size of the column
    u32 lo;       // lowest layer of this pass
    u32 k;        // layers [lo, lo+k)
    u32 c;        // log2(words per row); 0 for the bottom pass (lo == 0, one contiguous row)
    u32 logt;     // c + k = log2(tile words)
    u32 scale;    // interpolate's 2^-n, applied by the last pass; 0 = no scaling
    const u32 *tw_end;  // tree + L
};

__device__ __forceinline__ u32 phys(u32 e) { return e + (e >> 5); }

// One register stage: G consecutive layers on LDS bits [q, q+G).
// CIRCLE: bit 0 of this stage is the circle layer (only when q == 0 and lo == 0; needs G >= 3).
template <int G, bool INV, bool CIRCLE>
__device__ __forceinline__ void run_stage(u32 *lds, const u32 q, const PassParams &pp, const u32 hi) {
    const u32 ngroups = 1u << (pp.logt - G);
    for (u32 gid = threadIdx.x; gid < ngroups; gid += kThreads) {
        const u32 low = gid & ((1u << q) - 1u), high = gid >> q;
        const u32 e0 = (high << (q + G)) | low;
        u32 v[1 << G];
#pragma unroll
        for (int m = 0; m < (1 << G); m++) v[m] = lds[phys(e0 + ((u32)m << q))];

        u32 tw1[(G >= 2) ? (1 << (G - 2)) : 1];   // layer-1 twiddles of the group, shared with the circle layer
        (void)tw1;
#pragma unroll
        for (int step = 0; step < G; step++) {
            const int l = INV ? step : (G - 1 - step);
            const u32 b = q + (u32)l;                       // LDS bit of this layer
            if (CIRCLE && l == 0) {
                if (INV) {                                  // inverse runs the circle layer first: fetch tw1 now
                    const u32 *seg1 = pp.tw_end - (1u << (pp.n - 1));
                    const u32 hb1 = (hi << (pp.logt - 2)) | (high << (G - 2));
#pragma unroll
                    for (int j = 0; j < (1 << (G - 2)); j++) tw1[j] = seg1[hb1 + j];
                }
#pragma unroll
                for (int j = 0; j < (1 << (G - 1)); j++) {
                    u32 t = tw1[(j >> 1) ^ 1];
                    if ((j ^ (j >> 1)) & 1) t = m31_neg(t);
                    if (INV) m31_ibutterfly(v[2 * j], v[2 * j + 1], t);
                    else m31_butterfly(v[2 * j], v[2 * j + 1], t);
                }
            } else {
                const u32 i = pp.lo + (b - pp.c);           // global layer
                const u32 *seg = pp.tw_end - (1u << (pp.n - i));
                const u32 hb = (hi << (pp.logt - 1 - b)) | (high << (G - 1 - l));
#pragma unroll
                for (int j = 0; j < (1 << (G - 1 - l)); j++) {
                    u32 t;
                    if (CIRCLE && l == 1 && INV) t = tw1[j];
                    else t = seg[hb + j];
                    if (CIRCLE && l == 1 && !INV) tw1[j] = t;
#pragma unroll
                    for (int r = 0; r < (1 << l); r++) {
                        const int m0 = (j << (l + 1)) | r;
                        if (INV) m31_ibutterfly(v[m0], v[m0 + (1 << l)], t);
                        else m31_butterfly(v[m0], v[m0 + (1 << l)], t);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << G); m++) lds[phys(e0 + ((u32)m << q))] = v[m];
    }
}

template <bool INV, bool CIRCLE>
__device__ __forceinline__ void dispatch_stage(int g, u32 *lds, u32 q, const PassParams &pp, u32 hi) {
    switch (g) {
        case 5: run_stage<5, INV, CIRCLE>(lds, q, pp, hi); break;
        case 4: run_stage<4, INV, CIRCLE>(lds, q, pp, hi); break;
        case 3: run_stage<3, INV, CIRCLE>(lds, q, pp, hi); break;
        case 2: if (!CIRCLE) run_stage<2, INV, false>(lds, q, pp, hi); break;
        case 1: if (!CIRCLE) run_stage<1, INV, false>(lds, q, pp, hi); break;
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

template <bool INV>
__global__ void __launch_bounds__(kThreads) k_cfft_pass(ColPtrs cols, u32 n_cols, PassParams pp) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    // column is the fastest-varying block coordinate: blocks sharing a tile (hence twiddles) are adjacent
    const u32 col = blockIdx.x % n_cols;
    const u32 tile = blockIdx.x / n_cols;
    u32 *__restrict__ data = cols.p[col];

    const u32 mid_bits = pp.lo - pp.c;                     // 0 for the bottom pass
    const u32 mid = tile & ((1u << mid_bits) - 1u);
    const u32 hi = tile >> mid_bits;
    const size_t base = ((size_t)hi << (pp.lo + pp.k)) | ((size_t)mid << pp.c);
    const u32 tile_words = 1u << pp.logt;
    const u32 cmask = (1u << pp.c) - 1u;

    // ---- global -> LDS, 16 bytes per lane
    if (pp.logt >= 2) {
        for (u32 e = 4u * threadIdx.x; e < tile_words; e += 4u * kThreads) {
            const size_t g = base + ((size_t)(e >> pp.c) << pp.lo) + (e & cmask);
            const uint4 x = *reinterpret_cast<const uint4 *>(data + g);
            const u32 p = phys(e);
            lds[p] = x.x; lds[p + 1] = x.y; lds[p + 2] = x.z; lds[p + 3] = x.w;
        }
    }
    __syncthreads();

    int gs[8];
    const int n_stages = plan_stages(pp, gs);
    if (!INV) {
        u32 q = pp.logt;
        for (int s = n_stages - 1; s >= 0; s--) {
            q -= (u32)gs[s];
            if (pp.lo == 0 && q == 0) dispatch_stage<false, true>(gs[s], lds, q, pp, hi);
            else dispatch_stage<false, false>(gs[s], lds, q, pp, hi);
            __syncthreads();
        }
    } else {
        u32 q = pp.c;
        for (int s = 0; s < n_stages; s++) {
            if (pp.lo == 0 && q == 0) dispatch_stage<true, true>(gs[s], lds, q, pp, hi);
            else dispatch_stage<true, false>(gs[s], lds, q, pp, hi);
            q += (u32)gs[s];
            __syncthreads();
        }
    }

    // ---- LDS -> global (fused 2^-n scaling on interpolate's last pass)
    for (u32 e = 4u * threadIdx.x; e < tile_words; e += 4u * kThreads) {
        const size_t g = base + ((size_t)(e >> pp.c) << pp.lo) + (e & cmask);
        const u32 p = phys(e);
        uint4 x = make_uint4(lds[p], lds[p + 1], lds[p + 2], lds[p + 3]);
        if (INV && pp.scale) {
            x.x = m31_mul(x.x, pp.scale); x.y = m31_mul(x.y, pp.scale);
            x.z = m31_mul(x.z, pp.scale); x.w = m31_mul(x.w, pp.scale);
        }
        *reinterpret_cast<uint4 *>(data + g) = x;
    }
}

// log_size 1 and 2 (backend/cpu/circle.ts:93-110,153-185): twiddles are coordinates of the half
// coset's initial point, supplied by the host.  tx/ty (and the scale) are already inverted for INV.
template <bool INV>
__global__ void k_cfft_small(ColPtrs cols, u32 n_cols, u32 n, u32 tx, u32 ty, u32 scale) {
    u32 col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    u32 *v = cols.p[col];
    if (n == 1) {
        u32 v0 = v[0], v1 = v[1];
        if (!INV) m31_butterfly(v0, v1, ty);
        else { m31_ibutterfly(v0, v1, ty); v0 = m31_mul(v0, scale); v1 = m31_mul(v1, scale); }
        v[0] = v0; v[1] = v1;
    } else {
        u32 v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
        if (!INV) {
            m31_butterfly(v0, v2, tx); m31_butterfly(v1, v3, tx);
            m31_butterfly(v0, v1, ty); m31_butterfly(v2, v3, m31_neg(ty));
        } else {
            m31_ibutterfly(v0, v1, ty); m31_ibutterfly(v2, v3, m31_neg(ty));
            m31_ibutterfly(v0, v2, tx); m31_ibutterfly(v1, v3, tx);
            v0 = m31_mul(v0, scale); v1 = m31_mul(v1, scale); v2 = m31_mul(v2, scale); v3 = m31_mul(v3, scale);
        }
        v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
    }
}

struct Pass { u32 lo, k, c; };
// passes low -> high
int plan_passes(u32 n, Pass *out) {
    int cnt = 0;
    if (n <= kMaxLogTileB) {
        out[cnt++] = {0, n, 0};
        return cnt;
    }
    out[cnt++] = {0, kMaxLogTileB, 0};
    u32 rem = n - kMaxLogTileB, lo = kMaxLogTileB;
    u32 np = (rem + kMaxKA - 1) / kMaxKA;
    u32 base = rem / np, extra = rem % np;
    for (u32 s = 0; s < np; s++) {
        u32 k = base + (s < extra ? 1 : 0);
        u32 c = kLogTileA - k;
        if (c > lo) c = lo;
        out[cnt++] = {lo, k, c};
        lo += k;
    }
    return cnt;
}

template <bool INV>
int launch_pass(u32 *const *cols, size_t n_cols, u32 n, const Pass &ps, const u32 *tw, u32 tw_log, u32 scale) {
    Context &c = ctx();
    PassParams pp;
    pp.n = n; pp.lo = ps.lo; pp.k = ps.k; pp.c = ps.c; pp.logt = ps.c + ps.k; pp.scale = scale;
    pp.tw_end = tw + ((size_t)1 << tw_log);
    size_t tiles = (size_t)1 << (n - pp.logt);
    size_t lds_bytes = (((size_t)1 << pp.logt) + (((size_t)1 << pp.logt) >> 5) + 4) * sizeof(u32);
    static bool lds_attr_set = false;   // tiles above 64 KiB need the opt-in (160 KiB LDS per CU on gfx950)
    if (!lds_attr_set) {
        TSTWO_HIP(hipFuncSetAttribute((const void *)k_cfft_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TSTWO_HIP(hipFuncSetAttribute((const void *)k_cfft_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_attr_set = true;
    }
    for (size_t i = 0; i < n_cols; i++)
        if (((uintptr_t)cols[i]) & 15) return set_error(TSTWO_ERR_BAD_ARG, "cfft: columns must be 16-byte aligned");
    for (size_t b0 = 0; b0 < n_cols; b0 += kMaxColsPerLaunch) {
        size_t cnt = n_cols - b0 < (size_t)kMaxColsPerLaunch ? n_cols - b0 : (size_t)kMaxColsPerLaunch;
        ColPtrs cp;
        for (size_t i = 0; i < cnt; i++) cp.p[i] = cols[b0 + i];
        size_t blocks = tiles * cnt;
        if (blocks > 0x7fffffffu) return set_error(TSTWO_ERR_BAD_ARG, "cfft: grid too large");
        hipLaunchKernelGGL(k_cfft_pass<INV>, dim3((unsigned)blocks), dim3(kThreads), lds_bytes, c.stream, cp, (u32)cnt, pp);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

template <bool INV>
int cfft(u32 *const *cols, size_t n_cols, u32 n, u32 half_initial, const u32 *tw, u32 tw_log) {
    TSTWO_REQUIRE_READY();
    if (n == 0 || n > 31) return set_error(TSTWO_ERR_BAD_ARG, "cfft: log_size out of range");
    if (n_cols == 0) return TSTWO_OK;
    if (!cols) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null column table");
    Context &c = ctx();
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
        for (size_t b0 = 0; b0 < n_cols; b0 += kMaxColsPerLaunch) {
            size_t cnt = n_cols - b0 < (size_t)kMaxColsPerLaunch ? n_cols - b0 : (size_t)kMaxColsPerLaunch;
            ColPtrs cp;
            for (size_t i = 0; i < cnt; i++) cp.p[i] = cols[b0 + i];
            hipLaunchKernelGGL(k_cfft_small<INV>, dim3(1), dim3(64), 0, c.stream, cp, (u32)cnt, n, tx, ty, scale);
        }
        TSTWO_LAUNCH_CHECK();
        return TSTWO_OK;
    }
    if (!tw) return set_error(TSTWO_ERR_BAD_ARG, "cfft: null twiddle buffer");
    if (tw_log > 31 || ((size_t)1 << (n - 1)) > ((size_t)1 << tw_log)) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
    Pass passes[8];
    int np = plan_passes(n, passes);
    if (!INV) {
        for (int s = np - 1; s >= 0; s--) {
            int rc = launch_pass<false>(cols, n_cols, n, passes[s], tw, tw_log, 0);
            if (rc) return rc;
        }
    } else {
        for (int s = 0; s < np; s++) {
            int rc = launch_pass<true>(cols, n_cols, n, passes[s], tw, tw_log, s == np - 1 ? n_inv : 0);
            if (rc) return rc;
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

}  // extern "C"
