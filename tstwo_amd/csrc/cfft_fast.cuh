// cfft_fast.cuh — the specialised Circle-FFT pass kernels used for log_size >= 13 (included by cfft.hip).
//
// Geometry is compile-time so that addressing folds into immediates and the register budget stays
// near 64 VGPRs (8 waves/SIMD): every lane owns 16 words of the tile.
//   k_cfft_b<INV, LOGT> bottom pass: layers 0..LOGT-1 on a contiguous 2^LOGT-word tile (LOGT = 13: 512 lanes).
//   k_cfft_a<INV, K, EXT, LOGT> strided pass: K layers [lo, lo+K) on a tile of 2^K rows x 2^(LOGT-K) words, 2^(LOGT-4) lanes (LOGT = 14: 1024).
// Structure of one tile (forward; the inverse mirrors it):
//   1. the tile arrives as four 16-byte loads per lane a quarter-tile apart, so the pass's two top layers
//      are a radix-4 butterfly in registers before anything touches LDS;
//   2. the middle layers run as LDS radix-8/16 stages (compile-time strides, twiddles from a small LDS heap
//      that is staged once per workgroup and reused for every column the workgroup owns);
//   3. the last stage leaves each lane with its final 16 words and stores them straight to HBM.
//   The next column's tile is prefetched into registers while the current one is transformed; barriers
//   drain LDS only (lds_barrier), so global loads/stores stay in flight across them.
// LDS layout: word e lives at e + (e >> 5) (one pad word per 32).  Every access pattern used here maps the
// 32 lanes of a half-wave onto 32 distinct banks (4-word runs at lane stride 4, 16-word runs at lane stride
// 16, unit-stride runs for stage bits >= 5; the bit-4 stage assigns groups to lanes with bits 4/5 of the
// lane id swapped), and the address of word e0 + (m << Q) of a register group is pad(e0) + constant(m), so
// one address register and immediate offsets serve a whole group.
#pragma once

namespace fast {

__device__ __forceinline__ u32 pad(u32 e) { return e + (e >> 5); }
// offset of word (m << Q) relative to pad(e0) when e0 has zeros in the group's bit range (no carries: see DESIGN.md)
template <int Q>
__device__ __forceinline__ constexpr u32 off(int m) { return ((u32)m << Q) + (((u32)m << Q) >> 5); }

template <bool INV>
__device__ __forceinline__ void bfly(u32 &a, u32 &b, u32 t2) {
    if (INV) ibf_dbl(a, b, t2);
    else bf_dbl(a, b, t2);
}

// G layers on LDS bits [Q, Q+G) of NG register groups of 2^G words each (NG << G = 16 words per lane); twiddles from the
// heap: layer bit b lives at twl[2^(LOGT-1-b) + (e >> (b+1))].  Every layer is ONE bf_layer of 8 butterflies (priority
// phases: cfft.hip); returns at kPrioLight.
template <int G, int NG, int Q, int LOGT, bool INV>
__device__ __forceinline__ void groups_layers(u32 (&v)[NG][1 << G], const u32 *twl, const u32 (&high)[NG]) {
    constexpr int HALF = 1 << (G - 1), N = NG * HALF;
#pragma unroll
    for (int step = 0; step < G; step++) {
        const int l = INV ? step : (G - 1 - step);
        u32 x[N], y[N], tw[N];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const u32 *tl = twl + (1u << (LOGT - 1 - (Q + l))) + (high[g] << (G - 1 - l));
#pragma unroll
            for (int j = 0; j < (1 << (G - 1 - l)); j++) {
                const u32 t2 = tl[j];
#pragma unroll
                for (int r = 0; r < (1 << l); r++) {
                    const int m0 = (j << (l + 1)) | r, i = g * HALF + (j << l) + r;
                    x[i] = v[g][m0]; y[i] = v[g][m0 + (1 << l)]; tw[i] = t2;
                }
            }
        }
        bf_layer<INV, N>(x, y, tw);
        if (step == G - 1) phase<kPrioLight>(x, y);
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int j = 0; j < (1 << (G - 1 - l)); j++)
#pragma unroll
                for (int r = 0; r < (1 << l); r++) {
                    const int m0 = (j << (l + 1)) | r, i = g * HALF + (j << l) + r;
                    v[g][m0] = x[i]; v[g][m0 + (1 << l)] = y[i];
                }
    }
}

// In-place LDS stage: every lane handles 16 >> G groups of 2^G words.
template <int G, int Q, int LOGT, int THREADS, bool INV, int WPL = 16>
__device__ __forceinline__ void lds_stage(u32 *lds, const u32 *twl, u32 tid) {
    constexpr int NG = WPL >> G;
    u32 v[NG][1 << G], high[NG];
    u32 *p[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        u32 gid = tid + (u32)g * THREADS;
        if (Q == 4) gid = (gid & ~0x30u) | ((gid & 0x10u) << 1) | ((gid & 0x20u) >> 1);   // bank-conflict-free lane -> group map
        // (Q = 3, the 8-words-per-lane kernel: the word address is 33 high + low + const, so 32 lanes need distinct high + low mod 32:
        // lane bits 3, 4 go to high bits 3, 4 and lane bits 5..7 to high bits 0..2 — banks 8 a + low + const)
        if (Q == 3) gid = (gid & ~0xF8u) | (((gid >> 3) & 3u) << 6) | (((gid >> 5) & 7u) << 3);
        const u32 low = gid & ((1u << Q) - 1u);
        high[g] = gid >> Q;
        p[g] = lds + pad((high[g] << (Q + G)) | low);
#pragma unroll
        for (int m = 0; m < (1 << G); m++) v[g][m] = p[g][off<Q>(m)];
    }
    groups_layers<G, NG, Q, LOGT, INV>(v, twl, high);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int m = 0; m < (1 << G); m++) p[g][off<Q>(m)] = v[g][m];
}

// The pass's two top layers on the four quarter-tile vectors of a lane (component-wise radix-4).
// x[j], j = (top bit, second bit).  ta: top-layer twiddle; tb0/tb1: second-layer twiddles of the two halves.
// (a, c), (b, d) with ta — the pass's top layer; leaves the wave at kPrioHeavy like bf_layer unless LAST
template <bool INV, bool LAST>
__device__ __forceinline__ void quarter_layer_top(uint4 (&q)[4], u32 ta) {
    u32 *p0 = reinterpret_cast<u32 *>(&q[0]), *p1 = reinterpret_cast<u32 *>(&q[1]);
    u32 *p2 = reinterpret_cast<u32 *>(&q[2]), *p3 = reinterpret_cast<u32 *>(&q[3]);
    u32 x[8], y[8], tw[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = p0[k]; y[k] = p2[k]; x[4 + k] = p1[k]; y[4 + k] = p3[k]; tw[k] = ta; tw[4 + k] = ta; }
    bf_layer<INV, 8>(x, y, tw);
    if (LAST) phase<kPrioLight>(x, y);
#pragma unroll
    for (int k = 0; k < 4; k++) { p0[k] = x[k]; p2[k] = y[k]; p1[k] = x[4 + k]; p3[k] = y[4 + k]; }
}
// (a, b) with tb0, (c, d) with tb1 — the second layer
template <bool INV, bool LAST>
__device__ __forceinline__ void quarter_layer_second(uint4 (&q)[4], u32 tb0, u32 tb1) {
    u32 *p0 = reinterpret_cast<u32 *>(&q[0]), *p1 = reinterpret_cast<u32 *>(&q[1]);
    u32 *p2 = reinterpret_cast<u32 *>(&q[2]), *p3 = reinterpret_cast<u32 *>(&q[3]);
    u32 x[8], y[8], tw[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { x[k] = p0[k]; y[k] = p1[k]; x[4 + k] = p2[k]; y[4 + k] = p3[k]; tw[k] = tb0; tw[4 + k] = tb1; }
    bf_layer<INV, 8>(x, y, tw);
    if (LAST) phase<kPrioLight>(x, y);
#pragma unroll
    for (int k = 0; k < 4; k++) { p0[k] = x[k]; p1[k] = y[k]; p2[k] = x[4 + k]; p3[k] = y[4 + k]; }
}
template <bool INV, bool TWO>
__device__ __forceinline__ void top_layers(uint4 (&q)[4], u32 ta, u32 tb0, u32 tb1) {
    if (!INV) {
        quarter_layer_top<false, !TWO>(q, ta);
        if (TWO) quarter_layer_second<false, true>(q, tb0, tb1);
    } else {
        if (TWO) quarter_layer_second<true, false>(q, tb0, tb1);
        quarter_layer_top<true, true>(q, ta);
    }
}

// The four lowest layers of the bottom pass on a lane's 16 consecutive words: layer 3 (t3), layer 2 (t2[2]), layer 1
// (t1[4]) and the circle layer (t1 again, permuted and negated: backend/cpu/circle.ts:153-185), 8 butterflies each.
template <bool INV>
__device__ __forceinline__ void low_layers(u32 (&v)[16], const u32 (&t1)[4], const u32 (&t2)[2], u32 t3) {
    u32 tc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        tc[j] = t1[(j >> 1) ^ 1];
        if ((j ^ (j >> 1)) & 1) tc[j] = 0xFFFFFFFEu - tc[j];
    }
#pragma unroll
    for (int step = 0; step < 4; step++) {
        const int layer = INV ? step : 3 - step;          // 0 = circle layer
        const int l = layer;                              // partner distance 2^l (circle layer: adjacent words)
        u32 x[8], y[8], tw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = ((i >> l) << (l + 1)) | (i & ((1 << l) - 1));       // the i-th word with bit l clear
            x[i] = v[m]; y[i] = v[m + (1 << l)];
            tw[i] = layer == 0 ? tc[m >> 1] : layer == 1 ? t1[m >> 2] : layer == 2 ? t2[m >> 3] : t3;
        }
        bf_layer<INV, 8>(x, y, tw);
        if (step == 3) phase<kPrioLight>(x, y);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = ((i >> l) << (l + 1)) | (i & ((1 << l) - 1));
            v[m] = x[i]; v[m + (1 << l)] = y[i];
        }
    }
}

__device__ __forceinline__ uint4 scale4(uint4 x, u32 s) {
    return make_uint4(m31_mul(x.x, s), m31_mul(x.y, s), m31_mul(x.z, s), m31_mul(x.w, s));
}
// all 16 words of a lane times `s` (the 2^-n of the last inverse pass), in priority phases; returns at kPrioLight
__device__ __forceinline__ void scale16(uint4 (&x)[4], u32 s) {
    const u32 s2 = s + s;
    u32 a[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
    u32 b[8] = {x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
    mul8_dbl(a, s2);
    mul8_dbl(b, s2);
    phase<kPrioLight>(a, b);
    x[0] = make_uint4(a[0], a[1], a[2], a[3]); x[1] = make_uint4(a[4], a[5], a[6], a[7]);
    x[2] = make_uint4(b[0], b[1], b[2], b[3]); x[3] = make_uint4(b[4], b[5], b[6], b[7]);
}

// optional second pointer table of a kernel (a read-only source); an empty struct when the kernel works in place
struct NoSrc {};
template <int EXT> struct SrcTable { using type = ColPtrs; };
template <> struct SrcTable<0> { using type = NoSrc; };

// ------------------------------------------------------------------------------------------------
// Bottom pass: layers 0..LOGT-1 (circle layer included) of a contiguous 2^LOGT-word tile, LOGT in 11..13
// (2^(LOGT-4) lanes).  LOGT = 13 is the default; smaller tiles give more workgroups when there are few columns.
// OOP: tiles are read from `src` (left untouched) and written to `cols` — the first pass of an out-of-place interpolation.
#ifndef TSTWO_B_WAVES
#define TSTWO_B_WAVES 6
#endif
template <bool INV, int LOGT, bool OOP = false>
__global__ void __launch_bounds__(1 << (LOGT - 4), TSTWO_B_WAVES) k_cfft_b(ColPtrs cols, typename SrcTable<OOP ? 1 : 0>::type src, u32 n_cols, u32 total_items,
                                                 u32 n, const u32 *__restrict__ tw_end, u32 scale) {
    constexpr int THREADS = 1 << (LOGT - 4);
    constexpr int GM = LOGT - 10;              // layers of the middle LDS stage (bits [8, LOGT-2))
    constexpr u32 T = 1u << LOGT, QT = T / 4;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *twl = lds + T + T / 32;                          // 2^(LOGT-4)-entry heap: layer bits 4..LOGT-3
    const u32 t = threadIdx.x;
    // Work items are (tile, column) pairs in tile-major order; this workgroup owns a contiguous range of them (equal shares,
    // the grid is sized to what is resident at once: no tail round), cut below into runs of columns that share a tile.
    const u32 share = total_items / gridDim.x, extra = total_items % gridDim.x;
    u32 item = blockIdx.x * share + min(blockIdx.x, extra);
    const u32 item_end = item + share + (blockIdx.x < extra ? 1u : 0u);
#pragma unroll 1
    while (item < item_end) {
    // (the quotient is wave-uniform but comes out of the VALU's float-reciprocal division sequence: readfirstlane moves it
    // back to an SGPR so that everything derived from it — tile base, column pointers — stays scalar)
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(item / n_cols));   // tile index
    const u32 col0 = item - hi * n_cols;
    const u32 col1 = min(n_cols, col0 + (item_end - item));
    item += col1 - col0;
    const size_t base = (size_t)hi << LOGT;

    auto src_of = [&](u32 col) -> const u32 * {
        if constexpr (OOP) return colp_u<kSecondTableOff>(src, col) + base;
        else return colp_u(cols, col) + base;
    };
    // Workgroup start: EVERY global load of the prologue is issued back to back — the first tile, the lane's register
    // twiddles (layers 1..3), its heap entry, the two top-layer twiddles — and only then is anything used.  A memory round
    // trip costs ~2300 cycles here (s_memtime stamps); written in "load, use, load, use" order the compiler
    // waited for each one in turn (five round trips, ~5 us per launch of a few-column transform).
    uint4 pf[4];
    u32 t1[4], t2[2], t3, ta, tb0, tb1;
    // (forward pass, final stores: e0 = ((t >> 6) << 10) + 4 (t & 63) is the word offset of the lane's 16-byte piece in its
    // wave's 1024 consecutive words, x 4 pieces 256 words apart — the coalesced form of the lane's 16 consecutive words)
    {
        u32 tp = t;                          // opaque per tile run: keeps the prologue's lane addresses out of kernel-lifetime registers
        asm volatile("" : "+v"(tp));
        const u32 *__restrict__ d = src_of(col0);
#pragma unroll
        for (int j = 0; j < 4; j++)
            pf[j] = INV ? gload4(d, 16 * tp + 4 * j) : gload4(d + 4 * tp + j * QT);
        // (twiddle loads as wave-uniform base + 32-bit lane offset: a 64-bit address pair per load, all of them live at once
        // because the loads are issued back to back, is what used to spill in this prologue)
        uint4 q1 = gload4(tw_end - ((size_t)1 << (n - 1)) + ((size_t)hi << (LOGT - 2)), 4 * tp);
        uint2 q2 = gload2(tw_end - ((size_t)1 << (n - 2)) + ((size_t)hi << (LOGT - 3)), 2 * tp);
        u32 q3 = gload1(tw_end - ((size_t)1 << (n - 3)) + ((size_t)hi << (LOGT - 4)), tp);
        // heap: level lv in 2..LOGT-5 holds layer bit b = LOGT-1 - lv (lanes 0..3 have no entry: they load lane 4's and drop it)
        const u32 th = max(tp, 4u);
        const u32 lv = 31u - (u32)__clz(th);
        const u32 hb = (u32)(LOGT - 1) - lv;
        // its word lies 2^(n-hb) - (hi << lv) - (th - 2^lv) words below tw_end, hb >= 4: within 2^(n-4) words of it
        const u32 hoff = (1u << (n - 4)) - (1u << (n - hb)) + (hi << lv) + (th - (1u << lv));
        u32 hv = gload1(tw_end - ((size_t)1 << (n - 4)), hoff);
        u32 a = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 1))) + (ptrdiff_t)hi];           // layer LOGT-1 / LOGT-2 (wave-uniform)
        u32 b0 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 2))) + (ptrdiff_t)(2 * (size_t)hi)];
        u32 b1 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 2))) + (ptrdiff_t)(2 * (size_t)hi + 1)];
        asm volatile("" : "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y), "+v"(q3), "+v"(hv));   // one wait for all of them
        t1[0] = q1.x + q1.x; t1[1] = q1.y + q1.y; t1[2] = q1.z + q1.z; t1[3] = q1.w + q1.w;
        t2[0] = q2.x + q2.x; t2[1] = q2.y + q2.y;
        t3 = q3 + q3;
        if (tp >= 4) twl[tp] = hv + hv;
        ta = a + a; tb0 = b0 + b0; tb1 = b1 + b1;
    }
    if (INV) lds_barrier();      // the inverse reads the heap in its first LDS stage, which no workgroup barrier precedes

    for (u32 col = col0; col < col1; col++) {
        u32 *__restrict__ data = colp_u(cols, col) + base;
        const u32 *__restrict__ next = src_of(min(col + 1, col1 - 1));
        // Opaque per iteration (inverse and the 2^11 tile): the LDS addresses of the stages (about twenty, all functions of the
        // lane id) are then recomputed per column instead of living in registers for the whole kernel — hoisted, they spilled
        // (one scratch reload inside this loop).  The forward 2^12 / 2^13 kernels fit without it, and there the ~20 extra address
        // instructions per column cost more than they free: 3.92 against 3.85 ms for 256 x 2^22 (gpurun_out/r03b/ab1.log).
        u32 tt = t;
        if (INV || LOGT == 11) asm volatile("" : "+v"(tt));
        if (!INV) {
            top_layers<false, true>(pf, ta, tb0, tb1);                    // layers LOGT-1, LOGT-2
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32 *p = lds + pad(4 * tt) + j * (QT + QT / 32);
                p[0] = pf[j].x; p[1] = pf[j].y; p[2] = pf[j].z; p[3] = pf[j].w;
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) pf[j] = gload4(next + 4 * t + j * QT);   // (pointer form: the base + 32-bit offset form costs this kernel registers it does not have: +17 us)
            lds_stage<GM, 8, LOGT, THREADS, false>(lds, twl, tt);              // layers LOGT-3..8
            lds_barrier();
            lds_stage<4, 4, LOGT, THREADS, false>(lds, twl, tt);               // layers 7..4
            lds_wave_fence();    // the 256-word blocks a wave wrote in that stage are the ones it reads now: no workgroup barrier
            u32 v[16];
#pragma unroll
            for (int m = 0; m < 16; m++) v[m] = lds[pad(16 * tt) + m];
            low_layers<false>(v, t1, t2, t3);                              // layers 3, 2, 1 and the circle layer
            // The lane now holds 16 consecutive words (64 bytes): stored as they are, every store instruction would touch 64
            // separate 64-byte segments a quarter each.  One more trip through the wave's own 1024 words of LDS turns them
            // into four 1 KiB-contiguous 16-byte-per-lane stores (wave-local: no workgroup barrier).
            lds_wave_fence();
#pragma unroll
            for (int m = 0; m < 16; m++) lds[pad(16 * tt) + m] = v[m];
            lds_wave_fence();
            const u32 e0 = ((tt >> 6) << 10) + 4 * (tt & 63);
            uint4 o[4];
            const u32 *pe0 = lds + pad(e0);       // pad(e0 + 256 j) = pad(e0) + 264 j: e0's bits 8, 9 are clear
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 *p = pe0 + 264 * j;
                o[j] = make_uint4(p[0], p[1], p[2], p[3]);
            }
            lds_barrier();       // last LDS access of this column: the next column's tile may overwrite it
#pragma unroll
            for (int j = 0; j < 4; j++) gstore4(data + e0 + 256 * j, o[j]);
        } else {
            u32 v[16];
            // (the lane's 64 consecutive bytes arrive as four 16-byte loads; the coalesced form with a wave-local LDS exchange,
            // which pays on the forward pass's store side, measured 2 % slower here: loads share their lines through the cache)
#pragma unroll
            for (int j = 0; j < 4; j++) { v[4 * j] = pf[j].x; v[4 * j + 1] = pf[j].y; v[4 * j + 2] = pf[j].z; v[4 * j + 3] = pf[j].w; }
            low_layers<true>(v, t1, t2, t3);
#pragma unroll
            for (int m = 0; m < 16; m++) lds[pad(16 * tt) + m] = v[m];
            lds_wave_fence();    // the next stage reads the blocks this wave has just written
#pragma unroll
            for (int j = 0; j < 4; j++) pf[j] = gload4(next, 16 * t + 4 * j);
            lds_stage<4, 4, LOGT, THREADS, true>(lds, twl, tt);
            lds_barrier();
            lds_stage<GM, 8, LOGT, THREADS, true>(lds, twl, tt);
            lds_barrier();
            uint4 x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 *p = lds + pad(4 * tt) + j * (QT + QT / 32);
                x[j] = make_uint4(p[0], p[1], p[2], p[3]);
            }
            lds_barrier();       // last LDS access of this column (see the forward branch)
            top_layers<true, true>(x, ta, tb0, tb1);
            if (scale) scale16(x, scale);
#pragma unroll
            for (int j = 0; j < 4; j++) gstore4(data, 4 * t + j * QT, x[j]);
        }
    }
    }   // runs of one tile
}

#ifdef TSTWO_EXPERIMENTS
// ------------------------------------------------------------------------------------------------
// Bottom pass with 8 words per lane (experiments build only; the round-3 verdict's "one structural attempt"): a 2^LOGT tile on
// 2^(LOGT-3) lanes — 1024 at LOGT = 13, two workgroups per CU = 8 waves per SIMD where k_cfft_b has 6 — at the price of five
// exchanges of 8 words where k_cfft_b has 3.5 of 16: layers LOGT-1, LOGT-2 in registers on four 8-byte loads a quarter-tile apart |
// LOGT-3..LOGT-5 | LOGT-6..LOGT-8 | 4, 3 (two groups of 4 words per lane) | 2, 1, 0 on the lane's 8 consecutive words, stored as
// two 16-byte pieces.  4 butterflies per lane and layer: priority phases of 4.  In place only.  Measured: DESIGN.md 4.1.
template <bool INV, int LOGT>
__global__ void __launch_bounds__(1 << (LOGT - 3), 8) k_cfft_b8(ColPtrs cols, NoSrc, u32 n_cols, u32 total_items, u32 n,
                                                                const u32 *__restrict__ tw_end, u32 scale) {
    static_assert(LOGT == 13, "stage bits below are written for the 2^13 tile");
    constexpr int THREADS = 1 << (LOGT - 3);
    constexpr u32 T = 1u << LOGT, QT = T / 4;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *twl = lds + T + T / 32;                          // 2^(LOGT-3)-entry heap: layer bits 3..LOGT-3
    const u32 t = threadIdx.x;
    const u32 share = total_items / gridDim.x, extra = total_items % gridDim.x;
    u32 item = blockIdx.x * share + min(blockIdx.x, extra);
    const u32 item_end = item + share + (blockIdx.x < extra ? 1u : 0u);
#pragma unroll 1
    while (item < item_end) {
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(item / n_cols));   // tile index
    const u32 col0 = item - hi * n_cols;
    const u32 col1 = min(n_cols, col0 + (item_end - item));
    item += col1 - col0;
    const size_t base = (size_t)hi << LOGT;
    uint2 pf2[4];           // forward: four 8-byte pieces a quarter-tile apart
    uint4 pf4[2];           // inverse: the lane's 8 consecutive words
    u32 t1[2], t2, ta, tb0, tb1;
    {
        const u32 *__restrict__ d = colp_u(cols, col0) + base;
        if (!INV) {
#pragma unroll
            for (int j = 0; j < 4; j++) pf2[j] = gload2(d, 2 * t + j * QT);
        } else {
            pf4[0] = gload4(d, 8 * t); pf4[1] = gload4(d, 8 * t + 4);
        }
        const uint2 q1 = gload2(tw_end - ((size_t)1 << (n - 1)) + ((size_t)hi << (LOGT - 2)), 2 * t);        // layer 1: h = idx >> 2 = 2 t, 2 t + 1
        const u32 q2 = gload1(tw_end - ((size_t)1 << (n - 2)) + ((size_t)hi << (LOGT - 3)), t);              // layer 2: h = t
        const u32 th = max(t, 4u);
        const u32 lv = 31u - (u32)__clz(th);
        const u32 hb = (u32)(LOGT - 1) - lv;                                                                   // heap: layer bit hb in 3..LOGT-3
        const u32 hoff = (1u << (n - 3)) - (1u << (n - hb)) + (hi << lv) + (th - (1u << lv));
        const u32 hv = gload1(tw_end - ((size_t)1 << (n - 3)), hoff);
        const u32 a = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 1))) + (ptrdiff_t)hi];
        const u32 b0 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 2))) + (ptrdiff_t)(2 * (size_t)hi)];
        const u32 b1 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (LOGT - 2))) + (ptrdiff_t)(2 * (size_t)hi + 1)];
        t1[0] = q1.x + q1.x; t1[1] = q1.y + q1.y; t2 = q2 + q2;
        if (t >= 4) twl[t] = hv + hv;
        ta = a + a; tb0 = b0 + b0; tb1 = b1 + b1;
    }
    lds_barrier();          // heap visible before any stage reads it
    // the three lowest layers on 8 consecutive words: layer 2 (t2), layer 1 (t1[2]), circle layer (t1 permuted / negated)
    auto low3 = [&](u32 (&v)[8]) {
        u32 tc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            tc[j] = t1[(j >> 1) ^ 1];
            if ((j ^ (j >> 1)) & 1) tc[j] = 0xFFFFFFFEu - tc[j];
        }
#pragma unroll
        for (int step = 0; step < 3; step++) {
            const int l = INV ? step : 2 - step;
            u32 x[4], y[4], tw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = ((i >> l) << (l + 1)) | (i & ((1 << l) - 1));
                x[i] = v[m]; y[i] = v[m + (1 << l)];
                tw[i] = l == 0 ? tc[m >> 1] : l == 1 ? t1[m >> 2] : t2;
            }
            bf_layer<INV, 4>(x, y, tw);
            if (step == 2) phase<kPrioLight>(x, y);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = ((i >> l) << (l + 1)) | (i & ((1 << l) - 1));
                v[m] = x[i]; v[m + (1 << l)] = y[i];
            }
        }
    };
    // the two top layers on the four quarter-tile pairs w[2 j], w[2 j + 1]: layer LOGT-1 pairs quarters (0, 2), (1, 3) with ta;
    // layer LOGT-2 pairs (0, 1) with tb0, (2, 3) with tb1
    auto top2 = [&](u32 (&w)[8]) {
#pragma unroll
        for (int step = 0; step < 2; step++) {
            const bool top = INV ? step == 1 : step == 0;
            u32 x[4], y[4], tw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // butterfly i: component i & 1 of quarter pair (i >> 1)
                const int qa = top ? (i >> 1) : 2 * (i >> 1), qb = top ? qa + 2 : qa + 1;
                x[i] = w[2 * qa + (i & 1)]; y[i] = w[2 * qb + (i & 1)];
                tw[i] = top ? ta : ((i >> 1) ? tb1 : tb0);
            }
            bf_layer<INV, 4>(x, y, tw);
            if (step == 1) phase<kPrioLight>(x, y);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qa = top ? (i >> 1) : 2 * (i >> 1), qb = top ? qa + 2 : qa + 1;
                w[2 * qa + (i & 1)] = x[i]; w[2 * qb + (i & 1)] = y[i];
            }
        }
    };
    for (u32 col = col0; col < col1; col++) {
        u32 *__restrict__ data = colp_u(cols, col) + base;
        const u32 *__restrict__ next = colp_u(cols, min(col + 1, col1 - 1)) + base;
        u32 tt = t;
        asm volatile("" : "+v"(tt));
        if (!INV) {
            u32 w[8] = {pf2[0].x, pf2[0].y, pf2[1].x, pf2[1].y, pf2[2].x, pf2[2].y, pf2[3].x, pf2[3].y};
            top2(w);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32 *p = lds + pad(2 * tt) + j * (QT + QT / 32);
                p[0] = w[2 * j]; p[1] = w[2 * j + 1];
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) pf2[j] = gload2(next, 2 * t + j * QT);
            lds_stage<3, LOGT - 5, LOGT, THREADS, false, 8>(lds, twl, tt);       // layers LOGT-3 .. LOGT-5
            lds_barrier();
            lds_stage<3, LOGT - 8, LOGT, THREADS, false, 8>(lds, twl, tt);       // layers LOGT-6 .. LOGT-8
            lds_barrier();
            lds_stage<2, 3, LOGT, THREADS, false, 8>(lds, twl, tt);              // layers 4, 3
            lds_barrier();
            u32 v[8];
#pragma unroll
            for (int m = 0; m < 8; m++) v[m] = lds[pad(8 * tt) + m];
            lds_barrier();       // last LDS access of this column
            low3(v);
            gstore4(data, 8 * t, make_uint4(v[0], v[1], v[2], v[3]));
            gstore4(data, 8 * t + 4, make_uint4(v[4], v[5], v[6], v[7]));
        } else {
            u32 v[8] = {pf4[0].x, pf4[0].y, pf4[0].z, pf4[0].w, pf4[1].x, pf4[1].y, pf4[1].z, pf4[1].w};
            low3(v);
#pragma unroll
            for (int m = 0; m < 8; m++) lds[pad(8 * tt) + m] = v[m];
            lds_barrier();
            pf4[0] = gload4(next, 8 * t); pf4[1] = gload4(next, 8 * t + 4);
            lds_stage<2, 3, LOGT, THREADS, true, 8>(lds, twl, tt);
            lds_barrier();
            lds_stage<3, LOGT - 8, LOGT, THREADS, true, 8>(lds, twl, tt);
            lds_barrier();
            lds_stage<3, LOGT - 5, LOGT, THREADS, true, 8>(lds, twl, tt);
            lds_barrier();
            u32 w[8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 *p = lds + pad(2 * tt) + j * (QT + QT / 32);
                w[2 * j] = p[0]; w[2 * j + 1] = p[1];
            }
            lds_barrier();       // last LDS access of this column
            top2(w);
            if (scale) {
                mul8_dbl(w, scale + scale);
                phase<kPrioLight>(w);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) gstore2(data, 2 * t + j * QT, make_uint2(w[2 * j], w[2 * j + 1]));
        }
    }
    lds_barrier();       // the stages read the twiddle heap: the next run restages it
    }   // runs of one tile
}
#endif

// ------------------------------------------------------------------------------------------------
// Strided pass: K layers [lo, lo+K) on a tile of 2^K rows x 2^C words (C = LOGT - K).
// EXT > 0 (forward only, the pass that holds the transform's top layers): the input is a polynomial of log size
// n - EXT in its own buffers (`src`), zero-extended to log size n on the fly.  The top EXT layers of a zero-padded input
// only replicate (butterfly(a, 0, t) = (a, a)), so the quarter-tile vectors a lane needs are copies of each other:
// they are loaded once from the small polynomial, the replicated layers are skipped, and the result goes to `cols`.
// LOGT (12..15) = log2 of the tile.  14 (1024 lanes x 16 words) is the default; the smaller tiles exist for transforms of few
// columns, where 2^(n-14) tiles would leave most of the 256 CUs without a workgroup.  LOGT = 15 is the 128 KiB tile: still 1024
// lanes (the workgroup limit) and still one workgroup per CU, every lane working as V = 2 "virtual lanes" of 16 words each, one
// after the other inside every stage — the same stage code, twice the work between two barriers, twice the bytes in flight per
// CU — so that a 10-layer pass keeps 128-byte rows (n = 24 = 14 + 10: two passes over HBM instead of three) and a 9-layer pass
// has 256-byte rows.
// V = 2 at LOGT = 14 (512 lanes x 32 words, 66 KiB: TWO workgroups per CU, i.e. two independent barrier domains where the 2^15
// tile has one) is instantiated for A/B timing (experiments build: TSTWO_CFFT_AV=2).
#ifdef TSTWO_A_WAVES          // experiments: minimum waves per SIMD asked of the strided pass (8 = two 1024-lane workgroups per CU)
#define TSTWO_A_BOUNDS(LOGT, V) __launch_bounds__((1 << ((LOGT) - 4)) / (V), TSTWO_A_WAVES)
#else
#define TSTWO_A_BOUNDS(LOGT, V) __launch_bounds__((1 << ((LOGT) - 4)) / (V))
#endif
template <bool INV, int K, int EXT = 0, int LOGT = 14, int V = (LOGT == 15 ? 2 : 1)>
__global__ void TSTWO_A_BOUNDS(LOGT, V) k_cfft_a(ColPtrs cols, typename SrcTable<EXT>::type src, u32 n_cols, u32 total_items, u32 n,
                                                   u32 lo, const u32 *__restrict__ tw_end, u32 scale) {
    static_assert(EXT == 0 || (!INV && K >= 2 && EXT <= 2), "fused extension: forward pass with two register layers");
    static_assert(LOGT >= 12 && LOGT <= 15 && LOGT - K >= 4, "strided tile: rows of at least 16 words");
    static_assert((V == 1 || V == 2) && (1 << (LOGT - 4)) / V <= 1024, "virtual lanes (16 words each) per lane");
#ifdef TSTWO_A_SB
    constexpr bool SB = true;
#else
    constexpr bool SB = V == 2;                // scalar-base addressing (below)
#endif
    constexpr int VT = 1 << (LOGT - 4);        // virtual lanes per tile
    constexpr int THREADS = VT / V, C = LOGT - K;
    constexpr u32 T = 1u << LOGT, QT = T / 4;
    constexpr int F = K >= 2 ? 2 : 1;          // layers fused into the load / store
    constexpr int R = K - F;                   // layers run as LDS stages
    constexpr int G2 = R > 4 ? 4 : R;          // stage on bits [C, C+G2)
    constexpr int G1 = R - G2;                 // stage on bits [C+4, C+4+G1)
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32 *twl = lds + T + T / 32;               // heap of 2^K entries (levels F..K-1)
    const u32 t = threadIdx.x;
    // contiguous range of (tile, column) work items, as in k_cfft_b
    const u32 share = total_items / gridDim.x, extra = total_items % gridDim.x;
    u32 item = blockIdx.x * share + min(blockIdx.x, extra);
    const u32 item_end = item + share + (blockIdx.x < extra ? 1u : 0u);
#pragma unroll 1
    while (item < item_end) {
    const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)(item / n_cols));
    const u32 col0 = item - tile * n_cols;
    const u32 col1 = min(n_cols, col0 + (item_end - item));
    item += col1 - col0;
    const u32 mid_bits = lo - C;
    const u32 mid = tile & ((1u << mid_bits) - 1u);
    const u32 hi = tile >> mid_bits;
    const size_t base = ((size_t)hi << (lo + K)) | ((size_t)mid << C);
    // tile-relative word offset: < 2^(lo + K) <= 2^30 words (log_size <= 30), so 32 bits hold it
    auto goff = [&](u32 e) -> u32 { return ((e >> C) << lo) + (e & ((1u << C) - 1u)); };
    // Scalar-base addressing (SB; the 2^15 tile, and -DTSTWO_A_SB for A/B timing of the others).  A word offset of this kernel is
    // goff(e) with e a bit-rearrangement of (virtual lane id, g, m): every bit of the inputs lands on its own output bit, so
    //   goff(e(t + c0, m)) = goff(e(t, 0)) + goff(e(c0, m))        (c0 = v THREADS + g VT: compile-time, above t's bits)
    // — ONE lane offset per access pattern (loop-invariant, a single VGPR) plus a wave-uniform term that goes into the scalar
    // base of `global_load/store v_off, s[base]`.  The pointer form costs a 64-bit VGPR address (and the 64-bit adds that make
    // it) per access: 32 of them per virtual lane in the final stage, which is what spilled at 32 words per lane.
    const u32 lane4 = goff(4 * t);                                   // quarter-tile vectors: word 4 t of each quarter

    // Twiddle staging (LDS heap for the stage layers, ta / tb for the register layers).  Called AFTER the first tile's loads
    // have been issued: the heap fill waits for its own global loads, and a memory round trip costs ~2300 cycles here
    // (s_memtime stamps), so tile loads issued behind it would pay that latency a second time.
    u32 ta = 0, tb0 = 0, tb1 = 0;
    auto stage_twiddles = [&]() {
        static_assert((1 << K) <= THREADS, "one heap entry per lane at most");
        u32 hv = 0;
        if constexpr (R > 0) {      // unconditional load (lanes without an entry read entry 2^F and drop it): no branch around it
            const u32 idx = min(max(t, (u32)(1 << F)), (u32)(1 << K) - 1u);
            const u32 lv = 31u - (u32)__clz(idx);
            const u32 i = lo + (K - 1 - lv);           // b = LOGT-1-lv, i = lo + b - C
            hv = tw_end[-(ptrdiff_t)((size_t)1 << (n - i)) + (ptrdiff_t)(((size_t)hi << lv) + (idx - (1u << lv)))];
        }
        const u32 a = tw_end[-(ptrdiff_t)((size_t)1 << (n - (lo + K - 1))) + (ptrdiff_t)hi];
        u32 b0 = 0, b1 = 0;
        if constexpr (F == 2) {
            b0 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (lo + K - 2))) + (ptrdiff_t)(2 * (size_t)hi)];
            b1 = tw_end[-(ptrdiff_t)((size_t)1 << (n - (lo + K - 2))) + (ptrdiff_t)(2 * (size_t)hi + 1)];
        }
        if constexpr (R > 0) {
            asm volatile("" : "+v"(hv));
            if (t >= (1u << F) && t < (1u << K)) twl[t] = hv + hv;
        }
        ta = a + a; tb0 = b0 + b0; tb1 = b1 + b1;
    };
    // element offsets of a virtual lane's 16 words in the final-stage layout (R > 0)
    auto e_final = [&](u32 vt, int g, int m) -> u32 {
        const u32 gid = vt + (u32)g * VT;
        const u32 low = gid & ((1u << C) - 1u), high = gid >> C;
        return (high << (C + G2)) | ((u32)m << C) | low;
    };
    const u32 lane_f = goff(e_final(t, 0, 0));                       // SB: lane part of the final-stage offsets
    auto u_final = [&](int v, int g, int m) -> u32 { return goff(e_final((u32)v * THREADS, g, m)); };    // SB: uniform part
    auto u_quarter = [&](int v, int j) -> u32 { return goff(4u * (u32)v * THREADS + (u32)j * QT); };

    if constexpr (!INV || R == 0) {
        // (column accesses in pointer form here: with the base + 32-bit-offset form of the inverse branch this kernel is 4 us
        // faster and the bottom pass that follows it 17 us slower — same bottom-pass code, measured twice on one box)
        // forward (and the LDS-free case): four quarter-tile vectors per virtual lane
        uint4 pf[V][4];
        // loads of a virtual lane's four quarter-tile vectors (j = 2 * top bit + second bit); with EXT the vectors that differ
        // only in replicated bits are one load from the small polynomial
        auto load_tile = [&](const u32 *__restrict__ d) {
#pragma unroll
            for (int v = 0; v < V; v++) {
                const u32 vt = t + (u32)v * THREADS;
                if constexpr (SB) {
#pragma unroll
                    for (int j = 0; j < (EXT == 0 ? 4 : EXT == 1 ? 2 : 1); j++) pf[v][j] = gload4(d + u_quarter(v, j), lane4);
                } else if constexpr (EXT == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) pf[v][j] = gload4(d + goff(4 * vt + j * QT));
                } else if constexpr (EXT == 1) {
                    pf[v][0] = gload4(d + goff(4 * vt));
                    pf[v][1] = gload4(d + goff(4 * vt + QT));
                } else {
                    pf[v][0] = gload4(d + goff(4 * vt));
                }
            }
        };
        auto src_of = [&](u32 col) -> const u32 * {
            if constexpr (EXT == 0) return colp_u(cols, col) + base;
            else return colp_u<kSecondTableOff>(src, col) + base;       // the first pass has hi == 0: base only carries bits below the replicated ones
        };
        load_tile(src_of(col0));
        stage_twiddles();
        for (u32 col = col0; col < col1; col++) {
            u32 *__restrict__ data = colp_u(cols, col) + base;
            const u32 *__restrict__ next = src_of(min(col + 1, col1 - 1));
            u32 tt = t;
            asm volatile("" : "+v"(tt));     // opaque per iteration: keeps the 16 scatter addresses out of loop-invariant registers
#pragma unroll
            for (int v = 0; v < V; v++) {
                if constexpr (EXT == 0) {
                    top_layers<INV, F == 2>(pf[v], ta, tb0, tb1);
                } else if constexpr (EXT == 1) {     // top layer replicates: x2 = x0, x3 = x1; second layer is real
                    pf[v][2] = pf[v][0]; pf[v][3] = pf[v][1];
                    quarter_layer_second<false, true>(pf[v], tb0, tb1);
                } else {                              // both register layers replicate
                    pf[v][1] = pf[v][0]; pf[v][2] = pf[v][0]; pf[v][3] = pf[v][0];
                }
            }
            if constexpr (R == 0) {
#pragma unroll
                for (int v = 0; v < V; v++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint4 x = pf[v][j];
                        if (INV && scale) x = scale4(x, scale);
                        if constexpr (SB) gstore4(data + u_quarter(v, j), lane4, x);
                        else gstore4(data + goff(4 * (t + (u32)v * THREADS) + j * QT), x);
                    }
                load_tile(next);
            } else {
#pragma unroll
                for (int v = 0; v < V; v++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        // pad(4 (t + v THREADS)) = pad(4 t) + v (4 THREADS + THREADS / 8): 4 THREADS is a multiple of 32 above 4 t's bits
                        u32 *p = lds + pad(4 * t) + v * (4 * THREADS + THREADS / 8) + j * (QT + QT / 32);
                        p[0] = pf[v][j].x; p[1] = pf[v][j].y; p[2] = pf[v][j].z; p[3] = pf[v][j].w;
                    }
                lds_barrier();
                load_tile(next);
                if constexpr (G1 > 0) {
#pragma unroll
                    for (int v = 0; v < V; v++) lds_stage<G1, C + 4, LOGT, VT, false>(lds, twl, t + (u32)v * THREADS);
                    lds_barrier();
                }
                // final stage: butterflies, then straight to HBM (rows of 2^C words: >= 128 B per half-wave)
                {
                    // (virtual lanes one after the other: 16 tile words live at a time beside the 16 V words of the prefetch)
                    constexpr int NG = 16 >> G2;
#pragma unroll
                    for (int v = 0; v < V; v++) {
                        u32 w[NG][1 << G2], high[NG];
#pragma unroll
                        for (int g = 0; g < NG; g++) {
                            high[g] = (t + (u32)v * THREADS + (u32)g * VT) >> C;
#pragma unroll
                            for (int m = 0; m < (1 << G2); m++) w[g][m] = lds[pad(e_final(tt + (u32)v * THREADS, g, 0)) + off<C>(m)];
                        }
                        if (v == V - 1) lds_barrier();   // last access to the tile in LDS: the next column may overwrite it while this stage computes
                        groups_layers<G2, NG, C, LOGT, false>(w, twl, high);
#pragma unroll
                        for (int g = 0; g < NG; g++)
#pragma unroll
                            for (int m = 0; m < (1 << G2); m++) {
                                if constexpr (SB) gstore1(data + u_final(v, g, m), lane_f, w[g][m]);
                                else gstore1(data + goff(e_final(tt + (u32)v * THREADS, g, m)), w[g][m]);
                            }
                    }
                }
            }
        }
        if constexpr (R > 0) lds_barrier();      // the final stage read the twiddle heap: the next run restages it
    } else {
        // inverse with LDS stages: a virtual lane's 16 words arrive in the first-stage layout
        constexpr int NG = 16 >> G2;
        u32 pfs[V][16];
        {
            const u32 tt = t;
            const u32 *__restrict__ d = colp_u(cols, col0) + base;
#pragma unroll
            for (int v = 0; v < V; v++)
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int m = 0; m < (1 << G2); m++)
                        pfs[v][g * (1 << G2) + m] = SB ? gload1(d + u_final(v, g, m), lane_f) : gload1(d, goff(e_final(tt + (u32)v * THREADS, g, m)));
        }
        stage_twiddles();
        lds_barrier();       // the inverse reads the heap in its first stage, before any other barrier
        for (u32 col = col0; col < col1; col++) {
            u32 *__restrict__ data = colp_u(cols, col) + base;
            const u32 *__restrict__ next = colp_u(cols, min(col + 1, col1 - 1)) + base;
            u32 tt = t;
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int v = 0; v < V; v++) {
                u32 w[NG][1 << G2], high[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    high[g] = (t + (u32)v * THREADS + (u32)g * VT) >> C;
#pragma unroll
                    for (int m = 0; m < (1 << G2); m++) w[g][m] = pfs[v][g * (1 << G2) + m];
                }
                groups_layers<G2, NG, C, LOGT, true>(w, twl, high);
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int m = 0; m < (1 << G2); m++) lds[pad(e_final(tt + (u32)v * THREADS, g, 0)) + off<C>(m)] = w[g][m];
            }
            lds_barrier();
#pragma unroll
            for (int v = 0; v < V; v++)
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int m = 0; m < (1 << G2); m++)
                        pfs[v][g * (1 << G2) + m] = SB ? gload1(next + u_final(v, g, m), lane_f) : gload1(next, goff(e_final(tt + (u32)v * THREADS, g, m)));
            if constexpr (G1 > 0) {
#pragma unroll
                for (int v = 0; v < V; v++) lds_stage<G1, C + 4, LOGT, VT, true>(lds, twl, t + (u32)v * THREADS);
                lds_barrier();
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                uint4 x[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 *p = lds + pad(4 * t) + v * (4 * THREADS + THREADS / 8) + j * (QT + QT / 32);
                    x[j] = make_uint4(p[0], p[1], p[2], p[3]);
                }
                if (v == V - 1) lds_barrier();       // last LDS access of this column
                top_layers<true, F == 2>(x, ta, tb0, tb1);
                if (scale) scale16(x, scale);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (SB) gstore4(data + u_quarter(v, j), lane4, x[j]);
                    else gstore4(data, goff(4 * (t + (u32)v * THREADS) + j * QT), x[j]);
                }
            }
        }
    }
    }   // runs of one tile
}

}  // namespace fast
