// merkle.hip — MerkleOps.commitOnLayer / MerkleProver.commit with BLAKE2s-256 (RFC 7693, unkeyed,
// 32-byte digest — what @noble/hashes blake2s computes for the reference, vcs/blake2_hash.ts:53).
//
// One lane per tree node.  Node message = [left32 || right32]? || LE32(col_0[i]) || ... || LE32(col_{C-1}[i])
// (vcs/blake2_merkle.ts:9-24).  Column-major columns make lane i read word i of every column: one
// coalesced 256-byte access per column per wave.  The 16 message words live in VGPRs; the 10 rounds
// are fully unrolled so SIGMA indexes registers at compile time; rotations are v_alignbit_b32.
// The compression is VALU-bound (~1.2k lane-ops per 64-byte block, ~19 ops/byte) — DESIGN.md §Merkle
// prices it against the integer-issue ceiling as well as the HBM roofline the bench reports.
// Algorithmic bytes for a layer of n nodes: 4*C*n (+ 64*n children) read, 32*n written.
#include <string.h>

#include <vector>

#include "common.h"
#include <stdlib.h>

using namespace tstwo;

namespace {

constexpr u32 IV0 = 0x6A09E667u, IV1 = 0xBB67AE85u, IV2 = 0x3C6EF372u, IV3 = 0xA54FF53Au, IV4 = 0x510E527Fu,
              IV5 = 0x9B05688Cu, IV6 = 0x1F83D9ABu, IV7 = 0x5BE0CD19u;

__device__ __forceinline__ u32 rotr32(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }

#define B2S_G(a, b, c, d, x, y)                     \
    do {                                            \
        a = a + b + (x); d = rotr32(d ^ a, 16);     \
        c = c + d;       b = rotr32(b ^ c, 12);     \
        a = a + b + (y); d = rotr32(d ^ a, 8);      \
        c = c + d;       b = rotr32(b ^ c, 7);      \
    } while (0)

// Four independent G functions (a column step or a diagonal step of a round) issued opcode by opcode in priority phases
// (common.h: heavy = v_add3 / v_alignbit on port 0 at high priority, light = v_xor / v_add on either port): per step
// 24 heavy + 24 light instructions.  Entered and left at kPrioHeavy.
#define B2S_4(OP) OP(0) OP(1) OP(2) OP(3)
#define B2S_STEP4(a0, b0, c0, d0, a1, b1, c1, d1, a2, b2, c2, d2, a3, b3, c3, d3, x0, y0, x1, y1, x2, y2, x3, y3) \
    do {                                                                                                          \
        a0 = a0 + b0 + (x0); a1 = a1 + b1 + (x1); a2 = a2 + b2 + (x2); a3 = a3 + b3 + (x3);                       \
        phase<kPrioLight>(a0, a1, a2, a3);                                                                        \
        d0 ^= a0; d1 ^= a1; d2 ^= a2; d3 ^= a3;                                                                   \
        phase<kPrioHeavy>(d0, d1, d2, d3);                                                                        \
        d0 = rotr32(d0, 16); d1 = rotr32(d1, 16); d2 = rotr32(d2, 16); d3 = rotr32(d3, 16);                       \
        phase<kPrioLight>(d0, d1, d2, d3);                                                                        \
        c0 += d0; c1 += d1; c2 += d2; c3 += d3;                                                                   \
        b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3;                                                                   \
        phase<kPrioHeavy>(b0, b1, b2, b3);                                                                        \
        b0 = rotr32(b0, 12); b1 = rotr32(b1, 12); b2 = rotr32(b2, 12); b3 = rotr32(b3, 12);                       \
        a0 = a0 + b0 + (y0); a1 = a1 + b1 + (y1); a2 = a2 + b2 + (y2); a3 = a3 + b3 + (y3);                       \
        phase<kPrioLight>(a0, a1, a2, a3);                                                                        \
        d0 ^= a0; d1 ^= a1; d2 ^= a2; d3 ^= a3;                                                                   \
        phase<kPrioHeavy>(d0, d1, d2, d3);                                                                        \
        d0 = rotr32(d0, 8); d1 = rotr32(d1, 8); d2 = rotr32(d2, 8); d3 = rotr32(d3, 8);                           \
        phase<kPrioLight>(d0, d1, d2, d3);                                                                        \
        c0 += d0; c1 += d1; c2 += d2; c3 += d3;                                                                   \
        b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3;                                                                   \
        phase<kPrioHeavy>(b0, b1, b2, b3);                                                                        \
        b0 = rotr32(b0, 7); b1 = rotr32(b1, 7); b2 = rotr32(b2, 7); b3 = rotr32(b3, 7);                           \
    } while (0)

// One compression (vcs/blake2s_ref.ts:176-230): h <- F(h, m, t, last)
__device__ __forceinline__ void b2s_compress(u32 h[8], const u32 m[16], u32 t_lo, bool last) {
    u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u32 v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = IV4 ^ t_lo, v13 = IV5, v14 = last ? ~IV6 : IV6, v15 = IV7;
    phase<kPrioHeavy>(v0, v1, v2, v3);
    // message schedule SIGMA (vcs/blake2s_ref.ts:9-20) written out so every m[] index is a literal
#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                    \
    B2S_STEP4(v0, v4, v8, v12, v1, v5, v9, v13, v2, v6, v10, v14, v3, v7, v11, v15,                                         \
              m[s0], m[s1], m[s2], m[s3], m[s4], m[s5], m[s6], m[s7]);                                                      \
    B2S_STEP4(v0, v5, v10, v15, v1, v6, v11, v12, v2, v7, v8, v13, v3, v4, v9, v14,                                         \
              m[s8], m[s9], m[s10], m[s11], m[s12], m[s13], m[s14], m[s15]);
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef B2S_ROUND
    phase<kPrioLight>(v4, v5, v6, v7);
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

struct LayerParams {
    u32 total_words;   // W: message length in 32-bit words (16 if children, plus one per column)
    u32 w_begin;       // first message word handled by this launch (multiple of 16)
    u32 w_end;         // one past the last word handled (multiple of 16, or >= W on the final launch)
    u32 col_word0;     // message word index of cols.p[0]
    u32 n_cols;        // columns in this launch's table
    u32 load_state;    // 1: resume from the 8-word state stored in out[] by the previous launch
    u32 is_final;      // 1: this launch holds the last block (finalise)
};

// One lane hashes several nodes (grid-stride) and software-pipelines the message: the 16 words of the next
// 64-byte block are fetched into a second register set while the current block is compressed, so the
// compression (VALU-bound, ~1k ops) hides the HBM latency.  Loads are never branched around (a per-word
// branch makes hipcc wait vmcnt(0) per element): out-of-range words read a clamped column and are zeroed
// by a select.
template <bool HAS_PREV>
__device__ __forceinline__ void load_block(u32 (&m)[16], const uint4 *__restrict__ prev, const HashColPtrs &cols,
                                           const LayerParams &lp, size_t node, u32 w) {
    if (HAS_PREV && w == 0) {      // wave-uniform
        const uint4 *c = prev + 4 * node;
        uint4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
        m[0] = c0.x; m[1] = c0.y; m[2] = c0.z; m[3] = c0.w; m[4] = c1.x; m[5] = c1.y; m[6] = c1.z; m[7] = c1.w;
        m[8] = c2.x; m[9] = c2.y; m[10] = c2.z; m[11] = c2.w; m[12] = c3.x; m[13] = c3.y; m[14] = c3.z; m[15] = c3.w;
    } else if (lp.n_cols == 0) {   // wave-uniform: empty message block
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = 0u;
    } else {
        const u32 last = lp.n_cols - 1u;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 ci = w + (u32)k - lp.col_word0;                // wave-uniform
            const u32 v = cols.p[min(ci, last)][node];
            m[k] = (ci <= last) ? v : 0u;
        }
    }
}

template <bool HAS_PREV>
__global__ void __launch_bounds__(256) k_merkle_layer(const uint4 *__restrict__ prev, HashColPtrs cols, uint4 *__restrict__ out,
                                                     size_t n_nodes, LayerParams lp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t node0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 W = lp.total_words;
    const u32 w_stop = lp.w_end < W ? lp.w_end : W;
    // wave-uniform trip structure: `rows` nodes per lane (tail lanes clamp their loads and skip their stores),
    // `nb` 64-byte blocks per node
    const u32 nb = w_stop > lp.w_begin ? (w_stop - lp.w_begin + 15u) / 16u : 1u;
    const u32 rows = (u32)((n_nodes + stride - 1) / stride);
    const u32 total = rows * nb;
    const size_t last_node = n_nodes - 1;

    u32 ma[16], mb[16], h[8];
    u32 j = 0, blk = 0;                                   // uniform: row number / block number of the block in flight
    load_block<HAS_PREV>(ma, prev, cols, lp, min(node0, last_node), lp.w_begin);

#define MERKLE_STEP(CUR, NXT)                                                                                         \
    {                                                                                                                 \
        const size_t node = node0 + (size_t)j * stride;                                                               \
        const size_t nc = min(node, last_node);                                                                       \
        const u32 wc = lp.w_begin + 16u * blk;                                                                        \
        if (blk == 0) {                                                                                               \
            if (lp.load_state) {                                                                                      \
                uint4 a = out[2 * nc], b = out[2 * nc + 1];                                                           \
                h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;       \
            } else {                                                                                                  \
                h[0] = IV0 ^ 0x01010020u; h[1] = IV1; h[2] = IV2; h[3] = IV3; h[4] = IV4; h[5] = IV5; h[6] = IV6; h[7] = IV7; \
            }                                                                                                         \
        }                                                                                                             \
        u32 jn = j, bn = blk + 1;                                                                                     \
        if (bn == nb) { bn = 0; jn = j + 1; }                                                                         \
        if (it + 1 < total)                                                                                           \
            load_block<HAS_PREV>(NXT, prev, cols, lp, min(node0 + (size_t)jn * stride, last_node), lp.w_begin + 16u * bn); \
        const u32 bytes_end = (wc + 16 < W ? wc + 16 : W) * 4u;                                                       \
        b2s_compress(h, CUR, bytes_end, lp.is_final && (wc + 16 >= W));                                               \
        if (blk + 1 == nb && node < n_nodes) {                                                                        \
            out[2 * node] = make_uint4(h[0], h[1], h[2], h[3]);                                                       \
            out[2 * node + 1] = make_uint4(h[4], h[5], h[6], h[7]);                                                   \
        }                                                                                                             \
        j = jn; blk = bn;                                                                                             \
    }

    for (u32 it = 0; it < total; it += 2) {
        MERKLE_STEP(ma, mb)
        if (it + 1 < total) {
            const u32 it_save = it;
            it = it_save + 1;
            MERKLE_STEP(mb, ma)
            it = it_save;
        }
    }
#undef MERKLE_STEP
}

// ---- lean special cases of k_merkle_layer (same results, fewer non-hash instructions per compression) ----
// (1) bottom layer whose column count is exactly 16*NBLK (e.g. the 32-column trace shard): every message word has
//     a compile-time column index, so the column pointers are fetched once (scalar registers) instead of a
//     clamp + scalar load + select per word per block.
// Up to kMaxTrees equally shaped trees per launch (tstwo_merkle_commit_many): blockIdx.y selects the tree — its columns are
// cols.p[blockIdx.y * 16 * NBLK ...] of the by-value table, its layers buffer ts.t[blockIdx.y].
constexpr int kMaxTrees = 8;
struct TreeSet { uint4 *t[kMaxTrees]; };
template <int NBLK>
__global__ void __launch_bounds__(256) k_merkle_leaf_static(HashColPtrs cols, TreeSet outs, size_t n_nodes) {
    uint4 *__restrict__ out = outs.t[blockIdx.y];
    const u32 col0 = blockIdx.y * (16 * NBLK);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t node0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 rows = (u32)((n_nodes + stride - 1) / stride);
    const size_t last_node = n_nodes - 1;
    u32 cur[16], nxt[16];
    // word `node` of column k = scalar base (kernel argument) + ONE 32-bit byte offset shared by all columns: global_load with
    // an SGPR base and a VGPR offset, instead of a 64-bit address pair per column in VGPRs (columns are at most 4 GiB: the
    // host takes this kernel for log_size <= 30 only)
    auto word = [&](int k, u32 byte_off) -> u32 { return *(const TSTWO_GLOBAL u32 *)((const TSTWO_GLOBAL char *)cols.p[col0 + k] + byte_off); };
    {
        const u32 oc = (u32)min(node0, last_node) * 4u;
#pragma unroll
        for (int k = 0; k < 16; k++) cur[k] = word(k, oc);
    }
    // The digest of node j is stored one compression LATER (behind the loads of node j + 1's first block): the compiler waits
    // vmcnt(0) at the loop header, so a store issued at the end of an iteration has its whole write latency exposed there
    // (gfx9 stores count on vmcnt); issued here it has a full compression to complete.  Costs 8 VGPRs.
    u32 hp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t pnode = n_nodes;                                       // node whose digest is waiting in hp (none yet)
    for (u32 j = 0; j < rows; j++) {
        const size_t node = node0 + (size_t)j * stride;
        const size_t nn = min(node + stride, last_node);          // next node of this lane (clamped: loads are never branched around)
        const size_t nc = min(node, last_node);
        u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            // fetch the next 64-byte block (next block of this node, or block 0 of the lane's next node) while this one is compressed
            const int bn = (b + 1) % NBLK;
            const u32 src = (u32)((b + 1 < NBLK) ? nc : nn) * 4u;
#pragma unroll
            for (int k = 0; k < 16; k++) nxt[k] = word(16 * bn + k, src);
            if (b == 0 && pnode < n_nodes) {
                out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
            }
            b2s_compress(h, cur, 64u * (b + 1), b == NBLK - 1);
#pragma unroll
            for (int k = 0; k < 16; k++) cur[k] = nxt[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) hp[k] = h[k];
        pnode = node;
    }
    if (pnode < n_nodes) {
        out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
    }
}

// fold_line fused into the leaf hashing of the NEXT FRI layer's tree (the folded row IS that tree's 16-byte leaf message,
// vcs/blake2_merkle.ts:9-24 over the 4 coordinate columns): FoldSpec describes the layer being folded; the leaf kernels then
// compute row i = f0 + alpha f1 from rows 2i, 2i+1 (fri.ts:120-152), store it to the new evaluation and hash it from registers.
struct FoldSpec { const u32 *in[4]; const u32 *inv_x; const u32 *alpha; };       // alpha: 4 words in device memory (the drawn QM31)
struct FoldRow { uint2 a, b, c, d; u32 t; };
__device__ __forceinline__ FoldRow fold_row_load(const FoldSpec &fs, size_t i) {
    return {gload2(fs.in[0] + 2 * i), gload2(fs.in[1] + 2 * i), gload2(fs.in[2] + 2 * i), gload2(fs.in[3] + 2 * i), gload1(fs.inv_x + i)};
}
__device__ __forceinline__ qm31 fold_row(const FoldRow &r, qm31 alpha) {
    const qm31 f0 = {m31_add(r.a.x, r.a.y), m31_add(r.b.x, r.b.y), m31_add(r.c.x, r.c.y), m31_add(r.d.x, r.d.y)};
    const qm31 f1 = qm31_mul_m31({m31_sub(r.a.x, r.a.y), m31_sub(r.b.x, r.b.y), m31_sub(r.c.x, r.c.y), m31_sub(r.d.x, r.d.y)}, r.t);
    return qm31_add(f0, qm31_mul(alpha, f1));
}

// (1b) bottom layer of exactly 4 columns — every FRI layer (the 4 coordinate columns of a QM31 column): a 16-byte
//      message, one final block whose words 4..15 are zero.
template <bool FOLD>
__global__ void __launch_bounds__(256) k_merkle_leaf4(u32 *__restrict__ c0, u32 *__restrict__ c1, u32 *__restrict__ c2, u32 *__restrict__ c3,
                                                     uint4 *__restrict__ out, size_t n_nodes, FoldSpec fs) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t node0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 rows = (u32)((n_nodes + stride - 1) / stride);
    const size_t last_node = n_nodes - 1;
    auto word = [&](const u32 *col, u32 byte_off) -> u32 { return *(const TSTWO_GLOBAL u32 *)((const TSTWO_GLOBAL char *)col + byte_off); };
    const u32 o0 = (u32)min(node0, last_node) * 4u;       // scalar base + one 32-bit offset (log_size <= 30: host)
    u32 a = 0, b = 0, c = 0, d = 0;
    FoldRow fr = {};
    qm31 alpha = {0, 0, 0, 0};
    if (FOLD) {
        alpha = {fs.alpha[0], fs.alpha[1], fs.alpha[2], fs.alpha[3]};
        fr = fold_row_load(fs, min(node0, last_node));
    } else {
        a = word(c0, o0); b = word(c1, o0); c = word(c2, o0); d = word(c3, o0);
    }
    u32 hp[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // deferred store: see k_merkle_leaf_static
    size_t pnode = n_nodes;
    for (u32 j = 0; j < rows; j++) {
        const size_t node = node0 + (size_t)j * stride;
        const size_t nn = min(node + stride, last_node);
        const u32 on = (u32)nn * 4u;
        u32 na = 0, nb = 0, ncc = 0, nd = 0;
        FoldRow nfr = {};
        if (FOLD) {
            nfr = fold_row_load(fs, nn);                             // next node's rows in flight during the compression
        } else {
            na = word(c0, on); nb = word(c1, on); ncc = word(c2, on); nd = word(c3, on);      // next node's words in flight during the compression
        }
        if (pnode < n_nodes) {
            out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
            out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
        }
        if (FOLD) {
            const qm31 r = fold_row(fr, alpha);
            a = r.a; b = r.b; c = r.c; d = r.d;
            if (node < n_nodes) { gstore1(c0 + node, a); gstore1(c1 + node, b); gstore1(c2 + node, c); gstore1(c3 + node, d); }
        }
        u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
        const u32 m[16] = {a, b, c, d, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        b2s_compress(h, m, 16u, true);
#pragma unroll
        for (int k = 0; k < 8; k++) hp[k] = h[k];
        pnode = node;
        if (FOLD) fr = nfr;
        else { a = na; b = nb; c = ncc; d = nd; }
    }
    if (pnode < n_nodes) {
        out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
    }
}

// (2) inner layer without columns: node = Blake2s(left || right), one 64-byte block.
__device__ __forceinline__ void merkle_inner_body(const uint4 *__restrict__ prev, uint4 *__restrict__ out, size_t n_nodes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t node0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 rows = (u32)((n_nodes + stride - 1) / stride);
    const size_t last_node = n_nodes - 1;
    uint4 c[4], cn[4];
    {
        const uint4 *p = prev + 4 * min(node0, last_node);
        c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; c[3] = p[3];
    }
    u32 hp[8] = {0, 0, 0, 0, 0, 0, 0, 0};                        // deferred store: see k_merkle_leaf_static
    size_t pnode = n_nodes;
    for (u32 j = 0; j < rows; j++) {
        const size_t node = node0 + (size_t)j * stride;
        const uint4 *pn = prev + 4 * min(node + stride, last_node);
        cn[0] = pn[0]; cn[1] = pn[1]; cn[2] = pn[2]; cn[3] = pn[3];
        if (pnode < n_nodes) {
            out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
            out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
        }
        u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
        const u32 m[16] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w,
                           c[2].x, c[2].y, c[2].z, c[2].w, c[3].x, c[3].y, c[3].z, c[3].w};
        b2s_compress(h, m, 64u, true);
#pragma unroll
        for (int k = 0; k < 8; k++) hp[k] = h[k];
        pnode = node;
        c[0] = cn[0]; c[1] = cn[1]; c[2] = cn[2]; c[3] = cn[3];
    }
    if (pnode < n_nodes) {
        out[2 * pnode] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        out[2 * pnode + 1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
    }
}

__global__ void __launch_bounds__(256) k_merkle_inner(const uint4 *__restrict__ prev, uint4 *__restrict__ out, size_t n_nodes) {
    merkle_inner_body(prev, out, n_nodes);
}
// the same for layer log_out of every tree of a set (children = its layer log_out + 1)
__global__ void __launch_bounds__(256) k_merkle_inner_set(TreeSet ts, u32 log_out) {
    uint4 *layers = ts.t[blockIdx.y];
    merkle_inner_body(layers + 2 * (((size_t)1 << (log_out + 1)) - 1), layers + 2 * (((size_t)1 << log_out) - 1), (size_t)1 << log_out);
}

// (3) LEVELS column-free layers in one launch, no exchange at all: a lane owns 2^LEVELS consecutive nodes of layer
// `log_child` — a whole subtree — and produces the 2^(LEVELS-1), ..., 1 nodes above them in post-order (left, right, parent),
// so that a parent's message is the two digests still in registers.  Every lane is busy at every level (a layer-per-launch
// chain halves the grid each time and ends in launches that are all latency), intermediate layers are written once and
// never read back, and two or three launches disappear.  All levels are written to their places in the layers buffer
// (MerkleProver keeps all layers, vcs/prover.ts:24-29; layer k at byte offset 32*(2^k - 1)).
struct Digest { u32 w[8]; };
__device__ __forceinline__ Digest hash_pair(const Digest &l, const Digest &r) {
    Digest d = {{IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7}};
    const u32 m[16] = {l.w[0], l.w[1], l.w[2], l.w[3], l.w[4], l.w[5], l.w[6], l.w[7], r.w[0], r.w[1], r.w[2], r.w[3], r.w[4], r.w[5], r.w[6], r.w[7]};
    b2s_compress(d.w, m, 64u, true);
    return d;
}
__device__ __forceinline__ Digest load_digest(const uint4 *p) {
    const uint4 a = p[0], b = p[1];
    return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ void store_digest(uint4 *p, const Digest &d) {
    p[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
    p[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}
// node `idx` of layer `log_child - LVL` (LVL levels above the children) of the lane's subtree; children read from HBM at LVL == 1.
// The two sub-subtrees are a rolled loop (one copy of each level's compression in the code: fully inlined, the 7 or 15
// compressions of a kernel took the compiler tens of minutes).
template <int LVL>
__device__ __forceinline__ Digest subtree_node(uint4 *__restrict__ layers, u32 log_child, size_t idx) {
    Digest l, r;
    if constexpr (LVL == 1) {
        const uint4 *c = layers + 2 * ((((size_t)1 << log_child) - 1) + 2 * idx);
        l = load_digest(c);
        r = load_digest(c + 2);
    } else {
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            const Digest d = subtree_node<LVL - 1>(layers, log_child, 2 * idx + s);
            if (s == 0) l = d;
            else r = d;
        }
    }
    const Digest d = hash_pair(l, r);
    store_digest(layers + 2 * ((((size_t)1 << (log_child - LVL)) - 1) + idx), d);
    return d;
}
template <int LEVELS>
__global__ void __launch_bounds__(256) k_merkle_subtree(TreeSet ts, u32 log_child) {
    uint4 *__restrict__ layers = ts.t[blockIdx.y];
    const size_t top = (size_t)blockIdx.x * blockDim.x + threadIdx.x;         // node index in layer log_child - LEVELS
    if (top >= ((size_t)1 << (log_child - LEVELS))) return;
    if constexpr (LEVELS == 2) {
        // straight-line form of the two-level subtree: all four children are requested before anything is hashed (the rolled
        // recursion loads a pair, waits, hashes, stores, and waits vmcnt(0) for that store before it loads the next pair), and
        // every store is followed by a compression or by the end of the wave, so no write latency is ever waited for
        const uint4 *c = layers + 2 * ((((size_t)1 << log_child) - 1) + 4 * top);
        const Digest c0 = load_digest(c), c1 = load_digest(c + 2), c2 = load_digest(c + 4), c3 = load_digest(c + 6);
        uint4 *mid = layers + 2 * ((((size_t)1 << (log_child - 1)) - 1) + 2 * top);
        const Digest l = hash_pair(c0, c1);
        store_digest(mid, l);
        const Digest r = hash_pair(c2, c3);
        store_digest(mid + 2, r);
        store_digest(layers + 2 * ((((size_t)1 << (log_child - 2)) - 1) + top), hash_pair(l, r));
    } else {
        (void)subtree_node<LEVELS>(layers, log_child, top);
    }
}

// The two-level subtree with every global access a 1 KiB-contiguous wave access (round 4).  In k_merkle_subtree<2> a lane reads its
// four children as eight 16-byte loads at a 128-byte lane stride and writes its digests at 64- and 32-byte lane strides: every
// instruction touches 64 lines a piece each, the pieces of a line arrive a compression apart, and with 32 waves per CU the lines
// do not survive in the caches in between — the counters showed 1.31 x the child bytes fetched and 1.16-1.24 x the digest bytes
// written (profiles/r03_cfft_pmc.json), on launches that move 4.3 TB/s.  Here a wave loads its 256 children as eight 1 KiB rows,
// transposes them to "lane owns 128 consecutive bytes" through its OWN 4.5 KiB of LDS (two halves of 4 KiB, one pad slot per
// 8 chunks: conflict-free both ways; wave-local, so no barrier — the LDS executes a wave's instructions in order), and stores
// the 128 + 64 digests it produced the same way, all six store instructions back to back at the end of the wave.
// Needs the top layer (2^(log_child-2) nodes) to be a multiple of 256 nodes: the host falls back to k_merkle_subtree<2> otherwise.
__device__ __forceinline__ u32 xslot(u32 chunk) { return chunk + (chunk >> 3); }
__global__ void __launch_bounds__(256) k_merkle_subtree2c(TreeSet ts, u32 log_child) {
    __shared__ uint4 xch[4][288];                       // per wave: 256 chunks of 16 bytes + 32 pad slots
    uint4 *__restrict__ layers = ts.t[blockIdx.y];
    const u32 lane = threadIdx.x & 63u;
    uint4 *x = xch[threadIdx.x >> 6];
    const size_t top0 = (size_t)blockIdx.x * 256u + (threadIdx.x & ~63u);        // the wave's first node of layer log_child - 2
    const u32 *cbase = (const u32 *)(layers + 2 * ((((size_t)1 << log_child) - 1) + 4 * top0));
    uint4 in[8];
#pragma unroll
    for (int k = 0; k < 8; k++) in[k] = gload4(cbase, 4u * (64u * k + lane));
    Digest ch[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 4; k++) x[xslot(64u * k + lane)] = in[4 * h + k];
        asm volatile("" ::: "memory");
        if ((lane >> 5) == (u32)h) {                    // chunks 8 l .. 8 l + 7 of this half belong to lane 32 h + l
            const uint4 *mine = x + 9u * (lane & 31u);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 a = mine[2 * j], b = mine[2 * j + 1];
                ch[j] = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
            }
        }
        asm volatile("" ::: "memory");
    }
    const Digest l = hash_pair(ch[0], ch[1]);
    const Digest r = hash_pair(ch[2], ch[3]);
    const Digest top = hash_pair(l, r);
    // layer log_child - 1: the wave's 128 digests = 256 chunks; lane's chunks 4 lane .. 4 lane + 3
    {
        uint4 *mine = x + xslot(4u * lane);             // 4 lane + j, j < 4, stays inside one group of 8: same pad
        mine[0] = make_uint4(l.w[0], l.w[1], l.w[2], l.w[3]); mine[1] = make_uint4(l.w[4], l.w[5], l.w[6], l.w[7]);
        mine[2] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]); mine[3] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
    }
    asm volatile("" ::: "memory");
    uint4 o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = x[xslot(64u * k + lane)];
    asm volatile("" ::: "memory");
    {
        uint4 *mine = x + xslot(2u * lane);
        mine[0] = make_uint4(top.w[0], top.w[1], top.w[2], top.w[3]); mine[1] = make_uint4(top.w[4], top.w[5], top.w[6], top.w[7]);
    }
    asm volatile("" ::: "memory");
    uint4 o2[2];
#pragma unroll
    for (int k = 0; k < 2; k++) o2[k] = x[xslot(64u * k + lane)];
    u32 *mid = (u32 *)(layers + 2 * ((((size_t)1 << (log_child - 1)) - 1) + 2 * top0));
    u32 *up = (u32 *)(layers + 2 * ((((size_t)1 << (log_child - 2)) - 1) + top0));
#pragma unroll
    for (int k = 0; k < 4; k++) gstore4(mid, 4u * (64u * k + lane), o[k]);
#pragma unroll
    for (int k = 0; k < 2; k++) gstore4(up, 4u * (64u * k + lane), o2[k]);
}

// Several column-free levels per launch: a workgroup of WG lanes owns 2*WG consecutive nodes of layer `log_child`
// and produces the WG, WG/2, ... nodes above them (LEVELS levels), exchanging digests through LDS.  Every level is
// still written to its place in the layers buffer (MerkleProver keeps all layers, vcs/prover.ts:24-29).
// layers: device buffer in tstwo_merkle_commit's layout (layer k at byte offset 32*(2^k - 1)).
template <int WG>
__global__ void __launch_bounds__(WG) k_merkle_up(uint4 *__restrict__ layers, u32 log_child, u32 levels) {
    __shared__ uint4 sh[WG * 2];                                  // digests of the level just produced (2 x uint4 each)
    const u32 t = threadIdx.x;
    const uint4 *__restrict__ child = layers + 2 * (((size_t)1 << log_child) - 1);
    u32 m[16];
    u32 active = min((u32)WG, 1u << (log_child - 1));             // parents this workgroup produces at the first level
    if (t < active) {
        const size_t node = (size_t)blockIdx.x * WG + t;           // parent index in layer log_child-1
        const uint4 *c = child + 4 * node;
        uint4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
        m[0] = c0.x; m[1] = c0.y; m[2] = c0.z; m[3] = c0.w; m[4] = c1.x; m[5] = c1.y; m[6] = c1.z; m[7] = c1.w;
        m[8] = c2.x; m[9] = c2.y; m[10] = c2.z; m[11] = c2.w; m[12] = c3.x; m[13] = c3.y; m[14] = c3.z; m[15] = c3.w;
    }
    for (u32 lv = 1; lv <= levels; lv++) {
        const u32 log_out = log_child - lv;
        u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
        if (t < active) {
            b2s_compress(h, m, 64u, true);
            uint4 *out = layers + 2 * (((size_t)1 << log_out) - 1) + 2 * ((size_t)blockIdx.x * active + t);
            out[0] = make_uint4(h[0], h[1], h[2], h[3]);
            out[1] = make_uint4(h[4], h[5], h[6], h[7]);
            sh[2 * t] = make_uint4(h[0], h[1], h[2], h[3]);
            sh[2 * t + 1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
        __syncthreads();
        active >>= 1;
        if (lv < levels && t < active) {
            uint4 c0 = sh[4 * t], c1 = sh[4 * t + 1], c2 = sh[4 * t + 2], c3 = sh[4 * t + 3];
            m[0] = c0.x; m[1] = c0.y; m[2] = c0.z; m[3] = c0.w; m[4] = c1.x; m[5] = c1.y; m[6] = c1.z; m[7] = c1.w;
            m[8] = c2.x; m[9] = c2.y; m[10] = c2.z; m[11] = c2.w; m[12] = c3.x; m[13] = c3.y; m[14] = c3.z; m[15] = c3.w;
        }
        __syncthreads();
    }
}

// ---- Upper tree, latency path: one compression spread over a QUAD of lanes (lane j of the quad owns column j of the
// 4x4 Blake2s state).  The column step is lane-local; the diagonal step rotates rows b, c, d by 1, 2, 3 lanes with DPP
// quad_perm moves and rotates them back.  A lane needs message words m[SIGMA[r][2j..]], i.e. a lane-dependent choice
// among registers that are literal per round: three v_cndmask on the constant lane masks j==1, j==2, j==3.
// ~1/2.4 of the dependent-instruction chain of the one-lane compression, which is what bounds the top of a tree.
__device__ __forceinline__ u32 quad_rot(u32 x, int by) {   // value held by lane (j + by) & 3 of this lane's quad
    return by == 1 ? (u32)__builtin_amdgcn_mov_dpp((int)x, 0x39, 0xF, 0xF, false)
         : by == 2 ? (u32)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false)
                   : (u32)__builtin_amdgcn_mov_dpp((int)x, 0x93, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 sel4(u32 x0, u32 x1, u32 x2, u32 x3, u32 j) {
    u32 r = x0;
    r = j == 1 ? x1 : r;
    r = j == 2 ? x2 : r;
    r = j == 3 ? x3 : r;
    return r;
}
// Single 64-byte final block from the initial state (a node of children only: hashNode, vcs/blake2_merkle.ts:9-24).
// Returns the digest words j (o_lo) and 4+j (o_hi) in lane j of the quad.
__device__ __forceinline__ void b2s_quad_block64(const u32 (&m)[16], u32 j, u32 &o_lo, u32 &o_hi) {
    const u32 ivlo = sel4(IV0, IV1, IV2, IV3, j), ivhi = sel4(IV4, IV5, IV6, IV7, j);
    const u32 h_lo = ivlo ^ (j == 0 ? 0x01010020u : 0u), h_hi = ivhi;
    u32 a = h_lo, b = h_hi, c = ivlo, d = ivhi ^ sel4(64u, 0u, 0xFFFFFFFFu, 0u, j);
#define B2SQ_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)          \
    {                                                                                            \
        u32 x = sel4(m[s0], m[s2], m[s4], m[s6], j), y = sel4(m[s1], m[s3], m[s5], m[s7], j);     \
        B2S_G(a, b, c, d, x, y);                                                                  \
        b = quad_rot(b, 1); c = quad_rot(c, 2); d = quad_rot(d, 3);                               \
        x = sel4(m[s8], m[s10], m[s12], m[s14], j); y = sel4(m[s9], m[s11], m[s13], m[s15], j);   \
        B2S_G(a, b, c, d, x, y);                                                                  \
        b = quad_rot(b, 3); c = quad_rot(c, 2); d = quad_rot(d, 1);                               \
    }
    B2SQ_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2SQ_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2SQ_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2SQ_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2SQ_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2SQ_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2SQ_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2SQ_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2SQ_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2SQ_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef B2SQ_ROUND
    o_lo = h_lo ^ a ^ c;
    o_hi = h_hi ^ b ^ d;
}

// The same with the message in LDS instead of registers: lane j of the quad reads its words of round r — m[SIGMA[r][2j]],
// m[SIGMA[r][2j+1]] for the column step, m[SIGMA[r][8+2j]], m[SIGMA[r][8+2j+1]] for the diagonal step — from 40 LDS byte addresses
// it computed ONCE (quad_msg_addrs: the quad's message slot does not move between tree levels).  40 ds_read_b32 per compression
// instead of 120 v_cndmask (the three selects per word above): a third fewer issue slots on a path where one wave issues alone.
typedef __attribute__((address_space(3))) const u32 lds_cu32;
struct QuadMsgAddrs { u32 a[40]; };
__device__ __forceinline__ void quad_msg_addrs(QuadMsgAddrs &qa, u32 msg_byte_base, u32 j) {
#define B2SQ_ADDR(r, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                   \
    qa.a[4 * r + 0] = msg_byte_base + sel4(4u * s0, 4u * s2, 4u * s4, 4u * s6, j);                             \
    qa.a[4 * r + 1] = msg_byte_base + sel4(4u * s1, 4u * s3, 4u * s5, 4u * s7, j);                             \
    qa.a[4 * r + 2] = msg_byte_base + sel4(4u * s8, 4u * s10, 4u * s12, 4u * s14, j);                          \
    qa.a[4 * r + 3] = msg_byte_base + sel4(4u * s9, 4u * s11, 4u * s13, 4u * s15, j);
    B2SQ_ADDR(0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2SQ_ADDR(1, 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2SQ_ADDR(2, 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2SQ_ADDR(3, 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2SQ_ADDR(4, 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2SQ_ADDR(5, 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2SQ_ADDR(6, 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2SQ_ADDR(7, 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2SQ_ADDR(8, 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2SQ_ADDR(9, 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef B2SQ_ADDR
}
__device__ __forceinline__ u32 lds_word(u32 byte_addr) { return *(lds_cu32 *)(uintptr_t)byte_addr; }
__device__ __forceinline__ void b2s_quad_block64_lds(const QuadMsgAddrs &qa, u32 j, u32 &o_lo, u32 &o_hi) {
    const u32 ivlo = sel4(IV0, IV1, IV2, IV3, j), ivhi = sel4(IV4, IV5, IV6, IV7, j);
    const u32 h_lo = ivlo ^ (j == 0 ? 0x01010020u : 0u), h_hi = ivhi;
    u32 a = h_lo, b = h_hi, c = ivlo, d = ivhi ^ sel4(64u, 0u, 0xFFFFFFFFu, 0u, j);
    u32 w[40];
#pragma unroll
    for (int k = 0; k < 40; k++) w[k] = lds_word(qa.a[k]);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        B2S_G(a, b, c, d, w[4 * r], w[4 * r + 1]);
        b = quad_rot(b, 1); c = quad_rot(c, 2); d = quad_rot(d, 3);
        B2S_G(a, b, c, d, w[4 * r + 2], w[4 * r + 3]);
        b = quad_rot(b, 3); c = quad_rot(c, 2); d = quad_rot(d, 1);
    }
    o_lo = h_lo ^ a ^ c;
    o_hi = h_hi ^ b ^ d;
}

// ---- Blake2sChannel on the device (channel/blake2.ts:25-224, Rust semantics).  State = 10 words: digest[8], n_challenges,
// n_sent.  One quad of lanes runs the (latency-bound) compressions; used by the FRI commit loop so that a layer's root
// never has to travel to the host before the next fold can be launched.
__device__ __forceinline__ void chan_hash64(const u32 (&m)[16], u32 j, u32 (&digest)[8]) {
    u32 lo, hi;
    b2s_quad_block64(m, j, lo, hi);
    // every lane of the quad needs the whole digest: word k lives in lane k & 3 (lo for k < 4, hi for k >= 4)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        digest[k] = (u32)__builtin_amdgcn_readlane((int)lo, k);
        digest[4 + k] = (u32)__builtin_amdgcn_readlane((int)hi, k);
    }
}
// mix_root (vcs/blake2_merkle.ts:28-31): digest <- H(digest || root), n_challenges += 1, n_sent <- 0; then (optionally)
// draw_felt (blake2.ts:158-184): H(digest || LE32(n_sent) || 0^28) until all 8 words < 2P; felt = first 4 words reduced.
// state in registers of every lane of a wave (d, n_chal, n_sent); root: 8 words (global or LDS); felt: the drawn QM31 (valid in
// every lane).  Executed by one whole wave (chan_hash64 broadcasts through readlane of lanes 0..3).
__device__ __forceinline__ void chan_mix_draw(u32 (&d)[8], u32 &n_chal, u32 &n_sent, const u32 *root, bool do_mix, bool do_draw, u32 (&felt)[4]) {
    const u32 j = threadIdx.x & 3;
    if (do_mix) {
        u32 m[16];
#pragma unroll
        for (int k = 0; k < 8; k++) { m[k] = d[k]; m[8 + k] = root[k]; }
        chan_hash64(m, j, d);
        n_chal += 1;
        n_sent = 0;
    }
    if (do_draw) {
        u32 w[8];
        bool ok = false;
        // retry probability per round ~ 2^-28; the loop is bounded so that the kernel always terminates (64 rejections in a
        // row have probability 2^-1792)
        for (int tries = 0; tries < 64 && !ok; tries++) {
            u32 m[16];
#pragma unroll
            for (int k = 0; k < 8; k++) { m[k] = d[k]; m[8 + k] = 0; }
            m[8] = n_sent;
            n_sent += 1;
            chan_hash64(m, j, w);
            ok = true;
#pragma unroll
            for (int k = 0; k < 8; k++) ok = ok && (w[k] < 2u * M31_P);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) felt[k] = w[k] >= M31_P ? w[k] - M31_P : w[k];       // M31.reduce of a value < 2P
    }
}
// Levels log_child-1 .. log_child-levels, 4 lanes per node: a workgroup of WG lanes owns WG/4 consecutive parents of the
// first level and everything above them (WG/4 -> 1 is log2(WG/4)+1 levels).  Children digests live in LDS between levels.
// the level loop shared by k_merkle_upq and k_merkle_leaf4_upq: `sh` holds the 2*active child digests of this workgroup
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int WG>
__device__ __forceinline__ void upq_levels(uint4 *__restrict__ layers, u32 *sh, u32 log_child, u32 levels, u32 active, u32 blk) {
    const u32 t = threadIdx.x, q = t >> 2, j = t & 3;
    QuadMsgAddrs qa;                                               // this lane's 40 message-word addresses: the quad's slot sh[16q..16q+15]
    quad_msg_addrs(qa, (u32)(uintptr_t)(__attribute__((address_space(3))) u32 *)sh + 64u * q, j);
    for (u32 lv = 1; lv <= levels; lv++) {
        const u32 log_out = log_child - lv;
        u32 o_lo = 0, o_hi = 0;
        const bool on = q < active;                                // quad-uniform
        if (on) {
            b2s_quad_block64_lds(qa, j, o_lo, o_hi);
            u32 *out = reinterpret_cast<u32 *>(layers + 2 * (((size_t)1 << log_out) - 1) + 2 * ((size_t)blk * active + q));
            out[j] = o_lo;
            out[4 + j] = o_hi;
        }
        // LDS-only barriers: __syncthreads() also waits for the digest stores above to be acknowledged by memory (vmcnt(0)) —
        // nobody in this launch reads them back, and that wait was most of a level's time on this latency-bound path
        lds_only_barrier();                                        // every quad has read its children
        if (on) {
            sh[8 * q + j] = o_lo;
            sh[8 * q + 4 + j] = o_hi;
        }
        lds_only_barrier();
        active >>= 1;
    }
}
// The launch that produces a tree's root can run the channel's mix_root + draw_felt on it right away (wave 0, root still in LDS):
// the FRI commit loop's "tree, then channel" pair as one launch (ChanHook; null pointers: no channel step).
struct ChanHook { u32 *chan, *felt; };
__device__ __forceinline__ void chan_step_from_lds(const ChanHook &hk, const u32 *root_lds) {
    if (!hk.chan || threadIdx.x >= 64) return;
    u32 d[8], f[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = hk.chan[k];
    u32 n_chal = hk.chan[8], n_sent = hk.chan[9];
    chan_mix_draw(d, n_chal, n_sent, root_lds, true, hk.felt != nullptr, f);
    if (hk.felt && threadIdx.x < 4) hk.felt[threadIdx.x] = threadIdx.x == 0 ? f[0] : threadIdx.x == 1 ? f[1] : threadIdx.x == 2 ? f[2] : f[3];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) hk.chan[k] = d[k];
        hk.chan[8] = n_chal;
        hk.chan[9] = n_sent;
    }
}
template <int WG>
__global__ void __launch_bounds__(WG) k_merkle_upq(TreeSet ts, u32 log_child, u32 levels, ChanHook hk) {
    uint4 *__restrict__ layers = ts.t[blockIdx.y];
    constexpr u32 Q = WG / 4;
    __shared__ __attribute__((aligned(16))) u32 sh[Q * 16];      // 2Q child digests x 8 words
    const u32 t = threadIdx.x;
    const u32 active = min(Q, 1u << (log_child - 1));             // parents this workgroup produces at the first level
    {
        const uint4 *child = layers + 2 * (((size_t)1 << log_child) - 1) + (size_t)blockIdx.x * (4 * active);
        if (t < 4 * active) reinterpret_cast<uint4 *>(sh)[t] = child[t];      // 2*active digests = 4*active uint4
    }
    __syncthreads();
    upq_levels<WG>(layers, sh, log_child, levels, active, blockIdx.x);
    if (log_child == levels) chan_step_from_lds(hk, sh);          // this launch reached layer 0: the root is sh[0..7]
}
// A small 4-column tree (every FRI layer below 2^17 rows) without a launch of its own for the leaves: the first 2*active lanes
// of the workgroup hash one leaf each (16-byte message, vcs/blake2_merkle.ts:9-24), write it to the leaf layer and to LDS,
// and the quad levels follow in the same launch.
template <int WG, bool FOLD>
__global__ void __launch_bounds__(WG) k_merkle_leaf4_upq(u32 *__restrict__ c0, u32 *__restrict__ c1, u32 *__restrict__ c2, u32 *__restrict__ c3,
                                                        uint4 *__restrict__ layers, u32 log_leaf, u32 levels, ChanHook hk, FoldSpec fs) {
    constexpr u32 Q = WG / 4;
    __shared__ __attribute__((aligned(16))) u32 sh[Q * 16];
    const u32 t = threadIdx.x;
    const u32 active = min(Q, 1u << (log_leaf - 1));              // parents of the first level in this workgroup
    if (t < 2 * active) {
        const size_t node = (size_t)blockIdx.x * (2 * active) + t;
        u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
        u32 m0, m1, m2, m3;
        if (FOLD) {
            const qm31 r = fold_row(fold_row_load(fs, node), {fs.alpha[0], fs.alpha[1], fs.alpha[2], fs.alpha[3]});
            m0 = r.a; m1 = r.b; m2 = r.c; m3 = r.d;
            gstore1(c0 + node, m0); gstore1(c1 + node, m1); gstore1(c2 + node, m2); gstore1(c3 + node, m3);
        } else {
            m0 = c0[node]; m1 = c1[node]; m2 = c2[node]; m3 = c3[node];
        }
        const u32 m[16] = {m0, m1, m2, m3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        b2s_compress(h, m, 16u, true);
        uint4 *leaf = layers + 2 * ((((size_t)1 << log_leaf) - 1) + node);
        const uint4 lo = make_uint4(h[0], h[1], h[2], h[3]), hi = make_uint4(h[4], h[5], h[6], h[7]);
        leaf[0] = lo; leaf[1] = hi;
        reinterpret_cast<uint4 *>(sh)[2 * t] = lo;
        reinterpret_cast<uint4 *>(sh)[2 * t + 1] = hi;
    }
    __syncthreads();
    upq_levels<WG>(layers, sh, log_leaf, levels, active, blockIdx.x);
    if (log_leaf == levels) chan_step_from_lds(hk, sh);
}

// Column-free levels log_child-1 .. log_stop of the tree, a few fused launches instead of one launch per level.
// Set by the FRI commit loop around a tstwo_merkle_commit call: the single-workgroup launch that produces the root takes the
// channel step with it and clears the hook; a hook still set afterwards means the tree went another way (the caller then
// launches k_channel_mix_draw itself).
// (thread_local: the hooks belong to the call chain that set them — a second host thread committing a tree of its own while
// tstwo_fri_commit_layers is between "set" and "consumed" must not pick them up.)
thread_local ChanHook g_chan_hook = {nullptr, nullptr};
// Likewise for a fold: set by merkle_commit4_folded around a tstwo_merkle_commit of the 4 NEW coordinate columns; the leaf launch
// of that tree folds the previous layer into them on the way (FoldSpec) and clears it.
thread_local FoldSpec g_fold = {};
thread_local bool g_fold_set = false;
int commit_upper_levels(uint8_t *layers, u32 log_child, u32 log_stop);
int commit_upper_levels(TreeSet ts, unsigned n_trees, u32 log_child, u32 log_stop) {
    Context &c = ctx();
    const ChanHook none = {nullptr, nullptr};
    auto take_hook = [&](bool reaches_root) {
        if (!reaches_root || n_trees != 1 || !g_chan_hook.chan) return none;
        const ChanHook h = g_chan_hook;
        g_chan_hook = none;
        return h;
    };
    uint8_t *layers = (uint8_t *)ts.t[0];             // (the one-lane scheme kept for A/B timing handles one tree)
    const bool one_lane = knobs().merkle_up_onelane;     // previous scheme, kept for A/B timing
    const bool small_wg = knobs().merkle_up_smallwg;     // 256-lane workgroups only (A/B timing)
    const bool narrow_first = knobs().merkle_up_narrow_first;   // 64-quad workgroups first, one 256-quad workgroup last (A/B timing)
    while (log_child > log_stop) {
        const u32 remaining = log_child - log_stop;
        const u32 parents_log = log_child - 1;
        if (one_lane && n_trees == 1) {
            if (parents_log >= 8) {               // >= 256 parents: 256-lane workgroups, up to 5 levels each
                u32 levels = remaining < 5 ? remaining : 5;
                hipLaunchKernelGGL(k_merkle_up<256>, dim3(1u << (parents_log - 8)), dim3(256), 0, c.stream, (uint4 *)layers, log_child, levels);
                log_child -= levels;
            } else {                              // the top of the tree (< 256 parents): one workgroup finishes it
                hipLaunchKernelGGL(k_merkle_up<256>, dim3(1), dim3(256), 0, c.stream, (uint4 *)layers, log_child, remaining);
                log_child -= remaining;
            }
        } else if (parents_log >= 9 && remaining >= 9 && remaining <= 16 && !small_wg && !narrow_first) {
            // 256 quads per workgroup, 9 levels each, FIRST: the wide levels (256 and 128 compressions on one CU are the slow
            // part of a single-workgroup tree top) run on 2^(parents_log-8) CUs side by side, and what is left (<= 7 levels)
            // fits one 64-quad workgroup, one wave per SIMD.  (The other order — 64-quad workgroups first, one 256-quad
            // workgroup to finish — put those wide levels on one CU.)
            hipLaunchKernelGGL(k_merkle_upq<1024>, dim3(1u << (parents_log - 8), n_trees), dim3(1024), 0, c.stream, ts, log_child, 9u, none);
            log_child -= 9;
        } else if (parents_log <= 6 && !narrow_first) {   // <= 64 parents: one 64-quad workgroup finishes the tree
            hipLaunchKernelGGL(k_merkle_upq<256>, dim3(1, n_trees), dim3(256), 0, c.stream, ts, log_child, remaining, take_hook(log_stop == 0));
            log_child -= remaining;
        } else if (parents_log <= 8 && !small_wg) {   // <= 256 parents: ONE workgroup of 256 quads finishes the tree (up to 9 levels)
            hipLaunchKernelGGL(k_merkle_upq<1024>, dim3(1, n_trees), dim3(1024), 0, c.stream, ts, log_child, remaining, take_hook(log_stop == 0));
            log_child -= remaining;
        } else if (parents_log >= 6) {            // >= 64 parents: 64 quads per workgroup, 64 -> 1 = up to 7 levels each
            u32 levels = remaining < 7 ? remaining : 7;
            hipLaunchKernelGGL(k_merkle_upq<256>, dim3(1u << (parents_log - 6), n_trees), dim3(256), 0, c.stream, ts, log_child, levels, none);
            log_child -= levels;
        } else {
            hipLaunchKernelGGL(k_merkle_upq<256>, dim3(1, n_trees), dim3(256), 0, c.stream, ts, log_child, remaining, take_hook(log_stop == 0));
            log_child -= remaining;
        }
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
int commit_upper_levels(uint8_t *layers, u32 log_child, u32 log_stop) {
    TreeSet one = {};
    one.t[0] = (uint4 *)layers;
    return commit_upper_levels(one, 1, log_child, log_stop);
}

__global__ void __launch_bounds__(64) k_channel_mix_draw(u32 *__restrict__ chan, const u32 *__restrict__ root, u32 *__restrict__ felt,
                                                        u32 do_mix, u32 do_draw) {
    u32 d[8], f[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = chan[k];
    u32 n_chal = chan[8], n_sent = chan[9];
    chan_mix_draw(d, n_chal, n_sent, root, do_mix != 0, do_draw != 0, f);
    if (do_draw && threadIdx.x < 4) {
        u32 v = f[0];
        v = threadIdx.x == 1 ? f[1] : v;
        v = threadIdx.x == 2 ? f[2] : v;
        v = threadIdx.x == 3 ? f[3] : v;
        felt[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) chan[k] = d[k];
        chan[8] = n_chal;
        chan[9] = n_sent;
    }
}

// ---- FRI commit tail: every layer from 2^log0 <= 2^9 rows down to the last one in ONE launch of one workgroup.  Per layer the
// host loop costs three launches (tree, channel, fold) of a few microseconds of work each plus the gaps between dependent
// kernels; here a layer is: the 4-column tree in LDS (leaves one per lane, levels by quads: the k_merkle_leaf4_upq body), the
// channel's mix_root + draw_felt by wave 0 (state kept in its registers across the layers), fold_line by the first 2^(lg-1)
// lanes.  Evaluations and trees go to the same buffers, in the same layouts, as the per-layer path.
struct FriTail {
    u32 *eval[11][4];        // eval[0]: the input evaluation (2^log0 rows, already folded); eval[i + 1]: output of fold i
    uint4 *tree[10];         // tree[i]: layers buffer of the tree over eval[i]
    u32 n_layers, log0;
    const u32 *pre[4];       // non-null: eval[0] is still to be computed — the fold_line of this evaluation of 2^(log0+1) rows with
    const u32 *pre_alpha;    // the alpha at pre_alpha (drawn by an earlier launch); the kernel writes eval[0]
};
__global__ void __launch_bounds__(1024) k_fri_tail(FriTail ft, const u32 *__restrict__ itw, u32 tw_log, u32 *__restrict__ chan,
                                                  u32 *__restrict__ alphas) {
    constexpr u32 Q = 256;
    __shared__ __attribute__((aligned(16))) u32 sh[Q * 16];
    __shared__ __attribute__((aligned(16))) u32 evl[4][512];         // the current layer's evaluation, coordinate-major
    __shared__ u32 alpha_sh[4];
    const u32 t = threadIdx.x;
    u32 d[8], n_chal = 0, n_sent = 0;
    if (t < 64) {
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = chan[k];
        n_chal = chan[8]; n_sent = chan[9];
    }
    // Between the layers the evaluation never goes through memory: row t of the next layer is computed by lane t — the lane that
    // hashes leaf t of the next tree (registers) — and the fold reads rows 2t, 2t+1 from LDS.  (The first form read the folded
    // rows back from global memory behind a device-scope fence, twice per layer: most of a layer's time outside its tree levels.)
    u32 row[4] = {0u, 0u, 0u, 0u};
    if (t < (1u << ft.log0)) {
        if (ft.pre_alpha) {          // the fold into the first layer of the tail rides along (one launch fewer)
            const qm31 alpha = *reinterpret_cast<const qm31 *>(ft.pre_alpha);
            const u32 tw0 = gload1(itw + ((size_t)1 << tw_log) - ((size_t)2 << ft.log0) + t);
            const uint2 a = gload2(ft.pre[0] + 2 * t), b = gload2(ft.pre[1] + 2 * t), c = gload2(ft.pre[2] + 2 * t), e = gload2(ft.pre[3] + 2 * t);
            const qm31 f0 = {m31_add(a.x, a.y), m31_add(b.x, b.y), m31_add(c.x, c.y), m31_add(e.x, e.y)};
            const qm31 f1 = qm31_mul_m31({m31_sub(a.x, a.y), m31_sub(b.x, b.y), m31_sub(c.x, c.y), m31_sub(e.x, e.y)}, tw0);
            const qm31 r0 = qm31_add(f0, qm31_mul(alpha, f1));
            row[0] = r0.a; row[1] = r0.b; row[2] = r0.c; row[3] = r0.d;
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) { gstore1(ft.eval[0][c2] + t, row[c2]); evl[c2][t] = row[c2]; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) { row[c] = gload1(ft.eval[0][c] + t); evl[c][t] = row[c]; }
        }
    }
    for (u32 i = 0; i < ft.n_layers; i++) {
        const u32 lg = ft.log0 - i;                                  // 1 <= lg <= 9
        uint4 *layers = ft.tree[i];
        const u32 active = 1u << (lg - 1);
        // the fold's x^-1 of this layer: requested now, used behind the tree and the channel step
        const u32 tw = t < active ? gload1(itw + ((size_t)1 << tw_log) - ((size_t)1 << lg) + t) : 0u;
        // tree over the 4 coordinate columns (vcs/blake2_merkle.ts:9-24): leaves, then all levels
        if (t < 2 * active) {
            u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
            const u32 m[16] = {row[0], row[1], row[2], row[3], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            b2s_compress(h, m, 16u, true);
            uint4 *leaf = layers + 2 * ((((size_t)1 << lg) - 1) + t);
            const uint4 lo = make_uint4(h[0], h[1], h[2], h[3]), hi = make_uint4(h[4], h[5], h[6], h[7]);
            leaf[0] = lo; leaf[1] = hi;
            reinterpret_cast<uint4 *>(sh)[2 * t] = lo;
            reinterpret_cast<uint4 *>(sh)[2 * t + 1] = hi;
        }
        lds_only_barrier();
        upq_levels<1024>(layers, sh, lg, lg, active, 0u);            // the root is in sh[0..7] afterwards
        // channel: mix the root, draw alpha (wave 0; channel/blake2.ts:115-184, Rust draw semantics)
        if (t < 64) {
            u32 f[4];
            chan_mix_draw(d, n_chal, n_sent, sh, true, true, f);
            if (t < 4) {
                const u32 v = t == 0 ? f[0] : t == 1 ? f[1] : t == 2 ? f[2] : f[3];
                alpha_sh[t] = v;
                alphas[4 * i + t] = v;
            }
        }
        lds_only_barrier();
        // fold_line (fri.ts:120-152): row t of the next evaluation from rows 2t, 2t + 1 of this one
        qm31 r = {0u, 0u, 0u, 0u};
        if (t < active) {
            const qm31 alpha = {alpha_sh[0], alpha_sh[1], alpha_sh[2], alpha_sh[3]};
            const uint2 a = *reinterpret_cast<const uint2 *>(&evl[0][2 * t]), b = *reinterpret_cast<const uint2 *>(&evl[1][2 * t]),
                        c = *reinterpret_cast<const uint2 *>(&evl[2][2 * t]), e = *reinterpret_cast<const uint2 *>(&evl[3][2 * t]);
            const qm31 f0 = {m31_add(a.x, a.y), m31_add(b.x, b.y), m31_add(c.x, c.y), m31_add(e.x, e.y)};
            const qm31 f1 = qm31_mul_m31({m31_sub(a.x, a.y), m31_sub(b.x, b.y), m31_sub(c.x, c.y), m31_sub(e.x, e.y)}, tw);
            r = qm31_add(f0, qm31_mul(alpha, f1));
            gstore1(ft.eval[i + 1][0] + t, r.a); gstore1(ft.eval[i + 1][1] + t, r.b);
            gstore1(ft.eval[i + 1][2] + t, r.c); gstore1(ft.eval[i + 1][3] + t, r.d);
        }
        lds_only_barrier();              // every lane has read its two rows of this layer
        if (t < active) { evl[0][t] = r.a; evl[1][t] = r.b; evl[2][t] = r.c; evl[3][t] = r.d; }
        row[0] = r.a; row[1] = r.b; row[2] = r.c; row[3] = r.d;
        lds_only_barrier();
    }
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) chan[k] = d[k];
        chan[8] = n_chal;
        chan[9] = n_sent;
    }
}

struct GatherItem { const u32 *src; unsigned long long idx; };
__global__ void __launch_bounds__(256) k_gather_words(const GatherItem *__restrict__ items, u32 words, size_t total, u32 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t item = i / words, w = i % words;
    out[i] = items[item].src[items[item].idx * words + w];
}

// Proof-of-work grind: lane i of a batch tests nonce base + i; digest' = Blake2s(digest || LE64(nonce)) is one 40-byte block.
struct GrindDigest { u32 w[8]; };
__global__ void __launch_bounds__(256) k_grind(GrindDigest d, u32 pow_bits, unsigned long long base, unsigned long long count,
                                              unsigned long long *__restrict__ best) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long nonce = base + i;
    u32 m[16] = {d.w[0], d.w[1], d.w[2], d.w[3], d.w[4], d.w[5], d.w[6], d.w[7], (u32)nonce, (u32)(nonce >> 32), 0, 0, 0, 0, 0, 0};
    u32 h[8] = {IV0 ^ 0x01010020u, IV1, IV2, IV3, IV4, IV5, IV6, IV7};
    b2s_compress(h, m, 40u, true);
    u32 tz = 0;                                   // trailing zeros of the first 16 bytes as a little-endian u128
    if (h[0]) tz = __ffs(h[0]) - 1;
    else if (h[1]) tz = 32 + __ffs(h[1]) - 1;
    else if (h[2]) tz = 64 + __ffs(h[2]) - 1;
    else if (h[3]) tz = 96 + __ffs(h[3]) - 1;
    else tz = 128;
    if (tz >= pow_bits) atomicMin(best, nonce);
}

int commit_layer(u32 log_size, const uint8_t *prev, const u32 *const *cols, size_t n_cols, uint8_t *out) {
    Context &c = ctx();
    if (log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "merkle: log size out of range");
    if (!out) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null output layer");
    if ((((uintptr_t)out) & 15) || (((uintptr_t)prev) & 15)) return set_error(TSTWO_ERR_BAD_ARG, "merkle: layers must be 16-byte aligned");
    const size_t n_nodes = (size_t)1 << log_size;
    const u32 child_words = prev ? 16u : 0u;
    const u32 W = child_words + (u32)n_cols;
    unsigned blocks = ceil_div(n_nodes, 256);
    const unsigned cap_mult = (unsigned)knobs().merkle_cap;
    const unsigned cap = (unsigned)c.n_cus * cap_mult;     // workgroups per CU before lanes grid-stride over more nodes
    if (blocks > cap) blocks = cap;
    if (!prev && log_size <= 30 && (n_cols == 16 || n_cols == 32 || n_cols == 48 || n_cols == 64) && !knobs().merkle_generic) {
        HashColPtrs hp;
        for (size_t k = 0; k < n_cols; k++) hp.p[k] = cols[k];
        TreeSet one = {};
        one.t[0] = (uint4 *)out;
        switch (n_cols / 16) {
            case 1: hipLaunchKernelGGL(k_merkle_leaf_static<1>, dim3(blocks), dim3(256), 0, c.stream, hp, one, n_nodes); break;
            case 2: hipLaunchKernelGGL(k_merkle_leaf_static<2>, dim3(blocks), dim3(256), 0, c.stream, hp, one, n_nodes); break;
            case 3: hipLaunchKernelGGL(k_merkle_leaf_static<3>, dim3(blocks), dim3(256), 0, c.stream, hp, one, n_nodes); break;
            default: hipLaunchKernelGGL(k_merkle_leaf_static<4>, dim3(blocks), dim3(256), 0, c.stream, hp, one, n_nodes); break;
        }
        TSTWO_LAUNCH_CHECK();
        return TSTWO_OK;
    }
    if (!prev && log_size <= 30 && n_cols == 4 && !knobs().merkle_generic) {
        u32 *w0 = const_cast<u32 *>(cols[0]), *w1 = const_cast<u32 *>(cols[1]), *w2 = const_cast<u32 *>(cols[2]), *w3 = const_cast<u32 *>(cols[3]);
        if (g_fold_set) {
            hipLaunchKernelGGL(k_merkle_leaf4<true>, dim3(blocks), dim3(256), 0, c.stream, w0, w1, w2, w3, (uint4 *)out, n_nodes, g_fold);
            g_fold_set = false;
        } else {
            hipLaunchKernelGGL(k_merkle_leaf4<false>, dim3(blocks), dim3(256), 0, c.stream, w0, w1, w2, w3, (uint4 *)out, n_nodes, FoldSpec{});
        }
        TSTWO_LAUNCH_CHECK();
        return TSTWO_OK;
    }
    if (prev && n_cols == 0 && !knobs().merkle_generic) {
        hipLaunchKernelGGL(k_merkle_inner, dim3(blocks), dim3(256), 0, c.stream, (const uint4 *)prev, (uint4 *)out, n_nodes);
        TSTWO_LAUNCH_CHECK();
        return TSTWO_OK;
    }
    // columns are absorbed kMaxHashCols per launch; launch boundaries fall on 64-byte block boundaries
    size_t col_base = 0;
    bool first = true;
    do {
        size_t avail = n_cols - col_base;
        size_t take = avail;
        // words available to this launch must end on a block boundary unless it is the final launch
        size_t max_cols = first && prev ? (size_t)kMaxHashCols : (size_t)kMaxHashCols;
        if (take > max_cols) take = max_cols;
        bool final_launch = (col_base + take == n_cols);
        if (!final_launch) {
            // make (child_words + col_base + take) a multiple of 16
            size_t end_word = child_words + col_base + take;
            take -= end_word % 16;
        }
        HashColPtrs hp;
        for (size_t k = 0; k < take; k++) hp.p[k] = cols[col_base + k];
        LayerParams lp;
        lp.total_words = W;
        lp.w_begin = first ? 0u : (u32)(child_words + col_base);
        lp.w_end = final_launch ? (W > 0 ? W : 1u) + 16u : (u32)(child_words + col_base + take);
        lp.col_word0 = (u32)(child_words + col_base);
        lp.n_cols = (u32)take;
        lp.load_state = first ? 0u : 1u;
        lp.is_final = final_launch ? 1u : 0u;
        if (prev)
            hipLaunchKernelGGL(k_merkle_layer<true>, dim3(blocks), dim3(256), 0, c.stream, (const uint4 *)prev, hp, (uint4 *)out, n_nodes, lp);
        else
            hipLaunchKernelGGL(k_merkle_layer<false>, dim3(blocks), dim3(256), 0, c.stream, (const uint4 *)nullptr, hp, (uint4 *)out, n_nodes, lp);
        col_base += take;
        first = false;
    } while (col_base < n_cols);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

}  // namespace

namespace tstwo {
// fri.hip's commit loop: tstwo_merkle_commit(…) followed by mix_root + draw_felt on its root, the channel step riding on the tree's
// last launch when that launch is a single workgroup (every FRI layer's tree), a k_channel_mix_draw launch otherwise.
int merkle_commit_then_channel(const u32 *const *cols, const u32 *log_sizes, size_t n_cols, uint8_t *layers, u32 *chan, u32 *felt) {
    g_chan_hook = {chan, felt};
    int rc = tstwo_merkle_commit(cols, log_sizes, n_cols, layers, nullptr);
    const bool pending = g_chan_hook.chan != nullptr;
    g_chan_hook = {nullptr, nullptr};
    if (rc) return rc;
    return pending ? tstwo_channel_mix_root_draw_felt(chan, layers, felt) : TSTWO_OK;
}
// fri.hip's commit loop: fold_line of a layer INTO the leaf hashing of the next layer's tree.  new_cols (2^log_new rows each) receive
// the folded evaluation; `layers` the tree over them; then mix_root + draw_felt as in merkle_commit_then_channel.  Bit-identical to
// tstwo_fri_fold_line_dev + tstwo_merkle_commit + tstwo_channel_mix_root_draw_felt (which it falls back to when an environment
// override routes 4-column trees away from the leaf4 kernels).
int merkle_commit4_folded(const u32 *const prev[4], u32 log_new, const u32 *inv_x, const u32 *alpha_dev, u32 *const new_cols[4],
                          uint8_t *layers, u32 *chan, u32 *felt) {
    const bool fusable = !knobs().merkle_generic && !knobs().merkle_no_fused_leaf4 && !knobs().fri_no_fold_fusion && log_new >= 1 &&
                         log_new <= 30;
    const u32 lg4[4] = {log_new, log_new, log_new, log_new};
    if (!fusable) return -1;                                     // caller takes the unfused path
    g_fold = {{prev[0], prev[1], prev[2], prev[3]}, inv_x, alpha_dev};
    g_fold_set = true;
    int rc = merkle_commit_then_channel(new_cols, lg4, 4, layers, chan, felt);
    const bool consumed = !g_fold_set;
    g_fold_set = false;
    if (rc) return rc;
    return consumed ? TSTWO_OK : set_error(TSTWO_ERR_HIP, "fri commit: the fold was not carried by the leaf launch");
}
// fri.hip's commit loop hands the layers from 2^log0 <= 2^9 rows down to the last one to k_fri_tail (see there).
int launch_fri_tail(u32 *const (*eval)[4], uint8_t *const *trees, u32 n_layers, u32 log0, const u32 *itw, u32 tw_log, u32 *chan, u32 *alphas,
                    const u32 *const *pre, const u32 *pre_alpha) {
    if (n_layers == 0 || n_layers > 10 || log0 < n_layers || log0 > 9 || log0 - n_layers + 1 < 1)
        return set_error(TSTWO_ERR_BAD_ARG, "fri tail: layer range out of bounds");
    FriTail ft = {};
    for (u32 i = 0; i <= n_layers; i++)
        for (int k = 0; k < 4; k++) ft.eval[i][k] = eval[i][k];
    for (u32 i = 0; i < n_layers; i++) ft.tree[i] = (uint4 *)trees[i];
    ft.n_layers = n_layers;
    ft.log0 = log0;
    if (pre && pre_alpha) {
        if (tw_log > 31 || log0 + 1 > tw_log) return set_error(TSTWO_ERR_TWIDDLES, "Not enough twiddles!");
        for (int k = 0; k < 4; k++) ft.pre[k] = pre[k];
        ft.pre_alpha = pre_alpha;
    }
    hipLaunchKernelGGL(k_fri_tail, dim3(1), dim3(1024), 0, ctx().stream, ft, itw, tw_log, chan, alphas);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
}  // namespace tstwo

extern "C" {

size_t tstwo_merkle_layers_bytes(u32 max_log) { return 32u * (((size_t)2 << max_log) - 1); }

int tstwo_grind_blake2s(const uint8_t digest[32], u32 pow_bits, uint64_t start_nonce, uint64_t *nonce_out) {
    TSTWO_REQUIRE_READY();
    if (!digest || !nonce_out) return set_error(TSTWO_ERR_BAD_ARG, "grind: null argument");
    if (pow_bits > 128) return set_error(TSTWO_ERR_BAD_ARG, "grind: pow_bits > 128");
    Context &c = ctx();
    int rc = ensure_scratch(64);
    if (rc) return rc;
    unsigned long long *best = (unsigned long long *)c.scratch;
    GrindDigest d;
    for (int i = 0; i < 8; i++)
        d.w[i] = (u32)digest[4 * i] | ((u32)digest[4 * i + 1] << 8) | ((u32)digest[4 * i + 2] << 16) | ((u32)digest[4 * i + 3] << 24);
    unsigned long long base = start_nonce;
    const unsigned long long none = ~0ull;
    for (;;) {
        // batches grow with the expected work so easy targets return after one small launch
        unsigned long long batch = 1ull << 20;
        if (base - start_nonce >= (1ull << 22)) batch = 1ull << 26;
        if (none - base < batch) batch = none - base;
        if (batch == 0) return set_error(TSTWO_ERR_BAD_ARG, "grind: nonce space exhausted");
        TSTWO_HIP(hipMemsetAsync(best, 0xFF, sizeof(none), c.stream));      // none = all ones
        hipLaunchKernelGGL(k_grind, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, c.stream, d, pow_bits, base, batch, best);
        TSTWO_LAUNCH_CHECK();
        unsigned long long found = none;
        { int rc2 = small_d2h(&found, best, sizeof(found)); if (rc2) return rc2; }
        if (found != none) { *nonce_out = found; return TSTWO_OK; }
        base += batch;
    }
}

int tstwo_gather_words(const void *const *srcs, const uint64_t *idx, u32 words, size_t n_items, u32 *host_out) {
    TSTWO_REQUIRE_READY();
    if (n_items == 0 || words == 0) return TSTWO_OK;
    if (!srcs || !idx || !host_out) return set_error(TSTWO_ERR_BAD_ARG, "gather: null argument");
    Context &c = ctx();
    const size_t total = n_items * words;
    const size_t items_bytes = ((n_items * sizeof(GatherItem) + 63) / 64) * 64;
    int rc = ensure_scratch(items_bytes + total * sizeof(u32));
    if (rc) return rc;
    GatherItem *h = new GatherItem[n_items];
    for (size_t i = 0; i < n_items; i++) { h[i].src = (const u32 *)srcs[i]; h[i].idx = idx[i]; }
    rc = small_h2d(c.scratch, h, n_items * sizeof(GatherItem));     // stream-ordered behind whatever still reads the scratch
    delete[] h;
    if (rc) return rc;
    u32 *const page = (u32 *)result_target(total * sizeof(u32));
    u32 *d_out = page ? page : (u32 *)((unsigned char *)c.scratch + items_bytes);
    hipLaunchKernelGGL(k_gather_words, dim3(ceil_div(total, 256)), dim3(256), 0, c.stream, (const GatherItem *)c.scratch, words, total, d_out);
    TSTWO_LAUNCH_CHECK();
    if (page) {
        const void *view = nullptr;
        rc = result_wait(&view);
        if (rc) return rc;
        memcpy(host_out, view, total * sizeof(u32));
        return TSTWO_OK;
    }
    return small_d2h(host_out, d_out, total * sizeof(u32));
}

// MerkleProver.decommit (vcs/prover.ts:32-109) against device-resident layers and columns: the walk over the layers
// (which nodes are visited, which child digests / column values the verifier cannot recompute) runs here on the host
// side of the library; the selected words are then fetched with two gathers.
// The walk of one tree (vcs/prover.ts:32-109): appends the requests to the shared lists.
struct DecommitLists {
    std::vector<GatherItem> hashes, queried, witness;      // (device base, element index); digests are 8 words, values 1
};
static int plan_decommit(const uint8_t *layers, u32 max_log, const u32 *const *cols, const u32 *col_log_sizes, size_t n_cols,
                         const u32 *query_logs, const uint64_t *const *queries, const size_t *n_queries, size_t n_query_sets,
                         DecommitLists &out) {
    if (!layers || (n_cols && (!cols || !col_log_sizes)) || (n_query_sets && (!query_logs || !queries || !n_queries)))
        return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: null argument");
    if (max_log > 31) return set_error(TSTWO_ERR_BAD_ARG, "merkle: log size out of range");
    TSTWO_REQUIRE_TABLE(cols, n_cols);
    for (size_t i = 0; i < n_cols; i++)
        if (col_log_sizes[i] > max_log) return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: column larger than the tree");
    std::vector<uint64_t> last, cur;
    for (int lg = (int)max_log; lg >= 0; lg--) {
        const uint64_t *direct = nullptr;
        size_t nd = 0;
        for (size_t k = 0; k < n_query_sets; k++)
            if (query_logs[k] == (u32)lg) { direct = queries[k]; nd = n_queries[k]; }
        if (nd && !direct) return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: null argument");
        for (size_t k = 0; k < nd; k++)
            if (direct[k] >> lg) return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: query position outside its layer");
        const bool has_child = (u32)lg < max_log;
        const u32 *child_layer = has_child ? (const u32 *)(layers + 32 * (((size_t)1 << (lg + 1)) - 1)) : nullptr;
        size_t pi = 0, di = 0;
        cur.clear();
        for (;;) {
            bool any = false;
            uint64_t node = 0;
            if (pi < last.size()) { node = last[pi] >> 1; any = true; }
            if (di < nd && (!any || direct[di] < node)) { node = direct[di]; any = true; }
            if (!any) break;
            if (has_child)
                for (uint64_t k = 2 * node; k <= 2 * node + 1; k++) {
                    if (pi < last.size() && last[pi] == k) pi++;
                    else out.hashes.push_back({child_layer, k});
                }
            const bool queried = di < nd && direct[di] == node;
            if (queried) di++;
            for (size_t i = 0; i < n_cols; i++)          // columns of this layer, in the caller's order (stable sort by size)
                if (col_log_sizes[i] == (u32)lg) (queried ? out.queried : out.witness).push_back({cols[i], node});
            cur.push_back(node);
        }
        last.swap(cur);
    }
    return TSTWO_OK;
}

// One upload of all request items, two launches (8-word digests, 1-word column values), one read-back.
// (the last n_extra entries of l.hashes go to `extra` instead of hash_witness: the roots of a FRI proof's trees)
static int run_decommit(const DecommitLists &l, u32 *queried_values, uint8_t *hash_witness, u32 *column_witness, size_t n_extra = 0,
                        uint8_t *extra = nullptr) {
    const size_t nh = l.hashes.size(), nq = l.queried.size(), nw = l.witness.size(), nv = nq + nw;
    if (nh + nv == 0) return TSTWO_OK;
    Context &c = ctx();
    std::vector<GatherItem> items;
    items.reserve(nh + nv);
    items.insert(items.end(), l.hashes.begin(), l.hashes.end());
    items.insert(items.end(), l.queried.begin(), l.queried.end());
    items.insert(items.end(), l.witness.begin(), l.witness.end());
    const size_t items_bytes = ((items.size() * sizeof(GatherItem) + 63) / 64) * 64;
    const size_t out_words = 8 * nh + nv;
    int rc = ensure_scratch(items_bytes + out_words * sizeof(u32));
    if (rc) return rc;
    rc = small_h2d(c.scratch, items.data(), items.size() * sizeof(GatherItem));
    if (rc) return rc;
    const GatherItem *d_items = (const GatherItem *)c.scratch;
    // the gathers write straight into the result page when the words fit: the read-back is a synchronisation, not a copy
    u32 *const page = (u32 *)result_target(out_words * sizeof(u32));
    u32 *d_out = page ? page : (u32 *)((unsigned char *)c.scratch + items_bytes);
    if (nh) hipLaunchKernelGGL(k_gather_words, dim3(ceil_div(8 * nh, 256)), dim3(256), 0, c.stream, d_items, 8u, 8 * nh, d_out);
    if (nv) hipLaunchKernelGGL(k_gather_words, dim3(ceil_div(nv, 256)), dim3(256), 0, c.stream, d_items + nh, 1u, nv, d_out + 8 * nh);
    TSTWO_LAUNCH_CHECK();
    std::vector<u32> host(out_words);
    if (page) {
        const void *view = nullptr;
        rc = result_wait(&view);
        if (rc) return rc;
        memcpy(host.data(), view, out_words * sizeof(u32));
    } else {
        rc = small_d2h(host.data(), d_out, out_words * sizeof(u32));
        if (rc) return rc;
    }
    if (nh - n_extra) memcpy(hash_witness, host.data(), 32 * (nh - n_extra));
    if (n_extra) memcpy(extra, host.data() + 8 * (nh - n_extra), 32 * n_extra);
    if (nq) memcpy(queried_values, host.data() + 8 * nh, 4 * nq);
    if (nw) memcpy(column_witness, host.data() + 8 * nh + nq, 4 * nw);
    return TSTWO_OK;
}

int tstwo_merkle_decommit(const uint8_t *layers, u32 max_log, const u32 *const *cols, const u32 *col_log_sizes, size_t n_cols,
                          const u32 *query_logs, const uint64_t *const *queries, const size_t *n_queries, size_t n_query_sets,
                          u32 *queried_values, size_t *n_queried, uint8_t *hash_witness, size_t *n_hashes,
                          u32 *column_witness, size_t *n_column_witness) {
    TSTWO_REQUIRE_READY();
    if (!n_queried || !n_hashes || !n_column_witness) return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: null argument");
    DecommitLists l;
    int rc = plan_decommit(layers, max_log, cols, col_log_sizes, n_cols, query_logs, queries, n_queries, n_query_sets, l);
    if (rc) return rc;
    const size_t cap_q = *n_queried, cap_h = *n_hashes, cap_w = *n_column_witness;
    *n_queried = l.queried.size(); *n_hashes = l.hashes.size(); *n_column_witness = l.witness.size();
    if (l.queried.size() > cap_q || l.hashes.size() > cap_h || l.witness.size() > cap_w ||
        (l.queried.size() && !queried_values) || (l.hashes.size() && !hash_witness) || (l.witness.size() && !column_witness))
        return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: output buffer too small (required counts returned)");
    return run_decommit(l, queried_values, hash_witness, column_witness);
}

// Several trees in one round trip (every layer of a FRI proof, every tree of a commitment scheme): request r is described by
// reqs[r]; the outputs are the concatenation of the per-tree outputs in request order, counts[3r..3r+2] = (queried values,
// hashes, column witness words) of request r.  totals[3] is in/out like the single-tree call (capacities / required sizes).
int tstwo_merkle_decommit_many(const tstwo_decommit_request *reqs, size_t n_reqs, u32 *queried_values, uint8_t *hash_witness,
                               u32 *column_witness, size_t *counts, size_t totals[3]) {
    TSTWO_REQUIRE_READY();
    if ((n_reqs && (!reqs || !counts)) || !totals) return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: null argument");
    DecommitLists l;
    for (size_t r = 0; r < n_reqs; r++) {
        const size_t q0 = l.queried.size(), h0 = l.hashes.size(), w0 = l.witness.size();
        const tstwo_decommit_request &q = reqs[r];
        int rc = plan_decommit(q.layers, q.max_log, q.cols, q.col_log_sizes, q.n_cols, q.query_logs, q.queries, q.n_queries,
                               q.n_query_sets, l);
        if (rc) return rc;
        counts[3 * r] = l.queried.size() - q0;
        counts[3 * r + 1] = l.hashes.size() - h0;
        counts[3 * r + 2] = l.witness.size() - w0;
    }
    const size_t cap_q = totals[0], cap_h = totals[1], cap_w = totals[2];
    totals[0] = l.queried.size(); totals[1] = l.hashes.size(); totals[2] = l.witness.size();
    if (l.queried.size() > cap_q || l.hashes.size() > cap_h || l.witness.size() > cap_w ||
        (l.queried.size() && !queried_values) || (l.hashes.size() && !hash_witness) || (l.witness.size() && !column_witness))
        return set_error(TSTWO_ERR_BAD_ARG, "merkle decommit: output buffer too small (required counts returned)");
    return run_decommit(l, queried_values, hash_witness, column_witness);
}

// ---- FriProver.decommit_on_queries (fri.ts:768-785) in ONE call: the position logic of
// computeDecommitmentPositionsAndWitnessEvals (fri.ts:346-384) for every layer, the Merkle walk of every layer's tree
// (vcs/prover.ts:32-109, plan_decommit above) and ONE gather round trip for all witness evaluations, hash witnesses and
// column witnesses of the proof.
namespace {
// Queries.fold (queries.ts:140-158): positions >> n, de-duplicated (the input is ascending, so is the output)
void fold_queries(std::vector<uint64_t> &q, u32 n) {
    size_t w = 0;
    for (size_t i = 0; i < q.size(); i++) {
        const uint64_t v = q[i] >> n;
        if (w == 0 || q[w - 1] != v) q[w++] = v;
    }
    q.resize(w);
}
// fri.ts:346-384: every position of the folding cosets the queries touch (-> Merkle query set), and those among them the
// verifier cannot compute itself (-> witness evaluations)
void decommitment_positions(const std::vector<uint64_t> &q, u32 fold_step, std::vector<uint64_t> &positions, std::vector<uint64_t> &witness) {
    size_t i = 0;
    while (i < q.size()) {
        const uint64_t coset = q[i] >> fold_step, start = coset << fold_step;
        const size_t first = i;
        while (i < q.size() && (q[i] >> fold_step) == coset) i++;
        size_t k = first;
        for (uint64_t pos = start; pos < start + ((uint64_t)1 << fold_step); pos++) {
            positions.push_back(pos);
            if (k < i && q[k] == pos) { k++; continue; }       // the verifier can calculate this one
            witness.push_back(pos);
        }
    }
}
}  // namespace

int tstwo_fri_decommit(const tstwo_fri_layer *fri_layers, size_t n_layers, const uint64_t *queries, size_t n_queries, u32 log_domain_size,
                       u32 first_fold_step, u32 fold_step, u32 *witness_evals, uint8_t *hash_witness, u32 *column_witness, uint8_t *commitments,
                       size_t *counts, size_t totals[3]) {
    TSTWO_REQUIRE_READY();
    if ((n_layers && (!fri_layers || !counts)) || (n_queries && !queries) || !totals) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: null argument");
    if (log_domain_size > 31 || first_fold_step > 31 || fold_step > 31 || fold_step == 0) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: log size / fold step out of range");
    std::vector<uint64_t> q(queries, queries + n_queries);
    for (size_t i = 0; i < n_queries; i++) {
        if (q[i] >> log_domain_size) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: query position outside the domain");
        if (i && q[i - 1] >= q[i]) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: query positions must be ascending and distinct");
    }
    DecommitLists l;
    std::vector<GatherItem> evals;            // one item per coordinate word of a witness evaluation, layer by layer
    std::vector<std::vector<uint64_t>> pos_sets;
    for (size_t r = 0; r < n_layers; r++) {
        const tstwo_fri_layer &fl = fri_layers[r];
        if (!fl.layers || !fl.n_evals || !fl.cols || !fl.eval_logs) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: null argument");
        TSTWO_REQUIRE_TABLE(fl.cols, 4 * fl.n_evals);
        const size_t h0 = l.hashes.size(), w0 = l.witness.size(), e0 = evals.size();
        // Merkle query sets of this tree: one per distinct evaluation size (first layer: the circle evaluations folded to their
        // own size, get_query_positions_by_log_size fri.ts:470-480; inner layers: the one line evaluation)
        pos_sets.clear();
        std::vector<u32> set_logs;
        std::vector<u32> col_logs(4 * fl.n_evals);
        const u32 step = r == 0 ? first_fold_step : fold_step;
        for (size_t e = 0; e < fl.n_evals; e++) {
            const u32 lg = fl.eval_logs[e];
            if (lg > log_domain_size || lg > fl.max_log) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: evaluation larger than its tree / the query domain");
            if (r > 0 && (fl.n_evals != 1 || lg != fl.max_log)) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: an inner layer commits one line evaluation");
            for (int k = 0; k < 4; k++) col_logs[4 * e + k] = lg;
            std::vector<uint64_t> cq = q;
            if (r == 0) fold_queries(cq, log_domain_size - lg);
            else if (cq.size() && (cq.back() >> lg)) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: layer sizes do not follow the fold steps");
            std::vector<uint64_t> pos, wit;
            decommitment_positions(cq, step, pos, wit);
            if (pos.size() && (pos.back() >> lg)) return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: fold step larger than the evaluation");
            for (uint64_t p : wit)
                for (int k = 0; k < 4; k++) evals.push_back({fl.cols[4 * e + k], p});
            bool seen = false;
            for (u32 sl : set_logs) seen = seen || sl == lg;
            if (!seen) { set_logs.push_back(lg); pos_sets.push_back(std::move(pos)); }
        }
        std::vector<const uint64_t *> qp(pos_sets.size());
        std::vector<size_t> qn(pos_sets.size());
        for (size_t k = 0; k < pos_sets.size(); k++) { qp[k] = pos_sets[k].data(); qn[k] = pos_sets[k].size(); }
        const size_t q0 = l.queried.size();
        int rc = plan_decommit(fl.layers, fl.max_log, fl.cols, col_logs.data(), 4 * fl.n_evals, set_logs.data(), qp.data(), qn.data(), pos_sets.size(), l);
        if (rc) return rc;
        l.queried.resize(q0);                  // the queried values themselves are not part of a FRI layer proof (fri.ts:262-269)
        counts[3 * r] = (evals.size() - e0) / 4;
        counts[3 * r + 1] = l.hashes.size() - h0;
        counts[3 * r + 2] = l.witness.size() - w0;
        // the next layer is queried at the folded positions (fri.ts:776-783)
        fold_queries(q, r == 0 ? first_fold_step : fold_step);
    }
    const size_t cap_e = totals[0], cap_h = totals[1], cap_w = totals[2];
    totals[0] = evals.size() / 4; totals[1] = l.hashes.size(); totals[2] = l.witness.size();
    if (totals[0] > cap_e || totals[1] > cap_h || totals[2] > cap_w || (totals[0] && !witness_evals) || (totals[1] && !hash_witness) ||
        (totals[2] && !column_witness))
        return set_error(TSTWO_ERR_BAD_ARG, "fri decommit: output buffer too small (required counts returned)");
    l.queried = evals;                         // travel as the "queried" 1-word items of the shared gather
    if (commitments)                           // the trees' roots (FriLayerProof.commitment) ride along: digest 0 of every layers buffer
        for (size_t r = 0; r < n_layers; r++) l.hashes.push_back({(const u32 *)fri_layers[r].layers, 0});
    return run_decommit(l, witness_evals, hash_witness, column_witness, commitments ? n_layers : 0, commitments);
}

// Device-resident Blake2sChannel (state: 10 words = digest[8], n_challenges, n_sent).  root: 32 bytes in device memory
// (e.g. offset 0 of a tstwo_merkle_commit layers buffer) or NULL to skip the mix; felt: 4 words in device memory or NULL
// to skip the draw.  Nothing is synchronised: the next kernel on the stream can consume `felt`.
int tstwo_channel_mix_root_draw_felt(u32 *chan, const uint8_t *root, u32 *felt) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_PTRS(chan);
    if ((((uintptr_t)root) & 3) || (((uintptr_t)felt) & 3)) return set_error(TSTWO_ERR_BAD_ARG, "channel: unaligned pointer");
    hipLaunchKernelGGL(k_channel_mix_draw, dim3(1), dim3(64), 0, ctx().stream, chan, (const u32 *)root, felt, root ? 1u : 0u, felt ? 1u : 0u);
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}

int tstwo_merkle_commit_layer(u32 log_size, const uint8_t *prev, const u32 *const *cols, size_t n_cols, uint8_t *out) {
    TSTWO_REQUIRE_READY();
    if (n_cols && !cols) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null column table");
    TSTWO_REQUIRE_TABLE(cols, n_cols);
    return commit_layer(log_size, prev, cols, n_cols, out);
}

int tstwo_merkle_commit(const u32 *const *cols, const u32 *log_sizes, size_t n_cols, uint8_t *layers, uint8_t root[32]) {
    TSTWO_REQUIRE_READY();
    if (!layers) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null layers buffer");
    if (n_cols && !log_sizes) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null log size table");
    TSTWO_REQUIRE_TABLE(cols, n_cols);
    u32 max_log = 0;
    for (size_t i = 0; i < n_cols; i++) {
        if (log_sizes[i] > 31) return set_error(TSTWO_ERR_BAD_ARG, "merkle: log size out of range");
        if (log_sizes[i] > max_log) max_log = log_sizes[i];
    }
    // layers below 2^up_log nodes are latency-bound: fused multi-level launches (k_merkle_upq) instead of one per level
    const int up_log = knobs().merkle_up_log ? knobs().merkle_up_log : (knobs().merkle_up_onelane ? 15 : 16);
    // a tree of exactly 4 equally long columns with at most 2^up_log rows (every FRI layer but the first few): leaves and the
    // first 7 (or all, below 2^10 rows) levels in one launch
    if (n_cols == 4 && max_log >= 1 && (int)max_log <= up_log && log_sizes[0] == max_log && log_sizes[1] == max_log && log_sizes[2] == max_log &&
        log_sizes[3] == max_log && !knobs().merkle_no_fused_leaf4) {
        Context &c = ctx();
        u32 log_child = max_log;
        u32 *w0 = const_cast<u32 *>(cols[0]), *w1 = const_cast<u32 *>(cols[1]), *w2 = const_cast<u32 *>(cols[2]), *w3 = const_cast<u32 *>(cols[3]);
        const bool fold = g_fold_set;
        const FoldSpec fs = fold ? g_fold : FoldSpec{};
        g_fold_set = false;
        if (max_log <= 9) {
            if (fold) hipLaunchKernelGGL((k_merkle_leaf4_upq<1024, true>), dim3(1), dim3(1024), 0, c.stream, w0, w1, w2, w3, (uint4 *)layers, max_log, max_log, g_chan_hook, fs);
            else hipLaunchKernelGGL((k_merkle_leaf4_upq<1024, false>), dim3(1), dim3(1024), 0, c.stream, w0, w1, w2, w3, (uint4 *)layers, max_log, max_log, g_chan_hook, fs);
            g_chan_hook = {nullptr, nullptr};
            log_child = 0;
        } else {
            if (fold) hipLaunchKernelGGL((k_merkle_leaf4_upq<256, true>), dim3(1u << (max_log - 7)), dim3(256), 0, c.stream, w0, w1, w2, w3, (uint4 *)layers, max_log, 7u, ChanHook{nullptr, nullptr}, fs);
            else hipLaunchKernelGGL((k_merkle_leaf4_upq<256, false>), dim3(1u << (max_log - 7)), dim3(256), 0, c.stream, w0, w1, w2, w3, (uint4 *)layers, max_log, 7u, ChanHook{nullptr, nullptr}, fs);
            log_child = max_log - 7;
        }
        TSTWO_LAUNCH_CHECK();
        int rc = log_child ? commit_upper_levels(layers, log_child, 0) : TSTWO_OK;
        if (rc) return rc;
        if (root) return small_d2h(root, layers, 32);
        return TSTWO_OK;
    }
    const int sub_levels = knobs().merkle_subtree;   // measured: 2 (0.308 ms) < off (0.313) < 3 (0.326) < 4 (0.332) for 32 x 2^22
    const u32 **lc = n_cols ? new const u32 *[n_cols] : nullptr;
    const uint8_t *prev = nullptr;
    int rc = TSTWO_OK;
    int lg = (int)max_log;
    while (lg >= 0 && rc == TSTWO_OK) {   // vcs/prover.ts:24-27
        size_t k = 0;
        for (size_t i = 0; i < n_cols; i++)
            if (log_sizes[i] == (u32)lg) lc[k++] = cols[i];
        uint8_t *dst = layers + 32 * (((size_t)1 << lg) - 1);
        // layer k starts at 32*(2^k-1): 16-byte aligned for every k >= 0 when `layers` is
        if (k == 0 && prev != nullptr && lg < up_log) {
            // a run of column-free layers below lg+1: fuse them (stop above the next layer that has columns)
            int stop = lg;
            while (stop > 0) {
                bool has = false;
                for (size_t i = 0; i < n_cols; i++) has = has || log_sizes[i] == (u32)(stop - 1);
                if (has) break;
                stop--;
            }
            rc = commit_upper_levels(layers, (u32)lg + 1, (u32)stop);
            prev = layers + 32 * (((size_t)1 << stop) - 1);
            lg = stop - 1;
            continue;
        }
        if (k == 0 && prev != nullptr && sub_levels >= 2) {
            // a run of column-free layers at or above 2^up_log nodes: in-lane subtrees, up to sub_levels layers per launch
            int run = 1;
            while (run < sub_levels && lg - run >= up_log && lg - run >= 0) {
                bool has = false;
                for (size_t i = 0; i < n_cols; i++) has = has || log_sizes[i] == (u32)(lg - run);
                if (has) break;
                run++;
            }
            if (run >= 2) {
                const size_t tops = (size_t)1 << (lg - run + 1);
                const unsigned blocks = ceil_div(tops, 256);
                Context &c = ctx();
                TreeSet one = {};
                one.t[0] = (uint4 *)layers;
                switch (run) {
                    case 2:
                        if (tops % 256 == 0 && !knobs().merkle_subtree_lane_stride) hipLaunchKernelGGL(k_merkle_subtree2c, dim3(blocks), dim3(256), 0, c.stream, one, (u32)lg + 1);
                        else hipLaunchKernelGGL(k_merkle_subtree<2>, dim3(blocks), dim3(256), 0, c.stream, one, (u32)lg + 1);
                        break;
                    case 3: hipLaunchKernelGGL(k_merkle_subtree<3>, dim3(blocks), dim3(256), 0, c.stream, one, (u32)lg + 1); break;
                    default: hipLaunchKernelGGL(k_merkle_subtree<4>, dim3(blocks), dim3(256), 0, c.stream, one, (u32)lg + 1); break;
                }
                if (hipGetLastError() != hipSuccess) rc = set_error(TSTWO_ERR_HIP, "merkle: subtree kernel launch failed");
                lg -= run;
                prev = layers + 32 * (((size_t)1 << (lg + 1)) - 1);
                continue;
            }
        }
        rc = commit_layer((u32)lg, prev, lc, k, dst);
        prev = dst;
        lg--;
    }
    delete[] lc;
    if (rc) return rc;
    if (root) {
        int rc2 = small_d2h(root, layers, 32);
        if (rc2) return rc2;
    }
    return TSTWO_OK;
}


// Several trees in ONE launch sequence (a TreeVec committed together: the 8 trees of BASELINE config 5's trace on one GPU,
// pcs/prover.ts:62-64).  A tree ends in ~65 us of launches with almost nothing to do (the layers below 2^19 nodes: 1 M of a
// 32-column log-22 tree's 12.6 M compressions); committed one after the other, 8 trees pay that 8 times.  When the trees
// have ONE shape that the static leaf kernel serves (16 / 32 / 48 / 64 columns of one log size, at most 8 trees and 256
// columns in all), every launch covers all trees (blockIdx.y = tree) and the tails run side by side; any other input is
// committed tree by tree.  Bit-identical to tstwo_merkle_commit per tree (same kernels, same nodes).
int tstwo_merkle_commit_many(const tstwo_commit_request *reqs, size_t n_trees, uint8_t *roots) {
    TSTWO_REQUIRE_READY();
    if (n_trees && !reqs) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null request table");
    bool uniform = n_trees >= 2 && n_trees <= (size_t)kMaxTrees && !knobs().merkle_generic && !knobs().merkle_no_batch;
    const int up_log = knobs().merkle_up_log ? knobs().merkle_up_log : 16;
    size_t n_cols = n_trees ? reqs[0].n_cols : 0;
    u32 lg = 0;
    for (size_t r = 0; r < n_trees && uniform; r++) {
        const tstwo_commit_request &q = reqs[r];
        if (!q.layers || !q.cols || !q.log_sizes) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null argument");
        uniform = q.n_cols == n_cols && (n_cols == 16 || n_cols == 32 || n_cols == 48 || n_cols == 64) && n_cols * n_trees <= (size_t)kMaxHashCols &&
                  (((uintptr_t)q.layers) & 15) == 0;
        for (size_t k = 0; k < q.n_cols && uniform; k++) {
            if (!q.cols[k]) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer in table");
            if (r == 0 && k == 0) lg = q.log_sizes[0];
            uniform = q.log_sizes[k] == lg;
        }
    }
    uniform = uniform && lg >= (u32)up_log + 1 && lg <= 30;
    if (!uniform) {
        for (size_t r = 0; r < n_trees; r++) {
            int rc = tstwo_merkle_commit(reqs[r].cols, reqs[r].log_sizes, reqs[r].n_cols, reqs[r].layers, nullptr);
            if (rc) return rc;
        }
    } else {
        Context &c = ctx();
        HashColPtrs hp;
        TreeSet leaf = {}, ts = {};
        for (size_t r = 0; r < n_trees; r++) {
            for (size_t k = 0; k < n_cols; k++) hp.p[r * n_cols + k] = reqs[r].cols[k];
            ts.t[r] = (uint4 *)reqs[r].layers;
            leaf.t[r] = (uint4 *)(reqs[r].layers + 32 * (((size_t)1 << lg) - 1));
        }
        const size_t n_nodes = (size_t)1 << lg;
        unsigned blocks = ceil_div(n_nodes, 256);
        const unsigned cap_mult = (unsigned)knobs().merkle_cap;
        const unsigned cap = (unsigned)c.n_cus * cap_mult / (unsigned)n_trees;          // the same lanes in flight as one tree's launch
        if (blocks > cap) blocks = cap ? cap : 1;
        const dim3 grid(blocks, (unsigned)n_trees);
        switch (n_cols / 16) {
            case 1: hipLaunchKernelGGL(k_merkle_leaf_static<1>, grid, dim3(256), 0, c.stream, hp, leaf, n_nodes); break;
            case 2: hipLaunchKernelGGL(k_merkle_leaf_static<2>, grid, dim3(256), 0, c.stream, hp, leaf, n_nodes); break;
            case 3: hipLaunchKernelGGL(k_merkle_leaf_static<3>, grid, dim3(256), 0, c.stream, hp, leaf, n_nodes); break;
            default: hipLaunchKernelGGL(k_merkle_leaf_static<4>, grid, dim3(256), 0, c.stream, hp, leaf, n_nodes); break;
        }
        // column-free layers lg-1 .. up_log two per launch (in-lane subtrees), a single leftover layer on its own, then the
        // quad-lane levels: the launch sequence of tstwo_merkle_commit for a tree whose columns all sit on the leaf layer
        int cur = (int)lg - 1;
        while (cur - 1 >= up_log) {
            const size_t tops = (size_t)1 << (cur - 1);
            if (tops % 256 == 0 && !knobs().merkle_subtree_lane_stride) hipLaunchKernelGGL(k_merkle_subtree2c, dim3((unsigned)(tops / 256), (unsigned)n_trees), dim3(256), 0, c.stream, ts, (u32)cur + 1);
            else hipLaunchKernelGGL(k_merkle_subtree<2>, dim3(ceil_div(tops, 256), (unsigned)n_trees), dim3(256), 0, c.stream, ts, (u32)cur + 1);
            cur -= 2;
        }
        if (cur >= up_log) {
            unsigned b1 = ceil_div((size_t)1 << cur, 256);
            if (b1 > cap) b1 = cap ? cap : 1;
            hipLaunchKernelGGL(k_merkle_inner_set, dim3(b1, (unsigned)n_trees), dim3(256), 0, c.stream, ts, (u32)cur);
            cur -= 1;
        }
        TSTWO_LAUNCH_CHECK();
        int rc = commit_upper_levels(ts, (unsigned)n_trees, (u32)cur + 1, 0);
        if (rc) return rc;
    }
    if (roots) {
        std::vector<GatherItem> items(n_trees);
        for (size_t r = 0; r < n_trees; r++) items[r] = {(const u32 *)reqs[r].layers, 0};
        DecommitLists l;
        l.hashes = items;
        return run_decommit(l, nullptr, roots, nullptr);
    }
    return TSTWO_OK;
}

}  // extern "C"
