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
#include "common.h"

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

// One compression (vcs/blake2s_ref.ts:176-230): h <- F(h, m, t, last)
__device__ __forceinline__ void b2s_compress(u32 h[8], const u32 m[16], u32 t_lo, bool last) {
    u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u32 v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = IV4 ^ t_lo, v13 = IV5, v14 = last ? ~IV6 : IV6, v15 = IV7;
    // message schedule SIGMA (vcs/blake2s_ref.ts:9-20) written out so every m[] index is a literal
#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]);   B2S_G(v1, v5, v9, v13, m[s2], m[s3]);       \
    B2S_G(v2, v6, v10, v14, m[s4], m[s5]);  B2S_G(v3, v7, v11, v15, m[s6], m[s7]);      \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]);  B2S_G(v1, v6, v11, v12, m[s10], m[s11]);    \
    B2S_G(v2, v7, v8, v13, m[s12], m[s13]); B2S_G(v3, v4, v9, v14, m[s14], m[s15]);
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

template <bool HAS_PREV>
__global__ void __launch_bounds__(256) k_merkle_layer(const uint4 *__restrict__ prev, HashColPtrs cols, uint4 *__restrict__ out,
                                                     size_t n_nodes, LayerParams lp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    u32 h[8];
    if (lp.load_state) {
        uint4 a = out[2 * i], b = out[2 * i + 1];
        h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
    } else {
        h[0] = IV0 ^ 0x01010020u; h[1] = IV1; h[2] = IV2; h[3] = IV3; h[4] = IV4; h[5] = IV5; h[6] = IV6; h[7] = IV7;
    }
    const u32 W = lp.total_words;
    u32 w = lp.w_begin;
    do {
        u32 m[16];
        if (HAS_PREV && w == 0) {
            uint4 c0 = prev[4 * i], c1 = prev[4 * i + 1], c2 = prev[4 * i + 2], c3 = prev[4 * i + 3];
            m[0] = c0.x; m[1] = c0.y; m[2] = c0.z; m[3] = c0.w; m[4] = c1.x; m[5] = c1.y; m[6] = c1.z; m[7] = c1.w;
            m[8] = c2.x; m[9] = c2.y; m[10] = c2.z; m[11] = c2.w; m[12] = c3.x; m[13] = c3.y; m[14] = c3.z; m[15] = c3.w;
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                u32 wi = w + (u32)k;                 // wave-uniform
                u32 ci = wi - lp.col_word0;
                m[k] = (wi < W && ci < lp.n_cols) ? cols.p[ci][i] : 0u;
            }
        }
        u32 bytes_end = (w + 16 < W ? w + 16 : W) * 4u;   // t counter after this block
        bool last = lp.is_final && (w + 16 >= W);
        b2s_compress(h, m, bytes_end, last);
        w += 16;
    } while (w < lp.w_end && w < W);
    out[2 * i] = make_uint4(h[0], h[1], h[2], h[3]);
    out[2 * i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
}

int commit_layer(u32 log_size, const uint8_t *prev, const u32 *const *cols, size_t n_cols, uint8_t *out) {
    Context &c = ctx();
    if (log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "merkle: log size out of range");
    if (!out) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null output layer");
    if ((((uintptr_t)out) & 15) || (((uintptr_t)prev) & 15)) return set_error(TSTWO_ERR_BAD_ARG, "merkle: layers must be 16-byte aligned");
    const size_t n_nodes = (size_t)1 << log_size;
    const u32 child_words = prev ? 16u : 0u;
    const u32 W = child_words + (u32)n_cols;
    const unsigned blocks = ceil_div(n_nodes, 256);
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

extern "C" {

size_t tstwo_merkle_layers_bytes(u32 max_log) { return 32u * (((size_t)2 << max_log) - 1); }

int tstwo_merkle_commit_layer(u32 log_size, const uint8_t *prev, const u32 *const *cols, size_t n_cols, uint8_t *out) {
    TSTWO_REQUIRE_READY();
    if (n_cols && !cols) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null column table");
    return commit_layer(log_size, prev, cols, n_cols, out);
}

int tstwo_merkle_commit(const u32 *const *cols, const u32 *log_sizes, size_t n_cols, uint8_t *layers, uint8_t root[32]) {
    TSTWO_REQUIRE_READY();
    if (!layers) return set_error(TSTWO_ERR_BAD_ARG, "merkle: null layers buffer");
    u32 max_log = 0;
    for (size_t i = 0; i < n_cols; i++) {
        if (log_sizes[i] > 31) return set_error(TSTWO_ERR_BAD_ARG, "merkle: log size out of range");
        if (log_sizes[i] > max_log) max_log = log_sizes[i];
    }
    const u32 **lc = n_cols ? new const u32 *[n_cols] : nullptr;
    const uint8_t *prev = nullptr;
    int rc = TSTWO_OK;
    for (int lg = (int)max_log; lg >= 0 && rc == TSTWO_OK; lg--) {   // vcs/prover.ts:24-27
        size_t k = 0;
        for (size_t i = 0; i < n_cols; i++)
            if (log_sizes[i] == (u32)lg) lc[k++] = cols[i];
        uint8_t *dst = layers + 32 * (((size_t)1 << lg) - 1);
        // layer k starts at 32*(2^k-1): 16-byte aligned for every k >= 0 when `layers` is
        rc = commit_layer((u32)lg, prev, lc, k, dst);
        prev = dst;
    }
    delete[] lc;
    if (rc) return rc;
    if (root) {
        TSTWO_HIP(hipMemcpyAsync(root, layers, 32, hipMemcpyDeviceToHost, ctx().stream));
        TSTWO_HIP(hipStreamSynchronize(ctx().stream));
    }
    return TSTWO_OK;
}

}  // extern "C"
