// microbench5.hip — Blake2s compressions on register data only (no loads): the G functions in program order (round 1's form)
// against the priority-phased form of tstwo_amd/csrc/merkle.hip (B2S_STEP4).  What the ALUs alone allow for the Merkle kernels.
//   hipcc --offload-arch=gfx950 -O3 -I tstwo_amd/csrc -o tools/microbench5.bin.so tools/microbench5.hip && tools/microbench5.bin.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
#define TSTWO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x90)
template <int PRIO>
__device__ __forceinline__ void phase(u32 &a, u32 &b, u32 &c, u32 &d) {
    TSTWO_SCHED_FENCE();
    asm volatile("s_setprio %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(PRIO));
    TSTWO_SCHED_FENCE();
}
constexpr int kPrioHeavy = 3, kPrioLight = 0;
__device__ __forceinline__ u32 rotr32(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }
#define G(a, b, c, d, x, y) do { a = a + b + (x); d = rotr32(d ^ a, 16); c = c + d; b = rotr32(b ^ c, 12); a = a + b + (y); d = rotr32(d ^ a, 8); c = c + d; b = rotr32(b ^ c, 7); } while (0)
#define STEP_PLAIN(a0, b0, c0, d0, a1, b1, c1, d1, a2, b2, c2, d2, a3, b3, c3, d3, x0, y0, x1, y1, x2, y2, x3, y3) \
    G(a0, b0, c0, d0, x0, y0); G(a1, b1, c1, d1, x1, y1); G(a2, b2, c2, d2, x2, y2); G(a3, b3, c3, d3, x3, y3);
#define STEP_PHASED(a0, b0, c0, d0, a1, b1, c1, d1, a2, b2, c2, d2, a3, b3, c3, d3, x0, y0, x1, y1, x2, y2, x3, y3) \
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
    } while (0);
#define ROUND(STEP, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    STEP(v0, v4, v8, v12, v1, v5, v9, v13, v2, v6, v10, v14, v3, v7, v11, v15, m[s0], m[s1], m[s2], m[s3], m[s4], m[s5], m[s6], m[s7]) \
    STEP(v0, v5, v10, v15, v1, v6, v11, v12, v2, v7, v8, v13, v3, v4, v9, v14, m[s8], m[s9], m[s10], m[s11], m[s12], m[s13], m[s14], m[s15])
#define TEN_ROUNDS(STEP) \
    ROUND(STEP, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND(STEP, 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3) \
    ROUND(STEP, 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND(STEP, 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8) \
    ROUND(STEP, 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND(STEP, 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9) \
    ROUND(STEP, 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND(STEP, 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10) \
    ROUND(STEP, 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND(STEP, 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
template <bool PHASED, bool STORE = false>
__global__ void __launch_bounds__(256) k_blake2s(u32 *out, u32 seed, int n_compress) {
    u32 h[8], m[16];
    for (int j = 0; j < 8; j++) h[j] = threadIdx.x * (j + 1) + seed;
    for (int j = 0; j < 16; j++) m[j] = threadIdx.x + j * seed;
#pragma unroll 1
    for (int i = 0; i < n_compress; i++) {
        u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
        u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au, v12 = 0x510E527Fu ^ (u32)i, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu, v15 = 0x5BE0CD19u;
        if (PHASED) {
            phase<kPrioHeavy>(v0, v1, v2, v3);
            TEN_ROUNDS(STEP_PHASED)
            phase<kPrioLight>(v4, v5, v6, v7);
        } else {
            TEN_ROUNDS(STEP_PLAIN)
        }
        h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
        m[i & 15] ^= h[0];
    }
    if (STORE) {      // the leaf kernel's output: a 32-byte digest per lane
        uint4 *o = reinterpret_cast<uint4 *>(out) + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
        return;
    }
    u32 r = 0;
    for (int j = 0; j < 8; j++) r ^= h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// the same with the loop body instantiated twice (two copies of the compression in the code, as in the unrolled NBLK = 2 leaf kernel)
template <bool PHASED, bool STORE = false>
__global__ void __launch_bounds__(256) k_blake2s_x2(u32 *out, u32 seed, int n_compress) {
    u32 h[8], m[16];
    for (int j = 0; j < 8; j++) h[j] = threadIdx.x * (j + 1) + seed;
    for (int j = 0; j < 16; j++) m[j] = threadIdx.x + j * seed;
#pragma unroll 1
    for (int i0 = 0; i0 < n_compress; i0 += 2) {
        { const int i = i0;
        u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
        u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au, v12 = 0x510E527Fu ^ (u32)i, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu, v15 = 0x5BE0CD19u;
        if (PHASED) {
            phase<kPrioHeavy>(v0, v1, v2, v3);
            TEN_ROUNDS(STEP_PHASED)
            phase<kPrioLight>(v4, v5, v6, v7);
        } else {
            TEN_ROUNDS(STEP_PLAIN)
        }
        h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
        m[i & 15] ^= h[0];
            }
        { const int i = i0 + 1;
        u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
        u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au, v12 = 0x510E527Fu ^ (u32)i, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu, v15 = 0x5BE0CD19u;
        if (PHASED) {
            phase<kPrioHeavy>(v0, v1, v2, v3);
            TEN_ROUNDS(STEP_PHASED)
            phase<kPrioLight>(v4, v5, v6, v7);
        } else {
            TEN_ROUNDS(STEP_PLAIN)
        }
        h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
        m[i & 15] ^= h[0];
            }
    }
    if (STORE) {      // the leaf kernel's output: a 32-byte digest per lane
        uint4 *o = reinterpret_cast<uint4 *>(out) + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
        return;
    }
    u32 r = 0;
    for (int j = 0; j < 8; j++) r ^= h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    u32 *out;
    if (hipMalloc(&out, (size_t)32 << 22) != hipSuccess) return 1;
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_blake2s<false>, dim3(cus * 8), dim3(256), 0, 0, out, 3u, 64);
    (void)hipDeviceSynchronize();
    printf("{\"note\": \"Blake2s compressions/s on register data, G = 1e9\"");
    for (int w = 1; w <= 8; w *= 2) {
        float p = time_ms([&] { hipLaunchKernelGGL(k_blake2s<false>, dim3(cus * w), dim3(256), 0, 0, out, 3u, 256); }, 5);
        float q = time_ms([&] { hipLaunchKernelGGL(k_blake2s<true>, dim3(cus * w), dim3(256), 0, 0, out, 3u, 256); }, 5);
        printf(",\n \"%d waves per SIMD x 256 compressions\": {\"program_order_G_per_s\": %.1f, \"phased_G_per_s\": %.1f}", w,
               (double)cus * w * 256 * 256 / (p * 1e-3) / 1e9, (double)cus * w * 256 * 256 / (q * 1e-3) / 1e9);
    }
    float p = time_ms([&] { hipLaunchKernelGGL(k_blake2s<false>, dim3((1 << 22) / 256), dim3(256), 0, 0, out, 3u, 2); }, 20);
    float q = time_ms([&] { hipLaunchKernelGGL(k_blake2s<true>, dim3((1 << 22) / 256), dim3(256), 0, 0, out, 3u, 2); }, 20);
    float qs = time_ms([&] { hipLaunchKernelGGL((k_blake2s<true, true>), dim3((1 << 22) / 256), dim3(256), 0, 0, out, 3u, 2); }, 20);
    printf(",\n \"leaf-kernel shape with the 32-byte digest stores (134 MB)\": {\"phased_us\": %.1f}", qs * 1e3);
    float q2 = time_ms([&] { hipLaunchKernelGGL((k_blake2s_x2<true, true>), dim3((1 << 22) / 256), dim3(256), 0, 0, out, 3u, 2); }, 20);
    printf(",\n \"the same with two copies of the compression in the loop body\": {\"phased_us\": %.1f}", q2 * 1e3);
    printf(",\n \"leaf-kernel shape: 2^22 lanes x 2 compressions\": {\"program_order_G_per_s\": %.1f, \"phased_G_per_s\": %.1f, \"phased_us\": %.1f}\n}\n",
           (double)(1 << 22) * 2 / (p * 1e-3) / 1e9, (double)(1 << 22) * 2 / (q * 1e-3) / 1e9, q * 1e3);
    return 0;
}
