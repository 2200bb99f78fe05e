// microbench.hip — instruction-rate / bandwidth calibration for the hot-path kernels on gfx950.
// Standalone (hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip); prints one JSON object.
// Confirms the per-op costs SURVEY.md §8(d) estimated: full-rate integer VALU, v_mad_u64_u32, the M31
// multiply / butterfly / Blake2s compression in registers, and streaming HBM copy bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../tstwo_amd/csrc/m31.cuh"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ void __launch_bounds__(256) k_add(u32 *out, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    for (int i = 0; i < ITERS; i++) {
        a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(256) k_alignbit(u32 *out, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    for (int i = 0; i < ITERS; i++) {
        a0 = __builtin_amdgcn_alignbit(a0, a1, 7); a1 = __builtin_amdgcn_alignbit(a1, a2, 7);
        a2 = __builtin_amdgcn_alignbit(a2, a3, 7); a3 = __builtin_amdgcn_alignbit(a3, a4, 7);
        a4 = __builtin_amdgcn_alignbit(a4, a5, 7); a5 = __builtin_amdgcn_alignbit(a5, a6, 7);
        a6 = __builtin_amdgcn_alignbit(a6, a7, 7); a7 = __builtin_amdgcn_alignbit(a7, a0, 7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(256) k_mad64(u32 *out, u32 seed) {
    u64 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    u32 m = seed | 1;
    for (int i = 0; i < ITERS; i++) {
        a0 = (u64)(u32)a0 * m + a1; a1 = (u64)(u32)a1 * m + a2; a2 = (u64)(u32)a2 * m + a3; a3 = (u64)(u32)a3 * m + a0;
        a0 = (u64)(u32)a0 * m + a1; a1 = (u64)(u32)a1 * m + a2; a2 = (u64)(u32)a2 * m + a3; a3 = (u64)(u32)a3 * m + a0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3) ^ (u32)((a0 ^ a1 ^ a2 ^ a3) >> 32);
}
__global__ void __launch_bounds__(256) k_m31_mul(u32 *out, u32 seed) {
    u32 a0 = (threadIdx.x + seed) & M31_P, a1 = (a0 * 3) & M31_P, a2 = (a0 * 5) & M31_P, a3 = (a0 * 7) & M31_P;
    u32 t = (seed * 2654435761u) % M31_P;
    for (int i = 0; i < ITERS; i++) {
        a0 = m31_mul(a0, t); a1 = m31_mul(a1, t); a2 = m31_mul(a2, t); a3 = m31_mul(a3, t);
        a0 = m31_mul(a0, a1); a1 = m31_mul(a1, a2); a2 = m31_mul(a2, a3); a3 = m31_mul(a3, a0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void __launch_bounds__(256) k_butterfly(u32 *out, u32 seed) {
    u32 v[8];
    for (int j = 0; j < 8; j++) v[j] = ((threadIdx.x + seed) * (2 * j + 3)) % M31_P;
    u32 t = (seed * 2654435761u) % M31_P;
    for (int i = 0; i < ITERS; i++) {
        m31_butterfly(v[0], v[4], t); m31_butterfly(v[1], v[5], t); m31_butterfly(v[2], v[6], t); m31_butterfly(v[3], v[7], t);
        m31_butterfly(v[0], v[2], t); m31_butterfly(v[1], v[3], t); m31_butterfly(v[4], v[6], t); m31_butterfly(v[5], v[7], t);
    }
    u32 r = 0;
    for (int j = 0; j < 8; j++) r ^= v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}


#define RATE_KERNEL(NAME, EXPR)                                                                       \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                 \
        u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        const u32 k = seed * 2654435761u;                                                             \
        for (int i = 0; i < ITERS; i++) {                                                             \
            a0 = EXPR(a0, a1, k); a1 = EXPR(a1, a2, k); a2 = EXPR(a2, a3, k); a3 = EXPR(a3, a4, k);   \
            a4 = EXPR(a4, a5, k); a5 = EXPR(a5, a6, k); a6 = EXPR(a6, a7, k); a7 = EXPR(a7, a0, k);   \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }
#define E_ADD3(x, y, k) ((x) + (y) + (k))
#define E_XOR(x, y, k) ((x) ^ (y))
#define E_XOR3(x, y, k) ((x) ^ (y) ^ (k))
#define E_PERM(x, y, k) __builtin_amdgcn_perm((x), (y), 0x01000302u)
#define E_LSHLOR(x, y, k) (((x) << 7) | (y))
#define E_ANDOR(x, y, k) (((x) & (k)) | (y))
#define E_BFI(x, y, k) (((x) & (k)) | ((y) & ~(k)))
#define E_MUL24(x, y, k) (((x) & 0xffffffu) * ((y) & 0xffffffu))
#define E_MULLO(x, y, k) ((x) * (y))
#define E_MULHI(x, y, k) __umulhi((x), (y))
#define E_MIN(x, y, k) min((x), (y))
#define E_LSHR(x, y, k) (((x) >> 3) + (y))
#define E_ROT16(x, y, k) __builtin_amdgcn_alignbit((x) ^ (y), (x) ^ (y), 16)
#define E_FMA(x, y, k) __float_as_uint(__builtin_fmaf(__uint_as_float(x), __uint_as_float(y), __uint_as_float(k)))
RATE_KERNEL(k_add3, E_ADD3) RATE_KERNEL(k_xor, E_XOR) RATE_KERNEL(k_xor3, E_XOR3) RATE_KERNEL(k_perm, E_PERM)
RATE_KERNEL(k_lshlor, E_LSHLOR) RATE_KERNEL(k_andor, E_ANDOR) RATE_KERNEL(k_bfi, E_BFI) RATE_KERNEL(k_mul24, E_MUL24)
RATE_KERNEL(k_mullo, E_MULLO) RATE_KERNEL(k_mulhi, E_MULHI) RATE_KERNEL(k_min, E_MIN) RATE_KERNEL(k_lshr_add, E_LSHR)
RATE_KERNEL(k_xor_rot16, E_ROT16) RATE_KERNEL(k_fma, E_FMA)

__device__ __forceinline__ u32 rotr32(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }
#define G(a, b, c, d, x, y) do { a = a + b + (x); d = rotr32(d ^ a, 16); c = c + d; b = rotr32(b ^ c, 12); a = a + b + (y); d = rotr32(d ^ a, 8); c = c + d; b = rotr32(b ^ c, 7); } while (0)
#define ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    G(v0, v4, v8, v12, m[s0], m[s1]); G(v1, v5, v9, v13, m[s2], m[s3]); G(v2, v6, v10, v14, m[s4], m[s5]); G(v3, v7, v11, v15, m[s6], m[s7]); \
    G(v0, v5, v10, v15, m[s8], m[s9]); G(v1, v6, v11, v12, m[s10], m[s11]); G(v2, v7, v8, v13, m[s12], m[s13]); G(v3, v4, v9, v14, m[s14], m[s15]);
__global__ void __launch_bounds__(256) k_blake2s(u32 *out, u32 seed, int n_compress) {
    u32 h[8], m[16];
    for (int j = 0; j < 8; j++) h[j] = threadIdx.x * (j + 1) + seed;
    for (int j = 0; j < 16; j++) m[j] = threadIdx.x + j * seed;
    for (int i = 0; i < n_compress; i++) {
        u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
        u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au, v12 = 0x510E527Fu ^ (u32)i, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu, v15 = 0x5BE0CD19u;
        ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
        ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
        ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
        ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
        ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
        h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
        m[i & 15] ^= h[0];
    }
    u32 r = 0;
    for (int j = 0; j < 8; j++) r ^= h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4 *__restrict__ in, u32 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { uint4 x = in[i]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> float time_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;      // 8 blocks of 256 threads per CU = 32 waves/CU
    u32 *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const double lanes = (double)blocks * 256;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.gcnArchName, cus, prop.clockRate / 1000);
    // warm the clocks
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 20);
    printf(", \"v_add_u32_Tops\": %.2f", lanes * ITERS * 8 / (ms * 1e-3) / 1e12);

#define RATE(NAME, LABEL, OPS)                                                                        \
    ms = time_ms([&] { hipLaunchKernelGGL(NAME, dim3(blocks), dim3(256), 0, 0, out, 3u); }, 20);       \
    printf(", \"" LABEL "\": %.2f", lanes * ITERS * 8 * OPS / (ms * 1e-3) / 1e12);
    RATE(k_fma, "v_fma_f32_Tops", 1) RATE(k_xor, "v_xor_Tops", 1) RATE(k_min, "v_min_u32_Tops", 1) RATE(k_add3, "v_add3_Tops", 1)
    RATE(k_xor3, "v_xor3_Tops", 1) RATE(k_perm, "v_perm_Tops", 1) RATE(k_lshlor, "v_lshl_or_Tops", 1) RATE(k_andor, "v_and_or_Tops", 1)
    RATE(k_bfi, "v_bfi_Tops", 1) RATE(k_mul24, "v_mul_u32_u24_Tops", 1) RATE(k_mullo, "v_mul_lo_u32_Tops", 1) RATE(k_mulhi, "v_mul_hi_u32_Tops", 1)
    RATE(k_lshr_add, "lshr_then_add_pairs_T", 1) RATE(k_xor_rot16, "xor_then_rot16_pairs_T", 1)
    ms = time_ms([&] { hipLaunchKernelGGL(k_alignbit, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 20);
    printf(", \"v_alignbit_Tops\": %.2f", lanes * ITERS * 8 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(256), 0, 0, out, 3u); }, 20);
    printf(", \"v_mad_u64_u32_Tops\": %.2f", lanes * ITERS * 8 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_m31_mul, dim3(blocks), dim3(256), 0, 0, out, 3u); }, 20);
    printf(", \"m31_mul_Tops\": %.2f", lanes * ITERS * 8 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_butterfly, dim3(blocks), dim3(256), 0, 0, out, 3u); }, 20);
    printf(", \"m31_butterfly_T_per_s\": %.3f", lanes * ITERS * 8 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL(k_blake2s, dim3(blocks), dim3(256), 0, 0, out, 3u, 256); }, 20);
    printf(", \"blake2s_compress_G_per_s\": %.2f", lanes * 256 / (ms * 1e-3) / 1e9);
    {   // same work shape as the Merkle leaf kernel: 2^22 lanes x 2 compressions, ~0.25 ms per launch
        u32 *big;
        CHECK(hipMalloc(&big, (size_t)4 << 22));
        ms = time_ms([&] { hipLaunchKernelGGL(k_blake2s, dim3((1 << 22) / 256), dim3(256), 0, 0, big, 3u, 2); }, 20);
        printf(", \"blake2s_compress_short_kernel_G_per_s\": %.2f", (double)(1 << 22) * 2 / (ms * 1e-3) / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_blake2s, dim3((1 << 21) / 256), dim3(256), 0, 0, big, 3u, 4); }, 20);
        printf(", \"blake2s_compress_short_kernel_4per_G_per_s\": %.2f", (double)(1 << 21) * 4 / (ms * 1e-3) / 1e9);
        hipFree(big);
    }
    // HBM streaming: 2 GiB buffers (>> 256 MiB Infinity Cache)
    size_t bytes = (size_t)2 << 30;
    uint4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / 16); }, 10);
    printf(", \"hbm_copy_GBps\": %.0f", 2.0 * bytes / (ms * 1e-3) / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(cus * 8), dim3(256), 0, 0, a, out, bytes / 16); }, 10);
    printf(", \"hbm_read_GBps\": %.0f", 1.0 * bytes / (ms * 1e-3) / 1e9);
    // 512 MiB working set copied in place-ish (fits 2x in nothing; partially Infinity-Cache resident)
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, ((size_t)64 << 20) / 16); }, 20);
    printf(", \"copy_64MiB_GBps\": %.0f", 2.0 * ((size_t)64 << 20) / (ms * 1e-3) / 1e9);
    printf("}\n");
    return 0;
}
