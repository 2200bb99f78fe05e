// microbench4.hip — heterogeneous waves on one SIMD: waves of role A run one instruction stream, waves of role B another
// (role = bit 2 of the wave index inside a 512- or 1024-thread workgroup, so every SIMD holds both roles).  Per-wave cycles
// come from s_memtime.  Question: can a "slow" (4-cycle) VALU stream of one wave overlap a "fast" (2.x-cycle) stream of another?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench4.bin.so tools/microbench4.hip && tools/microbench4.bin.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32;
typedef unsigned long long u64;
constexpr int ITERS = 1024;
#define R8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"
#define I_ADD(d) "v_add_u32 " d ", " d ", %8\n"
#define I_XOR(d) "v_xor_b32 " d ", " d ", %8\n"
#define I_MIN(d) "v_min_u32 " d ", " d ", %8\n"
#define I_ALIGN(d) "v_alignbit_b32 " d ", " d ", " d ", 7\n"
#define X64(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)
#define MIX64 R8(I_ADD) R8(I_MIN) R8(I_XOR) R8(I_ALIGN) R8(I_ADD) R8(I_MIN) R8(I_XOR) R8(I_ALIGN)
// MODE 0: A = add, B = min   1: A = add, B = add   2: A = min, B = min   3: both the mixed stream   4: A = add, B = mixed
template <int MODE>
__global__ void __launch_bounds__(1024) k_roles(u64 *cycles, u32 *out, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    u32 b = a0 * 2654435761u + 1, c = a0 ^ 0x55555555u;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool roleB = (wave >> 2) & 1;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 3 || (MODE == 4 && roleB)) {
#pragma unroll 1
        for (int i = 0; i < ITERS; i++) asm volatile(MIX64 OPS);
    } else if ((MODE == 0 && roleB) || MODE == 2) {
#pragma unroll 1
        for (int i = 0; i < ITERS; i++) asm volatile(X64(I_MIN) OPS);
    } else {
#pragma unroll 1
        for (int i = 0; i < ITERS; i++) asm volatile(X64(I_ADD) OPS);
    }
    asm volatile("s_nop 0" ::: "memory");
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
typedef void (*kern_t)(u64 *, u32 *, u32);
int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    u32 *out; u64 *cyc;
    if (hipMalloc(&out, (size_t)cus * 4 * 1024 * 4) != hipSuccess || hipMalloc(&cyc, (size_t)cus * 4 * 16 * 8) != hipSuccess) return 1;
    kern_t ks[] = {k_roles<0>, k_roles<1>, k_roles<2>, k_roles<3>, k_roles<4>};
    const char *names[] = {"A=add B=min", "A=add B=add", "A=min B=min", "A=B=mixed(add,min,xor,alignbit runs of 8)", "A=add B=mixed"};
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(ks[1], dim3(cus), dim3(1024), 0, 0, cyc, out, 1u);
    (void)hipDeviceSynchronize();
    printf("{\"note\": \"s_memtime ticks per wave64 instruction, per role, W waves per SIMD (half of each role); wall = ms for the launch\",\n");
    for (int m = 0; m < 5; m++) {
        for (int cfg = 0; cfg < 3; cfg++) {
            const int threads = cfg == 0 ? 512 : 1024, blocks_per_cu = cfg == 2 ? 2 : 1;
            const int blocks = cus * blocks_per_cu, wpb = threads / 64;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(threads), 0, 0, cyc, out, 3u);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(threads), 0, 0, cyc, out, 3u);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<u64> h((size_t)blocks * wpb);
            (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            double sa = 0, sb = 0; size_t na = 0, nb = 0;
            for (size_t i = 0; i < h.size(); i++) { if (((i % wpb) >> 2) & 1) { sb += h[i]; nb++; } else { sa += h[i]; na++; } }
            printf(" \"%s, W=%d\": {\"A_ticks_per_instr\": %.3f, \"B_ticks_per_instr\": %.3f, \"wall_ms\": %.4f, \"wall_cyc_per_instr_per_wave_2400\": %.3f}%s\n",
                   names[m], wpb / 4 * blocks_per_cu, sa / na / (ITERS * 64.0), sb / nb / (ITERS * 64.0), ms,
                   ms * 1e-3 * 2.4e9 / (ITERS * 64.0 * (wpb / 4 * blocks_per_cu)), (m == 4 && cfg == 2) ? "}" : ",");
        }
    }
    return 0;
}
