// microbench2.hip — VALU issue rates on gfx950 measured with inline assembly (nothing for the compiler to fold),
// at 1 / 2 / 4 / 8 waves per SIMD.  Replaces the C-level rate loops of tools/microbench.hip whose xor / bfi / add
// entries were partly folded by the compiler (profiles/r01_microbench.json: 114 / 76 / 59 T lane-ops/s).
//
//   hipcc --offload-arch=gfx950 -O3 -o microbench2 microbench2.hip && ./microbench2 > profiles/r02_microbench.json
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o microbench2.s microbench2.hip   (ISA check)
//
// Every kernel runs ITERS x 32 copies of ONE instruction on 8 independent destination registers (dependency distance 8),
// so cycles per wave-instruction = waves_per_simd * elapsed_cycles / (ITERS * 32).  Output: JSON, one object per
// instruction: {"cyc": [c1, c2, c4, c8]} = SIMD cycles per wave64 instruction at that many waves per SIMD, and
// "Tops" = chip-wide lane-ops/s at the best occupancy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define R8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define BODY32(I) asm volatile(R8(I) R8(I) R8(I) R8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(k) : "vcc");
#define ASM_KERNEL(NAME, I)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                        \
        u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        u32 b = a0 * 2654435761u + 1, c = a0 ^ 0x55555555u;                                                 \
        u32 k = seed * 40503u + 7;                                                                           \
        _Pragma("unroll 1") for (int i = 0; i < ITERS; i++) { BODY32(I) }                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                 \
    }
#define I_ADD(d) "v_add_u32 " d ", " d ", %8\n"
#define I_SUB(d) "v_sub_u32 " d ", " d ", %8\n"
#define I_MIN(d) "v_min_u32 " d ", " d ", %8\n"
#define I_XOR(d) "v_xor_b32 " d ", " d ", %8\n"
#define I_AND(d) "v_and_b32 " d ", " d ", %8\n"
#define I_MOV(d) "v_mov_b32 " d ", %8\n"
#define I_LSHR(d) "v_lshrrev_b32 " d ", 1, " d "\n"
#define I_ADD3(d) "v_add3_u32 " d ", " d ", %8, %9\n"
#define I_XOR3(d) "v_xor3_b32 " d ", " d ", %8, %9\n"
#define I_LSHLADD(d) "v_lshl_add_u32 " d ", " d ", 1, %8\n"
#define I_ADDLSHL(d) "v_add_lshl_u32 " d ", " d ", %8, 1\n"
#define I_ANDOR(d) "v_and_or_b32 " d ", " d ", %8, %9\n"
#define I_BFI(d) "v_bfi_b32 " d ", %8, " d ", %9\n"
#define I_BFE(d) "v_bfe_u32 " d ", " d ", 1, 31\n"
#define I_PERM(d) "v_perm_b32 " d ", " d ", %8, %9\n"
#define I_ALIGNBIT(d) "v_alignbit_b32 " d ", " d ", %8, 7\n"
#define I_MULLO(d) "v_mul_lo_u32 " d ", " d ", %8\n"
#define I_MULHI(d) "v_mul_hi_u32 " d ", " d ", %8\n"
#define I_MUL24(d) "v_mul_u32_u24 " d ", " d ", %8\n"
#define I_MAD24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n"
#define I_FMA(d) "v_fma_f32 " d ", " d ", %8, %9\n"
#define I_FMAC(d) "v_fmac_f32 " d ", %8, %9\n"
#define I_CNDMASK(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define I_ADDCO(d) "v_add_co_u32 " d ", vcc, " d ", %8\n"
#define I_ADDC(d) "v_addc_co_u32 " d ", vcc, " d ", %8, vcc\n"
#define I_ADDS(d) "v_add_u32 " d ", " d ", %10\n"
#define I_MINS(d) "v_min_u32 " d ", %10, " d "\n"
#define I_MULLOS(d) "v_mul_lo_u32 " d ", " d ", %10\n"
ASM_KERNEL(k_add, I_ADD) ASM_KERNEL(k_sub, I_SUB) ASM_KERNEL(k_min, I_MIN) ASM_KERNEL(k_xor, I_XOR) ASM_KERNEL(k_and, I_AND)
ASM_KERNEL(k_mov, I_MOV) ASM_KERNEL(k_lshr, I_LSHR) ASM_KERNEL(k_add3, I_ADD3)
ASM_KERNEL(k_lshladd, I_LSHLADD) ASM_KERNEL(k_addlshl, I_ADDLSHL) ASM_KERNEL(k_andor, I_ANDOR) ASM_KERNEL(k_bfi, I_BFI)
ASM_KERNEL(k_bfe, I_BFE) ASM_KERNEL(k_perm, I_PERM) ASM_KERNEL(k_alignbit, I_ALIGNBIT) ASM_KERNEL(k_mullo, I_MULLO)
ASM_KERNEL(k_mulhi, I_MULHI) ASM_KERNEL(k_mul24, I_MUL24) ASM_KERNEL(k_mad24, I_MAD24) ASM_KERNEL(k_fma, I_FMA)
ASM_KERNEL(k_fmac, I_FMAC) ASM_KERNEL(k_cndmask, I_CNDMASK) ASM_KERNEL(k_addco, I_ADDCO) ASM_KERNEL(k_addc, I_ADDC)
ASM_KERNEL(k_adds, I_ADDS) ASM_KERNEL(k_mins, I_MINS) ASM_KERNEL(k_mullos, I_MULLOS)

// 64-bit destination: v_mad_u64_u32 d[0:1] = s0 * s1 + d[0:1]  (the M31 multiply), 8 independent accumulators
#define R8Q(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define I_MAD64(d) "v_mad_u64_u32 " d ", vcc, %8, %9, " d "\n"
#define I_MAD64S(d) "v_mad_u64_u32 " d ", vcc, %8, %10, " d "\n"
#define I_MAD64Z(d) "v_mad_u64_u32 " d ", vcc, %8, %9, 0\n"
#define ASM_KERNEL64(NAME, I)                                                                                \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                        \
        u64 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        u32 b = (u32)a0 * 2654435761u + 1, c = (u32)a0 ^ 0x55555555u;                                       \
        u32 k = seed * 40503u + 7;                                                                           \
        _Pragma("unroll 1") for (int i = 0; i < ITERS; i++) {                                                \
            asm volatile(R8Q(I) R8Q(I) R8Q(I) R8Q(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(k) : "vcc"); \
        }                                                                                                    \
        u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)r ^ (u32)(r >> 32);                                \
    }
ASM_KERNEL64(k_mad64, I_MAD64) ASM_KERNEL64(k_mad64s, I_MAD64S) ASM_KERNEL64(k_mad64z, I_MAD64Z)

// the M31 butterfly as the CFFT kernels execute it (doubled twiddle): 11 VALU instructions, written out
//   p = b * t2 ; s = hi(p) + (lo(p) >> 1) ; m = min(s, s - P) ; a' = min(a + m, a + m - P) ; b' = min(a - m, a - m + P)
__device__ __forceinline__ u32 m31_mul_dbl(u32 x, u32 t2) {
    u64 p = (u64)x * (u64)t2;
    u32 s = (u32)(p >> 32) + ((u32)p >> 1);
    return min(s, s - 2147483647u);
}
__device__ __forceinline__ void bf_dbl(u32 &v0, u32 &v1, u32 t2) {
    u32 m = m31_mul_dbl(v1, t2);
    u32 s = v0 + m, d = v0 - m;
    v0 = min(s, s - 2147483647u);
    v1 = min(d, d + 2147483647u);
}
// compiled butterflies: 16 values, 4 layers of 8 butterflies per iteration (32 butterflies), wave-uniform twiddles
__global__ void __launch_bounds__(256) k_bf16(u32 *out, u32 seed) {
    u32 v[16];
    for (int j = 0; j < 16; j++) v[j] = ((threadIdx.x + seed) * (2 * j + 3)) % 2147483647u;
    u32 t = ((seed * 2654435761u) % 2147483647u) * 2;
#pragma unroll 1
    for (int i = 0; i < ITERS / 4; i++) {
#pragma unroll
        for (int l = 3; l >= 0; l--)
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (!(m & (1 << l))) bf_dbl(v[m], v[m | (1 << l)], t + l);
    }
    u32 r = 0;
    for (int j = 0; j < 16; j++) r ^= v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the same with the modulus in an (opaque) VGPR instead of a literal: the s - P / d + P become VGPR-VGPR VOP2 (double rate)
__device__ __forceinline__ void bf_dbl_v(u32 &v0, u32 &v1, u32 t2, u32 P) {
    u64 p = (u64)v1 * (u64)t2;
    u32 s = (u32)(p >> 32) + ((u32)p >> 1);
    u32 m = min(s, s - P);
    u32 a = v0 + m, d = v0 - m;
    v0 = min(a, a - P);
    v1 = min(d, d + P);
}
__global__ void __launch_bounds__(256) k_bf16v(u32 *out, u32 seed) {
    u32 v[16];
    for (int j = 0; j < 16; j++) v[j] = ((threadIdx.x + seed) * (2 * j + 3)) % 2147483647u;
    u32 t = ((seed * 2654435761u) % 2147483647u) * 2;
    u32 P = 2147483647u;
    asm("" : "+v"(P));
#pragma unroll 1
    for (int i = 0; i < ITERS / 4; i++) {
#pragma unroll
        for (int l = 3; l >= 0; l--)
#pragma unroll
            for (int m = 0; m < 16; m++)
                if (!(m & (1 << l))) bf_dbl_v(v[m], v[m | (1 << l)], t + l, P);
    }
    u32 r = 0;
    for (int j = 0; j < 16; j++) r ^= v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- what decides the rate inside a mixed stream?  (a) alternating fast / slow opcodes on independent registers,
// (b) a fully dependent chain of the fast opcode, (c) the butterfly hand-interleaved 4 ways (instruction k of four independent
// butterflies back to back), modulus in a VGPR: every instruction's inputs were produced >= 4 instructions earlier.
#define I_ADDMIN(d) "v_add_u32 " d ", " d ", %8\n" "v_min_u32 " d ", " d ", %9\n"
ASM_KERNEL(k_addmin, I_ADDMIN)
__global__ void __launch_bounds__(256) k_addchain(u32 *out, u32 seed) {
    u32 a0 = threadIdx.x + seed, b = a0 * 2654435761u + 1;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n" : "+v"(a0) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
// 4 butterflies (a_i = %0..%3, b_i = %4..%7); 64-bit temporaries in fixed registers v[200:207]; %8 = P (VGPR), %9 = doubled twiddle
#define BF4(OP) OP("%0", "%4", "200", "201") OP("%1", "%5", "202", "203") OP("%2", "%6", "204", "205") OP("%3", "%7", "206", "207")
#define S_MAD(a, b, lo, hi) "v_mad_u64_u32 v[" lo ":" hi "], vcc, " b ", %9, 0\n"
#define S_LSHR(a, b, lo, hi) "v_lshrrev_b32 " b ", 1, v" lo "\n"
#define S_ADDH(a, b, lo, hi) "v_add_u32 " b ", " b ", v" hi "\n"
#define S_SUBP(a, b, lo, hi) "v_sub_u32 v" lo ", " b ", %8\n"
#define S_MINM(a, b, lo, hi) "v_min_u32 " b ", " b ", v" lo "\n"
#define S_APM(a, b, lo, hi) "v_add_u32 v" lo ", " a ", " b "\n"
#define S_AMM(a, b, lo, hi) "v_sub_u32 v" hi ", " a ", " b "\n"
#define S_APMP(a, b, lo, hi) "v_sub_u32 " b ", v" lo ", %8\n"
#define S_MINA(a, b, lo, hi) "v_min_u32 " a ", v" lo ", " b "\n"
#define S_AMMP(a, b, lo, hi) "v_add_u32 " b ", v" hi ", %8\n"
#define S_MINB(a, b, lo, hi) "v_min_u32 " b ", v" hi ", " b "\n"
#define BF_ROUND BF4(S_MAD) BF4(S_LSHR) BF4(S_ADDH) BF4(S_SUBP) BF4(S_MINM) BF4(S_APM) BF4(S_AMM) BF4(S_APMP) BF4(S_MINA) BF4(S_AMMP) BF4(S_MINB)
__global__ void __launch_bounds__(256) k_bf_asm4(u32 *out, u32 seed) {
    u32 a0 = (threadIdx.x + seed) & 0x3fffffffu, a1 = (a0 * 3) & 0x3fffffffu, a2 = (a0 * 5) & 0x3fffffffu, a3 = (a0 * 7) & 0x3fffffffu;
    u32 b0 = (a0 + 11) & 0x3fffffffu, b1 = (a0 + 13) & 0x3fffffffu, b2 = (a0 + 17) & 0x3fffffffu, b3 = (a0 + 19) & 0x3fffffffu;
    u32 P = 2147483647u;
    u32 t2 = ((seed * 2654435761u) % 2147483647u) * 2;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {      // 8 butterflies per iteration (two rounds of four)
        asm volatile(BF_ROUND BF_ROUND
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
                     : "v"(P), "s"(t2) : "vcc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;
}

typedef void (*kern_t)(u32 *, u32);
struct Entry { const char *name; kern_t k; double ops_per_iter; };

static float time_kernel(kern_t k, int blocks, u32 *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;   // Hz (nominal; the chip may run lower under load)
    u32 *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry es[] = {
        {"v_add_u32", k_add, 32}, {"v_sub_u32", k_sub, 32}, {"v_min_u32", k_min, 32}, {"v_xor_b32", k_xor, 32}, {"v_and_b32", k_and, 32},
        {"v_mov_b32", k_mov, 32}, {"v_lshrrev_b32", k_lshr, 32}, {"v_add3_u32", k_add3, 32},
        {"v_lshl_add_u32", k_lshladd, 32}, {"v_add_lshl_u32", k_addlshl, 32}, {"v_and_or_b32", k_andor, 32}, {"v_bfi_b32", k_bfi, 32},
        {"v_bfe_u32", k_bfe, 32}, {"v_perm_b32", k_perm, 32}, {"v_alignbit_b32", k_alignbit, 32}, {"v_mul_lo_u32", k_mullo, 32},
        {"v_mul_hi_u32", k_mulhi, 32}, {"v_mul_u32_u24", k_mul24, 32}, {"v_mad_u32_u24", k_mad24, 32}, {"v_fma_f32", k_fma, 32},
        {"v_fmac_f32", k_fmac, 32}, {"v_cndmask_b32", k_cndmask, 32}, {"v_add_co_u32", k_addco, 32}, {"v_addc_co_u32", k_addc, 32},
        {"v_add_u32_sgpr", k_adds, 32}, {"v_min_u32_sgpr", k_mins, 32}, {"v_mul_lo_u32_sgpr", k_mullos, 32},
        {"v_mad_u64_u32", k_mad64, 32}, {"v_mad_u64_u32_sgpr", k_mad64s, 32}, {"v_mad_u64_u32_zero_addend", k_mad64z, 32},
        {"m31_butterfly_compiled(11 instr)", k_bf16, 32.0 / 4},   // 32 butterflies per 4 ITERS-units
        {"m31_butterfly_compiled_P_in_vgpr", k_bf16v, 32.0 / 4},
        {"alternating v_add_u32 / v_min_u32 (per instruction)", k_addmin, 64},
        {"v_add_u32 fully dependent chain", k_addchain, 32},
        {"m31_butterfly asm, 4-way interleaved, P in vgpr", k_bf_asm4, 8},
    };
    // warm up clocks
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_add, dim3(cus * 8), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"iters\": %d,\n \"note\": \"cyc = SIMD cycles per wave64 instruction at 1/2/4/8 waves per SIMD (nominal clock); Tops = chip lane-ops/s at the best occupancy; inline asm, 8 independent destinations\",\n \"rates\": {\n", prop.gcnArchName, cus, clk / 1e6, ITERS);
    const int n = sizeof(es) / sizeof(es[0]);
    for (int e = 0; e < n; e++) {
        double cyc[4], best_tops = 0;
        for (int w = 0; w < 4; w++) {
            const int wps = 1 << w;                 // waves per SIMD: one 256-thread block = 1 wave on each of the 4 SIMDs
            const int blocks = cus * wps;
            float ms = time_kernel(es[e].k, blocks, out);
            const double instr = (double)ITERS * es[e].ops_per_iter;          // per wave
            cyc[w] = ms * 1e-3 * clk / (instr * wps);
            const double tops = (double)blocks * 256 * instr / (ms * 1e-3) / 1e12;
            if (tops > best_tops) best_tops = tops;
        }
        printf("  \"%s\": {\"cyc\": [%.2f, %.2f, %.2f, %.2f], \"Tops\": %.2f}%s\n", es[e].name, cyc[0], cyc[1], cyc[2], cyc[3], best_tops, e + 1 < n ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
