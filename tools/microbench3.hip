// microbench3.hip — follow-up to microbench2: WHY do v_add/v_sub/v_xor/v_and/v_lshr/v_mov issue at ~2.5 cycles per wave64
// instruction in a pure stream but at ~4 inside the M31 butterfly?  Each kernel isolates one candidate cause (operand reuse,
// in-place destinations, opcode switches, run length of "fast" opcodes between "slow" ones).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench3.bin.so tools/microbench3.hip && tools/microbench3.bin.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
constexpr int ITERS = 2048;
#define R8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(k) : "vcc"
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                        \
        u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        u32 b = a0 * 2654435761u + 1, c = a0 ^ 0x55555555u;                                                 \
        u32 k = seed * 40503u + 7;                                                                           \
        _Pragma("unroll 1") for (int i = 0; i < ITERS; i++) { asm volatile(BODY OPS); }                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                 \
    }
#define I_ADD(d) "v_add_u32 " d ", " d ", %8\n"
#define I_ADDC(d) "v_add_u32 " d ", " d ", %9\n"
#define I_XOR(d) "v_xor_b32 " d ", " d ", %8\n"
#define I_MIN(d) "v_min_u32 " d ", " d ", %8\n"
#define I_MINC(d) "v_min_u32 " d ", " d ", %9\n"
#define I_LSHR(d) "v_lshrrev_b32 " d ", 1, " d "\n"
#define I_ALIGN(d) "v_alignbit_b32 " d ", " d ", " d ", 7\n"
#define I_ADD3(d) "v_add3_u32 " d ", " d ", %8, %9\n"
#define I_ADD_XOR(d) I_ADD(d) I_XOR(d)
#define I_ADD_ADDC(d) I_ADD(d) I_ADDC(d)
#define I_ADD_MIN(d) I_ADD(d) I_MIN(d)
#define I_ADD_MINC(d) I_ADD(d) I_MINC(d)
#define I_ADD_LSHR(d) I_ADD(d) I_LSHR(d)
#define I_XOR_ALIGN(d) I_XOR(d) I_ALIGN(d)
// 64 instructions per iteration in every kernel
KERNEL(k_add, R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD))
KERNEL(k_min, R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN))
KERNEL(k_add_xor, R8(I_ADD_XOR) R8(I_ADD_XOR) R8(I_ADD_XOR) R8(I_ADD_XOR))                // opcode switch, both fast, same src1
KERNEL(k_add_addc, R8(I_ADD_ADDC) R8(I_ADD_ADDC) R8(I_ADD_ADDC) R8(I_ADD_ADDC))           // same opcode, src1 alternates
KERNEL(k_add_lshr, R8(I_ADD_LSHR) R8(I_ADD_LSHR) R8(I_ADD_LSHR) R8(I_ADD_LSHR))
KERNEL(k_add_min, R8(I_ADD_MIN) R8(I_ADD_MIN) R8(I_ADD_MIN) R8(I_ADD_MIN))                // fast/slow alternating, same src1
KERNEL(k_add_minc, R8(I_ADD_MINC) R8(I_ADD_MINC) R8(I_ADD_MINC) R8(I_ADD_MINC))           // ... different src1
KERNEL(k_run8, R8(I_ADD) R8(I_MIN) R8(I_ADD) R8(I_MIN) R8(I_ADD) R8(I_MIN) R8(I_ADD) R8(I_MIN))     // runs of 8
KERNEL(k_run16, R8(I_ADD) R8(I_ADD) R8(I_MIN) R8(I_MIN) R8(I_ADD) R8(I_ADD) R8(I_MIN) R8(I_MIN))    // runs of 16
KERNEL(k_run32, R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN))    // runs of 32
KERNEL(k_run24_8, R8(I_ADD) R8(I_XOR) R8(I_LSHR) R8(I_MIN) R8(I_ADD) R8(I_XOR) R8(I_LSHR) R8(I_MIN)) // 24 fast, 8 slow
KERNEL(k_xor_align, R8(I_XOR_ALIGN) R8(I_XOR_ALIGN) R8(I_XOR_ALIGN) R8(I_XOR_ALIGN))      // Blake2s: d = rotr(d ^ a, r)
KERNEL(k_blake_run, R8(I_ADD) R8(I_XOR) R8(I_ALIGN) R8(I_ADD) R8(I_XOR) R8(I_ALIGN) R8(I_ADD3) R8(I_XOR)) // G-like mix in runs of 8

// not in place: d_i = e_i + f  (e_i never written), and d_i = e_i + f_i (two fresh VGPR sources)
__global__ void __launch_bounds__(256) k_add_notinplace(u32 *out, u32 seed) {
    u32 a0, a1, a2, a3, a4, a5, a6, a7;
    u32 e0 = threadIdx.x + seed, e1 = e0 * 3, e2 = e0 * 5, e3 = e0 * 7, e4 = e0 + 11, e5 = e0 + 13, e6 = e0 + 17, e7 = e0 + 19, b = e0 * 77;
#define NIP(d, e) "v_add_u32 " d ", " e ", %16\n"
#define NIP8 NIP("%0", "%8") NIP("%1", "%9") NIP("%2", "%10") NIP("%3", "%11") NIP("%4", "%12") NIP("%5", "%13") NIP("%6", "%14") NIP("%7", "%15")
#pragma unroll 1
    for (int i = 0; i < ITERS; i++)
        asm volatile(NIP8 NIP8 NIP8 NIP8 NIP8 NIP8 NIP8 NIP8
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                     : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e6), "v"(e7), "v"(b));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(256) k_add_two_fresh(u32 *out, u32 seed) {
    u32 a0, a1, a2, a3, a4, a5, a6, a7;
    u32 e0 = threadIdx.x + seed, e1 = e0 * 3, e2 = e0 * 5, e3 = e0 * 7, e4 = e0 + 11, e5 = e0 + 13, e6 = e0 + 17, e7 = e0 + 19;
#define TF(d, e, f) "v_add_u32 " d ", " e ", " f "\n"
#define TF8 TF("%0", "%8", "%9") TF("%1", "%9", "%10") TF("%2", "%10", "%11") TF("%3", "%11", "%12") TF("%4", "%12", "%13") TF("%5", "%13", "%14") TF("%6", "%14", "%15") TF("%7", "%15", "%8")
#pragma unroll 1
    for (int i = 0; i < ITERS; i++)
        asm volatile(TF8 TF8 TF8 TF8 TF8 TF8 TF8 TF8
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                     : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e6), "v"(e7));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// in place with a rotating second source: a_i += a_{i+4}  (two VGPR sources, both recently written)
KERNEL(k_add_ring, "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n"
                   "v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %6\nv_add_u32 %3, %3, %7\n"
                   "v_add_u32 %4, %4, %0\nv_add_u32 %5, %5, %1\nv_add_u32 %6, %6, %2\nv_add_u32 %7, %7, %3\n")


// ---- s_setprio experiments.  Model from microbench4: each SIMD has two VALU issue ports; port 0 takes the next instruction of
// the highest-priority (then oldest) ready wave whatever it is, port 1 a "fast" opcode of ANOTHER wave; a wave issues at most
// one VALU instruction per ~4.4 cycles.  With equal priorities the oldest wave owns port 0 and the others stall on their first
// "slow" opcode.  Raising the priority of a wave while it runs slow opcodes hands port 0 to it and leaves port 1 to waves in a
// fast run.
#define P_HI "s_setprio 3\n"
#define P_LO "s_setprio 0\n"
#define R4A(I) I("%0") I("%1") I("%2") I("%3")
#define R4B(I) I("%4") I("%5") I("%6") I("%7")
KERNEL(k_run8_prio, P_LO R8(I_ADD) P_HI R8(I_MIN) P_LO R8(I_ADD) P_HI R8(I_MIN) P_LO R8(I_ADD) P_HI R8(I_MIN) P_LO R8(I_ADD) P_HI R8(I_MIN))
KERNEL(k_run8_prio_inv, P_HI R8(I_ADD) P_LO R8(I_MIN) P_HI R8(I_ADD) P_LO R8(I_MIN) P_HI R8(I_ADD) P_LO R8(I_MIN) P_HI R8(I_ADD) P_LO R8(I_MIN))
KERNEL(k_run4_prio, P_LO R4A(I_ADD) P_HI R4A(I_MIN) P_LO R4B(I_ADD) P_HI R4B(I_MIN) P_LO R4A(I_ADD) P_HI R4A(I_MIN) P_LO R4B(I_ADD) P_HI R4B(I_MIN)
                    P_LO R4A(I_ADD) P_HI R4A(I_MIN) P_LO R4B(I_ADD) P_HI R4B(I_MIN) P_LO R4A(I_ADD) P_HI R4A(I_MIN) P_LO R4B(I_ADD) P_HI R4B(I_MIN))
KERNEL(k_run32_prio, P_LO R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) P_HI R8(I_MIN) R8(I_MIN) R8(I_MIN) R8(I_MIN))
KERNEL(k_run24_8_prio, P_LO R8(I_ADD) R8(I_XOR) R8(I_LSHR) P_HI R8(I_MIN) P_LO R8(I_ADD) R8(I_XOR) R8(I_LSHR) P_HI R8(I_MIN))
KERNEL(k_blake_run_prio, P_LO R8(I_ADD) R8(I_XOR) P_HI R8(I_ALIGN) P_LO R8(I_ADD) R8(I_XOR) P_HI R8(I_ALIGN) R8(I_ADD3) P_LO R8(I_XOR))
// the 4-way interleaved butterfly of microbench2 (11 instructions: mad, lshr, add, sub, min, add, sub, sub, min, add, min), with priorities
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
#define BF_PLAIN BF4(S_MAD) BF4(S_LSHR) BF4(S_ADDH) BF4(S_SUBP) BF4(S_MINM) BF4(S_APM) BF4(S_AMM) BF4(S_APMP) BF4(S_MINA) BF4(S_AMMP) BF4(S_MINB)
// reordered so that the two final mins are adjacent: mad | lshr add sub | min | add sub sub add | min min
#define BF_PRIO P_HI BF4(S_MAD) P_LO BF4(S_LSHR) BF4(S_ADDH) BF4(S_SUBP) P_HI BF4(S_MINM) P_LO BF4(S_APM) BF4(S_AMM) BF4(S_APMP) P_HI BF4(S_MINA) P_LO BF4(S_AMMP) P_HI BF4(S_MINB)
// variant with a better port balance: a + m - P as ONE v_add3_u32 (heavy; -P in a VGPR) instead of a light v_sub after the
// light v_add — 6 light + 5 heavy instead of 7 + 4, same count
#define S_APMP3(a, b, lo, hi) "v_add3_u32 " b ", " a ", " b ", %10\n"
#define BF4X(OP) OP("%0", "%4", "200", "201", "208") OP("%1", "%5", "202", "203", "209") OP("%2", "%6", "204", "205", "210") OP("%3", "%7", "206", "207", "211")
#define S_AMMP3(a, b, lo, hi, x) "v_add_u32 v" x ", v" hi ", %8\n"
#define S_MINB3(a, b, lo, hi, x) "v_min_u32 " b ", v" hi ", v" x "\n"
#define BF_PRIO65 P_HI BF4(S_MAD) P_LO BF4(S_LSHR) BF4(S_ADDH) BF4(S_SUBP) P_HI BF4(S_MINM) P_LO BF4(S_APM) BF4(S_AMM) BF4X(S_AMMP3) P_HI BF4(S_APMP3) BF4(S_MINA) BF4X(S_MINB3)
#define BF_KERNEL(NAME, ROUND)                                                                               \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                        \
        u32 a0 = (threadIdx.x + seed) & 0x3fffffffu, a1 = (a0 * 3) & 0x3fffffffu, a2 = (a0 * 5) & 0x3fffffffu, a3 = (a0 * 7) & 0x3fffffffu; \
        u32 b0 = (a0 + 11) & 0x3fffffffu, b1 = (a0 + 13) & 0x3fffffffu, b2 = (a0 + 17) & 0x3fffffffu, b3 = (a0 + 19) & 0x3fffffffu; \
        u32 P = 2147483647u, negP = 0x80000001u;                                                             \
        u32 t2 = ((seed * 2654435761u) % 2147483647u) * 2;                                                   \
        _Pragma("unroll 1") for (int i = 0; i < ITERS; i++) {                                                \
            asm volatile(ROUND ROUND ROUND ROUND ROUND ROUND "s_nop 0\n" "s_nop 0\n"                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)    \
                         : "v"(P), "s"(t2), "v"(negP) : "vcc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211"); \
        }                                                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                 \
    }
BF_KERNEL(k_bf_plain, BF_PLAIN)      // 6 rounds x 4 butterflies x 11 = 264 VALU instructions per iteration
BF_KERNEL(k_bf_prio, BF_PRIO)
BF_KERNEL(k_bf_prio65, BF_PRIO65)

typedef void (*kern_t)(u32 *, u32);
struct Entry { const char *name; kern_t k; double n; };
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
    return best;
}
int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    u32 *out;
    if (hipMalloc(&out, (size_t)cus * 8 * 256 * 4) != hipSuccess) return 1;
    Entry es[] = {{"runs of 8, prio: slow=3 fast=0", k_run8_prio, 64}, {"runs of 8, prio inverted", k_run8_prio_inv, 64}, {"runs of 4, prio", k_run4_prio, 64},
                  {"runs of 32, prio", k_run32_prio, 64}, {"24 fast / 8 min, prio", k_run24_8_prio, 64}, {"blake-like, prio", k_blake_run_prio, 64},
                  {"butterfly asm 4-way plain (per instruction)", k_bf_plain, 264}, {"butterfly asm 4-way prio (per instruction)", k_bf_prio, 264}, {"butterfly 6 light + 5 heavy (v_add3), prio", k_bf_prio65, 264},
                  {"add", k_add}, {"min", k_min}, {"add,xor alternating (same src1)", k_add_xor}, {"add,add alternating src1", k_add_addc},
                  {"add,lshr alternating", k_add_lshr}, {"add,min alternating same src1", k_add_min}, {"add,min alternating other src1", k_add_minc},
                  {"runs of 8 add / 8 min", k_run8}, {"runs of 16", k_run16}, {"runs of 32", k_run32}, {"24 fast (add,xor,lshr) / 8 min", k_run24_8},
                  {"xor,alignbit alternating", k_xor_align}, {"blake-like runs of 8 (add xor align add xor align add3 xor)", k_blake_run},
                  {"add not in place, shared src1", k_add_notinplace}, {"add two fresh sources", k_add_two_fresh}, {"add ring (a_i += a_i+4)", k_add_ring}};
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_add, dim3(cus * 8), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    if (argc > 1) {      // sustained run: the same kernel back to back for ~1.5 s, rate per 50 launches (clock behaviour under load)
        kern_t k = argv[1][0] == 'b' ? k_bf_prio : argv[1][0] == 'p' ? k_bf_plain : argv[1][0] == 'k' ? k_blake_run_prio : k_add;
        const double n = (argv[1][0] == 'b' || argv[1][0] == 'p') ? 264 : 64;
        printf("sustained %s: nominal cycles per instruction, 8 waves/SIMD, per 50 launches:", argv[1]);
        for (int round = 0; round < 40; round++) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k, dim3(cus * 8), dim3(256), 0, 0, out, 3u);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf(" %.2f", ms / 50 * 1e-3 * clk / ((double)ITERS * n * 8));
        }
        printf("\n");
        return 0;
    }
    printf("{\"note\": \"cycles per wave64 instruction (nominal %.0f MHz) at 1/2/4/8 waves per SIMD; 64 instructions per loop iteration\",\n", clk / 1e6);
    const int n = sizeof(es) / sizeof(es[0]);
    for (int e = 0; e < n; e++) {
        double cyc[4];
        for (int w = 0; w < 4; w++) {
            const int wps = 1 << w;
            float ms = time_kernel(es[e].k, cus * wps, out);
            cyc[w] = ms * 1e-3 * clk / ((double)ITERS * (es[e].n ? es[e].n : 64) * wps);
        }
        printf(" \"%s\": [%.2f, %.2f, %.2f, %.2f]%s\n", es[e].name, cyc[0], cyc[1], cyc[2], cyc[3], e + 1 < n ? "," : "}");
    }
    return 0;
}
