// microbench6.hip — what a single-launch Circle FFT of ONE column of 2^20 (BASELINE config 2) would trade: the kernel boundary
// between its two passes against a grid-wide barrier inside one launch.  256 workgroups of 256 lanes (one per CU, all resident),
// each moving its own 16 KiB tile (read 4 x 16 B per lane, one multiply-add, write) per phase — the memory shape of the two
// 2^12-word passes — (a) as two launches, (b) as one launch with an agent-scope counter barrier between the phases (release
// fence + arrive, bounded sc1 poll, acquire fence: cdna_hip_programming.md §6 Guideline 16 in its counter form), (c) the same
// with the counter sharded per XCD group (blockIdx % 8) and one top counter.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb6 tools/microbench6.hip && /tmp/mb6
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;

__device__ __forceinline__ void phase_work(u32 *data, u32 wg, u32 salt) {
    uint4 *p = reinterpret_cast<uint4 *>(data) + (size_t)wg * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = p[256 * j];
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j].x = v[j].x * 3u + salt; v[j].y ^= v[j].x; v[j].z += v[j].y; v[j].w ^= v[j].z; p[256 * j] = v[j]; }
}
__global__ void __launch_bounds__(256) k_phase(u32 *data, u32 salt, u32 swap) {
    phase_work(data, swap ? (blockIdx.x ^ 85u) : blockIdx.x, salt);
}
// one monotonic counter; `target` = arrivals expected so far.  Every spin is bounded (a grid that is not resident must not hang).
__device__ __forceinline__ bool barrier_counter(unsigned *ctr, unsigned target) {
    __syncthreads();                        // (waits vmcnt(0) too: every wave's stores have been issued and acknowledged)
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
        ok = spins < (1u << 20);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok != 0;
}
__global__ void __launch_bounds__(256) k_fused_counter(u32 *data, u32 salt, unsigned *ctr, unsigned target, unsigned *fail) {
    phase_work(data, blockIdx.x, salt);
    if (!barrier_counter(ctr, target)) { if (threadIdx.x == 0) atomicAdd(fail, 1u); return; }
    phase_work(data, blockIdx.x ^ 85u, salt + 1);        // reads tiles other workgroups wrote in phase 1
}
// hierarchical: 8 group counters (blockIdx % 8 shares an XCD), the last arriver of a group arrives on the top counter and
// publishes the group's generation; everyone polls its own group's generation word
__global__ void __launch_bounds__(256) k_fused_xcd(u32 *data, u32 salt, unsigned *state /* [8 grp ctr][8 grp gen][1 top] x 16 words apart */,
                                                   unsigned gen, unsigned *fail) {
    phase_work(data, blockIdx.x, salt);
    __syncthreads();
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7u, per = gridDim.x / 8u;
        unsigned *gctr = state + 16 * g, *ggen = state + 16 * (8 + g), *top = state + 16 * 16;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned mine = __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        unsigned spins = 0;
        if (mine == gen * per) {                      // last of the group: arrive on the top counter, wait for all 8 groups, release the group
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8u && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(ggen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(ggen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
        }
        ok = spins < (1u << 20);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicAdd(fail, 1u); return; }
    phase_work(data, blockIdx.x ^ 85u, salt + 1);
}

template <class F>
static float time_us(F f, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; r++) f(20 + r);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    const int WG = 256;
    u32 *data; unsigned *state, *fail;
    if (hipMalloc(&data, (size_t)WG * 16384) != hipSuccess || hipMalloc(&state, 4096) != hipSuccess || hipMalloc(&fail, 4) != hipSuccess) return 1;
    (void)hipMemset(data, 1, (size_t)WG * 16384); (void)hipMemset(state, 0, 4096); (void)hipMemset(fail, 0, 4);
    const int reps = 2000;
    float one = time_us([&](int i) { hipLaunchKernelGGL(k_phase, dim3(WG), dim3(256), 0, 0, data, (u32)i, 0u); }, reps);
    float two = time_us([&](int i) {
        hipLaunchKernelGGL(k_phase, dim3(WG), dim3(256), 0, 0, data, (u32)i, 0u);
        hipLaunchKernelGGL(k_phase, dim3(WG), dim3(256), 0, 0, data, (u32)i + 1, 1u);
    }, reps);
    unsigned n_launch = 0;
    float fc = time_us([&](int i) {
        n_launch++;
        hipLaunchKernelGGL(k_fused_counter, dim3(WG), dim3(256), 0, 0, data, (u32)i, state + 512, n_launch * WG, fail);
    }, reps);
    unsigned gen = 0;
    float fx = time_us([&](int i) {
        gen++;
        hipLaunchKernelGGL(k_fused_xcd, dim3(WG), dim3(256), 0, 0, data, (u32)i, state, gen, fail);
    }, reps);
    unsigned nf = 0;
    (void)hipMemcpy(&nf, fail, 4, hipMemcpyDeviceToHost);
    printf("{\"note\": \"256 workgroups x 256 lanes, 16 KiB tile per workgroup and phase; us per back-to-back repetition (launch-rate bound where a single kernel is shorter than the launch interval)\",\n"
           " \"one_phase_one_launch_us\": %.2f, \"two_phases_two_launches_us\": %.2f,\n"
           " \"two_phases_one_launch_counter_barrier_us\": %.2f, \"two_phases_one_launch_xcd_barrier_us\": %.2f, \"barrier_timeouts\": %u}\n",
           one, two, fc, fx, nf);
    return 0;
}
