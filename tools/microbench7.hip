// microbench7.hip — does the 256 MB memory-side cache (Infinity Cache / MALL) serve a pass that reads what the previous kernel wrote?
// The two CFFT passes exchange the whole column set through memory (write 16 MiB per column, read it back); if a working set
// of S bytes written by one kernel were read back at more than HBM speed for S <= 128 MiB, column groups of that size would
// pay.  For S = 16 MiB ... 1 GiB: (a) read of a buffer the previous kernel wrote, (b) the same buffer read again (read after
// read), (c) write, (d) in-place read-modify-write (the shape of a pass), (e) out-of-place copy of one half onto the other.  16 bytes per lane, grid-stride, 2048 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb7 tools/microbench7.hip && /tmp/mb7
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;

__global__ void __launch_bounds__(256) k_write(uint4 *p, size_t n16, u32 salt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(salt, (u32)i, salt ^ (u32)i, 7u);
}
__global__ void __launch_bounds__(256) k_read(const uint4 *p, size_t n16, u32 *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = p[i];
        acc ^= v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) *sink = acc;          // (never: keeps the loads)
}
__global__ void __launch_bounds__(256) k_rmw(uint4 *p, size_t n16, u32 salt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 v = p[i];
        v.x += salt; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
        p[i] = v;
    }
}
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, u32 salt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 v = src[i];
        v.x += salt; v.y ^= v.x; v.z += v.y; v.w ^= v.z;
        dst[i] = v;
    }
}
template <class F>
static float once_us(F f) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms * 1e3f;
}
int main() {
    const size_t max_bytes = (size_t)1 << 30;
    uint4 *buf; u32 *sink;
    if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    const dim3 grid(2048), wg(256);
    // clocks up
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_rmw, grid, wg, 0, 0, buf, max_bytes / 16, (u32)i);
    (void)hipDeviceSynchronize();
    printf("{\"note\": \"GB/s of the bytes each kernel touches (rmw: read + written); median of 9\", \"rows\": [\n");
    const size_t sizes_mib[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024};
    for (size_t si = 0; si < sizeof sizes_mib / sizeof *sizes_mib; si++) {
        const size_t bytes = sizes_mib[si] << 20, n16 = bytes / 16;
        float raw[9], rar[9], wr[9], rmw[9], cp[9];
        for (int r = 0; r < 9; r++) {
            wr[r] = once_us([&] { hipLaunchKernelGGL(k_write, grid, wg, 0, 0, buf, n16, (u32)r); });
            raw[r] = once_us([&] { hipLaunchKernelGGL(k_read, grid, wg, 0, 0, buf, n16, sink); });
            rar[r] = once_us([&] { hipLaunchKernelGGL(k_read, grid, wg, 0, 0, buf, n16, sink); });
            rmw[r] = once_us([&] { hipLaunchKernelGGL(k_rmw, grid, wg, 0, 0, buf, n16, (u32)r); });
            cp[r] = once_us([&] { hipLaunchKernelGGL(k_copy, grid, wg, 0, 0, buf, buf + n16 / 2, n16 / 2, (u32)r); });      // out of place: first half -> second half
        }
        auto med = [](float *x) { for (int i = 0; i < 9; i++) for (int j = i + 1; j < 9; j++) if (x[j] < x[i]) { float t = x[i]; x[i] = x[j]; x[j] = t; } return x[4]; };
        const float w = med(wr), a = med(raw), b = med(rar), m = med(rmw), c = med(cp);
        printf("  {\"MiB\": %zu, \"write_us\": %.1f, \"write_GBps\": %.0f, \"read_after_write_us\": %.1f, \"read_after_write_GBps\": %.0f, "
               "\"read_after_read_us\": %.1f, \"read_after_read_GBps\": %.0f, \"rmw_us\": %.1f, \"rmw_GBps\": %.0f, \"copy_half_to_half_GBps\": %.0f}%s\n",
               sizes_mib[si], w, bytes / w / 1e3, a, bytes / a / 1e3, b, bytes / b / 1e3, m, 2.0 * bytes / m / 1e3, 1.0 * bytes / c / 1e3,
               si + 1 < sizeof sizes_mib / sizeof *sizes_mib ? "," : "");
    }
    printf("]}\n");
    return 0;
}
