// repro_hipmallocasync.hip — standalone (no libtstwo_hip) reproduction attempt of the fault seen with HIP's stream-ordered
// allocator under the library's 32 x 2^22 evaluate + commit sequence (tools/diag_async_alloc.py).
// Same shape of work, trivial kernels: per pass allocate 32 x 16 MiB + 8 MiB with hipMallocAsync on one non-blocking
// stream, upload each column from PAGEABLE host memory (hipMemcpyAsync + hipStreamSynchronize, like tstwo_upload),
// run kernel A (fills "twiddles"), kernel B (transforms every column with them), allocate 256 MiB while B is in flight,
// run kernel C (folds the columns into it), read 32 bytes back, compare with the host's answer, hipFreeAsync everything.
//   hipcc --offload-arch=gfx950 -O2 -o repro repro_hipmallocasync.hip && ./repro [passes] [async|malloc] [poison] [pinned] [syncbig] [keep] [syncfree]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NC = 32;
constexpr size_t N = (size_t)1 << 22;

__global__ void k_tw(uint32_t *tw, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tw[i] = (uint32_t)(i * 2654435761u + 12345u);
}
struct Cols { uint32_t *p[NC]; };
__global__ void k_xform(Cols c, const uint32_t *tw, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < NC; k++) c.p[k][i] = c.p[k][i] * 3u + tw[i >> 1] + (uint32_t)k;
}
__global__ void k_fold(Cols c, uint32_t *out, size_t n) {     // out[i*8 + j] = mix of the row; also written sparsely across the big buffer
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = 0x9E3779B9u;
    for (int k = 0; k < NC; k++) h = (h ^ c.p[k][i]) * 0x85EBCA6Bu + (h >> 13);
    for (int j = 0; j < 8; j++) out[i * 8 + j] = h + j;
}

int main(int argc, char **argv) {
    int passes = argc > 1 ? atoi(argv[1]) : 8;
    bool use_async = !(argc > 2 && !strcmp(argv[2], "malloc"));
    bool poison = false, pinned = false, sync_before_big = false, keep = false, devsync_free = false;
    for (int a = 3; a < argc; a++) {
        if (!strcmp(argv[a], "poison")) poison = true;             // memset every block after allocating it
        if (!strcmp(argv[a], "pinned")) pinned = true;             // upload from page-locked host memory
        if (!strcmp(argv[a], "syncbig")) sync_before_big = true;   // drain the stream before the 256 MiB allocation
        if (!strcmp(argv[a], "keep")) keep = true;                 // release threshold = max on the default pool
        if (!strcmp(argv[a], "syncfree")) devsync_free = true;     // hipStreamSynchronize after the hipFreeAsync calls
    }
    hipStream_t s;
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (keep) {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t th = UINT64_MAX;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &th));
    }
    std::vector<uint32_t *> host(NC);
    for (int k = 0; k < NC; k++) {
        if (pinned) CK(hipHostMalloc((void **)&host[k], N * 4, hipHostMallocDefault)); else host[k] = (uint32_t *)malloc(N * 4);
        for (size_t i = 0; i < N; i++) host[k][i] = (uint32_t)(i * 40503u + k * 7919u + 1);
    }
    auto expect_row0 = [&](const int *order, uint32_t *out8) {
        uint32_t h = 0x9E3779B9u;
        for (int k = 0; k < NC; k++) {
            uint32_t v = host[order[k]][0] * 3u + 12345u + (uint32_t)k;
            h = (h ^ v) * 0x85EBCA6Bu + (h >> 13);
        }
        for (int j = 0; j < 8; j++) out8[j] = h + j;
    };
    int fails = 0;
    for (int pass = 0; pass < passes; pass++) {
        int order[NC];
        for (int k = 0; k < NC; k++) order[k] = (pass & 1) ? NC - 1 - k : k;
        Cols c;
        for (int k = 0; k < NC; k++) {
            if (use_async) CK(hipMallocAsync((void **)&c.p[k], N * 4, s)); else CK(hipMalloc((void **)&c.p[k], N * 4));
            if (poison) CK(hipMemsetAsync(c.p[k], 0xA5, N * 4, s));
            CK(hipMemcpyAsync(c.p[k], host[order[k]], N * 4, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
        }
        uint32_t *tw, *big;
        if (use_async) CK(hipMallocAsync((void **)&tw, N * 2, s)); else CK(hipMalloc((void **)&tw, N * 2));
        if (poison) CK(hipMemsetAsync(tw, 0xA5, N * 2, s));
        hipLaunchKernelGGL(k_tw, dim3((N / 2 + 255) / 256), dim3(256), 0, s, tw, N / 2);
        hipLaunchKernelGGL(k_xform, dim3((N + 255) / 256), dim3(256), 0, s, c, (const uint32_t *)tw, N);
        if (sync_before_big) CK(hipStreamSynchronize(s));
        const size_t big_bytes = 32 * ((2 * N) - 1);
        if (use_async) CK(hipMallocAsync((void **)&big, big_bytes, s)); else CK(hipMalloc((void **)&big, big_bytes));
        if (poison) CK(hipMemsetAsync(big, 0xA5, big_bytes, s));
        hipLaunchKernelGGL(k_fold, dim3((N + 255) / 256), dim3(256), 0, s, c, big, N);
        CK(hipGetLastError());
        uint32_t got[8], want[8];
        CK(hipMemcpyAsync(got, big, 32, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        expect_row0(order, want);
        bool ok = !memcmp(got, want, 32);
        // post-mortem: is column 0 still what was uploaded?
        uint32_t c0[4];
        CK(hipMemcpyAsync(c0, c.p[0], 16, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        bool untouched = c0[0] == host[order[0]][0] && c0[1] == host[order[0]][1];
        printf("pass %d (%s%s%s%s%s%s, order %s): %s%s  col0 %p tw %p big %p\n", pass, use_async ? "hipMallocAsync" : "hipMalloc", poison ? "+poison" : "",
               pinned ? "+pinned" : "", sync_before_big ? "+syncbig" : "", keep ? "+keep" : "", devsync_free ? "+syncfree" : "",
               (pass & 1) ? "rev" : "fwd", ok ? "OK" : "WRONG", untouched ? " [column 0 reads back as uploaded: the kernels did not touch it]" : "",
               (void *)c.p[0], (void *)tw, (void *)big);
        if (!ok) {
            fails++;
            // post-mortem: which stage is wrong?  (all copies below are ordinary stream-ordered D2H copies + sync)
            std::vector<uint32_t> t(N / 2), col(N);
            CK(hipMemcpyAsync(t.data(), tw, N * 2, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            size_t tw_bad = 0;
            for (size_t i = 0; i < N / 2; i++) tw_bad += t[i] != (uint32_t)(i * 2654435761u + 12345u);
            size_t up_same = 0, xf_bad = 0, first_bad = (size_t)-1;
            int worst_col = -1;
            for (int k = 0; k < NC; k++) {
                CK(hipMemcpyAsync(col.data(), c.p[k], N * 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                size_t bad = 0;
                for (size_t i = 0; i < N; i++) {
                    uint32_t want_v = host[order[k]][i] * 3u + (uint32_t)((i >> 1) * 2654435761u + 12345u) + (uint32_t)k;
                    if (col[i] != want_v) { bad++; if (first_bad == (size_t)-1) first_bad = i; }
                    up_same += col[i] == host[order[k]][i];
                }
                if (bad && worst_col < 0) worst_col = k;
                xf_bad += bad;
            }
            printf("   post-mortem: twiddle words wrong %zu / %zu; transformed column words wrong %zu / %zu (first column with errors: slot %d, first index %zu); "
                   "words still equal to the upload %zu\n", tw_bad, N / 2, xf_bad, (size_t)NC * N, worst_col, first_bad, up_same);
            printf("   got  %08x %08x ...  want %08x %08x ...\n", got[0], got[1], want[0], want[1]);
        }
        for (int k = 0; k < NC; k++) { if (use_async) CK(hipFreeAsync(c.p[k], s)); else CK(hipFree(c.p[k])); }
        if (use_async) { CK(hipFreeAsync(tw, s)); CK(hipFreeAsync(big, s)); } else { CK(hipFree(tw)); CK(hipFree(big)); }
        if (devsync_free) CK(hipStreamSynchronize(s));
    }
    printf("%d of %d passes wrong\n", fails, passes);
    return 0;
}
