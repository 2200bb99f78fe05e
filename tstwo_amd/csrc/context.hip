// context.hip — lifecycle, device memory, stream and event plumbing of the C ABI (include/tstwo_hip.h).
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace tstwo {

static Context g_ctx;
static thread_local std::string g_last_error;

Context &ctx() { return g_ctx; }

#ifdef TSTWO_EXPERIMENTS
// The experiments build: every switch of common.h's Knobs from its TSTWO_* variable, read ONCE (first use), never per call.
static Knobs read_knobs() {
    Knobs k;
    auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; };
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    k.cfft_rounds = num("TSTWO_CFFT_ROUNDS", 1); if (k.cfft_rounds < 1) k.cfft_rounds = 1;
    k.cfft_maxwg = num("TSTWO_CFFT_MAXWG", 0); if (k.cfft_maxwg < 0) k.cfft_maxwg = 0;
    k.cfft_lds_pad = num("TSTWO_CFFT_LDS_PAD", 0);
    k.cfft_kb = num("TSTWO_CFFT_KB", 0); if (k.cfft_kb < 11 || k.cfft_kb > 15) k.cfft_kb = 0;
    k.cfft_ka = num("TSTWO_CFFT_KA", 0); if (k.cfft_ka < 1 || k.cfft_ka > 10) k.cfft_ka = 0;
    k.cfft_logta = num("TSTWO_CFFT_LOGTA", 0); if (k.cfft_logta < 12 || k.cfft_logta > 15) k.cfft_logta = 0;
    k.cfft_av = num("TSTWO_CFFT_AV", 0);
    k.cfft_b8 = on("TSTWO_CFFT_B8");
    k.cfft_generic = num("TSTWO_CFFT_GENERIC", 0);
    k.cfft_group = num("TSTWO_CFFT_GROUP", 0);
    k.cfft_trace = on("TSTWO_CFFT_TRACE"); k.cfft_sync = on("TSTWO_CFFT_SYNC");
    k.cfft_no_oop = on("TSTWO_CFFT_NO_OOP"); k.cfft_no_fused_extend = on("TSTWO_CFFT_NO_FUSED_EXTEND");
    k.merkle_cap = num("TSTWO_MERKLE_CAP", 32);
    k.merkle_up_log = num("TSTWO_MERKLE_UP_LOG", 0);
    k.merkle_subtree = num("TSTWO_MERKLE_SUBTREE", 2); if (k.merkle_subtree > 4) k.merkle_subtree = 4;
    k.merkle_generic = on("TSTWO_MERKLE_GENERIC"); k.merkle_up_onelane = on("TSTWO_MERKLE_UP_ONELANE");
    k.merkle_up_smallwg = on("TSTWO_MERKLE_UP_SMALLWG"); k.merkle_up_narrow_first = on("TSTWO_MERKLE_UP_NARROW_FIRST");
    k.merkle_no_fused_leaf4 = on("TSTWO_MERKLE_NO_FUSED_LEAF4"); k.merkle_no_batch = on("TSTWO_MERKLE_NO_BATCH");
    k.merkle_subtree_lane_stride = on("TSTWO_MERKLE_SUBTREE_LANE_STRIDE");
    k.fri_no_tail = on("TSTWO_FRI_NO_TAIL"); k.fri_no_fold_fusion = on("TSTWO_FRI_NO_FOLD_FUSION");
    k.fold_cap = num("TSTWO_FOLD_CAP", 64); k.fold1 = on("TSTWO_FOLD1");
    k.qinv_k = num("TSTWO_QINV_K", 0); k.qinv_montgomery = on("TSTWO_QINV_MONTGOMERY");
    k.quot_no_lazy = on("TSTWO_QUOT_NO_LAZY"); k.quot_no_pair = on("TSTWO_QUOT_NO_PAIR");
    k.quot_no_triple = on("TSTWO_QUOT_NO_TRIPLE");
    k.quot_no_rowpair = on("TSTWO_QUOT_NO_ROWPAIR");
    k.device_flag = on("TSTWO_DEVICE_FLAG");
    k.no_fast_wait = on("TSTWO_NO_FAST_WAIT");
    return k;
}
const Knobs &knobs() {
    static const Knobs k = read_knobs();
    return k;
}
#endif

int set_error(int code, const char *msg) {
    g_last_error = msg ? msg : "";
    return code;
}
int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    (void)hipGetLastError();
    return TSTWO_ERR_HIP;
}
int require_ready() {
    if (!g_ctx.ready) {
        int rc = tstwo_init(0);
        if (rc) return rc;
    }
    return TSTWO_OK;
}
int ensure_scratch(size_t bytes) {
    Context &c = g_ctx;
    if (c.scratch_bytes >= bytes) return TSTWO_OK;
    if (c.scratch) {
        TSTWO_HIP(hipStreamSynchronize(c.stream));
        TSTWO_HIP(hipFree(c.scratch));
        c.scratch = nullptr;
        c.scratch_bytes = 0;
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    TSTWO_HIP(hipMalloc((void **)&c.scratch, want));
    c.scratch_bytes = want;
    return TSTWO_OK;
}
__global__ void k_signal(u32 *seq, u32 value) { *(volatile TSTWO_GLOBAL u32 *)seq = value; }
int wait_stream() {
    Context &c = g_ctx;
    if (!c.seq_host || knobs().no_fast_wait) {
        TSTWO_HIP(hipStreamSynchronize(c.stream));
        return TSTWO_OK;
    }
    const u32 want = ++c.seq_next;
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, c.stream, c.seq_dev, want);
    if (hipGetLastError() != hipSuccess) {              // (e.g. a stream in capture mode: let the runtime report it)
        TSTWO_HIP(hipStreamSynchronize(c.stream));
        return TSTWO_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (__atomic_load_n(c.seq_host, __ATOMIC_ACQUIRE) == want) return TSTWO_OK;
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
    }
    TSTWO_HIP(hipStreamSynchronize(c.stream));           // long-running work: block instead of burning a core
    return TSTWO_OK;
}
int small_d2h(void *host_dst, const void *dev_src, size_t bytes) {
    Context &c = g_ctx;
    if (bytes == 0) return TSTWO_OK;
    if (!c.pinned || bytes > kPinnedBytes) {
        TSTWO_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c.stream));
        TSTWO_HIP(hipStreamSynchronize(c.stream));
        return TSTWO_OK;
    }
    TSTWO_HIP(hipMemcpyAsync(c.pinned, dev_src, bytes, hipMemcpyDeviceToHost, c.stream));
    if (int rc = wait_stream()) return rc;
    memcpy(host_dst, c.pinned, bytes);
    return TSTWO_OK;
}
void *result_target(size_t bytes) {
    Context &c = g_ctx;
    return (c.result_dev && bytes <= kResultBytes) ? c.result_dev : nullptr;
}
int result_wait(const void **host_view) {
    Context &c = g_ctx;
    if (int rc = wait_stream()) return rc;
    *host_view = c.result_host;
    return TSTWO_OK;
}
// A host-array upload cannot be part of a captured graph: the memcpy node would read a ring slot (or the caller's array) at
// REPLAY time, long after it has been overwritten, and the slot's event would become a captured event.  Fail loudly instead
// (include/tstwo_hip.h, "Rules while capturing").
static int refuse_if_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return TSTWO_OK; }
    if (st != hipStreamCaptureStatusNone)
        return set_error(TSTWO_ERR_BAD_ARG, "host-array upload during graph capture (column tables beyond 64 pointers, gather "
                                            "requests and quotient constants cannot be recorded)");
    return TSTWO_OK;
}
int small_h2d(void *dev_dst, const void *host_src, size_t bytes) {
    Context &c = g_ctx;
    if (bytes == 0) return TSTWO_OK;
    if (int rc = refuse_if_capturing(c.stream)) return rc;
    if (!c.up_ring || bytes > kUpSlotBytes) {
        TSTWO_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c.stream));
        TSTWO_HIP(hipStreamSynchronize(c.stream));
        return TSTWO_OK;
    }
    // Ring of page-locked slots: the copy is enqueued and the call returns; a slot is rewritten only after the event
    // recorded behind its previous copy has completed (kUpSlots uploads later — in practice long done, no wait).
    const int k = c.up_next;
    c.up_next = (k + 1) % kUpSlots;
    if (c.up_busy[k]) TSTWO_HIP(hipEventSynchronize(c.up_done[k]));
    unsigned char *slot = (unsigned char *)c.up_ring + (size_t)k * kUpSlotBytes;
    memcpy(slot, host_src, bytes);
    TSTWO_HIP(hipMemcpyAsync(dev_dst, slot, bytes, hipMemcpyHostToDevice, c.stream));
    TSTWO_HIP(hipEventRecord(c.up_done[k], c.stream));
    c.up_busy[k] = true;
    return TSTWO_OK;
}
static std::vector<const u32 *> g_coltab_last[2];       // host copy of what each device table slot holds
static u32 **g_coltab_last_base[2] = {nullptr, nullptr};
static void coltab_cache_reset() { for (int i = 0; i < 2; i++) { g_coltab_last[i].clear(); g_coltab_last_base[i] = nullptr; } }

int fill_col_table(ColPtrs &out, const u32 *const *cols, size_t n_cols, int slot) {
    Context &c = g_ctx;
    out.ext = nullptr;
    if (n_cols <= (size_t)kMaxColsPerLaunch) {
        for (size_t i = 0; i < (size_t)kMaxColsPerLaunch; i++) out.p[i] = const_cast<u32 *>(cols[i < n_cols ? i : 0]);
        return TSTWO_OK;
    }
    // A table in device memory cannot be part of a captured graph, whether or not this call would have to upload it: the launch
    // would be recorded against a slot that any later > 64-column call rewrites before the graph is replayed (a cache HIT used
    // to slip through here — only the upload itself was refused).
    if (int rc = refuse_if_capturing(c.stream)) return rc;
    if (c.coltab_cap < n_cols) {
        size_t cap = 1024;
        while (cap < n_cols) cap *= 2;
        if (c.coltab) {
            TSTWO_HIP(hipStreamSynchronize(c.stream));
            TSTWO_HIP(hipFree(c.coltab));
            c.coltab = nullptr;
            c.coltab_cap = 0;
            coltab_cache_reset();
        }
        TSTWO_HIP(hipMalloc((void **)&c.coltab, 2 * cap * sizeof(u32 *)));
        c.coltab_cap = cap;
    }
    u32 **dst = c.coltab + (size_t)(slot & 1) * c.coltab_cap;
    // the passes of one transform ask for the same table again: skip the upload when the slot already holds it
    std::vector<const u32 *> &lc = g_coltab_last[slot & 1];
    u32 ***last_base = g_coltab_last_base;
    if (!(last_base[slot & 1] == dst && lc.size() == n_cols && memcmp(lc.data(), cols, n_cols * sizeof(u32 *)) == 0)) {
        int rc = small_h2d(dst, cols, n_cols * sizeof(u32 *));  // stream-ordered behind kernels still reading the table
        if (rc) return rc;
        lc.assign(cols, cols + n_cols);
        last_base[slot & 1] = dst;
    }
    for (int i = 0; i < kMaxColsPerLaunch; i++) out.p[i] = nullptr;
    out.ext = dst;
    return TSTWO_OK;
}
int read_and_clear_flag(u32 *value) {
    Context &c = g_ctx;
    if (c.flag_host) {          // the flag is host memory the kernels write through the bus: a completion wait, no copy
        if (int rc = wait_stream()) return rc;
        *value = *(volatile u32 *)c.flag_host;
        if (*value) *(volatile u32 *)c.flag_host = 0u;      // nothing is in flight on the stream: no kernel can race this store
        return TSTWO_OK;
    }
    int rc = small_d2h(value, c.flag, sizeof(u32));
    if (rc) return rc;
    if (*value) TSTWO_HIP(hipMemsetAsync(c.flag, 0, sizeof(u32), c.stream));
    return TSTWO_OK;
}

// host-side M31 helpers for the one-time generator table (circle.ts:101-105,137)
static u32 h_mul(u32 a, u32 b) {
    u64 p = (u64)a * b;
    u64 s = (p & M31_P) + (p >> 31);
    s = (s & M31_P) + (s >> 31);
    return s >= M31_P ? (u32)(s - M31_P) : (u32)s;
}
static u32 h_add(u32 a, u32 b) { u32 s = a + b; return s >= M31_P ? s - M31_P : s; }
static u32 h_sub(u32 a, u32 b) { return a >= b ? a - b : a + M31_P - b; }

}  // namespace tstwo

using namespace tstwo;

namespace {
// Device allocator state.  tstwo_malloc / tstwo_free / tstwo_trim take `mu`, so a block may be released from any thread
// (a garbage collector's finaliser thread, say) while another thread is inside the library; every other entry point is
// thread-compatible only (include/tstwo_hip.h "Threading").
struct Pool {
    std::mutex mu;
    std::map<size_t, std::vector<void *>> free_lists;     // size class -> cached blocks (mode POOL)
    struct Live { size_t bytes; int mode; };
    std::unordered_map<void *, Live> live;                 // every block handed out by tstwo_malloc -> how to release it
    size_t cached_bytes = 0;
    int mode = TSTWO_ALLOC_POOL;
    bool poison = false, probed = false;
};
Pool g_pool;
size_t size_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    size_t p = 4096;
    while (p < bytes) p <<= 1;                            // 2^k ...
    if (bytes <= (p >> 1) + (p >> 2)) return (p >> 1) + (p >> 2);   // ... or 1.5 * 2^(k-1): at most 33 % slack
    return p;
}
// ASYNC (HIP's stream-ordered pool) is KNOWN TO RETURN WRONG DATA on ROCm 7.2 / gfx950: from the second or third pass of an
// allocate / transform / free cycle on, kernels access other memory than the copy engines, with either release threshold
// (profiles/r02_hipmallocasync_fault.txt: "[keep] 4 of 5 passes wrong"; tools/repro_hipmallocasync.hip reproduces it with no
// code of this library).  The mode is therefore refused unless TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC=1 opts in (re-checks on a
// newer runtime).
bool async_alloc_allowed() {
    const char *a = getenv("TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC");
    return a && *a && strcmp(a, "0") != 0;
}
const char *kAsyncRefused = "TSTWO_ALLOC_ASYNC refused: hipMallocAsync's pool returns wrong data on ROCm 7.2 / gfx950 "
                            "(set TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC=1 to force it)";
int probe_env() {           // TSTWO_ALLOC=pool|direct|async, TSTWO_NO_POOL=1 (= direct), TSTWO_POISON=1
    if (g_pool.probed) return TSTWO_OK;
    const char *m = getenv("TSTWO_ALLOC");
    if (m && !strcmp(m, "async") && !async_alloc_allowed()) return set_error(TSTWO_ERR_BAD_ARG, kAsyncRefused);
    g_pool.probed = true;
    if (m && !strcmp(m, "direct")) g_pool.mode = TSTWO_ALLOC_DIRECT;
    else if (m && !strcmp(m, "async")) g_pool.mode = TSTWO_ALLOC_ASYNC;
    else if (getenv("TSTWO_NO_POOL")) g_pool.mode = TSTWO_ALLOC_DIRECT;
    const char *p = getenv("TSTWO_POISON");
    g_pool.poison = p && *p && strcmp(p, "0") != 0;
    return TSTWO_OK;
}
// HIP's stream-ordered pool (opt-in only, see above): freed blocks stay mapped (release threshold = max) unless
// TSTWO_ASYNC_RELEASE=1 asks for HIP's default (threshold 0).  NEITHER setting avoids the fault described above.
int configure_async_pool() {
    hipMemPool_t pool = nullptr;
    TSTWO_HIP(hipDeviceGetDefaultMemPool(&pool, g_ctx.device));
    const char *r = getenv("TSTWO_ASYNC_RELEASE");
    uint64_t threshold = (r && *r && strcmp(r, "0") != 0) ? 0 : UINT64_MAX;
    TSTWO_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold));
    return TSTWO_OK;
}
int trim_locked() {
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    for (auto &kv : g_pool.free_lists)
        for (void *p : kv.second) (void)hipFree(p);
    g_pool.free_lists.clear();
    g_pool.cached_bytes = 0;
    return TSTWO_OK;
}
}  // namespace

extern "C" {

const char *tstwo_last_error(void) { return g_last_error.c_str(); }
#ifdef TSTWO_EXPERIMENTS
const char *tstwo_version(void) { return "tstwo_hip 0.1 (gfx950) +experiments"; }
#else
const char *tstwo_version(void) { return "tstwo_hip 0.1 (gfx950)"; }
#endif

int tstwo_device_count(int *out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (out) *out = n;
    return TSTWO_OK;
}

int tstwo_init(int device) {
    Context &c = g_ctx;
    if (c.ready && c.device == device) return TSTWO_OK;
    if (c.ready) tstwo_shutdown();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return set_error(TSTWO_ERR_HIP, "no HIP device available: libtstwo_hip has no CPU fallback");
    }
    if (device < 0 || device >= n) return set_error(TSTWO_ERR_BAD_ARG, "tstwo_init: device index out of range");
    TSTWO_HIP(hipSetDevice(device));
    TSTWO_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    c.device = device;
    hipDeviceProp_t prop;
    TSTWO_HIP(hipGetDeviceProperties(&prop, device));
    c.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // GEN * 2^k table
    cpoint tab[31];
    cpoint g = {2u, 1268011823u};
    for (int k = 0; k < 31; k++) {
        tab[k] = g;
        cpoint d = {h_sub(h_mul(g.x, g.x), h_mul(g.y, g.y)), h_add(h_mul(g.x, g.y), h_mul(g.y, g.x))};
        g = d;
    }
    TSTWO_HIP(hipMalloc((void **)&c.gen_pow2, sizeof(tab)));
    TSTWO_HIP(hipMemcpy(c.gen_pow2, tab, sizeof(tab), hipMemcpyHostToDevice));
    {   // windowed multiples: win[w][k] = k * (2^(8w) GEN), k < 256 (the identity at k = 0)
        static cpoint win[4 * 256];
        for (int w = 0; w < 4; w++) {
            const cpoint base = tab[8 * w];
            cpoint acc = {1u, 0u};
            for (int k = 0; k < 256; k++) {
                win[256 * w + k] = acc;
                const cpoint nx = {h_sub(h_mul(acc.x, base.x), h_mul(acc.y, base.y)), h_add(h_mul(acc.x, base.y), h_mul(acc.y, base.x))};
                acc = nx;
            }
        }
        TSTWO_HIP(hipMalloc((void **)&c.gen_win, sizeof(win)));
        TSTWO_HIP(hipMemcpy(c.gen_win, win, sizeof(win), hipMemcpyHostToDevice));
    }
    {   // error flag: page-locked host memory mapped into the device (read-back of a status word = 9.5 us of synchronisation instead
        // of 15-18 us with a copy); device memory if the mapping is not available
        void *h = nullptr, *d = nullptr;
        if (!knobs().device_flag && hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
            memset(h, 0, 64);
            c.flag_host = (u32 *)h;
            c.flag = (u32 *)d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            TSTWO_HIP(hipMalloc((void **)&c.flag, 64));
            TSTWO_HIP(hipMemset(c.flag, 0, 64));
        }
    }
    if (hipHostMalloc(&c.pinned, kPinnedBytes, hipHostMallocDefault) != hipSuccess) { c.pinned = nullptr; (void)hipGetLastError(); }
    {   // result page for tstwo_download_many (same mechanism as the flag)
        void *h = nullptr, *d = nullptr;
        if (!knobs().device_flag && hipHostMalloc(&h, kResultBytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
            c.result_host = h;
            c.result_dev = d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
        }
    }
    {   // sequence word of wait_stream()
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
            memset(h, 0, 64);
            c.seq_host = (u32 *)h;
            c.seq_dev = (u32 *)d;
            c.seq_next = 0;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
        }
    }
    if (hipHostMalloc(&c.up_ring, kUpSlots * kUpSlotBytes, hipHostMallocDefault) != hipSuccess) { c.up_ring = nullptr; (void)hipGetLastError(); }
    if (c.up_ring)
        for (int k = 0; k < kUpSlots; k++) TSTWO_HIP(hipEventCreateWithFlags(&c.up_done[k], hipEventDisableTiming));
    c.ready = true;
    return TSTWO_OK;
}

int tstwo_shutdown(void) {
    Context &c = g_ctx;
    if (!c.ready) return TSTWO_OK;
    (void)hipStreamSynchronize(c.stream);
    (void)tstwo_comm_destroy();          // communicator and collective stream, if any
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        (void)trim_locked();
        for (auto &kv : g_pool.live) (void)hipFree(kv.first);
        g_pool.live.clear();
    }
    if (c.gen_pow2) (void)hipFree(c.gen_pow2);
    if (c.gen_win) (void)hipFree(c.gen_win);
    if (c.flag_host) (void)hipHostFree(c.flag_host);
    else if (c.flag) (void)hipFree(c.flag);
    if (c.result_host) (void)hipHostFree(c.result_host);
    c.result_host = c.result_dev = nullptr;
    if (c.pinned) (void)hipHostFree(c.pinned);
    if (c.seq_host) (void)hipHostFree(c.seq_host);
    if (c.up_ring) {
        for (int k = 0; k < kUpSlots; k++) (void)hipEventDestroy(c.up_done[k]);
        (void)hipHostFree(c.up_ring);
    }
    if (c.coltab) (void)hipFree(c.coltab);
    coltab_cache_reset();
    if (c.scratch) (void)hipFree(c.scratch);
    if (c.copy_stream[0]) {
        for (int i = 0; i < kCopyStreams; i++) {
            (void)hipStreamSynchronize(c.copy_stream[i]);
            (void)hipEventDestroy(c.copy_done[i]);
            (void)hipStreamDestroy(c.copy_stream[i]);
        }
        (void)hipEventDestroy(c.copy_after);
    }
    if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
    c = Context();
    return TSTWO_OK;
}

int tstwo_device_name(char *buf, size_t buflen) {
    TSTWO_REQUIRE_READY();
    hipDeviceProp_t prop;
    TSTWO_HIP(hipGetDeviceProperties(&prop, g_ctx.device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return TSTWO_OK;
}

int tstwo_set_stream(void *hip_stream) {
    TSTWO_REQUIRE_READY();
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    g_ctx.stream = hip_stream ? (hipStream_t)hip_stream : g_ctx.own_stream;
    return TSTWO_OK;
}

// ---- hipGraph capture of a launch sequence on the library's stream (launch-bound loops such as the small layers of a FRI
// commit): everything enqueued between begin and end is recorded instead of executed; the instantiated graph replays the
// whole sequence with one call.  Only asynchronous entry points may be called while capturing (anything that reads back to
// the host synchronises and fails the capture); allocations must be served by the caching allocator (warm it with one eager
// run of the same sequence) and every buffer the sequence touches must stay alive as long as the graph is replayed.
int tstwo_graph_begin_capture(void) {
    TSTWO_REQUIRE_READY();
    TSTWO_HIP(hipStreamBeginCapture(g_ctx.stream, hipStreamCaptureModeRelaxed));
    return TSTWO_OK;
}
int tstwo_graph_end_capture(void **graph_exec) {
    TSTWO_REQUIRE_READY();
    if (!graph_exec) return set_error(TSTWO_ERR_BAD_ARG, "graph: null out pointer");
    *graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    TSTWO_HIP(hipStreamEndCapture(g_ctx.stream, &graph));
    if (!graph) return set_error(TSTWO_ERR_HIP, "graph: capture produced no graph");
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate");
    *graph_exec = (void *)exec;
    return TSTWO_OK;
}
int tstwo_graph_launch(void *graph_exec) {
    TSTWO_REQUIRE_READY();
    if (!graph_exec) return set_error(TSTWO_ERR_BAD_ARG, "graph: null handle");
    TSTWO_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, g_ctx.stream));
    return TSTWO_OK;
}
int tstwo_graph_destroy(void *graph_exec) {
    if (!graph_exec) return TSTWO_OK;
    TSTWO_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return TSTWO_OK;
}

int tstwo_sync(void) {
    TSTWO_REQUIRE_READY();
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    for (int i = 0; i < kCopyStreams; i++)          // copies of tstwo_upload_async that nothing fenced: "everything the library was asked to do is done"
        if (g_ctx.copy_pending[i]) { TSTWO_HIP(hipStreamSynchronize(g_ctx.copy_stream[i])); g_ctx.copy_pending[i] = false; }
    return TSTWO_OK;
}

// Caching device allocator.  Wrappers allocate a result column per operation (value semantics of the reference's
// Column API) and a raw hipMalloc / hipFree pair per call costs more than most kernels, so freed blocks are kept in
// size-class free lists and handed out again.  Every use of device memory by this library is ordered on one stream
// (kernels, memsets, copies), so a block released after operation k and reused by operation k+1 needs no host
// synchronisation: k+1 cannot start before k has finished.  Blocks return to HIP at tstwo_shutdown / tstwo_trim.

int tstwo_trim(void) {
    if (!g_ctx.ready) return TSTWO_OK;
    std::lock_guard<std::mutex> lock(g_pool.mu);
    return trim_locked();
}

int tstwo_set_alloc_mode(int mode) {
    TSTWO_REQUIRE_READY();
    const int base = mode & 0xF;
    if (base != TSTWO_ALLOC_POOL && base != TSTWO_ALLOC_DIRECT && base != TSTWO_ALLOC_ASYNC)
        return set_error(TSTWO_ERR_BAD_ARG, "tstwo_set_alloc_mode: unknown mode");
    if (base == TSTWO_ALLOC_ASYNC && !async_alloc_allowed()) return set_error(TSTWO_ERR_BAD_ARG, kAsyncRefused);
    std::lock_guard<std::mutex> lock(g_pool.mu);
    g_pool.probed = true;
    int rc = trim_locked();            // cached blocks go back to HIP; live blocks remember the mode they came from
    if (rc) return rc;
    g_pool.mode = base;
    g_pool.poison = (mode & TSTWO_ALLOC_POISON) != 0;
    if (base == TSTWO_ALLOC_ASYNC) return configure_async_pool();
    return TSTWO_OK;
}

int tstwo_malloc(void **dev, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (!dev) return set_error(TSTWO_ERR_BAD_ARG, "tstwo_malloc: null out pointer");
    *dev = nullptr;
    std::lock_guard<std::mutex> lock(g_pool.mu);
    if (!g_pool.probed) {
        if (int rc_env = probe_env()) return rc_env;
        if (g_pool.mode == TSTWO_ALLOC_ASYNC) { int rc_cfg = configure_async_pool(); if (rc_cfg) return rc_cfg; }
    }
    if (bytes == 0) bytes = 16;
    const int mode = g_pool.mode;
    const size_t cls = mode == TSTWO_ALLOC_POOL ? size_class(bytes) : bytes;
    void *p = nullptr;
    if (mode == TSTWO_ALLOC_POOL) {
        auto it = g_pool.free_lists.find(cls);
        if (it != g_pool.free_lists.end() && !it->second.empty()) {
            p = it->second.back();
            it->second.pop_back();
            g_pool.cached_bytes -= cls;
        }
    }
    if (!p) {
        hipError_t e = mode == TSTWO_ALLOC_ASYNC ? hipMallocAsync(&p, cls, g_ctx.stream) : hipMalloc(&p, cls);
        if (e != hipSuccess && g_pool.cached_bytes) {           // out of memory with blocks cached: give them back and retry
            (void)hipGetLastError();
            int rc = trim_locked();
            if (rc) return rc;
            e = mode == TSTWO_ALLOC_ASYNC ? hipMallocAsync(&p, cls, g_ctx.stream) : hipMalloc(&p, cls);
        }
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    }
    g_pool.live[p] = {cls, mode};
    // Debugging aid: a block handed out full of 0xA5 turns any read of memory the library never wrote into a
    // deterministic mismatch (a recycled block otherwise holds whatever the previous owner left, often the right data).
    if (g_pool.poison) TSTWO_HIP(hipMemsetAsync(p, 0xA5, cls, g_ctx.stream));
    *dev = p;
    return TSTWO_OK;
}
int tstwo_free(void *dev) {
    if (!dev) return TSTWO_OK;
    TSTWO_REQUIRE_READY();
    std::lock_guard<std::mutex> lock(g_pool.mu);
    auto it = g_pool.live.find(dev);
    if (it == g_pool.live.end()) {           // not ours (or already released): the safe thing is a synchronous free
        TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
        TSTWO_HIP(hipFree(dev));
        return TSTWO_OK;
    }
    const Pool::Live l = it->second;
    g_pool.live.erase(it);
    if (l.mode == TSTWO_ALLOC_POOL && g_pool.mode == TSTWO_ALLOC_POOL) {
        // Reuse is stream-ordered: the next owner's first access is enqueued behind every operation that used the block.
        g_pool.free_lists[l.bytes].push_back(dev);
        g_pool.cached_bytes += l.bytes;
        return TSTWO_OK;
    }
    if (l.mode == TSTWO_ALLOC_ASYNC) {
        TSTWO_HIP(hipFreeAsync(dev, g_ctx.stream));
        return TSTWO_OK;
    }
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    TSTWO_HIP(hipFree(dev));
    return TSTWO_OK;
}
int tstwo_upload(void *dev_dst, const void *host_src, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (bytes == 0) return TSTWO_OK;
    if (bytes <= kPinnedBytes) return small_h2d(dev_dst, host_src, bytes);
    TSTWO_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, g_ctx.stream));
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    return TSTWO_OK;
}
// ---- Host hand-over beside the kernels (the boundary the TypeScript prover crosses: createBaseFieldColumn(data),
// backend/index.ts:20-31, backend/cpu/index.ts:85-90).  tstwo_upload is synchronous and, from pageable memory, staged by the
// runtime (34-39 GB/s measured: 19 steps' worth of time for the config-5 trace).  From page-locked memory — the caller's own
// buffers registered with tstwo_host_register, or buffers from tstwo_host_alloc — a copy is one DMA at link rate, and issued on
// the library's copy stream it runs BESIDE the kernels of the main stream: upload of column group k+1 under the transform of group k.
// Ordering: a copy starts only after everything enqueued on the main stream BEFORE the tstwo_upload_async call (so a
// destination block the allocator has just recycled is no longer in use), and nothing enqueued on the main stream after
// tstwo_upload_fence() starts before the copies issued so far have landed.  The host never blocks in either; tstwo_upload_wait
// blocks until the copies are done (the source may then be reused).
int tstwo_host_register(void *host, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (!host || !bytes) return set_error(TSTWO_ERR_BAD_ARG, "host_register: null or empty range");
    TSTWO_HIP(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return TSTWO_OK;
}
int tstwo_host_unregister(void *host) {
    TSTWO_REQUIRE_READY();
    if (!host) return set_error(TSTWO_ERR_BAD_ARG, "host_unregister: null pointer");
    TSTWO_HIP(hipHostUnregister(host));
    return TSTWO_OK;
}
int tstwo_host_alloc(void **host, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (!host) return set_error(TSTWO_ERR_BAD_ARG, "host_alloc: null output");
    *host = nullptr;
    TSTWO_HIP(hipHostMalloc(host, bytes ? bytes : 16, hipHostMallocDefault));
    return TSTWO_OK;
}
int tstwo_host_free(void *host) {
    TSTWO_REQUIRE_READY();
    if (!host) return TSTWO_OK;
    TSTWO_HIP(hipHostFree(host));
    return TSTWO_OK;
}
int tstwo_upload_async(void *dev_dst, const void *host_src, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (bytes == 0) return TSTWO_OK;
    TSTWO_REQUIRE_PTRS(dev_dst, host_src);
    Context &c = g_ctx;
    if (int rc = refuse_if_capturing(c.stream)) return rc;
    if (!c.copy_stream[0]) {
        for (int i = 0; i < kCopyStreams; i++) {
            TSTWO_HIP(hipStreamCreateWithFlags(&c.copy_stream[i], hipStreamNonBlocking));
            TSTWO_HIP(hipEventCreateWithFlags(&c.copy_done[i], hipEventDisableTiming));
        }
        TSTWO_HIP(hipEventCreateWithFlags(&c.copy_after, hipEventDisableTiming));
    }
    const int i = c.copy_next;
    c.copy_next = (i + 1) % kCopyStreams;
    TSTWO_HIP(hipEventRecord(c.copy_after, c.stream));
    TSTWO_HIP(hipStreamWaitEvent(c.copy_stream[i], c.copy_after, 0));
    TSTWO_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c.copy_stream[i]));
    TSTWO_HIP(hipEventRecord(c.copy_done[i], c.copy_stream[i]));
    c.copy_pending[i] = true;
    return TSTWO_OK;
}
int tstwo_upload_fence(void) {
    TSTWO_REQUIRE_READY();
    Context &c = g_ctx;
    for (int i = 0; i < kCopyStreams; i++)
        if (c.copy_pending[i]) TSTWO_HIP(hipStreamWaitEvent(c.stream, c.copy_done[i], 0));
    return TSTWO_OK;
}
int tstwo_upload_wait(void) {
    TSTWO_REQUIRE_READY();
    Context &c = g_ctx;
    for (int i = 0; i < kCopyStreams; i++)
        if (c.copy_pending[i]) { TSTWO_HIP(hipStreamSynchronize(c.copy_stream[i])); c.copy_pending[i] = false; }
    return TSTWO_OK;
}
int tstwo_download(void *host_dst, const void *dev_src, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (bytes == 0) return TSTWO_OK;
    if (bytes <= kPinnedBytes) return small_d2h(host_dst, dev_src, bytes);
    TSTWO_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
    TSTWO_HIP(hipStreamSynchronize(g_ctx.stream));
    return TSTWO_OK;
}
// Several small device buffers in ONE round trip.  A read-back costs a synchronisation whatever its size (~10 us; 25 us with a
// copy behind it), so six 40-byte .. 4 KiB pieces fetched one by one — the end of a FRI commit: channel state, the last layer's
// four coordinate columns, a twiddle slice — cost more than the layer kernels in front of them.  One launch packs up to 16
// pieces (pointers by value in the kernel arguments: no upload) into the mapped result page; then one stream synchronisation.
namespace {
struct Pieces { const u32 *src[16]; u32 off[16]; u32 words[16]; };
__global__ void __launch_bounds__(256) k_pack_pieces(Pieces p, u32 *dst) {
    const u32 k = blockIdx.y, n = p.words[k];
    const u32 *__restrict__ s = p.src[k];
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) gstore1(dst, p.off[k] + i, gload1(s, i));
}
}  // namespace
int tstwo_download_many(const void *const *srcs, const size_t *n_bytes, size_t n_pieces, void *host_out) {
    TSTWO_REQUIRE_READY();
    if (n_pieces == 0) return TSTWO_OK;
    if (!srcs || !n_bytes || !host_out) return set_error(TSTWO_ERR_BAD_ARG, "download_many: null argument");
    size_t total = 0;
    for (size_t i = 0; i < n_pieces; i++) {
        if (n_bytes[i] % 4 || (n_bytes[i] && (!srcs[i] || ((uintptr_t)srcs[i] & 3))))
            return set_error(TSTWO_ERR_BAD_ARG, "download_many: pieces are whole, 4-byte aligned words");
        total += n_bytes[i];
    }
    if (total == 0) return TSTWO_OK;
    Context &c = g_ctx;
    if (!c.result_dev || total > kResultBytes) {          // large or no mapped page: piece by piece
        size_t off = 0;
        for (size_t i = 0; i < n_pieces; i++) {
            int rc = tstwo_download((unsigned char *)host_out + off, srcs[i], n_bytes[i]);
            if (rc) return rc;
            off += n_bytes[i];
        }
        return TSTWO_OK;
    }
    size_t off_words = 0;
    for (size_t i0 = 0; i0 < n_pieces; i0 += 16) {
        Pieces p = {};
        u32 k = 0, longest = 0;
        for (size_t i = i0; i < n_pieces && i < i0 + 16; i++) {
            if (!n_bytes[i]) continue;
            p.src[k] = (const u32 *)srcs[i];
            p.off[k] = (u32)off_words;
            p.words[k] = (u32)(n_bytes[i] / 4);
            off_words += n_bytes[i] / 4;
            longest = p.words[k] > longest ? p.words[k] : longest;
            k++;
        }
        if (k) hipLaunchKernelGGL(k_pack_pieces, dim3((unsigned)ceil_div((size_t)longest, (size_t)256), k), dim3(256), 0, c.stream, p, (u32 *)c.result_dev);
    }
    TSTWO_LAUNCH_CHECK();
    if (int rc = wait_stream()) return rc;
    memcpy(host_out, c.result_host, total);
    return TSTWO_OK;
}
int tstwo_copy(void *dev_dst, const void *dev_src, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (bytes == 0) return TSTWO_OK;
    TSTWO_HIP(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, g_ctx.stream));
    return TSTWO_OK;
}
int tstwo_zero(void *dev, size_t bytes) {
    TSTWO_REQUIRE_READY();
    if (bytes == 0) return TSTWO_OK;
    TSTWO_HIP(hipMemsetAsync(dev, 0, bytes, g_ctx.stream));
    return TSTWO_OK;
}

int tstwo_event_create(void **ev) {
    TSTWO_REQUIRE_READY();
    hipEvent_t e;
    TSTWO_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return TSTWO_OK;
}
int tstwo_event_record(void *ev) {
    TSTWO_REQUIRE_READY();
    TSTWO_HIP(hipEventRecord((hipEvent_t)ev, g_ctx.stream));
    return TSTWO_OK;
}
int tstwo_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms) {
    TSTWO_REQUIRE_READY();
    TSTWO_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    TSTWO_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return TSTWO_OK;
}
int tstwo_event_destroy(void *ev) {
    if (!ev) return TSTWO_OK;
    TSTWO_HIP(hipEventDestroy((hipEvent_t)ev));
    return TSTWO_OK;
}

}  // extern "C"
