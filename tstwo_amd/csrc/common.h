// common.h — shared host-side plumbing of libtstwo_hip.so (context, error text, launch helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <initializer_list>
#include <string>

#include "../../include/tstwo_hip.h"
#include "m31.cuh"

namespace tstwo {

struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;     // stream every launch goes to (own_stream or a borrowed one)
    cpoint *gen_pow2 = nullptr;       // device: GEN * 2^k, k = 0..30 (circle.ts:137)
    cpoint *gen_win = nullptr;        // device: 4 windows x 256 entries, [w][k] = (k * 2^(8w)) * GEN: idx * GEN in 3 point additions
    u32 *flag = nullptr;              // error flag word (zero-inverse detection) as the kernels address it: page-locked HOST memory mapped
                                      // into the device when available (flag_host), else device memory
    u32 *flag_host = nullptr;         // the same word as the host reads it after a stream synchronisation (no copy); nullptr: device flag
    u32 *scratch = nullptr;           // device scratch for reductions (decompose / eval_at_point)
    size_t scratch_bytes = 0;
    int n_cus = 256;
    u32 **coltab = nullptr;           // device: 2 pointer tables of coltab_cap entries each (fill_col_table)
    size_t coltab_cap = 0;
    void *pinned = nullptr;           // page-locked host staging for small read-backs (kPinnedBytes)
    void *result_host = nullptr;      // page-locked host memory mapped into the device (kResultBytes): kernels write small results
    void *result_dev = nullptr;       // straight into it (tstwo_download_many: one synchronisation, no copy); nullptr: not available
    void *up_ring = nullptr;          // page-locked ring of kUpSlots upload slots (small_h2d: asynchronous uploads)
    hipEvent_t up_done[16] = {};      // recorded behind the copy that last used the slot
    bool up_busy[16] = {};
    int up_next = 0;
    // tstwo_upload_async: host -> device copies beside the kernels of `stream`, on kCopyStreams copy streams taken in turn (one
    // stream = one DMA engine at a time; two keep the link busy across the gap between consecutive copies)
    u32 *seq_host = nullptr;             // wait_stream(): a sequence word in page-locked host memory mapped into the device ...
    u32 *seq_dev = nullptr;              // ... as the one-lane signal kernel addresses it
    u32 seq_next = 0;
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    hipEvent_t copy_after = nullptr;     // recorded on `stream`, waited for by the copy stream: a copy never overtakes work enqueued before it
    hipEvent_t copy_done[2] = {nullptr, nullptr};   // recorded behind the last copy of each stream: tstwo_upload_fence makes `stream` wait for them
    bool copy_pending[2] = {false, false};
    int copy_next = 0;
};
constexpr size_t kPinnedBytes = 64 * 1024;
constexpr size_t kResultBytes = 256 * 1024;
constexpr int kUpSlots = 16;
constexpr int kCopyStreams = 2;
constexpr size_t kUpSlotBytes = 16 * 1024;

// Experiment and A/B-timing switches.  The SHIPPED library never takes them from its caller's environment:
// knobs() is a compile-time constant holding the defaults below, so every branch on a non-default value folds away and a
// prover's environment can change neither results nor kernel plans.  Built with -DTSTWO_EXPERIMENTS (python -m tstwo_amd.build
// --experiments -> libtstwo_hip_exp.so; tools/build_variant.sh passes it too) the struct is filled ONCE, at the first use, from
// the TSTWO_* variables named in the comments (context.hip: read_knobs) — never per call.
struct Knobs {
    // cfft.hip
    int cfft_rounds = 1;               // TSTWO_CFFT_ROUNDS: workgroups per launch as a multiple of the resident slots
    int cfft_maxwg = 0;                // TSTWO_CFFT_MAXWG: cap on the workgroups per launch (0 = none)
    int cfft_lds_pad = 0;              // TSTWO_CFFT_LDS_PAD: extra dynamic LDS per workgroup (lowers residency)
    int cfft_kb = 0, cfft_ka = 0;      // TSTWO_CFFT_KB (11-15) / TSTWO_CFFT_KA (1-10): bottom tile / strided layer limit (0 = planner)
    int cfft_logta = 0;                // TSTWO_CFFT_LOGTA (12-15): strided tile (0 = planner)
    bool cfft_b8 = false;              // TSTWO_CFFT_B8: the 8-words-per-lane bottom pass k_cfft_b8 (in-place transforms with a 2^13 bottom tile)
    int cfft_av = 0;                   // TSTWO_CFFT_AV=2: 2^14-word strided tiles on 512 lanes x 32 words (two workgroups per CU)
    int cfft_generic = 0;              // TSTWO_CFFT_GENERIC: bit 0 / 1 generic kernel for bottom / strided passes, bit 2 SKIP the bottom pass
    int cfft_group = 0;                // TSTWO_CFFT_GROUP: Infinity-Cache column grouping
    bool cfft_trace = false, cfft_sync = false;             // TSTWO_CFFT_TRACE / TSTWO_CFFT_SYNC
    bool cfft_no_oop = false, cfft_no_fused_extend = false; // TSTWO_CFFT_NO_OOP / TSTWO_CFFT_NO_FUSED_EXTEND
    // merkle.hip
    int merkle_cap = 32;               // TSTWO_MERKLE_CAP: workgroups per CU before lanes grid-stride
    int merkle_up_log = 0;             // TSTWO_MERKLE_UP_LOG: first quad-lane level (0 = default 16; 15 with UP_ONELANE)
    int merkle_subtree = 2;            // TSTWO_MERKLE_SUBTREE: column-free layers per launch of the in-lane subtree kernel (0, 2-4)
    bool merkle_generic = false, merkle_up_onelane = false, merkle_up_smallwg = false, merkle_up_narrow_first = false;
    bool merkle_no_fused_leaf4 = false, merkle_no_batch = false;
    bool merkle_subtree_lane_stride = false;   // TSTWO_MERKLE_SUBTREE_LANE_STRIDE: round 3's k_merkle_subtree<2> instead of the coalesced k_merkle_subtree2c (A/B)
    // fri.hip
    bool fri_no_tail = false, fri_no_fold_fusion = false;   // TSTWO_FRI_NO_TAIL / TSTWO_FRI_NO_FOLD_FUSION
    int fold_cap = 64;                 // TSTWO_FOLD_CAP: workgroups per CU of the fold kernels
    bool fold1 = false;                // TSTWO_FOLD1: circle fold with one output row per lane
    // field_ops.hip / quotients.hip
    int qinv_k = 0;                    // TSTWO_QINV_K
    bool qinv_montgomery = false, quot_no_lazy = false, quot_no_pair = false;
    bool quot_no_rowpair = false;      // TSTWO_QUOT_NO_ROWPAIR: 3+ batches over one column list through k_quotients8_multi sweeps instead of k_quotients_rp (A/B)
    bool quot_no_triple = false;       // TSTWO_QUOT_NO_TRIPLE: k batches over one column list as sweeps of 2 (+ 1) instead of 3 / 2 (A/B)
    // context.hip
    bool device_flag = false;          // TSTWO_DEVICE_FLAG: zero-inverse flag / result page in device memory
    bool no_fast_wait = false;         // TSTWO_NO_FAST_WAIT: wait_stream() = hipStreamSynchronize (A/B of the polled sequence word)
};
#ifdef TSTWO_EXPERIMENTS
const Knobs &knobs();
#else
inline constexpr Knobs kShippedKnobs{};
inline constexpr const Knobs &knobs() { return kShippedKnobs; }
#endif

Context &ctx();
int set_error(int code, const char *msg);
int set_error(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
int require_ready();
int ensure_scratch(size_t bytes);
// "Everything enqueued on the stream so far has completed" for the SMALL synchronous results of the boundary (a zero flag, a Merkle
// root, eval_at_point's value, tstwo_download_many): hipStreamSynchronize costs 9.5 us on this stack whatever the work was, so a
// one-lane kernel stores a sequence number into page-locked host memory behind the work and the host polls that word (bounded: after
// ~100 us — work that long does not care about 9 us — it falls back to hipStreamSynchronize).  Results written by earlier kernels
// into mapped host memory, or copied into page-locked memory by the stream, are visible before the number is (stream order, then
// ordered bus writes).  Measured (profiles/r04_latency.txt): synchronous m31_batch_inverse of 2^20 12.9 against 16.5 us, a Merkle root 26.2
// against 30.8; polling a HIP event instead (hipEventRecord + hipEventQuery) is slower than hipStreamSynchronize (17.3 us).
int wait_stream();
// reads the device error flag (synchronises the stream) and clears it
int read_and_clear_flag(u32 *value);
// Small device->host / host->device transfers through page-locked staging: a pageable hipMemcpy of a few bytes costs
// ~25 us on this stack, a pinned one ~10 us.  small_d2h synchronises the stream (the data is in host memory on return);
// small_h2d copies the source into a ring slot and returns with the transfer enqueued (stream-ordered, no host sync) for
// sizes up to kUpSlotBytes, and synchronises for larger ones.
int small_d2h(void *host_dst, const void *dev_src, size_t bytes);
// Small results without the copy: a kernel writes them straight into the result page (page-locked host memory mapped into the
// device, kResultBytes), and the read-back is one stream synchronisation.  result_target: the page's device address when `bytes`
// fit (and the page exists), else nullptr — the caller then goes through device scratch + small_d2h.  result_wait: synchronises
// and returns the page as the host sees it.
void *result_target(size_t bytes);
int result_wait(const void **host_view);
int small_h2d(void *dev_dst, const void *host_src, size_t bytes);

#define TSTWO_HIP(call)                                        \
    do {                                                       \
        hipError_t _e = (call);                                \
        if (_e != hipSuccess) return ::tstwo::hip_fail(_e, #call); \
    } while (0)

#define TSTWO_REQUIRE_READY()                      \
    do {                                           \
        int _rc = ::tstwo::require_ready();        \
        if (_rc) return _rc;                       \
    } while (0)

#define TSTWO_LAUNCH_CHECK() TSTWO_HIP(hipGetLastError())

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// by-value pointer tables passed as kernel arguments (no device-side pointer arrays to manage)
constexpr int kMaxColsPerLaunch = 64;    // CFFT / bit-reverse batch chunk
// Up to 64 columns travel by value in the kernel argument; more than that go through a pointer table in device memory
// (`ext`), so that a batch of thousands of small columns is still ONE launch instead of one per 64 columns.
struct ColPtrs { u32 *p[kMaxColsPerLaunch]; u32 *const *ext; };
#ifdef __HIPCC__
// (VALU issue model, phase<PRIO>(values...) and vgpr_P(): phase.cuh, included by m31.cuh)
// Accesses to COLUMN data go through these: a column pointer comes out of a pointer table, so to the compiler it is a generic
// address and a plain dereference is a flat_load / flat_store — vector-memory instructions that ALSO count on lgkmcnt, which
// makes every LDS-only wait (s_waitcnt lgkmcnt(0) before s_barrier) wait for in-flight global traffic as well.  The explicit
// global address space gives global_load / global_store (vmcnt only).
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
#define TSTWO_GLOBAL __attribute__((address_space(1)))
// The zero-inverse flag lives in host-coherent memory: a plain store of 1 (every writer writes the same value, and only when an
// input WAS zero — the rare, failing case) is visible to the host once the kernel has completed.
__device__ __forceinline__ void raise_flag(u32 *flag) { *(volatile TSTWO_GLOBAL u32 *)flag = 1u; }
__device__ __forceinline__ uint4 gload4(const u32 *p) {
    const u32x4_t v = *(const TSTWO_GLOBAL u32x4_t *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gload2(const u32 *p) {
    const u32x2_t v = *(const TSTWO_GLOBAL u32x2_t *)p;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ u32 gload1(const u32 *p) { return *(const TSTWO_GLOBAL u32 *)p; }
__device__ __forceinline__ void gstore4(u32 *p, uint4 x) {
    u32x4_t v;
    v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    *(TSTWO_GLOBAL u32x4_t *)p = v;
}
__device__ __forceinline__ void gstore1(u32 *p, u32 x) { *(TSTWO_GLOBAL u32 *)p = x; }
// The same with the address split into a (wave-uniform) base and a 32-bit WORD offset below 2^30: the byte offset is formed in
// 32 bits, so the access is `global_load/store v, v_off, s[base]` — no 64-bit address pair per access in VGPRs and no
// v_lshl_add_u64 / v_add_co + v_addc (heavy VALU) to build one.
__device__ __forceinline__ uint4 gload4(const u32 *base, u32 word_off) {
    const u32x4_t v = *(const TSTWO_GLOBAL u32x4_t *)((const TSTWO_GLOBAL char *)base + (word_off << 2));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gload2(const u32 *base, u32 word_off) {
    const u32x2_t v = *(const TSTWO_GLOBAL u32x2_t *)((const TSTWO_GLOBAL char *)base + (word_off << 2));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ u32 gload1(const u32 *base, u32 word_off) { return *(const TSTWO_GLOBAL u32 *)((const TSTWO_GLOBAL char *)base + (word_off << 2)); }
__device__ __forceinline__ void gstore4(u32 *base, u32 word_off, uint4 x) {
    u32x4_t v;
    v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    *(TSTWO_GLOBAL u32x4_t *)((TSTWO_GLOBAL char *)base + (word_off << 2)) = v;
}
__device__ __forceinline__ void gstore2(u32 *base, u32 word_off, uint2 x) {
    u32x2_t v;
    v.x = x.x; v.y = x.y;
    *(TSTWO_GLOBAL u32x2_t *)((TSTWO_GLOBAL char *)base + (word_off << 2)) = v;
}
__device__ __forceinline__ void gstore1(u32 *base, u32 word_off, u32 x) { *(TSTWO_GLOBAL u32 *)((TSTWO_GLOBAL char *)base + (word_off << 2)) = x; }
// Column pointer i.  Written as a branch, not as `c.ext ? c.ext[i] : c.p[i]`: the compiler merged that into ONE load through a
// selected generic address — a flat_load for the pointer and flat_loads for every column access derived from it.
__device__ __forceinline__ u32 *colp(const ColPtrs &c, u32 i) {
    u32 *p;
    if (c.ext) {
        asm volatile("");              // keeps the two loads in their own blocks (no if-conversion into a selected address)
        p = c.ext[i];
    } else {
        p = c.p[i];                    // kernel-argument table
    }
    return p;
}
// The same for a wave-uniform i (every tiled kernel), as ONE scalar load: both tables are read through the constant address
// space — the device table (written before the launch, read-only while the kernel runs) or the by-value table inside the
// kernel-argument segment itself, which starts KERNARG_OFF bytes into it (0: ColPtrs is the first kernel argument; the
// second table of the out-of-place / fused-extension passes follows it directly).  The pointer lands in SGPRs, so column
// accesses (gload*/gstore*) are global_load / global_store with a scalar base, and nothing waits on vector memory for a pointer.
template <int KERNARG_OFF = 0>
__device__ __forceinline__ u32 *colp_u(const ColPtrs &c, u32 i) {
    typedef const unsigned long long __attribute__((address_space(4))) *k64;
    typedef const char __attribute__((address_space(4))) *kbytes;
    const k64 tab = c.ext ? (k64)(unsigned long long)c.ext : (k64)((kbytes)__builtin_amdgcn_kernarg_segment_ptr() + KERNARG_OFF);
    u32 *p = (u32 *)tab[(u32)__builtin_amdgcn_readfirstlane((int)i)];
    // (the pointer itself is still "generic" to the compiler: dereference it through gload*/gstore* only)
    return p;
}
constexpr int kSecondTableOff = (int)sizeof(ColPtrs);     // kernel signature (ColPtrs cols, <ColPtrs | NoSrc> src, ...)
#endif
// merkle.hip: tstwo_merkle_commit + the channel's mix_root / draw_felt on its root, as one launch sequence without a separate
// channel launch where the tree's last launch can carry it
int merkle_commit_then_channel(const u32 *const *cols, const u32 *log_sizes, size_t n_cols, uint8_t *layers, u32 *chan, u32 *felt);
// merkle.hip: fold_line fused into the next layer's leaf hashing + tree + channel step; returns -1 when an environment override
// forbids the fusion (the caller then folds and commits separately)
int merkle_commit4_folded(const u32 *const prev[4], u32 log_new, const u32 *inv_x, const u32 *alpha_dev, u32 *const new_cols[4],
                          uint8_t *layers, u32 *chan, u32 *felt);
// merkle.hip: the FRI commit's last layers (2^log0 <= 2^9 rows and below) in one single-workgroup launch
int launch_fri_tail(u32 *const (*eval)[4], uint8_t *const *trees, u32 n_layers, u32 log0, const u32 *itw, u32 tw_log, u32 *chan, u32 *alphas,
                    const u32 *const *pre, const u32 *pre_alpha);
// Host: describe columns [0, n_cols) of `cols` in `out`; slot 0/1 = which of the two device tables to use when a launch needs two.
int fill_col_table(ColPtrs &out, const u32 *const *cols, size_t n_cols, int slot);
constexpr int kMaxHashCols = 256;        // Merkle: columns absorbed per launch (multiple of 16)
struct HashColPtrs { const u32 *p[kMaxHashCols]; };
struct Soa4 { u32 *p[4]; };
struct CSoa4 { const u32 *p[4]; };

// Argument hygiene of the C ABI: a null device pointer must come back as an error, never reach a kernel (a GPU page fault
// can take the whole node down).
inline bool has_null(std::initializer_list<const void *> l) {
    for (const void *p : l) if (!p) return true;
    return false;
}
template <typename T>
inline bool table_has_null(T *const *t, size_t n) {
    if (!t) return n > 0;
    for (size_t i = 0; i < n; i++) if (!t[i]) return true;
    return false;
}
#define TSTWO_REQUIRE_PTRS(...) do { if (::tstwo::has_null({__VA_ARGS__})) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer"); } while (0)
#define TSTWO_REQUIRE_TABLE(t, n) do { if (::tstwo::table_has_null((t), (n))) return set_error(TSTWO_ERR_BAD_ARG, "null device pointer in table"); } while (0)

}  // namespace tstwo
