/*
 * tstwo_hip.h — C ABI of libtstwo_hip.so, the MI355X (gfx950) backend for tstwo's data-parallel
 * hot path.  This is the drop-in boundary: exactly the entry points a `HipBackend` for
 * teddyjfpender/tstwo binds with bun:ffi (INTEGRATION.md shows the TypeScript stub).  The reference
 * has no FFI today; each entry point below cites the TypeScript interface/function it replaces
 * (paths relative to packages/core/src of the reference).
 *
 * Conventions
 *  - Every function returns 0 on success, nonzero on failure; tstwo_last_error() then holds the
 *    reference's own error text where the reference throws (e.g. "0 has no inverse",
 *    "length is not power of two", "Not enough twiddles!") so the wrapper can `throw new Error(msg)`.
 *  - Columns are plain little-endian uint32 device buffers holding canonical M31 values in [0, P),
 *    P = 2^31-1 (M31.intoSlice layout, fields/m31.ts:272-284).  QM31 / SecureColumnByCoords data
 *    is struct-of-arrays: 4 coordinate columns (fields/secure_columns.ts:124).  Hashes are 32-byte
 *    Blake2s digests, layers are arrays of digests (vcs/blake2_hash.ts:5-49).
 *  - "dev" pointers are device addresses (from tstwo_malloc, or any HIP allocation of the same
 *    process, e.g. a torch tensor's data_ptr()).  `const uint32_t *const *cols` style arguments are
 *    HOST arrays of device pointers, borrowed for the duration of the call.
 *  - Work is enqueued on one HIP stream per process (tstwo_set_stream to borrow the caller's).
 *    Calls that hand results to host memory synchronise; the others are asynchronous and ordered
 *    on that stream; tstwo_sync() drains it.
 *  - Threading: the library is thread-compatible — one thread at a time inside it, the caller
 *    serialises calls — with one exception: tstwo_malloc / tstwo_free / tstwo_trim lock the
 *    allocator, so a block may be released from any thread (a finaliser / GC thread under ctypes or
 *    bun:ffi, which drop the interpreter lock during a call) while another thread is in a call.
 *  - No CPU fallback exists: every entry point fails with an error if no GPU is present.
 */
#ifndef TSTWO_HIP_H
#define TSTWO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSTWO_OK 0
#define TSTWO_ERR_HIP 1            /* HIP runtime failure (text has the hipError string) */
#define TSTWO_ERR_ZERO_INVERSE 2   /* "0 has no inverse"                      fields/m31.ts:139 */
#define TSTWO_ERR_NOT_POW2 3       /* "length is not power of two"            backend/cpu/index.ts:65 */
#define TSTWO_ERR_TWIDDLES 4       /* "Not enough twiddles!"                  poly/utils.ts:86 */
#define TSTWO_ERR_TOO_SMALL 5      /* "fold_line: Evaluation too small, ..."  fri.ts:127 */
#define TSTWO_ERR_LEN_MISMATCH 6   /* "fold_circle_into_line: Length mismatch ..." fri.ts:168 */
#define TSTWO_ERR_BAD_ARG 7
#define TSTWO_ERR_LOG_SIZE 8       /* "log size too small"                    backend/cpu/circle.ts:72 */
#define TSTWO_ERR_COMM 9           /* RCCL missing or failing (text has the ncclResult string) */

/* ---------------------------------------------------------------- lifecycle / plumbing */
int tstwo_init(int device);                 /* select GPU `device`, create the stream; idempotent */
int tstwo_shutdown(void);
const char *tstwo_last_error(void);
const char *tstwo_version(void);
int tstwo_device_count(int *out);
int tstwo_device_name(char *buf, size_t buflen);
int tstwo_set_stream(void *hip_stream);     /* borrow a caller stream (NULL = back to the library's) */
int tstwo_sync(void);
int tstwo_malloc(void **dev, size_t bytes);
int tstwo_free(void *dev);                   /* returns the block to the library's caching allocator (no sync) */
int tstwo_trim(void);                        /* synchronises and gives every cached block back to HIP */
/* Where tstwo_malloc gets its blocks.  POOL (default): size-class free lists over hipMalloc, reuse ordered on the
 * library's stream.  DIRECT: hipMalloc / hipFree per call (free synchronises).  ASYNC: HIP's stream-ordered pool
 * (hipMallocAsync / hipFreeAsync on the library's stream) — UNSAFE: on ROCm 7.2 / gfx950 that pool silently returns wrong
 * data from the second or third allocate / compute / free cycle on (reproduced without any code of this library,
 * tools/repro_hipmallocasync.hip), so tstwo_set_alloc_mode(ASYNC) and TSTWO_ALLOC=async FAIL with TSTWO_ERR_BAD_ARG unless
 * the environment holds TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC=1 (for re-checking a newer runtime).  OR in POISON to have every
 * block handed out filled with 0xA5 bytes first (debugging aid: reads of memory the library never wrote stop looking right
 * by accident).  Environment, read at the first tstwo_malloc: TSTWO_ALLOC=pool|direct|async, TSTWO_POISON=1. */
#define TSTWO_ALLOC_POOL 0
#define TSTWO_ALLOC_DIRECT 1
#define TSTWO_ALLOC_ASYNC 2
#define TSTWO_ALLOC_POISON 0x10
int tstwo_set_alloc_mode(int mode);
/* The host buffer may be reused as soon as tstwo_upload returns; the copy itself is ordered on the stream (small
 * uploads travel through a page-locked ring without a host synchronisation, large ones synchronise). */
int tstwo_upload(void *dev_dst, const void *host_src, size_t bytes);
/* Host hand-over beside the kernels — the boundary createBaseFieldColumn(data) crosses (backend/index.ts:20-31,
 * backend/cpu/index.ts:85-90).  tstwo_upload is synchronous and, from pageable memory, staged by the runtime (35-39 GB/s on the
 * MI355X box).  Page-locked memory makes a copy ONE DMA at link rate, and tstwo_upload_async issues it on the library's copy
 * stream, beside the kernels of the main stream (upload of column group k+1 under the transform of group k):
 *   tstwo_host_register / _unregister   page-lock (and unlock) a range of the CALLER's memory (a Uint32Array's backing store);
 *   tstwo_host_alloc / _free            page-locked memory from the library (bun:ffi toArrayBuffer wraps it);
 *   tstwo_upload_async(dst, src, n)     enqueue the copy and return.  It starts after everything enqueued on the main stream
 *                                       BEFORE this call (a block tstwo_malloc has just recycled is no longer in use by then);
 *                                       src must stay valid and unchanged until tstwo_upload_wait / tstwo_sync returns, dst must
 *                                       not be freed before.  Pageable src works, at tstwo_upload's rate.  Refused during
 *                                       graph capture;
 *   tstwo_upload_fence()                everything enqueued on the main stream AFTER this call waits (on the device) for the
 *                                       copies issued so far; the host does not block;
 *   tstwo_upload_wait()                 the host blocks until the copies issued so far have landed. */
int tstwo_host_register(void *host, size_t bytes);
int tstwo_host_unregister(void *host);
int tstwo_host_alloc(void **host, size_t bytes);
int tstwo_host_free(void *host);
int tstwo_upload_async(void *dev_dst, const void *host_src, size_t bytes);
int tstwo_upload_fence(void);
int tstwo_upload_wait(void);
int tstwo_download(void *host_dst, const void *dev_src, size_t bytes);   /* synchronous */
/* n_pieces small device buffers in one round trip: piece i = n_bytes[i] bytes (whole 4-byte aligned words) at srcs[i]; the pieces
 * land back to back in host_out.  Up to 256 KiB in all they are packed on the device into page-locked host memory and cost one
 * stream synchronisation (the read-backs that end a FRI commit — channel state, the last layer's four coordinate columns
 * `LineEvaluation.interpolate` wants on the host, poly/line.ts:312-329 — took six); larger requests are fetched piece by piece.
 * srcs / n_bytes are host arrays.  Synchronous. */
int tstwo_download_many(const void *const *srcs, const size_t *n_bytes, size_t n_pieces, void *host_out);
int tstwo_copy(void *dev_dst, const void *dev_src, size_t bytes);        /* async d2d */
int tstwo_zero(void *dev, size_t bytes);                                  /* async; Column.zeros, backend/index.ts:56 */
/* hipGraph capture of a launch sequence (launch-bound loops: e.g. the 20+ small kernels of a FRI commit with the device
 * channel).  Between begin and end every asynchronous entry point is RECORDED on the library's stream instead of executed;
 * tstwo_graph_launch replays the recorded sequence with one call.  Rules while capturing: no entry point that returns data
 * to the host (they synchronise) and none that uploads a host array (column tables beyond 64 pointers, gather requests,
 * quotient constants: the staging slot they travel through is only valid at capture time — such a call fails with
 * TSTWO_ERR_BAD_ARG "host-array upload during graph capture" and records nothing); allocations must hit the
 * library's caching allocator (run the sequence once eagerly first); all buffers the sequence uses must outlive the
 * graph, which addresses them by value. */
int tstwo_graph_begin_capture(void);
int tstwo_graph_end_capture(void **graph_exec);
int tstwo_graph_launch(void *graph_exec);
int tstwo_graph_destroy(void *graph_exec);
/* HIP events on the library's stream (bench.py times kernels with these) */
int tstwo_event_create(void **ev);
int tstwo_event_record(void *ev);
int tstwo_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);    /* synchronises on ev_stop */
int tstwo_event_destroy(void *ev);

/* ---------------------------------------------------------------- field column ops
 * M31.add/sub/mul/neg applied per element (fields/m31.ts:147-173); bench/m31.bench.ts workload. */
int tstwo_m31_add(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n);
int tstwo_m31_sub(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n);
int tstwo_m31_mul(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n);
int tstwo_m31_neg(const uint32_t *a, uint32_t *out, size_t n);
/* batchInverse (fields/fields.ts:66-207).  Result = elementwise inverse; any zero input fails with
 * TSTWO_ERR_ZERO_INVERSE like the reference's single final inverse().  Synchronises (error flag). */
int tstwo_m31_batch_inverse(const uint32_t *in, uint32_t *out, size_t n);
int tstwo_cm31_batch_inverse(const uint32_t *const in[2], uint32_t *const out[2], size_t n);
int tstwo_qm31_batch_inverse(const uint32_t *const in[4], uint32_t *const out[4], size_t n);
/* Deferred error reporting for a phase of many small calls: the *_async variants only enqueue the kernel; a zero input
 * sets a sticky flag in device memory instead of failing the call (the affected outputs are unspecified).
 * tstwo_check_zero_flag() synchronises ONCE, clears the flag and fails with TSTWO_ERR_ZERO_INVERSE ("0 has no
 * inverse", fields/m31.ts:139) if any call since the last check met a zero.  The synchronous calls above are
 * unchanged (= async + check). */
int tstwo_m31_batch_inverse_async(const uint32_t *in, uint32_t *out, size_t n);
int tstwo_cm31_batch_inverse_async(const uint32_t *const in[2], uint32_t *const out[2], size_t n);
int tstwo_qm31_batch_inverse_async(const uint32_t *const in[4], uint32_t *const out[4], size_t n);
int tstwo_check_zero_flag(void);
/* QM31.mul / QM31.add per element on SoA columns (fields/qm31.ts:168-233) */
int tstwo_qm31_mul(const uint32_t *const a[4], const uint32_t *const b[4], uint32_t *const out[4], size_t n);
/* AccumulationOps.accumulate: col[i] += other[i] (backend/cpu/accumulation.ts:38-49) */
int tstwo_secure_accumulate(uint32_t *const col[4], const uint32_t *const other[4], size_t n);

/* ---------------------------------------------------------------- ColumnOps.bitReverseColumn
 * In-place bit-reversal permutation of each column (backend/index.ts:20, backend/cpu/index.ts:62-79).
 * n == 0 or not a power of two -> TSTWO_ERR_NOT_POW2. */
int tstwo_bit_reverse(uint32_t *const *cols, size_t n_cols, size_t n);

/* ---------------------------------------------------------------- PolyOps.precomputeTwiddles
 * Twiddle tree of the coset (initial index `coset_initial`, log size `log_size`): 2^log_size words
 * (backend/cpu/circle.ts:210-239, poly/twiddles.ts:10-29).  itw = elementwise inverse (may be NULL).
 * Generated on the device. */
int tstwo_twiddles_build(uint32_t coset_initial, uint32_t log_size, uint32_t *tw, uint32_t *itw);

/* ---------------------------------------------------------------- PolyOps.evaluate / interpolate
 * Circle FFT over n_cols independent columns of 2^log_size words, in place, bit-reversed evaluation
 * order (backend/cpu/circle.ts:84-134 / :136-207).  The domain is CircleDomain(half coset =
 * (half_initial, log_size-1)); tw / itw is the (inverse) twiddle tree of a root coset of log
 * tw_log of which that half coset is a doubling (the wrapper checks is_doubling_of and throws
 * "twiddle tree mismatch").  evaluate expects coefficients already extended to 2^log_size
 * (tstwo_poly_extend).  The true transform (Rust-exact) is computed; the reference's log_size==3
 * output swap (circle.ts:123-131) is offered by the wrapper as a compat option, not here.
 * 1 <= log_size <= 30 = MAX_CIRCLE_DOMAIN_LOG_SIZE (poly/circle/domain.ts:4; a 4 GiB column); larger sizes return
 * TSTWO_ERR_BAD_ARG. */
int tstwo_cfft_evaluate(uint32_t *const *cols, size_t n_cols, uint32_t log_size, uint32_t half_initial,
                        const uint32_t *tw, uint32_t tw_log);
int tstwo_cfft_interpolate(uint32_t *const *cols, size_t n_cols, uint32_t log_size, uint32_t half_initial,
                           const uint32_t *itw, uint32_t tw_log);
/* Out-of-place interpolate: src[i] (evaluations, read only) -> dst[i] (coefficients).  The reference's interpolate has
 * value semantics (the evaluation survives, backend/cpu/circle.ts:136-207 works on a copy); here the copy is folded into
 * the first pass.  Bit-identical to copying and calling tstwo_cfft_interpolate. */
int tstwo_cfft_interpolate_to(const uint32_t *const *src, uint32_t *const *dst, size_t n_cols, uint32_t log_size,
                              uint32_t half_initial, const uint32_t *itw, uint32_t tw_log);
/* CirclePoly.extend + evaluate (backend/cpu/circle.ts:71-134; pcs/prover.ts Rust text "Extension": evaluate_polynomials
 * on the blown-up domain) without materialising the zero padding: polys[i] = 2^log_poly coefficients (read only),
 * out[i] = 2^log_size evaluations.  Bit-identical to tstwo_poly_extend followed by tstwo_cfft_evaluate.
 * log_size < log_poly -> TSTWO_ERR_LOG_SIZE ("log size too small"). */
int tstwo_cfft_evaluate_extended(const uint32_t *const *polys, uint32_t log_poly, uint32_t *const *out, size_t n_cols,
                                 uint32_t log_size, uint32_t half_initial, const uint32_t *tw, uint32_t tw_log);
/* How many passes over HBM (= kernel launches, each reading and writing every column once) tstwo_cfft_evaluate /
 * tstwo_cfft_interpolate take for n_cols columns of 2^log_size words — the planner's own answer, for callers that price a
 * transform against the memory roofline (bench.py: roofline.launches_per_step).  Host-side only, touches no device state. */
int tstwo_cfft_plan_passes(uint32_t log_size, size_t n_cols, uint32_t *n_passes);
/* PolyOps.extend (circle.ts:71-82): dst[0..2^log_dst) = src[0..2^log_src) zero-padded.
 * log_dst < log_src -> TSTWO_ERR_LOG_SIZE ("log size too small"). */
int tstwo_poly_extend(const uint32_t *src, uint32_t log_src, uint32_t *dst, uint32_t log_dst);
/* PolyOps.eval_at_point (circle.ts:52-69): point and result are QM31 as 4 host words. */
int tstwo_eval_at_point(const uint32_t *coeffs, uint32_t log_size, const uint32_t point_x[4],
                        const uint32_t point_y[4], uint32_t out[4]);
/* The same for n_cols polynomials of one size at one point (prove_values samples every column of a tree at the same
 * out-of-domain point): one launch sequence and one read-back.  coeffs: host array of device pointers; out: 4 words per
 * column (host). */
int tstwo_eval_at_point_batch(const uint32_t *const *coeffs, size_t n_cols, uint32_t log_size, const uint32_t px[4],
                              const uint32_t py[4], uint32_t *out);

/* ---------------------------------------------------------------- FriOps (fri.ts:93-110)
 * Twiddles come from the inverse twiddle tree `itw` (root coset log tw_log) — a FRI fold is one
 * inverse-CFFT layer followed by f0 + alpha*f1.
 * fold_line (fri.ts:120-152): in = 2^log_n rows on LineDomain(coset of log log_n, a doubling of the
 * tree's root), out = 2^(log_n-1) rows.  log_n == 0 -> TSTWO_ERR_TOO_SMALL. */
int tstwo_fri_fold_line(const uint32_t *const in[4], uint32_t log_n, const uint32_t *itw, uint32_t tw_log,
                        const uint32_t alpha[4], uint32_t *const out[4]);
/* LineEvaluation.interpolate (poly/line.ts:312-329: bit reversal, lineIfft :354-390, scaling by 1/n) of an evaluation of
 * 2^log_n <= 2^12 rows on LineDomain(coset of log log_n, a doubling of the tree's root): in = the evaluation's four coordinate
 * columns as LineEvaluation stores them (bit-reversed order), out = the LinePoly coefficients in the reference's bit-reversed
 * order, per coordinate.  The last FRI layer (fri.ts:718-754) is the caller.  Asynchronous. */
int tstwo_line_interpolate(const uint32_t *const in[4], uint32_t log_n, const uint32_t *itw, uint32_t tw_log,
                           uint32_t *const out[4]);
/* fold_circle_into_line (fri.ts:162-192): src = 2^log_n rows on a CircleDomain whose half coset is a
 * doubling of the tree's root; dst (dst_len rows) is updated in place: dst*alpha^2 + (alpha*f1 + f0).
 * dst_len != 2^(log_n-1) -> TSTWO_ERR_LEN_MISMATCH. */
int tstwo_fri_fold_circle_into_line(uint32_t *const dst[4], size_t dst_len, const uint32_t *const src[4],
                                    uint32_t log_n, const uint32_t *itw, uint32_t tw_log, const uint32_t alpha[4]);
/* Variants for a commit loop that never leaves the device: alpha_dev points to 4 words (16-byte aligned) in device memory,
 * written earlier on the stream by tstwo_channel_mix_root_draw_felt.  Same results as the by-value calls. */
int tstwo_fri_fold_line_dev(const uint32_t *const in[4], uint32_t log_n, const uint32_t *itw, uint32_t tw_log,
                            const uint32_t *alpha_dev, uint32_t *const out[4]);
int tstwo_fri_fold_circle_into_line_dev(uint32_t *const dst[4], size_t dst_len, const uint32_t *const src[4],
                                        uint32_t log_n, const uint32_t *itw, uint32_t tw_log, const uint32_t *alpha_dev);
/* FriProver.commit's whole layer loop (commitInnerLayers, fri.ts:676-716, with the Merkle / channel wiring of the Rust text) in
 * ONE call, enqueued on the library's stream with nothing read back: a tree over the coordinate columns of every circle
 * evaluation (first layer), then per line layer: mix its root and draw alpha on the device channel
 * (tstwo_channel_mix_root_draw_felt), fold (tstwo_fri_fold_*_dev), commit the folded evaluation — until the evaluation has
 * 2^log_last_layer_size rows.
 * circle_cols: host array of 4 * n_columns device pointers (coordinate columns of the circle evaluations, canonic domains whose
 * half cosets are doublings of the tree's root — the wrapper checks, as for tstwo_fri_fold_circle_into_line); col_logs: their
 * log sizes, strictly decreasing ("column sizes not decreasing" otherwise), each >= 3.  chan: the device channel state
 * (10 words); alphas: device, 4 words per drawn alpha (16-byte aligned, alphas_cap entries >= number of trees).
 * Outputs: *first_tree = the first layer's tree; out[0 .. *n_out - 2] = the inner layers (evaluation + tree, largest first);
 * out[*n_out - 1] = the last layer's evaluation (layers = NULL), which the caller interpolates (fri.ts:718-754).  Every
 * buffer returned is a tstwo_malloc block the CALLER owns (tstwo_free); on failure nothing is returned and nothing leaks. */
typedef struct {
    uint32_t log_size;                     /* the line evaluation has 2^log_size rows */
    uint32_t *cols[4];                     /* device: its coordinate columns */
    uint8_t *layers;                       /* device: the tree committed over them (tstwo_merkle_commit layout), or NULL */
} tstwo_fri_layer_out;
int tstwo_fri_commit_layers(const uint32_t *const *circle_cols, const uint32_t *col_logs, size_t n_columns, const uint32_t *itw,
                            uint32_t tw_log, uint32_t log_last_layer_size, uint32_t *chan, uint32_t *alphas, size_t alphas_cap,
                            uint8_t **first_tree, tstwo_fri_layer_out *out, size_t out_cap, size_t *n_out);
/* Blake2sChannel on the device (channel/blake2.ts:25-224; Rust draw semantics).  chan = 10 words of device memory:
 * digest[8], n_challenges, n_sent (upload the host channel's state, download it back when done).
 * root != NULL: mix_root (vcs/blake2_merkle.ts:28-31) of the 32 bytes at `root` (device memory, e.g. byte 0 of a
 * tstwo_merkle_commit layers buffer).  felt != NULL: draw_felt into the 4 words at `felt` (device memory).
 * Asynchronous: later calls on the stream see the results; nothing reaches the host. */
int tstwo_channel_mix_root_draw_felt(uint32_t *chan, const uint8_t *root, uint32_t *felt);
/* Row shards of a FRI layer for multi-GPU provers (SURVEY.md 8e "contiguous row sharding"; output i of fri.ts:120-192
 * depends on inputs 2i, 2i+1 only).  `in`/`src` hold input rows [2*row_offset, 2*(row_offset+n_rows)) and `out`/`dst`
 * output rows [row_offset, row_offset+n_rows) of a layer of 2^log_n input rows.  row_offset and n_rows are multiples
 * of 4 (TSTWO_ERR_BAD_ARG otherwise).  The concatenation over shards equals tstwo_fri_fold_line /
 * tstwo_fri_fold_circle_into_line on the whole layer. */
int tstwo_fri_fold_line_rows(const uint32_t *const in[4], uint32_t log_n, size_t row_offset, size_t n_rows,
                             const uint32_t *itw, uint32_t tw_log, const uint32_t alpha[4], uint32_t *const out[4]);
int tstwo_fri_fold_circle_into_line_rows(uint32_t *const dst[4], const uint32_t *const src[4], uint32_t log_n,
                                         size_t row_offset, size_t n_rows, const uint32_t *itw, uint32_t tw_log,
                                         const uint32_t alpha[4]);
/* Variants taking the n/2 per-output inverse twiddles explicitly (domains that are not a doubling
 * of a precomputed tree, and log_n < 3 for the circle fold). */
int tstwo_fri_fold_line_tw(const uint32_t *const in[4], uint32_t log_n, const uint32_t *inv_x,
                           const uint32_t alpha[4], uint32_t *const out[4]);
int tstwo_fri_fold_circle_into_line_tw(uint32_t *const dst[4], size_t dst_len, const uint32_t *const src[4],
                                       uint32_t log_n, const uint32_t *inv_y, const uint32_t alpha[4]);
/* decompose (backend/cpu/fri.ts:97-164): lambda (host, 4 words) and g = f -/+ lambda per half. */
int tstwo_fri_decompose(const uint32_t *const in[4], size_t n, uint32_t *const out[4], uint32_t lambda[4]);

/* ---------------------------------------------------------------- MerkleOps (vcs/ops.ts:16-26)
 * commitOnLayer: node i = Blake2s( [prev[2i] || prev[2i+1]]  ||  LE32(cols[0][i]) || ... ), i < 2^log_size
 * (vcs/blake2_merkle.ts:9-24).  prev = NULL for the bottom layer.  out: 2^log_size * 32 bytes (device). */
int tstwo_merkle_commit_layer(uint32_t log_size, const uint8_t *prev, const uint32_t *const *cols,
                              size_t n_cols, uint8_t *out);
/* MerkleProver.commit (vcs/prover.ts:13-30): columns of mixed log sizes join at their layer (input
 * order kept within a size class).  layers (device) receives every layer, root first: layer k (2^k
 * digests) starts at byte offset 32*(2^k - 1); total 32*(2^(max_log+1) - 1) bytes.  root (host, 32 B)
 * may be NULL (then nothing is synchronised).  n_cols == 0 -> one hash of the empty message. */
int tstwo_merkle_commit(const uint32_t *const *cols, const uint32_t *log_sizes, size_t n_cols,
                        uint8_t *layers, uint8_t root[32]);
/* Several trees in one launch sequence (a TreeVec committed together, pcs/prover.ts:62-64: the 8 trees of 32 columns that
 * BASELINE config 5's trace makes on one GPU).  Request r = the arguments of tstwo_merkle_commit for tree r; every tree's layers
 * buffer receives exactly what tstwo_merkle_commit writes.  When the trees share one shape the static leaf kernel serves (16,
 * 32, 48 or 64 columns of one log size >= 17; at most 8 trees, 256 columns in all) each launch covers all of them, so that
 * the latency-bound top of the trees (layers below 2^19 nodes) runs side by side instead of 8 times in a row; any other input
 * is committed tree by tree.  roots (host, n_trees * 32 bytes) may be NULL (then nothing is synchronised). */
typedef struct {
    const uint32_t *const *cols;           /* host array of n_cols device column pointers */
    const uint32_t *log_sizes;
    size_t n_cols;
    uint8_t *layers;                       /* device: tstwo_merkle_layers_bytes(max log) bytes */
} tstwo_commit_request;
int tstwo_merkle_commit_many(const tstwo_commit_request *reqs, size_t n_trees, uint8_t *roots);
size_t tstwo_merkle_layers_bytes(uint32_t max_log);
/* MerkleProver.decommit (vcs/prover.ts:32-109) on a tree built by tstwo_merkle_commit: `layers` is that call's buffer,
 * `cols` / `col_log_sizes` the same columns in the same order.  Query set k = n_queries[k] ascending positions
 * queries[k][] (host) into the layer of log size query_logs[k].  Outputs (host): the queried column values (layer by
 * layer from the largest, node by node, column by column — the order MerkleVerifier.verify consumes), the hash
 * witness (32 bytes each) and the column witness.  The three size_t are in/out: capacity in elements on entry, count on
 * return; if a buffer is too small the counts are returned with TSTWO_ERR_BAD_ARG and nothing is written. */
int tstwo_merkle_decommit(const uint8_t *layers, uint32_t max_log, const uint32_t *const *cols,
                          const uint32_t *col_log_sizes, size_t n_cols, const uint32_t *query_logs,
                          const uint64_t *const *queries, const size_t *n_queries, size_t n_query_sets,
                          uint32_t *queried_values, size_t *n_queried, uint8_t *hash_witness, size_t *n_hashes,
                          uint32_t *column_witness, size_t *n_column_witness);
/* The same for several trees in ONE round trip (all layers of a FRI proof; all trees of a commitment scheme).  Outputs are the
 * per-request outputs concatenated in request order; counts[3r..3r+2] = (queried values, hashes, column-witness words) of
 * request r; totals[3] is in/out (capacities in elements / required sizes), like the single-tree call. */
typedef struct {
    const uint8_t *layers;                 /* device: the tree's tstwo_merkle_commit buffer */
    uint32_t max_log;
    const uint32_t *const *cols;           /* host array of device column pointers */
    const uint32_t *col_log_sizes;
    size_t n_cols;
    const uint32_t *query_logs;            /* query set k: n_queries[k] ascending positions queries[k][] of layer query_logs[k] */
    const uint64_t *const *queries;
    const size_t *n_queries;
    size_t n_query_sets;
} tstwo_decommit_request;
int tstwo_merkle_decommit_many(const tstwo_decommit_request *reqs, size_t n_reqs, uint32_t *queried_values,
                               uint8_t *hash_witness, uint32_t *column_witness, size_t *counts, size_t totals[3]);
/* FriProver.decommit_on_queries (fri.ts:768-785) for a whole FRI proof in ONE round trip: per layer the position logic of
 * computeDecommitmentPositionsAndWitnessEvals (fri.ts:346-384), the witness evaluations, and the Merkle decommitment of the
 * layer's tree at those positions (the queried values are not part of a FriLayerProof, fri.ts:262-269).
 * layers[0] = the first layer: n_evals circle evaluations (possibly of several sizes) under one tree, each queried at the
 * positions folded to its own size (fri.ts:470-480) with fold step `first_fold_step` (CIRCLE_TO_LINE_FOLD_STEP = 1);
 * layers[1..] = the inner layers: one line evaluation each, queried at the positions folded so far, with `fold_step`
 * (FOLD_STEP = 1).  cols: host array of 4 * n_evals device pointers — the coordinate columns of each evaluation in the order
 * the tree was committed with (tstwo_merkle_commit).  queries: ascending distinct positions in [0, 2^log_domain_size).
 * Outputs (host), concatenated layer by layer: witness_evals (4 words per evaluation, in the order fri.ts:346-384 pushes
 * them; first layer: evaluation by evaluation), hash_witness (32 bytes each), column_witness; counts[3r..3r+2] = (witness
 * evaluations, hashes, column-witness words) of layer r; totals[3] is in/out (capacities / required sizes) like
 * tstwo_merkle_decommit_many.  commitments (host, n_layers * 32 bytes, may be NULL): every tree's root
 * (FriLayerProof.commitment) in the same round trip. */
typedef struct {
    const uint8_t *layers;                 /* device: the layer's tstwo_merkle_commit buffer */
    uint32_t max_log;                      /* log size of the tree */
    const uint32_t *const *cols;           /* host array of 4 * n_evals device column pointers */
    const uint32_t *eval_logs;             /* log size of each evaluation */
    size_t n_evals;
} tstwo_fri_layer;
int tstwo_fri_decommit(const tstwo_fri_layer *layers, size_t n_layers, const uint64_t *queries, size_t n_queries,
                       uint32_t log_domain_size, uint32_t first_fold_step, uint32_t fold_step, uint32_t *witness_evals,
                       uint8_t *hash_witness, uint32_t *column_witness, uint8_t *commitments, size_t *counts, size_t totals[3]);
/* Gather for MerkleProver.decommit (vcs/prover.ts:32-109): item i = `words` consecutive uint32 words starting at
 * word index idx[i]*words of the device buffer srcs[i] (a column: words = 1; a layer of digests: words = 8).
 * Results land contiguously in host_out (n_items * words words).  srcs / idx are host arrays.  Synchronises. */
int tstwo_gather_words(const void *const *srcs, const uint64_t *idx, uint32_t words, size_t n_items, uint32_t *host_out);

/* ---------------------------------------------------------------- GrindOps (backend/cpu/grind.ts:31-42, proof_of_work.ts)
 * Smallest nonce >= start_nonce such that Blake2sChannel.mix_u64(nonce) applied to a channel with digest `digest`
 * yields a digest with at least pow_bits trailing zero bits (channel/blake2.ts:96-111: the first 16 digest bytes read
 * as a little-endian u128).  The reference's sequential loop returns the same (first) nonce.  Synchronises. */
int tstwo_grind_blake2s(const uint8_t digest[32], uint32_t pow_bits, uint64_t start_nonce, uint64_t *nonce_out);

/* ---------------------------------------------------------------- QuotientOps
 * accumulateQuotients row loop (backend/cpu/quotients.ts:52-116,160-178) with the per-batch constants
 * computed by the wrapper (quotientConstants, quotients.ts:124-191; constraints.ts:117-128):
 *   batch b covers entries [batch_off[b], batch_off[b+1]) of col_idx / abc;
 *   abc[3*j..3*j+2] = (alpha^j a, alpha^j b, alpha^j c) as QM31 (4 words each);
 *   den_b(row) = (prx[b] - p.x) * piy[b] - (pry[b] - p.y) * pix[b]  in CM31 (2 words each);
 *   acc = acc * batch_coeff[b] + num_b * den_b^-1.
 * Domain = CircleDomain(half coset (half_initial, log_size-1)), rows in bit-reversed order.
 * All constant arrays are host memory.  A vanishing denominator fails with TSTWO_ERR_ZERO_INVERSE. */
int tstwo_quotients_accumulate(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols,
                               size_t n_cols, size_t n_batches, const uint32_t *batch_off,
                               const uint32_t *col_idx, const uint32_t *abc, const uint32_t *batch_coeff,
                               const uint32_t *prx, const uint32_t *pry, const uint32_t *pix,
                               const uint32_t *piy, uint32_t *const out[4]);
/* The same from the samples themselves: the quotient constants (quotientConstants, backend/cpu/quotients.ts:124-152,183-191;
 * complexConjugateLineCoeffs, constraints.ts:117-128; Rust conjugation semantics) are computed by the library.
 * points: 8 words per batch (QM31 x, QM31 y); values: 4 words per entry; batch b owns entries [batch_off[b], batch_off[b+1]).
 * A sample point equal to its own conjugate -> TSTWO_ERR_BAD_ARG "Cannot evaluate a line with a single point". */
int tstwo_quotients_accumulate_samples(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols, size_t n_cols,
                                       size_t n_batches, const uint32_t *batch_off, const uint32_t *col_idx,
                                       const uint32_t *points, const uint32_t *values, const uint32_t random_coeff[4],
                                       uint32_t *const out[4]);
/* The same two calls without the read-back: a vanishing denominator sets the sticky zero flag (tstwo_check_zero_flag). */
int tstwo_quotients_accumulate_async(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols,
                                     size_t n_cols, size_t n_batches, const uint32_t *batch_off,
                                     const uint32_t *col_idx, const uint32_t *abc, const uint32_t *batch_coeff,
                                     const uint32_t *prx, const uint32_t *pry, const uint32_t *pix,
                                     const uint32_t *piy, uint32_t *const out[4]);
int tstwo_quotients_accumulate_samples_async(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols,
                                             size_t n_cols, size_t n_batches, const uint32_t *batch_off,
                                             const uint32_t *col_idx, const uint32_t *points, const uint32_t *values,
                                             const uint32_t random_coeff[4], uint32_t *const out[4]);

/* ---------------------------------------------------------------- multi-GPU: the one exchange on the path
 * Column sharding: rank g commits its own Merkle tree over its own trace columns and the ranks all-gather the
 * 32-byte roots, which every rank then mixes into its channel in rank order (TreeVec order; pcs/prover.ts:62-64,227-228).
 * One process per GPU; the collective is RCCL's ncclAllGather over xGMI, enqueued on the library's stream behind the
 * kernels that produce the root (asynchronous: no host synchronisation).  RCCL is bound at run time (dlopen), so
 * single-GPU hosts need no librccl.
 *   rank 0: tstwo_comm_unique_id(id); the 128 bytes travel to the other ranks by any host channel (file, socket, env);
 *   every rank: tstwo_comm_init(rank, world, id)        — collective, returns when all ranks have joined;
 *   per commit: tstwo_allgather_roots(layers (byte 0 = root), roots_out (world * 32 bytes, device)).
 * Without a communicator (a world of one) the gathers degenerate to a device copy.  tstwo_allgather moves any small
 * per-rank record the same way (the subtree roots of a row-sharded FRI layer). */
#define TSTWO_COMM_ID_BYTES 128
int tstwo_comm_unique_id(uint8_t id[TSTWO_COMM_ID_BYTES]);
int tstwo_comm_init(int rank, int world, const uint8_t id[TSTWO_COMM_ID_BYTES]);
int tstwo_comm_destroy(void);
int tstwo_comm_info(int *rank, int *world);
int tstwo_allgather_roots(const uint8_t *root_dev, uint8_t *roots_out_dev);
int tstwo_allgather(const void *send_dev, void *recv_dev, size_t bytes_per_rank);
/* Overlapped form: the collective runs on a second stream of the library, behind everything enqueued so far, while later
 * calls (the next commit's CFFT) proceed on the main stream.  Neither buffer may be touched by later work until
 * tstwo_comm_wait() has been called: it makes the main stream wait (on the device, not the host) for the last async
 * collective; tstwo_sync() after it covers both. */
int tstwo_allgather_async(const void *send_dev, void *recv_dev, size_t bytes_per_rank);
int tstwo_comm_wait(void);

#ifdef __cplusplus
}
#endif
#endif
