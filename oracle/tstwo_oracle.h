/*
 * tstwo_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * Scalar C restatement of the algorithms on tstwo's data-parallel hot path
 * (reference = teddyjfpender/tstwo, TypeScript; paths below are relative to
 * /root/reference/packages/core/src unless stated).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (tstwo_amd/, libtstwo_hip.so) never does.
 *
 * Parity status: PINNED for the field arithmetic (test-vectors/{m31,cm31,qm31,securecolumn}-test-vectors.json, 817
 * Rust-generated vectors), Blake2s (absolute KATs in test/vcs/{blake2_hash,blake2s_ref}.test.ts) and
 * the twiddle-slicing rule (test/poly/domainLineTwiddles.test.ts).  CFFT, FRI
 * folds, quotients and Merkle layers have no golden vectors in the reference;
 * they are pinned by (i) the reference's own mathematical property tests,
 * ported in tests/test_oracle_*.py, and (ii) fixtures in tests/golden/ produced
 * by an independent pure-Python big-int model (tests/golden/gen_golden.py).
 * The reference itself cannot run here (no Bun/TS toolchain).
 *
 * All values are canonical M31 residues in [0, P) stored as uint32_t.
 * QM31 / SecureColumnByCoords data is SoA: four separate uint32_t columns.
 */
#ifndef TSTWO_ORACLE_H
#define TSTWO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_P 2147483647u

#define ORC_OK 0
#define ORC_ERR_ZERO_INVERSE 1     /* "0 has no inverse"            fields/m31.ts:139 */
#define ORC_ERR_NOT_POW2 2         /* "length is not power of two"  backend/cpu/index.ts:65 */
#define ORC_ERR_NOT_ENOUGH_TWIDDLES 3 /* "Not enough twiddles!"     poly/utils.ts:86 */
#define ORC_ERR_TOO_SMALL 4        /* fold_line n<2                  fri.ts:127 */
#define ORC_ERR_LEN_MISMATCH 5     /* fold_circle_into_line          fri.ts:168 */
#define ORC_ERR_BAD_ARG 6

typedef struct { uint32_t a, b; } orc_cm31;         /* a + b*i            fields/cm31.ts */
typedef struct { uint32_t a, b, c, d; } orc_qm31;   /* (a+bi) + (c+di)u   fields/qm31.ts */
typedef struct { uint32_t x, y; } orc_point;        /* circle.ts:19 */
typedef struct { orc_qm31 x, y; } orc_spoint;       /* CirclePoint<SecureField> */

/* ---- M31 (fields/m31.ts) ---- */
uint32_t orc_m31_reduce(uint64_t x);          /* m31.ts:89-101, x < P^2 */
uint32_t orc_m31_partial_reduce(uint32_t x);  /* m31.ts:60-64, x < 2P */
uint32_t orc_m31_from_i32(int32_t v);         /* m31.ts:252-267 */
uint32_t orc_m31_from_u32(uint32_t v);
uint32_t orc_m31_add(uint32_t a, uint32_t b);
uint32_t orc_m31_sub(uint32_t a, uint32_t b);
uint32_t orc_m31_neg(uint32_t a);
uint32_t orc_m31_mul(uint32_t a, uint32_t b);
uint32_t orc_m31_pow2147483645(uint32_t v);   /* m31.ts:305-326, 37-mult chain */
int orc_m31_inverse(uint32_t v, uint32_t *out); /* m31.ts:137-142 */

/* ---- CM31 / QM31 ---- */
orc_cm31 orc_cm31_add(orc_cm31 x, orc_cm31 y);
orc_cm31 orc_cm31_sub(orc_cm31 x, orc_cm31 y);
orc_cm31 orc_cm31_neg(orc_cm31 x);
orc_cm31 orc_cm31_mul(orc_cm31 x, orc_cm31 y);
int orc_cm31_inverse(orc_cm31 x, orc_cm31 *out);
orc_qm31 orc_qm31_add(orc_qm31 x, orc_qm31 y);
orc_qm31 orc_qm31_sub(orc_qm31 x, orc_qm31 y);
orc_qm31 orc_qm31_neg(orc_qm31 x);
orc_qm31 orc_qm31_mul(orc_qm31 x, orc_qm31 y);
orc_qm31 orc_qm31_mul_m31(orc_qm31 x, uint32_t m);
orc_qm31 orc_qm31_mul_cm31(orc_qm31 x, orc_cm31 m);
int orc_qm31_inverse(orc_qm31 x, orc_qm31 *out);

/* ---- batch inverse (fields/fields.ts:66-207): classic / WIDTH=4 interleave ---- */
int orc_m31_batch_inverse(const uint32_t *col, uint32_t *dst, size_t n);
int orc_cm31_batch_inverse(const orc_cm31 *col, orc_cm31 *dst, size_t n);
/* SoA QM31 column (SecureColumnByCoords, fields/secure_columns.ts:124) */
int orc_qm31_batch_inverse_soa(const uint32_t *const in[4], uint32_t *const out[4], size_t n);

/* elementwise column ops (semantics of M31.add/sub/mul/neg applied per element) */
void orc_m31_col_add(const uint32_t *a, const uint32_t *b, uint32_t *o, size_t n);
void orc_m31_col_sub(const uint32_t *a, const uint32_t *b, uint32_t *o, size_t n);
void orc_m31_col_mul(const uint32_t *a, const uint32_t *b, uint32_t *o, size_t n);
void orc_m31_col_neg(const uint32_t *a, uint32_t *o, size_t n);
void orc_qm31_col_mul_soa(const uint32_t *const a[4], const uint32_t *const b[4], uint32_t *const o[4], size_t n);

/* ---- bit reverse (utils.ts:15-22, backend/cpu/index.ts:62-79) ---- */
uint32_t orc_bit_reverse_index(uint32_t idx, uint32_t log_size);
int orc_bit_reverse_u32(uint32_t *v, size_t n);

/* ---- circle group / cosets (circle.ts) ---- */
orc_point orc_point_add(orc_point p, orc_point q);        /* circle.ts:101-105 */
orc_point orc_index_to_point(uint32_t idx);               /* circle.ts:172-174 (idx mod 2^31) */
uint32_t orc_subgroup_gen(uint32_t log_size);             /* circle.ts:167-170 */
uint32_t orc_half_odds_initial(uint32_t log_size);        /* circle.ts:232-234 */
uint32_t orc_odds_initial(uint32_t log_size);             /* circle.ts:227-229 */
orc_point orc_coset_at(uint32_t initial, uint32_t log_size, uint32_t i); /* circle.ts:276-282 */
/* CircleDomain.at (poly/circle/domain.ts:64-88) for half coset (initial, half_log) */
orc_point orc_circle_domain_at(uint32_t half_initial, uint32_t half_log, uint32_t i);

/* ---- twiddles (backend/cpu/circle.ts:210-239) ---- */
/* buf / ibuf have 2^log_size entries; ibuf may be NULL */
int orc_precompute_twiddles(uint32_t coset_initial, uint32_t log_size, uint32_t *buf, uint32_t *ibuf);

/* ---- CFFT (backend/cpu/circle.ts:84-207,243-278) ----
 * values: 2^log_size entries, in place.  Domain = CircleDomain(half coset (half_initial, log_size-1)).
 * tw: the (i)twiddle tree buffer of a root coset of log tw_log (2^tw_log entries) of which the
 * domain's half coset is a doubling.  compat_log3_swap reproduces circle.ts:123-131,145-151. */
int orc_cfft_evaluate(uint32_t *values, uint32_t log_size, uint32_t half_initial,
                      const uint32_t *tw, uint32_t tw_log, int compat_log3_swap);
int orc_cfft_interpolate(uint32_t *values, uint32_t log_size, uint32_t half_initial,
                         const uint32_t *itw, uint32_t tw_log, int compat_log3_swap);
/* eval_at_point (circle.ts:52-69, poly/utils.ts:36-59) */
orc_qm31 orc_eval_at_point(const uint32_t *coeffs, uint32_t log_size, orc_spoint p);

/* ---- FRI (fri.ts:120-192, backend/cpu/fri.ts:97-164) ---- SoA QM31 ---- */
/* LineDomain coset = (coset_initial, log_n); in has 2^log_n rows, out 2^(log_n-1) */
int orc_fold_line(const uint32_t *const in[4], uint32_t log_n, uint32_t coset_initial,
                  orc_qm31 alpha, uint32_t *const out[4]);
/* src on CircleDomain(half coset (half_initial, log_n-1)), 2^log_n rows; dst 2^(log_n-1) rows in place */
int orc_fold_circle_into_line(uint32_t *const dst[4], size_t dst_len, const uint32_t *const src[4],
                              uint32_t log_n, uint32_t half_initial, orc_qm31 alpha);
int orc_decompose(const uint32_t *const in[4], size_t n, uint32_t *const out[4], orc_qm31 *lambda);

/* ---- Blake2s (vcs/blake2_hash.ts, vcs/blake2s_ref.ts; RFC 7693 unkeyed, 32-byte digest) ---- */
void orc_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]);
void orc_blake2s_compress(const uint32_t h[8], const uint32_t m[16], uint32_t count_lo,
                          uint32_t count_hi, uint32_t lastblock, uint32_t lastnode, uint32_t out[8]);
/* hashNode (vcs/blake2_merkle.ts:9-24) */
void orc_hash_node(const uint8_t *left32, const uint8_t *right32, const uint32_t *values, size_t n_values,
                   uint8_t out[32]);
/* commitOnLayer (vcs/test_utils.ts:17-43 == Rust MerkleOps): prev may be NULL; out 2^log * 32 bytes */
void orc_commit_on_layer(uint32_t log_size, const uint8_t *prev, const uint32_t *const *cols, size_t n_cols,
                         uint8_t *out);
/* MerkleProver.commit (vcs/prover.ts:13-30). log_sizes[i] = log2(len(cols[i])).  layers_out receives
 * layers root-first: layer k has 2^k hashes, k = 0..max_log; total (2^(max_log+1)-1)*32 bytes. */
int orc_merkle_commit(const uint32_t *const *cols, const uint32_t *log_sizes, size_t n_cols,
                      uint8_t *layers_out, uint8_t root[32]);

/* ---- quotients (backend/cpu/quotients.ts, constraints.ts:117-128) ---- */
typedef struct {
    orc_spoint point;
    size_t n_cols;
    const uint32_t *col_idx;     /* n_cols */
    const orc_qm31 *values;      /* n_cols sampled values */
} orc_sample_batch;
/* Rust semantics: conj(v) = (c0, -c1); Pr = c0 part, Pi = c1 part of the point (pcs/quotients.rs). */
orc_qm31 orc_qm31_complex_conjugate(orc_qm31 v);
void orc_line_coeffs(orc_spoint point, orc_qm31 value, orc_qm31 alpha, orc_qm31 out_abc[3]); /* constraints.ts:117-128 */
int orc_accumulate_quotients(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols, size_t n_cols,
                             orc_qm31 random_coeff, const orc_sample_batch *batches, size_t n_batches,
                             uint32_t *const out[4]);
/* Generic row kernel with host-supplied constants (lets tests exercise the TS-quirk variant too):
 * per batch b: den = (prx[b]-p.x)*piy[b] - (pry[b]-p.y)*pix[b] (CM31), line coeffs abc[off[b]+j][3],
 * batch_coeff[b]. */
int orc_accumulate_quotients_consts(uint32_t half_initial, uint32_t log_size, const uint32_t *const *cols,
                                    size_t n_batches, const size_t *batch_off /* n_batches+1 */,
                                    const uint32_t *col_idx, const orc_qm31 *abc /* 3 per entry */,
                                    const orc_qm31 *batch_coeff, const orc_cm31 *prx, const orc_cm31 *pry,
                                    const orc_cm31 *pix, const orc_cm31 *piy, uint32_t *const out[4]);

/* ---- accumulation (backend/cpu/accumulation.ts:38-63) ---- */
void orc_accumulate(uint32_t *const col[4], const uint32_t *const other[4], size_t n);
void orc_generate_secure_powers(orc_qm31 felt, size_t n, orc_qm31 *out);

/* ---- pthread drivers (tstwo_oracle_mt.c): the same functions above run by several threads, for full-size parity tests and
 * the all-cores CPU baseline.  cols are transformed in place, one column per task; the Merkle root is the single-tree root. */
int orc_mt_cfft_evaluate(uint32_t *const *cols, size_t n_cols, uint32_t log_size, uint32_t half_initial,
                         const uint32_t *tw, uint32_t tw_log, unsigned threads);
int orc_mt_merkle_root(const uint32_t *const *cols, size_t n_cols, uint32_t log_size, unsigned threads, uint8_t root[32]);

#ifdef __cplusplus
}
#endif
#endif
