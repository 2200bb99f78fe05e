// quotients.hip — QuotientOps.accumulateQuotients row loop (backend/cpu/quotients.ts:52-116,160-178).
//
// Per row r (bit-reversed order) with p = domain.at(bitrev(r)):
//   den_b = (Pr_b.x - p.x) * Pi_b.y - (Pr_b.y - p.y) * Pi_b.x                          (CM31)
//   num_b = sum_j ( c_j * f_{col_j}(r) - (a_j * p.y + b_j) )
//         = sum_j c_j * f_{col_j}(r)  -  (A_b * p.y + B_b),   A_b = sum_j a_j, B_b = sum_j b_j   (exact)
//   acc   = acc * coeff_b + num_b * den_b^-1
//
// One lane owns 8 consecutive rows.  In bit-reversed order those are p0, conj p0, -p0, conj -p0,
// p0+Q, ..., with Q the order-4 point, so one double-and-add per lane (amortised to <= 11 M31
// multiplications per row) replaces the reference's per-row scalar multiplication, and the 8
// denominators of a batch share one Montgomery inversion (the unique inverse, same value as the
// reference's per-row batchInverse).  Constants are tiny and wave-uniform (scalar loads).
// Algorithmic bytes per row: 4 per column entry read + 16 written.
#include <vector>

#include "common.h"
#include "host_field.h"

using namespace tstwo;

namespace {

struct BatchConst {          // 24 words
    cm31 prx, pry, pix, piy;
    qm31 coeff, A, B;
    u32 begin, end;          // entry range
    u32 pad[2];
};
struct Entry {               // 8 words
    qm31 c;
    u32 col;
    u32 pad[3];
};

__device__ __forceinline__ cpoint domain_point8(cpoint p0, cpoint q4, int s) {
    // s = row & 7 = (s2 s1 s0): s0 -> conjugate, s1 -> antipode, s2 -> + order-4 point
    cpoint p = p0;
    if (s & 4) p = cpoint_add(p, q4);
    if (s & 2) p = {m31_neg(p.x), m31_neg(p.y)};
    if (s & 1) p.y = m31_neg(p.y);
    return p;
}

__global__ void __launch_bounds__(256) k_quotients8(u32 half_initial, u32 log_size, const u32 *const *__restrict__ cols,
                                                   const BatchConst *__restrict__ batches, u32 n_batches,
                                                   const Entry *__restrict__ entries, Soa4 out,
                                                   const cpoint *__restrict__ gen_pow2, cpoint q4, u32 *flag) {
    const size_t n_threads = (size_t)1 << (log_size - 3);
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    // natural index of row 8t: bitrev(t, log_size-3) inside the half coset (log_size-1), step 2^(31-(log_size-1))
    u32 bt = log_size > 3 ? (__brev((u32)t) >> (32 - (log_size - 3))) : 0u;
    u32 idx0 = (half_initial + (bt << (32 - log_size))) & 0x7fffffffu;
    cpoint p0 = cpoint_from_index_win(idx0, gen_pow2);      // gen_pow2 = Context::gen_win for this kernel
    cpoint pt[8];
#pragma unroll
    for (int s = 0; s < 8; s++) pt[s] = domain_point8(p0, q4, s);

    qm31 acc[8];
#pragma unroll
    for (int s = 0; s < 8; s++) acc[s] = {0u, 0u, 0u, 0u};
    bool zero = false;
    const size_t row0 = t << 3;

    for (u32 b = 0; b < n_batches; b++) {
        const BatchConst bc = batches[b];
        // numerator: sum_j c_j * f_j(row) accumulated lazily — up to 4 products of < 2^62 (plus the 31-bit running value) fit
        // 64 bits, so a group of 4 column entries costs 4 v_mad_u64_u32 and ONE reduction per coordinate and row instead of
        // 4 multiplications with a reduction each and 4 modular additions
        u32 num[4][8];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int s = 0; s < 8; s++) num[k][s] = 0u;
        for (u32 j = bc.begin; j < bc.end; j += 4) {
            const u32 cnt = min(4u, bc.end - j);                    // wave-uniform
            u32 cw[4][4], f[4][8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const Entry en = entries[j + (e < (int)cnt ? e : 0)];          // loads are never branched around; unused products get c = 0
                const u32 *col = cols[en.col];
                const uint4 f0 = gload4(col + row0), f1 = gload4(col + row0 + 4);
                f[e][0] = f0.x; f[e][1] = f0.y; f[e][2] = f0.z; f[e][3] = f0.w; f[e][4] = f1.x; f[e][5] = f1.y; f[e][6] = f1.z; f[e][7] = f1.w;
                const bool on = e < (int)cnt;
                cw[e][0] = on ? en.c.a : 0u; cw[e][1] = on ? en.c.b : 0u; cw[e][2] = on ? en.c.c : 0u; cw[e][3] = on ? en.c.d : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    u64 a = (u64)num[k][s];
#pragma unroll
                    for (int e = 0; e < 4; e++) a += (u64)cw[e][k] * (u64)f[e][s];
                    num[k][s] = m31_reduce_u64(a);
                }
        }
        // denominator, linear in the point: (Pr.x - x) Pi.y - (Pr.y - y) Pi.x = (Pr.x Pi.y - Pr.y Pi.x) - x Pi.y + y Pi.x, i.e. per
        // coordinate C0 + x (P - Pi.y) + y Pi.x: two multiply-adds on top of a per-batch constant and one reduction
        const cm31 c0 = cm31_sub(cm31_mul(bc.prx, bc.piy), cm31_mul(bc.pry, bc.pix));
        const u32 npya = M31_P - bc.piy.a, npyb = M31_P - bc.piy.b;
        cm31 den[8], pre[8];
        qm31 numq[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            numq[s] = qm31_sub({num[0][s], num[1][s], num[2][s], num[3][s]}, qm31_add(qm31_mul_m31(bc.A, pt[s].y), bc.B));
            cm31 d;
            d.a = m31_reduce_u64((u64)c0.a + (u64)pt[s].x * npya + (u64)pt[s].y * bc.pix.a);
            d.b = m31_reduce_u64((u64)c0.b + (u64)pt[s].x * npyb + (u64)pt[s].y * bc.pix.b);
            if (cm31_is_zero(d)) { zero = true; d = {1u, 0u}; }
            den[s] = d;
            pre[s] = s == 0 ? d : cm31_mul(pre[s - 1], d);
        }
        cm31 cur = cm31_inv(pre[7]);
#pragma unroll
        for (int s = 7; s >= 0; s--) {
            cm31 dinv = s == 0 ? cur : cm31_mul(pre[s - 1], cur);
            cur = cm31_mul(cur, den[s]);
            const qm31 term = qm31_mul_cm31(numq[s], dinv);
            acc[s] = b == 0 ? term : qm31_add(qm31_mul(acc[s], bc.coeff), term);     // the accumulator is zero before the first batch
        }
    }
    if (zero) atomicOr(flag, 1u);
    gstore4(out.p[0] + row0, make_uint4(acc[0].a, acc[1].a, acc[2].a, acc[3].a));
    gstore4(out.p[0] + row0 + 4, make_uint4(acc[4].a, acc[5].a, acc[6].a, acc[7].a));
    gstore4(out.p[1] + row0, make_uint4(acc[0].b, acc[1].b, acc[2].b, acc[3].b));
    gstore4(out.p[1] + row0 + 4, make_uint4(acc[4].b, acc[5].b, acc[6].b, acc[7].b));
    gstore4(out.p[2] + row0, make_uint4(acc[0].c, acc[1].c, acc[2].c, acc[3].c));
    gstore4(out.p[2] + row0 + 4, make_uint4(acc[4].c, acc[5].c, acc[6].c, acc[7].c));
    gstore4(out.p[3] + row0, make_uint4(acc[0].d, acc[1].d, acc[2].d, acc[3].d));
    gstore4(out.p[3] + row0 + 4, make_uint4(acc[4].d, acc[5].d, acc[6].d, acc[7].d));
}

// Any log_size (used for log_size < 3): one row per lane, the reference's formulation verbatim.
__global__ void __launch_bounds__(256) k_quotients_row(u32 half_initial, u32 log_size, const u32 *const *__restrict__ cols,
                                                      const BatchConst *__restrict__ batches, u32 n_batches,
                                                      const Entry *__restrict__ entries, Soa4 out,
                                                      const cpoint *__restrict__ gen_pow2, u32 *flag) {
    const size_t N = (size_t)1 << log_size;
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    u32 i = log_size ? (__brev((u32)row) >> (32 - log_size)) : 0u;   // domain.at(bitrev(row)), poly/circle/domain.ts:76-88
    const u32 half = 1u << (log_size - 1), step = log_size > 1 ? (1u << (32 - log_size)) : 0u;
    u32 idx = i < half ? half_initial + i * step : 0x80000000u - ((half_initial + (i - half) * step) & 0x7fffffffu);
    cpoint p = cpoint_from_index(idx, gen_pow2);
    qm31 acc = {0u, 0u, 0u, 0u};
    bool zero = false;
    for (u32 b = 0; b < n_batches; b++) {
        const BatchConst bc = batches[b];
        qm31 num = {0u, 0u, 0u, 0u};
        for (u32 j = bc.begin; j < bc.end; j++) num = qm31_add(num, qm31_mul_m31(entries[j].c, cols[entries[j].col][row]));
        num = qm31_sub(num, qm31_add(qm31_mul_m31(bc.A, p.y), bc.B));
        cm31 dx = cm31_sub(bc.prx, cm31{p.x, 0u}), dy = cm31_sub(bc.pry, cm31{p.y, 0u});
        cm31 d = cm31_sub(cm31_mul(dx, bc.piy), cm31_mul(dy, bc.pix));
        if (cm31_is_zero(d)) { zero = true; d = {1u, 0u}; }
        acc = qm31_add(qm31_mul(acc, bc.coeff), qm31_mul_cm31(num, cm31_inv(d)));
    }
    if (zero) atomicOr(flag, 1u);
    out.p[0][row] = acc.a; out.p[1][row] = acc.b; out.p[2][row] = acc.c; out.p[3][row] = acc.d;
}

qm31 q_from(const u32 *w) { return {w[0], w[1], w[2], w[3]}; }
cm31 c_from(const u32 *w) { return {w[0], w[1]}; }

}  // namespace

extern "C" {

// batch_off is caller data that sizes host buffers: it must start at 0 and never decrease
static int check_batch_off(const u32 *batch_off, size_t n_batches) {
    if (!n_batches) return TSTWO_OK;
    if (batch_off[0] != 0) return set_error(TSTWO_ERR_BAD_ARG, "quotients: batch_off[0] must be 0");
    for (size_t b = 0; b < n_batches; b++)
        if (batch_off[b] > batch_off[b + 1]) return set_error(TSTWO_ERR_BAD_ARG, "quotients: batch_off must be non-decreasing");
    return TSTWO_OK;
}

int tstwo_quotients_accumulate_async(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols, size_t n_batches,
                                     const u32 *batch_off, const u32 *col_idx, const u32 *abc, const u32 *batch_coeff,
                                     const u32 *prx, const u32 *pry, const u32 *pix, const u32 *piy, u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    if (log_size == 0 || log_size > 31) return set_error(TSTWO_ERR_BAD_ARG, "quotients: log size out of range");
    TSTWO_REQUIRE_TABLE(cols, n_cols); TSTWO_REQUIRE_TABLE(out, 4);
    if (n_batches) TSTWO_REQUIRE_PTRS(batch_off, col_idx, abc, batch_coeff, prx, pry, pix, piy);
    { int rc_off = check_batch_off(batch_off, n_batches); if (rc_off) return rc_off; }
    Context &c = ctx();
    const size_t n_entries = n_batches ? batch_off[n_batches] : 0;
    for (size_t j = 0; j < n_entries; j++)
        if (col_idx[j] >= n_cols) return set_error(TSTWO_ERR_BAD_ARG, "quotients: column index out of range");
    // host blob: [column pointers][batch consts][entries]
    const size_t ptr_bytes = ((n_cols * sizeof(u32 *) + 63) / 64) * 64;
    const size_t bc_bytes = ((n_batches * sizeof(BatchConst) + 63) / 64) * 64;
    const size_t en_bytes = ((n_entries * sizeof(Entry) + 63) / 64) * 64;
    std::vector<unsigned char> blob(ptr_bytes + bc_bytes + en_bytes + 64, 0);
    const u32 **hp = (const u32 **)blob.data();
    for (size_t i = 0; i < n_cols; i++) {
        if (((uintptr_t)cols[i]) & 15) return set_error(TSTWO_ERR_BAD_ARG, "quotients: columns must be 16-byte aligned");
        hp[i] = cols[i];
    }
    BatchConst *hb = (BatchConst *)(blob.data() + ptr_bytes);
    Entry *he = (Entry *)(blob.data() + ptr_bytes + bc_bytes);
    for (size_t b = 0; b < n_batches; b++) {
        BatchConst &x = hb[b];
        x.prx = c_from(prx + 2 * b); x.pry = c_from(pry + 2 * b); x.pix = c_from(pix + 2 * b); x.piy = c_from(piy + 2 * b);
        x.coeff = q_from(batch_coeff + 4 * b);
        host::Q A = {{0, 0, 0, 0}}, B = {{0, 0, 0, 0}};
        for (size_t j = batch_off[b]; j < batch_off[b + 1]; j++) {
            host::Q a, bb;
            for (int k = 0; k < 4; k++) { a.v[k] = abc[12 * j + k]; bb.v[k] = abc[12 * j + 4 + k]; }
            A = host::qadd(A, a);
            B = host::qadd(B, bb);
            he[j].c = q_from(abc + 12 * j + 8);
            he[j].col = col_idx[j];
        }
        x.A = {A.v[0], A.v[1], A.v[2], A.v[3]};
        x.B = {B.v[0], B.v[1], B.v[2], B.v[3]};
        x.begin = batch_off[b];
        x.end = batch_off[b + 1];
    }
    int rc = ensure_scratch(blob.size());
    if (rc) return rc;
    rc = small_h2d(c.scratch, blob.data(), blob.size());   // stream-ordered: nothing in flight still reads the scratch when it lands
    if (rc) return rc;
    const u32 *const *d_cols = (const u32 *const *)c.scratch;
    const BatchConst *d_b = (const BatchConst *)((unsigned char *)c.scratch + ptr_bytes);
    const Entry *d_e = (const Entry *)((unsigned char *)c.scratch + ptr_bytes + bc_bytes);
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    bool aligned = true;
    for (int k = 0; k < 4; k++) aligned = aligned && ((((uintptr_t)out[k]) & 15) == 0);
    if (log_size >= 3 && aligned) {
        u32 qx, qy;
        host::point(1u << 29, &qx, &qy);   // order-4 point
        cpoint q4 = {qx, qy};
        size_t n_threads = (size_t)1 << (log_size - 3);
        hipLaunchKernelGGL(k_quotients8, dim3(ceil_div(n_threads, 256)), dim3(256), 0, c.stream, half_initial & 0x7fffffffu,
                           log_size, d_cols, d_b, (u32)n_batches, d_e, o4, c.gen_win, q4, c.flag);
    } else {
        size_t N = (size_t)1 << log_size;
        hipLaunchKernelGGL(k_quotients_row, dim3(ceil_div(N, 256)), dim3(256), 0, c.stream, half_initial & 0x7fffffffu, log_size,
                           d_cols, d_b, (u32)n_batches, d_e, o4, c.gen_pow2, c.flag);
    }
    TSTWO_LAUNCH_CHECK();
    return TSTWO_OK;
}
int tstwo_quotients_accumulate(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols, size_t n_batches,
                               const u32 *batch_off, const u32 *col_idx, const u32 *abc, const u32 *batch_coeff,
                               const u32 *prx, const u32 *pry, const u32 *pix, const u32 *piy, u32 *const out[4]) {
    int rc = tstwo_quotients_accumulate_async(half_initial, log_size, cols, n_cols, n_batches, batch_off, col_idx, abc, batch_coeff,
                                              prx, pry, pix, piy, out);
    return rc ? rc : tstwo_check_zero_flag();
}


// QuotientOps.accumulate_quotients from the SAMPLES (backend/cpu/quotients.ts:52-75 + quotientConstants :124-152,183-191 +
// complexConjugateLineCoeffs, constraints.ts:117-128; Rust semantics: conj(a + bu) = a - bu, Pr = c0, Pi = c1): the per-entry
// line coefficients (alpha^j a, alpha^j b, alpha^j c) and the per-batch alpha^{#cols} are computed here on the host side of
// the library — a few QM31 multiplications per sampled column — and handed to tstwo_quotients_accumulate.
// points: 8 words per batch (x then y, QM31 each); values: 4 words per entry, entries of batch b are
// [batch_off[b], batch_off[b+1]).
static int quotients_from_samples(bool async, u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols, size_t n_batches,
                                  const u32 *batch_off, const u32 *col_idx, const u32 *points, const u32 *values,
                                  const u32 random_coeff[4], u32 *const out[4]) {
    TSTWO_REQUIRE_READY();
    if (n_batches) TSTWO_REQUIRE_PTRS(batch_off, col_idx, points, values, random_coeff);
    { int rc_off = check_batch_off(batch_off, n_batches); if (rc_off) return rc_off; }
    const size_t n_entries = n_batches ? batch_off[n_batches] : 0;
    auto load = [](const u32 *p) { host::Q q; for (int k = 0; k < 4; k++) q.v[k] = p[k]; return q; };
    auto conj = [](host::Q q) { q.v[2] = host::neg(q.v[2]); q.v[3] = host::neg(q.v[3]); return q; };   // (c0, -c1)
    const host::Q rc = load(random_coeff);
    std::vector<u32> abc(12 * n_entries + 4), bco(4 * n_batches + 4), prx(2 * n_batches + 2), pry(2 * n_batches + 2),
        pix(2 * n_batches + 2), piy(2 * n_batches + 2);
    for (size_t b = 0; b < n_batches; b++) {
        const host::Q px = load(points + 8 * b), py = load(points + 8 * b + 4);
        const host::Q cy = conj(py);
        bool same = true;
        for (int k = 0; k < 4; k++) same = same && cy.v[k] == py.v[k];
        if (same) return set_error(TSTWO_ERR_BAD_ARG, "Cannot evaluate a line with a single point");   // constraints.ts:120
        const host::Q c = host::qsub(cy, py);
        host::Q alpha = {{1, 0, 0, 0}}, bc = {{1, 0, 0, 0}};
        for (size_t j = batch_off[b]; j < batch_off[b + 1]; j++) {
            alpha = host::qmul(alpha, rc);
            bc = host::qmul(bc, rc);
            const host::Q v = load(values + 4 * j);
            const host::Q a = host::qsub(conj(v), v);
            const host::Q bb = host::qsub(host::qmul(v, c), host::qmul(a, py));
            const host::Q ea = host::qmul(alpha, a), eb = host::qmul(alpha, bb), ec = host::qmul(alpha, c);
            for (int k = 0; k < 4; k++) { abc[12 * j + k] = ea.v[k]; abc[12 * j + 4 + k] = eb.v[k]; abc[12 * j + 8 + k] = ec.v[k]; }
        }
        for (int k = 0; k < 4; k++) bco[4 * b + k] = bc.v[k];
        prx[2 * b] = px.v[0]; prx[2 * b + 1] = px.v[1]; pix[2 * b] = px.v[2]; pix[2 * b + 1] = px.v[3];
        pry[2 * b] = py.v[0]; pry[2 * b + 1] = py.v[1]; piy[2 * b] = py.v[2]; piy[2 * b + 1] = py.v[3];
    }
    return (async ? tstwo_quotients_accumulate_async : tstwo_quotients_accumulate)(
        half_initial, log_size, cols, n_cols, n_batches, batch_off, col_idx, abc.data(), bco.data(), prx.data(), pry.data(),
        pix.data(), piy.data(), out);
}
int tstwo_quotients_accumulate_samples(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols, size_t n_batches,
                                       const u32 *batch_off, const u32 *col_idx, const u32 *points, const u32 *values,
                                       const u32 random_coeff[4], u32 *const out[4]) {
    return quotients_from_samples(false, half_initial, log_size, cols, n_cols, n_batches, batch_off, col_idx, points, values,
                                  random_coeff, out);
}
int tstwo_quotients_accumulate_samples_async(u32 half_initial, u32 log_size, const u32 *const *cols, size_t n_cols, size_t n_batches,
                                             const u32 *batch_off, const u32 *col_idx, const u32 *points, const u32 *values,
                                             const u32 random_coeff[4], u32 *const out[4]) {
    return quotients_from_samples(true, half_initial, log_size, cols, n_cols, n_batches, batch_off, col_idx, points, values,
                                  random_coeff, out);
}

}  // extern "C"
