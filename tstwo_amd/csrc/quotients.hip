// quotients.hip — QuotientOps.accumulateQuotients row loop (backend/cpu/quotients.ts:52-116,160-178).
//
// Per row r (bit-reversed order) with p = domain.at(bitrev(r)):
//   den_b = (Pr_b.x - p.x) * Pi_b.y - (Pr_b.y - p.y) * Pi_b.x                          (CM31)
//   num_b = sum_j ( c_j * f_{col_j}(r) - (a_j * p.y + b_j) )
//         = sum_j c_j * f_{col_j}(r)  -  (A_b * p.y + B_b),   A_b = sum_j a_j, B_b = sum_j b_j   (exact)
//   acc   = acc * coeff_b + num_b * den_b^-1
//
// One lane owns two quads of 4 consecutive rows.  In bit-reversed order a quad is p0, conj p0, -p0, conj -p0 and the
// second quad is the same around p0 + Q (Q a fixed point: the quads lie 2^bsel rows apart, chosen so that a wave's loads
// cover whole cache lines), so one double-and-add per lane (amortised to <= 11 M31
// multiplications per row) replaces the reference's per-row scalar multiplication, and the 8
// denominators of a batch share one Montgomery inversion (the unique inverse, same value as the
// reference's per-row batchInverse).  Constants are tiny and wave-uniform (scalar loads).
// Algorithmic bytes per row: 4 per column entry read + 16 written.
#include <vector>

#include "common.h"
#include "field8.cuh"
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

// One lane = 8 rows = the points p0, conj p0, -p0, conj -p0, p1, conj p1, -p1, conj -p1 with p1 = p0 + Q
// (bit-reversed order; Q the point 2^bsel rows away): row s has x = sx[s] * x_{s>>2}, y = sy[s] * y_{s>>2}, sx = + + - -, sy = + - - +.
// Everything below is the same operation on the 8 rows, issued in priority phases (field8.cuh), and uses the signs:
//   * numerator  sum_j c_j f_j(row): lazily in 64 bits, 4 column entries per reduction (4 units of (P-1)P);
//   * A p.y + B and the denominator  C0 + x (P - Pi.y) + y Pi.x  need the products with x0, x1, y0, y1 only (8 + 8
//     multiplications per batch and lane instead of 32 + 32), rows differ by add / subtract (f8::addsub);
//   * the 8 CM31 denominators are inverted through their M31 norms n = re^2 + im^2: d^-1 = conj(d) / n, one Fermat chain
//     per 8 rows (f8::inverse8) — the unique inverses, the values the reference's per-row batchInverse gives;
//   * term = num * d^-1 (QM31 x CM31): 8 multiply-adds + 4 short reductions per row.
constexpr unsigned kSignX = 0xCC, kSignY = 0x66;       // bit s set: row s takes the negative x / y of its half
template <class T>
__device__ __forceinline__ void bcast(T (&r)[8], T v) {
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = v;
}

// SINGLE: one sample batch (the common shape: BASELINE config 3) — a half's rows go straight to memory; otherwise the
// accumulator of all 8 rows lives in registers across the batches (32 more VGPRs).
// LAZY: some batch has more than 4 column entries — the numerator's 64-bit sums are folded, not reduced, between groups of 4.
template <bool SINGLE, bool LAZY>
__global__ void __launch_bounds__(256) k_quotients8(u32 half_initial, u32 log_size, const u32 *const *__restrict__ cols,
                                                   const BatchConst *__restrict__ batches, u32 n_batches,
                                                   const Entry *__restrict__ entries, Soa4 out,
                                                   const cpoint *__restrict__ gen_pow2, cpoint qb, u32 bsel, u32 *flag) {
    const size_t n_threads = (size_t)1 << (log_size - 3);
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    // The lane's two quads of rows: 4t with a zero bit inserted at position bsel, and that + 2^bsel.  bsel = 8 (domains of at
    // least 512 rows): a wave's 64 quads A are 1 KiB of consecutive rows and so are its quads B — every 16-byte load and store
    // of the wave covers whole cache lines (with bsel = 2, 8 consecutive rows per lane, each access used half of every line
    // and the other half was fetched again later: at 32 columns the lines did not survive in L2 in between).
    // Row r sits at natural index bitrev(r): row bit k >= 1 moves the point by 2^(31-k) generator steps (k = 1: -p, k = bsel: + qb).
    // (rows as 32-bit word offsets from wave-uniform column bases: log_size <= 30, checked by the host)
    const u32 u4 = (u32)t << 2;
    const u32 rowA = ((u4 >> bsel) << (bsel + 1)) | (u4 & ((1u << bsel) - 1u)), rowB = rowA + (1u << bsel);
    u32 idx0 = (half_initial + __brev(rowA)) & 0x7fffffffu;
    const cpoint p0 = cpoint_from_index_win(idx0, gen_pow2);      // gen_pow2 = Context::gen_win for this kernel
    const cpoint p1 = cpoint_add(p0, qb);
    u32 xy[8] = {p0.x, p1.x, p0.y, p1.y, p0.x, p1.x, p0.y, p1.y};        // operands of the denominator products (a | b halves)
    u32 yy[8] = {p0.y, p0.y, p0.y, p0.y, p1.y, p1.y, p1.y, p1.y};        // operands of A * y

    u32 acc[SINGLE ? 1 : 4][8];       // [coordinate][row]
    if (!SINGLE) {                    // no batch at all: the quotient is zero
#pragma unroll
        for (int k = 0; k < 4; k++) bcast(acc[SINGLE ? 0 : k], 0u);
    }
    bool zero = false;

    for (u32 b = 0; b < n_batches; b++) {
        const BatchConst bc = batches[b];
        // ---- denominator, linear in the point: (Pr.x - x) Pi.y - (Pr.y - y) Pi.x = C0 - x Pi.y + y Pi.x per CM31 coordinate
        const cm31 c0 = cm31_sub(cm31_mul(bc.prx, bc.piy), cm31_mul(bc.pry, bc.pix));       // wave-uniform (scalar unit)
        u32 ir[8], ii[8];
        {
            u32 m8[8] = {M31_P - bc.piy.a, M31_P - bc.piy.a, bc.pix.a, bc.pix.a, M31_P - bc.piy.b, M31_P - bc.piy.b, bc.pix.b, bc.pix.b}, pr[8];
            f8::mul(pr, xy, m8);       // {x0 npya, x1 npya, y0 pixa, y1 pixa, x0 npyb, x1 npyb, y0 pixb, y1 pixb}
            u32 da[8], db[8], c8[8], tx[8], ty[8], u[8];
            bcast(c8, c0.a);
#pragma unroll
            for (int s = 0; s < 8; s++) { tx[s] = pr[s >> 2]; ty[s] = pr[2 + (s >> 2)]; }
            f8::addsub<kSignX>(u, c8, tx);
            f8::addsub<kSignY>(da, u, ty);
            bcast(c8, c0.b);
#pragma unroll
            for (int s = 0; s < 8; s++) { tx[s] = pr[4 + (s >> 2)]; ty[s] = pr[6 + (s >> 2)]; }
            f8::addsub<kSignX>(u, c8, tx);
            f8::addsub<kSignY>(db, u, ty);
            f8::done();
#pragma unroll
            for (int s = 0; s < 8; s++)
                if ((da[s] | db[s]) == 0) { zero = true; da[s] = 1u; }
            // d^-1 = conj(d) / (re^2 + im^2)
            u64 nn[8];
            u32 n[8], ninv[8], ndb[8];
            f8::boundary<kPrioHeavy>(da, db);
            f8::mul64(nn, da, da); f8::mad(nn, db, db);
            f8::reduce<false>(n, nn);
            f8::inverse8(ninv, n);
            f8::neg_operand(ndb, db);
            f8::mul(ir, da, ninv);
            f8::mul(ii, ndb, ninv);
        }
        // ---- A y: products A_k y0, A_k y1 (one run of 8); rows differ by the sign of y
        u32 ay[8];
        {
            u32 a8[8] = {bc.A.a, bc.A.b, bc.A.c, bc.A.d, bc.A.a, bc.A.b, bc.A.c, bc.A.d};
            f8::mul(ay, a8, yy);
        }
        // ---- per half (4 rows = one 16-byte load per column entry): numerator, num - (A y + B), term, accumulate.
        // 8-wide arrays hold [coordinate 2h of the 4 rows | coordinate 2h + 1 of the 4 rows].
#pragma unroll
        for (int half = 0; half < 2; half++) {
            u32 num[2][8];
            if constexpr (LAZY) {
                // running 64-bit sums, [coordinate 2h of the 4 rows | coordinate 2h + 1 of the 4 rows].  A group of 4 column entries adds 4
                // products of < 2^62; between groups the sum is FOLDED, not reduced: x = lo + 2^32 hi = lo + 2 hi (mod P, 2^31 = 1) is one
                // multiply-add (hi * 2 + lo < 2^34), after which four more products fit again (4 (P-1)^2 + 2^34 < 2^64).  One full
                // reduction (13 instructions) per coordinate and row at the end instead of one per group: with 32 sampled columns the
                // reductions were three quarters of the kernel's instructions.
                u64 accq[2][8];
#pragma unroll
                for (int s = 0; s < 8; s++) accq[0][s] = accq[1][s] = 0ull;
                u32 two = 2u;                                   // in a VGPR: hi * two + lo stays ONE v_mad_u64_u32 (a literal 2 becomes shift + add-with-carry)
                asm volatile("" : "+v"(two));
                for (u32 j = bc.begin; j < bc.end; j += 4) {
                    const u32 cnt = min(4u, bc.end - j);                    // wave-uniform
                    u32 cw[4][4], f[4][4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const Entry en = entries[j + (e < (int)cnt ? e : 0)];          // loads are never branched around; unused products get c = 0
                        const uint4 fv = gload4(cols[en.col], half ? rowB : rowA);
                        f[e][0] = fv.x; f[e][1] = fv.y; f[e][2] = fv.z; f[e][3] = fv.w;
                        const bool on = e < (int)cnt;
                        cw[e][0] = on ? en.c.a : 0u; cw[e][1] = on ? en.c.b : 0u; cw[e][2] = on ? en.c.c : 0u; cw[e][3] = on ? en.c.d : 0u;
                    }
                    u32 fp[2][8] = {{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]},
                                    {f[2][0], f[2][1], f[2][2], f[2][3], f[3][0], f[3][1], f[3][2], f[3][3]}};
                    f8::boundary<kPrioHeavy>(fp[0], fp[1]);
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int s = 0; s < 8; s++) {
                            const int k = 2 * h + (s >> 2), r = s & 3;
                            u64 a = (u64)(u32)(accq[h][s] >> 32) * (u64)two + (u64)(u32)accq[h][s];          // the fold (0 stays 0)
#pragma unroll
                            for (int e = 0; e < 4; e++) a += (u64)cw[e][k] * (u64)fp[e >> 1][4 * (e & 1) + r];
                            accq[h][s] = a;
                        }
                    f8::pin(accq[0]); f8::pin(accq[1]);
                    f8::done();
                }
                f8::reduce(num[0], accq[0]);
                f8::reduce(num[1], accq[1]);

            } else {
                // at most 4 column entries per batch (BASELINE config 3): one group, reduced directly
                bool first = true;
                for (u32 j = bc.begin; j < bc.end; j += 4) {
                    const u32 cnt = min(4u, bc.end - j);                    // wave-uniform
                    u32 cw[4][4], f[4][4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const Entry en = entries[j + (e < (int)cnt ? e : 0)];          // loads are never branched around; unused products get c = 0
                        const uint4 fv = gload4(cols[en.col], half ? rowB : rowA);
                        f[e][0] = fv.x; f[e][1] = fv.y; f[e][2] = fv.z; f[e][3] = fv.w;
                        const bool on = e < (int)cnt;
                        cw[e][0] = on ? en.c.a : 0u; cw[e][1] = on ? en.c.b : 0u; cw[e][2] = on ? en.c.c : 0u; cw[e][3] = on ? en.c.d : 0u;
                    }
                    u32 fp[2][8] = {{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]},
                                    {f[2][0], f[2][1], f[2][2], f[2][3], f[3][0], f[3][1], f[3][2], f[3][3]}};
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        u64 a64[8];
                        f8::boundary<kPrioHeavy>(fp[0], fp[1]);
#pragma unroll
                        for (int s = 0; s < 8; s++) {           // up to 4 products of < 2^62 plus the 31-bit running value
                            const int k = 2 * h + (s >> 2), r = s & 3;
                            u64 a = first ? 0ull : (u64)num[h][s];
#pragma unroll
                            for (int e = 0; e < 4; e++) a += (u64)cw[e][k] * (u64)fp[e >> 1][4 * (e & 1) + r];
                            a64[s] = a;
                        }
                        f8::reduce(num[h], a64);
                    }
                    first = false;
                }
                if (first) { bcast(num[0], 0u); bcast(num[1], 0u); }
            }
            u32 term[2][8];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // num - (A y + B)
                u32 b8[8], t8[8], nb[8], nq[8];
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int k = 2 * h + (s >> 2);
                    b8[s] = k == 0 ? bc.B.a : k == 1 ? bc.B.b : k == 2 ? bc.B.c : bc.B.d;
                    t8[s] = ay[k + 4 * half];
                }
                f8::sub(nb, num[h], b8);
                f8::addsub<(~kSignY) & 0xFFu>(nq, nb, t8);       // minus (+ A y) on the rows with +y, plus on the rows with -y
                // term = (nA + nB i) (ir + ii i): [re of the 4 rows | im of the 4 rows]
                u32 U[8], V[8], W[8], Z[8];
                const u32 P = vgpr_P();
                f8::done();
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int r = 4 * half + (s & 3);
                    U[s] = nq[s & 3]; W[s] = nq[4 + (s & 3)];
                    V[s] = s < 4 ? ir[r] : ii[r];
                    Z[s] = s < 4 ? P - ii[r] : ir[r];
                }
                u64 a64[8];
                f8::boundary<kPrioHeavy>(U, V, W, Z);
                f8::mul64(a64, U, V); f8::mad(a64, W, Z);
                f8::reduce<false>(term[h], a64);
            }
            f8::done();
            if (SINGLE) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    gstore4(out.p[2 * h], half ? rowB : rowA, make_uint4(term[h][0], term[h][1], term[h][2], term[h][3]));
                    gstore4(out.p[2 * h + 1], half ? rowB : rowA, make_uint4(term[h][4], term[h][5], term[h][6], term[h][7]));
                }
            } else if (b == 0) {              // the accumulator is zero before the first batch
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[SINGLE ? 0 : k][4 * half + r] = term[k >> 1][4 * (k & 1) + r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int s = 4 * half + r;
                    const qm31 m = qm31_mul({acc[0][s], acc[SINGLE ? 0 : 1][s], acc[SINGLE ? 0 : 2][s], acc[SINGLE ? 0 : 3][s]}, bc.coeff);
                    acc[0][s] = m31_add(m.a, term[0][r]); acc[SINGLE ? 0 : 1][s] = m31_add(m.b, term[0][4 + r]);
                    acc[SINGLE ? 0 : 2][s] = m31_add(m.c, term[1][r]); acc[SINGLE ? 0 : 3][s] = m31_add(m.d, term[1][4 + r]);
                }
            }
        }
    }
    if (zero) raise_flag(flag);
    if (!SINGLE) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            gstore4(out.p[k], rowA, make_uint4(acc[SINGLE ? 0 : k][0], acc[SINGLE ? 0 : k][1], acc[SINGLE ? 0 : k][2], acc[SINGLE ? 0 : k][3]));
            gstore4(out.p[k], rowB, make_uint4(acc[SINGLE ? 0 : k][4], acc[SINGLE ? 0 : k][5], acc[SINGLE ? 0 : k][6], acc[SINGLE ? 0 : k][7]));
        }
    }
}

// NB (2 or 3) sample batches over the SAME column list (every column opened at two points, z and z·g: the common AIR shape; three
// for columns that also look one row back).  The batches' numerators are sums over the same column words with different
// coefficients, so the words are loaded ONCE and feed all of them (k_quotients8 reads every column again per batch: 32 columns x
// 2^22 took 0.235 ms for two batches against 0.149 for one, the difference being the second trip of 512 MiB through memory).
// Same arithmetic as k_quotients8<., LAZY> otherwise; the result (..(acc * coeff_0 + term_0) * coeff_1 + term_1 ..) of a half goes
// straight to memory, with acc = 0, or — ACCUM — the rows a previous launch left in `out`: a column list opened at k points is
// ceil(k / 3) sweeps over the columns (k = 4: two sweeps of two) instead of k.  NB = 3 holds 96 VGPRs of 64-bit sums: 3 waves per SIMD.
template <int NB, bool ACCUM>
__global__ void __launch_bounds__(256) k_quotients8_multi(u32 half_initial, u32 log_size, const u32 *const *__restrict__ lp,
                                                         const BatchConst *__restrict__ batches, const qm31 *__restrict__ lc, u32 n_entries, Soa4 out,
                                                         const cpoint *__restrict__ gen_pow2, cpoint qb, u32 bsel, u32 *flag) {
    const size_t n_threads = (size_t)1 << (log_size - 3);
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    const u32 u4 = (u32)t << 2;
    const u32 rowA = ((u4 >> bsel) << (bsel + 1)) | (u4 & ((1u << bsel) - 1u)), rowB = rowA + (1u << bsel);
    u32 idx0 = (half_initial + __brev(rowA)) & 0x7fffffffu;
    const cpoint p0 = cpoint_from_index_win(idx0, gen_pow2);
    const cpoint p1 = cpoint_add(p0, qb);
    u32 xy[8] = {p0.x, p1.x, p0.y, p1.y, p0.x, p1.x, p0.y, p1.y};
    u32 yy[8] = {p0.y, p0.y, p0.y, p0.y, p1.y, p1.y, p1.y, p1.y};
    bool zero = false;
    u32 ir[NB][8], ii[NB][8], ay[NB][8];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const BatchConst bc = batches[b];
        const cm31 c0 = cm31_sub(cm31_mul(bc.prx, bc.piy), cm31_mul(bc.pry, bc.pix));
        u32 m8[8] = {M31_P - bc.piy.a, M31_P - bc.piy.a, bc.pix.a, bc.pix.a, M31_P - bc.piy.b, M31_P - bc.piy.b, bc.pix.b, bc.pix.b}, pr[8];
        f8::mul(pr, xy, m8);
        u32 da[8], db[8], c8[8], tx[8], ty[8], u[8];
        bcast(c8, c0.a);
#pragma unroll
        for (int s = 0; s < 8; s++) { tx[s] = pr[s >> 2]; ty[s] = pr[2 + (s >> 2)]; }
        f8::addsub<kSignX>(u, c8, tx);
        f8::addsub<kSignY>(da, u, ty);
        bcast(c8, c0.b);
#pragma unroll
        for (int s = 0; s < 8; s++) { tx[s] = pr[4 + (s >> 2)]; ty[s] = pr[6 + (s >> 2)]; }
        f8::addsub<kSignX>(u, c8, tx);
        f8::addsub<kSignY>(db, u, ty);
        f8::done();
#pragma unroll
        for (int s = 0; s < 8; s++)
            if ((da[s] | db[s]) == 0) { zero = true; da[s] = 1u; }
        u64 nn[8];
        u32 n[8], ninv[8], ndb[8];
        f8::boundary<kPrioHeavy>(da, db);
        f8::mul64(nn, da, da); f8::mad(nn, db, db);
        f8::reduce<false>(n, nn);
        f8::inverse8(ninv, n);
        f8::neg_operand(ndb, db);
        f8::mul(ir[b], da, ninv);
        f8::mul(ii[b], ndb, ninv);
        u32 a8[8] = {bc.A.a, bc.A.b, bc.A.c, bc.A.d, bc.A.a, bc.A.b, bc.A.c, bc.A.d};
        f8::mul(ay[b], a8, yy);
    }
    // (the shared column list as two compact tables: lp[j] = column pointer of position j, lc[b * n_entries + j] = batch b's
    // coefficient there — 8 + 16 NB bytes per position instead of NB 32-byte Entry records and a dependent pointer load: a
    // 256-column list of two batches is 10 KiB and stays in the 16 KiB scalar cache, where 18 KiB of records did not)
#pragma unroll 1          // (rolled on purpose: one copy of the body; ay / ir / ii are then indexed by `half` at run time — eight LDS accesses per lane)
    for (int half = 0; half < 2; half++) {
        u64 accq[NB][2][8];
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int s = 0; s < 8; s++) accq[b][0][s] = accq[b][1][s] = 0ull;
        u32 two = 2u;
        asm volatile("" : "+v"(two));
        for (u32 j = 0; j < n_entries; j += 4) {
            const u32 cnt = min(4u, n_entries - j);                    // wave-uniform
            u32 cw[NB][4][4], f[4][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const u32 je = j + (e < (int)cnt ? e : 0);
                const bool on = e < (int)cnt;
                {
                    const uint4 fv = gload4(lp[je], half ? rowB : rowA);
                    f[e][0] = fv.x; f[e][1] = fv.y; f[e][2] = fv.z; f[e][3] = fv.w;
                }
#pragma unroll
                for (int b = 0; b < NB; b++) {                               // same column, NB coefficient sets
                    const qm31 cq = lc[(u32)b * n_entries + je];
                    cw[b][e][0] = on ? cq.a : 0u; cw[b][e][1] = on ? cq.b : 0u; cw[b][e][2] = on ? cq.c : 0u; cw[b][e][3] = on ? cq.d : 0u;
                }
            }
            u32 fp[2][8] = {{f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]},
                            {f[2][0], f[2][1], f[2][2], f[2][3], f[3][0], f[3][1], f[3][2], f[3][3]}};
            f8::boundary<kPrioHeavy>(fp[0], fp[1]);
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        const int k = 2 * h + (s >> 2), r = s & 3;
                        u64 a = (u64)(u32)(accq[b][h][s] >> 32) * (u64)two + (u64)(u32)accq[b][h][s];          // the fold (0 stays 0)
#pragma unroll
                        for (int e = 0; e < 4; e++) a += (u64)cw[b][e][k] * (u64)fp[e >> 1][4 * (e & 1) + r];
                        accq[b][h][s] = a;
                    }
#pragma unroll
            for (int b = 0; b < NB; b++) { f8::pin(accq[b][0]); f8::pin(accq[b][1]); }
            f8::done();
        }
        u32 term[NB][2][8];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const BatchConst bc = batches[b];
            u32 num[2][8];
            f8::reduce(num[0], accq[b][0]);
            f8::reduce(num[1], accq[b][1]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u32 b8[8], t8[8], nb[8], nq[8];
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int k = 2 * h + (s >> 2);
                    b8[s] = k == 0 ? bc.B.a : k == 1 ? bc.B.b : k == 2 ? bc.B.c : bc.B.d;
                    t8[s] = ay[b][k + 4 * half];
                }
                f8::sub(nb, num[h], b8);
                f8::addsub<(~kSignY) & 0xFFu>(nq, nb, t8);
                u32 U[8], V[8], W[8], Z[8];
                const u32 P = vgpr_P();
                f8::done();
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int r = 4 * half + (s & 3);
                    U[s] = nq[s & 3]; W[s] = nq[4 + (s & 3)];
                    V[s] = s < 4 ? ir[b][r] : ii[b][r];
                    Z[s] = s < 4 ? P - ii[b][r] : ir[b][r];
                }
                u64 a64[8];
                f8::boundary<kPrioHeavy>(U, V, W, Z);
                f8::mul64(a64, U, V); f8::mad(a64, W, Z);
                f8::reduce<false>(term[b][h], a64);
            }
        }
        f8::done();
        u32 o[4][4];
        if constexpr (ACCUM) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = gload4(out.p[k], half ? rowB : rowA);
                o[k][0] = v.x; o[k][1] = v.y; o[k][2] = v.z; o[k][3] = v.w;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (b == 0 && !ACCUM) {                  // the accumulator is zero before the first batch
#pragma unroll
                for (int r = 0; r < 4; r++) { o[0][r] = term[0][0][r]; o[1][r] = term[0][0][4 + r]; o[2][r] = term[0][1][r]; o[3][r] = term[0][1][4 + r]; }
                continue;
            }
            const qm31 cf = batches[b].coeff;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const qm31 m = qm31_mul({o[0][r], o[1][r], o[2][r], o[3][r]}, cf);
                o[0][r] = m31_add(m.a, term[b][0][r]); o[1][r] = m31_add(m.b, term[b][0][4 + r]);
                o[2][r] = m31_add(m.c, term[b][1][r]); o[3][r] = m31_add(m.d, term[b][1][4 + r]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) gstore4(out.p[k], half ? rowB : rowA, make_uint4(o[k][0], o[k][1], o[k][2], o[k][3]));
    }
    if (zero) raise_flag(flag);
}

// NB = 3 or 4 sample batches over ONE column list in ONE sweep (round 4).  k_quotients8_multi keeps a batch's numerator as 16 64-bit
// sums (4 coordinates x 4 rows): three batches are 96 VGPRs of accumulators, 171 registers and 2 waves per SIMD in all.  Here a
// lane's 8 rows are FOUR PAIRS of rows (r, r + 1) = (p, conj p), 128 rows apart, taken one after the other: a pair's numerators are
// 8 sums per batch (4 coordinates x 2 rows), so four batches fit the registers the pair kernel uses for two.  A wave's 8-byte
// accesses cover 512 consecutive bytes — whole lines — and its four sub-blocks are 512 consecutive rows in all.  What used to be
// one Fermat chain per batch over a lane's 8 rows (inverse8) is one chain per PAIR over the pair's 2 NB denominators; the points of
// the pairs are p0, p0 + Q7, p0 + Q8, p0 + Q7 + Q8 (row bits 7 and 8 <-> 2^24 and 2^23 generator steps).  Same arithmetic per row
// as k_quotients8<., LAZY>; ACCUM continues from the rows already in `out`.  Needs log_size >= 9.
template <int NB, bool ACCUM>
__global__ void __launch_bounds__(256) k_quotients_rp(u32 half_initial, u32 log_size, const u32 *const *__restrict__ lp,
                                                     const BatchConst *__restrict__ batches, const qm31 *__restrict__ lc, u32 n_entries, Soa4 out,
                                                     const cpoint *__restrict__ gen_win, cpoint q7, cpoint q8, u32 *flag) {
    static_assert(NB == 3 || NB == 4, "2 NB <= 8 denominators per pair share one inversion");
    const size_t n_threads = (size_t)1 << (log_size - 3);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    const u32 row0 = (((u32)t >> 6) << 9) + 2u * ((u32)t & 63u);          // first row of the lane's pair 0; pair j: + 128 j
    const cpoint p0 = cpoint_from_index_win((half_initial + __brev(row0)) & 0x7fffffffu, gen_win);
    const cpoint p1 = cpoint_add(p0, q7), p2 = cpoint_add(p0, q8);
    const cpoint p3 = cpoint_add(p2, q7);
    bool zero = false;
#pragma unroll 1          // (rolled: one copy of the body)
    for (int j = 0; j < 4; j++) {
        const u32 row = row0 + 128u * (u32)j;
        const cpoint pj = j == 0 ? p0 : j == 1 ? p1 : j == 2 ? p2 : p3;
        // ---- numerators: 8 running 64-bit sums per batch, index 2 k + r (coordinate k, row r of the pair), folded between groups of
        //      4 column entries (x = lo + 2^32 hi = lo + 2 hi mod P: one multiply-add, see k_quotients8<., LAZY>)
        u64 accq[NB][8];
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int s = 0; s < 8; s++) accq[b][s] = 0ull;
        u32 two = 2u;
        asm volatile("" : "+v"(two));
        // The column words of a group of 4 entries are 4 loads of 8 bytes per lane — 2 KiB per wave — so the NEXT group's loads are
        // issued before this group's multiply-adds (two register sets, the loop unrolled by two): 4 KiB per wave in flight, what the
        // 16-byte kernels have; with one group in flight the kernel ran at half their memory rate on wide column lists.
        auto load_group = [&](u32 (&f)[8], u32 jn) {
            const u32 cnt = jn < n_entries ? min(4u, n_entries - jn) : 0u;          // wave-uniform; a group past the end re-reads entry 0
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const u32 je = e < (int)cnt ? jn + e : 0u;
                const uint2 fv = gload2(lp[je], row);      // loads are never branched around
                f[2 * e] = fv.x; f[2 * e + 1] = fv.y;
            }
        };
        auto use_group = [&](u32 (&f)[8], u32 jn) {
            const u32 cnt = min(4u, n_entries - jn);                    // wave-uniform
            u32 cw[NB][4][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const u32 je = jn + (e < (int)cnt ? e : 0);
                const bool on = e < (int)cnt;
#pragma unroll
                for (int b = 0; b < NB; b++) {                               // same column, NB coefficient sets; unused products get c = 0
                    const qm31 cq = lc[(u32)b * n_entries + je];
                    cw[b][e][0] = on ? cq.a : 0u; cw[b][e][1] = on ? cq.b : 0u; cw[b][e][2] = on ? cq.c : 0u; cw[b][e][3] = on ? cq.d : 0u;
                }
            }
            f8::boundary<kPrioHeavy>(f);
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int k = s >> 1, r = s & 1;
                    u64 a = (u64)(u32)(accq[b][s] >> 32) * (u64)two + (u64)(u32)accq[b][s];          // the fold (0 stays 0)
#pragma unroll
                    for (int e = 0; e < 4; e++) a += (u64)cw[b][e][k] * (u64)f[2 * e + r];
                    accq[b][s] = a;
                }
#pragma unroll
            for (int b = 0; b < NB; b++) f8::pin(accq[b]);
            f8::done();
        };
        u32 fa[8], fb[8];
        load_group(fa, 0);
        for (u32 jn = 0; jn < n_entries; jn += 8) {
            load_group(fb, jn + 4);
            use_group(fa, jn);
            load_group(fa, jn + 8);
            if (jn + 4 < n_entries) use_group(fb, jn + 4);
        }
        // ---- the pair's 2 NB denominators (element 2 b + r: batch b, row r; r = 1 is the conjugate point: y -> -y), one inversion
        u32 ir[8], ii[8];
        {
            u32 xs[8], ys[8], mx[8], my[8], px[8], py[8];
            f8::pin(accq[0]);
#pragma unroll
            for (int e = 0; e < 8; e++) { xs[e] = pj.x; ys[e] = pj.y; }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const BatchConst bc = batches[b < NB ? b : 0];
                mx[b] = M31_P - bc.piy.a; mx[4 + b] = M31_P - bc.piy.b;
                my[b] = bc.pix.a; my[4 + b] = bc.pix.b;
            }
            f8::mul(px, xs, mx);            // x (P - Pi.y): [re part of batch 0..3 | im part of batch 0..3]
            f8::mul(py, ys, my);            // y Pi.x
            u32 da[8], db[8], c8[8], tx[8], ty[8], u[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const BatchConst bc = batches[(e >> 1) < NB ? (e >> 1) : 0];
                const cm31 c0 = cm31_sub(cm31_mul(bc.prx, bc.piy), cm31_mul(bc.pry, bc.pix));       // wave-uniform (scalar unit)
                c8[e] = c0.a; tx[e] = px[e >> 1]; ty[e] = py[e >> 1];
            }
            f8::add(u, c8, tx);
            f8::addsub<0xAAu>(da, u, ty);           // + y Pi.x on the row with +y, - on the conjugate row
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const BatchConst bc = batches[(e >> 1) < NB ? (e >> 1) : 0];
                const cm31 c0 = cm31_sub(cm31_mul(bc.prx, bc.piy), cm31_mul(bc.pry, bc.pix));
                c8[e] = c0.b; tx[e] = px[4 + (e >> 1)]; ty[e] = py[4 + (e >> 1)];
            }
            f8::add(u, c8, tx);
            f8::addsub<0xAAu>(db, u, ty);
            f8::done();
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (e >= 2 * NB) { da[e] = 1u; db[e] = 0u; }                          // unused slots of the inversion
                else if ((da[e] | db[e]) == 0) { zero = true; da[e] = 1u; }
            }
            u64 nn[8];
            u32 n[8], ninv[8], ndb[8];
            f8::boundary<kPrioHeavy>(da, db);
            f8::mul64(nn, da, da); f8::mad(nn, db, db);
            f8::reduce<false>(n, nn);
            f8::inverse8(ninv, n);
            f8::neg_operand(ndb, db);
            f8::mul(ir, da, ninv);
            f8::mul(ii, ndb, ninv);
        }
        // ---- per batch: num - (A y + B), term = that x d^-1, acc = acc * coeff + term
        u32 o[4][2];
        if constexpr (ACCUM) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint2 v = gload2(out.p[k], row);
                o[k][0] = v.x; o[k][1] = v.y;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const BatchConst bc = batches[b];
            u32 num[8], ay[8], a8[8], y8[8], b8[8], nb[8], nq[8];
            f8::reduce(num, accq[b]);
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int k = s >> 1;
                a8[s] = k == 0 ? bc.A.a : k == 1 ? bc.A.b : k == 2 ? bc.A.c : bc.A.d;
                b8[s] = k == 0 ? bc.B.a : k == 1 ? bc.B.b : k == 2 ? bc.B.c : bc.B.d;
                y8[s] = pj.y;
            }
            f8::mul(ay, a8, y8);                                  // A_k y (the same for both rows; the conjugate row takes it with the other sign)
            f8::sub(nb, num, b8);
            f8::addsub<0x55u>(nq, nb, ay);                        // minus (+ A y) on the row with +y (even index), plus on the row with -y
            // term = (n0 + n1 i | n2 + n3 i) (ir + ii i): index 4 h + 2 q + r (CM31 half h, part q: re / im, row r)
            u32 U[8], V[8], W[8], Z[8], term[8];
            const u32 P = vgpr_P();
            f8::done();
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int h = s >> 2, q = (s >> 1) & 1, r = s & 1;
                U[s] = nq[2 * (2 * h) + r]; W[s] = nq[2 * (2 * h + 1) + r];
                V[s] = q == 0 ? ir[2 * b + r] : ii[2 * b + r];
                Z[s] = q == 0 ? P - ii[2 * b + r] : ir[2 * b + r];
            }
            u64 a64[8];
            f8::boundary<kPrioHeavy>(U, V, W, Z);
            f8::mul64(a64, U, V); f8::mad(a64, W, Z);
            f8::reduce<false>(term, a64);
            f8::done();
            // coordinate k of row r: term[4 (k >> 1) + 2 (k & 1) + r]
            if (b == 0 && !ACCUM) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int r = 0; r < 2; r++) o[k][r] = term[4 * (k >> 1) + 2 * (k & 1) + r];
            } else {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const qm31 m = qm31_mul({o[0][r], o[1][r], o[2][r], o[3][r]}, bc.coeff);
                    o[0][r] = m31_add(m.a, term[r]); o[1][r] = m31_add(m.b, term[2 + r]);
                    o[2][r] = m31_add(m.c, term[4 + r]); o[3][r] = m31_add(m.d, term[6 + r]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) gstore2(out.p[k], row, make_uint2(o[k][0], o[k][1]));
    }
    if (zero) raise_flag(flag);
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
    if (zero) raise_flag(flag);
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
    // k >= 2 batches whose column lists overlap take the shared-load kernels, which read ONE list — the union of the batches'
    // columns, in first-seen order — as two compact tables: per position the column pointer, per batch and position the
    // coefficient (zero where a batch does not sample that column: it then adds nothing, and its a_j, b_j are not in the batch's
    // A, B either; a column listed twice in a batch gets the sum of its coefficients).  Every column opened at the same k points is
    // the case with no zeros; "every column at z, half of them also at z / g" is the common AIR shape with some.  Taken when the
    // batches hold at least 1.4 entries per union column on average — below that the zero products cost more than the shared loads save.
    std::vector<u32> ulist;
    std::vector<int> upos(n_cols, -1);
    for (size_t j = 0; j < n_entries; j++)
        if (upos[col_idx[j]] < 0) { upos[col_idx[j]] = (int)ulist.size(); ulist.push_back(col_idx[j]); }
    const u32 per = (u32)ulist.size();
    const bool same_list = n_batches >= 2 && per > 0 && !knobs().quot_no_pair && 10 * n_entries >= 14 * (size_t)per;
    const size_t lp_bytes = same_list ? (((size_t)per * sizeof(u32 *) + 63) / 64) * 64 : 0;
    const size_t lc_bytes = same_list ? ((n_batches * (size_t)per * sizeof(qm31) + 63) / 64) * 64 : 0;
    std::vector<unsigned char> blob(ptr_bytes + bc_bytes + en_bytes + lp_bytes + lc_bytes + 64, 0);
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
    if (same_list) {
        const u32 **hlp = (const u32 **)(blob.data() + ptr_bytes + bc_bytes + en_bytes);
        qm31 *hlc = (qm31 *)(blob.data() + ptr_bytes + bc_bytes + en_bytes + lp_bytes);
        for (size_t j = 0; j < per; j++) hlp[j] = cols[ulist[j]];
        for (size_t bb = 0; bb < n_batches; bb++)
            for (size_t j = batch_off[bb]; j < batch_off[bb + 1]; j++) {
                qm31 &dst = hlc[bb * per + (size_t)upos[col_idx[j]]];          // (the blob is zero-initialised)
                const host::Q sum = host::qadd({{dst.a, dst.b, dst.c, dst.d}}, {{he[j].c.a, he[j].c.b, he[j].c.c, he[j].c.d}});
                dst = {sum.v[0], sum.v[1], sum.v[2], sum.v[3]};
            }
    }
    int rc = ensure_scratch(blob.size());
    if (rc) return rc;
    rc = small_h2d(c.scratch, blob.data(), blob.size());   // stream-ordered: nothing in flight still reads the scratch when it lands
    if (rc) return rc;
    const u32 *const *d_cols = (const u32 *const *)c.scratch;
    const BatchConst *d_b = (const BatchConst *)((unsigned char *)c.scratch + ptr_bytes);
    const Entry *d_e = (const Entry *)((unsigned char *)c.scratch + ptr_bytes + bc_bytes);
    const u32 *const *d_lp = (const u32 *const *)((unsigned char *)c.scratch + ptr_bytes + bc_bytes + en_bytes);
    const qm31 *d_lc = (const qm31 *)((unsigned char *)c.scratch + ptr_bytes + bc_bytes + en_bytes + lp_bytes);
    Soa4 o4 = {{out[0], out[1], out[2], out[3]}};
    bool aligned = true;
    for (int k = 0; k < 4; k++) aligned = aligned && ((((uintptr_t)out[k]) & 15) == 0);
    if (log_size >= 3 && log_size <= 30 && aligned) {
        const u32 bsel = log_size >= 9 ? 8u : 2u;          // row bit that separates a lane's two quads (k_quotients8)
        u32 qx, qy;
        host::point(1u << (31 - bsel), &qx, &qy);
        cpoint qb = {qx, qy};
        size_t n_threads = (size_t)1 << (log_size - 3);
        bool lazy = false;               // a batch with more than 4 column entries: fold the numerator sums between groups
        const bool no_lazy = knobs().quot_no_lazy;          // (measurement knob: reduce after every group of 4)
        for (size_t b = 0; b < n_batches; b++) lazy = lazy || (!no_lazy && batch_off[b + 1] - batch_off[b] > 4);
        const dim3 grid(ceil_div(n_threads, 256));
        // two batches over one column list (same columns in the same order): the column words are loaded once for both
        const bool pair = same_list;
        if (pair) {
            size_t done = 0;
            // sweeps: 2 batches -> k_quotients8_multi<2>; 3 or 4 -> the row-pair kernel k_quotients_rp<3 | 4> (log_size >= 9);
            // more -> 4 (or 3) at a time, the later sweeps continuing from the rows the earlier ones wrote (5 = 3 + 2, 6 = 3 + 3, 7 = 4 + 3)
            const bool rp_ok = log_size >= 9 && !knobs().quot_no_rowpair;
            u32 q7x, q7y, q8x, q8y;
            host::point(1u << 24, &q7x, &q7y);
            host::point(1u << 23, &q8x, &q8y);
            const cpoint q7 = {q7x, q7y}, q8 = {q8x, q8y};
            while (done < n_batches) {
                const size_t left = n_batches - done;
                if (rp_ok && left >= 3) {
                    const int nb = (left == 3 || left == 5 || left == 6) ? 3 : 4;
#define TSTWO_QRP(NBV, ACC) hipLaunchKernelGGL((k_quotients_rp<NBV, ACC>), grid, dim3(256), 0, c.stream, half_initial & 0x7fffffffu, log_size, \
                                               d_lp, d_b + done, d_lc + done * per, per, o4, c.gen_win, q7, q8, c.flag)
                    if (nb == 3 && done == 0) TSTWO_QRP(3, false);
                    else if (nb == 3) TSTWO_QRP(3, true);
                    else if (done == 0) TSTWO_QRP(4, false);
                    else TSTWO_QRP(4, true);
#undef TSTWO_QRP
                    TSTWO_LAUNCH_CHECK();
                    done += (size_t)nb;
                    continue;
                }
                const int nb = knobs().quot_no_triple ? (left >= 2 ? 2 : 1) : ((left == 2 || left == 4) ? 2 : 3);
#define TSTWO_QMULTI(NBV, ACC) hipLaunchKernelGGL((k_quotients8_multi<NBV, ACC>), grid, dim3(256), 0, c.stream, half_initial & 0x7fffffffu, log_size, \
                                                  d_lp, d_b + done, d_lc + done * per, per, o4, c.gen_win, qb, bsel, c.flag)
                if (nb == 1) TSTWO_QMULTI(1, true);
                else if (nb == 2 && done == 0) TSTWO_QMULTI(2, false);
                else if (nb == 2) TSTWO_QMULTI(2, true);
                else if (done == 0) TSTWO_QMULTI(3, false);
                else TSTWO_QMULTI(3, true);
#undef TSTWO_QMULTI
                TSTWO_LAUNCH_CHECK();
                done += (size_t)nb;
            }
            return TSTWO_OK;
        }
#define TSTWO_QLAUNCH(S, Z) hipLaunchKernelGGL((k_quotients8<S, Z>), grid, dim3(256), 0, c.stream, half_initial & 0x7fffffffu, log_size, d_cols, d_b, \
                                               (u32)n_batches, d_e, o4, c.gen_win, qb, bsel, c.flag)
        if (n_batches == 1 && lazy) TSTWO_QLAUNCH(true, true);
        else if (n_batches == 1) TSTWO_QLAUNCH(true, false);
        else if (lazy) TSTWO_QLAUNCH(false, true);
        else TSTWO_QLAUNCH(false, false);
#undef TSTWO_QLAUNCH
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
