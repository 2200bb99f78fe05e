// host_field.h — host-side M31/QM31 scalars for the handful of constants the launch code needs
// (2^-n, the n <= 2 CFFT twiddles, alpha^2, eval_at_point's folding factors).  Not a compute path.
#pragma once
#include <stdint.h>

namespace tstwo {
namespace host {

typedef uint32_t u32;
typedef uint64_t u64;
constexpr u32 P = 2147483647u;

inline u32 mul(u32 a, u32 b) {
    u64 p = (u64)a * b;
    u64 s = (p & P) + (p >> 31);
    s = (s & P) + (s >> 31);
    return s >= P ? (u32)(s - P) : (u32)s;
}
inline u32 add(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
inline u32 sub(u32 a, u32 b) { return a >= b ? a - b : a + P - b; }
inline u32 neg(u32 a) { return a ? P - a : 0; }
inline u32 pow(u32 a, u32 e) {
    u32 r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
inline u32 inv(u32 a) { return pow(a, P - 2); }
// idx * GEN (circle.ts:58-70,137)
inline void point(u32 idx, u32 *x, u32 *y) {
    u32 rx = 1, ry = 0, cx = 2, cy = 1268011823u;
    idx &= 0x7fffffffu;
    while (idx) {
        if (idx & 1) {
            u32 nx = sub(mul(rx, cx), mul(ry, cy)), ny = add(mul(rx, cy), mul(ry, cx));
            rx = nx; ry = ny;
        }
        u32 dx = sub(mul(cx, cx), mul(cy, cy)), dy = add(mul(cx, cy), mul(cy, cx));
        cx = dx; cy = dy;
        idx >>= 1;
    }
    *x = rx; *y = ry;
}
struct Q { u32 v[4]; };
inline Q qadd(Q x, Q y) { Q r; for (int i = 0; i < 4; i++) r.v[i] = add(x.v[i], y.v[i]); return r; }
inline Q qsub(Q x, Q y) { Q r; for (int i = 0; i < 4; i++) r.v[i] = sub(x.v[i], y.v[i]); return r; }
inline void cmul(const u32 *x, const u32 *y, u32 *o) {
    u32 re = sub(mul(x[0], y[0]), mul(x[1], y[1])), im = add(mul(x[0], y[1]), mul(x[1], y[0]));
    o[0] = re; o[1] = im;
}
// fields/qm31.ts:223-233
inline Q qmul(Q x, Q y) {
    u32 a0b0[2], a1b1[2], r[2], t1[2], t2[2];
    cmul(x.v, y.v, a0b0);
    cmul(x.v + 2, y.v + 2, a1b1);
    const u32 R[2] = {2, 1};
    cmul(R, a1b1, r);
    cmul(x.v, y.v + 2, t1);
    cmul(x.v + 2, y.v, t2);
    Q o;
    o.v[0] = add(a0b0[0], r[0]); o.v[1] = add(a0b0[1], r[1]);
    o.v[2] = add(t1[0], t2[0]); o.v[3] = add(t1[1], t2[1]);
    return o;
}

}  // namespace host
}  // namespace tstwo
