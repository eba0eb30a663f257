// hhe_modarith.h -- 64-bit modular arithmetic in integer lanes (moduli up to 61 bits).
// Device code uses __umul64hi (v_mul_hi_u32 / v_mad_u64_u32 chains on gfx950); the host
// path (table generation, tests-only emulator) uses unsigned __int128.
#pragma once
#include "hhe_common.h"

HD u64 mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// x*w mod q in [0,2q) for any x < 2^64, with ws = floor(w * 2^64 / q), w < q.
HD u64 shoup_lazy(u64 x, u64 w, u64 ws, u64 q) { return x * w - mulhi64(x, ws) * q; }
// the same value computed as x*w + hi*(2^64 - q): with nq taken from ModDev (opaque to the compiler) this is a v_mad_u64_u32
// chain that can absorb one more 64-bit addend, instead of two products and a v_sub_co / v_subb_co pair
HD u64 shoup_lazy_n(u64 x, u64 w, u64 ws, u64 nq) { return x * w + mulhi64(x, ws) * nq; }
// Truncated high product: x1*s1 + hi32(x0*s1) + hi32(x1*s0).  It drops hi32(x0*s0) and the carries of the two middle terms'
// low halves, so it is at most 2 below mulhi64(x, s) and never above (no overflow: (2^32-1)^2 + 2(2^32-2) < 2^64).  On gfx950
// two v_mul_hi_u32, one v_mad_u64_u32 and one 64-bit add instead of v_mul_hi_u32 + three v_mad_u64_u32 + four register moves
// that re-pair 32-bit halves with zeros for the 64-bit addends.
HD u64 mulhi64_trunc(u64 x, u64 s)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)s, s1 = (u32)(s >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    return (u64)x1 * s1 + __umulhi(x0, s1) + __umulhi(x1, s0);
#else
    return (u64)x1 * s1 + (u32)(((u64)x0 * s1) >> 32) + (u32)(((u64)x1 * s0) >> 32);
#endif
}
// x*w mod q in [0,4q) for any x < 2^64 (the quotient estimate is at most 2 low on top of Shoup's 1)
HD u64 shoup_lazy_t(u64 x, u64 w, u64 ws, u64 nq) { return x * w + mulhi64_trunc(x, ws) * nq; }
// pseudo-Mersenne fold (ModDev::pm_*): congruent to x mod q and below 2^b + 2^(64-b) c <= 2q for ANY x < 2^64
HD u64 pm_fold(u64 x, const ModDev &m)
{
    const u32 hi = (u32)(x >> 32);
    const u64 r = ((u64)(hi & m.pm_mask) << 32) | (u32)x;
    return (u64)(hi >> m.pm_sh) * m.pm_c + r;
}
// Range bookkeeping of the lazy paths, checked where it can be: the tests-only emulator is built with -DHHE_RANGE_CHECK and
// aborts when a lazy sum wraps 64 bits or a lazy difference goes negative (the adversarial-residue tests drive the bounds);
// in the product these are a plain add / subtract.
#if defined(HHE_RANGE_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
void hhe_range_violation(const char *what);
inline u64 add_nw(u64 a, u64 b) { const u64 s = a + b; if (s < a) hhe_range_violation("lazy sum wrapped 2^64"); return s; }
inline u64 sub_nn(u64 a, u64 b) { if (a < b) hhe_range_violation("lazy difference went negative"); return a - b; }
#else
HD u64 add_nw(u64 a, u64 b) { return a + b; }
HD u64 sub_nn(u64 a, u64 b) { return a - b; }
#endif
// x >= c ? x - c : x given nc = 2^64 - c, for c <= 2^63 and x < 2c: one 64-bit add and a select on the sign of the sum
HD u64 csub(u64 x, u64 nc)
{
    const u64 t = x + nc;
    return (long long)t < 0 ? x : t;
}
HD u64 shoup_mul(u64 x, u64 w, u64 ws, u64 q)
{
    u64 r = shoup_lazy(x, w, ws, q);
    return r >= q ? r - q : r;
}

HD u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
HD u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// Quotient estimate floor((x1:x0) * R / 2^128), R = (r_hi:r_lo) = floor(2^128/q).
// Never above the true quotient and at most 3 below it.
HD u64 barrett_quo(u64 x0, u64 x1, u64 r_lo, u64 r_hi)
{
    u64 carry = mulhi64(x0, r_lo);
    u64 t2lo = x0 * r_hi, t2hi = mulhi64(x0, r_hi);
    u64 s1 = t2lo + carry;
    u64 t3 = t2hi + (s1 < t2lo);
    u64 ulo = x1 * r_lo, uhi = mulhi64(x1, r_lo);
    u64 s2 = s1 + ulo;
    u64 c2 = uhi + (s2 < ulo);
    return x1 * r_hi + t3 + c2;
}
// (x1:x0) mod q, q < 2^61
HD u64 barrett128(u64 x0, u64 x1, const ModDev &m)
{
    u64 quo = barrett_quo(x0, x1, m.r_lo, m.r_hi);
    u64 r = x0 - quo * m.q;
    u64 q2 = m.q << 1;
    r -= (r >= q2) ? q2 : 0;
    r -= (r >= m.q) ? m.q : 0;
    return r;
}
HD u64 mulmod(u64 a, u64 b, const ModDev &m) { return barrett128(a * b, mulhi64(a, b), m); }
// x mod q for a single word
HD u64 reduce64(u64 x, const ModDev &m)
{
    u64 r = x - mulhi64(x, m.r_hi) * m.q;
    return r >= m.q ? r - m.q : r;
}
// floor(w * 2^64 / q) for w < q: the Shoup quotient of a fixed multiplicand
HD u64 shoup_quotient(u64 w, const ModDev &m)
{
    u64 quo = barrett_quo(0, w, m.r_lo, m.r_hi);  // at most 3 below the true quotient
    u64 rem = 0 - quo * m.q;                      // w * 2^64 - quo * q < 4q, exact in 64 bits
    while (rem >= m.q) { rem -= m.q; quo++; }
    return quo;
}
// 128-bit lazy accumulator
struct Acc128 {
    u64 lo, hi;
};
HD void acc_mac(Acc128 &a, u64 x, u64 y)
{
    u64 plo = x * y, phi = mulhi64(x, y);
    a.lo += plo;
    a.hi += phi + (a.lo < plo);
}
HD void acc_add(Acc128 &a, u64 x)
{
    a.lo += x;
    a.hi += (a.lo < x);
}
