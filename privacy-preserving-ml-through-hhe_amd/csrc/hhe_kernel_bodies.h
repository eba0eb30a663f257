// hhe_kernel_bodies.h -- the bodies of every gfx950 kernel on the transciphering path,
// written once as host/device inline functions of (block, thread) so that the
// tests-only emulator (tests/emu) can drive exactly the same index arithmetic on the
// CPU.  hhe_kernels.hip wraps each body in a __global__ kernel; nothing here is a
// CPU fallback for the product.
//
// NTT: negacyclic Cooley-Tukey forward / Gentleman-Sande inverse with the transform
// order of SEAL (seal/util/dwthandler.h:94-356; root_powers[bitrev(k)] = psi^k), Harvey
// lazy butterflies in [0,4q) / [0,2q) (seal/util/ntt.h:30-61), split in two passes
// (N = N1 x N2).  One workgroup stages a tile of 2^logm points x 2^logc independent
// lanes in LDS and runs radix-16/8/4 register rounds over it.
#pragma once
#include "hhe_modarith.h"

constexpr int NTT_THREADS = 256;
// 4096-point tiles, radix-16/8 register rounds (16 points per lane).  (A 2048-point / radix-8 geometry at 5-8 waves per SIMD
// was measured in round 1: butterflies-only 0.31 vs 0.28 us per transform, and removed.)
struct NttTile { static constexpr int LOG = 12; };
// register-radix schedule per sub-transform size and workgroup width: T = 256 lanes hold 16 points each (radix-16/8
// rounds), T = 512 lanes hold 8 points each (radix-8/4 rounds: half the registers per lane, one more round)
template <int LOGM, int T = NTT_THREADS> struct NttSched;
template <> struct NttSched<5, 256> { static constexpr int R = 2; static constexpr int rho(int i) { return i == 0 ? 3 : 2; } };
template <> struct NttSched<6, 256> { static constexpr int R = 2; static constexpr int rho(int i) { return 3; } };
template <> struct NttSched<7, 256> { static constexpr int R = 2; static constexpr int rho(int i) { return i == 0 ? 4 : 3; } };
template <> struct NttSched<8, 256> { static constexpr int R = 2; static constexpr int rho(int i) { return 4; } };
template <> struct NttSched<6, 512> { static constexpr int R = 2; static constexpr int rho(int i) { return 3; } };
template <> struct NttSched<7, 512> { static constexpr int R = 3; static constexpr int rho(int i) { return i == 0 ? 3 : 2; } };
template <> struct NttSched<8, 512> { static constexpr int R = 3; static constexpr int rho(int i) { return i == 2 ? 2 : 3; } };
struct NttLds { static constexpr int ELEMS = (1 << NttTile::LOG) + 512; };  // rows of pitch C+1

HD u32 bitrev_n(u32 v, int bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - bits);
#else
    u32 r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
#endif
}
// index map of a Galois automorphism in SEAL's (bit-reversed) NTT domain (seal/util/galois.h:143-153):
// NTT(galois_elt(a))[x] = NTT(a)[ntt_perm_index(x)]
HD u32 ntt_perm_index(u32 x, int logn, u32 elt)
{
    const u32 e = 2 * bitrev_n(x, logn) + 1;
    const u32 e2 = (u32)(((u64)e * elt) & ((2u << logn) - 1));
    return bitrev_n((e2 - 1) >> 1, logn);
}

struct NttGeom {
    int n, M, C, pitch, N_over_M, logm, logc;
    int poly, mod_index, tile;
};
// CM / CC >= 0: pass size and tile columns known at compile time (full tiles: every launch with N >= 4096), so the LDS
// pitch, masks and per-pair strides fold into immediates; -1: taken from the launch arguments (ragged small-N tiles)
template <int CM = -1, int CC = -1> HD NttGeom ntt_geom(const NttArgs &a, int bx, int by)
{
    NttGeom g;
    g.logm = CM >= 0 ? CM : a.logm;
    g.logc = CC >= 0 ? CC : a.logc;
    g.n = 1 << a.logn;
    g.M = 1 << g.logm;
    g.C = 1 << g.logc;
    g.pitch = g.C + 1;
    g.N_over_M = g.n >> g.logm;
    g.poly = by;
    g.mod_index = a.mod_base + by % a.mod_cycle;
    g.tile = bx;
    return g;
}
// global coefficient index of (x, lane)
template <bool STRIDED> HD int ntt_gidx(const NttGeom &g, int x, int lane)
{
    if (STRIDED) return x * g.N_over_M + g.tile * g.C + lane;
    return (g.tile * g.C + lane) * g.M + x;
}

// A pointer that was itself loaded from memory (twiddle tables out of ModDev, per-item operand pointers) has no known address
// space, so its loads compile to flat_load: those count on lgkmcnt as well as vmcnt, every LDS wait behind one becomes a wait
// for the table load too, and the scheduler may not move them across LDS traffic.  Only a pointer TYPE in the global address
// space makes them global_load (an assume on is_shared / is_private, or a cast there and back, is folded away).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(1))) u64 *gptr;
#else
typedef const u64 *gptr;
#endif
HD gptr as_global(const u64 *p) { return (gptr)p; }
struct alignas(16) U2 { u64 a, b; };
HD U2 ld2(const u64 *p) { return *reinterpret_cast<const U2 *>(p); }   // 16 B per lane: the coalescing sweet spot
HD void st2(u64 *p, U2 v) { *reinterpret_cast<U2 *>(p) = v; }
HD U2 ld2g(gptr p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 w = *(const __attribute__((address_space(1))) u64x2 *)p;
    return U2{w.x, w.y};
#else
    return ld2(p);
#endif
}
// Streaming variants for data that is written once and read by a later kernel / read once.  NTT_NT bits, measured in-call on
// MI355X (transcipherings/s, default 1): 1 = non-temporal stores of the transform outputs (intermediate, lazy digits T):
// 229 vs 223 without; 2 = non-temporal transform loads: neutral; 4 = inner-product output S: -1 %; 8 = c0-branch output: neutral;
// 16 / 32 = non-temporal loads of T / of the diagonal operand in the inner product: neutral.
#ifndef NTT_NT
#define NTT_NT 1
#endif
template <int BIT = 1> HD void st2_stream(u64 *p, U2 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NTT_NT & BIT) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        u64x2 w; w.x = v.a; w.y = v.b;
        __builtin_nontemporal_store(w, reinterpret_cast<u64x2 *>(p));
        return;
    }
#endif
    st2(p, v);
}
template <int BIT = 2> HD U2 ld2_stream(const u64 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (NTT_NT & BIT) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 w = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(p));
        return U2{w.x, w.y};
    }
#endif
    return ld2(p);
}
// (in[pi(x)], in[pi(x+1)]) for even x with one 16-byte load: x and x+1 differ in the top bit of their bit reversal, the
// odd exponents 2*bitrev+1 then differ by N, so do their products with the (odd) Galois element, hence pi(x+1) = pi(x) ^ 1
HD U2 ld2_perm(const u64 *base, u32 x_even, int logn, u32 elt)
{
    const u32 p0 = ntt_perm_index(x_even, logn, elt);
    const U2 v = ld2(base + (p0 & ~1u));
    return (p0 & 1) ? U2{v.b, v.a} : v;
}

// (galois(src)[k], galois(src)[k+1]) for even k (GaloisTool::apply_galois as a gather, seal/util/galois.h:32): two 8-byte loads
HD U2 ld2_galois(const u64 *src, u32 k, int logn, u32 einv, u64 q)
{
    const u32 n = 1u << logn;
    const u32 j0 = (u32)(((u64)k * einv) & (2 * n - 1)), j1 = (u32)((j0 + einv) & (2 * n - 1));
    U2 v;
    v.a = (j0 < n) ? src[j0] : negmod(src[j0 - n], q);
    v.b = (j1 < n) ? src[j1] : negmod(src[j1 - n], q);
    return v;
}

// prime of the SOURCE polynomial of poly g.poly in a first pass (digits: limb I of c1, whatever the transform's own modulus)
HD u64 ntt_src_q(const NttArgs &a, const NttGeom &g)
{
    const int ip = a.src_item_polys > 0 ? a.src_item_polys : a.count;
    return mod_at(a.mods, (g.poly % ip) / a.src_div).q;
}

// element pair handled by one lane in the load/store phases: (x, lane) and its neighbour in global memory
template <bool STRIDED> HD void ntt_pair(const NttArgs &a, const NttGeom &g, int e2, int &x, int &lane, int &gi, int &lds0, int &lds1)
{
    if (STRIDED) { lane = (e2 & ((g.C >> 1) - 1)) << 1; x = e2 >> (g.logc - 1); lds0 = x * g.pitch + lane; lds1 = lds0 + 1; }
    else { x = (e2 & ((g.M >> 1) - 1)) << 1; lane = e2 >> (g.logm - 1); lds0 = x * g.pitch + lane; lds1 = lds0 + g.pitch; }
    gi = ntt_gidx<STRIDED>(g, x, lane);
}

// load op of the first pass applied to one 16-byte pair
template <bool FIRST> HD U2 ntt_load_op(const NttArgs &a, const ModDev &m, int poly, U2 v)
{
    if (FIRST) {
        if (a.load_op == LOAD_DIGIT) {
            if (a.zero_flag && (v.a == 0 || v.b == 0)) *a.zero_flag = 1;
            if (a.digit_reduce) { v.a = reduce64(v.a, m); v.b = reduce64(v.b, m); }
        }
        else if (a.load_op == LOAD_LIFT) {
            const u64 thr = (a.t + 1) >> 1, inc = m.q - a.t;
            v.a = (v.a >= thr) ? v.a + inc : v.a;
            v.b = (v.b >= thr) ? v.b + inc : v.b;
        } else if (a.load_op == LOAD_RNEG) {
            const u64 h = a.ks.half_mod[poly % a.L];
            v.a = submod(reduce64(v.a, m), h, m.q);
            v.b = submod(reduce64(v.b, m), h, m.q);
        }
    }
    return v;
}
// full tiles (every launch with N >= 4096): all NP global loads of a lane are issued before the first LDS write, so
// a workgroup's load phase costs one memory round trip instead of NP
template <bool STRIDED, bool INVERSE, int NP, int T = NTT_THREADS>
HD void ntt_load_full(const NttArgs &a, const NttGeom &g, const ModDev &m, const u64 *src, int tid, u64 *lds)
{
    constexpr bool FIRST = (STRIDED != INVERSE);
    U2 v[NP];
    int l0[NP], l1[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        int x, lane, gi;
        ntt_pair<STRIDED>(a, g, tid + k * T, x, lane, gi, l0[k], l1[k]);
        if (FIRST && a.load_einv) v[k] = ld2_galois(src, (u32)gi, a.logn, a.load_einv, ntt_src_q(a, g));
        else v[k] = ld2_stream(src + gi);
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const U2 w = ntt_load_op<FIRST>(a, m, g.poly, v[k]);
        lds[l0[k]] = w.a;
        lds[l1[k]] = w.b;
    }
}

template <bool STRIDED, bool INVERSE, int CM = -1, int CC = -1, int T = NTT_THREADS>
HD void ntt_body_load(const NttArgs &a, int bx, int by, int tid, u64 *lds)
{
    constexpr bool FIRST = (STRIDED != INVERSE);
    const NttGeom g = ntt_geom<CM, CC>(a, bx, by);
    const ModDev m = mod_at_u(a.mods, g.mod_index);
    const u64 *src;
    if (FIRST) {
        const int ip = a.src_item_polys > 0 ? a.src_item_polys : a.count;
        src = a.src + (size_t)(g.poly / ip) * a.src_item_stride + (size_t)((g.poly % ip) / a.src_div) * g.n;
    } else src = a.dst + (size_t)g.poly * g.n;
    const int E2 = (g.M * g.C) >> 1;
    if (E2 == 8 * T) { ntt_load_full<STRIDED, INVERSE, 8, T>(a, g, m, src, tid, lds); return; }
    if (E2 == 4 * T) { ntt_load_full<STRIDED, INVERSE, 4, T>(a, g, m, src, tid, lds); return; }
    for (int e2 = tid; e2 < E2; e2 += T) {
        int x, lane, gi, l0, l1;
        ntt_pair<STRIDED>(a, g, e2, x, lane, gi, l0, l1);
        const U2 raw = (FIRST && a.load_einv) ? ld2_galois(src, (u32)gi, a.logn, a.load_einv, ntt_src_q(a, g)) : ld2(src + gi);
        const U2 v = ntt_load_op<FIRST>(a, m, g.poly, raw);
        lds[l0] = v.a;
        lds[l1] = v.b;
    }
}

// one register round: RHO stages starting at local stage S0 of a 2^LOGM transform.
// Forward (CT) butterflies X' = X + wY, Y' = X - wY + kq.  Harvey's schedule (LAZY8 = false, any modulus below 2^61): wY leaves
// the Shoup product in [0,2q) for ANY 64-bit Y, X is folded below 2q at every stage, [0,4q) invariant.
// LAZY8 (every modulus of the launch has the pseudo-Mersenne form ModDev::pm_ok, so 16q <= 2^64): the product uses the
// TRUNCATED high product (shoup_lazy_t: [0,4q) for any 64-bit Y, 4 instructions less than the exact one) and X is folded with
// pm_fold (below 2q for ANY 64-bit value, 3 instructions, no compare) at the first stage of a round, and at the third stage of a
// radix-16 round: a fold leaves X < 2q, every stage adds at most 4q to both outputs, so they stay below 2q + 3 * 4q = 14q < 2^64.  The values are
// congruent to Harvey's, so every fully reduced result is unchanged.
// TWL (row kernel, 256-point rows): the twiddles of stages 0..TWL_STAGES-1 of the tile's rows sit in LDS (ks_row_twiddle_fill)
// and the rounds inside those stages read them there instead of from the global tables.
constexpr int TWL_STAGES = 6;                                 // the first two radix-8 rounds of NttSched<8, 512>
constexpr int TWL_ROW = (1 << TWL_STAGES) + 1;                // heap index 2^s + block, one pad pair (rows on distinct 16-byte slots)
HD int twl_slot(int row, int h) { return row * TWL_ROW + h; }
template <int LOGM, int S0, int RHO, bool STRIDED, bool INVERSE, bool LAZY8 = false, int CC = -1, int T = NTT_THREADS, bool FW16 = false, bool TWL = false>
HD void ntt_body_round(const NttArgs &a, int bx, int by, int tid, u64 *lds, const u64 *twl = nullptr)
{
    const NttGeom g = ntt_geom<(CC >= 0 ? LOGM : -1), CC>(a, bx, by);
    const ModDev m = mod_at_u(a.mods, g.mod_index);
    const u64 q = m.q, q2 = q << 1, q4 = q << 2, nq = m.nq, nq2 = nq << 1;
    const gptr W = as_global(INVERSE ? m.iw : FW16 ? m.fw : m.w), WS = as_global(m.ws);
    constexpr int LO_BITS = LOGM - S0 - RHO;
    constexpr int RAD = 1 << RHO;
    const int groups = (g.M >> RHO) * g.C;
    for (int grp = tid; grp < groups; grp += T) {
        const int lane = grp & (g.C - 1);
        const int sub = grp >> g.logc;
        const int hi = sub >> LO_BITS;
        const int lo = sub & ((1 << LO_BITS) - 1);
        const int x0 = (hi << (LOGM - S0)) + lo;
        const int P = STRIDED ? 1 : g.N_over_M + g.tile * g.C + lane;
        const int tb = (P << S0) + hi;
        constexpr bool IN_LDS = TWL && !STRIDED && S0 + RHO <= TWL_STAGES;
        u64 v[RAD];
#pragma unroll
        for (int k = 0; k < RAD; k++) v[k] = lds[(x0 + (k << LO_BITS)) * g.pitch + lane];
        if (!INVERSE) {
#pragma unroll
            for (int u = 0; u < RHO; u++) {
                const int half = 1 << (RHO - 1 - u);
#pragma unroll
                for (int b = 0; b < (1 << u); b++) {
                    u64 w, ws;
                    if (IN_LDS) { const U2 tw = ld2(twl + 2 * twl_slot(lane, (1 << (S0 + u)) + (hi << u) + b)); w = tw.a; ws = tw.b; }
                    else if (FW16) { const U2 tw = ld2g(W + 2 * (size_t)((tb << u) + b)); w = tw.a; ws = tw.b; }
                    else { w = W[(tb << u) + b]; ws = WS[(tb << u) + b]; }
#pragma unroll
                    for (int j = 0; j < half; j++) {
                        const int k0 = b * 2 * half + j, k1 = k0 + half;
                        u64 x = v[k0];
                        if (LAZY8) {
                            if (u == 0 || (RHO == 4 && u == 2)) x = pm_fold(x, m);
                            const u64 y = shoup_lazy_t(v[k1], w, ws, nq);
                            v[k0] = add_nw(x, y);
                            v[k1] = sub_nn(add_nw(x, q4), y);
                        } else {
                            x = csub(x, nq2);
                            const u64 y = shoup_lazy_n(v[k1], w, ws, nq);
                            v[k0] = x + y;
                            v[k1] = x + q2 - y;
                        }
                    }
                }
            }
        } else {
            // Inverse (GS) butterflies X' = X + Y, Y' = (X - Y) w.  The product leaves the Shoup multiplication in [0,2q) ([0,4q) with
            // the truncated high product) whatever its input, so only the sums grow, doubling per stage.  Harvey folds every sum
            // back below 2q.  LAZY8 (16q <= 2^64) tracks the bound of each of the thread's values as a compile-time multiple of q
            // (bnd[]; the butterfly pattern of a register round is fixed), folds an operand with pm_fold only when a sum would pass
            // 16q, and restores the [0,2q) invariant once at the end of the round.
            int bnd[RAD];
#pragma unroll
            for (int k = 0; k < RAD; k++) bnd[k] = 2;
#pragma unroll
            for (int u = RHO - 1; u >= 0; u--) {
                const int half = 1 << (RHO - 1 - u);
#pragma unroll
                for (int b = 0; b < (1 << u); b++) {
                    const U2 tw = IN_LDS ? ld2(twl + 2 * twl_slot(lane, (1 << (S0 + u)) + (hi << u) + b)) : ld2g(W + 2 * (size_t)((tb << u) + b));
                    const u64 w = tw.a, ws = tw.b;
#pragma unroll
                    for (int j = 0; j < half; j++) {
                        const int k0 = b * 2 * half + j, k1 = k0 + half;
                        u64 x = v[k0], y = v[k1];
                        if (LAZY8) {
                            if (bnd[k0] + bnd[k1] > 16) { x = pm_fold(x, m); bnd[k0] = 2; }
                            if (bnd[k0] + bnd[k1] > 16) { y = pm_fold(y, m); bnd[k1] = 2; }
                            v[k0] = add_nw(x, y);
                            v[k1] = shoup_lazy_t(sub_nn(add_nw(x, q * (u64)bnd[k1]), y), w, ws, nq);  // + bnd q keeps the difference non-negative; below 16q
                            bnd[k0] = bnd[k0] + bnd[k1];
                            bnd[k1] = 4;
                        } else {
                            v[k0] = csub(x + y, nq2);
                            v[k1] = shoup_lazy_n(x + q2 - y, w, ws, nq);
                        }
                    }
                }
            }
            if (LAZY8) {
#pragma unroll
                for (int k = 0; k < RAD; k++)
                    if (bnd[k] > 2) v[k] = pm_fold(v[k], m);
            }
        }
#pragma unroll
        for (int k = 0; k < RAD; k++) lds[(x0 + (k << LO_BITS)) * g.pitch + lane] = v[k];
    }
}

// global operands of the fused store epilogues of one pair.  Fetched separately from their use so that a full tile can
// put the loads of several pairs in flight before the first store (stores to acc / aux_out may alias them as far as the
// compiler knows, which would otherwise serialise one memory round trip per pair).
struct StorePre { U2 d, s, acc, gp; };
template <bool STRIDED, bool INVERSE>
HD StorePre ntt_store_fetch(const NttArgs &a, const NttGeom &g, size_t pbase, int gi)
{
    constexpr bool LAST = (STRIDED == INVERSE);
    StorePre p = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if (!LAST) return p;
    if (INVERSE) {
        if (a.store_op == STORE_KS1) p.s = ld2(a.aux_r + ((size_t)(g.poly / a.L) * 2 + 1) * g.n + gi);
        else if (a.store_op == STORE_KSF) {
            const size_t bk = g.poly / a.L;  // (item, k)
            p.s = ld2(a.aux_r + bk * g.n + gi);
            if ((a.base_mask >> (bk & 1)) & 1) {
                const u64 *bp = a.aux_in + (bk >> 1) * a.base_stride + ((bk & 1) * a.L + g.poly % a.L) * g.n;
                p.d = a.gal_einv ? ld2_galois(bp, (u32)gi, a.logn, a.gal_einv, mod_at(a.mods, a.mod_base + g.poly % a.mod_cycle).q) : ld2(bp + gi);
            }
            if (a.acc) p.acc = ld2(a.acc + pbase + gi);
        }
    } else if (a.store_op == STORE_MUL || a.store_op == STORE_MAC) {
        const gptr mp = as_global(a.mul_ptrs ? a.mul_ptrs[g.poly / a.mul_item_polys] : a.mul);
        p.d = ld2g(mp + a.mul_shift + (size_t)(g.poly % a.mul_cycle) * g.n + gi);
        if (a.store_op == STORE_MAC) p.acc = ld2(a.acc + pbase + gi);
    } else if (a.store_op == STORE_KS0) {
        const int j = g.poly % a.L;
        const size_t item = g.poly / a.L;
        p.d = ld2g(as_global(a.mul_ptrs[item]) + a.mul_shift + (size_t)j * g.n + gi);
        p.s = ld2(a.aux_r + ((item * 2 + 0) * a.K + j) * g.n + gi);
        p.acc = ld2(a.acc + pbase + gi);
        p.gp = ld2_perm(a.aux_in + pbase, (u32)gi, a.logn, a.gal_elt);
    }
    return p;
}
template <bool STRIDED, bool INVERSE>
HD void ntt_store_pair(const NttArgs &a, const NttGeom &g, const ModDev &m, u64 *dst, size_t pbase, int gi, int l0, int l1,
                       const StorePre &pre, const u64 *lds)
{
    constexpr bool LAST = (STRIDED == INVERSE);
    const u64 q = m.q, q2 = q << 1;
    u64 v[2] = {lds[l0], lds[l1]};
    if (!LAST || (!INVERSE && a.store_op == STORE_LAZY)) { st2_stream(dst + gi, U2{v[0], v[1]}); return; }
    if (INVERSE) {
        const bool st = a.store_op == STORE_SCALE_T;
        if (a.store_op != STORE_KS1 && a.store_op != STORE_KSF)  // the mod-down epilogues fold N^-1 into their own product
        for (int k = 0; k < 2; k++) {
            v[k] = csub(shoup_lazy_n(v[k], st ? m.ninv_t : m.ninv, st ? m.ninv_t_s : m.ninv_s, m.nq), m.nq);
        }
        if (a.store_op == STORE_RSP) { v[0] = addmod(v[0], a.ks.half, q); v[1] = addmod(v[1], a.ks.half, q); }
        if (a.store_op == STORE_KSF) {  // Evaluator::switch_key_inplace mod-down, poly = (item, k, j)
            const int j = g.poly % a.L;
            const bool base = (a.base_mask >> ((g.poly / a.L) & 1)) & 1;
            const u64 rr[2] = {pre.s.a, pre.s.b}, bb[2] = {pre.d.a, pre.d.b};
            u64 o[2];
            for (int k = 0; k < 2; k++) {
                const u64 pa = shoup_lazy_n(v[k], a.ks.ninv_qinv[j], a.ks.ninv_qinv_s[j], m.nq);   // as in KS1 below
                const u64 pb = shoup_lazy_n(rr[k], a.ks.qsp_inv[j], a.ks.qsp_inv_s[j], m.nq);
                const u64 t = csub(csub(csub(pa + a.ks.hq2[j] - pb, m.nq << 2), m.nq << 1), m.nq);
                o[k] = base ? addmod(t, bb[k], q) : t;
            }
            st2(a.aux_out + pbase + gi, U2{o[0], o[1]});
            if (a.acc) st2(a.acc + pbase + gi, U2{addmod(pre.acc.a, o[0], q), addmod(pre.acc.b, o[1], q)});  // acc [B][2][L][N] += the finished ciphertext
            return;
        }
        else if (a.store_op == STORE_KS1) {
            const int j = g.poly % a.L;
            const u64 rr[2] = {pre.s.a, pre.s.b};
            for (int k = 0; k < 2; k++) {
                // (v N^-1 - r + half) q_sp^-1 mod q as two lazy products (any 64-bit input): no separate N^-1 scaling, no reduction of r
                const u64 pa = shoup_lazy_n(v[k], a.ks.ninv_qinv[j], a.ks.ninv_qinv_s[j], m.nq);
                const u64 pb = shoup_lazy_n(rr[k], a.ks.qsp_inv[j], a.ks.qsp_inv_s[j], m.nq);
                u64 o = csub(csub(csub(pa + a.ks.hq2[j] - pb, m.nq << 2), m.nq << 1), m.nq);
                u32 idx = (u32)(gi + k);
                if (a.gal_elt) {
                    const u64 raw = (u64)(gi + k) * a.gal_elt;
                    idx = (u32)(raw & (g.n - 1));
                    if ((raw >> a.logn) & 1) o = negmod(o, q);
                }
                a.aux_out[pbase + idx] = o;
            }
            return;
        }
    } else {
        // forward results arrive in [0,4q), or [0,14q) from LAZY8 rounds; a Barrett product takes them as they are
        if (a.store_op != STORE_MUL && a.store_op != STORE_MAC)
            for (int k = 0; k < 2; k++) {
                if (a.lazy8) v[k] = csub(pm_fold(v[k], m), m.nq);
                else v[k] = csub(csub(v[k], m.nq << 1), m.nq);
            }
        if (a.store_op == STORE_MUL || a.store_op == STORE_MAC) {
            const U2 d = pre.d;
            v[0] = mulmod(v[0], d.a, m);
            v[1] = mulmod(v[1], d.b, m);
            if (a.store_op == STORE_MAC) {
                U2 acc = pre.acc;
                acc.a = addmod(acc.a, v[0], q);
                acc.b = addmod(acc.b, v[1], q);
                st2(a.acc + pbase + gi, acc);
                return;
            }
        } else if (a.store_op == STORE_KS0) {
            const int j = g.poly % a.L;
            const U2 d = pre.d, s0 = pre.s;
            U2 acc = pre.acc;
            const u64 g0 = pre.gp.a, g1 = pre.gp.b;
            if (a.mul_s_off) {  // plaintext diagonal with its Shoup quotient: a lazy product in [0,2q) instead of a Barrett product
                const U2 ds = ld2g(as_global(a.mul_ptrs[g.poly / a.L]) + a.mul_shift + a.mul_s_off + (size_t)j * g.n + gi);
                acc.a = csub(csub(acc.a + shoup_lazy_n(g0, d.a, ds.a, m.nq), m.nq << 1), m.nq);
                acc.b = csub(csub(acc.b + shoup_lazy_n(g1, d.b, ds.b, m.nq), m.nq << 1), m.nq);
            } else {
                acc.a = addmod(acc.a, mulmod(g0, d.a, m), q);
                acc.b = addmod(acc.b, mulmod(g1, d.b, m), q);
            }
            st2(a.acc + pbase + gi, acc);
            U2 o;
            o.a = addmod(g0, shoup_mul(submod(s0.a, v[0], q), a.ks.qsp_inv[j], a.ks.qsp_inv_s[j], q), q);
            o.b = addmod(g1, shoup_mul(submod(s0.b, v[1], q), a.ks.qsp_inv[j], a.ks.qsp_inv_s[j], q), q);
            st2_stream<8>(a.aux_out + pbase + gi, o);
            return;
        }
    }
    st2(dst + gi, U2{v[0], v[1]});
}

// NP pairs per lane, compile-time: the epilogue operands of G pairs are fetched before the first of them is stored (G > 1
// trades registers for memory-level parallelism)
template <bool STRIDED, bool INVERSE, int NP, int T = NTT_THREADS>
HD void ntt_store_full(const NttArgs &a, const NttGeom &g, const ModDev &m, u64 *dst, size_t pbase, int tid, const u64 *lds)
{
#ifndef NTT_STORE_G
#define NTT_STORE_G 1  // measured on MI355X (in-call A/B, transcipherings/s): G=1 227, G=2 226, G=4 219 (the prefetched operands push the row pass past 128 VGPRs)
#endif
    constexpr int G = NTT_STORE_G;
#pragma unroll
    for (int k0 = 0; k0 < NP; k0 += G) {
        StorePre pre[G];
        int gi[G], l0[G], l1[G];
#pragma unroll
        for (int k = 0; k < G; k++) {
            int x, lane;
            ntt_pair<STRIDED>(a, g, tid + (k0 + k) * T, x, lane, gi[k], l0[k], l1[k]);
            pre[k] = ntt_store_fetch<STRIDED, INVERSE>(a, g, pbase, gi[k]);
        }
#pragma unroll
        for (int k = 0; k < G; k++) ntt_store_pair<STRIDED, INVERSE>(a, g, m, dst, pbase, gi[k], l0[k], l1[k], pre[k], lds);
    }
}
template <bool STRIDED, bool INVERSE, int CM = -1, int CC = -1, int T = NTT_THREADS>
HD void ntt_body_store(const NttArgs &a, int bx, int by, int tid, const u64 *lds)
{
    const NttGeom g = ntt_geom<CM, CC>(a, bx, by);
    const ModDev m = mod_at_u(a.mods, g.mod_index);
    u64 *dst = a.dst + (size_t)g.poly * g.n;
    const size_t pbase = (size_t)g.poly * g.n;
    const int E2 = (g.M * g.C) >> 1;
    if (E2 == 8 * T) { ntt_store_full<STRIDED, INVERSE, 8, T>(a, g, m, dst, pbase, tid, lds); return; }  // full tiles
    if (E2 == 4 * T) { ntt_store_full<STRIDED, INVERSE, 4, T>(a, g, m, dst, pbase, tid, lds); return; }
    for (int e2 = tid; e2 < E2; e2 += T) {
        int x, lane, gi, l0, l1;
        ntt_pair<STRIDED>(a, g, e2, x, lane, gi, l0, l1);
        const StorePre pre = ntt_store_fetch<STRIDED, INVERSE>(a, g, pbase, gi);
        ntt_store_pair<STRIDED, INVERSE>(a, g, m, dst, pbase, gi, l0, l1, pre, lds);
    }
}

// ------------------------------------------------------------------ element-wise

HD void elt_body(const EltArgs &a, int op, size_t gid)
{
    const int n = 1 << a.logn;
    const size_t p = gid >> a.logn;
    if (p >= (size_t)a.count) return;
    const size_t i = gid & (n - 1);
    const ModDev m = mod_at(a.mods, a.mod_base + (int)(p % a.mod_cycle));
    const size_t bp = a.b_cycle ? p % a.b_cycle : p;
    const u64 x = op == ELT_BCAST ? 0 : a.a[gid];
    if (op == ELT_SHOUP) { a.out[gid] = shoup_quotient(x, m); return; }
    u64 r;
    switch (op) {
    case ELT_BCAST: r = a.b[bp * n + i]; break;
    case ELT_ADD: r = addmod(x, a.b[bp * n + i], m.q); break;
    case ELT_SUB: r = submod(x, a.b[bp * n + i], m.q); break;
    case ELT_NEG: r = negmod(x, m.q); break;
    case ELT_MUL: r = mulmod(x, a.b[bp * n + i], m); break;
    case ELT_MAC: r = addmod(a.out[gid], mulmod(x, a.b[bp * n + i], m), m.q); break;
    default: r = x; break;
    }
    a.out[gid] = r;
}

// strided gather / scatter of whole ciphertexts (16 bytes per lane): gid over [count][words / 2]
HD void copy_items_body(const CopyItemsArgs &a, size_t gid)
{
    const size_t half = a.words >> 1;
    const size_t s = gid / half, w = (gid % half) << 1;
    if (s >= a.count) return;
    st2(a.dst + (s * a.dst_stride + a.dst_off) * a.words + w, ld2(a.src + (s * a.src_stride + a.src_off) * a.words + w));
}

// GaloisTool::apply_galois as a gather (seal/util/galois.h:32): gid over [count][N / 2], two adjacent output
// coefficients per lane (16-byte store / read-modify-write, two 8-byte gathers)
HD void galois_body(const GaloisArgs &a, size_t gid)
{
    const u32 n = 1u << a.logn;
    const size_t p = gid >> (a.logn - 1);
    if (p >= (size_t)a.count) return;
    const u32 k = (u32)(gid & ((n >> 1) - 1)) << 1;
    const size_t item = p / a.L, limb = p % a.L;
    const u64 q = mod_at(a.mods, limb).q;
    const u64 *src = a.in + item * a.in_item_stride + limb * n;
    const u32 j0 = (u32)(((u64)k * a.einv) & (2 * n - 1)), j1 = (u32)((j0 + a.einv) & (2 * n - 1));
    U2 v;
    v.a = (j0 < n) ? src[j0] : negmod(src[j0 - n], q);
    v.b = (j1 < n) ? src[j1] : negmod(src[j1 - n], q);
    u64 *o = a.out + item * a.out_item_stride + limb * n + k;
    if (a.accumulate) {
        const U2 c = ld2(o);
        v.a = addmod(c.a, v.a, q);
        v.b = addmod(c.b, v.b, q);
    }
    st2(o, v);
}

// NTT-domain Galois gather, optionally multiply-accumulating with a per-item table: gid over [count][N]
HD void perm_body(const PermArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t p = gid >> a.logn;
    if (p >= (size_t)a.count) return;
    const u32 x = (u32)(gid & (n - 1));
    const size_t item = p / a.L, j = p % a.L;
    const ModDev m = mod_at(a.mods, j);
    const u64 v = a.in[p * n + ntt_perm_index(x, a.logn, a.elt)];
    u64 *o = a.out + item * a.out_item_stride + j * n + x;
    if (a.mac) *o = addmod(*o, mulmod(v, a.mul_ptrs[item][a.mul_shift + j * n + x], m), m.q);
    else *o = v;
}

// key-switch inner product: gid over [B][K][N/2], two adjacent coefficients per lane (16-B accesses)
HD void ks_mac_body(const KsMacArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bj = gid >> (a.logn - 1);
    const int J = (int)(bj % a.K);
    const size_t b = bj / a.K;
    if (b >= (size_t)a.B) return;
    const ModDev m = mod_at_u(a.mods, J);
    Acc128 s0[2] = {{0, 0}, {0, 0}}, s1[2] = {{0, 0}, {0, 0}};
    u32 p0 = (u32)i;
    if (a.perm_elt) p0 = ntt_perm_index((u32)i, a.logn, a.perm_elt);
    for (int I = 0; I < a.L; I++) {
        const u64 *tp = a.T + ((b * a.L + I) * a.K + J) * n;
        U2 t = ld2(tp + (p0 & ~1u));  // shared digits: pair (pi(i), pi(i+1)) = aligned pair, possibly swapped (ld2_perm)
        if (p0 & 1) t = U2{t.b, t.a};
        const U2 k0 = ld2(a.key + (((size_t)I * 2 + 0) * a.K + J) * n + i);
        const U2 k1 = ld2(a.key + (((size_t)I * 2 + 1) * a.K + J) * n + i);
        if (a.acc && I == J) {  // the diagonal digit is NTT_J(galois(c1)): reuse it for the plain product
            const U2 d = ld2g(as_global(a.mul_ptrs[b]) + a.mul_shift + (size_t)J * n + i);
            u64 *ap = a.acc + (b * a.L + J) * n + i;
            U2 acc = ld2(ap);
            acc.a = addmod(acc.a, mulmod(t.a, d.a, m), m.q);
            acc.b = addmod(acc.b, mulmod(t.b, d.b, m), m.q);
            st2(ap, acc);
        }
        acc_mac(s0[0], t.a, k0.a); acc_mac(s0[1], t.b, k0.b);
        acc_mac(s1[0], t.a, k1.a); acc_mac(s1[1], t.b, k1.b);
        if ((I & 3) == 3) {  // q < 2^61: four products stay below 2^124
            for (int k = 0; k < 2; k++) {
                s0[k].lo = barrett128(s0[k].lo, s0[k].hi, m); s0[k].hi = 0;
                s1[k].lo = barrett128(s1[k].lo, s1[k].hi, m); s1[k].hi = 0;
            }
        }
    }
    if (a.corr) {
        const U2 e0 = ld2(a.corr + ((size_t)0 * a.K + J) * n + i), e1 = ld2(a.corr + ((size_t)1 * a.K + J) * n + i);
        acc_add(s0[0], e0.a); acc_add(s0[1], e0.b);
        acc_add(s1[0], e1.a); acc_add(s1[1], e1.b);
    }
    const U2 r0 = {barrett128(s0[0].lo, s0[0].hi, m), barrett128(s0[1].lo, s0[1].hi, m)};
    const U2 r1 = {barrett128(s1[0].lo, s1[0].hi, m), barrett128(s1[1].lo, s1[1].hi, m)};
    if (a.s_acc && J < a.L) {  // leaf of the FC rotation trie: only the sum over leaves is ever inverse-transformed
        u64 *p0 = a.s_acc + ((b * 2 + 0) * a.L + J) * n + i, *p1 = a.s_acc + ((b * 2 + 1) * a.L + J) * n + i;
        U2 c0 = ld2(p0), c1 = ld2(p1);
        c0.a = addmod(c0.a, r0.a, m.q); c0.b = addmod(c0.b, r0.b, m.q);
        c1.a = addmod(c1.a, r1.a, m.q); c1.b = addmod(c1.b, r1.b, m.q);
        st2(p0, c0); st2(p1, c1);
        return;
    }
    if (a.S_sp) {
        u64 *o0 = J == a.K - 1 ? a.S_sp + (b * 2 + 0) * n : a.S + ((b * 2 + 0) * a.L + J) * n;
        u64 *o1 = J == a.K - 1 ? a.S_sp + (b * 2 + 1) * n : a.S + ((b * 2 + 1) * a.L + J) * n;
        st2(o0 + i, r0); st2(o1 + i, r1);
        return;
    }
    st2(a.S + ((b * 2 + 0) * a.K + J) * n + i, r0);
    st2(a.S + ((b * 2 + 1) * a.K + J) * n + i, r1);
}

// The same inner product for a compile-time digit count LL <= 4 and a compile-time variant MODE (KS_PLAIN: write S;
// KS_ACC: + diagonal product of the fused matmul; KS_PERM: shared digits read through the Galois index map + correction
// table; KS_LEAF: KS_PERM with the data limbs summed into s_acc).  Straight-line code: every global load of the lane is
// issued before the first dependent instruction, which is what a bandwidth-bound kernel at 4 waves/SIMD needs (the generic
// body above makes L + 1 dependent round trips).
enum { KS_PLAIN = 0, KS_ACC = 1, KS_PERM = 2, KS_LEAF = 3 };
template <int LL, int MODE> HD void ks_mac_body_t(const KsMacArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bj = gid >> (a.logn - 1);
    const int J = (int)(bj % a.K);
    const size_t b = bj / a.K;
    if (b >= (size_t)a.B) return;
    ModDev m;  // the three words the Barrett reductions need, fetched with the first wave of loads
    m.q = mod_at_u(a.mods, J).q; m.r_lo = mod_at_u(a.mods, J).r_lo; m.r_hi = mod_at_u(a.mods, J).r_hi;
    constexpr bool PERM = MODE == KS_PERM || MODE == KS_LEAF;
    u32 p0 = (u32)i;
    if (PERM) p0 = ntt_perm_index((u32)i, a.logn, a.perm_elt);
    const bool diag = MODE == KS_ACC && J < LL, leaf = MODE == KS_LEAF && J < LL;
    const u64 *dptr = diag ? a.mul_ptrs[b] : nullptr;  // pointer first: the operand load behind it is a dependent round trip
    U2 t[LL], k0[LL], k1[LL];
#pragma unroll
    for (int I = 0; I < LL; I++) {
        t[I] = ld2_stream<16>(a.T + ((b * LL + I) * a.K + J) * n + (PERM ? (size_t)(p0 & ~1u) : i));
        k0[I] = ld2(a.key + (((size_t)I * 2 + 0) * a.K + J) * n + i);
        k1[I] = ld2(a.key + (((size_t)I * 2 + 1) * a.K + J) * n + i);
    }
    U2 d = {0, 0}, accv = {0, 0}, e0 = {0, 0}, e1 = {0, 0}, c0 = {0, 0}, c1 = {0, 0};
    u64 *ap = a.acc + (b * LL + J) * n + i;
    u64 *q0 = a.s_acc + ((b * 2 + 0) * LL + J) * n + i, *q1 = a.s_acc + ((b * 2 + 1) * LL + J) * n + i;
    if (diag) { d = ld2_stream<32>(dptr + a.mul_shift + (size_t)J * n + i); accv = ld2(ap); }
    if (PERM) { e0 = ld2(a.corr + ((size_t)0 * a.K + J) * n + i); e1 = ld2(a.corr + ((size_t)1 * a.K + J) * n + i); }
    if (leaf) { c0 = ld2(q0); c1 = ld2(q1); }
    Acc128 s0[2] = {{e0.a, 0}, {e0.b, 0}}, s1[2] = {{e1.a, 0}, {e1.b, 0}};
    U2 tj = t[0];
#pragma unroll
    for (int I = 0; I < LL; I++) {
        if (PERM && (p0 & 1)) t[I] = U2{t[I].b, t[I].a};
        if (I == J) tj = t[I];
        acc_mac(s0[0], t[I].a, k0[I].a); acc_mac(s0[1], t[I].b, k0[I].b);
        acc_mac(s1[0], t[I].a, k1[I].a); acc_mac(s1[1], t[I].b, k1[I].b);
    }
    const U2 r0 = {barrett128(s0[0].lo, s0[0].hi, m), barrett128(s0[1].lo, s0[1].hi, m)};
    const U2 r1 = {barrett128(s1[0].lo, s1[0].hi, m), barrett128(s1[1].lo, s1[1].hi, m)};
    if (diag) {  // the diagonal digit is NTT_J(galois(c1)): reuse it for the plain product
        accv.a = addmod(accv.a, mulmod(tj.a, d.a, m), m.q);
        accv.b = addmod(accv.b, mulmod(tj.b, d.b, m), m.q);
        st2(ap, accv);
    }
    if (leaf) {
        c0.a = addmod(c0.a, r0.a, m.q); c0.b = addmod(c0.b, r0.b, m.q);
        c1.a = addmod(c1.a, r1.a, m.q); c1.b = addmod(c1.b, r1.b, m.q);
        st2(q0, c0); st2(q1, c1);
        return;
    }
    if (a.S_sp) {
        u64 *o0 = J == a.K - 1 ? a.S_sp + (b * 2 + 0) * n : a.S + ((b * 2 + 0) * LL + J) * n;
        u64 *o1 = J == a.K - 1 ? a.S_sp + (b * 2 + 1) * n : a.S + ((b * 2 + 1) * LL + J) * n;
        st2_stream<4>(o0 + i, r0); st2_stream<4>(o1 + i, r1);
        return;
    }
    st2_stream<4>(a.S + ((b * 2 + 0) * a.K + J) * n + i, r0);
    st2_stream<4>(a.S + ((b * 2 + 1) * a.K + J) * n + i, r1);
}
// variant of a launch (-1: none of the specialisations applies, use the generic body)
inline int ks_mac_mode(const KsMacArgs &a)
{
    if (a.L < 1 || a.L > 4) return -1;
    if (a.perm_elt) return (a.corr && !a.acc) ? (a.s_acc ? KS_LEAF : KS_PERM) : -1;
    if (a.s_acc || a.corr) return -1;
    return a.acc ? KS_ACC : KS_PLAIN;
}

// gid over [2][K][N]
HD void ks_corr_body(const KsCorrArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    if (gid >= (size_t)2 * a.K * n) return;
    const size_t x = gid & (n - 1);
    const int J = (int)((gid >> a.logn) % a.K), k = (int)(gid >> a.logn) / a.K;
    const ModDev m = mod_at_u(a.mods, J);
    u64 sum = 0;
    for (int I = 0; I < a.L; I++)
        if (I != J) sum = addmod(sum, mulmod(a.key[(((size_t)I * 2 + k) * a.K + J) * n + x], a.qmod[I * a.K + J], m), m.q);
    a.corr[gid] = mulmod(sum, a.shat[(size_t)J * n + x], m);
}

// ------------------------------------------------------------------ fused key-switch row kernel (KsRowArgs)
// Phases of one workgroup (b, J, row tile); fa / ia are the argument blocks of the forward / inverse row pass whose
// geometry and moduli the shared round bodies read (poly index `by` only selects the modulus there).
// Tile = 512 points on ONE wave (64 lanes x 8 points): sixteen independent single-wave workgroups per CU, whose barriers cost
// nothing and whose phases interleave freely.  Measured (transcipherings/s, one box, three alternating runs each): 2048 points
// on 256 lanes 292.8, 1024 points on 128 lanes 294.6, 512 points on 64 lanes 297.5.
#ifndef KSROW_TL
#define KSROW_TL 9
#endif
constexpr int KSROW_TILE_LOG = KSROW_TL;                             // points per tile of this kernel (log2)
constexpr int KSROW_THREADS = (1 << KSROW_TILE_LOG) / 8;             // 8 points per lane: 2 x 8 lazy sums + a radix-8 round fit the 128-VGPR budget of 4 waves per SIMD.
                                                                     // (4 points per lane on 256-point tiles, radix-4 rounds: 90 VGPRs, 5-6 waves per SIMD -- measured 481-484 us per
                                                                     // launch against 433: one more LDS round trip per transform and twice the per-workgroup start-up)
constexpr int KSROW_SCHED = 512;                                     // NttSched selector of the 8-points-per-lane (radix-8/4) schedules
constexpr int KSROW_NP = (1 << KSROW_TILE_LOG) / 2 / KSROW_THREADS;  // pairs per lane of a tile
constexpr int KSROW_LDS = (1 << KSROW_TILE_LOG) + 256;  // words: M rows of pitch C + 1 = 2^TL + M, M <= 256
// (w, ws) of stages 0..TWL_STAGES-1 for the C rows of this tile -> LDS, heap order per row.  The table index of stage s,
// block k of row P is (P << s) + k (the same index ntt_body_round forms), P = N/M + global row.
constexpr int KSROW_TWL = (1 << (KSROW_TILE_LOG - 8)) * TWL_ROW * 2;  // words, for 256-point rows
template <int CM, int CC>
HD void ks_row_twiddle_fill(const NttArgs &a, int bx, int J, bool inverse, int tid, u64 *twl)
{
    const NttGeom g = ntt_geom<CM, CC>(a, bx, J);
    const ModDev m = mod_at_u(a.mods, g.mod_index);
    const gptr tab = as_global(inverse ? m.iw : m.fw);
    for (int e = tid; e < (g.C << TWL_STAGES); e += KSROW_THREADS) {
        const int row = e >> TWL_STAGES, h = e & ((1 << TWL_STAGES) - 1);
        if (h == 0) continue;
        int s = 0;
        while ((2 << s) <= h) s++;
        const int P = g.N_over_M + g.tile * g.C + row;
        const U2 tw = ld2g(tab + 2 * (size_t)((P << s) + (h - (1 << s))));
        st2(twl + 2 * (size_t)twl_slot(row, h), tw);
    }
}
// The row pass of a digit transform reads the strided pass's output as it is (no load op in a second pass): the tile of the NEXT
// digit is requested (fetch) before the rounds and key products of the current one and written to LDS (commit) after them, so
// the wave never waits for a tile -- 2 * KSROW_NP more live VGPRs across the phases (the truncated products made room for them)
template <int CM, int CC>
HD void ks_row_tile_fetch(const NttArgs &fa, int bx, int by, int tid, U2 *pf)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, by);
    const u64 *src = fa.dst + (size_t)g.poly * g.n;
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        int xx, lane, gi, l0, l1;
        ntt_pair<false>(fa, g, tid + k * KSROW_THREADS, xx, lane, gi, l0, l1);
        pf[k] = ld2_stream(src + gi);
    }
}
template <int CM, int CC>
HD void ks_row_tile_commit(const NttArgs &fa, int bx, int by, int tid, const U2 *pf, u64 *lds)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, by);
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        int xx, lane, gi, l0, l1;
        ntt_pair<false>(fa, g, tid + k * KSROW_THREADS, xx, lane, gi, l0, l1);
        lds[l0] = pf[k].a;
        lds[l1] = pf[k].b;
    }
}
// after the forward rounds of digit I: acc_k[pair] += T * key[I][k][J].  The kernel only runs on pseudo-Mersenne moduli (ModDev::pm_ok,
// 16q <= 2^64): truncated Shoup products in [0,4q), sums folded below 2q by pm_fold after every third digit, so they never
// exceed 2q + 3 * 4q = 14q < 2^64
#ifndef KSROW_DEPTH
#define KSROW_DEPTH 1   // measured (config 2, /s, row kernel us): depth 1 320 / 428, depth 2 (9 spilled registers) 310 / 451, loads fenced by the stores 318 / 438
#endif
struct KsIdx { int gi, l0, l1; };
template <int CM, int CC> HD KsIdx ks_row_idx(const NttArgs &fa, const NttGeom &g, int tid, int k)
{
    int xx, lane;
    KsIdx r;   // recomputed where needed (a few integer operations) rather than kept in 3 x NP registers
    ntt_pair<false>(fa, g, tid + k * KSROW_THREADS, xx, lane, r.gi, r.l0, r.l1);
    return r;
}
template <int CM, int CC>
HD void ks_row_mac_phase(const KsRowArgs &x, const NttArgs &fa, int bx, int b, int J, int I, int tid, const u64 *lds, u64 *acc0, u64 *acc1)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, J);
    const ModDev m = mod_at_u(fa.mods, J);
    const u64 nq = m.nq;
    const size_t kofs = (((size_t)I * 2) * x.K + J) * g.n, kstep = (size_t)x.K * g.n;
    const bool diag = x.acc && I == J;
    const gptr dptr = diag ? as_global(x.mul_ptrs[b]) + x.mul_shift + (size_t)J * g.n : as_global(nullptr);
    u64 *ap = diag ? x.acc + ((size_t)b * x.L + J) * g.n : nullptr;
    const bool fold = (I % 3) == 2 && I != x.L - 1;   // the flush phase folds after the last digit
    // The key words of a pair are requested KSROW_DEPTH pairs ahead of their products, and the stores of the plain product come
    // after every key load of the phase (a store fences the loads behind it for the compiler): the wave waits for about one
    // round trip per phase instead of one per pair.
    constexpr int DEPTH = KSROW_DEPTH < KSROW_NP ? KSROW_DEPTH : KSROW_NP;
    U2 k0[KSROW_NP], k0s[KSROW_NP], k1[KSROW_NP], k1s[KSROW_NP];
    auto request = [&](int k) {
        const int gi = ks_row_idx<CM, CC>(fa, g, tid, k).gi;
        k0[k] = ld2(x.key + kofs + gi); k0s[k] = ld2(x.key_s + kofs + gi);
        k1[k] = ld2(x.key + kofs + kstep + gi); k1s[k] = ld2(x.key_s + kofs + kstep + gi);
    };
#pragma unroll
    for (int k = 0; k < DEPTH; k++) request(k);
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        const KsIdx ix = ks_row_idx<CM, CC>(fa, g, tid, k);
        const u64 v0 = lds[ix.l0], v1 = lds[ix.l1];
        acc0[2 * k] = add_nw(acc0[2 * k], shoup_lazy_t(v0, k0[k].a, k0s[k].a, nq));
        acc0[2 * k + 1] = add_nw(acc0[2 * k + 1], shoup_lazy_t(v1, k0[k].b, k0s[k].b, nq));
        acc1[2 * k] = add_nw(acc1[2 * k], shoup_lazy_t(v0, k1[k].a, k1s[k].a, nq));
        acc1[2 * k + 1] = add_nw(acc1[2 * k + 1], shoup_lazy_t(v1, k1[k].b, k1s[k].b, nq));
        if (k + DEPTH < KSROW_NP) request(k + DEPTH);
        if (fold) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                acc0[2 * k + e] = pm_fold(acc0[2 * k + e], m);
                acc1[2 * k + e] = pm_fold(acc1[2 * k + e], m);
            }
        }
    }
    if (diag) {  // the diagonal digit is NTT_J(galois(c1)): reuse it for the plain product (lazy input); one pair requested ahead
        U2 d[KSROW_NP], ds[KSROW_NP], ac[KSROW_NP];
        auto drequest = [&](int k) {
            const int gi = ks_row_idx<CM, CC>(fa, g, tid, k).gi;
            d[k] = ld2g(dptr + gi);
            ac[k] = ld2(ap + gi);
            ds[k] = x.mul_s_off ? ld2g(dptr + x.mul_s_off + gi) : U2{0, 0};
        };
        drequest(0);
#pragma unroll
        for (int k = 0; k < KSROW_NP; k++) {
            const KsIdx ix = ks_row_idx<CM, CC>(fa, g, tid, k);
            const u64 v0 = lds[ix.l0], v1 = lds[ix.l1];
            if (x.mul_s_off) {  // Shoup product with the table's quotients: canonical sum + [0,4q) -> fold below 2q -> canonical
                ac[k].a = csub(pm_fold(add_nw(ac[k].a, shoup_lazy_t(v0, d[k].a, ds[k].a, nq)), m), nq);
                ac[k].b = csub(pm_fold(add_nw(ac[k].b, shoup_lazy_t(v1, d[k].b, ds[k].b, nq)), m), nq);
            } else {
                ac[k].a = addmod(ac[k].a, mulmod(v0, d[k].a, m), m.q);
                ac[k].b = addmod(ac[k].b, mulmod(v1, d[k].b, m), m.q);
            }
            st2(ap + ix.gi, ac[k]);
            if (k + 1 < KSROW_NP) drequest(k + 1);
        }
    }
}
// The same products for a child of an FC trie node (shared digits): the digit transforms T of the node's UN-rotated c1 are complete
// (both passes) and read through the NTT-domain Galois map -- no tile, no rounds; the map sends aligned pairs to aligned pairs (possibly
// swapped), so a lane's pair is one 16-byte load.  The correction table of the key (KsCorrArgs) joins the sums after the last digit.
template <int CM, int CC>
HD void ks_row_mac_gather(const KsRowArgs &x, const NttArgs &fa, int bx, int b, int J, int I, int tid, u64 *acc0, u64 *acc1)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, J);
    const ModDev m = mod_at_u(fa.mods, J);
    const u64 nq = m.nq;
    const size_t kofs = (((size_t)I * 2) * x.K + J) * g.n, kstep = (size_t)x.K * g.n;
    const u64 *T = x.T + (((size_t)b * x.L + I) * x.K + J) * g.n;
    const bool fold = (I % 3) == 2 && I != x.L - 1;
    U2 t[KSROW_NP], k0[KSROW_NP], k0s[KSROW_NP], k1[KSROW_NP], k1s[KSROW_NP];
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {   // every gathered pair of the digit is requested before the first product
        const u32 p0 = ntt_perm_index((u32)ks_row_idx<CM, CC>(fa, g, tid, k).gi, fa.logn, x.perm_elt);
        t[k] = ld2(T + (p0 & ~1u));
        if (p0 & 1) t[k] = U2{t[k].b, t[k].a};
    }
    auto request = [&](int k) {
        const int gi = ks_row_idx<CM, CC>(fa, g, tid, k).gi;
        k0[k] = ld2(x.key + kofs + gi); k0s[k] = ld2(x.key_s + kofs + gi);
        k1[k] = ld2(x.key + kofs + kstep + gi); k1s[k] = ld2(x.key_s + kofs + kstep + gi);
    };
    request(0);
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        if (k + 1 < KSROW_NP) request(k + 1);
        acc0[2 * k] = add_nw(acc0[2 * k], shoup_lazy_t(t[k].a, k0[k].a, k0s[k].a, nq));
        acc0[2 * k + 1] = add_nw(acc0[2 * k + 1], shoup_lazy_t(t[k].b, k0[k].b, k0s[k].b, nq));
        acc1[2 * k] = add_nw(acc1[2 * k], shoup_lazy_t(t[k].a, k1[k].a, k1s[k].a, nq));
        acc1[2 * k + 1] = add_nw(acc1[2 * k + 1], shoup_lazy_t(t[k].b, k1[k].b, k1s[k].b, nq));
        if (fold) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                acc0[2 * k + e] = pm_fold(acc0[2 * k + e], m);
                acc1[2 * k + e] = pm_fold(acc1[2 * k + e], m);
            }
        }
    }
    if (I == x.L - 1) {   // + corr[k][J] (canonical words): the sums stay below 2q + 3 * 4q + q
        const u64 *e0 = x.corr + ((size_t)0 * x.K + J) * g.n, *e1 = x.corr + ((size_t)1 * x.K + J) * g.n;
#pragma unroll
        for (int k = 0; k < KSROW_NP; k++) {
            const int gi = ks_row_idx<CM, CC>(fa, g, tid, k).gi;
            const U2 c0 = ld2(e0 + gi), c1 = ld2(e1 + gi);
            acc0[2 * k] = add_nw(acc0[2 * k], c0.a); acc0[2 * k + 1] = add_nw(acc0[2 * k + 1], c0.b);
            acc1[2 * k] = add_nw(acc1[2 * k], c1.a); acc1[2 * k + 1] = add_nw(acc1[2 * k + 1], c1.b);
        }
        if (x.c0hat && J < x.L) {   // + q_sp * NTT_J(galois(c0)): c0hat holds q_sp * NTT(c0) (canonical words: one more q on the bound above)
            const u64 *ch = x.c0hat + ((size_t)b * x.L + J) * g.n;
            U2 v[KSROW_NP];
#pragma unroll
            for (int k = 0; k < KSROW_NP; k++) {
                const u32 p0 = ntt_perm_index((u32)ks_row_idx<CM, CC>(fa, g, tid, k).gi, fa.logn, x.perm_elt);
                v[k] = ld2(ch + (p0 & ~1u));
                if (p0 & 1) v[k] = U2{v[k].b, v[k].a};
            }
#pragma unroll
            for (int k = 0; k < KSROW_NP; k++) {
                acc0[2 * k] = add_nw(acc0[2 * k], v[k].a);
                acc0[2 * k + 1] = add_nw(acc0[2 * k + 1], v[k].b);
            }
        }
    }
}
// sums -> LDS in [0,2q) (input range of the inverse rounds); optionally also canonical to global (S_0[j])
template <int CM, int CC>
HD void ks_row_flush_phase(const NttArgs &fa, int bx, int J, int tid, u64 *lds, const u64 *acc, u64 *canon_out)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, J);
    const ModDev m = mod_at_u(fa.mods, J);
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        const KsIdx ix = ks_row_idx<CM, CC>(fa, g, tid, k);
        u64 v0 = pm_fold(acc[2 * k], m), v1 = pm_fold(acc[2 * k + 1], m);
        if (canon_out) {
            v0 = csub(v0, m.nq); v1 = csub(v1, m.nq);
#ifdef KSROW_S0_STREAM
            st2_stream(canon_out + ix.gi, U2{v0, v1});
#else
            st2(canon_out + ix.gi, U2{v0, v1});   // plain: a non-temporal store here took ~2.4x as long to complete (17,000 vs 7,000 cycles under load)
#endif
        } else { lds[ix.l0] = v0; lds[ix.l1] = v1; }
    }
}
// inverse row pass output (first inverse pass: plain stream store of the tile)
template <int CM, int CC>
HD void ks_row_store_phase(const NttArgs &fa, int bx, int J, int tid, const u64 *lds, u64 *out)
{
    const NttGeom g = ntt_geom<CM, CC>(fa, bx, J);
#pragma unroll
    for (int k = 0; k < KSROW_NP; k++) {
        int xx, lane, gi, l0, l1;
        ntt_pair<false>(fa, g, tid + k * KSROW_THREADS, xx, lane, gi, l0, l1);
        st2_stream(out + gi, U2{lds[l0], lds[l1]});
    }
}

// closes the leaf sums of the FC rotation trie: gid over [B][2][L][N]
HD void leaf_sum_body(const LeafSumArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    size_t r = gid >> a.logn;
    const int j = (int)(r % a.L); r /= a.L;
    const int k = (int)(r & 1);
    const size_t b = r >> 1;
    if (b >= (size_t)a.B) return;
    const ModDev m = mod_at(a.mods, j);
    u64 v = addmod(a.accS[gid], a.accH[gid], m.q);
    v = shoup_mul(v, a.ks.qsp_inv[j], a.ks.qsp_inv_s[j], m.q);
    a.out[gid] = addmod(a.out[gid], v, m.q);
}

// Rounding terms of a leaf key switch of the FC rotation trie, element-wise: gid over [B][2][N/2], two adjacent coefficients per lane.
// Every load of a lane (r, the L sums, the L Galois-mapped c0 words) is requested before its first store.  This was the epilogue of
// the special limb's last inverse pass until round 3: there every limb's read-modify-write sat behind the previous limb's store in the
// wave's in-order memory counter, 24 dependent round trips per lane and 355 us per launch at 2.2 TB/s.
// M = leaves of the launch as a compile-time constant: the loads of ALL leaves (r and the Galois-mapped c0 words, 8 bytes each from a
// different cache line) are in flight before the first of them is used
template <int JC, int M>
HD void leaf_round_limbs(const LeafRoundArgs &a, size_t bk, size_t i, int j0, bool with_c0)
{
    const size_t n = (size_t)1 << a.logn;
    U2 ac[JC], r[M], c0[M][JC];
#pragma unroll
    for (int jj = 0; jj < JC; jj++)
        if (j0 + jj < a.L) ac[jj] = ld2(a.accH + (bk * a.L + j0 + jj) * n + i);
#pragma unroll
    for (int l = 0; l < M; l++) {
        r[l] = ld2(a.r + (bk * M + l) * n + i);
#pragma unroll
        for (int jj = 0; jj < JC; jj++)
            if (with_c0 && a.base[l] && j0 + jj < a.L)
                c0[l][jj] = ld2_galois(a.base[l] + (bk >> 1) * a.base_stride + (size_t)(j0 + jj) * n, (u32)i, a.logn, a.gal_einv[l], mod_at(a.mods, j0 + jj).q);
    }
#pragma unroll
    for (int l = 0; l < M; l++) {
#pragma unroll
        for (int jj = 0; jj < JC; jj++) {
            const int j = j0 + jj;
            if (j < a.L) {
                const ModDev mj = mod_at(a.mods, j);
                ac[jj].a = addmod(ac[jj].a, submod(a.ks.half_mod[j], reduce64(r[l].a, mj), mj.q), mj.q);
                ac[jj].b = addmod(ac[jj].b, submod(a.ks.half_mod[j], reduce64(r[l].b, mj), mj.q), mj.q);
                if (with_c0 && a.base[l]) {  // k = 0: + q_sp * galois(c0)[j] (the sum is multiplied by q_sp^-1 when it is closed); null: the c0 terms come from the per-element sums (csum_c0)
                    ac[jj].a = addmod(ac[jj].a, shoup_mul(c0[l][jj].a, a.ks.qsp_mod[j], a.ks.qsp_mod_s[j], mj.q), mj.q);
                    ac[jj].b = addmod(ac[jj].b, shoup_mul(c0[l][jj].b, a.ks.qsp_mod[j], a.ks.qsp_mod_s[j], mj.q), mj.q);
                }
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < JC; jj++)
        if (j0 + jj < a.L) st2(a.accH + (bk * a.L + j0 + jj) * n + i, ac[jj]);
}
template <int M> HD void leaf_round_body_m(const LeafRoundArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bk = gid >> (a.logn - 1);   // (item, k)
    if (bk >= (size_t)a.B * 2) return;
    const bool with_c0 = !(bk & 1);
    for (int j0 = 0; j0 < a.L; j0 += 4) leaf_round_limbs<4, M>(a, bk, i, j0, with_c0);
}
HD void leaf_round_body(const LeafRoundArgs &a, size_t gid)
{
    static_assert(HHE_LEAF_GROUP == 4, "leaf_round_body dispatches on m = 1..4");
    switch (a.m) {
    case 1: leaf_round_body_m<1>(a, gid); break;
    case 2: leaf_round_body_m<2>(a, gid); break;
    case 3: leaf_round_body_m<3>(a, gid); break;
    default: leaf_round_body_m<4>(a, gid); break;
    }
}

// gid over [B][K][N/2] (as ks_mac_body_t): the inner products of m leaf children of one node, see KsMacLeavesArgs
template <int LL> HD void ks_mac_leaves_body(const KsMacLeavesArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bj = gid >> (a.logn - 1);
    const int J = a.sp_only ? a.K - 1 : (int)(bj % a.K);
    const size_t b = a.sp_only ? bj : bj / a.K;
    if (b >= (size_t)a.B) return;
    ModDev m;
    m.q = mod_at_u(a.mods, J).q; m.r_lo = mod_at_u(a.mods, J).r_lo; m.r_hi = mod_at_u(a.mods, J).r_hi;
    const bool data = J < LL;
    u64 *q0 = a.s_acc + ((b * 2 + 0) * LL + J) * n + i, *q1 = a.s_acc + ((b * 2 + 1) * LL + J) * n + i;
    U2 c0 = {0, 0}, c1 = {0, 0};
    if (data) { c0 = ld2(q0); c1 = ld2(q1); }
    for (int l = 0; l < a.m; l++) {
        const u32 p0 = ntt_perm_index((u32)i, a.logn, a.perm_elt[l]);
        const u64 *key = a.key[l], *corr = a.corr[l], *T = a.T[l];
        const int tk = a.t_polys[l], tj = tk == 1 ? 0 : J;
        U2 t[LL], k0[LL], k1[LL];
#pragma unroll
        for (int I = 0; I < LL; I++) {
            t[I] = ld2(T + ((b * LL + I) * tk + tj) * n + (size_t)(p0 & ~1u));   // leaves of one parent re-read the same words: cached
            k0[I] = ld2(key + (((size_t)I * 2 + 0) * a.K + J) * n + i);
            k1[I] = ld2(key + (((size_t)I * 2 + 1) * a.K + J) * n + i);
        }
        const U2 e0 = ld2(corr + ((size_t)0 * a.K + J) * n + i), e1 = ld2(corr + ((size_t)1 * a.K + J) * n + i);
        Acc128 s0[2] = {{e0.a, 0}, {e0.b, 0}}, s1[2] = {{e1.a, 0}, {e1.b, 0}};
#pragma unroll
        for (int I = 0; I < LL; I++) {
            if (p0 & 1) t[I] = U2{t[I].b, t[I].a};
            acc_mac(s0[0], t[I].a, k0[I].a); acc_mac(s0[1], t[I].b, k0[I].b);
            acc_mac(s1[0], t[I].a, k1[I].a); acc_mac(s1[1], t[I].b, k1[I].b);
        }
        const U2 r0 = {barrett128(s0[0].lo, s0[0].hi, m), barrett128(s0[1].lo, s0[1].hi, m)};
        const U2 r1 = {barrett128(s1[0].lo, s1[0].hi, m), barrett128(s1[1].lo, s1[1].hi, m)};
        if (data) {
            c0.a = addmod(c0.a, r0.a, m.q); c0.b = addmod(c0.b, r0.b, m.q);
            c1.a = addmod(c1.a, r1.a, m.q); c1.b = addmod(c1.b, r1.b, m.q);
        } else {
            st2_stream<4>(a.S_sp + ((b * 2 + 0) * a.m + l) * n + i, r0);
            st2_stream<4>(a.S_sp + ((b * 2 + 1) * a.m + l) * n + i, r1);
        }
    }
    if (data) { st2(q0, c0); st2(q1, c1); }
}

// gid over [B][L][N/2], see CsumArgs
HD void csum_add_body(const CsumArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bI = gid >> (a.logn - 1);
    const size_t b = bI / a.L, I = bI % a.L;
    if (b >= (size_t)a.B) return;
    U2 s = ld2(a.sums + bI * n + i), v[HHE_CSUM_GROUP], s0 = {0, 0}, v0[HHE_CSUM_GROUP];
    if (a.sums0) s0 = ld2(a.sums0 + bI * n + i);
#pragma unroll
    for (int l = 0; l < HHE_CSUM_GROUP; l++)
        if (l < a.m) {
            v[l] = ld2(a.src[l] + b * a.src_stride + (a.L + I) * n + i);
            if (a.sums0) v0[l] = ld2(a.src[l] + b * a.src_stride + I * n + i);
        }
    unsigned char *cp = a.carry + bI * n + i;
    unsigned ca = cp[0], cb = cp[1];
#pragma unroll
    for (int l = 0; l < HHE_CSUM_GROUP; l++)
        if (l < a.m) {
            s.a += v[l].a; ca += s.a < v[l].a;
            s.b += v[l].b; cb += s.b < v[l].b;
        }
    st2(a.sums + bI * n + i, s);
    cp[0] = (unsigned char)ca; cp[1] = (unsigned char)cb;
    if (a.sums0) {
        const u64 q = mod_at(a.mods, (int)I).q;
#pragma unroll
        for (int l = 0; l < HHE_CSUM_GROUP; l++)
            if (l < a.m) { s0.a = addmod(s0.a, v0[l].a, q); s0.b = addmod(s0.b, v0[l].b, q); }
        st2(a.sums0 + bI * n + i, s0);
    }
}
// gid over [B][L][N/2]: accH[b][0][j] += q_sp * galois(sums0[b][j])
HD void csum_c0_body(const CsumArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = (gid & ((n >> 1) - 1)) << 1;
    const size_t bj = gid >> (a.logn - 1);
    const size_t b = bj / a.L;
    const int j = (int)(bj % a.L);
    if (b >= (size_t)a.B) return;
    const u64 q = mod_at(a.mods, j).q;
    u64 *ap = a.accH + ((b * 2 + 0) * a.L + j) * n + i;
    U2 acc = ld2(ap);
    const U2 g = ld2_galois(a.sums0 + bj * n, (u32)i, a.logn, a.einv, q);
    acc.a = addmod(acc.a, shoup_mul(g.a, a.qsp_mod[j], a.qsp_mod_s[j], q), q);
    acc.b = addmod(acc.b, shoup_mul(g.b, a.qsp_mod[j], a.qsp_mod_s[j], q), q);
    st2(ap, acc);
}
HD void csum_digits_body(const CsumArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const u32 i = (u32)((gid & ((n >> 1) - 1)) << 1);
    const size_t bI = gid >> (a.logn - 1);
    const size_t b = bI / a.L, I = bI % a.L;
    if (b >= (size_t)a.B) return;
    const u64 *src = a.sums + bI * n;
    const unsigned char *car = a.carry + bI * n;
    const u64 qI = mod_at(a.mods, (int)I).q;
    const u32 j0 = (u32)(((u64)i * a.einv) & (2 * n - 1)), j1 = (u32)((j0 + a.einv) & (2 * n - 1));
    const bool f0 = j0 >= n, f1 = j1 >= n;   // sign flipped by the map: count * q_I - sum
    const u32 p0 = f0 ? j0 - (u32)n : j0, p1 = f1 ? j1 - (u32)n : j1;
    const u64 s0 = src[p0], s1 = src[p1], w0 = car[p0], w1 = car[p1];
    for (int J = 0; J < a.K; J++) {
        const ModDev mj = mod_at(a.mods, J);
        const u64 t64 = reduce64(mj.nq, mj);                                 // 2^64 mod q_J
        const u64 cq = mulmod((u64)a.count, reduce64(qI, mj), mj);           // count * q_I mod q_J
        u64 d0 = addmod(reduce64(s0, mj), mulmod(w0, t64, mj), mj.q), d1 = addmod(reduce64(s1, mj), mulmod(w1, t64, mj), mj.q);
        if (f0) d0 = submod(cq, d0, mj.q);
        if (f1) d1 = submod(cq, d1, mj.q);
        st2(a.out + ((b * a.L + I) * a.K + J) * n + i, U2{d0, d1});
    }
}

// key-switch mod-down by the special prime with rounding: gid over [B][2][L][N]
HD void ks_finish_body(const KsFinishArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    size_t r = gid >> a.logn;
    const int j = (int)(r % a.L); r /= a.L;
    const int k = (int)(r & 1);
    const size_t b = r >> 1;
    if (b >= (size_t)a.B) return;
    const ModDev m = mod_at(a.mods, j);
    const u64 qsp = mod_at(a.mods, a.K - 1).q;
    const u64 sp = a.S[((b * 2 + k) * a.K + (a.K - 1)) * n + i];
    const u64 rk = addmod(sp, a.half, qsp);
    const u64 rj = reduce64(rk, m);
    u64 v = a.S[((b * 2 + k) * a.K + j) * n + i];
    v = addmod(submod(v, rj, m.q), a.half_mod[j], m.q);
    v = shoup_mul(v, a.qsp_inv[j], a.qsp_inv_s[j], m.q);
    if ((a.base_mask >> k) & 1) v = addmod(v, a.base[b * a.base_item_stride + ((size_t)k * a.L + j) * n + i], m.q);
    a.out[gid] = v;
}

// add_plain / sub_plain with the BFV scaling variant (seal/util/scalingvariant.h:23)
// gid over [B][N]; also carries c1 through (optionally negated).
HD void add_plain_body(const AddPlainArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    const size_t b = gid >> a.logn;
    if (b >= (size_t)a.B) return;
    const u64 mval = a.plain_ptrs ? a.plain_ptrs[b][a.plain_shift + i] : a.plain[(a.plain_bcast ? 0 : b * n) + i];
    // fix = floor((m * (Q mod t) + (t+1)/2) / t)
    u64 lo = mval * a.q_mod_t, hi = mulhi64(mval, a.q_mod_t);
    lo += a.thr; hi += (lo < a.thr);
    u64 fix = barrett_quo(lo, hi, a.t_r_lo, a.t_r_hi);
    u64 rem = lo - fix * a.t;
    while (rem >= a.t) { rem -= a.t; fix++; }
    for (int j = 0; j < a.L; j++) {
        const ModDev m = mod_at(a.mods, j);
        Acc128 s = {0, 0};
        acc_mac(s, mval, a.delta[j]);
        acc_add(s, fix);
        const u64 sc = barrett128(s.lo, s.hi, m);
        const size_t o0 = ((b * 2 + 0) * a.L + j) * n + i, o1 = ((b * 2 + 1) * a.L + j) * n + i;
        u64 c0 = a.ct[o0], c1 = a.ct[o1];
        if (a.negate_ct) { c0 = negmod(c0, m.q); c1 = negmod(c1, m.q); }
        a.out[o0] = a.subtract ? submod(c0, sc, m.q) : addmod(c0, sc, m.q);
        a.out[o1] = c1;
    }
}

// BatchEncoder::encode slot scatter: gid over [B][count * (second_off>=0 ? 2 : 1)]
HD void encode_scatter_body(const EncodeArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const int per = a.count * (a.second_off >= 0 ? 2 : 1);
    const size_t b = gid / per;
    if (b >= (size_t)a.B) return;
    const int v = (int)(gid % per);
    const int slot = v < a.count ? v : a.second_off + (v - a.count);
    u64 x = a.vals[b * a.stride + v];
    x = x >= a.t ? x % a.t : x;
    a.out[b * n + a.slot_map[slot]] = x;
}

// diagonals of the two PASTA matrices placed in slot order (pasta_3_seal.cpp:390-401):
// diag_i[j] = M1[j][(j - i) mod 128], diag_i[j + N/2] = M2[j][(j - i) mod 128].
// gid over [4 layers][128 diag][2 halves][128 j]
HD void diag_body(const DiagArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const int j = (int)(gid & 127);
    const int h = (int)((gid >> 7) & 1);
    const int i = (int)((gid >> 8) & 127);
    const int layer = (int)(gid >> 15);
    if (layer >= PASTA_R + 1) return;
    const u64 v = a.mats[(((size_t)layer * 2 + h) * PASTA_T + j) * PASTA_T + ((j + PASTA_T - i) & 127)];
    const size_t slot = (size_t)j + (h ? n / 2 : 0);
    a.out[((size_t)layer * PASTA_T + i) * n + a.slot_map[slot]] = v;
}

// babystep-giantstep diagonals (pasta_3_seal.cpp:280-328): diagonal i is rotated left by k*N1 (k = i / N1) and, when
// the row is longer than 128 slots, its last k*N1 entries move to the end of the half row so that the later
// rotate_rows(-k*N1) of the inner sum brings them back.  gid over [4 layers][128 diag][2 halves][128 j']
HD void bsgs_diag_body(const BsgsDiagArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn, half = n >> 1;
    const int jp = (int)(gid & 127);
    const int h = (int)((gid >> 7) & 1);
    const int i = (int)((gid >> 8) & 127);
    const int layer = (int)(gid >> 15);
    if (layer >= PASTA_R + 1) return;
    const int shift = (i / a.n1) * a.n1;
    const int j = (jp + shift) & 127;
    const u64 v = a.mats[(((size_t)layer * 2 + h) * PASTA_T + j) * PASTA_T + ((j + PASTA_T - i) & 127)];
    size_t pos = (size_t)jp;
    if (half != PASTA_T && jp >= PASTA_T - shift) pos = half - PASTA_T + jp;
    a.out[((size_t)layer * PASTA_T + i) * n + a.slot_map[pos + (h ? half : 0)]] = v;
}

// ------------------------------------------------------------------ BEHZ
// fastbconv_m_tilde + sm_mrq (seal/util/rns.h:213-219): gid over [P][N]
HD void behz_extend_body(const BehzExtendArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    const size_t p = gid >> a.logn;
    if (p >= (size_t)a.P) return;
    const BehzDev &z = *a.bz;
    const int L = a.L;
    const u64 MTm = 0xffffffffULL, MT = 1ULL << 32;
    u64 tmp[HHE_MAXL];
    u64 ymt = 0;
    for (int l = 0; l < L; l++) {
        const ModDev m = mod_at(a.mods, l);
        u64 v = mulmod(a.x[(p * L + l) * n + i], z.mt_mod_q[l], m);
        v = shoup_mul(v, z.inv_punct_q[l], z.inv_punct_q_s[l], m.q);
        tmp[l] = v;
        ymt += v * z.punct_q_mt[l];
    }
    ymt &= MTm;
    const u64 r = (ymt * z.neg_inv_q_mt) & MTm;
    for (int pb = 0; pb <= L; pb++) {
        const ModDev mp = mod_at(a.mods, a.K + pb);
        Acc128 s = {0, 0};
        for (int l = 0; l < L; l++) {
            acc_mac(s, tmp[l], z.punct_q_bsk[l][pb]);
            if ((l & 3) == 3) { s.lo = barrett128(s.lo, s.hi, mp); s.hi = 0; }
        }
        const u64 rr = r >= (MT >> 1) ? r + (mp.q - MT) : r;
        acc_mac(s, z.q_mod_bsk[pb], rr);
        const u64 v = barrett128(s.lo, s.hi, mp);
        a.xb[(p * (L + 1) + pb) * n + i] = mulmod(v, z.inv_mt_bsk[pb], mp);
    }
}

// ciphertext tensor product in the NTT domain: gid over [B][limbs][N]
HD void tensor_body(const TensorArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    size_t r = gid >> a.logn;
    const int j = (int)(r % a.limbs);
    const size_t b = r / a.limbs;
    if (b >= (size_t)a.B) return;
    const ModDev m = mod_at(a.mods, a.mod_base + j);
    const size_t lim = a.limbs;
    const u64 a0 = a.a[((b * 2 + 0) * lim + j) * n + i], a1 = a.a[((b * 2 + 1) * lim + j) * n + i];
    const u64 b0 = a.b[((b * 2 + 0) * lim + j) * n + i], b1 = a.b[((b * 2 + 1) * lim + j) * n + i];
    a.d[((b * 3 + 0) * lim + j) * n + i] = mulmod(a0, b0, m);
    Acc128 s = {0, 0};
    acc_mac(s, a0, b1);
    acc_mac(s, a1, b0);
    a.d[((b * 3 + 1) * lim + j) * n + i] = barrett128(s.lo, s.hi, m);
    a.d[((b * 3 + 2) * lim + j) * n + i] = mulmod(a1, b1, m);
}

// fast_floor + fastbconv_sk (seal/util/rns.h:221-228): gid over [P][N]
HD void behz_floor_body(const BehzFloorArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    const size_t i = gid & (n - 1);
    const size_t p = gid >> a.logn;
    if (p >= (size_t)a.P) return;
    const BehzDev &z = *a.bz;
    const int L = a.L;
    u64 tq[HHE_MAXL], f[HHE_MAXL + 1], tb[HHE_MAXL];
    for (int l = 0; l < L; l++)
        tq[l] = shoup_mul(a.dq[(p * L + l) * n + i], z.inv_punct_q[l], z.inv_punct_q_s[l], mod_at(a.mods, l).q);
    for (int pb = 0; pb <= L; pb++) {
        const ModDev mp = mod_at(a.mods, a.K + pb);
        Acc128 s = {0, 0};
        for (int l = 0; l < L; l++) {
            acc_mac(s, tq[l], z.punct_q_bsk[l][pb]);
            if ((l & 3) == 3) { s.lo = barrett128(s.lo, s.hi, mp); s.hi = 0; }
        }
        const u64 conv = barrett128(s.lo, s.hi, mp);
        const u64 x = a.db[(p * (L + 1) + pb) * n + i];
        f[pb] = mulmod(addmod(x, mp.q - conv, mp.q), z.inv_q_bsk[pb], mp);
    }
    for (int l = 0; l < L; l++) tb[l] = mulmod(f[l], z.inv_punct_B[l], mod_at(a.mods, a.K + l));
    const ModDev ms = mod_at(a.mods, a.K + L);
    Acc128 s = {0, 0};
    for (int l = 0; l < L; l++) {
        acc_mac(s, tb[l], z.punct_B_msk[l]);
        if ((l & 3) == 3) { s.lo = barrett128(s.lo, s.hi, ms); s.hi = 0; }
    }
    const u64 conv = barrett128(s.lo, s.hi, ms);
    const u64 alpha = mulmod(addmod(conv, ms.q - f[L], ms.q), z.inv_B_msk, ms);
    const bool negative = alpha > (z.msk >> 1);
    for (int j = 0; j < L; j++) {
        const ModDev m = mod_at(a.mods, j);
        Acc128 t = {0, 0};
        for (int l = 0; l < L; l++) {
            acc_mac(t, tb[l], z.punct_B_q[l][j]);
            if ((l & 3) == 3) { t.lo = barrett128(t.lo, t.hi, m); t.hi = 0; }
        }
        u64 v = barrett128(t.lo, t.hi, m);
        Acc128 c = {v, 0};
        if (negative) acc_mac(c, z.msk - alpha, z.B_mod_q[j]);
        else acc_mac(c, alpha, z.neg_B_mod_q[j]);
        a.out[(p * L + j) * n + i] = barrett128(c.lo, c.hi, m);
    }
}
