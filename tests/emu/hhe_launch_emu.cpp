// hhe_launch_emu.cpp -- TESTS ONLY.  Implements csrc/hhe_launch.h on the CPU by looping the
// kernel bodies of csrc/hhe_kernel_bodies.h over (block, thread) with a barrier between
// phases, so the `-m "not gpu"` suite can check the kernels' index arithmetic and the host
// schedule against the oracle without a GPU.  It is never built into or loaded by the
// product library (libhhe_gfx950.so), which has no CPU path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "hhe_kernel_bodies.h"
#include "hhe_launch.h"

// -DHHE_RANGE_CHECK (this build only): a lazy value left its 64-bit range -- a bug in the range analysis, never a data error
void hhe_range_violation(const char *what)
{
    fprintf(stderr, "emu: range violation: %s\n", what);
    abort();
}

const char *rt_backend_name() { return "cpu-emulator(tests-only)"; }
const char *rt_last_error() { return "emu"; }
int rt_set_device(int) { return 0; }
void *rt_malloc(size_t b) { return malloc(b ? b : 8); }
void rt_free(void *p) { free(p); }
int rt_h2d(void *d, const void *s, size_t n, rt_stream) { memcpy(d, s, n); return 0; }
int rt_d2h(void *d, const void *s, size_t n, rt_stream) { memcpy(d, s, n); return 0; }
int rt_d2d(void *d, const void *s, size_t n, rt_stream) { memmove(d, s, n); return 0; }
int rt_memset(void *d, int v, size_t n, rt_stream) { memset(d, v, n); return 0; }
int rt_sync(rt_stream) { return 0; }
rt_stream rt_stream_create() { static int dummy[8]; static int n = 0; return (rt_stream)&dummy[(n++) & 7]; }
void rt_stream_destroy(rt_stream) {}
void *rt_event_create() { static int ev; return &ev; }
void rt_event_destroy(void *) {}
void *rt_event_create_timed() { static int ev; return &ev; }
float rt_event_elapsed_ms(void *, void *) { return 0.f; }
int rt_event_record(void *, rt_stream) { return 0; }
int rt_event_sync(void *) { return 0; }
int rt_stream_wait_event(rt_stream, void *) { return 0; }

template <int LOGM, bool STRIDED, bool INVERSE, int CC, int T, int SCH, int I, int S0, bool LAZY8 = false, bool TWL = false>
static void rounds_fwd(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl)
{
    if constexpr (I < NttSched<LOGM, SCH>::R) {
        constexpr int RHO = NttSched<LOGM, SCH>::rho(I);
        for (int t = 0; t < T; t++) ntt_body_round<LOGM, S0, RHO, STRIDED, false, LAZY8, CC, T, SCH == 512, TWL>(a, bx, by, t, lds, twl);
        rounds_fwd<LOGM, STRIDED, INVERSE, CC, T, SCH, I + 1, S0 + RHO, LAZY8, TWL>(a, bx, by, lds, twl);
    }
}
template <int LOGM, bool STRIDED, bool INVERSE, int CC, int T, int SCH, int I, int SEND, bool LAZY8 = false, bool TWL = false>
static void rounds_inv(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl)
{
    if constexpr (I >= 0) {
        constexpr int RHO = NttSched<LOGM, SCH>::rho(I);
        for (int t = 0; t < T; t++) ntt_body_round<LOGM, SEND - RHO, RHO, STRIDED, true, LAZY8, CC, T, false, TWL>(a, bx, by, t, lds, twl);
        rounds_inv<LOGM, STRIDED, INVERSE, CC, T, SCH, I - 1, SEND - RHO, LAZY8, TWL>(a, bx, by, lds, twl);
    }
}
// the register rounds of one pass over a staged tile (a barrier after each round = the end of the thread loop)
template <int LOGM, bool STRIDED, bool INVERSE, int CC, int T = NTT_THREADS, int SCH = T, bool TWL = false>
static void tile_rounds_emu(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl = nullptr)
{
    if constexpr (!INVERSE) {
        if (a.lazy8) rounds_fwd<LOGM, STRIDED, INVERSE, CC, T, SCH, 0, 0, true, TWL>(a, bx, by, lds, twl);
        else rounds_fwd<LOGM, STRIDED, INVERSE, CC, T, SCH, 0, 0, false, TWL>(a, bx, by, lds, twl);
    }
    else if (a.lazy8) rounds_inv<LOGM, STRIDED, INVERSE, CC, T, SCH, NttSched<LOGM, SCH>::R - 1, LOGM, true, TWL>(a, bx, by, lds, twl);
    else rounds_inv<LOGM, STRIDED, INVERSE, CC, T, SCH, NttSched<LOGM, SCH>::R - 1, LOGM, false, TWL>(a, bx, by, lds, twl);
}
template <int LOGM, bool STRIDED, bool INVERSE, bool FULL>
static void pass_emu(const NttArgs &a, int gx, int gy)
{
    constexpr int CM = FULL ? LOGM : -1, CC = FULL ? NttTile::LOG - LOGM : -1;
#pragma omp parallel
    {
        std::vector<u64> lds(NttLds::ELEMS);
#pragma omp for collapse(2)
        for (int by = 0; by < gy; by++)
            for (int bx = 0; bx < gx; bx++) {
                for (int t = 0; t < NTT_THREADS; t++) ntt_body_load<STRIDED, INVERSE, CM, CC>(a, bx, by, t, lds.data());
                tile_rounds_emu<LOGM, STRIDED, INVERSE, CC>(a, bx, by, lds.data());
                for (int t = 0; t < NTT_THREADS; t++) ntt_body_store<STRIDED, INVERSE, CM, CC>(a, bx, by, t, lds.data());
            }
    }
}
template <bool STRIDED, bool INVERSE>
static void launch_pass(NttArgs a, int logm, int other)
{
    a.logm = logm;
    int logc = NttTile::LOG - logm;
    if (logc > other) logc = other;
    a.logc = logc;
    const int gx = 1 << (other - logc), gy = a.count;
    const bool full = logc == NttTile::LOG - logm;
#define PASS_EMU(M_) case M_: if (full) pass_emu<M_, STRIDED, INVERSE, true>(a, gx, gy); else pass_emu<M_, STRIDED, INVERSE, false>(a, gx, gy); break;
    switch (logm) {
    PASS_EMU(5) PASS_EMU(6) PASS_EMU(7) PASS_EMU(8)
    default: fprintf(stderr, "emu: unsupported pass size\n"); abort();
    }
}
void k_ntt(const NttArgs &a, bool inverse, rt_stream)
{
    if (a.count <= 0) return;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    if (!inverse) { launch_pass<true, false>(a, n1, n2); launch_pass<false, false>(a, n2, n1); }
    else { launch_pass<false, true>(a, n2, n1); launch_pass<true, true>(a, n1, n2); }
}
void k_ntt_pass(const NttArgs &a, bool inverse, bool second, rt_stream)
{
    if (a.count <= 0) return;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    if (!inverse) { if (!second) launch_pass<true, false>(a, n1, n2); else launch_pass<false, false>(a, n2, n1); }
    else { if (!second) launch_pass<false, true>(a, n2, n1); else launch_pass<true, true>(a, n1, n2); }
}
void k_ntt2_fwd_first(const NttArgs &a1, const NttArgs &a2, rt_stream s) { k_ntt_pass(a1, false, false, s); k_ntt_pass(a2, false, false, s); }
template <int LOGM>
static void ks_row_emu(const NttArgs &a, const KsRowArgs &x, const NttArgs &c0, int gx, int gy)
{
    constexpr int CC = KSROW_TILE_LOG - LOGM, T = KSROW_THREADS, SCH = KSROW_SCHED;
    const size_t n = (size_t)1 << a.logn;
#pragma omp parallel
    {
        constexpr bool TWL = LOGM == 8;
        std::vector<u64> lds(KSROW_LDS + (TWL ? KSROW_TWL : 0));
        u64 *const twl = TWL ? lds.data() + KSROW_LDS : nullptr;
        std::vector<u64> acc0((size_t)T * 2 * KSROW_NP), acc1((size_t)T * 2 * KSROW_NP);
        std::vector<U2> pf((size_t)T * KSROW_NP);
        auto PF = [&](int t) { return &pf[(size_t)t * KSROW_NP]; };
        auto A0 = [&](int t) { return &acc0[(size_t)t * 2 * KSROW_NP]; };
        auto A1 = [&](int t) { return &acc1[(size_t)t * 2 * KSROW_NP]; };
#pragma omp for collapse(2)
        for (int by = 0; by < c0.count; by++)  // the c0-branch tiles of the grid: a plain forward row pass at this kernel's geometry
            for (int bx = 0; bx < gx; bx++) {
                if (TWL) for (int t = 0; t < T; t++) ks_row_twiddle_fill<LOGM, CC>(c0, bx, by, false, t, twl);
                for (int t = 0; t < T; t++) ntt_body_load<false, false, LOGM, CC, T>(c0, bx, by, t, lds.data());
                tile_rounds_emu<LOGM, false, false, CC, T, SCH, TWL>(c0, bx, by, lds.data(), twl);
                for (int t = 0; t < T; t++) ntt_body_store<false, false, LOGM, CC, T>(c0, bx, by, t, lds.data());
            }
#pragma omp for collapse(2)
        for (int y = 0; y < gy; y++)
            for (int bx = 0; bx < gx; bx++) {
                const int b = y / x.K, J = y % x.K;
                {
                    if (TWL) for (int t = 0; t < T; t++) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, false, t, twl);
                    std::fill(acc0.begin(), acc0.end(), 0);
                    std::fill(acc1.begin(), acc1.end(), 0);
                    for (int t = 0; t < T; t++) ks_row_tile_fetch<LOGM, CC>(a, bx, (b * x.L + 0) * x.K + J, t, PF(t));
                    for (int I = 0; I < x.L; I++) {
                        const int by = (b * x.L + I) * x.K + J;
                        for (int t = 0; t < T; t++) ks_row_tile_commit<LOGM, CC>(a, bx, by, t, PF(t), lds.data());
                        if (I + 1 < x.L) for (int t = 0; t < T; t++) ks_row_tile_fetch<LOGM, CC>(a, bx, by + x.K, t, PF(t));
                        tile_rounds_emu<LOGM, false, false, CC, T, SCH, TWL>(a, bx, by, lds.data(), twl);
                        if (TWL && I == x.L - 1) for (int t = 0; t < T; t++) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, true, t, twl);
                        for (int t = 0; t < T; t++) ks_row_mac_phase<LOGM, CC>(x, a, bx, b, J, I, t, lds.data(), A0(t), A1(t));
                    }
                    auto inverse_to = [&](std::vector<u64> &acc, u64 *out) {
                        for (int t = 0; t < T; t++) ks_row_flush_phase<LOGM, CC>(a, bx, J, t, lds.data(), &acc[(size_t)t * 2 * KSROW_NP], nullptr);
                        tile_rounds_emu<LOGM, false, true, CC, T, SCH, TWL>(a, bx, J, lds.data(), twl);
                        for (int t = 0; t < T; t++) ks_row_store_phase<LOGM, CC>(a, bx, J, t, lds.data(), out);
                    };
                    if (J < x.L) {
                        if (x.U0) inverse_to(acc0, x.U0 + (size_t)b * x.u_stride + (size_t)J * n);
                        else for (int t = 0; t < T; t++) ks_row_flush_phase<LOGM, CC>(a, bx, J, t, lds.data(), A0(t), x.S + (((size_t)b * 2 + 0) * x.K + J) * n);
                        inverse_to(acc1, x.U1 + (size_t)b * x.u_stride + (size_t)J * n);
                    } else {
                        inverse_to(acc0, x.Usp + ((size_t)b * 2 + 0) * n);
                        inverse_to(acc1, x.Usp + ((size_t)b * 2 + 1) * n);
                    }
                }
            }
    }
}
int k_ks_row(const NttArgs &a0, const KsRowArgs &x, const NttArgs *c0_row, rt_stream)
{
    // the device kernel only has the pseudo-Mersenne (lazy) rounds: a launch whose moduli do not qualify is a host bug
    if (!a0.lazy8 || (c0_row && !c0_row->lazy8)) { fprintf(stderr, "emu: k_ks_row on a modulus without the pseudo-Mersenne form\n"); abort(); }
    NttArgs a = a0, c0;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    a.logm = n2;
    a.logc = KSROW_TILE_LOG - n2;
    if (c0_row) { c0 = *c0_row; c0.logm = a.logm; c0.logc = a.logc; }
    else memset(&c0, 0, sizeof(c0));
    const int gx = 1 << (n1 - a.logc), gy = x.B * x.K;
    switch (n2) {
    case 6: ks_row_emu<6>(a, x, c0, gx, gy); break;
    case 7: ks_row_emu<7>(a, x, c0, gx, gy); break;
    case 8: ks_row_emu<8>(a, x, c0, gx, gy); break;
    default: return -1;
    }
    return 0;
}
template <int LOGM>
static void ks_perm_row_emu(const NttArgs &a, const KsRowArgs &x, int gx, int gy)
{
    constexpr int CC = KSROW_TILE_LOG - LOGM, T = KSROW_THREADS, SCH = KSROW_SCHED;
    const size_t n = (size_t)1 << a.logn;
#pragma omp parallel
    {
        constexpr bool TWL = LOGM == 8;
        std::vector<u64> lds(KSROW_LDS + (TWL ? KSROW_TWL : 0));
        u64 *const twl = TWL ? lds.data() + KSROW_LDS : nullptr;
        std::vector<u64> acc0((size_t)T * 2 * KSROW_NP), acc1((size_t)T * 2 * KSROW_NP);
#pragma omp for collapse(2)
        for (int y = 0; y < gy; y++)
            for (int bx = 0; bx < gx; bx++) {
                const int b = y / x.K, J = y % x.K;
                if (TWL) for (int t = 0; t < T; t++) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, true, t, twl);
                std::fill(acc0.begin(), acc0.end(), 0);
                std::fill(acc1.begin(), acc1.end(), 0);
                for (int I = 0; I < x.L; I++)
                    for (int t = 0; t < T; t++) ks_row_mac_gather<LOGM, CC>(x, a, bx, b, J, I, t, &acc0[(size_t)t * 2 * KSROW_NP], &acc1[(size_t)t * 2 * KSROW_NP]);
                auto inverse_to = [&](std::vector<u64> &acc, u64 *out) {
                    for (int t = 0; t < T; t++) ks_row_flush_phase<LOGM, CC>(a, bx, J, t, lds.data(), &acc[(size_t)t * 2 * KSROW_NP], nullptr);
                    tile_rounds_emu<LOGM, false, true, CC, T, SCH, TWL>(a, bx, J, lds.data(), twl);
                    for (int t = 0; t < T; t++) ks_row_store_phase<LOGM, CC>(a, bx, J, t, lds.data(), out);
                };
                inverse_to(acc0, J < x.L ? x.U0 + (size_t)b * x.u_stride + (size_t)J * n : x.Usp + ((size_t)b * 2 + 0) * n);
                inverse_to(acc1, J < x.L ? x.U1 + (size_t)b * x.u_stride + (size_t)J * n : x.Usp + ((size_t)b * 2 + 1) * n);
            }
    }
}
int k_ks_perm_row(const NttArgs &a0, const KsRowArgs &x, rt_stream)
{
    if (!a0.lazy8) { fprintf(stderr, "emu: k_ks_perm_row on a modulus without the pseudo-Mersenne form\n"); abort(); }
    NttArgs a = a0;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    a.logm = n2;
    a.logc = KSROW_TILE_LOG - n2;
    const int gx = 1 << (n1 - a.logc), gy = x.B * x.K;
    switch (n2) {
    case 6: ks_perm_row_emu<6>(a, x, gx, gy); break;
    case 7: ks_perm_row_emu<7>(a, x, gx, gy); break;
    case 8: ks_perm_row_emu<8>(a, x, gx, gy); break;
    default: return -1;
    }
    return 0;
}
void k_ntt2_fwd(const NttArgs &a1, const NttArgs &a2, rt_stream s) { k_ntt(a1, false, s); k_ntt(a2, false, s); }
void k_ntt2_inv(const NttArgs &a1, const NttArgs &a2, rt_stream s) { k_ntt(a1, true, s); k_ntt(a2, true, s); }
#define LOOP(total, call)                                         \
    do {                                                          \
        const long long _t = (long long)(total);                  \
        _Pragma("omp parallel for") for (long long g = 0; g < _t; g++) { call; } \
    } while (0)
void k_elt(const EltArgs &a, int op, rt_stream) { LOOP((size_t)a.count << a.logn, elt_body(a, op, (size_t)g)); }
void k_copy_items(const CopyItemsArgs &a, rt_stream) { LOOP(a.count * (a.words >> 1), copy_items_body(a, (size_t)g)); }
void k_galois(const GaloisArgs &a, rt_stream) { LOOP((size_t)a.count << (a.logn - 1), galois_body(a, (size_t)g)); }
void k_perm(const PermArgs &a, rt_stream) { LOOP((size_t)a.count << a.logn, perm_body(a, (size_t)g)); }
template <int MODE> static void ks_mac_t_emu(const KsMacArgs &a)
{
    const size_t total = ((size_t)a.B * a.K) << (a.logn - 1);
    switch (a.L) {
    case 1: LOOP(total, (ks_mac_body_t<1, MODE>(a, (size_t)g))); break;
    case 2: LOOP(total, (ks_mac_body_t<2, MODE>(a, (size_t)g))); break;
    case 3: LOOP(total, (ks_mac_body_t<3, MODE>(a, (size_t)g))); break;
    default: LOOP(total, (ks_mac_body_t<4, MODE>(a, (size_t)g))); break;
    }
}
void k_ks_mac(const KsMacArgs &a, rt_stream)
{
    switch (ks_mac_mode(a)) {
    case KS_PLAIN: ks_mac_t_emu<KS_PLAIN>(a); break;
    case KS_ACC: ks_mac_t_emu<KS_ACC>(a); break;
    case KS_PERM: ks_mac_t_emu<KS_PERM>(a); break;
    case KS_LEAF: ks_mac_t_emu<KS_LEAF>(a); break;
    default: LOOP(((size_t)a.B * a.K) << (a.logn - 1), ks_mac_body(a, (size_t)g)); break;
    }
}
int k_ks_mac_leaves(const KsMacLeavesArgs &a, rt_stream)
{
    const size_t total = ((size_t)a.B * (a.sp_only ? 1 : a.K)) << (a.logn - 1);
    switch (a.L) {
    case 1: LOOP(total, (ks_mac_leaves_body<1>(a, (size_t)g))); break;
    case 2: LOOP(total, (ks_mac_leaves_body<2>(a, (size_t)g))); break;
    case 3: LOOP(total, (ks_mac_leaves_body<3>(a, (size_t)g))); break;
    case 4: LOOP(total, (ks_mac_leaves_body<4>(a, (size_t)g))); break;
    default: return -1;
    }
    return 0;
}
void k_ks_corr(const KsCorrArgs &a, rt_stream) { LOOP(((size_t)2 * a.K) << a.logn, ks_corr_body(a, (size_t)g)); }
void k_ks_finish(const KsFinishArgs &a, rt_stream) { LOOP(((size_t)a.B * 2 * a.L) << a.logn, ks_finish_body(a, (size_t)g)); }
void k_leaf_sum(const LeafSumArgs &a, rt_stream) { LOOP(((size_t)a.B * 2 * a.L) << a.logn, leaf_sum_body(a, (size_t)g)); }
void k_csum_add(const CsumArgs &a, rt_stream) { LOOP(((size_t)a.B * a.L) << (a.logn - 1), csum_add_body(a, (size_t)g)); }
void k_csum_c0(const CsumArgs &a, rt_stream) { LOOP(((size_t)a.B * a.L) << (a.logn - 1), csum_c0_body(a, (size_t)g)); }
void k_csum_digits(const CsumArgs &a, rt_stream) { LOOP(((size_t)a.B * a.L) << (a.logn - 1), csum_digits_body(a, (size_t)g)); }
void k_leaf_round(const LeafRoundArgs &a, rt_stream) { LOOP(((size_t)a.B * 2) << (a.logn - 1), leaf_round_body(a, (size_t)g)); }
void k_add_plain(const AddPlainArgs &a, rt_stream) { LOOP((size_t)a.B << a.logn, add_plain_body(a, (size_t)g)); }
void k_encode_scatter(const EncodeArgs &a, rt_stream) { LOOP((size_t)a.B * a.count * (a.second_off >= 0 ? 2 : 1), encode_scatter_body(a, (size_t)g)); }
void k_diag(const DiagArgs &a, rt_stream) { LOOP((size_t)(PASTA_R + 1) * PASTA_T * 2 * PASTA_T, diag_body(a, (size_t)g)); }
void k_bsgs_diag(const BsgsDiagArgs &a, rt_stream) { LOOP((size_t)(PASTA_R + 1) * PASTA_T * 2 * PASTA_T, bsgs_diag_body(a, (size_t)g)); }
void k_behz_extend(const BehzExtendArgs &a, rt_stream) { LOOP((size_t)a.P << a.logn, behz_extend_body(a, (size_t)g)); }
void k_tensor(const TensorArgs &a, rt_stream) { LOOP(((size_t)a.B * a.limbs) << a.logn, tensor_body(a, (size_t)g)); }
void k_behz_floor(const BehzFloorArgs &a, rt_stream) { LOOP((size_t)a.P << a.logn, behz_floor_body(a, (size_t)g)); }

void k_pasta_xof(const PastaXofArgs &a, rt_stream) { LOOP(a.nblocks, pasta_xof_body(a, (size_t)g)); }
void k_pasta_plain(const PastaPlainArgs &a, rt_stream)
{
#pragma omp parallel for
    for (int blk = 0; blk < a.nblocks; blk++) {
        std::vector<u64> lds(PASTA_PLAIN_LDS, 0);
        pasta_plain_schedule([&](int kind, int layer, int i) {
            for (int t = 0; t < PASTA_PLAIN_THREADS; t++) pasta_plain_phase(a, blk, t, layer, kind, i, lds.data());
        });
    }
}
void k_pasta_crypt(const PastaCryptArgs &a, rt_stream) { LOOP(a.S * a.nwords, pasta_crypt_body(a, (size_t)g)); }
void k_decrypt_round(const DecryptArgs &a, rt_stream) { LOOP(a.B << a.logn, decrypt_round_body(a, (size_t)g)); }
void k_decode_gather(const DecodeArgs &a, rt_stream) { LOOP(a.B << a.logn, decode_gather_body(a, (size_t)g)); }
