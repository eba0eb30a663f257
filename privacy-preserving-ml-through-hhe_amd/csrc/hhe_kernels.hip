// hhe_kernels.hip -- gfx950 (MI355X, wave64) kernels for the PASTA-3 -> BFV transciphering
// path: RNS negacyclic NTT/INTT, coefficient-wise modular arithmetic, Galois permutation,
// key-switch gadget product / mod-down, BEHZ base conversions.  Integer lanes only
// (v_mul_hi_u32 / v_mad_u64_u32), no MFMA: this is exact modular arithmetic.
// Bodies live in hhe_kernel_bodies.h; this file is the __global__ wrappers + launchers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "hhe_kernel_bodies.h"
#include "hhe_launch.h"

static thread_local char g_rt_err[256] = "";
static int rt_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    snprintf(g_rt_err, sizeof(g_rt_err), "%s: %s", what, hipGetErrorString(e));
    return -1;
}
const char *rt_backend_name() { return "hip-gfx950"; }
const char *rt_last_error() { return g_rt_err; }
int rt_set_device(int d) { return rt_check(hipSetDevice(d), "hipSetDevice"); }
void *rt_malloc(size_t bytes)
{
    void *p = nullptr;
    if (rt_check(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc")) return nullptr;
    return p;
}
void rt_free(void *p) { if (p) (void)hipFree(p); }
int rt_h2d(void *d, const void *s, size_t n, rt_stream st) { return rt_check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st), "h2d"); }
int rt_d2h(void *d, const void *s, size_t n, rt_stream st) { return rt_check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st), "d2h"); }
int rt_d2d(void *d, const void *s, size_t n, rt_stream st) { return rt_check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st), "d2d"); }
int rt_memset(void *d, int v, size_t n, rt_stream st) { return rt_check(hipMemsetAsync(d, v, n, (hipStream_t)st), "memset"); }
int rt_sync(rt_stream st)
{
    if (rt_check(hipStreamSynchronize((hipStream_t)st), "sync")) return -1;
    return rt_check(hipGetLastError(), "kernel launch");  // a rejected launch (bad grid, missing code object) must not pass silently
}
rt_stream rt_stream_create()
{
    hipStream_t s = nullptr;
    if (rt_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    return (rt_stream)s;
}
void rt_stream_destroy(rt_stream s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
void *rt_event_create()
{
    hipEvent_t e = nullptr;
    if (rt_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return nullptr;
    return (void *)e;
}
void rt_event_destroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
void *rt_event_create_timed()
{
    hipEvent_t e = nullptr;
    if (rt_check(hipEventCreate(&e), "hipEventCreate")) return nullptr;
    return (void *)e;
}
float rt_event_elapsed_ms(void *ev0, void *ev1)
{
    float ms = -1.f;
    if (rt_check(hipEventElapsedTime(&ms, (hipEvent_t)ev0, (hipEvent_t)ev1), "hipEventElapsedTime")) return -1.f;
    return ms;
}
int rt_event_record(void *ev, rt_stream s) { return rt_check(hipEventRecord((hipEvent_t)ev, (hipStream_t)s), "hipEventRecord"); }
int rt_event_sync(void *ev) { return rt_check(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize"); }
int rt_stream_wait_event(rt_stream s, void *ev) { return rt_check(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0), "hipStreamWaitEvent"); }
// ---------------------------------------------------------------- NTT
// one-dimensional grids (gridDim.y is limited to 65535 polynomials): block -> (tile, poly), tiles per poly = 2^tiles_log
#define NTT_BX(a) ((int)(blockIdx.x & ((1u << (a).tiles_log) - 1)))
#define NTT_BY(a) ((int)(blockIdx.x >> (a).tiles_log))

// Synchronisation between the phases of a tile.  A workgroup of ONE wave (the fused row kernel) needs no barrier: its LDS operations
// execute in program order, so a wave-scope fence (ordering for the compiler, no instruction) is enough.  __syncthreads() would also
// be correct but its workgroup-scope release is an s_waitcnt vmcnt(0): every phase boundary would wait for ALL of the wave's
// outstanding global loads and stores (the tile and key words requested ahead, and stores whose completion takes thousands of
// cycles under load) -- measured: the S_0 stores alone held the next phase for ~7,000 cycles per workgroup.
template <int T> static __device__ __forceinline__ void tile_sync()
{
    if constexpr (T <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}
// CC = tile columns as a compile-time constant (full tiles) or -1 (ragged tiles of small N: taken from the arguments);
// T = lanes per workgroup (256: radix-16 rounds, 512: radix-8 rounds)
template <int LOGM, bool STRIDED, bool INVERSE, int CC, int T, int SCH = T, bool TWL = false>
struct NttRounds {
    static constexpr int R = NttSched<LOGM, SCH>::R;
    // forward: rounds 0..R-1 ascending; inverse: descending
    template <int I, int S0, bool LAZY8 = false>
    static __device__ __forceinline__ void fwd(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl)
    {
        if constexpr (I < R) {
            constexpr int RHO = NttSched<LOGM, SCH>::rho(I);
            ntt_body_round<LOGM, S0, RHO, STRIDED, false, LAZY8, CC, T, SCH == 512, TWL>(a, bx, by, threadIdx.x, lds, twl);
            tile_sync<T>();
            fwd<I + 1, S0 + RHO, LAZY8>(a, bx, by, lds, twl);
        }
    }
    template <int I, int SEND, bool LAZY8 = false>
    static __device__ __forceinline__ void inv(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl)
    {
        if constexpr (I >= 0) {
            constexpr int RHO = NttSched<LOGM, SCH>::rho(I);
            ntt_body_round<LOGM, SEND - RHO, RHO, STRIDED, true, LAZY8, CC, T, false, TWL>(a, bx, by, threadIdx.x, lds, twl);
            tile_sync<T>();
            inv<I - 1, SEND - RHO, LAZY8>(a, bx, by, lds, twl);
        }
    }
};
// the register rounds of one pass over the tile staged in LDS (each round ends with a barrier)
// PM = the caller guarantees pseudo-Mersenne moduli (the fused row kernel): only the lazy rounds are instantiated
template <int LOGM, bool STRIDED, bool INVERSE, int CC, int T = NTT_THREADS, int SCH = T, bool TWL = false, bool PM = false>
static __device__ __forceinline__ void ntt_tile_rounds(const NttArgs &a, int bx, int by, u64 *lds, const u64 *twl = nullptr)
{
    if constexpr (PM) {
        if constexpr (!INVERSE) NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template fwd<0, 0, true>(a, bx, by, lds, twl);
        else NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template inv<NttSched<LOGM, SCH>::R - 1, LOGM, true>(a, bx, by, lds, twl);
    } else if constexpr (!INVERSE) {
        if (a.lazy8) NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template fwd<0, 0, true>(a, bx, by, lds, twl);
        else NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template fwd<0, 0, false>(a, bx, by, lds, twl);
    } else {
        if (a.lazy8) NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template inv<NttSched<LOGM, SCH>::R - 1, LOGM, true>(a, bx, by, lds, twl);
        else NttRounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL>::template inv<NttSched<LOGM, SCH>::R - 1, LOGM, false>(a, bx, by, lds, twl);
    }
}

// one pass of one tile: load phase, register rounds through LDS, store phase
template <int LOGM, bool STRIDED, bool INVERSE, bool FULL, int T = NTT_THREADS, int SCH = T, int TL = NttTile::LOG, bool TWL = false, bool PM = false>
static __device__ __forceinline__ void ntt_pass_tile(const NttArgs &a, int bx, int by, u64 *lds, u64 *twl = nullptr)
{
    constexpr int CM = FULL ? LOGM : -1, CC = FULL ? TL - LOGM : -1;
    if constexpr (TWL) ks_row_twiddle_fill<CM, CC>(a, bx, by, INVERSE, threadIdx.x, twl);
    ntt_body_load<STRIDED, INVERSE, CM, CC, T>(a, bx, by, threadIdx.x, lds);
    tile_sync<T>();
    ntt_tile_rounds<LOGM, STRIDED, INVERSE, CC, T, SCH, TWL, PM>(a, bx, by, lds, twl);
    ntt_body_store<STRIDED, INVERSE, CM, CC, T>(a, bx, by, threadIdx.x, lds);
}

// FULL: the tile is 2^LOGM points x 2^(12 - LOGM) columns (every launch with N >= 4096) -> geometry folds into constants
template <int LOGM, bool STRIDED, bool INVERSE, bool FULL>
__global__ void __launch_bounds__(NTT_THREADS, 4) ntt_pass_kernel(NttArgs a)
{
    __shared__ u64 lds[NttLds::ELEMS];
    ntt_pass_tile<LOGM, STRIDED, INVERSE, FULL>(a, NTT_BX(a), NTT_BY(a), lds);
}
// Two independent batches of the same pass in ONE grid (polynomials [0, a1.count) use a1, the rest a2): a small batch
// rides in the tail of a big one instead of paying a launch of its own that cannot fill the 1024 workgroup slots.
// The argument block is selected per workgroup from the kernarg segment (uniform), the code is shared.
template <int LOGM, bool STRIDED, bool INVERSE, bool FULL>
__global__ void __launch_bounds__(NTT_THREADS, 4) ntt_pass2_kernel(NttArgs a1, NttArgs a2)
{
    __shared__ u64 lds[NttLds::ELEMS];
    const int by = NTT_BY(a1);
    const bool second = by >= a1.count;
    // Selecting between the two by-value blocks would copy them to scratch; index the kernarg segment instead (a1 at
    // offset 0, a2 right behind it), so every field stays a scalar load from constant memory.
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    typedef __attribute__((address_space(4))) const NttArgs *args_ptr;
    (void)a2;
    const args_ptr pa = (args_ptr)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + (second ? sizeof(NttArgs) : 0));
    ntt_pass_tile<LOGM, STRIDED, INVERSE, FULL>(*(const NttArgs *)pa, NTT_BX(a1), second ? by - a1.count : by, lds);
}

template <bool STRIDED, bool INVERSE>
static void launch_pass(NttArgs a, int logm, int other, hipStream_t st)
{
    a.logm = logm;
    int logc = NttTile::LOG - logm;
    if (logc > other) logc = other;
    a.logc = logc;
    a.tiles_log = other - logc;
    dim3 grid((unsigned)(((size_t)a.count) << a.tiles_log));
    const bool full = logc == NttTile::LOG - logm;
#define NTT_LAUNCH(M_)                                                                                                     \
    case M_:                                                                                                               \
        if (full) hipLaunchKernelGGL((ntt_pass_kernel<M_, STRIDED, INVERSE, true>), grid, dim3(NTT_THREADS), 0, st, a);    \
        else hipLaunchKernelGGL((ntt_pass_kernel<M_, STRIDED, INVERSE, false>), grid, dim3(NTT_THREADS), 0, st, a);        \
        break;
    switch (logm) {
        NTT_LAUNCH(5) NTT_LAUNCH(6) NTT_LAUNCH(7) NTT_LAUNCH(8)
    default: snprintf(g_rt_err, sizeof(g_rt_err), "unsupported NTT pass size 2^%d", logm); break;
    }
#undef NTT_LAUNCH
}
void k_ntt(const NttArgs &a, bool inverse, rt_stream s)
{
    if (a.count <= 0) return;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    hipStream_t st = (hipStream_t)s;
    if (!inverse) {
        launch_pass<true, false>(a, n1, n2, st);   // strided pass: global stages 0..n1-1
        launch_pass<false, false>(a, n2, n1, st);  // row pass: stages n1..n-1
    } else {
        launch_pass<false, true>(a, n2, n1, st);
        launch_pass<true, true>(a, n1, n2, st);
    }
}

// one pass only: `second` = false runs the first pass of the transform (its load ops), true the second (its store ops)
void k_ntt_pass(const NttArgs &a, bool inverse, bool second, rt_stream s)
{
    if (a.count <= 0) return;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    hipStream_t st = (hipStream_t)s;
    if (!inverse) { if (!second) launch_pass<true, false>(a, n1, n2, st); else launch_pass<false, false>(a, n2, n1, st); }
    else { if (!second) launch_pass<false, true>(a, n2, n1, st); else launch_pass<true, true>(a, n1, n2, st); }
}

template <bool STRIDED, bool INVERSE>
static void launch_pass2(NttArgs a1, NttArgs a2, int logm, int other, hipStream_t st)
{
    int logc = NttTile::LOG - logm;
    if (logc > other) logc = other;
    a1.logm = a2.logm = logm;
    a1.logc = a2.logc = logc;
    a1.tiles_log = a2.tiles_log = other - logc;
    dim3 grid((unsigned)(((size_t)a1.count + a2.count) << a1.tiles_log));
    const bool full = logc == NttTile::LOG - logm;
#define NTT_LAUNCH2(M_)                                                                                                   \
    case M_:                                                                                                              \
        if (full) hipLaunchKernelGGL((ntt_pass2_kernel<M_, STRIDED, INVERSE, true>), grid, dim3(NTT_THREADS), 0, st, a1, a2);      \
        else hipLaunchKernelGGL((ntt_pass2_kernel<M_, STRIDED, INVERSE, false>), grid, dim3(NTT_THREADS), 0, st, a1, a2);          \
        break;
    switch (logm) {
        NTT_LAUNCH2(5) NTT_LAUNCH2(6) NTT_LAUNCH2(7) NTT_LAUNCH2(8)
    default: snprintf(g_rt_err, sizeof(g_rt_err), "unsupported NTT pass size 2^%d", logm); break;
    }
#undef NTT_LAUNCH2
}
// forward transforms of two batches (same N) in shared grids
void k_ntt2_fwd(const NttArgs &a1, const NttArgs &a2, rt_stream s)
{
    if (a1.count <= 0) { k_ntt(a2, false, s); return; }
    if (a2.count <= 0) { k_ntt(a1, false, s); return; }
    int n1, n2;
    ntt_split(a1.logn, n1, n2);
    launch_pass2<true, false>(a1, a2, n1, n2, (hipStream_t)s);
    launch_pass2<false, false>(a1, a2, n2, n1, (hipStream_t)s);
}
// first (strided) pass of the forward transforms of two batches in one grid
void k_ntt2_fwd_first(const NttArgs &a1, const NttArgs &a2, rt_stream s)
{
    if (a1.count <= 0) { k_ntt_pass(a2, false, false, s); return; }
    if (a2.count <= 0) { k_ntt_pass(a1, false, false, s); return; }
    int n1, n2;
    ntt_split(a1.logn, n1, n2);
    launch_pass2<true, false>(a1, a2, n1, n2, (hipStream_t)s);
}
// inverse transforms of two batches where the store epilogue of the SECOND may read results of the first: the row passes
// (no such dependency yet) share one grid, the strided passes run one after the other
void k_ntt2_inv(const NttArgs &a1, const NttArgs &a2, rt_stream s)
{
    if (a1.count <= 0 || a2.count <= 0) { k_ntt(a1, true, s); k_ntt(a2, true, s); return; }
    int n1, n2;
    ntt_split(a1.logn, n1, n2);
    hipStream_t st = (hipStream_t)s;
    launch_pass2<false, true>(a1, a2, n2, n1, st);
    launch_pass<true, true>(a1, n1, n2, st);
    launch_pass<true, true>(a2, n1, n2, st);
}

// ---------------------------------------------------------------- fused key-switch row kernel
// 8 points per lane: the 2 x 8 lazy sums of a lane live in registers beside a radix-8 round inside the 128-VGPR budget of
// 4 waves per SIMD.  Measured alternatives (round 2, config 2, transcipherings/s): 4096-point tiles on 256 lanes x 16 points
// need 173+ VGPRs -- 200 at 3 waves per SIMD (spills), 237 at 2 waves per SIMD; 4096-point tiles on 512 lanes: 238.
// The tile size (csrc/hhe_kernel_bodies.h, KSROW_TL) is now one wave's worth; the launch bound asks for 4 waves per SIMD.
// Diagnostic build only (-DHHE_STAMPS, tools/stamps.sh; never the product): every 64th key-switch workgroup records the shader
// clock (s_memtime) at its phase boundaries into a buffer of its own that nothing else reads -- the timeline of one wave.
#ifdef HHE_STAMPS
constexpr int STAMP_SLOTS = 24, STAMP_WGS = 4096;
__device__ u64 g_ks_stamps[STAMP_WGS * STAMP_SLOTS];
__device__ int g_ks_stamp_launch;  // 1 while the ONE launch HHE_STAMP_LAUNCH selects runs (set from the host, in stream order)
// the clock values stay in scalar registers until the workgroup is done (a store per stamp would make the wave wait for all its
// outstanding vector memory operations at every phase boundary and distort the timeline it is meant to show)
#define KS_STAMP(i) do { if (st_rec) stamps_[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define KS_STAMP_I(base, I) do { const u64 t_ = __builtin_amdgcn_s_memtime(); if (!st_rec) { } else if ((I) == 0) stamps_[(base)] = t_; else if ((I) == 1) stamps_[(base) + 3] = t_; else if ((I) == 2) stamps_[(base) + 6] = t_; } while (0)
#define KS_STAMP_FLUSH() do { if (stamp_on && threadIdx.x == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 17; i_++) g_ks_stamps[stamp_wg * STAMP_SLOTS + i_] = stamps_[i_]; } } while (0)
extern "C" int hhe_debug_read_stamps(u64 *out, size_t words)
{
    if (words > (size_t)STAMP_WGS * STAMP_SLOTS) words = (size_t)STAMP_WGS * STAMP_SLOTS;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ks_stamps), words * 8) == hipSuccess ? 0 : -1;
}
#else
#define KS_STAMP(i) do { } while (0)
#define KS_STAMP_I(base, I) do { } while (0)
#define KS_STAMP_FLUSH() do { } while (0)
#endif
#ifndef KSROW_WAVES
#define KSROW_WAVES 4
#endif

template <int LOGM>
__global__ void __launch_bounds__(KSROW_THREADS, KSROW_WAVES) ks_row_kernel(NttArgs a, KsRowArgs x, NttArgs c0)
{
    // 256-point rows: the twiddles of the first two rounds of the tile's rows are staged in LDS once per direction (one
    // array: tile | twiddle heap) and serve the L digit transforms / the inverse transforms of the workgroup
    constexpr bool TWL = LOGM == 8;
    __shared__ u64 lds[KSROW_LDS + (TWL ? KSROW_TWL : 0)];
    u64 *const twl = TWL ? lds + KSROW_LDS : nullptr;
    constexpr int CC = KSROW_TILE_LOG - LOGM, T = KSROW_THREADS, SCH = KSROW_SCHED;
    // c0.count polynomials of the grid are an ordinary forward row pass (the c0 branch of the previous rotation step with
    // its mod-down epilogue): memory-bound tiles that run beside the arithmetic-bound key-switch tiles
    // the short c0 tiles come LAST in the grid: they fill the tail behind the long key-switch workgroups (c0 first: 290.7 /s,
    // c0 last: 292.0 /s on one box; alternating the two kinds: 2 % slower than either)
    const unsigned nmain = (unsigned)(x.B * x.K) << a.tiles_log;
#ifdef HHE_STAMPS
    const bool stamp_on = (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < STAMP_WGS && g_ks_stamp_launch;
    const unsigned stamp_wg = blockIdx.x >> 6;
    u64 stamps_[17];
#pragma unroll
    for (int i_ = 0; i_ < 17; i_++) stamps_[i_] = 0;
#endif
    // Block -> (item, limb, tile): the ITEM index runs fastest within groups of eight (limb, tile) units, and the unit's position in its
    // group is the block's XCD (blocks are dealt round-robin to the 8 XCDs).  Every item's workgroup for one (limb, tile) -- the readers of
    // the same 16-KB key slices and twiddles -- runs on the same XCD at about the same time, so a key is fetched into each L2 once per
    // launch instead of once per item (config 5: 63 MB of key + quotients against 4 MB of L2 per XCD; the item-major order read 2.7 GB per launch).
    const unsigned tmask = (1u << a.tiles_log) - 1;
    if (blockIdx.x >= nmain) {
        unsigned cb = blockIdx.x - nmain;
        if (c0.count == x.B * x.L) {   // (item b, limb j) polynomials: same order as the key-switch tiles
            const unsigned r = cb >> 3, u = (r / (unsigned)x.B) * 8 + (cb & 7);
            cb = (((r % (unsigned)x.B) * x.L + (u >> a.tiles_log)) << a.tiles_log) | (u & tmask);
        }
#ifdef HHE_STAMPS
        const bool st_rec = true;
#endif
        KS_STAMP(0);
        ntt_pass_tile<LOGM, false, false, true, T, SCH, KSROW_TILE_LOG, TWL, true>(c0, (int)(cb & ((1u << a.tiles_log) - 1)), (int)(cb >> a.tiles_log), lds, twl);
        KS_STAMP(16);
        KS_STAMP_FLUSH();
        return;
    }
    const unsigned rr = blockIdx.x >> 3, uu = (rr / (unsigned)x.B) * 8 + (blockIdx.x & 7);
    const int b = (int)(rr % (unsigned)x.B), J = (int)(uu >> a.tiles_log), bx = (int)(uu & tmask), tid = threadIdx.x;
    const size_t n = (size_t)1 << a.logn;
    {
#ifdef HHE_STAMPS
        const bool st_rec = true;
#endif
        KS_STAMP(0);
        u64 acc0[2 * KSROW_NP], acc1[2 * KSROW_NP];
#pragma unroll
        for (int k = 0; k < 2 * KSROW_NP; k++) { acc0[k] = 0; acc1[k] = 0; }
        U2 pf[KSROW_NP];  // the tile of the next digit, in flight
        ks_row_tile_fetch<LOGM, CC>(a, bx, (b * x.L + 0) * x.K + J, tid, pf);   // beside the twiddle fill's own loads
        if (TWL) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, false, tid, twl);
        KS_STAMP(1);
        for (int I = 0; I < x.L; I++) {
            const int by = (b * x.L + I) * x.K + J;
            ks_row_tile_commit<LOGM, CC>(a, bx, by, tid, pf, lds);
            tile_sync<T>();
            if (I + 1 < x.L) ks_row_tile_fetch<LOGM, CC>(a, bx, by + x.K, tid, pf);
            KS_STAMP_I(2, I);
            ntt_tile_rounds<LOGM, false, false, CC, T, SCH, TWL, true>(a, bx, by, lds, twl);
            KS_STAMP_I(3, I);
            if (TWL && I == x.L - 1) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, true, tid, twl);  // behind the key products: the forward rounds are over
            ks_row_mac_phase<LOGM, CC>(x, a, bx, b, J, I, tid, lds, acc0, acc1);
            tile_sync<T>();
            KS_STAMP_I(4, I);
        }
        if (J < x.L) {
            if (x.U0) {  // generic key switch: S_0[j] is inverse-transformed as well
                ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc0, nullptr);
                tile_sync<T>();
                ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
                ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, x.U0 + (size_t)b * x.u_stride + (size_t)J * n);
                tile_sync<T>();
            } else ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc0, x.S + (((size_t)b * 2 + 0) * x.K + J) * n);
            KS_STAMP(13);
            ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc1, nullptr);
            tile_sync<T>();
            KS_STAMP(14);
            ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
            KS_STAMP(15);
            ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, x.U1 + (size_t)b * x.u_stride + (size_t)J * n);
        } else {
            ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc0, nullptr);
            tile_sync<T>();
            KS_STAMP(11);
            ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
            KS_STAMP(12);
            ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, x.Usp + ((size_t)b * 2 + 0) * n);
            tile_sync<T>();
            KS_STAMP(13);
            ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc1, nullptr);
            tile_sync<T>();
            KS_STAMP(14);
            ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
            KS_STAMP(15);
            ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, x.Usp + ((size_t)b * 2 + 1) * n);
        }
        KS_STAMP(16);
    }
    KS_STAMP_FLUSH();
}
// A child of an FC trie node from the node's shared digit transforms: key inner product over the digits read through the Galois map
// (ks_row_mac_gather) + the inverse row pass of all 2K sums, one (item, key limb, row tile) per single-wave workgroup -- the sums never
// make a round trip (before: ks_mac_kernel wrote S, a row-pass launch read it back).  Same block order and outputs as the generic
// variant of ks_row_kernel (U0 / U1 / Usp).
template <int LOGM>
#ifndef PERMROW_WAVES
#define PERMROW_WAVES 3   // 154 VGPRs, no spills; at 4 waves per SIMD (128 VGPRs) 14 registers spill: 34.5 vs 32.5 ms per MNIST sample
#endif
__global__ void __launch_bounds__(KSROW_THREADS, PERMROW_WAVES) ks_perm_row_kernel(NttArgs a, KsRowArgs x)
{
    constexpr bool TWL = LOGM == 8;
    __shared__ u64 lds[KSROW_LDS + (TWL ? KSROW_TWL : 0)];
    u64 *const twl = TWL ? lds + KSROW_LDS : nullptr;
    constexpr int CC = KSROW_TILE_LOG - LOGM, T = KSROW_THREADS, SCH = KSROW_SCHED;
    const unsigned tmask = (1u << a.tiles_log) - 1;
    const unsigned rr = blockIdx.x >> 3, uu = (rr / (unsigned)x.B) * 8 + (blockIdx.x & 7);
    const int b = (int)(rr % (unsigned)x.B), J = (int)(uu >> a.tiles_log), bx = (int)(uu & tmask), tid = threadIdx.x;
    if (J >= x.K) return;  // the grid is padded to a multiple of eight (limb, tile) units
    const size_t n = (size_t)1 << a.logn;
    u64 acc0[2 * KSROW_NP], acc1[2 * KSROW_NP];
#pragma unroll
    for (int k = 0; k < 2 * KSROW_NP; k++) { acc0[k] = 0; acc1[k] = 0; }
    if (TWL) ks_row_twiddle_fill<LOGM, CC>(a, bx, J, true, tid, twl);
    for (int I = 0; I < x.L; I++) ks_row_mac_gather<LOGM, CC>(x, a, bx, b, J, I, tid, acc0, acc1);
    u64 *const out0 = J < x.L ? x.U0 + (size_t)b * x.u_stride + (size_t)J * n : x.Usp + ((size_t)b * 2 + 0) * n;
    u64 *const out1 = J < x.L ? x.U1 + (size_t)b * x.u_stride + (size_t)J * n : x.Usp + ((size_t)b * 2 + 1) * n;
    ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc0, nullptr);
    tile_sync<T>();
    ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
    ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, out0);
    tile_sync<T>();
    ks_row_flush_phase<LOGM, CC>(a, bx, J, tid, lds, acc1, nullptr);
    tile_sync<T>();
    ntt_tile_rounds<LOGM, false, true, CC, T, SCH, TWL, true>(a, bx, J, lds, twl);
    ks_row_store_phase<LOGM, CC>(a, bx, J, tid, lds, out1);
}
int k_ks_perm_row(const NttArgs &a0, const KsRowArgs &x, rt_stream s)
{
    NttArgs a = a0;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    a.logm = n2;
    a.logc = KSROW_TILE_LOG - n2;
    a.tiles_log = n1 - a.logc;
    const size_t units = ((size_t)x.K << a.tiles_log);   // (limb, tile) units; with tiles_log >= 3 a multiple of eight
    dim3 grid((unsigned)(((units + 7) & ~(size_t)7) * x.B));
    hipStream_t st = (hipStream_t)s;
    switch (n2) {
    case 6: hipLaunchKernelGGL((ks_perm_row_kernel<6>), grid, dim3(KSROW_THREADS), 0, st, a, x); break;
    case 7: hipLaunchKernelGGL((ks_perm_row_kernel<7>), grid, dim3(KSROW_THREADS), 0, st, a, x); break;
    case 8: hipLaunchKernelGGL((ks_perm_row_kernel<8>), grid, dim3(KSROW_THREADS), 0, st, a, x); break;
    default: snprintf(g_rt_err, sizeof(g_rt_err), "ks_perm_row: unsupported row pass size 2^%d", n2); return -1;
    }
    return 0;
}
int k_ks_row(const NttArgs &a0, const KsRowArgs &x, const NttArgs *c0_row, rt_stream s)
{
    NttArgs a = a0, c0;
    int n1, n2;
    ntt_split(a.logn, n1, n2);
    a.logm = n2;
    a.logc = KSROW_TILE_LOG - n2;
    a.tiles_log = n1 - a.logc;
    if (c0_row) { c0 = *c0_row; c0.logm = a.logm; c0.logc = a.logc; c0.tiles_log = a.tiles_log; }
    else { memset(&c0, 0, sizeof(c0)); }
    dim3 grid((unsigned)(((size_t)x.B * x.K + c0.count) << a.tiles_log));
    hipStream_t st = (hipStream_t)s;
#ifdef HHE_STAMPS
    {
        static const int flags[2] = {0, 1};
        static int launch = 0;
        const char *e = getenv("HHE_STAMP_LAUNCH");
        const int on = launch++ == (e ? atoi(e) : 300);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ks_stamp_launch), &flags[on], sizeof(int), 0, hipMemcpyHostToDevice, st);
    }
#endif
    switch (n2) {
    case 6: hipLaunchKernelGGL((ks_row_kernel<6>), grid, dim3(KSROW_THREADS), 0, st, a, x, c0); break;
    case 7: hipLaunchKernelGGL((ks_row_kernel<7>), grid, dim3(KSROW_THREADS), 0, st, a, x, c0); break;
    case 8: hipLaunchKernelGGL((ks_row_kernel<8>), grid, dim3(KSROW_THREADS), 0, st, a, x, c0); break;
    default: snprintf(g_rt_err, sizeof(g_rt_err), "ks_row: unsupported row pass size 2^%d", n2); return -1;
    }
    return 0;
}

// ---------------------------------------------------------------- element-wise family
constexpr int ELT_THREADS = 256;
static inline unsigned nblocks(size_t total) { return (unsigned)((total + ELT_THREADS - 1) / ELT_THREADS); }
#define GID ((size_t)blockIdx.x * ELT_THREADS + threadIdx.x)

__global__ void __launch_bounds__(ELT_THREADS) elt_kernel(EltArgs a, int op) { elt_body(a, op, GID); }
__global__ void __launch_bounds__(ELT_THREADS) copy_items_kernel(CopyItemsArgs a) { copy_items_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) galois_kernel(GaloisArgs a) { galois_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) perm_kernel(PermArgs a) { perm_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) ks_mac_kernel(KsMacArgs a) { ks_mac_body(a, GID); }
template <int LL, int MODE> __global__ void __launch_bounds__(ELT_THREADS) ks_mac_t_kernel(KsMacArgs a) { ks_mac_body_t<LL, MODE>(a, GID); }
template <int LL> __global__ void __launch_bounds__(ELT_THREADS) ks_mac_leaves_kernel(KsMacLeavesArgs a) { ks_mac_leaves_body<LL>(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) ks_corr_kernel(KsCorrArgs a) { ks_corr_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) ks_finish_kernel(KsFinishArgs a) { ks_finish_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) leaf_sum_kernel(LeafSumArgs a) { leaf_sum_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) leaf_round_kernel(LeafRoundArgs a) { leaf_round_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) csum_add_kernel(CsumArgs a) { csum_add_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) csum_digits_kernel(CsumArgs a) { csum_digits_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) csum_c0_kernel(CsumArgs a) { csum_c0_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) add_plain_kernel(AddPlainArgs a) { add_plain_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) encode_scatter_kernel(EncodeArgs a) { encode_scatter_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) diag_kernel(DiagArgs a) { diag_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) bsgs_diag_kernel(BsgsDiagArgs a) { bsgs_diag_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) behz_extend_kernel(BehzExtendArgs a) { behz_extend_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) tensor_kernel(TensorArgs a) { tensor_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) behz_floor_kernel(BehzFloorArgs a) { behz_floor_body(a, GID); }

#define LAUNCH1D(kern, total, s, ...)                                                                   \
    do {                                                                                                \
        size_t _t = (total);                                                                            \
        if (_t) hipLaunchKernelGGL(kern, dim3(nblocks(_t)), dim3(ELT_THREADS), 0, (hipStream_t)(s), __VA_ARGS__); \
    } while (0)

void k_elt(const EltArgs &a, int op, rt_stream s) { LAUNCH1D(elt_kernel, (size_t)a.count << a.logn, s, a, op); }
void k_copy_items(const CopyItemsArgs &a, rt_stream s) { LAUNCH1D(copy_items_kernel, a.count * (a.words >> 1), s, a); }
void k_galois(const GaloisArgs &a, rt_stream s) { LAUNCH1D(galois_kernel, (size_t)a.count << (a.logn - 1), s, a); }
void k_perm(const PermArgs &a, rt_stream s) { LAUNCH1D(perm_kernel, (size_t)a.count << a.logn, s, a); }
template <int MODE> static void launch_ks_mac_t(const KsMacArgs &a, rt_stream s)
{
    const size_t total = ((size_t)a.B * a.K) << (a.logn - 1);
    switch (a.L) {
    case 1: LAUNCH1D((ks_mac_t_kernel<1, MODE>), total, s, a); break;
    case 2: LAUNCH1D((ks_mac_t_kernel<2, MODE>), total, s, a); break;
    case 3: LAUNCH1D((ks_mac_t_kernel<3, MODE>), total, s, a); break;
    default: LAUNCH1D((ks_mac_t_kernel<4, MODE>), total, s, a); break;
    }
}
void k_ks_mac(const KsMacArgs &a, rt_stream s)
{
    switch (ks_mac_mode(a)) {
    case KS_PLAIN: launch_ks_mac_t<KS_PLAIN>(a, s); break;
    case KS_ACC: launch_ks_mac_t<KS_ACC>(a, s); break;
    case KS_PERM: launch_ks_mac_t<KS_PERM>(a, s); break;
    case KS_LEAF: launch_ks_mac_t<KS_LEAF>(a, s); break;
    default: LAUNCH1D(ks_mac_kernel, ((size_t)a.B * a.K) << (a.logn - 1), s, a); break;
    }
}
int k_ks_mac_leaves(const KsMacLeavesArgs &a, rt_stream s)
{
    const size_t total = ((size_t)a.B * (a.sp_only ? 1 : a.K)) << (a.logn - 1);
    switch (a.L) {
    case 1: LAUNCH1D((ks_mac_leaves_kernel<1>), total, s, a); break;
    case 2: LAUNCH1D((ks_mac_leaves_kernel<2>), total, s, a); break;
    case 3: LAUNCH1D((ks_mac_leaves_kernel<3>), total, s, a); break;
    case 4: LAUNCH1D((ks_mac_leaves_kernel<4>), total, s, a); break;
    default: return -1;
    }
    return 0;
}
void k_ks_corr(const KsCorrArgs &a, rt_stream s) { LAUNCH1D(ks_corr_kernel, ((size_t)2 * a.K) << a.logn, s, a); }
void k_ks_finish(const KsFinishArgs &a, rt_stream s) { LAUNCH1D(ks_finish_kernel, ((size_t)a.B * 2 * a.L) << a.logn, s, a); }
void k_leaf_sum(const LeafSumArgs &a, rt_stream s) { LAUNCH1D(leaf_sum_kernel, ((size_t)a.B * 2 * a.L) << a.logn, s, a); }
void k_csum_add(const CsumArgs &a, rt_stream s) { LAUNCH1D(csum_add_kernel, ((size_t)a.B * a.L) << (a.logn - 1), s, a); }
void k_csum_c0(const CsumArgs &a, rt_stream s) { LAUNCH1D(csum_c0_kernel, ((size_t)a.B * a.L) << (a.logn - 1), s, a); }
void k_csum_digits(const CsumArgs &a, rt_stream s) { LAUNCH1D(csum_digits_kernel, ((size_t)a.B * a.L) << (a.logn - 1), s, a); }
void k_leaf_round(const LeafRoundArgs &a, rt_stream s) { LAUNCH1D(leaf_round_kernel, ((size_t)a.B * 2) << (a.logn - 1), s, a); }
void k_add_plain(const AddPlainArgs &a, rt_stream s) { LAUNCH1D(add_plain_kernel, (size_t)a.B << a.logn, s, a); }
void k_encode_scatter(const EncodeArgs &a, rt_stream s)
{
    LAUNCH1D(encode_scatter_kernel, (size_t)a.B * a.count * (a.second_off >= 0 ? 2 : 1), s, a);
}
void k_diag(const DiagArgs &a, rt_stream s) { LAUNCH1D(diag_kernel, (size_t)(PASTA_R + 1) * PASTA_T * 2 * PASTA_T, s, a); }
void k_bsgs_diag(const BsgsDiagArgs &a, rt_stream s) { LAUNCH1D(bsgs_diag_kernel, (size_t)(PASTA_R + 1) * PASTA_T * 2 * PASTA_T, s, a); }
void k_behz_extend(const BehzExtendArgs &a, rt_stream s) { LAUNCH1D(behz_extend_kernel, (size_t)a.P << a.logn, s, a); }
void k_tensor(const TensorArgs &a, rt_stream s) { LAUNCH1D(tensor_kernel, ((size_t)a.B * a.limbs) << a.logn, s, a); }
void k_behz_floor(const BehzFloorArgs &a, rt_stream s) { LAUNCH1D(behz_floor_kernel, (size_t)a.P << a.logn, s, a); }

// ---------------------------------------------------------------- plain PASTA-3 (client / analyst side)
__global__ void __launch_bounds__(64) pasta_xof_kernel(PastaXofArgs a) { pasta_xof_body(a, (size_t)blockIdx.x * 64 + threadIdx.x); }
__global__ void __launch_bounds__(PASTA_PLAIN_THREADS) pasta_plain_kernel(PastaPlainArgs a)
{
    __shared__ u64 lds[PASTA_PLAIN_LDS];
    pasta_plain_schedule([&](int kind, int layer, int i) {
        pasta_plain_phase(a, (int)blockIdx.x, (int)threadIdx.x, layer, kind, i, lds);
        __syncthreads();
    });
}
__global__ void __launch_bounds__(ELT_THREADS) pasta_crypt_kernel(PastaCryptArgs a) { pasta_crypt_body(a, GID); }
void k_pasta_xof(const PastaXofArgs &a, rt_stream s)
{
    if (a.nblocks > 0) hipLaunchKernelGGL(pasta_xof_kernel, dim3((a.nblocks + 63) / 64), dim3(64), 0, (hipStream_t)s, a);
}
void k_pasta_plain(const PastaPlainArgs &a, rt_stream s)
{
    if (a.nblocks > 0) hipLaunchKernelGGL(pasta_plain_kernel, dim3(a.nblocks), dim3(PASTA_PLAIN_THREADS), 0, (hipStream_t)s, a);
}
void k_pasta_crypt(const PastaCryptArgs &a, rt_stream s) { LAUNCH1D(pasta_crypt_kernel, a.S * a.nwords, s, a); }
__global__ void __launch_bounds__(ELT_THREADS) decrypt_round_kernel(DecryptArgs a) { decrypt_round_body(a, GID); }
__global__ void __launch_bounds__(ELT_THREADS) decode_gather_kernel(DecodeArgs a) { decode_gather_body(a, GID); }
void k_decrypt_round(const DecryptArgs &a, rt_stream s) { LAUNCH1D(decrypt_round_kernel, a.B << a.logn, s, a); }
void k_decode_gather(const DecodeArgs &a, rt_stream s) { LAUNCH1D(decode_gather_kernel, a.B << a.logn, s, a); }
