// ubench_bfly3.hip -- sustained (about one second per mode: long enough for the socket power limit to set the clock) register-only rate
// of three spellings of the lazy 64-bit butterfly for q = 2^60 - c, c < 2^24, radix-8 rounds on 8 words per lane, 4 waves per SIMD:
// mode 1 = truncated Shoup product x*w + H'*(2^64 - q) and the pseudo-Mersenne fold of X (the product's code today);
// mode 2 = the same with H'*(2^64 - q) spelled H'*c - (H' << 60) (two multiplies less);
// mode 3 = no Shoup quotient at all: the 128-bit product x*w folded twice by 2^60 = c (mod q): 7 multiplies, 8-byte twiddles.
// Prints butterflies/s and the shader clock during the run (s_memtime ticks / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 mulhi_trunc(u64 x, u64 s)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)s, s1 = (u32)(s >> 32);
    return (u64)x1 * s1 + __umulhi(x0, s1) + __umulhi(x1, s0);
}
__device__ __forceinline__ u64 lazy_trunc(u64 x, u64 w, u64 ws, u64 nq) { return x * w + mulhi_trunc(x, ws) * nq; }
__device__ __forceinline__ u64 lazy_trunc_c(u64 x, u64 w, u64 ws, u32 c)
{
    const u64 h = mulhi_trunc(x, ws);
    return x * w + ((u64)(u32)h * c + ((u64)((u32)(h >> 32) * c) << 32)) - (h << 60);
}
__device__ __forceinline__ u64 pmfold60(u64 x, u32 c) { return (u64)((u32)(x >> 32) >> 28) * c + (x & ((1ull << 60) - 1)); }
// x * w mod q, result below 2^61 + 2^52 for any 64-bit x and w < 2^60
__device__ __forceinline__ u64 mul_fold(u64 x, u64 w, u32 c)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u64 p00 = (u64)x0 * w0;
    const u64 m = (u64)x0 * w1 + (p00 >> 32);          // < 2^60 + 2^32
    const u64 m2 = (u64)x1 * w0 + (u32)m;              // < 2^64
    const u64 hi = (u64)x1 * w1 + (m >> 32) + (m2 >> 32);
    const u32 l0 = (u32)p00, l1 = (u32)m2, h0 = (u32)hi, h1 = (u32)(hi >> 32);
    // P = hi:l1:l0 = ph * 2^60 + pl
    const u32 ph0 = __builtin_amdgcn_alignbit(h0, l1, 28), ph1 = __builtin_amdgcn_alignbit(h1, h0, 28);
    const u64 pl = ((u64)(l1 & 0x0fffffffu) << 32) | l0;
    // A = ph * c (< 2^88) = a1 * 2^32 + (u32)a0
    const u64 a0 = (u64)ph0 * c;
    const u64 a1 = (u64)ph1 * c + (a0 >> 32);
    const u32 ah60 = (u32)(a1 >> 28);                   // A >> 60 (< 2^28)
    const u64 am = ((u64)((u32)a1 & 0x0fffffffu) << 32) | (u32)a0;   // A mod 2^60
    return (u64)ah60 * c + (pl + am);
}
template <int MODE> __global__ void __launch_bounds__(256) k(u64 *out, int iters, u64 q, u64 w0, u64 ws0, u32 c, u64 nq, u64 *clk)
{
    u64 v[8];
    const u64 q4 = q << 2;
    for (int i = 0; i < 8; i++) v[i] = (u64)(threadIdx.x * 977 + i * 131 + blockIdx.x) * 0x9E3779B97F4A7C15ull >> 5;
    u64 w = w0 + threadIdx.x, ws = ws0 + threadIdx.x;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int half = 4 >> u;
#pragma unroll
            for (int b = 0; b < (1 << u); b++) {
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int k0 = b * 2 * half + j, k1 = k0 + half;
                    u64 x = v[k0];
                    if (u == 0) x = pmfold60(x, c);
                    u64 y;
                    if (MODE == 1) y = lazy_trunc(v[k1], w + b, ws + b, nq);
                    else if (MODE == 2) y = lazy_trunc_c(v[k1], w + b, ws + b, c);
                    else y = mul_fold(v[k1], w + b, c);
                    v[k0] = x + y;
                    v[k1] = x + q4 - y;
                }
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u64 s = 0;
    for (int i = 0; i < 8; i++) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <int MODE> void run(int iters)
{
    const int blocks = 256 * 4;
    u64 *out, *clk;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    (void)hipMalloc(&clk, 8);
    const u64 q = 1152921504595968001ULL;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, q, 88651361085495ULL, 123456789ULL, (u32)((1ull << 60) - q), 0 - q, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        u64 ticks = 0;
        (void)hipMemcpy(&ticks, clk, 8, hipMemcpyDeviceToHost);
        const double bf = (double)blocks * 256 * iters * 12;
        printf("mode %d: %.1f ms, %.3e butterflies/s, s_memtime %.0f MHz-equivalent\n", MODE, ms, bf / (ms * 1e-3), ticks / (ms * 1e3));
        fflush(stdout);
    }
    (void)hipFree(out); (void)hipFree(clk);
}
int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 700000;   // 700000 = about one second per launch
    run<1>(iters); run<2>(iters); run<3>(iters); run<1>(iters); run<3>(iters);
    return 0;
}
