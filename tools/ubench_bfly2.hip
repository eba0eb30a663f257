// ubench_bfly2.hip -- register-only rate of two spellings of the lazy 64-bit butterfly on gfx950 (radix-8 rounds on 8 words per lane):
// mode 0 = exact Shoup high product (3 v_mad_u64_u32 + v_mul_hi_u32 + re-pairing moves), products in [0,2q), X folded with a compare-select once per round;
// mode 1 = truncated high product x1*s1 + hi32(x0*s1) + hi32(x1*s0) (>= the exact one - 2: products in [0,4q)) and the pseudo-Mersenne fold
//          X -> (X mod 2^60) + (X >> 60) * c for q = 2^60 - c (no compare, no VCC) once per round.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 lazy_exact(u64 x, u64 w, u64 ws, u64 nq) { return x * w + __umul64hi(x, ws) * nq; }
__device__ __forceinline__ u64 mulhi_trunc(u64 x, u64 s)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)s, s1 = (u32)(s >> 32);
    return (u64)x1 * s1 + __umulhi(x0, s1) + __umulhi(x1, s0);
}
__device__ __forceinline__ u64 lazy_trunc(u64 x, u64 w, u64 ws, u64 nq) { return x * w + mulhi_trunc(x, ws) * nq; }
__device__ __forceinline__ u64 pmfold60(u64 x, u32 c) { return (u64)((u32)(x >> 32) >> 28) * c + (x & ((1ull << 60) - 1)); }
__device__ __forceinline__ u64 csub(u64 x, u64 nc) { const u64 t = x + nc; return (long long)t < 0 ? x : t; }
template <int MODE> __global__ void __launch_bounds__(256) k(u64 *out, int iters, u64 q, u64 w0, u64 ws0, u32 c, u64 nq)
{
    u64 v[8];
    const u64 q2 = q << 1, q4 = q << 2, nq8 = nq << 3;
    for (int i = 0; i < 8; i++) v[i] = (u64)(threadIdx.x * 977 + i * 131 + blockIdx.x) * 0x9E3779B97F4A7C15ull >> 5;
    u64 w = w0 + threadIdx.x, ws = ws0 + threadIdx.x;
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
                    if (MODE == 0) {
                        if (u == 0) x = csub(x, nq8);
                        const u64 y = lazy_exact(v[k1], w + b, ws + b, nq);
                        v[k0] = x + y;
                        v[k1] = x + q2 - y;
                    } else {
                        if (u == 0) x = pmfold60(x, c);
                        const u64 y = lazy_trunc(v[k1], w + b, ws + b, nq);
                        v[k0] = x + y;
                        v[k1] = x + q4 - y;
                    }
                }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; i++) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(int wps)
{
    int blocks = 256 * wps;
    u64 *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 4000;
    const u64 q = 1152921504595968001ULL;
    for (int rep = 0; rep < 2; rep++) k<MODE><<<blocks, 256>>>(out, iters, q, 88651361085495ULL, 123456789ULL, (u32)((1ull << 60) - q), 0 - q);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, q, 88651361085495ULL, 123456789ULL, (u32)((1ull << 60) - q), 0 - q);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double bf = (double)blocks * 256 * iters * 12;
    printf("mode %d waves/SIMD=%d: %.3f ms, %.3e butterflies/s => %.4f us per 2^15-point transform (butterflies only)\n", MODE, wps, ms, bf / (ms * 1e-3),
           245760.0 / (bf / (ms * 1e-3)) * 1e6);
    (void)hipFree(out);
}
int main()
{
    for (int w : {2, 4, 8}) { run<0>(w); run<1>(w); run<0>(w); run<1>(w); }
    return 0;
}
