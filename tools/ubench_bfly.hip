// ubench_bfly.hip -- register-only ceiling of the 64-bit Shoup (Harvey) butterfly on gfx950: each lane runs radix-16
// rounds on 16 words held in registers with per-lane twiddles; no memory traffic inside the loop.
// Prints butterflies per second chip-wide for several occupancies -> the VALU ceiling of a 2^15-point transform.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
__device__ __forceinline__ u64 shoup_lazy(u64 x, u64 w, u64 ws, u64 q) { return x * w - __umul64hi(x, ws) * q; }
template <int MODE> __global__ void __launch_bounds__(256) k(u64 *out, int iters, u64 q, u64 w0, u64 ws0)
{
    u64 v[16];
    const u64 q2 = q << 1, q8 = q << 3;
    for (int i = 0; i < 16; i++) v[i] = (threadIdx.x * 977 + i * 131 + blockIdx.x) % q;
    u64 w = w0 + threadIdx.x, ws = ws0 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int half = 8 >> u;
#pragma unroll
            for (int b = 0; b < (1 << u); b++) {
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int k0 = b * 2 * half + j, k1 = k0 + half;
                    u64 x = v[k0];
                    if (MODE == 0) x -= (x >= q2) ? q2 : 0;
                    else if (u == 0) x -= (x >= q8) ? q8 : 0;
                    const u64 y = shoup_lazy(v[k1], w + b, ws + b, q);
                    v[k0] = x + y;
                    v[k1] = x + q2 - y;
                }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 16; i++) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(int wps)
{
    int blocks = 256 * wps;
    u64 *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 2000;
    const u64 q = 1152921504595968001ULL;
    k<MODE><<<blocks, 256>>>(out, iters, q, 88651361085495ULL, 123456789ULL);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, q, 88651361085495ULL, 123456789ULL);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double bf = (double)blocks * 256 * iters * 32;
    double per_ntt = 245760.0;  // butterflies of one 2^15-point transform
    printf("mode %d waves/SIMD=%d: %.3f ms, %.3e butterflies/s, => %.4f us per 2^15-point transform (butterflies only)\n", MODE, wps, ms, bf / (ms * 1e-3),
           per_ntt / (bf / (ms * 1e-3)) * 1e6);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<0>(w); run<1>(w); }
    return 0;
}
