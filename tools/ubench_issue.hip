// ubench_issue.hip -- VALU issue rate of the integer instructions the modular arithmetic is built from, with the clock
// measured inside the kernel (s_memtime = shader cycles, s_memrealtime = 100 MHz) so that DVFS cannot distort the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int OP> __global__ void __launch_bounds__(256) k(u64 *out, int iters, u64 seed)
{
    u64 a[8];
    u32 x[8];
    u32 b = (u32)seed + threadIdx.x, c = (u32)(seed >> 32) | 1;
    for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + threadIdx.x; x[i] = (u32)a[i]; }
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 5) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u64 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { out[gridDim.x * 256 + 2 * blockIdx.x] = t1 - t0; out[gridDim.x * 256 + 2 * blockIdx.x + 1] = r1 - r0; }
}
template <int OP> void run(const char *name, int wps)
{
    int blocks = 256 * wps;
    u64 *out;
    (void)hipMalloc(&out, ((size_t)blocks * 256 + 2 * blocks) * 8);
    const int iters = 100000;
    for (int rep = 0; rep < 2; rep++) k<OP><<<blocks, 256>>>(out, iters, 12345);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 12345);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    u64 tr[2];
    (void)hipMemcpy(tr, out + (size_t)blocks * 256, 16, hipMemcpyDeviceToHost);
    const double insts = (double)iters * 8;
    const double ghz = (double)tr[0] / (double)tr[1] * 0.1;
    printf("%-16s waves/SIMD=%d  memtime/memrealtime*0.1 = %.2f  memtime ticks per wave-instruction per SIMD = %.2f  wall %.3f ms -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n",
           name, wps, ghz, (double)tr[0] / insts / wps, ms, ms * 1e6 / insts / wps, ms * 1e6 / insts / wps * 2.4);
    (void)hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<3>("v_add_u32", w); run<5>("v_add3_u32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_hi_u32", w); run<0>("v_mad_u64_u32", w); run<4>("v_lshl_add_u64", w);
    }
    return 0;
}
