// ubench_intmul.hip -- measures issue rates of the integer-multiply instructions the modular
// arithmetic is built from (gfx950).  Prints cycles per wave-instruction per SIMD.
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
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = 1.0 + i + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
        }
    }
    long long t1 = clock64();
    u64 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (u64)d[i] + x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[gridDim.x * 256 + blockIdx.x] = (u64)(t1 - t0);
}
template <int OP> void run(const char *name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block; blocks/CU = waves_per_simd
    u64 *out;
    hipMalloc(&out, (blocks * 256 + blocks) * 8);
    int iters = 4096;
    k<OP><<<blocks, 256>>>(out, iters, 12345);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 12345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    u64 cyc;
    hipMemcpy(&cyc, out + blocks * 256, 8, hipMemcpyDeviceToHost);
    double insts = (double)iters * 8;
    // per SIMD: waves_per_simd waves each issuing `insts` instructions in `cyc` cycles
    printf("%-18s waves/SIMD=%d  wall=%.3f ms  clock64 cycles/wave-inst (alone)=%.2f  => cycles per inst per SIMD=%.2f\n", name, waves_per_simd, ms,
           (double)cyc / insts, (double)cyc / insts / waves_per_simd);
    hipFree(out);
}
int main()
{
    for (int w : {1, 8}) {
        if (w == 1) {
            run<0>("v_mad_u64_u32", 1); run<1>("v_mul_lo_u32", 1); run<2>("v_mul_hi_u32", 1); run<3>("v_mul_u32_u24", 1);
            run<4>("v_mad_u32_u24", 1); run<8>("v_mul_hi_u32_u24", 1); run<5>("v_fma_f64", 1); run<6>("v_add_u32", 1); run<7>("v_lshl_add_u64", 1);
        } else {
            run<0>("v_mad_u64_u32", 8); run<1>("v_mul_lo_u32", 8); run<2>("v_mul_hi_u32", 8); run<3>("v_mul_u32_u24", 8);
            run<4>("v_mad_u32_u24", 8); run<8>("v_mul_hi_u32_u24", 8); run<5>("v_fma_f64", 8); run<6>("v_add_u32", 8); run<7>("v_lshl_add_u64", 8);
        }
    }
    return 0;
}
