// Micro-benchmark: issue cost of the VALU instructions the path tracer leans on (gfx950).
// Each kernel runs a long dependent-free stream of one instruction kind on 8 independent accumulators;
// waves/SIMD is set from the host.  Prints cycles per wave-instruction (from s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096
template <int KIND>
__global__ void k(unsigned *out, unsigned long long *cyc, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    float f0 = a0 * 1e-3f + 1.0f, f1 = f0 + 0.1f, f2 = f0 + 0.2f, f3 = f0 + 0.3f, f4 = f0 + 0.4f, f5 = f0 + 0.5f, f6 = f0 + 0.6f, f7 = f0 + 0.7f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f4, f5 }, p3 = { f6, f7 };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        if (KIND == 0) { asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(747796405u)); }
        if (KIND == 1) { asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0000001f)); }
        if (KIND == 2) { asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f2v{ 1.0000001f, 0.9999999f })); }
        if (KIND == 3) { asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x9E3779B9u)); }
        if (KIND == 4) { asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
        if (KIND == 5) { asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(7477961u)); }
        if (KIND == 6) { asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1e-9f)); }
        if (KIND == 7) { asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(*(unsigned long long *)&p0), "+v"(*(unsigned long long *)&p1), "+v"(*(unsigned long long *)&p2), "+v"(*(unsigned long long *)&p3) : "v"(a0), "v"(747796405u) : "vcc"); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.x + p2.y + p3.y);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd;   // one block per CU, 4 * wps waves
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(unsigned) * blocks * threads); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= blocks;
    // s_memtime ticks at a fixed 100 MHz on gfx9; convert with the measured kernel clock is not possible here,
    // so report ticks per (8*REP) instructions per wave and the ratio to v_mul_f32 printed alongside.
    printf("%-16s waves/SIMD=%d  ticks=%10.0f  ticks/inst/wave=%.5f\n", name, waves_per_simd, m, m / (8.0 * REP) / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int wps : { 1, 2, 4 }) {
        run<1>("v_mul_f32", wps); run<6>("v_fma_f32", wps); run<2>("v_pk_mul_f32", wps); run<3>("v_xor_b32", wps);
        run<0>("v_mul_lo_u32", wps); run<5>("v_mul_u32_u24", wps); run<7>("v_mad_u64_u32", wps); run<4>("v_sqrt_f32", wps);
    }
    return 0;
}
