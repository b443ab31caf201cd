// Micro-benchmark 2: issue cost of the instruction kinds and MIXES of the traversal loop (gfx950), at 4 / 6 / 8 waves per SIMD.
// Every kernel runs REP trips of a block of 8 independent instructions of one kind (or a VALU/SALU mix); the result is
// SIMD-cycles per wave-instruction = elapsed s_memtime ticks / (instructions per wave * waves on the SIMD) and, for reference, the
// wall-clock rate from hipEvents (instructions per ns per SIMD).  hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 8192
template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, unsigned seed) {
    float f0 = threadIdx.x * 1e-3f + 1.0f + seed, f1 = f0 + 0.1f, f2 = f0 + 0.2f, f3 = f0 + 0.3f, f4 = f0 + 0.4f, f5 = f0 + 0.5f, f6 = f0 + 0.6f, f7 = f0 + 0.7f;
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned long long m0 = 0, m1 = 0;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f4, f5 }, p3 = { f6, f7 };
    const f2v pc = { 1.0000001f, 0.9999999f }, pd = { 1e-9f, -1e-9f };
    const float c = 1.0000001f, d = 1e-9f;
    for (int i = 0; i < REP; i++) {
        if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));
        if (KIND == 1) asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));
        if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));
        if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c), "v"(d));
        if (KIND == 4) asm volatile("v_cmp_le_f32 %8, %0, %1\n v_cmp_le_f32 %9, %2, %3\n v_cmp_le_f32 %8, %4, %5\n v_cmp_le_f32 %9, %6, %7\n v_cmp_le_f32 %8, %1, %2\n v_cmp_le_f32 %9, %3, %4\n v_cmp_le_f32 %8, %5, %6\n v_cmp_le_f32 %9, %7, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+s"(m0), "+s"(m1));
        if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %9\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %9\n v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %9\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %9" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(0x5555555555555555ull), "s"(0x3333333333333333ull));
        if (KIND == 6) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
        if (KIND == 7) asm volatile("v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %9\n v_cvt_f32_u32 %2, %10\n v_cvt_f32_u32 %3, %11\n v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %10\n v_cvt_f32_u32 %7, %11" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        // VALU : SALU 1 : 1 -- does a scalar instruction of the same wave cost a vector issue slot?
        if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %10\n s_and_b64 %8, %8, %9\n v_mul_f32 %1, %1, %10\n s_or_b64 %9, %9, %8\n v_mul_f32 %2, %2, %10\n s_and_b64 %8, %8, %9\n v_mul_f32 %3, %3, %10\n s_or_b64 %9, %9, %8\n v_mul_f32 %4, %4, %10\n s_and_b64 %8, %8, %9\n v_mul_f32 %5, %5, %10\n s_or_b64 %9, %9, %8\n v_mul_f32 %6, %6, %10\n s_and_b64 %8, %8, %9\n v_mul_f32 %7, %7, %10\n s_or_b64 %9, %9, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+s"(m0), "+s"(m1) : "v"(c) : "scc");
        // a DEPENDENT chain of 8 (latency of one wave's back-to-back dependent VALU)
        if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(c));
        if (KIND == 12) asm volatile("v_perm_b32 %0, %0, %0, %4\n v_perm_b32 %1, %1, %1, %4\n v_perm_b32 %2, %2, %2, %4\n v_perm_b32 %3, %3, %3, %4\n v_perm_b32 %0, %0, %0, %4\n v_perm_b32 %1, %1, %1, %4\n v_perm_b32 %2, %2, %2, %4\n v_perm_b32 %3, %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x01000302u));
        if (KIND == 13) asm volatile("v_alignbit_b32 %0, %0, %0, %4\n v_alignbit_b32 %1, %1, %1, %4\n v_alignbit_b32 %2, %2, %2, %4\n v_alignbit_b32 %3, %3, %3, %4\n v_alignbit_b32 %0, %0, %0, %4\n v_alignbit_b32 %1, %1, %1, %4\n v_alignbit_b32 %2, %2, %2, %4\n v_alignbit_b32 %3, %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(16u));
        if (KIND == 14) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c), "v"(d));
        if (KIND == 15) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(d));
        if (KIND == 16) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(m0), "+v"(m1) : "v"(a0), "v"(747796405u) : "vcc");
        if (KIND == 17) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
        if (KIND == 18) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_le_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_le_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : : "vcc");
        if (KIND == 19) asm volatile("v_bfe_i32 %0, %0, 0, 16\n v_lshl_add_u32 %1, %1, 5, %0\n v_add3_u32 %2, %2, %1, %0\n v_and_b32 %3, %3, %2\n v_bfe_i32 %0, %0, 0, 16\n v_lshl_add_u32 %1, %1, 5, %0\n v_add3_u32 %2, %2, %1, %0\n v_and_b32 %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (KIND == 20) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
        if (KIND == 21) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));
        if (KIND == 10) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(747796405u));
        if (KIND == 11) asm volatile("v_lshrrev_b32 %0, 3, %0\n v_add_u32 %1, %1, %0\n v_xor_b32 %2, %2, %1\n v_lshrrev_b32 %3, 5, %3\n v_add_u32 %0, %0, %3\n v_xor_b32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(a0 ^ a1 ^ a2 ^ a3) + (float)(m0 ^ m1) + p0.x + p1.y + p2.x + p3.y;
}

template <int KIND>
static void run(const char *name, int blocks_per_cu, int waves_per_block_simd, int insts_per_trip = 8) {
    const int blocks = 256 * blocks_per_cu, threads = 256 * waves_per_block_simd;
    float *out; hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 2u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)insts_per_trip * REP * (blocks * (threads / 64));     // wave-instructions of this launch
    printf("%-22s waves/SIMD=%d  %.3f ms  %.3f wave-inst/ns/SIMD  (=> %.2f SIMD cycles per wave-instruction at 2.4 GHz)\n", name,
           blocks_per_cu * waves_per_block_simd, ms, wave_insts / (ms * 1e6) / 1024.0, 2.4 / (wave_insts / (ms * 1e6) / 1024.0));
    fflush(stdout);
    hipFree(out);
}

int main() {
    for (int b : { 2 }) for (int w : { 3 }) {
        run<0>("v_mul_f32", b, w); run<1>("v_min/max_f32", b, w); run<2>("v_min3/max3_f32", b, w); run<3>("v_fma_mix_f32", b, w);
        run<4>("v_cmp_le_f32 -> sgpr", b, w); run<5>("v_cndmask_b32 (sgpr)", b, w); run<6>("v_rcp_f32", b, w); run<7>("v_cvt_f32_u32", b, w);
        run<10>("v_mul_lo_u32", b, w); run<11>("shift/add/xor mix", b, w);
        run<12>("v_perm_b32", b, w); run<13>("v_alignbit_b32", b, w); run<14>("v_fma_f32", b, w); run<15>("v_add/sub_f32", b, w);
        run<16>("v_mad_u64_u32", b, w); run<17>("v_mov_b32", b, w); run<18>("v_cmp vcc + v_cndmask", b, w); run<19>("bfe/lshl_add/add3/and", b, w);
        run<20>("v_sqrt_f32", b, w); run<21>("v_pk_fma_f32 (2 fma)", b, w);
        run<8>("v_mul + s_and (1:1)", b, w, 16); run<9>("dependent v_mul chain", b, w);
    }
    return 0;
}
