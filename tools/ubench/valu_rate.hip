// tools/ubench/valu_rate.hip -- issue rate of the VALU forms k_step is made of, on the hardware (developer micro-benchmark; not
// part of the product).  Each case: every SIMD of the chip holds W waves that run the same unrolled stream of N independent
// instructions of one kind, ITER times; reported: shader-clock cycles per wave-instruction per SIMD (wall cycles of a wave's
// loop / instructions issued by all W waves of its SIMD) at W = 1, 2, 6.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(unsigned long long *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1;
    unsigned long long w0 = u0, w1 = u1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {          // v_fma_f32, 8 independent accumulators
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // v_pk_fma_f32, 4 independent accumulator pairs (2 fp32 fma per lane per instruction)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) {   // v_pk_mul_f32 / v_pk_add_f32 alternating
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 3) {   // v_mad_u64_u32 (Philox's 32 x 32 -> 64 multiply)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %3, %2, 0\n"
                              : "+v"(w0), "+v"(w1) : "v"(u0), "v"(u1) : "vcc");)
        } else if (KIND == 4) {   // transcendental: v_rsq_f32
            REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {   // v_alignbit_b32 / v_perm_b32 (VOP3 integer forms of the list build)
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 16\n v_perm_b32 %1, %1, %0, %2\n v_alignbit_b32 %0, %0, %1, 16\n v_perm_b32 %1, %1, %0, %2\n"
                              "v_alignbit_b32 %0, %0, %1, 16\n v_perm_b32 %1, %1, %0, %2\n v_alignbit_b32 %0, %0, %1, 16\n v_perm_b32 %1, %1, %0, %2\n"
                              : "+v"(u0), "+v"(u1) : "v"(0x01000706u));)
        } else if (KIND == 6) {   // v_add_f32 / v_mul_f32 (VOP2)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n"
                              "v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 7) {   // v_cvt_f32_f16 + v_fma_f32 pairs (the a/b unpack)
            REP8(asm volatile("v_cvt_f32_f16 %0, %1\n v_cvt_f32_f16 %2, %3\n v_cvt_f32_f16 %4, %5\n v_cvt_f32_f16 %6, %7\n v_cvt_f32_f16 %1, %0\n v_cvt_f32_f16 %3, %2\n v_cvt_f32_f16 %5, %4\n v_cvt_f32_f16 %7, %6\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1) + (float)(w0 + w1);
    if (s == 12345.678f) out[0] = 1;                    // (keeps every accumulator alive)
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cus)
{
    const int iters = 2000, per_iter = 64;
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * cus * 4 * 8);
    printf("%-34s", name);
    for (int W : {1, 2, 4, 6, 8}) {
        const int blocks = cus * 4 * W;                 // one-wave workgroups: the dispatcher spreads them over the SIMDs, W per SIMD
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
        // s_memtime ticks at a fixed 100 MHz on gfx9: convert through the wall time of the launch instead
        const double instr_per_simd = (double)iters * per_iter * W;
        printf("  W=%d: %6.2f ns/instr/SIMD (%5.1f us)", W, ms * 1e6 / instr_per_simd, ms * 1e3);
        (void)mean;
    }
    printf("\n");
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clock %d kHz => one cycle = %.3f ns\n", p.name, p.multiProcessorCount, p.clockRate, 1e6 / p.clockRate);
    const int cus = p.multiProcessorCount;
    run<0>("v_fma_f32", cus);
    run<6>("v_add_f32 / v_mul_f32", cus);
    run<1>("v_pk_fma_f32 (2 fma per lane)", cus);
    run<2>("v_pk_mul_f32 / v_pk_add_f32", cus);
    run<3>("v_mad_u64_u32", cus);
    run<4>("v_rsq_f32", cus);
    run<5>("v_alignbit_b32 / v_perm_b32", cus);
    run<7>("v_cvt_f32_f16", cus);
    return 0;
}
