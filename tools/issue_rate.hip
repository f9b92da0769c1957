// Micro-benchmark: what one SIMD of an MI355X issues per cycle when f64 vector instructions are
// mixed with scalar instructions and with small divergent regions, as a function of waves per
// SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/issue_rate.hip -o tools/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define SALU(n) asm volatile("s_add_u32 %0, %0, " #n : "+s"(sc) : : "scc")
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double s0, double s1, int iters, double thr)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t sc = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#define G(x) x = x * s0; x = x + s1
        if (MODE == 0) {         // 16 vector instructions
            G(a0); G(a1); G(a2); G(a3); G(a4); G(a5); G(a6); G(a7);
        } else if (MODE == 1) {  // 16 vector + 16 scalar, interleaved
            a0 = a0 * s0; SALU(1); a0 = a0 + s1; SALU(2); a1 = a1 * s0; SALU(3); a1 = a1 + s1; SALU(4);
            a2 = a2 * s0; SALU(1); a2 = a2 + s1; SALU(2); a3 = a3 * s0; SALU(3); a3 = a3 + s1; SALU(4);
            a4 = a4 * s0; SALU(1); a4 = a4 + s1; SALU(2); a5 = a5 * s0; SALU(3); a5 = a5 + s1; SALU(4);
            a6 = a6 * s0; SALU(1); a6 = a6 + s1; SALU(2); a7 = a7 * s0; SALU(3); a7 = a7 + s1; SALU(4);
        } else if (MODE == 2) {  // 16 vector + 8 scalar
            G(a0); SALU(1); G(a1); SALU(2); G(a2); SALU(3); G(a3); SALU(4);
            G(a4); SALU(1); G(a5); SALU(2); G(a6); SALU(3); G(a7); SALU(4);
        } else if (MODE == 3) {  // 16 vector + 4 small divergent regions (compare, exec mask, branch, 2 vector, restore)
            G(a0); if (a0 > thr + threadIdx.x) { a1 = a1 * s1; a1 = a1 + s0; }
            G(a2); if (a2 > thr + threadIdx.x) { a3 = a3 * s1; a3 = a3 + s0; }
            G(a4); if (a4 > thr + threadIdx.x) { a5 = a5 * s1; a5 = a5 + s0; }
            G(a6); if (a6 > thr + threadIdx.x) { a7 = a7 * s1; a7 = a7 + s0; }
            G(a1); G(a3); G(a5); G(a7);
        } else if (MODE == 4) {  // the same with selects instead of regions
            G(a0); { const bool c = a0 > thr + threadIdx.x; const double t = a1 * s1 + s0; a1 = c ? t : a1; }
            G(a2); { const bool c = a2 > thr + threadIdx.x; const double t = a3 * s1 + s0; a3 = c ? t : a3; }
            G(a4); { const bool c = a4 > thr + threadIdx.x; const double t = a5 * s1 + s0; a5 = c ? t : a5; }
            G(a6); { const bool c = a6 > thr + threadIdx.x; const double t = a7 * s1 + s0; a7 = c ? t : a7; }
            G(a1); G(a3); G(a5); G(a7);
        } else if (MODE == 5) {  // one dependent chain of 16
            G(a0); G(a0); G(a0); G(a0); G(a0); G(a0); G(a0); G(a0);
        } else if (MODE == 6) {  // 16 fused multiply-adds, three vector-register operands each
            a0 = __builtin_fma(a0, a1, a2); a1 = __builtin_fma(a1, a2, a3); a2 = __builtin_fma(a2, a3, a4); a3 = __builtin_fma(a3, a4, a5);
            a4 = __builtin_fma(a4, a5, a6); a5 = __builtin_fma(a5, a6, a7); a6 = __builtin_fma(a6, a7, a0); a7 = __builtin_fma(a7, a0, a1);
            a0 = __builtin_fma(a0, a1, a2); a1 = __builtin_fma(a1, a2, a3); a2 = __builtin_fma(a2, a3, a4); a3 = __builtin_fma(a3, a4, a5);
            a4 = __builtin_fma(a4, a5, a6); a5 = __builtin_fma(a5, a6, a7); a6 = __builtin_fma(a6, a7, a0); a7 = __builtin_fma(a7, a0, a1);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sc;
}

template <int MODE> void run(const char *name, int vector_per_iter)
{
    double *out;
    hipMalloc(&out, 256 * 256 * 64 * sizeof(double));
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = 256 * blocks_per_cu;  // 256 threads = 4 waves = 1 wave per SIMD per block
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters, 1e300);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters, MODE == 3 || MODE == 4 ? 100.0 : 1e300);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)grid * 4 * iters * vector_per_iter;
        printf("%-40s waves/SIMD %d: %.3f ms, %.2f SIMD-cycles per vector instruction (at 2.4 GHz)\n", name, blocks_per_cu, ms,
               ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
        fflush(stdout);
    }
    hipFree(out);
}

int main()
{
    run<0>("16 vector", 16);
    run<1>("16 vector + 16 scalar", 16);
    run<2>("16 vector + 8 scalar", 16);
    run<3>("24 vector in 4 divergent regions", 24);
    run<4>("28 vector with 4 selects", 28);
    run<5>("16 vector, one chain", 16);
    run<6>("16 fma, 3 vector operands", 16);
    return 0;
}
