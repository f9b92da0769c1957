// Micro-benchmark: sustained issue rate of f64 / f32 vector instructions on one MI355X,
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, double s0, double s1, int iters)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // independent fma
#define F(x) x = __builtin_fma(x, s0, s1)
            F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
            F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
        } else if (MODE == 1) {  // mul + add pairs (no contraction)
#define G(x) x = x * s0; x = x + s1
            G(a0); G(a1); G(a2); G(a3); G(a4); G(a5); G(a6); G(a7);
        } else if (MODE == 2) {  // dependent chain, mul/add
            a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1;
            a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1; a0 = a0 * s0; a0 = a0 + s1;
        } else if (MODE == 3) {  // two chains
            a0 = a0 * s0; a1 = a1 * s0; a0 = a0 + s1; a1 = a1 + s1; a0 = a0 * s0; a1 = a1 * s0; a0 = a0 + s1; a1 = a1 + s1;
            a0 = a0 * s0; a1 = a1 * s0; a0 = a0 + s1; a1 = a1 + s1; a0 = a0 * s0; a1 = a1 * s0; a0 = a0 + s1; a1 = a1 + s1;
        } else if (MODE == 4) {  // divisions
            a0 = a0 / a1 + s1; a2 = a2 / a3 + s1; a4 = a4 / a5 + s1; a6 = a6 / a7 + s1;
        } else if (MODE == 5) {  // sqrt
            a0 = __builtin_sqrt(a0) + s1; a1 = __builtin_sqrt(a1) + s1; a2 = __builtin_sqrt(a2) + s1; a3 = __builtin_sqrt(a3) + s1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name, int ops_per_iter)
{
    double *out;
    hipMalloc(&out, 256 * 256 * 64 * sizeof(double));
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = 256 * blocks_per_cu;  // 256 threads = 4 waves = 1 wave per SIMD per block
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)grid * 4 * iters * ops_per_iter;
        double cyc_per_instr_per_simd = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
        printf("%-22s waves/SIMD %d: %.3f ms, %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, blocks_per_cu, ms,
               cyc_per_instr_per_simd);
    }
    hipFree(out);
}

int main()
{
    run<0>("fma_f64 independent", 16);
    run<1>("mul+add f64 indep", 16);
    run<2>("mul/add f64 1 chain", 16);
    run<3>("mul/add f64 2 chains", 16);
    run<4>("div f64 (4/iter)+add", 4);
    run<5>("sqrt f64 (4/iter)+add", 4);
    return 0;
}
