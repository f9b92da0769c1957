// Micro-benchmark: what does a launch of very many very short waves cost, whatever they compute?
// The headline frame is 129 600 one-wave workgroups of a few hundred instructions each.  Variants:
// an empty kernel; a kernel that reads a 896-byte argument block and stores one pixel per lane;
// the same with a dependent chain of arithmetic in between; each for 1, 2 and 4 pixels per lane
// (fewer, longer waves) and for workgroups of 1 and 4 waves.
// hipcc --offload-arch=gfx950 -O3 tools/wave_rate.hip -o tools/wave_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Args { double v[108]; };  // 864 bytes + the rest: about what the camera kernel is given

template <typename T, int BLOCK, int PER_LANE, int CHAIN>
__global__ void __launch_bounds__(BLOCK) k_waves(Args a, T *fb, uint32_t n_pix, int mode)
{
    if (mode == 0) return;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int p = 0; p < PER_LANE; ++p) {
        const uint32_t k = (wave * PER_LANE + p) * 64u + lane;
        if (k >= n_pix) return;
        T x = (T)a.v[p] + (T)k, y = (T)a.v[100 + p], z = (T)a.v[50 + p];
#pragma unroll 8
        for (int i = 0; i < CHAIN; ++i) {  // dependent: x -> y -> z -> x
            x = x * y + z;
            y = y * z + x;
            z = z * x + y;
        }
        T *px = fb + (size_t)k * 3;
        px[0] = x; px[1] = y; px[2] = z;
    }
}

template <typename T, int BLOCK, int PER_LANE, int CHAIN>
int run(const char *name, T *fb, uint32_t n_pix, int mode)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args a;
    for (int i = 0; i < 108; ++i) a.v[i] = 1.0 / (i + 2);
    const uint32_t per_block = BLOCK * PER_LANE;
    const uint32_t grid = (n_pix + per_block - 1) / per_block;
    const int reps = 200;
    float ms = 0;
    for (int warm = 0; warm < 2; ++warm) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_waves<T, BLOCK, PER_LANE, CHAIN>), dim3(grid), dim3(BLOCK), 0, 0, a, fb, n_pix, mode);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-10s %s block %3d, %d px/lane, chain %3d x3: %7.2f us per launch (%u workgroups)\n", name, mode ? "store" : "empty", BLOCK, PER_LANE, CHAIN,
           ms / reps * 1e3, grid);
    return 0;
}

int main()
{
    const uint32_t n_pix = 3840u * 2160u;
    void *fb;
    CK(hipMalloc(&fb, (size_t)n_pix * 24));
    run<double, 64, 1, 0>("f64", (double *)fb, n_pix, 0);
    run<double, 256, 1, 0>("f64", (double *)fb, n_pix, 0);
    run<double, 64, 1, 0>("f64", (double *)fb, n_pix, 1);
    run<double, 64, 2, 0>("f64", (double *)fb, n_pix, 1);
    run<double, 64, 4, 0>("f64", (double *)fb, n_pix, 1);
    run<double, 256, 1, 0>("f64", (double *)fb, n_pix, 1);
    run<double, 64, 1, 40>("f64", (double *)fb, n_pix, 1);
    run<double, 64, 2, 40>("f64", (double *)fb, n_pix, 1);
    run<double, 64, 4, 40>("f64", (double *)fb, n_pix, 1);
    run<double, 256, 1, 40>("f64", (double *)fb, n_pix, 1);
    run<float, 64, 1, 0>("f32", (float *)fb, n_pix, 1);
    run<float, 64, 2, 0>("f32", (float *)fb, n_pix, 1);
    run<float, 64, 4, 0>("f32", (float *)fb, n_pix, 1);
    run<float, 256, 1, 0>("f32", (float *)fb, n_pix, 1);
    run<float, 64, 1, 40>("f32", (float *)fb, n_pix, 1);
    run<float, 64, 2, 40>("f32", (float *)fb, n_pix, 1);
    run<float, 64, 4, 40>("f32", (float *)fb, n_pix, 1);
    run<float, 256, 1, 40>("f32", (float *)fb, n_pix, 1);
    CK(hipFree(fb));
    return 0;
}
