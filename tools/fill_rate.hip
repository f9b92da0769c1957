// Micro-benchmark: how fast can a kernel shaped like the camera kernel's store side write a 4K
// f64 framebuffer?  One pixel (three doubles, 24 bytes) per lane, a wave covering 64 consecutive
// pixels, nothing else: the HBM-write floor under the camera kernel's time.
// hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o tools/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fill_pixels(double *fb, uint32_t n_pix, double v)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pix) return;
    double *px = fb + (size_t)k * 3;
    px[0] = v; px[1] = v + 1.0; px[2] = v + 2.0;
}
__global__ void __launch_bounds__(256) k_fill_wide(double2 *fb, size_t n2, double v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n2; k += stride) fb[k] = make_double2(v, v);
}

int main()
{
    const uint32_t n_pix = 3840u * 2160u;
    const size_t bytes = (size_t)n_pix * 24;
    double *fb;
    CK(hipMalloc((void **)&fb, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        const int reps = 200;
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_fill_pixels, dim3((n_pix + 255) / 256), dim3(256), 0, 0, fb, n_pix, (double)i);
                else hipLaunchKernelGGL(k_fill_wide, dim3(256 * 8), dim3(256), 0, 0, (double2 *)fb, bytes / 16, (double)i);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.4f ms per 4K f64 frame (%.0f MB), %.0f GB/s\n", variant == 0 ? "one pixel per lane, 3 x 8-byte stores" : "grid-stride 16-byte stores",
               ms / reps, bytes / 1e6, bytes / (ms / reps * 1e-3) / 1e9);
    }
    CK(hipFree(fb));
    return 0;
}
