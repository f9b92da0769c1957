// Micro-benchmark: rate of (a) returning u32 atomics from one lane per wave on 1..N counters,
// (b) non-returning f64 atomic adds scattered over a framebuffer-sized array.
// hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o tools/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_counter(uint32_t *counters, uint32_t n_counters, uint32_t stride_words, uint32_t *sink, int per_block)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t base = 0;
    if (per_block) {
        __shared__ uint32_t sh;
        if (threadIdx.x == 0) sh = atomicAdd(&counters[(blockIdx.x % n_counters) * stride_words], 256u);
        __syncthreads();
        base = sh;
    } else {
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&counters[(wave % n_counters) * stride_words], 64u);
        base = __shfl(base, 0);
    }
    if (base == 0xffffffffu) sink[threadIdx.x] = base;
}

__global__ void __launch_bounds__(256) k_scatter(double *fb, uint32_t n_pix, uint32_t n, int coherent_rows)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    // pseudo-random pixel, or a pixel near k (neighbouring lanes -> neighbouring pixels)
    uint32_t p = coherent_rows ? (k % n_pix) : (uint32_t)(((uint64_t)k * 2654435761ull) % n_pix);
    double *px = fb + (size_t)p * 3;
    unsafeAtomicAdd(px + 0, 1.0);
    unsafeAtomicAdd(px + 1, 2.0);
    unsafeAtomicAdd(px + 2, 3.0);
}

int main()
{
    uint32_t *counters, *sink;
    CK(hipMalloc(&counters, 1 << 20));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(counters, 0, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = 32768;  // 131072 waves
    for (int per_block = 0; per_block < 2; ++per_block)
        for (uint32_t nc : { 1u, 8u, 64u, 256u, 1024u }) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_counter, dim3(blocks), dim3(256), 0, 0, counters, nc, 32u, sink, per_block);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double n_at = per_block ? blocks : blocks * 4.0;
            printf("counter atomics: %s, %4u counters: %.3f ms for %.0f atomics = %.2f ns each\n", per_block ? "per block" : "per wave ", nc, ms, n_at, ms * 1e6 / n_at);
        }
    const uint32_t n_pix = 3840 * 2160;
    double *fb;
    CK(hipMalloc(&fb, (size_t)n_pix * 24));
    CK(hipMemset(fb, 0, (size_t)n_pix * 24));
    for (int coh = 0; coh < 2; ++coh)
        for (uint32_t n : { 1u << 20, 1u << 22, 1u << 23 }) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, 0, fb, n_pix, n, coh);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("f64 fb atomics (%s): %u rays x3: %.3f ms = %.2f G atomic/s\n", coh ? "coherent" : "scattered", n, ms, 3.0 * n / ms * 1e-6);
        }
    return 0;
}
