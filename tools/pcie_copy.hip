// How fast can a 4K f64 frame (199 MB) reach pageable host memory?
// hipcc --offload-arch=gfx950 -O3 -pthread tools/pcie_copy.hip -o tools/pcie_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)3840 * 2160 * 24;
    void *dev;
    CK(hipMalloc(&dev, bytes));
    CK(hipMemset(dev, 1, bytes));
    char *host = (char *)aligned_alloc(4096, bytes);
    memset(host, 0, bytes);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        CK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        printf("hipMemcpy to pageable: %.2f ms (%.1f GB/s)\n", now() - t, bytes / (now() - t) / 1e6);
    }
    {   // registering the caller's buffer is refused on some systems; report and go on
        hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
        if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
        else {
            double t1 = now();
            CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            printf("copy into registered memory: %.2f ms\n", now() - t1);
            CK(hipHostUnregister(host));
        }
    }
    for (int threads : { 1, 2, 4, 8 })
        for (size_t chunk : { (size_t)4 << 20, (size_t)16 << 20 }) {
            char *pin[2];
            CK(hipHostMalloc((void **)&pin[0], chunk));
            CK(hipHostMalloc((void **)&pin[1], chunk));
            hipEvent_t ev[2];
            CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t = now();
                const size_t n = (bytes + chunk - 1) / chunk;
                auto issue = [&](size_t i) {
                    const size_t len = std::min(chunk, bytes - i * chunk);
                    hipMemcpyAsync(pin[i & 1], (char *)dev + i * chunk, len, hipMemcpyDeviceToHost, s);
                    hipEventRecord(ev[i & 1], s);
                };
                issue(0);
                for (size_t i = 0; i < n; ++i) {
                    hipEventSynchronize(ev[i & 1]);
                    const size_t len = std::min(chunk, bytes - i * chunk);
                    // the buffer of chunk i+1 is the other one: its transfer overlaps this memcpy
                    if (i + 1 < n) issue(i + 1);
                    std::vector<std::thread> pool;
                    const size_t part = (len + threads - 1) / threads;
                    for (int k = 1; k < threads; ++k)
                        pool.emplace_back([&, k] { size_t a = k * part; if (a < len) memcpy(host + i * chunk + a, pin[i & 1] + a, std::min(part, len - a)); });
                    memcpy(host + i * chunk, pin[i & 1], std::min(part, len));
                    for (auto &th : pool) th.join();
                    // chunk i+2 reuses this buffer; it is issued in the next iteration, after this memcpy
                }
                best = std::min(best, now() - t);
            }
            printf("pinned staging, %d thread(s), %zu MiB chunks: %.2f ms (%.1f GB/s)\n", threads, chunk >> 20, best, bytes / best / 1e6);
            hipHostFree(pin[0]); hipHostFree(pin[1]);
        }
    return host[12345] == 1 ? 0 : 2;
}
