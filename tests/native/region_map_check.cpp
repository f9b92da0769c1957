// Host-side check of the screen-region arithmetic of camera launches (rt_launch.h): for a range of grids and
// region counts, every workgroup of the launch lands in a region below the count, no region is given more
// workgroups than the capacity formula of the queue planner allows (cols x rows), the rows the launch renders
// are each rendered exactly once, and the multiply-shift quotients equal the divisions they stand for.
// Prints "ok <cases>" or the first failure.
#include <cstdio>
#include <vector>

#include "../../ray-tracer_amd/csrc/rt_launch.h"

using namespace rtx;

static uint32_t mul_shift(uint32_t v, uint32_t magic) { return magic ? (uint32_t)(((uint64_t)v * magic) >> 32) : v; }

int main()
{
    unsigned long long cases = 0;
    const uint32_t widths[] = { 1, 2, 3, 5, 8, 16, 17, 31, 60, 64, 100, 120, 257 };
    const uint32_t heights[] = { 1, 2, 3, 7, 10, 64, 135, 136, 270, 701, 1080, 2160, 4320, 9999, 64511 };
    for (uint32_t gx : widths)
        for (uint32_t gy : heights)
            for (uint32_t n_regions = 1; n_regions <= 1024; n_regions *= 2) {
                uint32_t nx, cols, rows;
                region_split(gx, gy, n_regions, &nx, &cols, &rows);
                const uint32_t ny = n_regions / nx;
                if (nx == 0 || nx * ny != n_regions || (nx & (nx - 1)) || cols == 0 || rows == 0) {
                    std::printf("bad split gx %u gy %u n %u: nx %u cols %u rows %u\n", gx, gy, n_regions, nx, cols, rows);
                    return 1;
                }
                if ((uint64_t)cols * nx < gx || (uint64_t)rows * ny < gy) {
                    std::printf("regions do not cover the grid: gx %u gy %u n %u\n", gx, gy, n_regions);
                    return 1;
                }
                uint32_t ny_shift = 0;
                while ((1u << ny_shift) < ny) ++ny_shift;
                const uint32_t magic = region_magic(cols);
                const uint64_t launch_rows = (uint64_t)ny * rows;
                if (launch_rows - gy >= ny || (gy <= kCameraMaxRows && launch_rows > 65535u)) {  // a grid numbers 65 535 rows
                    std::printf("launch has too many rows: gy %u ny %u rows %u\n", gy, ny, rows);
                    return 1;
                }
                std::vector<uint32_t> per_region(n_regions, 0);
                std::vector<uint8_t> seen(gy, 0);
                for (uint32_t by = 0; by < launch_rows; ++by) {
                    const uint32_t region_row = by & ((1u << ny_shift) - 1u);
                    const uint32_t row = region_row * rows + (by >> ny_shift);
                    if (row >= gy) continue;  // past the frame: nothing to render
                    if (seen[row]++) { std::printf("row %u rendered twice (gy %u ny %u rows %u)\n", row, gy, ny, rows); return 1; }
                    if (row / rows != region_row) { std::printf("row %u in the wrong row of regions\n", row); return 1; }
                    for (uint32_t bx = 0; bx < gx; ++bx) {
                        const uint32_t q = mul_shift(bx, magic);
                        if (q != bx / cols) { std::printf("quotient %u / %u: %u\n", bx, cols, q); return 1; }
                        const uint32_t region = region_row * nx + q;
                        if (region >= n_regions) { std::printf("region %u of %u\n", region, n_regions); return 1; }
                        ++per_region[region];
                    }
                }
                for (uint32_t r = 0; r < gy; ++r)
                    if (!seen[r]) { std::printf("row %u never rendered (gy %u ny %u rows %u)\n", r, gy, ny, rows); return 1; }
                for (uint32_t r = 0; r < n_regions; ++r)
                    if (per_region[r] > (uint64_t)cols * rows) { std::printf("region %u holds %u workgroups > %u x %u\n", r, per_region[r], cols, rows); return 1; }
                ++cases;
            }
    std::printf("ok %llu\n", cases);
    return 0;
}
