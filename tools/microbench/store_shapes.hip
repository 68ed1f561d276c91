// Streaming-store rate of the SAD surface's output pattern (510 workgroups x 8 wavefronts, 129 rows each, four planes)
// with different bytes per lane. Build: hipcc --offload-arch=gfx950 -O3 store_shapes.hip -o store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kRows = 129, kCtus = 510;
constexpr size_t kRow8 = 16384, kRow16 = 4096, kRow32 = 2048, kRow64 = 512;

// MODE 0: the kernel's stores today (4 x 8 B, 8 B, 4 B, 4 B on a quarter of the lanes)
// MODE 1: 16-B stores for the two 16-bit planes (2 x 16 B, half the lanes 16 B), 4-B planes as today
// MODE 2: everything as 16-B stores of the same total size (the planes as one 23 040-byte row)
// MODE 3: as 0, but a workgroup writes 8 consecutive rows per plane before moving on (row-major bursts of 128 KiB)
template <int MODE>
__global__ void __launch_bounds__(512) store_kernel(char *o8, char *o16, char *o32, char *o64, u64 seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * kRows;
    const u64 v = seed + tid;
    for (int dy = 0; dy < kRows; dy++) {
        const size_t r = row0 + dy;
        if (MODE == 0) {
            char *r8 = o8 + r * kRow8 + wave * 2048 + lane * 8;
            *(u64 *)(r8) = v;
            *(u64 *)(r8 + 512) = v;
            *(u64 *)(r8 + 1024) = v;
            *(u64 *)(r8 + 1536) = v;
            *(u64 *)(o16 + r * kRow16 + wave * 512 + lane * 8) = v;
            *(u32 *)(o32 + r * kRow32 + wave * 256 + lane * 4) = (u32)v;
            if (lane < 16) *(u32 *)(o64 + r * kRow64 + wave * 64 + lane * 4) = (u32)v;
        } else if (MODE == 1) {
            char *r8 = o8 + r * kRow8 + wave * 2048 + lane * 16;
            const uint4 w = { (u32)v, (u32)(v >> 32), (u32)v, (u32)(v >> 32) };
            *(uint4 *)(r8) = w;
            *(uint4 *)(r8 + 1024) = w;
            if (lane < 32) *(uint4 *)(o16 + r * kRow16 + wave * 512 + lane * 16) = w;
            *(u32 *)(o32 + r * kRow32 + wave * 256 + lane * 4) = (u32)v;
            if (lane < 16) *(u32 *)(o64 + r * kRow64 + wave * 64 + lane * 4) = (u32)v;
        } else if (MODE >= 3 && MODE <= 6) {
            // 3: the 16 KiB plane alone; 4: + the 4 KiB plane; 5: + the 2 KiB plane; 6: all four, but as ONE row of 23 040 bytes
            // per (workgroup, dy) with today's bytes per lane
            char *r8 = (MODE == 6 ? o8 + r * 23040 : o8 + r * kRow8) + wave * 2048 + lane * 8;
            *(u64 *)(r8) = v;
            *(u64 *)(r8 + 512) = v;
            *(u64 *)(r8 + 1024) = v;
            *(u64 *)(r8 + 1536) = v;
            if (MODE == 6) {
                char *rr = o8 + r * 23040 + 16384;
                *(u64 *)(rr + wave * 512 + lane * 8) = v;
                *(u32 *)(rr + 4096 + wave * 256 + lane * 4) = (u32)v;
                if (lane < 16) *(u32 *)(rr + 6144 + wave * 64 + lane * 4) = (u32)v;
            } else {
                if (MODE >= 4) *(u64 *)(o16 + r * kRow16 + wave * 512 + lane * 8) = v;
                if (MODE >= 5) *(u32 *)(o32 + r * kRow32 + wave * 256 + lane * 4) = (u32)v;
            }
        } else if (MODE == 2) {
            char *rr = o8 + r * 23040;
            const uint4 w = { (u32)v, (u32)(v >> 32), (u32)v, (u32)(v >> 32) };
            for (int k = tid; k < 23040 / 16; k += 512) *(uint4 *)(rr + k * 16) = w;
        }
    }
}

int main()
{
    const size_t rows = (size_t)kCtus * kRows;
    char *o8, *o16, *o32, *o64;
    hipMalloc((void **)&o8, rows * 23040 + 4096);
    hipMalloc((void **)&o16, rows * kRow16);
    hipMalloc((void **)&o32, rows * kRow32);
    hipMalloc((void **)&o64, rows * kRow64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const double bytes = (double)rows * 23040;
    const double mode_bytes[7] = { 23040, 23040, 23040, 16384, 20480, 22528, 23040 };
    for (int mode = 0; mode < 7; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            const int n = rep == 0 ? 100 : 200;      // the first burst warms the clocks up
            hipEventRecord(e0);
            for (int i = 0; i < n; i++) {
                if (mode == 0) store_kernel<0><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 1) store_kernel<1><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 2) store_kernel<2><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 3) store_kernel<3><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 4) store_kernel<4><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 5) store_kernel<5><<<kCtus, 512>>>(o8, o16, o32, o64, i);
                if (mode == 6) store_kernel<6><<<kCtus, 512>>>(o8, o16, o32, o64, i);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.1f us per launch, %.2f TB/s\n", mode, ms / n * 1e3, (double)rows * mode_bytes[mode] / (ms / n * 1e-3) / 1e12);
        }
    }
    return 0;
}
