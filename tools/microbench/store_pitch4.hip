// The four planes of the SAD surface (rows of 16 KiB, 4 KiB, 2 KiB, 512 B per workgroup and step) by row padding.
// Build: hipcc --offload-arch=gfx950 -O3 store_pitch4.hip -o store_pitch4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int kRows = 129, kCtus = 510;
struct Pitch { size_t p8, p16, p32, p64; };

__global__ void __launch_bounds__(512) store_kernel(char *o8, char *o16, char *o32, char *o64, Pitch p, u64 seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * kRows;
    const u64 v = seed + tid;
    for (int dy = 0; dy < kRows; dy++) {
        const size_t r = row0 + dy;
        char *r8 = o8 + r * p.p8 + wave * 2048 + lane * 8;
        *(u64 *)(r8) = v;
        *(u64 *)(r8 + 512) = v;
        *(u64 *)(r8 + 1024) = v;
        *(u64 *)(r8 + 1536) = v;
        *(u64 *)(o16 + r * p.p16 + wave * 512 + lane * 8) = v;
        *(u32 *)(o32 + r * p.p32 + wave * 256 + lane * 4) = (u32)v;
        if (lane < 16) *(u32 *)(o64 + r * p.p64 + wave * 64 + lane * 4) = (u32)v;
    }
}

int main()
{
    const size_t rows = (size_t)kCtus * kRows;
    char *o8, *o16, *o32, *o64;
    (void)hipMalloc((void **)&o8, rows * 32768), (void)hipMalloc((void **)&o16, rows * 8192), (void)hipMalloc((void **)&o32, rows * 4096), (void)hipMalloc((void **)&o64, rows * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const Pitch ps[] = { { 16384, 4096, 2048, 512 }, { 16896, 4096, 2048, 512 }, { 16896, 4224, 2048, 512 }, { 16896, 4224, 2176, 512 }, { 16896, 4224, 2176, 640 },
                         { 16896, 4608, 2304, 640 }, { 16512, 4224, 2176, 640 }, { 17408, 4352, 2176, 544 } };
    for (const Pitch &p : ps)
        for (int rep = 0; rep < 2; rep++) {
            const int n = 150;
            (void)hipEventRecord(e0);
            for (int i = 0; i < n; i++) store_kernel<<<kCtus, 512>>>(o8, o16, o32, o64, p, i);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("pitches %zu %zu %zu %zu: %.1f us per launch, %.2f TB/s\n", p.p8, p.p16, p.p32, p.p64, ms / n * 1e3, (double)rows * 23040 / (ms / n * 1e-3) / 1e12);
        }
    return 0;
}
