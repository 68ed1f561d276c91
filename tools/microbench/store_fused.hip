// The SAD surface's output as ONE stream per (CTU, dy) row instead of four planes: the 16 KiB + 4 KiB + 2 KiB + 512 B of a
// row back to back (23 040 bytes), rows padded off the power-of-two pitch. Same grid, same bytes as store_pitch4.hip.
// Build: hipcc --offload-arch=gfx950 -O3 store_fused.hip -o store_fused
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int kRows = 129, kCtus = 510;

__global__ void __launch_bounds__(512) store_kernel(char *o, size_t pitch, u64 seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * kRows;
    const u64 v = seed + tid;
    for (int dy = 0; dy < kRows; dy++) {
        char *r = o + (row0 + dy) * pitch;
        char *r8 = r + wave * 2048 + lane * 8;
        *(u64 *)(r8) = v;
        *(u64 *)(r8 + 512) = v;
        *(u64 *)(r8 + 1024) = v;
        *(u64 *)(r8 + 1536) = v;
        *(u64 *)(r + 16384 + wave * 512 + lane * 8) = v;
        *(u32 *)(r + 20480 + wave * 256 + lane * 4) = (u32)v;
        if (lane < 16) *(u32 *)(r + 22528 + wave * 64 + lane * 4) = (u32)v;
    }
}

// the same bytes with every wavefront writing one contiguous 2 880-byte piece of the row (what a layout made for the
// stores alone would look like)
__global__ void __launch_bounds__(512) store_kernel_contig(char *o, size_t pitch, u64 seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * kRows;
    const u64 v = seed + tid;
    for (int dy = 0; dy < kRows; dy++) {
        char *r = o + (row0 + dy) * pitch + wave * 2880;
        *(u64 *)(r + lane * 8) = v;
        *(u64 *)(r + 512 + lane * 8) = v;
        *(u64 *)(r + 1024 + lane * 8) = v;
        *(u64 *)(r + 1536 + lane * 8) = v;
        *(u64 *)(r + 2048 + lane * 8) = v;
        *(u32 *)(r + 2560 + lane * 4) = (u32)v;
        if (lane < 16) *(u32 *)(r + 2816 + lane * 4) = (u32)v;
    }
}

int main()
{
    const size_t rows = (size_t)kCtus * kRows;
    char *o;
    (void)hipMalloc((void **)&o, rows * 32768);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (size_t pitch : { (size_t)23040, (size_t)23168, (size_t)23296, (size_t)23552, (size_t)24576, (size_t)32768 })
        for (int kind = 0; kind < 2; kind++)
            for (int rep = 0; rep < 2; rep++) {
                const int n = 150;
                (void)hipEventRecord(e0);
                for (int i = 0; i < n; i++) {
                    if (kind == 0) store_kernel<<<kCtus, 512>>>(o, pitch, i);
                    else store_kernel_contig<<<kCtus, 512>>>(o, pitch, i);
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%s pitch %zu: %.1f us per launch, %.2f TB/s\n", kind ? "contiguous per wavefront" : "four parts of a row       ", pitch, ms / n * 1e3,
                                (double)rows * 23040 / (ms / n * 1e-3) / 1e12);
            }
    return 0;
}
