// Streaming-store rate of 510 workgroups x 8 wavefronts writing 129 rows of 16 KiB each, by row pitch: a pitch that is a
// power of two against pitches with some lines of padding. Build: hipcc --offload-arch=gfx950 -O3 store_pitch.hip -o store_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
constexpr int kRows = 129, kCtus = 510;

__global__ void __launch_bounds__(512) store_kernel(char *o, size_t pitch, u64 seed)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * kRows;
    const u64 v = seed + tid;
    for (int dy = 0; dy < kRows; dy++) {
        char *r8 = o + (row0 + dy) * pitch + wave * 2048 + lane * 8;
        *(u64 *)(r8) = v;
        *(u64 *)(r8 + 512) = v;
        *(u64 *)(r8 + 1024) = v;
        *(u64 *)(r8 + 1536) = v;
    }
}

int main()
{
    const size_t rows = (size_t)kCtus * kRows;
    char *o;
    (void)hipMalloc((void **)&o, rows * 32768);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const size_t pitches[] = { 16384, 16384 + 128, 16384 + 256, 16384 + 512, 16384 + 1024, 16384 + 2048, 16384 + 4096, 16384 + 6656, 16384 + 8192, 32768 };
    for (size_t pitch : pitches)
        for (int rep = 0; rep < 2; rep++) {
            const int n = 150;
            (void)hipEventRecord(e0);
            for (int i = 0; i < n; i++) store_kernel<<<kCtus, 512>>>(o, pitch, i);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("pitch %6zu: %.1f us per launch, %.2f TB/s\n", pitch, ms / n * 1e3, (double)rows * 16384 / (ms / n * 1e-3) / 1e12);
        }
    return 0;
}
