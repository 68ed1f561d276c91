// Host -> GPU -> host round trips, the two ways the analysis evaluator could make them:
//   launch: write a word, launch a kernel that copies it to a pinned completion word, spin on that word
//   doorbell: the kernel is already running and polls a pinned doorbell word; the host rings it and spins on the completion word
// Build: hipcc --offload-arch=gfx950 -O3 round_trip.hip -o round_trip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void echo(volatile unsigned *done, unsigned seq)
{
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store((unsigned *)done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// serves `rounds` doorbells, then leaves; gives up after about a second without one
__global__ void server(unsigned *bell, unsigned *done, unsigned first, unsigned rounds)
{
    if (threadIdx.x != 0) return;
    for (unsigned seq = first; seq < first + rounds; seq++) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main()
{
    unsigned *bell, *done;
    (void)hipHostMalloc((void **)&bell, 256, hipHostMallocMapped | hipHostMallocCoherent);
    (void)hipHostMalloc((void **)&done, 256, hipHostMallocMapped | hipHostMallocCoherent);
    *bell = *done = 0;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int n = 20000;
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= n; i++) {
            const unsigned seq = rep * n + i;
            hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, st, done, seq);
            while (*(volatile unsigned *)done != seq) {}
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        if (rep) printf("launch per round trip:   %.2f us\n", us);
    }
    (void)hipStreamSynchronize(st);
    *done = 0;
    const unsigned first = 1000000;
    hipLaunchKernelGGL(server, dim3(1), dim3(64), 0, st, bell, done, first, (unsigned)n);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) {
        const unsigned seq = first + i;
        __atomic_store_n(bell, seq, __ATOMIC_RELEASE);
        while (*(volatile unsigned *)done != seq) {}
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("doorbell per round trip: %.2f us\n", us);
    (void)hipStreamSynchronize(st);
    return 0;
}
