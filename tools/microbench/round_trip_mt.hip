// Launch-and-spin round trips from several host threads at once, each with a stream and a completion word of its own - the
// pattern of several segment encoders sharing one GPU. Per thread count: round trips per second over all threads and the
// mean latency of one, for a kernel of one workgroup that ends at once and for one whose workgroups (64 of them) stay about
// 20 us (a spin on the clock), the length of an analysis round - without and with the round's traffic to host memory.
// Build: hipcc --offload-arch=gfx950 -O3 -pthread round_trip_mt.hip -o round_trip_mt
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

// host_io: every workgroup first reads a 256-byte record from mapped host memory and, before it reports, writes write_kb
// KiB of results there - the traffic of an analysis round
__global__ void __launch_bounds__(256) echo(unsigned *counter, unsigned target, unsigned *done, unsigned seq, unsigned hold_ticks, const unsigned *h_in, uint4 *h_out,
                                            unsigned write_kb)
{
    unsigned v = 0;
    if (h_in) v = h_in[blockIdx.x * 64 + (threadIdx.x & 63)];
    if (hold_ticks) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
    }
    if (h_out)
        for (unsigned k = 0; k < write_kb / 4; k++) h_out[((size_t)blockIdx.x * (write_kb / 4) + k) * 256 + threadIdx.x] = make_uint4(v, k, seq, threadIdx.x);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(counter, 1u) + 1u == target) {
        __threadfence_system();
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main()
{
    const int rounds = 4000;
    for (int mode = 0; mode < 6; mode++) {
        const int hold_us = mode == 0 ? 0 : 20;
        const unsigned write_kb = mode == 2 || mode == 5 ? 8 : mode == 3 ? 32 : 0;      // per workgroup
        const bool do_read = mode == 2 || mode == 3 || mode == 4;                  // a 256-byte record read in front
        for (int nthreads : { 1, 2, 4, 8, 16 }) {
            std::vector<std::thread> th;
            std::atomic<int> ready{ 0 };
            std::atomic<bool> go{ false };
            std::vector<double> lat(nthreads, 0.0);
            const int blocks = hold_us ? 64 : 1;
            for (int t = 0; t < nthreads; t++)
                th.emplace_back([&, t] {
                    unsigned *done = nullptr, *counter = nullptr, *h_in = nullptr;
                    uint4 *h_out = nullptr;
                    (void)hipHostMalloc((void **)&done, 256, hipHostMallocMapped | hipHostMallocCoherent);
                    if (do_read) (void)hipHostMalloc((void **)&h_in, 64 * 256, hipHostMallocMapped | hipHostMallocCoherent);
                    if (write_kb) (void)hipHostMalloc((void **)&h_out, (size_t)64 * write_kb * 1024, hipHostMallocMapped | hipHostMallocCoherent);
                    (void)hipMalloc((void **)&counter, 256);
                    (void)hipMemset(counter, 0, 256);
                    *done = 0;
                    hipStream_t st;
                    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    unsigned issued = 0;
                    for (int i = 1; i <= 200; i++) {      // warm up
                        issued += blocks;
                        hipLaunchKernelGGL(echo, dim3(blocks), dim3(256), 0, st, counter, issued, done, (unsigned)i, (unsigned)hold_us * 100u, h_in, h_out, write_kb);
                        while (*(volatile unsigned *)done != (unsigned)i) {}
                    }
                    ready++;
                    while (!go.load()) {}
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 201; i <= 200 + rounds; i++) {
                        issued += blocks;
                        hipLaunchKernelGGL(echo, dim3(blocks), dim3(256), 0, st, counter, issued, done, (unsigned)i, (unsigned)hold_us * 100u, h_in, h_out, write_kb);
                        while (*(volatile unsigned *)done != (unsigned)i) {}
                    }
                    lat[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
                    (void)hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                    (void)hipHostFree(done);
                    if (h_in) (void)hipHostFree(h_in);
                    if (h_out) (void)hipHostFree(h_out);
                    (void)hipFree(counter);
                });
            while (ready.load() < nthreads) {}
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            for (auto &t : th) t.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double mean = 0;
            for (double l : lat) mean += l / nthreads;
            std::printf("kernel of %2d workgroups holding %2d us, %s%2u KiB to host memory each, %2d threads: %7.0f round trips/s in all, %6.2f us each\n", blocks, hold_us,
                        do_read ? "a record from and " : "", write_kb, nthreads, nthreads * rounds / wall, mean);
        }
    }
    return 0;
}
