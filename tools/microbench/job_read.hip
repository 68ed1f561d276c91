// job_read.hip - where should a small job table live that the host writes before every launch and one workgroup reads at
// its start? Compares, per launch of one workgroup: a dependent chain of reads (record header -> payload) from
//   (a) coherent mapped host memory (hipHostMalloc mapped | coherent): the reads cross the bus,
//   (b) fine-grained device memory that the host writes through the PCIe aperture (hipExtMallocWithFlags
//       hipDeviceMallocFinegrained): the host's writes are posted, the kernel reads locally,
//   (c) device memory filled by hipMemcpyAsync on the launch stream.
// Prints the in-kernel time of the chain (s_memrealtime, 100 MHz) and the host's launch-to-completion round trip.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                          \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

struct Job {
    unsigned next;            // index of the record to read next (a dependent read)
    unsigned payload[63];
};

__global__ void __launch_bounds__(256) read_jobs(const Job *jobs, unsigned first, unsigned *flag, unsigned seq, unsigned long long *ticks, unsigned *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // what a round's workgroup does first: its record (256 B, every thread a dword), then one more record named by the first
    unsigned v = reinterpret_cast<const unsigned *>(jobs + first)[threadIdx.x & 63];
    const unsigned nxt = __shfl(v, 0) & 63u;
    v += reinterpret_cast<const unsigned *>(jobs + nxt)[threadIdx.x & 63];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (v == 0xdeadbeefu) sink[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ticks, t1 - t0);
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main()
{
    const int n = 64, iters = 2000;
    Job *h_map = nullptr, *d_fine = nullptr, *d_plain = nullptr, *h_pinned = nullptr;
    unsigned *flag = nullptr, *sink = nullptr;
    unsigned long long *ticks = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&h_map), n * sizeof(Job), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(reinterpret_cast<void **>(&h_pinned), n * sizeof(Job), hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipMalloc(reinterpret_cast<void **>(&d_plain), n * sizeof(Job)));
    CK(hipMalloc(reinterpret_cast<void **>(&sink), 64));
    CK(hipMalloc(reinterpret_cast<void **>(&ticks), 8));
    const hipError_t fe = hipExtMallocWithFlags(reinterpret_cast<void **>(&d_fine), n * sizeof(Job), hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(fe));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<Job> shadow(n);

    auto run = [&](const char *name, int kind) -> int {
        CK(hipMemsetAsync(ticks, 0, 8, st));
        CK(hipStreamSynchronize(st));
        *flag = 0;
        double host_us = 0, fill_us = 0;
        for (int i = 1; i <= iters; i++) {
            for (int k = 0; k < 8; k++) {      // the host writes eight records per round
                shadow[k].next = (unsigned)((k + i) & 7);
                for (int j = 0; j < 63; j++) shadow[k].payload[j] = (unsigned)(i * 131 + j);
            }
            const auto t0 = std::chrono::steady_clock::now();
            const Job *src = nullptr;
            if (kind == 0) std::memcpy(h_map, shadow.data(), 8 * sizeof(Job)), src = h_map;
            else if (kind == 1) std::memcpy(d_fine, shadow.data(), 8 * sizeof(Job)), src = d_fine;
            else {
                std::memcpy(h_pinned, shadow.data(), 8 * sizeof(Job));
                CK(hipMemcpyAsync(d_plain, h_pinned, 8 * sizeof(Job), hipMemcpyHostToDevice, st));
                src = d_plain;
            }
            const auto t1 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(read_jobs, dim3(1), dim3(256), 0, st, src, 0u, flag, (unsigned)i, ticks, sink);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned)i) {
            }
            const auto t2 = std::chrono::steady_clock::now();
            fill_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
            host_us += std::chrono::duration<double, std::micro>(t2 - t1).count();
        }
        unsigned long long tk = 0;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
        std::printf("%-44s in-kernel read chain %.2f us, host fill %.2f us, launch to flag %.2f us\n", name, tk / (double)iters / 100.0, fill_us / iters, host_us / iters);
        return 0;
    };
    if (run("mapped coherent host memory", 0)) return 1;
    if (fe == hipSuccess) {
        if (run("fine-grained device memory, host writes it", 1)) return 1;
    }
    if (run("device memory filled by hipMemcpyAsync", 2)) return 1;
    return 0;
}
