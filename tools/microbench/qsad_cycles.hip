// Issue cost of v_qsad_pk_u16_u8 and v_sad_u8 relative to v_add_u32, from s_memtime inside one wavefront per SIMD (no
// contention, 8 independent chains), with s_memrealtime (100 MHz) beside it to turn ticks into time.
// Build: hipcc --offload-arch=gfx950 -O3 qsad_cycles.hip -o qsad_cycles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int kIters = 8192;

template <int OP>
__global__ void __launch_bounds__(1024) k(const u32 *in, u64 *out, u64 *ticks)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 r = ((u64)in[t & 1023] << 32) | in[(t + 7) & 1023];
    u32 c = in[(t + 3) & 1023];
    u64 a[8];
    u32 b[8];
    for (int i = 0; i < 8; i++) a[i] = i, b[i] = i;
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(b[i]) : "v"(c));
            if (OP == 1) a[i] = __builtin_amdgcn_qsad_pk_u16_u8(r, c, a[i]);
            if (OP == 2) b[i] = __builtin_amdgcn_sad_u8((u32)r, c, b[i]);
            if (OP == 3) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b[i]) : "v"(c));
        }
        if (OP == 1)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(a[i]));
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u64 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i] ^ b[i];
    out[t] = s;
    // the oldest wavefront of a SIMD has issue priority: the span of the whole workgroup is what counts
    __shared__ u64 lo[2], hi[2];
    if (threadIdx.x == 0) lo[0] = lo[1] = ~0ull, hi[0] = hi[1] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&lo[0], t0), atomicMax(&hi[0], t1), atomicMin(&lo[1], r0), atomicMax(&hi[1], r1);
    }
    __syncthreads();
    if (threadIdx.x == 0) ticks[blockIdx.x * 2] = hi[0] - lo[0], ticks[blockIdx.x * 2 + 1] = hi[1] - lo[1];
}

int main()
{
    u32 *in;
    u64 *out, *ticks;
    const int blocks = 256;      // one 256-thread workgroup per CU: one wavefront per SIMD
    (void)hipMalloc((void **)&in, 4096), (void)hipMemset(in, 0x5a, 4096);
    (void)hipMalloc((void **)&out, (size_t)blocks * 1024 * 8);
    (void)hipMalloc((void **)&ticks, blocks * 16);
    const char *names[4] = { "v_add_u32", "v_qsad_pk_u16_u8", "v_sad_u8", "v_pk_add_u16" };
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int op = 0; op < 4; op++) {
            u64 h[512];
            for (int rep = 0; rep < 3; rep++) {
                const int nb = blocks, th = 256 * waves;      // waves per SIMD inside one workgroup, one workgroup per CU
                if (op == 0) k<0><<<nb, th>>>(in, out, ticks);
                if (op == 1) k<1><<<nb, th>>>(in, out, ticks);
                if (op == 2) k<2><<<nb, th>>>(in, out, ticks);
                if (op == 3) k<3><<<nb, th>>>(in, out, ticks);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
            double tk = 0, rt = 0;
            for (int i = 0; i < 256; i++) tk += h[2 * i], rt += h[2 * i + 1];
            tk /= 256, rt /= 256;
            printf("%d wave(s)/SIMD %-18s %.2f s_memtime ticks per instruction and SIMD, %.2f ns per instruction and SIMD, ticks at %.3f GHz\n", waves, names[op],
                   tk / (kIters * 8.0 * waves), rt * 10.0 / (kIters * 8.0 * waves), tk / (rt * 10.0));
        }
    return 0;
}
