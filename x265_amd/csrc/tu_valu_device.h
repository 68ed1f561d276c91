// tu_valu_device.h - the pieces of the VALU transform chain that more than one kernel uses: the core transform matrix, the
// quantiser constants and one transform stage on LDS operands (v_dot2 on int16 pairs). Include from a .hip file only.
#pragma once
#include "xmi_common.h"

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

// integers the HEVC core transform assigns to 64*sqrt(2)*cos(m*pi/64), m = 0..32 (m = 0 is the DC row value)
__constant__ int8_t k_cos64[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0 };
__constant__ int k_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };  // src/table.cpp:150
__constant__ int k_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                // src/table.cpp:155

__device__ __forceinline__ int dct_coef(int n, int k, int j)
{
    if (k == 0) return 64;
    const int m = (k * (32 / n) * (2 * j + 1)) & 127;
    if (m <= 32) return k_cos64[m];
    if (m <= 64) return -k_cos64[64 - m];
    if (m <= 96) return -k_cos64[m - 64];
    return k_cos64[128 - m];
}

__device__ __forceinline__ int clip16(int v) { return min(32767, max(-32768, v)); }

template <int N>
struct Cfg {
    static constexpr int LD = N + 2;                                   // int16 elements per LDS row (odd dword pitch)
    static constexpr int TPT = N * N < 256 ? N * N : 256;              // threads per TU
    static constexpr int TUS = 256 / TPT;                              // TUs per 256-thread workgroup
    static constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
};

// out[a][b] = f((sum_c mat[a][c] * in[b][c] + rnd) >> shift); all buffers [N][LD] int16 in LDS
template <int N, bool CLIP>
__device__ __forceinline__ void stage(const int16_t *mat, const int16_t *in, int16_t *out, int shift, int t)
{
    constexpr int LD = Cfg<N>::LD, TPT = Cfg<N>::TPT;
    const int rnd = 1 << (shift - 1);
#pragma unroll
    for (int e = t; e < N * N; e += TPT) {
        const int a = e / N, b = e % N;
        const s16x2 *mr = reinterpret_cast<const s16x2 *>(mat + a * LD);
        const s16x2 *ir = reinterpret_cast<const s16x2 *>(in + b * LD);
        int acc = 0;
#pragma unroll
        for (int c = 0; c < N / 2; c++) acc = __builtin_amdgcn_sdot2(mr[c], ir[c], acc, false);
        const int v = (acc + rnd) >> shift;
        out[a * LD + b] = (int16_t)(CLIP ? clip16(v) : v);
    }
}

}  // namespace
