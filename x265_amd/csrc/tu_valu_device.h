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
    static constexpr int LD = N + 8;                                   // int16 elements per LDS row: rows start on 16 bytes, and the 16
                                                                       // rows a b128 load group touches fall into different banks
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
        int acc = 0;
        if constexpr (N >= 8) {      // eight elements of both rows per 16-byte LDS load
            const uint4 *mr = reinterpret_cast<const uint4 *>(mat + a * LD);
            const uint4 *ir = reinterpret_cast<const uint4 *>(in + b * LD);
#pragma unroll
            for (int c = 0; c < N / 8; c++) {
                const uint4 m = mr[c], v = ir[c];
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.x), __builtin_bit_cast(s16x2, v.x), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.y), __builtin_bit_cast(s16x2, v.y), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.z), __builtin_bit_cast(s16x2, v.z), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.w), __builtin_bit_cast(s16x2, v.w), acc, false);
            }
        } else {
            const uint2 m = *reinterpret_cast<const uint2 *>(mat + a * LD), v = *reinterpret_cast<const uint2 *>(in + b * LD);
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.x), __builtin_bit_cast(s16x2, v.x), acc, false);
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, m.y), __builtin_bit_cast(s16x2, v.y), acc, false);
        }
        const int v = (acc + rnd) >> shift;
        out[a * LD + b] = (int16_t)(CLIP ? clip16(v) : v);
    }
}

}  // namespace
