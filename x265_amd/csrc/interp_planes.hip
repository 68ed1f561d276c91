// Quarter-pel luma interpolation planes for gfx950.
//
// xPredInterLumaBlk (src/pixel.cpp:8745-8795) builds the predicted block of a quarter-pel MV with the HEVC
// 8-tap filters (xg_lumaFilter, src/table.cpp:525-530): horizontal (src/pixel.cpp:7705), vertical (:8137) or
// horizontal-then-vertical through a 16-bit intermediate (:8690). Each output sample depends only on its
// position in the reference frame and on the phase (mvx & 3, mvy & 3), never on the block, so the 16 phase
// planes of a whole reference frame can be produced once per frame:
//     pred(x0, y0, mv)[i][j] = plane[(mv.y & 3) * 4 + (mv.x & 3)][y0 + (mv.y >> 2) + i][x0 + (mv.x >> 2) + j]
// Rounding follows the reference exactly: H and V: (sum + 32) >> 6; HV: first stage sum - 8192 kept in 16 bits,
// second stage (sum' + 8192*64 + 2048) >> 12, which equals (sum of unbiased first-stage sums * taps + 2048) >> 12
// because the taps add up to 64. All results are clipped to 0..255.
#include "xmi_common.h"

#include <type_traits>

namespace {

typedef unsigned int u32;
typedef short s16x2 __attribute__((ext_vector_type(2)));

// xg_lumaFilter (src/table.cpp:525-530), fractions 1..3, packed for v_dot4_i32_i8 (taps 0..3 | 4..7 as signed bytes)
// and for v_dot2_i32_i16 (tap pairs as signed 16-bit halves)
constexpr int kTap[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                             { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
constexpr int pack4(int f, int o) { return (kTap[f][o] & 255) | ((kTap[f][o + 1] & 255) << 8) | ((kTap[f][o + 2] & 255) << 16) | ((kTap[f][o + 3] & 255) << 24); }
constexpr int pack2(int f, int o) { return (kTap[f][o] & 0xffff) | ((kTap[f][o + 1] & 0xffff) << 16); }

// Output tile: 128 columns that start on a multiple of 128 of the row - with a 128-byte aligned buffer and stride a
// wavefront's store is two whole lines per plane - by 8 rows
constexpr int TW = 128, TH = 8;
constexpr int RW = TW + 8, RH = TH + 7;

__device__ __forceinline__ int clip8(int v) { return min(255, max(0, v)); }
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 8-tap sum over bytes held as two dwords of (sample - 128): sum c_t p_t = dot + 128 * 64
template <int F>
__device__ __forceinline__ int tap8_bytes(u32 lo, u32 hi)
{
    return __builtin_amdgcn_sdot4((int)hi, pack4(F, 4), __builtin_amdgcn_sdot4((int)lo, pack4(F, 0), 8192, false), false);
}
// 8-tap sum over 16-bit values held as four dwords of (row 2k, row 2k + 1) pairs
template <int F>
__device__ __forceinline__ int tap8_pairs(const u32 (&p)[4])
{
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = pack2(F, 2 * k);
        s = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p[k]), __builtin_bit_cast(s16x2, c), s, false);
    }
    return s;
}

template <int F>
__device__ __forceinline__ void h_sums(const u32 (&lo)[4], const u32 (&hi)[4], int16_t *dst)
{
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = tap8_bytes<F>(lo[j], hi[j]);
    *reinterpret_cast<uint2 *>(dst) = make_uint2((u32)(v[0] & 0xffff) | ((u32)v[1] << 16), (u32)(v[2] & 0xffff) | ((u32)v[3] << 16));
}

__device__ __forceinline__ void store4(uint8_t *dst, u32 v, int x, int cols, bool aligned)
{
    if (x >= 4 && x + 4 <= cols - 4 && aligned) {
        *reinterpret_cast<u32 *>(dst) = v;
    } else {
        for (int e = 0; e < 4; e++)
            if (x + e >= 4 && x + e < cols - 4) dst[e] = (uint8_t)(v >> (8 * e));
    }
}

// buffer: `rows` x `stride` bytes; samples within 4 of the buffer edge are left untouched in planes 1..15
// Output: sample (y, x) of phase ph at planes[ph * phase_pitch + y * row_pitch + x] - whole planes one after the other
// (phase_pitch = rows * stride, row_pitch = stride) or the 16 phases of a row next to each other (phase_pitch = stride,
// row_pitch = 16 * stride).
__global__ void __launch_bounds__(256) interp_kernel(const uint8_t *ref, uint8_t *planes, int stride, int rows, int cols, size_t row_pitch, size_t phase_pitch)
{
    __shared__ __align__(16) uint8_t s_ref[RH][RW];          // window rows y0-3 .., columns x0-4 ..
    __shared__ __align__(16) int16_t s_h[3][RH][TW];         // horizontal sums for xf = 1..3 (unbiased, fit 16 bits)
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = 4 + blockIdx.y * TH;       // first output sample of the tile (columns 0..3 are not written)
    // window: rows y0-3 .., columns x0-4 .. x0+131 (one column more on the left than the filters need, so that it starts
    // on a dword of the buffer); whole dwords when the buffer allows it, clamped bytes at the right edge otherwise. The
    // columns left of the buffer (first tile of a row) only feed the four outputs that are not written.
    const bool dwords = ((reinterpret_cast<uintptr_t>(ref) | (uintptr_t)stride) & 3) == 0 && x0 + RW - 4 <= cols;
    if (dwords) {
        for (int e = tid; e < RH * (RW / 4); e += 256) {
            const int ry = e / (RW / 4), c = e % (RW / 4);
            const int y = min(rows - 1, y0 - 3 + ry), x = max(0, x0 - 4 + 4 * c);
            reinterpret_cast<u32 *>(&s_ref[ry][0])[c] = *reinterpret_cast<const u32 *>(ref + (size_t)y * stride + x);
        }
    } else {
        for (int e = tid; e < RH * RW; e += 256) {
            const int ry = e / RW, rx = e % RW;
            const int y = min(rows - 1, y0 - 3 + ry), x = min(cols - 1, max(0, x0 - 4 + rx));
            s_ref[ry][rx] = ref[(size_t)y * stride + x];
        }
    }
    __syncthreads();
    // horizontal sums: an item = 4 consecutive columns of one window row (bytes 4g + 1 .. 4g + 11 of the row)
    for (int e = tid; e < RH * (TW / 4); e += 256) {
        const int ry = e / (TW / 4), g = e % (TW / 4);
        const u32 *w = reinterpret_cast<const u32 *>(&s_ref[ry][4 * g]);
        const u32 w0 = w[0] ^ 0x80808080u, w1 = w[1] ^ 0x80808080u, w2 = w[2] ^ 0x80808080u;
        u32 lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            lo[j] = __builtin_amdgcn_alignbyte(w1, w0, j + 1);
            hi[j] = __builtin_amdgcn_alignbyte(w2, w1, j + 1);
        }
        lo[3] = w1;
        hi[3] = w2;
        h_sums<1>(lo, hi, &s_h[0][ry][4 * g]);
        h_sums<2>(lo, hi, &s_h[1][ry][4 * g]);
        h_sums<3>(lo, hi, &s_h[2][ry][4 * g]);
    }
    __syncthreads();
    // each thread: 4 consecutive samples of one row, all 16 phases
    const int ty = tid / (TW / 4), tx = (tid % (TW / 4)) * 4;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= rows - 4) return;
    u32 out[16];

    // ---- integer and vertical phases: bytes (row ty + t, columns tx + 4 .. tx + 7) of the window, t = 0..7 ----
    u32 col_lo[4], col_hi[4];      // per output column j: the 8 rows as (sample - 128) bytes
    {
        u32 r[8];
#pragma unroll
        for (int t = 0; t < 8; t++)
            r[t] = *reinterpret_cast<const u32 *>(&s_ref[ty + t][tx + 4]);
        out[0] = r[3];
#pragma unroll
        for (int half = 0; half < 2; half++) {       // 4 x 4 byte transposes
            const u32 a = r[4 * half] ^ 0x80808080u, b = r[4 * half + 1] ^ 0x80808080u, c = r[4 * half + 2] ^ 0x80808080u,
                      d = r[4 * half + 3] ^ 0x80808080u;
            const u32 ab_lo = perm(b, a, 0x05010400u), ab_hi = perm(b, a, 0x07030602u);
            const u32 cd_lo = perm(d, c, 0x05010400u), cd_hi = perm(d, c, 0x07030602u);
            u32 *dst = half ? col_hi : col_lo;
            dst[0] = perm(cd_lo, ab_lo, 0x05040100u);
            dst[1] = perm(cd_lo, ab_lo, 0x07060302u);
            dst[2] = perm(cd_hi, ab_hi, 0x05040100u);
            dst[3] = perm(cd_hi, ab_hi, 0x07060302u);
        }
    }
    auto pack_v = [&](auto tag) {
        constexpr int F = decltype(tag)::value;
        u32 o = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) o |= (u32)clip8((tap8_bytes<F>(col_lo[j], col_hi[j]) + 32) >> 6) << (8 * j);
        return o;
    };
    out[4] = pack_v(std::integral_constant<int, 1>{});
    out[8] = pack_v(std::integral_constant<int, 2>{});
    out[12] = pack_v(std::integral_constant<int, 3>{});

    // ---- horizontal and diagonal phases from the 16-bit horizontal sums ----
#pragma unroll
    for (int xf = 1; xf < 4; xf++) {
        uint2 hrow[8];
#pragma unroll
        for (int t = 0; t < 8; t++) hrow[t] = *reinterpret_cast<const uint2 *>(&s_h[xf - 1][ty + t][tx]);
        {   // y fraction 0: row ty + 3
            const int v0 = (int)(short)(hrow[3].x & 0xffff), v1 = (int)hrow[3].x >> 16, v2 = (int)(short)(hrow[3].y & 0xffff), v3 = (int)hrow[3].y >> 16;
            out[xf] = (u32)clip8((v0 + 32) >> 6) | ((u32)clip8((v1 + 32) >> 6) << 8) | ((u32)clip8((v2 + 32) >> 6) << 16) | ((u32)clip8((v3 + 32) >> 6) << 24);
        }
        u32 pr[4][4];              // [column j][pair k] = (row 2k, row 2k + 1) of column j
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pr[0][k] = perm(hrow[2 * k + 1].x, hrow[2 * k].x, 0x05040100u);
            pr[1][k] = perm(hrow[2 * k + 1].x, hrow[2 * k].x, 0x07060302u);
            pr[2][k] = perm(hrow[2 * k + 1].y, hrow[2 * k].y, 0x05040100u);
            pr[3][k] = perm(hrow[2 * k + 1].y, hrow[2 * k].y, 0x07060302u);
        }
        auto pack_hv = [&](auto tag) {
            constexpr int F = decltype(tag)::value;
            u32 o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) o |= (u32)clip8((tap8_pairs<F>(pr[j]) + 2048) >> 12) << (8 * j);
            return o;
        };
        out[4 + xf] = pack_hv(std::integral_constant<int, 1>{});
        out[8 + xf] = pack_hv(std::integral_constant<int, 2>{});
        out[12 + xf] = pack_hv(std::integral_constant<int, 3>{});
    }
    const size_t at = (size_t)y * row_pitch + x;
    const bool aligned = (at & 3) == 0;
#pragma unroll
    for (int ph = 0; ph < 16; ph++) store4(planes + ph * phase_pitch + at, out[ph], x, cols, aligned);
}

}  // namespace

namespace xmi {
int interp_planes_launch(const uint8_t *d_ref_base, int stride, int rows, int cols, uint8_t *d_planes, bool phases_by_row, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_ref_base && d_planes, "xmi_interp_planes_dev: null argument");
    XMI_ARG(rows >= 16 && cols >= 16 && stride >= cols, "xmi_interp_planes_dev: bad geometry %d x %d, stride %d", cols, rows, stride);
    dim3 grid((cols - 4 + TW - 1) / TW, (rows - 8 + TH - 1) / TH);
    const size_t row_pitch = phases_by_row ? (size_t)16 * stride : (size_t)stride, phase_pitch = phases_by_row ? (size_t)stride : (size_t)rows * stride;
    hipLaunchKernelGGL(interp_kernel, grid, dim3(256), 0, xmi::as_stream(stream), d_ref_base, d_planes, stride, rows, cols, row_pitch, phase_pitch);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}
}  // namespace xmi

extern "C" int xmi_interp_planes_dev(const uint8_t *d_ref_base, int stride, int rows, int cols, uint8_t *d_planes, void *stream)
{
    return xmi::interp_planes_launch(d_ref_base, stride, rows, cols, d_planes, false, stream);
}
