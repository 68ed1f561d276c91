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

namespace {

__constant__ int k_luma[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                  { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };

constexpr int TW = 64, TH = 16;   // output tile
constexpr int RW = TW + 8, RH = TH + 7;

__device__ __forceinline__ int clip8(int v) { return min(255, max(0, v)); }

// buffer: `rows` x `stride` bytes; samples within 4 of the buffer edge are left untouched in planes 1..15
__global__ void __launch_bounds__(256) interp_kernel(const uint8_t *ref, uint8_t *planes, int stride, int rows, int cols)
{
    __shared__ uint8_t s_ref[RH][RW];
    __shared__ int16_t s_h[3][RH][TW];     // horizontal sums for xf = 1..3 (unbiased, fit 16 bits)
    const int tid = threadIdx.x;
    const int x0 = 4 + blockIdx.x * TW, y0 = 4 + blockIdx.y * TH;   // first output sample of the tile
    const size_t plane_bytes = (size_t)rows * stride;

    for (int e = tid; e < RH * RW; e += 256) {
        const int ry = e / RW, rx = e % RW;
        const int y = min(rows - 1, y0 - 3 + ry), x = min(cols - 1, x0 - 3 + rx);
        s_ref[ry][rx] = ref[(size_t)y * stride + x];
    }
    __syncthreads();
    for (int e = tid; e < RH * TW; e += 256) {
        const int ry = e / TW, x = e % TW;
#pragma unroll
        for (int xf = 1; xf < 4; xf++) {
            int s = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) s += (int)s_ref[ry][x + t] * k_luma[xf][t];
            s_h[xf - 1][ry][x] = (int16_t)s;
        }
    }
    __syncthreads();
    // each thread: 4 consecutive samples of one row, all 16 phases
    const int ty = tid / 16, tx = (tid % 16) * 4;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= rows - 4) return;
    unsigned out[16];
#pragma unroll
    for (int ph = 0; ph < 16; ph++) out[ph] = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int cx = tx + e;
        out[0] |= (unsigned)s_ref[ty + 3][cx + 3] << (8 * e);
#pragma unroll
        for (int xf = 1; xf < 4; xf++) out[xf] |= (unsigned)clip8((s_h[xf - 1][ty + 3][cx] + 32) >> 6) << (8 * e);
#pragma unroll
        for (int yf = 1; yf < 4; yf++) {
            int s = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) s += (int)s_ref[ty + t][cx + 3] * k_luma[yf][t];
            out[yf * 4] |= (unsigned)clip8((s + 32) >> 6) << (8 * e);
#pragma unroll
            for (int xf = 1; xf < 4; xf++) {
                int v = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) v += (int)s_h[xf - 1][ty + t][cx] * k_luma[yf][t];
                out[yf * 4 + xf] |= (unsigned)clip8((v + 2048) >> 12) << (8 * e);
            }
        }
    }
    const size_t at = (size_t)y * stride + x;
#pragma unroll
    for (int ph = 0; ph < 16; ph++) {
        uint8_t *dst = planes + ph * plane_bytes + at;
        if (x + 4 <= cols - 4 && ((at & 3) == 0)) {
            *reinterpret_cast<unsigned *>(dst) = out[ph];
        } else {
            for (int e = 0; e < 4; e++)
                if (x + e < cols - 4) dst[e] = (uint8_t)(out[ph] >> (8 * e));
        }
    }
}

}  // namespace

extern "C" int xmi_interp_planes_dev(const uint8_t *d_ref_base, int stride, int rows, int cols, uint8_t *d_planes, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_ref_base && d_planes, "xmi_interp_planes_dev: null argument");
    XMI_ARG(rows >= 16 && cols >= 16 && stride >= cols, "xmi_interp_planes_dev: bad geometry %d x %d, stride %d", cols, rows, stride);
    dim3 grid((cols - 8 + TW - 1) / TW, (rows - 8 + TH - 1) / TH);
    hipLaunchKernelGGL(interp_kernel, grid, dim3(256), 0, xmi::as_stream(stream), d_ref_base, d_planes, stride, rows, cols);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}
