// Internal helpers shared by the HIP translation units of libx265_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <mutex>
#include <cstdint>
#include <cstdio>

#include "../../include/x265_amd.h"

namespace xmi {

void set_error(const char *fmt, ...);
int  require_device();          // XMI_OK or XMI_ERR_*: binds on first use; the caller's current device must be the bound one
int  enter_device();            // the same for host-pointer / handle entry points: makes the bound device current for the thread
int  bound_device();            // -1 before the first binding
hipStream_t as_stream(void *s);
// xmi_interp_planes_dev with a choice of layout: whole planes one after the other (the documented layout of the entry point) or,
// phases_by_row, the 16 phases of a row next to each other - sample (y, x) of phase ph at d_planes[(y * 16 + ph) * stride + x]
int interp_planes_launch(const uint8_t *d_ref_base, int stride, int rows, int cols, uint8_t *d_planes, bool phases_by_row, void *stream);

// per-device set-up of sad_surface.hip (LDS opt-in of its kernels); called when the library binds a device
int sad_surface_prepare_device();
// per-device set-up of deblock.hip (address of its statically allocated staging array on `device`, which is current)
int deblock_prepare_device(int device);
// per-device set-up of rdoq.hip (coefficient scan tables in device memory)
int rdoq_prepare_device();

// scratch device buffer owned by the library, grown on demand (used by the host-pointer entry points)
void *scratch(int slot, size_t bytes);
// serialises the picture-level host-pointer entry points (SAD surface, deblocking, SAO, padding), which share scratch slots
std::mutex &picture_call_mutex();

#define XMI_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            xmi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return XMI_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define XMI_ARG(cond, ...)                  \
    do {                                    \
        if (!(cond)) {                      \
            xmi::set_error(__VA_ARGS__);    \
            return XMI_ERR_ARG;             \
        }                                   \
    } while (0)

// ---- layout of the SAD surface planes (sad_surface.hip writes it, me_search.hip reads it) ----
// level 0..3 = 8x8 / 16x16 / 32x32 / 64x64 blocks of a CTU; blk = raster index of the block at that level.
// Main part, dx in [-R, R-1]: [ctu][dy][dx / 16][slot][dx % 16]; edge part, dx = +R: after the main part, [ctu][dy][slot].
// The (ctu, dy) rows of the main part are surface_row_pitch elements apart: the payload (a power of two bytes) plus a few
// cache lines that nobody writes. Hundreds of workgroups stream rows at the same time; with a power-of-two distance
// between them the streaming-store rate of the chip drops from 5.4 to 4.6 TB/s (tools/microbench/store_pitch.hip).
__host__ __device__ constexpr int surface_level_blocks(int level) { return level == 0 ? 64 : level == 1 ? 16 : level == 2 ? 4 : 1; }
__host__ __device__ constexpr int surface_row_elems(int R, int level) { return (R / 8) * surface_level_blocks(level) * 16; }
__host__ __device__ constexpr int surface_row_pitch(int R, int level)
{
    const int esize = level < 2 ? 2 : 4, bytes = surface_row_elems(R, level) * esize;
    return (bytes + (bytes / 32 > 128 ? bytes / 32 : 128)) / esize;
}

__host__ __device__ inline int surface_slot(int level, int blk)
{
    if (level == 0) {
        const int by = blk >> 3, bx = blk & 7, by16 = by >> 1, bx16 = bx >> 1;
        const int z = (((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1));
        return ((by & 1) * 2 + (bx & 1)) * 16 + z;
    }
    if (level == 1) {
        const int by16 = blk >> 2, bx16 = blk & 3;
        return (((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1));
    }
    return level == 2 ? blk : 0;
}

__host__ __device__ inline size_t surface_index(int R, size_t nctu, int level, int ctu, int blk, int dy, int dx)
{
    const int dxi = dx + R, dyi = dy + R, slot = surface_slot(level, blk);
    const size_t nmv = (size_t)(2 * R + 1), nb = (size_t)surface_level_blocks(level);
    const size_t row = (size_t)ctu * nmv + (size_t)dyi;
    if (dxi == 2 * R) return nctu * nmv * (size_t)surface_row_pitch(R, level) + row * nb + (size_t)slot;
    return row * (size_t)surface_row_pitch(R, level) + ((size_t)(dxi >> 4) * nb + (size_t)slot) * 16 + (size_t)(dxi & 15);
}

// Bound on the integer displacement (either axis, either sign) of every block one motion search can read: predictors
// and start point |mvp >> 2| + 1, diamond points inside the clip box, sub-pel points within 5 of the box after the final
// clip (which bounds x by the Y maximum, src/encode.cpp:2950-2953); one more for good measure.
__host__ __device__ inline int me_request_reach(const xmi_me_request &rq)
{
    int reach = 0;
    for (int c = 0; c < 2; c++) {
        const int v[4] = { (rq.mvp[0][c] >> 2), (rq.mvp[1][c] >> 2), (int)rq.mv_min_ipel[c], (int)rq.mv_max_ipel[c] };
        for (int k = 0; k < 4; k++) {
            const int a = (v[k] < 0 ? -v[k] : v[k]) + (k < 2 ? 2 : 6);
            reach = a > reach ? a : reach;
        }
    }
    return reach;
}

}  // namespace xmi

#ifdef __HIPCC__
// Sum of |H_T D H_T| of a T x T residual tile (T = 4 or 8) held as packed 16-bit pairs of neighbouring columns
// (d[row][column pair]; |coefficient| <= 255 T^2 <= 16 320 fits). Column transform and the outer row stages are
// packed adds/subtracts; the last row stage, which would combine the two halves of a pair, is folded into the
// magnitude sum with |u + v| + |u - v| = 2 max(|u|, |v|).
namespace xmi {
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int T>
__device__ __forceinline__ unsigned hadamard_abs_sum_packed(s16x2 (&d)[T][T / 2])
{
    constexpr int P = T / 2;
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int len = 1; len < T; len <<= 1)
#pragma unroll
            for (int i = 0; i < T; i += 2 * len)
#pragma unroll
                for (int j = i; j < i + len; j++) {
                    const s16x2 x = d[j][p], y = d[j + len][p];
                    d[j][p] = x + y;
                    d[j + len][p] = x - y;
                }
    unsigned s = 0;
#pragma unroll
    for (int r = 0; r < T; r++) {
#pragma unroll
        for (int len = 1; len < P; len <<= 1)
#pragma unroll
            for (int i = 0; i < P; i += 2 * len)
#pragma unroll
                for (int j = i; j < i + len; j++) {
                    const s16x2 x = d[r][j], y = d[r][j + len];
                    d[r][j] = x + y;
                    d[r][j + len] = x - y;
                }
#pragma unroll
        for (int p = 0; p < P; p++) {
            const s16x2 m = __builtin_elementwise_max(d[r][p], -d[r][p]);
            s += 2u * (unsigned)(unsigned short)(m.x > m.y ? m.x : m.y);
        }
    }
    return s;
}
}  // namespace xmi
#endif
