// Internal helpers shared by the HIP translation units of libx265_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/x265_amd.h"

namespace xmi {

void set_error(const char *fmt, ...);
int  require_device();          // XMI_OK or XMI_ERR_NO_DEVICE (initialises on first use)
hipStream_t as_stream(void *s);

// scratch device buffer owned by the library, grown on demand (used by the host-pointer entry points)
void *scratch(int slot, size_t bytes);

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
