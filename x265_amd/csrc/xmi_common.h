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
