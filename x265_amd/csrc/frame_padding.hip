// Picture-border padding for gfx950: xFramePadding (src/encode.cpp:2043-2092). The reconstructed planes are extended by
// `pad` samples on every side by replicating the border sample (left / right per row, then whole rows up and down
// including the corners), which is what motion vectors pointing outside the picture read. One launch per plane: a
// thread writes one padding byte; rows above and below copy the already-extended first / last row by reading the
// nearest picture sample directly, so no ordering between the two steps is needed.
#include "xmi_common.h"

namespace {

__global__ void __launch_bounds__(256) frame_padding_kernel(uint8_t *plane, int stride, int width, int height, int pad)
{
    // padded rectangle: rows [-pad, height + pad), columns [-pad, width + pad), one thread per byte
    const int x = blockIdx.x * 256 + threadIdx.x - pad, y = (int)blockIdx.y - pad;
    if (x >= width + pad) return;
    if (x >= 0 && x < width && y >= 0 && y < height) return;       // a picture sample
    const int sx = min(max(x, 0), width - 1), sy = min(max(y, 0), height - 1);
    plane[(ptrdiff_t)y * stride + x] = plane[(ptrdiff_t)sy * stride + sx];
}

}  // namespace

extern "C" {

int xmi_frame_padding_dev(uint8_t *d_plane, int stride, int width, int height, int pad, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_plane, "xmi_frame_padding_dev: null plane");
    XMI_ARG(width > 0 && height > 0 && pad > 0 && stride >= width + 2 * pad, "xmi_frame_padding_dev: bad geometry %dx%d pad %d stride %d", width, height,
            pad, stride);
    hipLaunchKernelGGL(frame_padding_kernel, dim3((width + 2 * pad + 255) / 256, height + 2 * pad), dim3(256), 0, xmi::as_stream(stream), d_plane, stride,
                       width, height, pad);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_frame_padding(uint8_t *plane, int stride, int width, int height, int pad)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(plane && width > 0 && height > 0 && pad > 0 && stride >= width + 2 * pad, "xmi_frame_padding: bad argument");
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
    const size_t W = (size_t)width + 2 * pad, H = (size_t)height + 2 * pad;
    uint8_t *d = static_cast<uint8_t *>(xmi::scratch(10, W * H));
    if (!d) return XMI_ERR_HIP;
    uint8_t *top = plane - (ptrdiff_t)pad * stride - pad;
    XMI_HIP(hipMemcpy2D(d + (size_t)pad * W + pad, W, plane, (size_t)stride, (size_t)width, (size_t)height, hipMemcpyHostToDevice));
    rc = xmi_frame_padding_dev(d + (size_t)pad * W + pad, (int)W, width, height, pad, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy2D(top, (size_t)stride, d, W, W, H, hipMemcpyDeviceToHost));
    return XMI_OK;
}

}  // extern "C"
