// Per-call drop-ins for the reference's pixel primitives: HOST pointers, the reference's strides and
// argument meaning, executed on the GPU through the batch kernels with count = 1.
// They make an unmodified caller (the reference's function-pointer tables, src/pixel.cpp:529-542, 711-717,
// 810-816, 833-844, and xSubDct/xQuant/xDeQuant/xIDctAdd) runnable against this library; they are not a fast
// path: each call is a host->device copy, one kernel and a copy back.
#include "xmi_common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace {

std::mutex g_call_mu;  // the staging buffers below are shared by all callers

struct Stage {
    uint8_t *dev = nullptr;   // device staging, one allocation, carved by the helpers
    std::vector<uint8_t> host;
};
Stage g_st;

bool ensure(size_t bytes)
{
    // asked for on every call: the library's scratch table is the owner and xmi_shutdown() empties it
    void *p = xmi::scratch(7, bytes < 65536 ? 65536 : bytes);
    if (!p) return false;
    g_st.dev = static_cast<uint8_t *>(p);
    return true;
}

// pack an n x n block read with `stride` into dst (contiguous)
template <typename T>
void pack(T *dst, const T *src, size_t stride, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, src + (size_t)y * stride, (size_t)w * sizeof(T));
}
template <typename T>
void unpack(T *dst, size_t stride, const T *src, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * stride, src + (size_t)y * w, (size_t)w * sizeof(T));
}

int64_t metric_call(const char *who, int metric, int n, const uint8_t *a, size_t sa, const uint8_t *b, size_t sb)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(a && b, "%s: null argument", who);
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32 || n == 64, "%s: block size %d not in {4,8,16,32,64}", who, n);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)n * n;
    if (!ensure(2 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(2 * nn);
    pack(g_st.host.data(), a, sa, n, n);
    pack(g_st.host.data() + nn, b, sb, n, n);
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 2 * nn, hipMemcpyHostToDevice));
    uint32_t *d_out = reinterpret_cast<uint32_t *>(g_st.dev + 2 * nn);
    rc = xmi_block_metric_batch_dev(metric, n, 1, g_st.dev, g_st.dev + nn, d_out, nullptr);
    if (rc != XMI_OK) return rc;
    uint32_t v = 0;
    XMI_HIP(hipMemcpy(&v, d_out, 4, hipMemcpyDeviceToHost));
    return (int64_t)v;
}

}  // namespace

extern "C" {

int64_t xmi_sad(int n, const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs)
{
    return metric_call("xmi_sad", XMI_METRIC_SAD, n, src, ss, ref, rs);
}

int64_t xmi_ssd(int n, const uint8_t *src, intptr_t ss, const uint8_t *ref, intptr_t rs)
{
    return metric_call("xmi_ssd", XMI_METRIC_SSD, n, src, (size_t)ss, ref, (size_t)rs);
}

int64_t xmi_satd(int n, const uint8_t *src, intptr_t ss, const uint8_t *ref, intptr_t rs)
{
    if (n == 64) {
        xmi::set_error("xmi_satd: xSatdN has no 64x64 entry");
        return XMI_ERR_ARG;
    }
    return metric_call("xmi_satd", XMI_METRIC_SATD_TAB, n, src, (size_t)ss, ref, (size_t)rs);
}

int64_t xmi_satd_hm(int n, const uint8_t *src, uint32_t ss, const uint8_t *ref, uint32_t rs)
{
    return metric_call("xmi_satd_hm", XMI_METRIC_SATD_HM, n, src, ss, ref, rs);
}

int xmi_sub_dct(int16_t *dst, const uint8_t *src, uint32_t stride, const uint8_t *pred, uint32_t stride_pred,
                int width, uint32_t mode)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && src && pred, "xmi_sub_dct: null argument");
    XMI_ARG(width == 4 || width == 8 || width == 16 || width == 32, "xmi_sub_dct: width %d not in {4,8,16,32}", width);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)width * width;
    if (!ensure(4 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(2 * nn + 16);
    pack(g_st.host.data(), src, stride, width, width);
    pack(g_st.host.data() + nn, pred, stride_pred, width, width);
    g_st.host[2 * nn] = (uint8_t)((width == 4 && mode != XMI_MODE_INVALID) ? 2 : 0);
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 2 * nn + 16, hipMemcpyHostToDevice));
    int16_t *d_coef = reinterpret_cast<int16_t *>(g_st.dev + 2 * nn + 16);
    rc = xmi_sub_dct_batch_dev(width, 1, g_st.dev, g_st.dev + nn, g_st.dev + 2 * nn, d_coef, nullptr);
    if (rc != XMI_OK) return rc;
    std::vector<int16_t> out(nn);
    XMI_HIP(hipMemcpy(out.data(), d_coef, nn * 2, hipMemcpyDeviceToHost));
    unpack(dst, stride, out.data(), width, width);   // the reference writes dst with the source stride
    return XMI_OK;
}

int64_t xmi_quant(int16_t *dst, const int16_t *src, uint32_t stride, uint32_t qp, int width, int height, int slice_type)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && src, "xmi_quant: null argument");
    XMI_ARG((width == 4 || width == 8 || width == 16 || width == 32) && height == width, "xmi_quant: %dx%d is not a square TU of 4..32", width, height);
    XMI_ARG(qp <= 51, "xmi_quant: qp %u > 51", qp);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)width * width;
    if (!ensure(4 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(2 * nn + 16);
    pack(reinterpret_cast<int16_t *>(g_st.host.data()), src, stride, width, width);
    g_st.host[2 * nn] = (uint8_t)qp;
    g_st.host[2 * nn + 1] = (uint8_t)(slice_type == XMI_SLICE_I ? 1 : 0);
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 2 * nn + 16, hipMemcpyHostToDevice));
    int16_t *d_level = reinterpret_cast<int16_t *>(g_st.dev + 2 * nn + 16);
    uint32_t *d_sum = reinterpret_cast<uint32_t *>(g_st.dev + 4 * nn + 16);
    rc = xmi_quant_batch_dev(width, 1, reinterpret_cast<const int16_t *>(g_st.dev), g_st.dev + 2 * nn, g_st.dev + 2 * nn + 1, d_level, d_sum, nullptr);
    if (rc != XMI_OK) return rc;
    std::vector<int16_t> out(nn + 2);
    XMI_HIP(hipMemcpy(out.data(), d_level, nn * 2 + 4, hipMemcpyDeviceToHost));
    unpack(dst, stride, out.data(), width, width);
    uint32_t sum;
    memcpy(&sum, out.data() + nn, 4);
    return (int64_t)sum;
}

int xmi_dequant(int16_t *dst, const int16_t *src, uint32_t stride, uint32_t qp, int width, int height)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && src, "xmi_dequant: null argument");
    XMI_ARG((width == 4 || width == 8 || width == 16 || width == 32) && height == width, "xmi_dequant: %dx%d is not a square TU of 4..32", width, height);
    XMI_ARG(qp <= 51, "xmi_dequant: qp %u > 51", qp);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)width * width;
    if (!ensure(4 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(2 * nn + 16);
    pack(reinterpret_cast<int16_t *>(g_st.host.data()), src, stride, width, width);
    g_st.host[2 * nn] = (uint8_t)qp;
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 2 * nn + 16, hipMemcpyHostToDevice));
    int16_t *d_coef = reinterpret_cast<int16_t *>(g_st.dev + 2 * nn + 16);
    rc = xmi_dequant_batch_dev(width, 1, reinterpret_cast<const int16_t *>(g_st.dev), g_st.dev + 2 * nn, d_coef, nullptr);
    if (rc != XMI_OK) return rc;
    std::vector<int16_t> out(nn);
    XMI_HIP(hipMemcpy(out.data(), d_coef, nn * 2, hipMemcpyDeviceToHost));
    unpack(dst, stride, out.data(), width, width);
    return XMI_OK;
}

int xmi_idct_add(uint8_t *dst, uint32_t stride_dst, const int16_t *src, uint32_t stride, const uint8_t *pred,
                 int width, uint32_t mode)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && src && pred, "xmi_idct_add: null argument");
    XMI_ARG(width == 4 || width == 8 || width == 16 || width == 32, "xmi_idct_add: width %d not in {4,8,16,32}", width);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)width * width;
    if (!ensure(4 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(3 * nn + 16);
    pack(reinterpret_cast<int16_t *>(g_st.host.data()), src, stride, width, width);
    pack(g_st.host.data() + 2 * nn, pred, stride, width, width);   // the reference reads pred with the coefficient stride
    g_st.host[3 * nn] = (uint8_t)((width == 4 && mode != XMI_MODE_INVALID) ? 2 : 0);
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 3 * nn + 16, hipMemcpyHostToDevice));
    uint8_t *d_rec = g_st.dev + 3 * nn + 16;
    rc = xmi_idct_add_batch_dev(width, 1, reinterpret_cast<const int16_t *>(g_st.dev), g_st.dev + 2 * nn, g_st.dev + 3 * nn, d_rec, nullptr);
    if (rc != XMI_OK) return rc;
    std::vector<uint8_t> out(nn);
    XMI_HIP(hipMemcpy(out.data(), d_rec, nn, hipMemcpyDeviceToHost));
    unpack(dst, stride_dst, out.data(), width, width);
    return XMI_OK;
}

int64_t xmi_tu_chain(const uint8_t *pix, const uint8_t *pred, uint8_t *rec, int16_t *coef, int size, int stride,
                     int stride_rec, uint32_t qp, uint32_t mode, int slice_type)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(pix && pred && rec && coef, "xmi_tu_chain: null argument");
    XMI_ARG(size == 4 || size == 8 || size == 16 || size == 32, "xmi_tu_chain: size %d not in {4,8,16,32}", size);
    XMI_ARG(qp <= 51, "xmi_tu_chain: qp %u > 51", qp);
    std::lock_guard<std::mutex> lk(g_call_mu);
    const size_t nn = (size_t)size * size;
    if (!ensure(6 * nn + 64)) return XMI_ERR_HIP;
    g_st.host.resize(2 * nn + 16);
    pack(g_st.host.data(), pix, (size_t)stride, size, size);
    pack(g_st.host.data() + nn, pred, (size_t)stride, size, size);
    g_st.host[2 * nn] = (uint8_t)qp;
    g_st.host[2 * nn + 1] = (uint8_t)((slice_type == XMI_SLICE_I ? 1 : 0) | ((size == 4 && mode != XMI_MODE_INVALID) ? 2 : 0));
    XMI_HIP(hipMemcpy(g_st.dev, g_st.host.data(), 2 * nn + 16, hipMemcpyHostToDevice));
    int16_t *d_level = reinterpret_cast<int16_t *>(g_st.dev + 2 * nn + 16);
    uint32_t *d_sum = reinterpret_cast<uint32_t *>(g_st.dev + 4 * nn + 16);
    uint8_t *d_rec = g_st.dev + 4 * nn + 32;
    rc = xmi_tu_chain_batch_dev(size, 1, g_st.dev, g_st.dev + nn, g_st.dev + 2 * nn, g_st.dev + 2 * nn + 1, d_level, d_rec, d_sum, nullptr);
    if (rc != XMI_OK) return rc;
    std::vector<uint8_t> out(3 * nn + 16);
    XMI_HIP(hipMemcpy(out.data(), d_level, 3 * nn + 16, hipMemcpyDeviceToHost));
    unpack(coef, (size_t)stride, reinterpret_cast<const int16_t *>(out.data()), size, size);
    unpack(rec, (size_t)stride_rec, out.data() + 2 * nn + 16, size, size);
    uint32_t sum;
    memcpy(&sum, out.data() + 2 * nn, 4);
    return (int64_t)sum;
}

}  // extern "C"
