// Batched intra luma mode search for gfx950: the 35-mode loop of xEncDecideCU_Intra (src/encode.cpp:4005-4081).
//
// One wavefront per PU request. The wavefront stages the neighbouring samples in LDS, pads them (xPaddingRef,
// src/encode.cpp:3843), smooths them (xFilterRef, :3873) and stages the source block. The 35 predictions are
// never written anywhere: a lane owns one transform tile of one mode (4x4 tiles for the x264-style SATD bound
// for sizes 4..16, 8x8 tiles for xCalcHADs32x32_test at 32), evaluates xEncIntraPred (src/pixel.cpp:7338:
// planar :6489, DC :6704, angular :7243 incl. the edge filters and the transpose of the horizontal modes) pixel
// by pixel for that tile straight from the LDS samples, and runs the Hadamard transform of the residual in
// registers. 64 / tiles-per-block modes are evaluated at once; tile sums meet through a segmented DPP reduction.
// Cost = satd + {1,2,3} * sqrt_lambda in double, strict < in mode order, as the reference's loop does.
//
// xmi_intra_pred_batch_dev / xmi_enc_intra_pred write the prediction block itself (xEncIntraPred, one mode) with the
// same per-sample code: one wavefront per request.
#include "intra_device.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace {

// One kernel per PU size, each over the whole request array (wavefronts of other sizes leave at once): the
// register budget - and with it the number of resident wavefronts that hide the LDS and memory latency of this
// serial per-PU work - is then set by the size's own needs instead of by the 32x32 instance.
template <int N>
__global__ void __launch_bounds__(256) intra_search_kernel(const IntraParams p)
{
    constexpr int WPR = 1;       // wavefronts per request (4 was tried for 32x32 and lost: the unrolled code streams through the instruction cache either way)
    __shared__ __align__(16) uint8_t s_ref[4][272];
    __shared__ __align__(16) uint8_t s_src[4][N * N];
    __shared__ double s_best_cost[4];
    __shared__ int s_best_mode[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = WPR == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_intra_request rq = p.req[idx];
    if (rq.size != N) return;                   // another instance's size, or no PU size at all (then no result is written)
    search<N, WPR>(p, rq, idx, s_ref[wave], s_src[wave], lane, WPR == 4 ? wave : 0, s_best_cost, s_best_mode);
}

// ---- prediction blocks: one wavefront per request ----
struct IntraPredParams {
    const uint8_t *refs;
    const xmi_intra_pred_request *req;
    uint8_t *pred;
    int n;
};

__global__ void __launch_bounds__(256) intra_pred_kernel(const IntraPredParams p)
{
    __shared__ __align__(16) uint8_t s_ref[4][136];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_intra_pred_request rq = p.req[idx];
    if ((rq.size != 4 && rq.size != 8 && rq.size != 16 && rq.size != 32) || rq.mode > 34) return;   // no PU size or no mode: nothing is written
    const uint8_t *raw = p.refs + (size_t)idx * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < 4 * rq.size + 1; i += 64) s_ref[wave][i] = raw[i];
    __builtin_amdgcn_wave_barrier();
    uint8_t *dst = p.pred + rq.out;
    const bool is_luma = rq.is_luma != 0;
    switch (rq.size) {
    case 4: predict<4>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    case 8: predict<8>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    case 16: predict<16>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    default: predict<32>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    }
}

__global__ void __launch_bounds__(256) intra_chroma_search_kernel(const ChromaParams p)
{
    __shared__ __align__(16) uint8_t s_ref[4][2][136];
    __shared__ __align__(16) uint8_t s_src[4][2][256];
    __shared__ u32 s_satd[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_chroma_request rq = p.req[idx];
    const bool ok = (rq.size == 4 || rq.size == 8 || rq.size == 16) && rq.luma_mode < 35 && rq.x0 >= 0 && rq.y0 >= 0 && (rq.x0 & 3) == 0;
    if (!ok) {                   // marked, so that the caller can tell
        if (lane == 0) {
            xmi_chroma_result out = {};
            out.best_idx = out.best_mode = -1;
            p.res[idx] = out;
        }
        return;
    }
    switch (rq.size) {
    case 4: chroma_search<4>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    case 8: chroma_search<8>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    default: chroma_search<16>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    }
}

std::mutex g_pred_mu;   // the per-call form shares one staging area

// one kernel per PU size present in `sizes` (bit mask of 4, 8, 16, 32) over the whole request array
int launch_search(const uint8_t *d_src, int src_stride, const uint8_t *d_ref_samples, const xmi_intra_request *d_req, int n,
                  xmi_intra_result *d_res, int sizes, void *stream)
{
    IntraParams p = { d_src, d_ref_samples, d_req, d_res, src_stride, n };
    const dim3 grid((n + 3) / 4), block(256);
    if (sizes & 4) hipLaunchKernelGGL(intra_search_kernel<4>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 8) hipLaunchKernelGGL(intra_search_kernel<8>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 16) hipLaunchKernelGGL(intra_search_kernel<16>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 32) hipLaunchKernelGGL(intra_search_kernel<32>, grid, block, 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

}  // namespace

// ---- neighbour gather -------------------------------------------------------------------------------------------
// One wavefront per PU: the 4 size + 1 neighbouring samples in the order of src/encode.cpp:4010-4015 (left column from
// the bottom up, top-left, top row, top-right) out of the reconstruction plane, and the five availability bits of
// :3965-3974 out of the map of reconstructed 4x4 units. Samples of a segment that is not available are set to 0x80 (the
// reference leaves whatever its cache holds there and xPaddingRef overwrites them).
namespace {
struct GatherParams {
    const uint8_t *rec;
    const uint8_t *coded;
    xmi_intra_request *req;
    uint8_t *refs;
    int stride, width, height, coded_stride, n;
};

__global__ void __launch_bounds__(256) intra_gather_kernel(const GatherParams p)
{
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= p.n) return;
    const xmi_intra_request rq = p.req[idx];
    const int n = rq.size, x0 = rq.x0, y0 = rq.y0;
    auto unit = [&](int x, int y) -> bool {       // is the 4x4 unit that holds sample (x, y) reconstructed?
        return x >= 0 && y >= 0 && x < p.width && y < p.height && p.coded[(size_t)(y >> 2) * p.coded_stride + (x >> 2)] != 0;
    };
    const bool lb = unit(x0 - 1, y0 + n), l = unit(x0 - 1, y0), lt = unit(x0 - 1, y0 - 1), t = unit(x0, y0 - 1), tr = unit(x0 + n, y0 - 1);
    uint8_t *out = p.refs + (size_t)idx * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < 4 * n + 1; i += 64) {
        int x, y;
        bool ok;
        if (i < 2 * n) {
            x = x0 - 1, y = y0 + 2 * n - 1 - i;
            ok = i < n ? lb : l;
        } else {
            x = x0 - 1 + (i - 2 * n), y = y0 - 1;
            ok = i == 2 * n ? lt : i <= 3 * n ? t : tr;
        }
        // a segment is flagged by its first unit; the part of it that leaves the picture reads as 0x80
        ok = ok && x >= 0 && y >= 0 && x < p.width && y < p.height;
        out[i] = ok ? p.rec[(size_t)y * p.stride + x] : (uint8_t)0x80;
    }
    if (lane == 0) p.req[idx].valids = (uint8_t)((lb ? 1 : 0) | (l ? 2 : 0) | (lt ? 4 : 0) | (t ? 8 : 0) | (tr ? 16 : 0));
}
}  // namespace

extern "C" int xmi_intra_gather_dev(const uint8_t *d_rec, int rec_stride, int width, int height, const uint8_t *d_coded, int coded_stride,
                                    xmi_intra_request *d_req, int n, uint8_t *d_ref_samples, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_gather_dev: negative count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_rec && d_coded && d_req && d_ref_samples, "xmi_intra_gather_dev: null argument");
    XMI_ARG(width > 0 && height > 0 && width % 4 == 0 && height % 4 == 0 && rec_stride >= width && coded_stride >= width / 4,
            "xmi_intra_gather_dev: %dx%d, strides %d / %d", width, height, rec_stride, coded_stride);
    GatherParams p = { d_rec, d_coded, d_req, d_ref_samples, rec_stride, width, height, coded_stride, n };
    hipLaunchKernelGGL(intra_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_intra_pred_batch_dev(const uint8_t *d_ref_samples, const xmi_intra_pred_request *d_req, int n, uint8_t *d_pred, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_pred_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_ref_samples && d_req && d_pred, "xmi_intra_pred_batch_dev: null argument");
    IntraPredParams p = { d_ref_samples, d_req, d_pred, n };
    hipLaunchKernelGGL(intra_pred_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_enc_intra_pred(uint8_t *dst, const uint8_t *ref, uint32_t stride, uint32_t mode, uint32_t size, uint32_t is_luma)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && ref, "xmi_enc_intra_pred: null argument");
    XMI_ARG(size == 4 || size == 8 || size == 16 || size == 32, "xmi_enc_intra_pred: size %u not in {4,8,16,32}", size);
    XMI_ARG(mode < 35, "xmi_enc_intra_pred: mode %u > 34", mode);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    const size_t rq_off = 144, out_off = 160, blk = (size_t)size * size;
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, out_off + 1024));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[160];
    memset(host, 0, sizeof host);
    memcpy(host, ref, 4 * size + 1);
    xmi_intra_pred_request rq = { (uint8_t)size, (uint8_t)mode, (uint8_t)(is_luma ? 1 : 0), 0, 0 };
    memcpy(host + rq_off, &rq, sizeof rq);
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    IntraPredParams p = { dev, reinterpret_cast<const xmi_intra_pred_request *>(dev + rq_off), dev + out_off, 1 };
    hipLaunchKernelGGL(intra_pred_kernel, dim3(1), dim3(256), 0, nullptr, p);
    XMI_HIP(hipGetLastError());
    uint8_t out[1024];
    XMI_HIP(hipMemcpy(out, dev + out_off, blk, hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < size; y++) memcpy(dst + (size_t)y * stride, out + (size_t)y * size, size);
    return XMI_OK;
}

extern "C" int xmi_intra_search(const uint8_t *src, uint32_t src_stride, const uint8_t *ref_samples, const xmi_intra_request *req,
                                xmi_intra_result *res)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(src && ref_samples && req && res, "xmi_intra_search: null argument");
    const int n = req->size;
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32, "xmi_intra_search: size %d not in {4,8,16,32}", n);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    // staging: source block packed (rows n apart) | samples | request | result
    const size_t off_ref = 1024, off_req = off_ref + 144, off_res = off_req + 32, total = off_res + sizeof(xmi_intra_result);
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, 2048));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[off_res];
    memset(host, 0, sizeof host);
    for (int y = 0; y < n; y++) memcpy(host + (size_t)y * n, src + (size_t)y * src_stride, (size_t)n);
    memcpy(host + off_ref, ref_samples, (size_t)(4 * n + 1));
    xmi_intra_request rq = *req;
    rq.x0 = rq.y0 = 0;
    memcpy(host + off_req, &rq, sizeof rq);
    static_assert(sizeof(xmi_intra_request) <= 32 && 1024 + 144 + 32 + sizeof(xmi_intra_result) <= 2048, "staging layout");
    (void)total;
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    rc = launch_search(dev, n, dev + off_ref, reinterpret_cast<const xmi_intra_request *>(dev + off_req), 1,
                       reinterpret_cast<xmi_intra_result *>(dev + off_res), n, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(res, dev + off_res, sizeof(xmi_intra_result), hipMemcpyDeviceToHost));
    return XMI_OK;
}

extern "C" int xmi_intra_search_batch_dev(const uint8_t *d_src, int src_stride, const uint8_t *d_ref_samples,
                                          const xmi_intra_request *d_req, int n, xmi_intra_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_src && d_ref_samples && d_req && d_res, "xmi_intra_search_batch_dev: null argument");
    XMI_ARG(src_stride % 4 == 0 && ((uintptr_t)d_src & 3) == 0, "xmi_intra_search_batch_dev: source frame must be 4-byte aligned");
    return launch_search(d_src, src_stride, d_ref_samples, d_req, n, d_res, 4 | 8 | 16 | 32, stream);
}

extern "C" int xmi_intra_chroma_search_batch_dev(const uint8_t *d_src_u, const uint8_t *d_src_v, int src_stride, const uint8_t *d_ref_samples,
                                                 const xmi_chroma_request *d_req, int n, xmi_chroma_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_chroma_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_src_u && d_src_v && d_ref_samples && d_req && d_res, "xmi_intra_chroma_search_batch_dev: null argument");
    XMI_ARG(src_stride > 0 && src_stride % 4 == 0 && (((uintptr_t)d_src_u | (uintptr_t)d_src_v) & 3) == 0,
            "xmi_intra_chroma_search_batch_dev: chroma planes must be 4-byte aligned");
    ChromaParams p = { d_src_u, d_src_v, d_ref_samples, d_req, d_res, src_stride, n };
    hipLaunchKernelGGL(intra_chroma_search_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_intra_chroma_search(const uint8_t *src_u, const uint8_t *src_v, uint32_t src_stride, const uint8_t *ref_u, const uint8_t *ref_v,
                                       int size, int luma_mode, xmi_chroma_result *res)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(src_u && src_v && ref_u && ref_v && res, "xmi_intra_chroma_search: null argument");
    XMI_ARG(size == 4 || size == 8 || size == 16, "xmi_intra_chroma_search: size %d not in {4,8,16}", size);
    XMI_ARG(luma_mode >= 0 && luma_mode < 35, "xmi_intra_chroma_search: luma mode %d", luma_mode);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    // staging: U block | V block (rows `size` apart) | U samples | V samples | request | result
    const size_t blk = (size_t)size * size, off_ref = 512, off_req = off_ref + 2 * XMI_INTRA_REF_PITCH, off_res = off_req + 16;
    static_assert(sizeof(xmi_chroma_request) <= 16 && 512 + 2 * XMI_INTRA_REF_PITCH + 16 + sizeof(xmi_chroma_result) <= 2048, "staging layout");
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, 2048));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[512 + 2 * XMI_INTRA_REF_PITCH + 16];
    memset(host, 0, sizeof host);
    for (int y = 0; y < size; y++) {
        memcpy(host + (size_t)y * size, src_u + (size_t)y * src_stride, (size_t)size);
        memcpy(host + blk + (size_t)y * size, src_v + (size_t)y * src_stride, (size_t)size);
    }
    memcpy(host + off_ref, ref_u, (size_t)(4 * size + 1));
    memcpy(host + off_ref + XMI_INTRA_REF_PITCH, ref_v, (size_t)(4 * size + 1));
    xmi_chroma_request rq = { 0, 0, (uint8_t)size, (uint8_t)luma_mode, { 0, 0 } };
    memcpy(host + off_req, &rq, sizeof rq);
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    rc = xmi_intra_chroma_search_batch_dev(dev, dev + blk, size, dev + off_ref, reinterpret_cast<const xmi_chroma_request *>(dev + off_req), 1,
                                           reinterpret_cast<xmi_chroma_result *>(dev + off_res), nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(res, dev + off_res, sizeof(xmi_chroma_result), hipMemcpyDeviceToHost));
    return XMI_OK;
}
