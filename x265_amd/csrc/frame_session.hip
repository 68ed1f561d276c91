// Frame session: the host-pointer batch interface an encoder's CTU loop would call (no device pointers, no
// streams on the caller's side). The session owns the device copies of one picture pair - source luma, padded
// previous reconstruction and its 16 quarter-pel planes - and staging buffers for request and result tables.
//   xmi_frame_begin          once per P picture: upload source + padded reconstruction, build the planes
//                            (what xPredInterLumaBlk, src/pixel.cpp:8745, would compute per candidate)
//   xmi_frame_sad_surface    optional: the SAD surface of the pair; motion searches then look integer SADs up
//   xmi_frame_planes_host    the 16 planes back to host memory (a CPU search can then read predictions instead of filtering)
//   xmi_frame_me_search      a batch of xMotionSearch calls (src/encode.cpp:2815), host request/result tables
//   xmi_frame_intra_search   a batch of 35-mode intra searches (src/encode.cpp:4005-4081)
//   xmi_frame_tu_chain       a batch of xSubDctQuantRec chains (src/encode.cpp:3005), host arrays
// Every call is synchronous on the library's own stream; a session is used by one thread at a time.
#include "xmi_common.h"

#include <new>

struct xmi_frame {
    int width, height, pad, stride, rows;
    uint8_t *d_cur = nullptr, *d_ref = nullptr, *d_planes = nullptr;
    void *d_surf[4] = { nullptr, nullptr, nullptr, nullptr };   // SAD surface planes (8x8, 16x16, 32x32, 64x64), allocated on first use
    int surf_alloc_range = 0;         // range the planes were sized for
    int surf_range = 0;               // range of the surface computed for the current picture; 0 = none
    uint8_t *d_stage = nullptr;       // request / result / block staging, grown on demand
    size_t stage_bytes = 0;
    hipStream_t stream = nullptr;
    bool has_picture = false;
};

namespace {

int stage(xmi_frame *f, size_t bytes)
{
    if (f->stage_bytes >= bytes) return XMI_OK;
    if (f->d_stage) (void)hipFree(f->d_stage);
    f->d_stage = nullptr;
    f->stage_bytes = 0;
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    XMI_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_stage), want));
    f->stage_bytes = want;
    return XMI_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int xmi_frame_create(int width, int height, int pad, xmi_frame **out)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(out, "xmi_frame_create: null argument");
    XMI_ARG(width > 0 && height > 0 && width % 4 == 0, "xmi_frame_create: %dx%d: width must be a positive multiple of 4", width, height);
    XMI_ARG(pad >= 8 && pad % 4 == 0, "xmi_frame_create: pad %d must be a multiple of 4, at least 8", pad);
    xmi_frame *f = new (std::nothrow) xmi_frame;
    XMI_ARG(f, "xmi_frame_create: out of host memory");
    f->width = width;
    f->height = height;
    f->pad = pad;
    f->stride = width + 2 * pad;
    f->rows = height + 2 * pad;
    const size_t plane = (size_t)f->rows * f->stride;
    if (hipMalloc(reinterpret_cast<void **>(&f->d_cur), (size_t)width * height) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_ref), plane) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_planes), 16 * plane) != hipSuccess ||
        hipStreamCreate(&f->stream) != hipSuccess) {
        xmi::set_error("xmi_frame_create: device allocation for %dx%d (+%d) failed", width, height, pad);
        xmi_frame_destroy(f);
        return XMI_ERR_HIP;
    }
    // the interpolation kernel leaves the 4 outermost samples of every plane unwritten; xmi_frame_planes_host copies whole planes
    if (hipMemsetAsync(f->d_planes, 0, 16 * plane, f->stream) != hipSuccess) {
        xmi::set_error("xmi_frame_create: clearing the phase planes failed");
        xmi_frame_destroy(f);
        return XMI_ERR_HIP;
    }
    *out = f;
    return XMI_OK;
}

int xmi_frame_destroy(xmi_frame *f)
{
    if (!f) return XMI_OK;
    if (f->stream) (void)hipStreamDestroy(f->stream);
    if (f->d_cur) (void)hipFree(f->d_cur);
    if (f->d_ref) (void)hipFree(f->d_ref);
    if (f->d_planes) (void)hipFree(f->d_planes);
    if (f->d_stage) (void)hipFree(f->d_stage);
    for (void *q : f->d_surf)
        if (q) (void)hipFree(q);
    delete f;
    return XMI_OK;
}

int xmi_frame_begin(xmi_frame *f, const uint8_t *cur, int cur_stride, const uint8_t *ref_padded, int ref_stride)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f && cur && ref_padded, "xmi_frame_begin: null argument");
    XMI_ARG(cur_stride >= f->width && ref_stride >= f->stride, "xmi_frame_begin: strides %d / %d too small for %dx%d (+%d)", cur_stride,
            ref_stride, f->width, f->height, f->pad);
    XMI_HIP(hipMemcpy2DAsync(f->d_cur, (size_t)f->width, cur, (size_t)cur_stride, (size_t)f->width, (size_t)f->height,
                             hipMemcpyHostToDevice, f->stream));
    XMI_HIP(hipMemcpy2DAsync(f->d_ref, (size_t)f->stride, ref_padded, (size_t)ref_stride, (size_t)f->stride, (size_t)f->rows,
                             hipMemcpyHostToDevice, f->stream));
    f->surf_range = 0;
    int rc = xmi_interp_planes_dev(f->d_ref, f->stride, f->rows, f->stride, f->d_planes, f->stream);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipStreamSynchronize(f->stream));
    f->has_picture = true;
    return XMI_OK;
}

int xmi_frame_sad_surface(xmi_frame *f, int range)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f && f->has_picture, "xmi_frame_sad_surface: no picture (call xmi_frame_begin first)");
    XMI_ARG(f->width % 64 == 0 && f->height % 64 == 0, "xmi_frame_sad_surface: %dx%d is not a multiple of the 64x64 CTU", f->width, f->height);
    XMI_ARG(range == 16 || range == 32 || range == 64 || range == 128, "xmi_frame_sad_surface: range %d not in {16,32,64,128}", range);
    XMI_ARG(f->pad >= range + 4, "xmi_frame_sad_surface: pad %d is smaller than range + 4 = %d", f->pad, range + 4);
    const xmi_sad_geom g = { f->width, f->height, f->width, f->stride, range };
    if (f->surf_alloc_range != range) {
        for (void *&q : f->d_surf) {
            if (q) (void)hipFree(q);
            q = nullptr;
        }
        f->surf_alloc_range = 0;
        for (int level = 0; level < 4; level++)
            XMI_HIP(hipMalloc(&f->d_surf[level], xmi_sad_surface_elems(&g, level) * (level < 2 ? 2 : 4)));
        f->surf_alloc_range = range;
    }
    f->surf_range = 0;
    int rc = xmi_sad_surface_dev(&g, f->d_cur, f->d_ref + (size_t)f->pad * f->stride + f->pad, static_cast<uint16_t *>(f->d_surf[0]),
                                 static_cast<uint16_t *>(f->d_surf[1]), static_cast<uint32_t *>(f->d_surf[2]),
                                 static_cast<uint32_t *>(f->d_surf[3]), 0, (f->width / 64) * (f->height / 64), f->stream);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipStreamSynchronize(f->stream));
    f->surf_range = range;
    return XMI_OK;
}

int xmi_frame_planes_host(xmi_frame *f, uint8_t *planes)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f && f->has_picture && planes, "xmi_frame_planes_host: no picture or null destination");
    XMI_HIP(hipMemcpyAsync(planes, f->d_planes, 16 * (size_t)f->rows * f->stride, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipStreamSynchronize(f->stream));
    return XMI_OK;
}

int xmi_frame_me_search(xmi_frame *f, const xmi_me_request *req, int n, int me_range, xmi_me_result *res)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f && f->has_picture, "xmi_frame_me_search: no picture (call xmi_frame_begin first)");
    XMI_ARG(n >= 0 && (n == 0 || (req && res)), "xmi_frame_me_search: bad request table");
    if (n == 0) return XMI_OK;
    for (int i = 0; i < n; i++)      // the kernel trusts positions: check them where it is cheap
        XMI_ARG((req[i].size == 8 || req[i].size == 16 || req[i].size == 32 || req[i].size == 64) && req[i].x0 >= 0 && req[i].y0 >= 0 &&
                    req[i].x0 % 4 == 0 && req[i].x0 + req[i].size <= f->width && req[i].y0 + req[i].size <= f->height,
                "xmi_frame_me_search: request %d (%d,%d size %d) outside the %dx%d picture", i, req[i].x0, req[i].y0, req[i].size, f->width, f->height);
    for (int i = 0; i < n; i++)
        XMI_ARG(xmi::me_request_reach(req[i]) + 8 <= f->pad, "xmi_frame_me_search: request %d can read %d samples away from its PU; the border is %d", i,
                xmi::me_request_reach(req[i]), f->pad);
    const size_t rq_bytes = align256((size_t)n * sizeof(xmi_me_request)), rs_bytes = (size_t)n * sizeof(xmi_me_result);
    int rc = stage(f, rq_bytes + rs_bytes);
    if (rc != XMI_OK) return rc;
    xmi_me_request *d_req = reinterpret_cast<xmi_me_request *>(f->d_stage);
    xmi_me_result *d_res = reinterpret_cast<xmi_me_result *>(f->d_stage + rq_bytes);
    XMI_HIP(hipMemcpyAsync(d_req, req, (size_t)n * sizeof(xmi_me_request), hipMemcpyHostToDevice, f->stream));
    if (f->surf_range) {
        const xmi_sad_geom g = { f->width, f->height, f->width, f->stride, f->surf_range };
        rc = xmi_me_search_surface_batch_dev(f->d_cur, f->width, f->d_planes, f->stride, f->rows, f->pad, &g,
                                             static_cast<const uint16_t *>(f->d_surf[0]), static_cast<const uint16_t *>(f->d_surf[1]),
                                             static_cast<const uint32_t *>(f->d_surf[2]), static_cast<const uint32_t *>(f->d_surf[3]), d_req, n,
                                             me_range, d_res, f->stream);
    } else {
        rc = xmi_me_search_batch_dev(f->d_cur, f->width, f->d_planes, f->stride, f->rows, f->pad, d_req, n, me_range, d_res, f->stream);
    }
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpyAsync(res, d_res, rs_bytes, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipStreamSynchronize(f->stream));
    return XMI_OK;
}

int xmi_frame_intra_search(xmi_frame *f, const uint8_t *ref_samples, const xmi_intra_request *req, int n, xmi_intra_result *res)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f && f->has_picture, "xmi_frame_intra_search: no picture (call xmi_frame_begin first)");
    XMI_ARG(n >= 0 && (n == 0 || (req && res && ref_samples)), "xmi_frame_intra_search: bad request table");
    if (n == 0) return XMI_OK;
    for (int i = 0; i < n; i++)
        XMI_ARG((req[i].size == 4 || req[i].size == 8 || req[i].size == 16 || req[i].size == 32) && req[i].x0 >= 0 && req[i].y0 >= 0 &&
                    req[i].x0 % 4 == 0 && req[i].x0 + req[i].size <= f->width && req[i].y0 + req[i].size <= f->height,
                "xmi_frame_intra_search: request %d (%d,%d size %d) outside the %dx%d picture", i, req[i].x0, req[i].y0, (int)req[i].size, f->width,
                f->height);
    const size_t rq_bytes = align256((size_t)n * sizeof(xmi_intra_request)), rf_bytes = align256((size_t)n * XMI_INTRA_REF_PITCH);
    const size_t rs_bytes = (size_t)n * sizeof(xmi_intra_result);
    int rc = stage(f, rq_bytes + rf_bytes + rs_bytes);
    if (rc != XMI_OK) return rc;
    xmi_intra_request *d_req = reinterpret_cast<xmi_intra_request *>(f->d_stage);
    uint8_t *d_refs = f->d_stage + rq_bytes;
    xmi_intra_result *d_res = reinterpret_cast<xmi_intra_result *>(f->d_stage + rq_bytes + rf_bytes);
    XMI_HIP(hipMemcpyAsync(d_req, req, (size_t)n * sizeof(xmi_intra_request), hipMemcpyHostToDevice, f->stream));
    XMI_HIP(hipMemcpyAsync(d_refs, ref_samples, (size_t)n * XMI_INTRA_REF_PITCH, hipMemcpyHostToDevice, f->stream));
    rc = xmi_intra_search_batch_dev(f->d_cur, f->width, d_refs, d_req, n, d_res, f->stream);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpyAsync(res, d_res, rs_bytes, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipStreamSynchronize(f->stream));
    return XMI_OK;
}

int xmi_frame_tu_chain(xmi_frame *f, int n, int count, const uint8_t *pix, const uint8_t *pred, const uint8_t *qp, const uint8_t *flags,
                       int16_t *level, uint8_t *rec, uint32_t *sum)
{
    if (int rc0 = xmi::enter_device()) return rc0;   // also makes the library's device current for this thread
    XMI_ARG(f, "xmi_frame_tu_chain: null session");
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32, "xmi_frame_tu_chain: transform size %d not in {4,8,16,32}", n);
    XMI_ARG(count >= 0 && (count == 0 || (pix && pred && qp && flags && level && rec && sum)), "xmi_frame_tu_chain: bad arguments");
    if (count == 0) return XMI_OK;
    const size_t px = align256((size_t)count * n * n), tu = align256((size_t)count);
    // layout: pix | pred | rec | level (2 px) | qp | flags | sum (4 tu)
    int rc = stage(f, 5 * px + 2 * tu + 4 * tu);
    if (rc != XMI_OK) return rc;
    uint8_t *d_pix = f->d_stage, *d_pred = d_pix + px, *d_rec = d_pred + px;
    int16_t *d_level = reinterpret_cast<int16_t *>(d_rec + px);
    uint8_t *d_qp = d_rec + 3 * px, *d_fl = d_qp + tu;
    uint32_t *d_sum = reinterpret_cast<uint32_t *>(d_fl + tu);
    const size_t nn = (size_t)count * n * n;
    XMI_HIP(hipMemcpyAsync(d_pix, pix, nn, hipMemcpyHostToDevice, f->stream));
    XMI_HIP(hipMemcpyAsync(d_pred, pred, nn, hipMemcpyHostToDevice, f->stream));
    XMI_HIP(hipMemcpyAsync(d_qp, qp, (size_t)count, hipMemcpyHostToDevice, f->stream));
    XMI_HIP(hipMemcpyAsync(d_fl, flags, (size_t)count, hipMemcpyHostToDevice, f->stream));
    rc = xmi_tu_chain_batch_dev(n, count, d_pix, d_pred, d_qp, d_fl, d_level, d_rec, d_sum, f->stream);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpyAsync(level, d_level, nn * 2, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipMemcpyAsync(rec, d_rec, nn, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipMemcpyAsync(sum, d_sum, (size_t)count * 4, hipMemcpyDeviceToHost, f->stream));
    XMI_HIP(hipStreamSynchronize(f->stream));
    return XMI_OK;
}

}  // extern "C"
