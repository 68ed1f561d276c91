// SAO statistics for gfx950: calcSaoStatsCu (src/encode.cpp:364-622) for every CTU and component of a picture at once.
//
// After deblocking and before any offset is applied, the reference gathers per CTU, component and SAO type (edge
// offset classes EO_0..EO_3 = horizontal, vertical, 135 and 45 degree neighbours; band offset) the number of samples of
// each category and the sum of (source - reconstruction) over them; its serial parameter decision (rdoSaoUnitAll,
// src/encode.cpp:1348) then reads those tables. The inputs are frame-wide and fixed once the picture is deblocked, so
// all CTUs are done by one launch: one workgroup per (CTU, component), every thread classifies its samples for the
// five types and adds to per-wavefront histograms in LDS, which are summed at the end. Windows follow the reference:
// the bottom 4 (chroma 2) rows and right 5 (chroma 3) columns of a CTU are left out unless the CTU touches that picture
// border; edge classes skip samples whose neighbour lies outside the picture; EO_0 leaves the bottom rows out even in
// the last CTU row. Category of an edge sample: g_eoTable[sign(c - a) + sign(c - b) + 2] (src/encode.cpp:274), of a
// band sample 1 + (c >> 3).
#include "xmi_common.h"

namespace {

struct SaoParams {
    const uint8_t *rec[3], *org[3];
    xmi_sao_stats *out;
    int width, height, rec_stride[2], org_stride[2], wc;
};

constexpr int kBins = 5 * 33;

__device__ __forceinline__ int sgn(int x) { return (x > 0) - (x < 0); }

__global__ void __launch_bounds__(256) sao_stats_kernel(const SaoParams p)
{
    __shared__ int s_cnt[4][kBins], s_off[4][kBins];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 4 * kBins; i += 256) {
        (&s_cnt[0][0])[i] = 0;
        (&s_off[0][0])[i] = 0;
    }
    __syncthreads();
    const int comp = blockIdx.y, chroma = comp != 0;
    const int ctu = blockIdx.x, cx = ctu % p.wc, cy = ctu / p.wc;
    const int size = chroma ? 32 : 64, pw = p.width >> chroma, ph = p.height >> chroma;
    const int skip = chroma ? 2 : 4, skip_right = chroma ? 3 : 5;
    const int lpelx = cx * size, tpely = cy * size;
    const int rpelx = min(lpelx + size, pw), bpely = min(tpely + size, ph);
    const int lw = rpelx - lpelx, lh = bpely - tpely;
    const bool at_right = rpelx == pw, at_bottom = bpely == ph;
    const int rs = p.rec_stride[chroma], os = p.org_stride[chroma];
    const uint8_t *rec = p.rec[comp] + (size_t)tpely * rs + lpelx, *org = p.org[comp] + (size_t)tpely * os + lpelx;
    const int sx = lpelx == 0 ? 1 : 0, sy = tpely == 0 ? 1 : 0;
    const int ex_bo = at_right ? lw : lw - skip_right, ey_bo = at_bottom ? lh : lh - skip;
    const int ex_in = at_right ? lw - 1 : lw - skip_right, ey = at_bottom ? lh - 1 : lh - skip;
    const int eo_table = 1 | (2 << 4) | (0 << 8) | (3 << 12) | (4 << 16);      // g_eoTable, 4 bits per entry
    int *cnt = s_cnt[wave], *off = s_off[wave];
    auto add = [&](int type, int cls, int diff) {
        atomicAdd(&cnt[type * 33 + cls], 1);
        atomicAdd(&off[type * 33 + cls], diff);
    };
    for (int e = tid; e < lw * lh; e += 256) {
        const int y = e / lw, x = e - y * lw;
        const uint8_t *r = rec + (size_t)y * rs + x;
        const int c = r[0], diff = (int)org[(size_t)y * os + x] - c;
        if (x < ex_bo && y < ey_bo) add(4, 1 + (c >> 3), diff);
        auto eo = [&](int a, int b) { return (eo_table >> (4 * (sgn(c - a) + sgn(c - b) + 2))) & 15; };
        const bool x_in = x >= sx && x < ex_in, y_in = y >= sy && y < ey;
        if (x_in && y < lh - skip) add(0, eo(r[-1], r[1]), diff);
        if (y_in && x < ex_bo) add(1, eo(r[-rs], r[rs]), diff);
        if (x_in && y_in) {
            add(2, eo(r[-rs - 1], r[rs + 1]), diff);
            add(3, eo(r[-rs + 1], r[rs - 1]), diff);
        }
    }
    __syncthreads();
    xmi_sao_stats *o = p.out + (size_t)ctu * 3 + comp;
    for (int i = tid; i < kBins; i += 256) {
        (&o->count[0][0])[i] = s_cnt[0][i] + s_cnt[1][i] + s_cnt[2][i] + s_cnt[3][i];
        (&o->offset_org[0][0])[i] = s_off[0][i] + s_off[1][i] + s_off[2][i] + s_off[3][i];
    }
}

// ---- applying the decided offsets: processSaoUnitAll / processSaoCu (src/encode.cpp:1762-1904, 1524-1760) ----
// The reference filters a plane in place, CTU by CTU, and keeps the row above and the column to the left in side buffers
// so that every category is derived from samples BEFORE any offset; with separate source and destination planes that
// is simply "classify on src, write dst", one thread per sample.
struct SaoApplyParams {
    const uint8_t *src;
    uint8_t *dst;
    const xmi_sao_ctu_param *prm;
    int src_stride, dst_stride, width, height, ctu_log2, wc;
};

__global__ void __launch_bounds__(256) sao_apply_kernel(const SaoApplyParams p)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.width || y >= p.height) return;
    const xmi_sao_ctu_param q = p.prm[(y >> p.ctu_log2) * p.wc + (x >> p.ctu_log2)];
    const uint8_t *s = p.src + (size_t)y * p.src_stride + x;
    const int c = s[0];
    int v = c;
    if (q.type_idx == 4) {
        const int k = ((c >> 3) - q.band_position) & 31;
        if (k < 4) v = min(255, max(0, c + q.offset[k]));
    } else if (q.type_idx >= 0) {
        const int t = q.type_idx;
        const int dx = t == 1 ? 0 : t == 3 ? 1 : -1, dy = t == 0 ? 0 : -1;      // first neighbour; the second is its mirror
        const int xa = x + dx, ya = y + dy, xb = x - dx, yb = y - dy;
        if (min(xa, xb) >= 0 && max(xa, xb) < p.width && min(ya, yb) >= 0 && max(ya, yb) < p.height) {
            const int a = s[dy * p.src_stride + dx], b = s[-dy * p.src_stride - dx];
            const int eo_table = 1 | (2 << 4) | (0 << 8) | (3 << 12) | (4 << 16);
            const int cat = (eo_table >> (4 * (sgn(c - a) + sgn(c - b) + 2))) & 15;
            if (cat) v = min(255, max(0, c + q.offset[cat - 1]));
        }
    }
    p.dst[(size_t)y * p.dst_stride + x] = (uint8_t)v;
}

int check_geom(const char *who, const xmi_sao_geom *g)
{
    XMI_ARG(g, "%s: null geometry", who);
    XMI_ARG(g->width >= 16 && g->height >= 16 && g->width % 2 == 0 && g->height % 2 == 0, "%s: picture %dx%d", who, g->width, g->height);
    XMI_ARG(g->rec_stride_y >= g->width && g->org_stride_y >= g->width && g->rec_stride_c >= g->width / 2 && g->org_stride_c >= g->width / 2,
            "%s: strides do not cover the picture", who);
    return XMI_OK;
}

}  // namespace

extern "C" {

int xmi_sao_stats_dev(const xmi_sao_geom *g, const uint8_t *d_rec_y, const uint8_t *d_rec_u, const uint8_t *d_rec_v,
                      const uint8_t *d_org_y, const uint8_t *d_org_u, const uint8_t *d_org_v, xmi_sao_stats *d_stats, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    rc = check_geom("xmi_sao_stats_dev", g);
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_rec_y && d_rec_u && d_rec_v && d_org_y && d_org_u && d_org_v && d_stats, "xmi_sao_stats_dev: null argument");
    SaoParams p;
    p.rec[0] = d_rec_y;
    p.rec[1] = d_rec_u;
    p.rec[2] = d_rec_v;
    p.org[0] = d_org_y;
    p.org[1] = d_org_u;
    p.org[2] = d_org_v;
    p.out = d_stats;
    p.width = g->width;
    p.height = g->height;
    p.rec_stride[0] = g->rec_stride_y;
    p.rec_stride[1] = g->rec_stride_c;
    p.org_stride[0] = g->org_stride_y;
    p.org_stride[1] = g->org_stride_c;
    p.wc = (g->width + 63) / 64;
    const int nctu = p.wc * ((g->height + 63) / 64);
    hipLaunchKernelGGL(sao_stats_kernel, dim3(nctu, 3), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_sao_stats_frame(const xmi_sao_geom *g, const uint8_t *rec_y, const uint8_t *rec_u, const uint8_t *rec_v, const uint8_t *org_y,
                  const uint8_t *org_u, const uint8_t *org_v, xmi_sao_stats *stats)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    rc = check_geom("xmi_sao_stats_frame", g);
    if (rc != XMI_OK) return rc;
    XMI_ARG(rec_y && rec_u && rec_v && org_y && org_u && org_v && stats, "xmi_sao_stats_frame: null argument");
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
    const size_t w = (size_t)g->width, h = (size_t)g->height, luma = w * h, chroma = (w / 2) * (h / 2);
    const size_t nctu = ((w + 63) / 64) * ((h + 63) / 64);
    const size_t off_stats = (2 * (luma + 2 * chroma) + 15) & ~(size_t)15, total = off_stats + nctu * 3 * sizeof(xmi_sao_stats);
    uint8_t *d = static_cast<uint8_t *>(xmi::scratch(11, total));
    if (!d) return XMI_ERR_HIP;
    uint8_t *dr[3] = { d, d + luma, d + luma + chroma }, *dorg[3] = { d + luma + 2 * chroma, d + 2 * luma + 2 * chroma, d + 2 * luma + 3 * chroma };
    const uint8_t *hr[3] = { rec_y, rec_u, rec_v }, *ho[3] = { org_y, org_u, org_v };
    for (int c = 0; c < 3; c++) {
        const size_t pw = c ? w / 2 : w, ph = c ? h / 2 : h;
        XMI_HIP(hipMemcpy2D(dr[c], pw, hr[c], (size_t)(c ? g->rec_stride_c : g->rec_stride_y), pw, ph, hipMemcpyHostToDevice));
        XMI_HIP(hipMemcpy2D(dorg[c], pw, ho[c], (size_t)(c ? g->org_stride_c : g->org_stride_y), pw, ph, hipMemcpyHostToDevice));
    }
    const xmi_sao_geom dg = { g->width, g->height, g->width, g->width / 2, g->width, g->width / 2 };
    rc = xmi_sao_stats_dev(&dg, dr[0], dr[1], dr[2], dorg[0], dorg[1], dorg[2], reinterpret_cast<xmi_sao_stats *>(d + off_stats), nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(stats, d + off_stats, nctu * 3 * sizeof(xmi_sao_stats), hipMemcpyDeviceToHost));
    return XMI_OK;
}

int xmi_sao_apply_plane_dev(const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int width, int height, int ctu_size,
                            const xmi_sao_ctu_param *d_params, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_src && d_dst && d_params && d_src != d_dst, "xmi_sao_apply_plane_dev: null argument, or source and destination are the same plane");
    XMI_ARG(width > 0 && height > 0 && src_stride >= width && dst_stride >= width, "xmi_sao_apply_plane_dev: bad geometry %dx%d", width, height);
    XMI_ARG(ctu_size == 64 || ctu_size == 32, "xmi_sao_apply_plane_dev: CTU size %d (64 for luma, 32 for 4:2:0 chroma)", ctu_size);
    SaoApplyParams p = { d_src, d_dst, d_params, src_stride, dst_stride, width, height, ctu_size == 64 ? 6 : 5, (width + ctu_size - 1) / ctu_size };
    hipLaunchKernelGGL(sao_apply_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_sao_apply_plane(uint8_t *plane, int stride, int width, int height, int ctu_size, const xmi_sao_ctu_param *params)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(plane && params && width > 0 && height > 0 && stride >= width && (ctu_size == 64 || ctu_size == 32), "xmi_sao_apply_plane: bad argument");
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
    const size_t w = (size_t)width, h = (size_t)height, n = w * h;
    const size_t nctu = ((w + ctu_size - 1) / ctu_size) * ((h + ctu_size - 1) / ctu_size);
    const size_t off_prm = (2 * n + 15) & ~(size_t)15;
    uint8_t *d = static_cast<uint8_t *>(xmi::scratch(11, off_prm + nctu * sizeof(xmi_sao_ctu_param)));
    if (!d) return XMI_ERR_HIP;
    XMI_HIP(hipMemcpy2D(d, w, plane, (size_t)stride, w, h, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_prm, params, nctu * sizeof(xmi_sao_ctu_param), hipMemcpyHostToDevice));
    rc = xmi_sao_apply_plane_dev(d, width, d + n, width, width, height, ctu_size, reinterpret_cast<const xmi_sao_ctu_param *>(d + off_prm), nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy2D(plane, (size_t)stride, d + n, w, w, h, hipMemcpyDeviceToHost));
    return XMI_OK;
}

}  // extern "C"
