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
constexpr int kEoBins = 4 * 5;                 // histogram rows actually touched: 4 edge types x 5 categories ...
constexpr int kRows = kEoBins + 32;            // ... and the 32 bands

__device__ __forceinline__ int sgn(int x) { return min(max(x, -1), 1); }      // one v_med3_i32

// Every lane keeps a histogram of its own in LDS ([row][lane], so a wavefront's accesses to one row fall on distinct
// banks and nothing is contended, however uniform the picture is). One 32-bit word holds both statistics of a row:
// the sample count in the low half - a lane sees at most 64*64/256 = 16 samples - and the sum of (source -
// reconstruction) in the high half (|sum| <= 16 * 255); the low half never carries into the high one.
__global__ void __launch_bounds__(256) sao_stats_kernel(const SaoParams p)
{
    __shared__ int s_hist[kRows][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int r = 0; r < kRows; r++) s_hist[r][tid] = 0;
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
    auto classify = [&](int x, int y, int c, int diff, int l0, int r0, int up, int dn, int ul, int dr, int ur, int dl) {
        const int inc = (int)((unsigned)diff << 16) + 1;
        if (x < ex_bo && y < ey_bo) s_hist[kEoBins + (c >> 3)][tid] += inc;
        auto eo = [&](int a, int b) { return (eo_table >> (4 * (sgn(c - a) + sgn(c - b) + 2))) & 15; };
        const bool x_in = x >= sx && x < ex_in, y_in = y >= sy && y < ey;
        if (x_in && y < lh - skip) s_hist[0 * 5 + eo(l0, r0)][tid] += inc;
        if (y_in && x < ex_bo) s_hist[1 * 5 + eo(up, dn)][tid] += inc;
        if (x_in && y_in) {
            s_hist[2 * 5 + eo(ul, dr)][tid] += inc;
            s_hist[3 * 5 + eo(ur, dl)][tid] += inc;
        }
    };
    const bool dwords = ((((uintptr_t)p.rec[comp] | (uintptr_t)p.org[comp]) & 3) == 0) && ((rs | os | pw) & 3) == 0;
    if (dwords) {
        // four samples of a row per step: the three rows around them as dwords, left and right neighbours cut out of the
        // dwords beside them. Dwords that would lie outside the picture are not loaded (their samples are never used:
        // the edge classes skip samples whose neighbour is outside).
        typedef unsigned int u32;
        const int qw = lw >> 2;
        for (int e = tid; e < qw * lh; e += 256) {
            const int y = e / qw, x0 = (e - y * qw) * 4;
            const int gx = lpelx + x0, gy = tpely + y;
            u32 cw[3], lf[3], rt[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const bool row_ok = gy + k - 1 >= 0 && gy + k - 1 < ph;
                const u32 *r = reinterpret_cast<const u32 *>(rec + (ptrdiff_t)(y + k - 1) * rs + x0);
                cw[k] = row_ok ? r[0] : 0u;
                const u32 lw4 = (row_ok && gx >= 4) ? r[-1] : 0u, rw4 = (row_ok && gx + 8 <= pw) ? r[1] : 0u;
                lf[k] = __builtin_amdgcn_alignbyte(cw[k], lw4, 3);
                rt[k] = __builtin_amdgcn_alignbyte(rw4, cw[k], 1);
            }
            const u32 o4 = *reinterpret_cast<const u32 *>(org + (size_t)y * os + x0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                auto b = [&](u32 w) { return (int)((w >> (8 * i)) & 255); };
                const int c = b(cw[1]);
                classify(x0 + i, y, c, b(o4) - c, b(lf[1]), b(rt[1]), b(cw[0]), b(cw[2]), b(lf[0]), b(rt[2]), b(rt[0]), b(lf[2]));
            }
        }
    } else {
        for (int e = tid; e < lw * lh; e += 256) {
            const int y = e / lw, x = e - y * lw;
            const uint8_t *r = rec + (size_t)y * rs + x;
            const int c = r[0];
            const bool x_in = x >= sx && x < ex_in, y_in = y >= sy && y < ey;      // neighbours are read only where they exist
            const int l0 = x_in ? r[-1] : 0, r0 = x_in ? r[1] : 0, up = y_in ? r[-rs] : 0, dn = y_in ? r[rs] : 0;
            const bool both = x_in && y_in;
            classify(x, y, c, (int)org[(size_t)y * os + x] - c, l0, r0, up, dn, both ? r[-rs - 1] : 0, both ? r[rs + 1] : 0,
                     both ? r[-rs + 1] : 0, both ? r[rs - 1] : 0);
        }
    }
    __syncthreads();
    // rows are shared out over the wavefronts; a lane adds its four columns, the wavefront adds its lanes
    xmi_sao_stats *o = p.out + (size_t)ctu * 3 + comp;
    for (int i = tid; i < kBins; i += 256) {      // entries no sample can fall into: edge categories 5.., band 0
        const int type = i / 33, cls = i - type * 33;
        if (type < 4 ? cls >= 5 : cls == 0) {
            (&o->count[0][0])[i] = 0;
            (&o->offset_org[0][0])[i] = 0;
        }
    }
    for (int r = wave; r < kRows; r += 4) {
        int cnt = 0, sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = s_hist[r][lane + 64 * k];
            cnt += v & 0xffff;
            sum += v >> 16;
        }
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            cnt += __shfl_xor(cnt, k);
            sum += __shfl_xor(sum, k);
        }
        if (lane == 0) {
            const int type = r < kEoBins ? r / 5 : 4, cls = r < kEoBins ? r % 5 : 1 + (r - kEoBins);
            o->count[type][cls] = cnt;
            o->offset_org[type][cls] = sum;
        }
    }
}

// ---- applying the decided offsets: processSaoUnitAll / processSaoCu (src/encode.cpp:1762-1904, 1524-1760) ----
// The reference filters a plane in place, CTU by CTU, and keeps the row above and the column to the left in side buffers
// so that every category is derived from samples BEFORE any offset; with separate source and destination planes that
// is simply "classify on src, write dst".
struct SaoApplyParams {
    const uint8_t *src;
    uint8_t *dst;
    const xmi_sao_ctu_param *prm;
    int src_stride, dst_stride, width, height, ctu_log2, wc;
};

__device__ __forceinline__ int sao_sample(const xmi_sao_ctu_param &q, int c, int a, int b)
{
    const int eo_table = 1 | (2 << 4) | (0 << 8) | (3 << 12) | (4 << 16);
    const int cat = (eo_table >> (4 * (sgn(c - a) + sgn(c - b) + 2))) & 15;
    return cat ? min(255, max(0, c + q.offset[cat - 1])) : c;
}

__device__ __forceinline__ int sao_band(const xmi_sao_ctu_param &q, int c)
{
    const int k = ((c >> 3) - q.band_position) & 31;
    return k < 4 ? min(255, max(0, c + q.offset[k])) : c;
}

// Four neighbouring samples per thread. Away from the picture border and with 4-byte aligned planes they travel as
// dwords (the diagonal and horizontal neighbours are cut out of the dwords to the left and right with v_alignbyte);
// border threads and unaligned planes take the sample-by-sample form of the same rules.
__global__ void __launch_bounds__(256) sao_apply_kernel(const SaoApplyParams p)
{
    typedef unsigned int u32;
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.width || y >= p.height) return;
    const xmi_sao_ctu_param q = p.prm[(y >> p.ctu_log2) * p.wc + (x >> p.ctu_log2)];      // 4 | CTU size: one CTU for all four
    const uint8_t *s = p.src + (size_t)y * p.src_stride + x;
    uint8_t *d = p.dst + (size_t)y * p.dst_stride + x;
    const int t = q.type_idx;
    const int dx = t == 1 ? 0 : t == 3 ? 1 : -1, dy = t == 0 ? 0 : -1;      // first neighbour; the second is its mirror
    const bool aligned = ((((uintptr_t)p.src | (uintptr_t)p.dst) & 3) == 0) && ((p.src_stride | p.dst_stride) & 3) == 0;
    if (aligned && x >= 4 && x + 8 <= p.width && y >= 1 && y + 1 < p.height) {
        const u32 c4 = *reinterpret_cast<const u32 *>(s);
        u32 out = c4;
        if (t == 4) {
            out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) out |= (u32)sao_band(q, (c4 >> (8 * i)) & 255) << (8 * i);
        } else if (t >= 0) {
            auto row4 = [&](int ry, int rdx) -> u32 {      // samples x+rdx .. x+rdx+3 of row y+ry
                const u32 *r = reinterpret_cast<const u32 *>(s + (ptrdiff_t)ry * p.src_stride);
                return rdx == 0 ? r[0] : rdx < 0 ? __builtin_amdgcn_alignbyte(r[0], r[-1], 3) : __builtin_amdgcn_alignbyte(r[1], r[0], 1);
            };
            const u32 a4 = row4(dy, dx), b4 = row4(-dy, -dx);
            out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) out |= (u32)sao_sample(q, (c4 >> (8 * i)) & 255, (a4 >> (8 * i)) & 255, (b4 >> (8 * i)) & 255) << (8 * i);
        }
        *reinterpret_cast<u32 *>(d) = out;
        return;
    }
    for (int i = 0; i < 4 && x + i < p.width; i++) {
        const int c = s[i];
        int v = c;
        if (t == 4) {
            v = sao_band(q, c);
        } else if (t >= 0) {
            const int xa = x + i + dx, ya = y + dy, xb = x + i - dx, yb = y - dy;
            if (min(xa, xb) >= 0 && max(xa, xb) < p.width && min(ya, yb) >= 0 && max(ya, yb) < p.height)
                v = sao_sample(q, c, s[i + dy * p.src_stride + dx], s[i - dy * p.src_stride - dx]);
        }
        d[i] = (uint8_t)v;
    }
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
    int rc = xmi::enter_device();
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
    hipLaunchKernelGGL(sao_apply_kernel, dim3((width + 255) / 256, (height + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_sao_apply_plane(uint8_t *plane, int stride, int width, int height, int ctu_size, const xmi_sao_ctu_param *params)
{
    int rc = xmi::enter_device();
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
