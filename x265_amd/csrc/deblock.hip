// In-loop deblocking filter of one reconstructed picture for gfx950: xDeblockFrame (src/Deblock.cpp:524-559).
//
// The reference walks the CTBs in raster order and filters, per CTB, its vertical edges and then its horizontal edges
// shifted 8 samples to the left (ff_hevc_deblocking_filter_CTB, src/Deblock.cpp:244-342) - an arrangement whose
// result is that of filtering every vertical edge of the picture first and every horizontal edge afterwards, because
// an edge segment only modifies 3 samples on either side and reads 4. That is what runs here, as two kernels:
//   vertical edges:   one thread per (edge x = 8, 16, .., 4-row segment): 4 rows x 8 samples as two dwords per row
//   horizontal edges: one thread per (4-column segment, edge y = 8, 16, ..): 8 rows x one dword
// Every dword a thread rewrites belongs to its own segment, so the planes are filtered in place without atomics.
// Decisions and arithmetic are the reference's (hevc_loop_filter_luma :61-145, hevc_loop_filter_chroma :147-168, tables
// :195-207, chroma tc :209-234): a segment is filtered when the boundary strength the CTU loop stored for it is
// non-zero; chroma (4:2:0) only on the 16-sample luma grid and only for strength 2.
#include "xmi_common.h"

namespace {

typedef unsigned int u32;

__constant__ uint8_t k_tc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                  5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ uint8_t k_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26,
                                    28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__constant__ uint8_t k_qpc[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };

struct DbParams {
    uint8_t *y, *u, *v;
    const uint8_t *vbs, *hbs, *qp;
    int width, height, stride_y, stride_c, tc_offset, beta_offset, cb_qp_offset, cr_qp_offset;
    // first sample column / row whose vertical edges (v_) and horizontal edges (h_) xDeblockFrame gets to: 0 for pictures of at
    // least two CTB rows and columns (see xmi_deblock_frame_dev)
    int v_x0, v_y0, h_x0, h_y0;
};

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(hi, max(lo, v)); }
__device__ __forceinline__ int clip8(int v) { return min(255, max(0, v)); }

// one luma segment: s[line][position], position 0..7 = P3 P2 P1 P0 Q0 Q1 Q2 Q3; decisions from lines 0 and 3
__device__ __forceinline__ void filter_luma(int (&s)[4][8], int beta, int tc)
{
    const int dp0 = abs(s[0][1] - 2 * s[0][2] + s[0][3]), dq0 = abs(s[0][6] - 2 * s[0][5] + s[0][4]);
    const int dp3 = abs(s[3][1] - 2 * s[3][2] + s[3][3]), dq3 = abs(s[3][6] - 2 * s[3][5] + s[3][4]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    const bool strong = abs(s[0][0] - s[0][3]) + abs(s[0][7] - s[0][4]) < beta_3 && abs(s[0][3] - s[0][4]) < tc25 &&
                        abs(s[3][0] - s[3][3]) + abs(s[3][7] - s[3][4]) < beta_3 && abs(s[3][3] - s[3][4]) < tc25 &&
                        (d0 << 1) < beta_2 && (d3 << 1) < beta_2;
    if (strong) {
        const int tc2 = tc << 1;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int p3 = s[d][0], p2 = s[d][1], p1 = s[d][2], p0 = s[d][3], q0 = s[d][4], q1 = s[d][5], q2 = s[d][6], q3 = s[d][7];
            s[d][3] = clip3(p0 - tc2, p0 + tc2, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            s[d][2] = clip3(p1 - tc2, p1 + tc2, (p2 + p1 + p0 + q0 + 2) >> 2);
            s[d][1] = clip3(p2 - tc2, p2 + tc2, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            s[d][4] = clip3(q0 - tc2, q0 + tc2, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            s[d][5] = clip3(q1 - tc2, q1 + tc2, (p0 + q0 + q1 + q2 + 2) >> 2);
            s[d][6] = clip3(q2 - tc2, q2 + tc2, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        }
    } else {
        const int tc_2 = tc >> 1, side = (beta + (beta >> 1)) >> 3;
        const bool two_p = dp0 + dp3 < side, two_q = dq0 + dq3 < side;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int p2 = s[d][1], p1 = s[d][2], p0 = s[d][3], q0 = s[d][4], q1 = s[d][5], q2 = s[d][6];
            int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta0) < 10 * tc) {
                delta0 = clip3(-tc, tc, delta0);
                s[d][3] = clip8(p0 + delta0);
                s[d][4] = clip8(q0 - delta0);
                if (two_p) s[d][2] = clip8(p1 + clip3(-tc_2, tc_2, (((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1));
                if (two_q) s[d][5] = clip8(q1 + clip3(-tc_2, tc_2, (((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1));
            }
        }
    }
}

__device__ __forceinline__ int chroma_tc(int qp_y, int offset, int tc_offset)
{
    const int qp_i = clip3(0, 57, qp_y + offset);
    const int qp = qp_i < 30 ? qp_i : qp_i > 43 ? qp_i - 6 : (int)k_qpc[qp_i - 30];
    return k_tc[clip3(0, 53, qp + 2 + tc_offset)];
}

// chroma: p1 p0 | q0 q1 of one line -> new p0, q0
__device__ __forceinline__ void filter_chroma(int p1, int &p0, int &q0, int q1, int tc)
{
    const int delta0 = clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
    p0 = clip8(p0 + delta0);
    q0 = clip8(q0 - delta0);
}

__device__ __forceinline__ void edge_strengths(const DbParams &p, int bs, int qp, int &beta, int &tc)
{
    beta = k_beta[clip3(0, 51, qp + ((p.beta_offset >> 1) << 1))];
    tc = k_tc[clip3(0, 53, qp + 2 * (bs - 1) + ((p.tc_offset >> 1) << 1))];
}

__global__ void __launch_bounds__(256) deblock_vertical_kernel(const DbParams p)
{
    const int ex = blockIdx.x * 64 + (threadIdx.x & 63) + 1, sy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = 8 * ex, y = 4 * sy;
    if (x >= p.width || y >= p.height || x < p.v_x0 || y < p.v_y0) return;
    const int bs = p.vbs[ex + sy * (p.width >> 3)];
    if (!bs) return;
    const int qw = p.width >> 2;
    const int qp = (p.qp[((x - 1) >> 2) + (y >> 2) * qw] + p.qp[(x >> 2) + (y >> 2) * qw] + 1) >> 1;
    int beta, tc;
    edge_strengths(p, bs, qp, beta, tc);
    uint8_t *row = p.y + (size_t)y * p.stride_y + x - 4;
    int s[4][8];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint2 w = *reinterpret_cast<const uint2 *>(row + (size_t)d * p.stride_y);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            s[d][i] = (w.x >> (8 * i)) & 255;
            s[d][4 + i] = (w.y >> (8 * i)) & 255;
        }
    }
    filter_luma(s, beta, tc);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint2 w;
        w.x = (u32)s[d][0] | ((u32)s[d][1] << 8) | ((u32)s[d][2] << 16) | ((u32)s[d][3] << 24);
        w.y = (u32)s[d][4] | ((u32)s[d][5] << 8) | ((u32)s[d][6] << 16) | ((u32)s[d][7] << 24);
        *reinterpret_cast<uint2 *>(row + (size_t)d * p.stride_y) = w;
    }
    if ((x & 15) == 0 && (y & 7) == 0 && bs == 2) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int tcc = chroma_tc(qp, c ? p.cr_qp_offset : p.cb_qp_offset, p.tc_offset);
            uint8_t *pl = (c ? p.v : p.u) + (size_t)(y >> 1) * p.stride_c + (x >> 1);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint8_t *q = pl + (size_t)d * p.stride_c;
                int p0 = q[-1], q0 = q[0];
                filter_chroma(q[-2], p0, q0, q[1], tcc);
                q[-1] = (uint8_t)p0;
                q[0] = (uint8_t)q0;
            }
        }
    }
}

__global__ void __launch_bounds__(256) deblock_horizontal_kernel(const DbParams p)
{
    const int x4 = blockIdx.x * 64 + (threadIdx.x & 63), ey = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
    const int x = 4 * x4, y = 8 * ey;
    if (x >= p.width || y >= p.height || x < p.h_x0 || y < p.h_y0) return;
    const int bs = p.hbs[(x + y * (p.width >> 3)) >> 2];
    if (!bs) return;
    const int qw = p.width >> 2;
    const int qp = (p.qp[(x >> 2) + ((y - 1) >> 2) * qw] + p.qp[(x >> 2) + (y >> 2) * qw] + 1) >> 1;
    int beta, tc;
    edge_strengths(p, bs, qp, beta, tc);
    uint8_t *col = p.y + (size_t)(y - 4) * p.stride_y + x;
    int s[4][8];      // line = column of the segment, position = row across the edge
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 w = *reinterpret_cast<const u32 *>(col + (size_t)i * p.stride_y);
#pragma unroll
        for (int d = 0; d < 4; d++) s[d][i] = (w >> (8 * d)) & 255;
    }
    filter_luma(s, beta, tc);
#pragma unroll
    for (int i = 1; i < 7; i++)
        *reinterpret_cast<u32 *>(col + (size_t)i * p.stride_y) = (u32)s[0][i] | ((u32)s[1][i] << 8) | ((u32)s[2][i] << 16) | ((u32)s[3][i] << 24);
    if ((x & 7) == 0 && (y & 15) == 0 && bs == 2) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int tcc = chroma_tc(qp, c ? p.cr_qp_offset : p.cb_qp_offset, p.tc_offset);
            uint8_t *pl = (c ? p.v : p.u) + (size_t)(y >> 1) * p.stride_c + (x >> 1);
            const ptrdiff_t st = p.stride_c;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                int p0 = pl[d - st], q0 = pl[d];
                filter_chroma(pl[d - 2 * st], p0, q0, pl[d + st], tcc);
                pl[d - st] = (uint8_t)p0;
                pl[d] = (uint8_t)q0;
            }
        }
    }
}

// ---- boundary strengths of a picture from picture-level maps ----
// What the reference's per-unit calls of ff_hevc_deblocking_boundary_strengths (src/Deblock.cpp:369-503) leave in
// vertical_bs / horizontal_bs, computed per edge segment instead of per unit: the segment looks up the unit it lies in
// (units are aligned to their size), takes that unit's top-left motion vector and cbf as "current" (as the reference
// does, :403, :409) and the 8x8 / 4x4 entry across the edge as neighbour. boundary_strength (:344-366): on a unit edge 2
// in I slices, else 1 if either cbf is set; otherwise (and on the 8-sample grid inside a unit, P slices only) 1 if the
// vectors differ by 4 or more in a component. Nothing on the picture's left and top border.
struct BsParams {
    const uint8_t *unit_log2, *cbf;
    const short2 *mv;
    uint8_t *vbs, *hbs;
    int width, height, intra;
};

__global__ void __launch_bounds__(256) deblock_strengths_kernel(const BsParams p)
{
    const int w4 = p.width >> 2, w8 = p.width >> 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= w8 * (p.height >> 2)) return;
    const bool vertical = blockIdx.y == 0;
    // vertical: entry ex + sy * w8 is the segment at x = 8 ex, y = 4 sy; horizontal: entry x4 + ey * w4 at x = 4 x4, y = 8 ey
    const int x = vertical ? 8 * (t % w8) : 4 * (t % w4), y = vertical ? 4 * (t / w8) : 8 * (t / w4);
    int bs = 0;
    const int lg = p.unit_log2[(y >> 2) * w4 + (x >> 2)];
    if ((vertical ? x : y) > 0 && lg >= 2) {
        const int size = 1 << min(lg, 30), x0 = x & ~(size - 1), y0 = y & ~(size - 1);
        const bool unit_edge = vertical ? x == x0 : y == y0;
        if (unit_edge || (lg > 2 && !p.intra)) {
            const int cur_cbf = p.cbf[(y0 >> 2) * w4 + (x0 >> 2)];
            const int nb_cbf = vertical ? p.cbf[(y >> 2) * w4 + (x >> 2) - 1] : p.cbf[((y >> 2) - 1) * w4 + (x >> 2)];
            if (unit_edge && p.intra) bs = 2;
            else if (unit_edge && (cur_cbf || nb_cbf)) bs = 1;
            else if (!p.intra) {
                const short2 cur = p.mv[(y0 >> 3) * w8 + (x0 >> 3)];
                const short2 nb = vertical ? p.mv[(y >> 3) * w8 + (x >> 3) - 1] : p.mv[((y >> 3) - 1) * w8 + (x >> 3)];
                bs = abs(nb.x - cur.x) >= 4 || abs(nb.y - cur.y) >= 4;
            }
        }
    }
    (vertical ? p.vbs : p.hbs)[t] = (uint8_t)bs;
}

// ---- staging without allocation ----
// xmi_deblock_strengths_static stages its maps and tables in this array instead of a buffer from hipMalloc. Every device
// the library's code object is loaded on holds its own copy at its own address; g_bs_static[device] is that address,
// resolved once per device binding. Host code never passes the array's name to a kernel or a copy.
constexpr size_t kBsStaticMaxSamples = (size_t)4096 * 2304;           // luma samples of the largest picture served
constexpr size_t kBsStaticBytes = kBsStaticMaxSamples / 4 + 256;      // 2 maps (1/16 each) + vectors (1/16) + 2 tables (1/32 each)
constexpr int kMaxDevices = 64;
static uint8_t *g_bs_static[kMaxDevices] = {};

}  // namespace

// No kernel names the array (they get its address as an argument); the host-side use below is what makes the compiler
// give the device-side symbol external linkage, which the runtime's lookup by name needs - so no 'used' attribute here,
// which would pin it as a local symbol.
[[maybe_unused]] static __device__ __align__(256) uint8_t scratch[kBsStaticBytes];

// called when the library binds `device`, with that device current
int xmi::deblock_prepare_device(int device)
{
    XMI_ARG(device >= 0 && device < kMaxDevices, "device index %d beyond %d", device, kMaxDevices - 1);
    void *ptr = nullptr;
    XMI_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(::scratch)));   // the array at file scope, not the function xmi::scratch
    g_bs_static[device] = static_cast<uint8_t *>(ptr);
    return XMI_OK;
}

extern "C" {

int xmi_deblock_strengths_dev(int width, int height, int intra_slice, const uint8_t *d_unit_log2, const uint8_t *d_cbf, const int16_t *d_mv,
                              uint8_t *d_vertical_bs, uint8_t *d_horizontal_bs, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(width >= 16 && height >= 16 && width % 8 == 0 && height % 8 == 0, "xmi_deblock_strengths_dev: picture %dx%d must be a multiple of 8, at least 16x16",
            width, height);
    XMI_ARG(d_unit_log2 && d_cbf && d_vertical_bs && d_horizontal_bs && (intra_slice || d_mv), "xmi_deblock_strengths_dev: null argument");
    XMI_ARG(((uintptr_t)d_mv & 3) == 0, "xmi_deblock_strengths_dev: motion vectors must be 4-byte aligned");
    BsParams p = { d_unit_log2, d_cbf, reinterpret_cast<const short2 *>(d_mv), d_vertical_bs, d_horizontal_bs, width, height, intra_slice ? 1 : 0 };
    const int n = (width / 8) * (height / 4);
    hipLaunchKernelGGL(deblock_strengths_kernel, dim3((n + 255) / 256, 2), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_deblock_strengths(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                          uint8_t *vertical_bs, uint8_t *horizontal_bs)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(width >= 16 && height >= 16 && width % 8 == 0 && height % 8 == 0, "xmi_deblock_strengths: picture %dx%d must be a multiple of 8, at least 16x16", width,
            height);
    XMI_ARG(unit_log2 && cbf && vertical_bs && horizontal_bs && (intra_slice || mv), "xmi_deblock_strengths: null argument");
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
    const size_t n4 = (size_t)(width / 4) * (height / 4), n8 = (size_t)(width / 8) * (height / 8), nbs = (size_t)(width / 8) * (height / 4);
    const size_t off_cbf = n4, off_mv = (2 * n4 + 15) & ~(size_t)15, off_v = off_mv + 4 * n8, off_h = off_v + nbs;
    uint8_t *d = static_cast<uint8_t *>(xmi::scratch(10, off_h + nbs));
    if (!d) return XMI_ERR_HIP;
    XMI_HIP(hipMemcpy(d, unit_log2, n4, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_cbf, cbf, n4, hipMemcpyHostToDevice));
    if (!intra_slice) XMI_HIP(hipMemcpy(d + off_mv, mv, 4 * n8, hipMemcpyHostToDevice));
    rc = xmi_deblock_strengths_dev(width, height, intra_slice, d, d + off_cbf, reinterpret_cast<const int16_t *>(d + off_mv), d + off_v, d + off_h, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(vertical_bs, d + off_v, nbs, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy(horizontal_bs, d + off_h, nbs, hipMemcpyDeviceToHost));
    return XMI_OK;
}

int xmi_deblock_strengths_static(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                                 uint8_t *vertical_bs, uint8_t *horizontal_bs)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(width >= 16 && height >= 16 && width % 8 == 0 && height % 8 == 0, "xmi_deblock_strengths_static: picture %dx%d must be a multiple of 8, at least 16x16",
            width, height);
    XMI_ARG((size_t)width * (size_t)height <= kBsStaticMaxSamples,
            "xmi_deblock_strengths_static: picture %dx%d exceeds the %zu luma samples (4096x2304) this entry point stages; use xmi_deblock_strengths", width,
            height, kBsStaticMaxSamples);
    XMI_ARG(unit_log2 && cbf && vertical_bs && horizontal_bs && (intra_slice || mv), "xmi_deblock_strengths_static: null argument");
    int dev = -1;
    XMI_HIP(hipGetDevice(&dev));
    XMI_ARG(dev >= 0 && dev < kMaxDevices && g_bs_static[dev], "xmi_deblock_strengths_static: device %d was not bound by xmi_init", dev);
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());   // one staging array per device: calls of a process take turns
    uint8_t *d = g_bs_static[dev];
    const size_t n4 = (size_t)(width / 4) * (height / 4), n8 = (size_t)(width / 8) * (height / 8), nbs = (size_t)(width / 8) * (height / 4);
    const size_t off_cbf = n4, off_mv = (2 * n4 + 15) & ~(size_t)15, off_v = off_mv + 4 * n8, off_h = off_v + nbs;
    if (off_h + nbs > kBsStaticBytes) {
        xmi::set_error("xmi_deblock_strengths_static: %zu staging bytes exceed the array", off_h + nbs);
        return XMI_ERR_ARG;
    }
    // every byte the kernel reads is copied in first and both tables are written in full: nothing of an earlier call survives
    XMI_HIP(hipMemcpy(d, unit_log2, n4, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_cbf, cbf, n4, hipMemcpyHostToDevice));
    if (!intra_slice) XMI_HIP(hipMemcpy(d + off_mv, mv, 4 * n8, hipMemcpyHostToDevice));
    rc = xmi_deblock_strengths_dev(width, height, intra_slice, d, d + off_cbf, reinterpret_cast<const int16_t *>(d + off_mv), d + off_v, d + off_h, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(vertical_bs, d + off_v, nbs, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy(horizontal_bs, d + off_h, nbs, hipMemcpyDeviceToHost));
    return XMI_OK;
}

int xmi_deblock_frame_dev(uint8_t *d_y, uint8_t *d_u, uint8_t *d_v, const xmi_deblock_params *prm, const uint8_t *d_vertical_bs,
                          const uint8_t *d_horizontal_bs, const uint8_t *d_qp_tab, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(d_y && d_u && d_v && prm && d_vertical_bs && d_horizontal_bs && d_qp_tab, "xmi_deblock_frame_dev: null argument");
    XMI_ARG(prm->width >= 16 && prm->height >= 16 && prm->width % 8 == 0 && prm->height % 8 == 0,
            "xmi_deblock_frame_dev: picture %dx%d must be a multiple of 8, at least 16x16", prm->width, prm->height);
    XMI_ARG(prm->stride_y >= prm->width && prm->stride_y % 4 == 0 && ((uintptr_t)d_y & 3) == 0 && prm->stride_c >= prm->width / 2,
            "xmi_deblock_frame_dev: luma plane and stride (%d) must be 4-byte aligned, strides must cover the picture", prm->stride_y);
    // xDeblockFrame filters CTB (x - 1, y - 1) when it comes to CTB (x, y) with x, y > 0 - plus the last column and row from
    // their neighbours - and the last CTB at the end (hls_filters, src/Deblock.cpp:512-522, :556-559): a picture with one CTB
    // row or one CTB column never gets there, and only its last CTB is filtered - its vertical edges from its left border
    // on, its horizontal edges from 8 samples further left (ff_hevc_deblocking_filter_CTB, :274, :310-311).
    const int cols = (prm->width + 63) / 64, rows = (prm->height + 63) / 64;
    int v_x0 = 0, v_y0 = 0, h_x0 = 0, h_y0 = 0;
    if (rows == 1 && cols > 1) v_x0 = (cols - 1) * 64, h_x0 = v_x0 - 8;
    if (cols == 1 && rows > 1) v_y0 = h_y0 = (rows - 1) * 64;
    DbParams p = { d_y, d_u, d_v, d_vertical_bs, d_horizontal_bs, d_qp_tab, prm->width, prm->height, prm->stride_y, prm->stride_c,
                   prm->tc_offset, prm->beta_offset, prm->cb_qp_offset, prm->cr_qp_offset, v_x0, v_y0, h_x0, h_y0 };
    hipStream_t st = xmi::as_stream(stream);
    const int nev = prm->width / 8 - 1, nsv = prm->height / 4;      // vertical edges x segments
    hipLaunchKernelGGL(deblock_vertical_kernel, dim3((nev + 63) / 64, (nsv + 3) / 4), dim3(256), 0, st, p);
    const int nsh = prm->width / 4, neh = prm->height / 8 - 1;      // segments x horizontal edges
    hipLaunchKernelGGL(deblock_horizontal_kernel, dim3((nsh + 63) / 64, (neh + 3) / 4), dim3(256), 0, st, p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_deblock_frame(uint8_t *y, uint8_t *u, uint8_t *v, const xmi_deblock_params *prm, const uint8_t *vertical_bs,
                      const uint8_t *horizontal_bs, const uint8_t *qp_tab)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(y && u && v && prm && vertical_bs && horizontal_bs && qp_tab, "xmi_deblock_frame: null argument");
    XMI_ARG(prm->width >= 16 && prm->height >= 16 && prm->width % 8 == 0 && prm->height % 8 == 0 && prm->stride_y >= prm->width &&
                prm->stride_c >= prm->width / 2,
            "xmi_deblock_frame: bad geometry %dx%d, strides %d / %d", prm->width, prm->height, prm->stride_y, prm->stride_c);
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
    // strength tables: one entry per 4-sample segment of the 8-sample edge grid (what xmi_deblock_strengths writes and the
    // kernels index); QP table: one entry per 4x4 block
    const size_t w = (size_t)prm->width, h = (size_t)prm->height, tab = (w / 4) * (h / 4), nbs = (w / 8) * (h / 4);
    const size_t sy = (w + 3) & ~(size_t)3, sc = w / 2;
    const size_t off_u = sy * h, off_v = off_u + sc * (h / 2), off_vbs = (off_v + sc * (h / 2) + 15) & ~(size_t)15;
    const size_t off_hbs = off_vbs + tab, off_qp = off_hbs + tab, total = off_qp + tab;
    uint8_t *d = static_cast<uint8_t *>(xmi::scratch(10, total));
    if (!d) return XMI_ERR_HIP;
    XMI_HIP(hipMemcpy2D(d, sy, y, (size_t)prm->stride_y, w, h, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy2D(d + off_u, sc, u, (size_t)prm->stride_c, w / 2, h / 2, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy2D(d + off_v, sc, v, (size_t)prm->stride_c, w / 2, h / 2, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_vbs, vertical_bs, nbs, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_hbs, horizontal_bs, nbs, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d + off_qp, qp_tab, tab, hipMemcpyHostToDevice));
    xmi_deblock_params dp = *prm;
    dp.stride_y = (int)sy;
    dp.stride_c = (int)sc;
    rc = xmi_deblock_frame_dev(d, d + off_u, d + off_v, &dp, d + off_vbs, d + off_hbs, d + off_qp, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy2D(y, (size_t)prm->stride_y, d, sy, w, h, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy2D(u, (size_t)prm->stride_c, d + off_u, sc, w / 2, h / 2, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy2D(v, (size_t)prm->stride_c, d + off_v, sc, w / 2, h / 2, hipMemcpyDeviceToHost));
    return XMI_OK;
}

}  // extern "C"
