// Inter prediction blocks for gfx950: xPredInterLumaBlk (src/pixel.cpp:8745-8795) and xPredInterChromaBlk
// (src/pixel.cpp:9657-9703) for a batch of PUs, computed from the reconstruction planes on the device.
//
// One wavefront per (request, plane). The block's filter window is staged in LDS; then
//   no fraction:        copy
//   x fraction only:    clip((sum_h + 32) >> 6)                    src/pixel.cpp:7705 (luma), :8982 (chroma)
//   y fraction only:    clip((sum_v + 32) >> 6)                    :8137, :9257
//   both:               first stage sum_h - 8192 kept in 16 bits, second stage (sum_v' + 8192*64 + 2048) >> 12
//                       (:8690, :9611). Because the taps add up to 64 this equals (sum_v of the unbiased
//                       first-stage sums + 2048) >> 12, which is what is computed (the sums fit 16 bits).
// Filters: xg_lumaFilter (src/table.cpp:525-530), xg_chromaFilter (:532-541). Only the samples the reference's
// filters read are loaded (no extra columns when the x fraction is 0, no extra rows when the y fraction is 0).
//
// The per-call forms at the end take host pointers with the reference's argument meaning, stage the window and
// run the same kernel on one request.
#include "xmi_common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace {

__constant__ int k_luma[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                  { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int k_chroma[8][8] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

struct PredParams {
    const uint8_t *ref[3];   // pixel (0,0) of Y, U, V
    uint8_t *pred[3];
    const xmi_inter_pred_request *req;
    int stride_y, stride_c, n;
};

// One wavefront per block and plane: a picture has tens of thousands of small blocks, and a 16x16 luma block (an 8x8
// chroma block) does not feed more lanes than that. Two instances share the requests by size - MAXBS 16 takes the
// 8x8 and 16x16 PUs, MAXBS 64 the 32x32 and 64x64 ones - because the LDS a workgroup reserves decides how many of these
// short-lived workgroups a CU holds at once: 15 KB for a 64x64 window against 1.4 KB for a 16x16 one (a 1080p picture
// of 16x16 PUs: 27 us with one 256-thread, 15 KB workgroup per block, 15 us this way).
constexpr int kPredThreads = 64;

__device__ __forceinline__ int clip8(int v) { return min(255, max(0, v)); }

template <int MAXBS>
struct PredLds {
    static constexpr int kRows = MAXBS + 7;                          // window rows: block + 7 (8-tap filter)
    static constexpr int kPitch = (MAXBS + 7 + 3 + 3) & ~3;          // bytes per window row: samples + alignment, a multiple of 4
    uint8_t win[kRows + 1][kPitch];
    int16_t h[kRows][MAXBS];
};

// TAPS = 8 (luma, quarter-pel) or 4 (chroma, eighth-pel)
template <int TAPS, int MAXBS>
__device__ void predict_block(const uint8_t *ref, int stride, uint8_t *dst, int bs, int xf, int yf, PredLds<MAXBS> &lds)
{
    constexpr int B = TAPS / 2 - 1;                     // samples before the block the filter reaches
    constexpr int PITCH = PredLds<MAXBS>::kPitch;
    const int (*filt)[8] = TAPS == 8 ? k_luma : k_chroma;
    const int tid = threadIdx.x;
    const int c0 = xf ? -B : 0, c1 = xf ? bs + TAPS / 2 : bs;     // window columns [c0, c1) relative to the block
    const int r0 = yf ? -B : 0, r1 = yf ? bs + TAPS / 2 : bs;
    const int wc = c1 - c0, wr = r1 - r0;
    const uint8_t *org = ref + (ptrdiff_t)r0 * stride + c0;          // first sample of the window
    const int sh = (int)(reinterpret_cast<uintptr_t>(org) & 3);
    if ((stride & 3) == 0) {
        // the window as aligned dwords: every row starts sh bytes before its first sample (rows stay aligned because the
        // stride is a multiple of 4); the row pitch in LDS is a multiple of 4 as well
        const int wd = (wc + sh + 3) >> 2;
        for (int e = tid; e < wd * wr; e += kPredThreads) {
            const int r = e / wd, d = e - r * wd;
            *reinterpret_cast<unsigned *>(&lds.win[r][4 * d]) = *reinterpret_cast<const unsigned *>(org - sh + (ptrdiff_t)r * stride + 4 * d);
        }
    } else {
        for (int e = tid; e < wc * wr; e += kPredThreads) {
            const int r = e / wc, c = e - r * wc;
            lds.win[r][sh + c] = org[(ptrdiff_t)r * stride + c];
        }
    }
    __syncthreads();
    const uint8_t (*s_win)[PITCH] = reinterpret_cast<const uint8_t (*)[PITCH]>(&lds.win[0][sh]);      // samples start at column sh
    if (xf && yf) {
        for (int e = tid; e < wr * bs; e += kPredThreads) {
            const int r = e / bs, x = e - r * bs;
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)s_win[r][x + t] * filt[xf][t];
            lds.h[r][x] = (int16_t)s;
        }
        __syncthreads();
    }
    for (int e = tid; e < bs * bs; e += kPredThreads) {
        const int y = e / bs, x = e - y * bs;
        int v;
        if (!xf && !yf) {
            v = s_win[y][x];
        } else if (!yf) {
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)s_win[y][x + t] * filt[xf][t];
            v = clip8((s + 32) >> 6);
        } else if (!xf) {
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)s_win[y + t][x] * filt[yf][t];
            v = clip8((s + 32) >> 6);
        } else {
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)lds.h[y + t][x] * filt[yf][t];
            v = clip8((s + 2048) >> 12);
        }
        dst[e] = (uint8_t)v;
    }
}

template <int MAXBS>
__global__ void __launch_bounds__(kPredThreads) inter_pred_kernel(const PredParams p)
{
    __shared__ __align__(16) PredLds<MAXBS> lds;
    const int plane = blockIdx.y;
    if (!p.pred[plane]) return;                         // uniform per workgroup
    const xmi_inter_pred_request rq = p.req[blockIdx.x];
    // this instance's sizes; anything that is no PU size is written by neither
    if (MAXBS == 16 ? (rq.size != 8 && rq.size != 16) : (rq.size != 32 && rq.size != 64)) return;
    if (plane == 0) {
        const uint8_t *ref = p.ref[0] + (ptrdiff_t)(rq.y0 + (rq.mv[1] >> 2)) * p.stride_y + rq.x0 + (rq.mv[0] >> 2);
        predict_block<8, MAXBS>(ref, p.stride_y, p.pred[0] + rq.out_y, rq.size, rq.mv[0] & 3, rq.mv[1] & 3, lds);
    } else {
        const uint8_t *ref = p.ref[plane] + (ptrdiff_t)((rq.y0 >> 1) + (rq.mv[1] >> 3)) * p.stride_c + (rq.x0 >> 1) + (rq.mv[0] >> 3);
        predict_block<4, MAXBS>(ref, p.stride_c, p.pred[plane] + rq.out_c, rq.size >> 1, rq.mv[0] & 7, rq.mv[1] & 7, lds);
    }
}

void launch_pred(const PredParams &p, int n, hipStream_t st)
{
    hipLaunchKernelGGL(inter_pred_kernel<16>, dim3(n, 3), dim3(kPredThreads), 0, st, p);
    hipLaunchKernelGGL(inter_pred_kernel<64>, dim3(n, 3), dim3(kPredThreads), 0, st, p);
}

std::mutex g_mu;   // the per-call forms share one staging area

// One block through the kernel: window gathered on the host, uploaded as a tiny plane of its own.
int percall(const char *who, bool chroma, const uint8_t *ref, uint8_t *pred, uint32_t ref_stride, uint32_t sx, uint32_t sy, int mvx, int mvy)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(ref && pred, "%s: null argument", who);
    XMI_ARG(sx == sy && (chroma ? (sx == 4 || sx == 8 || sx == 16 || sx == 32) : (sx == 8 || sx == 16 || sx == 32 || sx == 64)),
            "%s: block %ux%u is not a square PU of this plane", who, sx, sy);
    const int bs = (int)sx, taps = chroma ? 4 : 8, B = taps / 2 - 1, shift = chroma ? 3 : 2, mask = chroma ? 7 : 3;
    const int xf = mvx & mask, yf = mvy & mask;
    const int c0 = xf ? -B : 0, c1 = xf ? bs + taps / 2 : bs, r0 = yf ? -B : 0, r1 = yf ? bs + taps / 2 : bs;
    const int W = bs + taps;                                  // pitch of the staged plane; block origin at (B, B)
    const uint8_t *org = ref + (ptrdiff_t)(mvy >> shift) * (ptrdiff_t)ref_stride + (mvx >> shift);
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t win_bytes = (size_t)W * W, blk = (size_t)bs * bs;
    const size_t rq_off = (win_bytes + 15) & ~(size_t)15, out_off = rq_off + 64;
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(8, out_off + blk));
    if (!dev) return XMI_ERR_HIP;
    static std::vector<uint8_t> host;
    host.assign(out_off, 0);
    for (int r = r0; r < r1; r++) memcpy(&host[(size_t)(B + r) * W + B + c0], org + (ptrdiff_t)r * (ptrdiff_t)ref_stride + c0, (size_t)(c1 - c0));
    xmi_inter_pred_request rq;
    memset(&rq, 0, sizeof rq);
    rq.size = chroma ? 2 * bs : bs;                           // chroma blocks are size/2 at (x0/2, y0/2) = (0, 0)
    rq.mv[0] = (int16_t)xf;                                   // the integer part is already in `org`
    rq.mv[1] = (int16_t)yf;
    memcpy(&host[rq_off], &rq, sizeof rq);
    XMI_HIP(hipMemcpy(dev, host.data(), out_off, hipMemcpyHostToDevice));
    PredParams p;
    memset(&p, 0, sizeof p);
    const int pl = chroma ? 1 : 0;
    p.ref[pl] = dev + (size_t)B * W + B;
    p.pred[pl] = dev + out_off;
    p.req = reinterpret_cast<const xmi_inter_pred_request *>(dev + rq_off);
    p.stride_y = p.stride_c = W;
    p.n = 1;
    launch_pred(p, 1, nullptr);
    XMI_HIP(hipGetLastError());
    static std::vector<uint8_t> out;
    out.resize(blk);
    XMI_HIP(hipMemcpy(out.data(), dev + out_off, blk, hipMemcpyDeviceToHost));
    const int ps = chroma ? 32 : 64;                          // MAX_CU_SIZE/2, MAX_CU_SIZE
    for (int y = 0; y < bs; y++) memcpy(pred + (size_t)y * ps, &out[(size_t)y * bs], (size_t)bs);
    return XMI_OK;
}

}  // namespace

extern "C" {

int xmi_pred_inter_batch_dev(const uint8_t *d_ref_y, int stride_y, const uint8_t *d_ref_u, const uint8_t *d_ref_v, int stride_c,
                             const xmi_inter_pred_request *d_req, int n, uint8_t *d_pred_y, uint8_t *d_pred_u, uint8_t *d_pred_v,
                             void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_pred_inter_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_req, "xmi_pred_inter_batch_dev: null request table");
    XMI_ARG(!d_pred_y || (d_ref_y && stride_y > 0), "xmi_pred_inter_batch_dev: luma output without a luma reference");
    XMI_ARG((!d_pred_u || (d_ref_u && stride_c > 0)) && (!d_pred_v || (d_ref_v && stride_c > 0)),
            "xmi_pred_inter_batch_dev: chroma output without a chroma reference");
    XMI_ARG(d_pred_y || d_pred_u || d_pred_v, "xmi_pred_inter_batch_dev: no output plane");
    PredParams p;
    p.ref[0] = d_ref_y;
    p.ref[1] = d_ref_u;
    p.ref[2] = d_ref_v;
    p.pred[0] = d_pred_y;
    p.pred[1] = d_pred_u;
    p.pred[2] = d_pred_v;
    p.req = d_req;
    p.stride_y = stride_y;
    p.stride_c = stride_c;
    p.n = n;
    launch_pred(p, n, xmi::as_stream(stream));
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int xmi_pred_inter_luma(const uint8_t *ref, uint8_t *pred, uint32_t ref_stride, uint32_t size_x, uint32_t size_y, int mvx, int mvy)
{
    return percall("xmi_pred_inter_luma", false, ref, pred, ref_stride, size_x, size_y, mvx, mvy);
}

int xmi_pred_inter_chroma(const uint8_t *ref, uint8_t *pred, uint32_t ref_stride, uint32_t size_x, uint32_t size_y, int mvx, int mvy)
{
    return percall("xmi_pred_inter_chroma", true, ref, pred, ref_stride, size_x, size_y, mvx, mvy);
}

}  // extern "C"
