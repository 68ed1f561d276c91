// analysis_gpu.hip - the P-picture analysis loop on the GPU: host/analysis_core.cpp (decision recursion, rate estimate,
// wavefront of CTU rows) bound to an evaluator that serves every batch of CU requests with the library's kernels, and the
// C ABI around it (xmi_analysis_*).
//
// One evaluator call = one round trip: the request tables go up in one copy, a fixed sequence of launches runs on one
// stream, everything the host needs comes back in one copy.
//   luma, per CU    : analysis_prepare_kernel  SSD of every usable merge candidate against the source, read from the
//                                              quarter-pel planes (a prediction is a plain copy out of the plane of its
//                                              phase); first smallest wins; its prediction and the source block are packed
//                                              as transform units; the search request learns whether that candidate was
//                                              the zero vector
//                     xmi_me_search_batch_dev  xMotionSearch replay
//                     analysis_me_pred_kernel  prediction at the vector found, packed as transform units
//                     xmi_tu_chain_batch_dev   per transform size: both candidates of every CU in one batch
//                     xmi_block_metric_batch_dev (SSD) per transform size: source against reconstruction
//   chroma, per CTU : analysis_gather_kernel   source blocks of U and V packed as transform units
//                     xmi_pred_inter_batch_dev prediction of U, then of V, at the CU's vector
//                     xmi_tu_chain_batch_dev   per transform size, U and V units in one batch
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "host/analysis_core.h"
#include "xmi_common.h"

using namespace xmi::analysis;

namespace {

constexpr int kPad = 80, kPadC = 40;          // PAD_YSIZE / PAD_CSIZE of the reference (inc/config.h:52-53)
constexpr int kMeRange = 32;                  // ME_RANGE (inc/config.h:50)

struct LumaJob {
    int32_t x0, y0, size;
    int32_t n_merge;
    int16_t merge_mv[5][2];
    uint8_t merge_ok[5];
    uint8_t cls;                  // transform class: 0 = 8x8, 1 = 16x16, 2 = 32x32
    uint8_t units;                // transform units per candidate: 1, or 4 for a 64x64 CU
    uint8_t reserved;
    int32_t unit_merge, unit_me;  // first unit of the two candidates inside the class arena
};
struct LumaOut {
    int32_t best_merge;
    uint32_t ssd_skip;
};
struct GatherJob {
    int32_t plane;                // 0 = U, 1 = V source
    int32_t x, y, n;              // block in plane samples
    uint32_t dst;                 // byte offset in the packed source arena
};

struct ClassPtrs {
    uint8_t *pix[3], *pred[3];
};

struct PrepareParams {
    const LumaJob *jobs;
    LumaOut *out;
    xmi_me_request *me_req;
    const xmi_me_result *me_res;
    const uint8_t *src;           // source luma, stride = width
    const uint8_t *planes;        // 16 quarter-pel planes of the padded reference
    int width, stride, rows;      // plane geometry (stride x rows bytes per plane)
    ClassPtrs c;
};

__device__ __forceinline__ const uint8_t *plane_block(const PrepareParams &p, int x0, int y0, int mvx, int mvy)
{
    const size_t plane = (size_t)((mvy & 3) * 4 + (mvx & 3)) * (size_t)p.stride * p.rows;
    return p.planes + plane + (size_t)(kPad + y0 + (mvy >> 2)) * p.stride + kPad + x0 + (mvx >> 2);
}

// byte offset of sample (x, y) of a CU inside its packed transform units
__device__ __forceinline__ uint32_t unit_offset(int first_unit, int size, int x, int y)
{
    if (size == 64) return (uint32_t)(first_unit + (y >> 5) * 2 + (x >> 5)) * 1024u + (uint32_t)((y & 31) * 32 + (x & 31));
    return (uint32_t)first_unit * (uint32_t)(size * size) + (uint32_t)(y * size + x);
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ void __launch_bounds__(256) analysis_prepare_kernel(const PrepareParams p)
{
    __shared__ uint32_t red[4];
    const LumaJob j = p.jobs[blockIdx.x];
    const int n = j.size, npx = n * n, tid = threadIdx.x;
    const uint8_t *src = p.src + (size_t)j.y0 * p.width + j.x0;
    uint32_t best_ssd = 0xffffffffu;
    int best = -1;
    for (int k = 0; k < j.n_merge; k++) {
        if (!j.merge_ok[k]) continue;                       // uniform
        const uint8_t *pr = plane_block(p, j.x0, j.y0, j.merge_mv[k][0], j.merge_mv[k][1]);
        uint32_t acc = 0;
        for (int i = tid * 4; i < npx; i += 1024) {         // four samples of one row (sizes are multiples of 8)
            const int y = i / n, x = i - y * n;
            const uint8_t *a = src + (size_t)y * p.width + x, *b = pr + (size_t)y * p.stride + x;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int d = (int)a[t] - (int)b[t];
                acc += (uint32_t)(d * d);
            }
        }
        const uint32_t ssd = block_sum(acc, red);
        if (ssd < best_ssd) best_ssd = ssd, best = k;       // every thread holds the same totals
    }
    if (tid == 0) {
        p.out[blockIdx.x].best_merge = best;
        p.out[blockIdx.x].ssd_skip = best_ssd;
        p.me_req[blockIdx.x].zero_mv_tested = best >= 0 && j.merge_mv[best][0] == 0 && j.merge_mv[best][1] == 0;
    }
    const uint8_t *pr = best >= 0 ? plane_block(p, j.x0, j.y0, j.merge_mv[best][0], j.merge_mv[best][1]) : nullptr;
    uint8_t *pix = p.c.pix[j.cls], *pred = p.c.pred[j.cls];
    for (int i = tid * 4; i < npx; i += 1024) {
        const int y = i / n, x = i - y * n;
        const uchar4 s = *reinterpret_cast<const uchar4 *>(src + (size_t)y * p.width + x);
        uchar4 q = make_uchar4(0, 0, 0, 0);
        if (pr) {
            const uint8_t *b = pr + (size_t)y * p.stride + x;
            q = make_uchar4(b[0], b[1], b[2], b[3]);
        }
        *reinterpret_cast<uchar4 *>(pix + unit_offset(j.unit_merge, n, x, y)) = s;
        *reinterpret_cast<uchar4 *>(pix + unit_offset(j.unit_me, n, x, y)) = s;
        *reinterpret_cast<uchar4 *>(pred + unit_offset(j.unit_merge, n, x, y)) = q;
    }
}

__global__ void __launch_bounds__(256) analysis_me_pred_kernel(const PrepareParams p)
{
    const LumaJob j = p.jobs[blockIdx.x];
    const xmi_me_result r = p.me_res[blockIdx.x];
    const int n = j.size, npx = n * n;
    const int mvx = r.mvp_idx < 0 ? 0 : r.mv[0], mvy = r.mvp_idx < 0 ? 0 : r.mv[1];   // a refused request predicts from the zero vector
    const uint8_t *pr = plane_block(p, j.x0, j.y0, mvx, mvy);
    uint8_t *pred = p.c.pred[j.cls];
    for (int i = threadIdx.x * 4; i < npx; i += 1024) {
        const int y = i / n, x = i - y * n;
        const uint8_t *b = pr + (size_t)y * p.stride + x;
        *reinterpret_cast<uchar4 *>(pred + unit_offset(j.unit_me, n, x, y)) = make_uchar4(b[0], b[1], b[2], b[3]);
    }
}

struct GatherParams {
    const GatherJob *jobs;
    const uint8_t *plane[2];
    int stride;
    uint8_t *dst;
};

__global__ void __launch_bounds__(64) analysis_gather_kernel(const GatherParams p)
{
    const GatherJob j = p.jobs[blockIdx.x];
    const uint8_t *src = p.plane[j.plane] + (size_t)j.y * p.stride + j.x;
    uint8_t *dst = p.dst + j.dst;
    for (int i = threadIdx.x * 4; i < j.n * j.n; i += 256) {
        const int y = i / j.n, x = i - y * j.n;
        *reinterpret_cast<uchar4 *>(dst + i) = *reinterpret_cast<const uchar4 *>(src + (size_t)y * p.stride + x);
    }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// Offsets of one class of transform units inside the download area: sums, SSDs, levels, reconstruction, prediction.
struct ClassLayout {
    int n = 0, units = 0;
    size_t sum = 0, ssd = 0, level = 0, rec = 0, pred = 0;
    size_t place(size_t at, int n_, int units_, bool with_ssd)
    {
        n = n_, units = units_;
        const size_t px = (size_t)units * n * n;
        sum = at, at = align_up(at + 4 * (size_t)units);
        ssd = at, at = with_ssd ? align_up(at + 4 * (size_t)units) : at;
        level = at, at = align_up(at + 2 * px);
        rec = at, at = align_up(at + px);
        pred = at, at = align_up(at + px);
        return at;
    }
};

class GpuEvaluator : public Evaluator {
public:
    int w = 0, h = 0, stride = 0, rows = 0, stride_c = 0, rows_c = 0, max_cu = 0, max_chroma = 0;
    uint8_t *d_src[3] = {}, *d_ref[3] = {}, *d_planes = nullptr;
    uint8_t *d_up = nullptr, *d_down = nullptr, *d_pix = nullptr, *d_qp = nullptr;   // d_qp: [luma qp x cap][chroma qp x cap][flags = 0 x cap]
    uint8_t *h_up = nullptr, *h_down = nullptr;
    size_t up_bytes = 0, down_bytes = 0, pix_bytes = 0, qp_cap = 0;
    hipStream_t st = nullptr;
    std::vector<uint8_t> untile_b;
    std::vector<int16_t> untile_w;
    int cur_qp = -1, cur_qpc = -1;
    char err[256] = "";

    ~GpuEvaluator() override { release(); }

    void release()
    {
        for (int i = 0; i < 3; i++) {
            if (d_src[i]) (void)hipFree(d_src[i]);
            if (d_ref[i]) (void)hipFree(d_ref[i]);
            d_src[i] = d_ref[i] = nullptr;
        }
        for (uint8_t **q : { &d_planes, &d_up, &d_down, &d_pix, &d_qp })
            if (*q) (void)hipFree(*q), *q = nullptr;
        if (h_up) (void)hipHostFree(h_up), h_up = nullptr;
        if (h_down) (void)hipHostFree(h_down), h_down = nullptr;
        if (st) (void)hipStreamDestroy(st), st = nullptr;
    }

    int init(int width, int height)
    {
        w = width, h = height;
        stride = w + 2 * kPad, rows = h + 2 * kPad, stride_c = w / 2 + 2 * kPadC, rows_c = h / 2 + 2 * kPadC;
        max_cu = h / 64;                      // one CU request per CTU row and round
        max_chroma = (h / 64) * 64;           // the CUs of one CTU per row: at most 64 of 8x8
        const size_t plane = (size_t)stride * rows, plane_c = (size_t)stride_c * rows_c;
        // worst cases per round: every CU a 64x64 (8 units of 32x32 each with both candidates), every chroma CU a 4x4 pair
        const size_t luma_px = (size_t)max_cu * 2 * 4096, chroma_px = (size_t)max_chroma * 2 * 64 * 4;
        up_bytes = align_up(sizeof(LumaJob) * max_cu) + align_up(sizeof(xmi_me_request) * max_cu) + align_up(sizeof(GatherJob) * max_chroma * 2 * 4) +
                   2 * align_up(sizeof(xmi_inter_pred_request) * max_chroma * 4) + 4096;
        down_bytes = align_up(sizeof(LumaOut) * max_cu) + align_up(sizeof(xmi_me_result) * max_cu) + 3 * 5 * 256 + 4 * luma_px + 8 * (size_t)max_cu * 8 +
                     3 * 5 * 256 + 4 * chroma_px + 8 * (size_t)max_chroma * 8 + 65536;
        pix_bytes = align_up(3 * luma_px) + align_up(3 * chroma_px) + 4096;
        qp_cap = align_up(std::max<size_t>((size_t)max_cu * 8, (size_t)max_chroma * 2 * 4) + 64);
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[0]), (size_t)w * h));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[1]), (size_t)w * h / 4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[2]), (size_t)w * h / 4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[0]), plane));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[1]), plane_c));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[2]), plane_c));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_planes), 16 * plane));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_up), up_bytes));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_down), down_bytes));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_pix), pix_bytes));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_qp), 3 * qp_cap));
        XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_up), up_bytes, hipHostMallocDefault));
        XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_down), down_bytes, hipHostMallocDefault));
        XMI_HIP(hipStreamCreate(&st));
        XMI_HIP(hipMemsetAsync(d_planes, 0, 16 * plane, st));
        XMI_HIP(hipMemsetAsync(d_pix, 0, pix_bytes, st));
        XMI_HIP(hipMemsetAsync(d_down, 0, down_bytes, st));
        untile_b.resize((size_t)max_cu * 3 * 4096);
        untile_w.resize((size_t)max_cu * 2 * 4096);
        return XMI_OK;
    }

    // source planes (w x h, w/2 x h/2) and padded reference planes from host memory; builds the quarter-pel planes
    int bind_pictures(const uint8_t *const src[3], const uint8_t *const ref[3], int qp, int qpc)
    {
        XMI_HIP(hipMemcpyAsync(d_src[0], src[0], (size_t)w * h, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_src[1], src[1], (size_t)w * h / 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_src[2], src[2], (size_t)w * h / 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_ref[0], ref[0], (size_t)stride * rows, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_ref[1], ref[1], (size_t)stride_c * rows_c, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_ref[2], ref[2], (size_t)stride_c * rows_c, hipMemcpyHostToDevice, st));
        int rc = xmi_interp_planes_dev(d_ref[0], stride, rows, stride, d_planes, st);
        if (rc != XMI_OK) return rc;
        if (qp != cur_qp || qpc != cur_qpc) {
            XMI_HIP(hipMemsetAsync(d_qp, qp, qp_cap, st));
            XMI_HIP(hipMemsetAsync(d_qp + qp_cap, qpc, qp_cap, st));
            XMI_HIP(hipMemsetAsync(d_qp + 2 * qp_cap, 0, qp_cap, st));
            cur_qp = qp, cur_qpc = qpc;
        }
        XMI_HIP(hipStreamSynchronize(st));
        return XMI_OK;
    }

    int evaluate(CuEval *const *cu, int ncu, ChromaEval *const *ch, int nch) override
    {
        const int rc = run(cu, ncu, ch, nch);
        if (rc != XMI_OK) snprintf(err, sizeof err, "%s", xmi_last_error());
        return rc;
    }

private:
    int run(CuEval *const *cu, int ncu, ChromaEval *const *ch, int nch)
    {
        XMI_ARG(ncu <= max_cu && nch <= max_chroma, "analysis: %d CU / %d chroma requests exceed the round capacity %d / %d", ncu, nch, max_cu, max_chroma);
        // ---- upload area ----
        size_t up = 0;
        LumaJob *jobs = reinterpret_cast<LumaJob *>(h_up + up);
        const size_t off_jobs = up;
        up = align_up(up + sizeof(LumaJob) * ncu);
        xmi_me_request *me_req = reinterpret_cast<xmi_me_request *>(h_up + up);
        const size_t off_me_req = up;
        up = align_up(up + sizeof(xmi_me_request) * ncu);

        // luma: transform classes; both candidates of a CU get the same relative unit index in the two halves of a class
        int units_in[3] = { 0, 0, 0 };
        for (int i = 0; i < ncu; i++) {
            const CuEval &q = *cu[i];
            LumaJob &j = jobs[i];
            XMI_ARG(q.size == 8 || q.size == 16 || q.size == 32 || q.size == 64, "analysis: CU size %d", q.size);
            XMI_ARG(q.x0 >= 0 && q.y0 >= 0 && q.x0 + q.size <= w && q.y0 + q.size <= h && q.n_merge >= 0 && q.n_merge <= 5, "analysis: CU outside the picture");
            j.x0 = q.x0, j.y0 = q.y0, j.size = q.size, j.n_merge = q.n_merge;
            for (int k = 0; k < 5; k++) {
                j.merge_mv[k][0] = k < q.n_merge ? q.merge_mv[k].x : 0, j.merge_mv[k][1] = k < q.n_merge ? q.merge_mv[k].y : 0;
                // a candidate the clip of the reference would change is left out by the caller; one that would leave the
                // padded planes can only come from a caller's mistake and is left out here
                const bool inside = std::abs(j.merge_mv[k][0] >> 2) <= kPad - 8 && std::abs(j.merge_mv[k][1] >> 2) <= kPad - 8;
                j.merge_ok[k] = k < q.n_merge && q.merge_ok[k] && inside;
            }
            j.cls = q.size == 8 ? 0 : q.size == 16 ? 1 : 2;
            j.units = q.size == 64 ? 4 : 1;
            j.reserved = 0;
            j.unit_merge = units_in[j.cls];
            units_in[j.cls] += j.units;
            xmi_me_request &r = me_req[i];
            r.x0 = q.x0, r.y0 = q.y0, r.size = q.size;
            r.mvp[0][0] = q.amvp[0].x, r.mvp[0][1] = q.amvp[0].y, r.mvp[1][0] = q.amvp[1].x, r.mvp[1][1] = q.amvp[1].y;
            r.zero_mv_tested = 0;
            r.mv_min_ipel[0] = q.mv_min_ipel.x, r.mv_min_ipel[1] = q.mv_min_ipel.y;
            r.mv_max_ipel[0] = q.mv_max_ipel.x, r.mv_max_ipel[1] = q.mv_max_ipel.y;
            r.sqrt_lambda = (uint32_t)std::sqrt(q.lambda);
        }
        for (int i = 0; i < ncu; i++) jobs[i].unit_me = units_in[jobs[i].cls] + jobs[i].unit_merge;

        // chroma: one unit list per transform class (4, 8, 16), U units first, then V units at the same relative index
        int cunits_in[3] = { 0, 0, 0 };
        struct ChromaPlace {
            int cls, first, units;
        };
        std::vector<ChromaPlace> cplace(nch);
        for (int i = 0; i < nch; i++) {
            const ChromaEval &c = *ch[i];
            XMI_ARG(c.size == 8 || c.size == 16 || c.size == 32 || c.size == 64, "analysis: chroma CU size %d", c.size);
            XMI_ARG(c.x0 >= 0 && c.y0 >= 0 && c.x0 + c.size <= w && c.y0 + c.size <= h, "analysis: chroma CU outside the picture");
            XMI_ARG(std::abs(c.mv.x >> 2) <= kPad - 8 && std::abs(c.mv.y >> 2) <= kPad - 8, "analysis: chroma vector (%d,%d) leaves the padded reference", c.mv.x, c.mv.y);
            const int cls = c.size == 8 ? 0 : c.size == 16 ? 1 : 2, units = c.size == 64 ? 4 : 1;
            cplace[i] = { cls, cunits_in[cls], units };
            cunits_in[cls] += units;
        }
        const int cn[3] = { 4, 8, 16 }, ln[3] = { 8, 16, 32 };
        int ctotal = 0;
        for (int c = 0; c < 3; c++) ctotal += cunits_in[c];
        GatherJob *gjobs = reinterpret_cast<GatherJob *>(h_up + up);
        const size_t off_gjobs = up;
        up = align_up(up + sizeof(GatherJob) * 2 * ctotal);
        xmi_inter_pred_request *preq[2];
        size_t off_preq[2];
        for (int pl = 0; pl < 2; pl++) {
            preq[pl] = reinterpret_cast<xmi_inter_pred_request *>(h_up + up);
            off_preq[pl] = up;
            up = align_up(up + sizeof(xmi_inter_pred_request) * ctotal);
        }
        XMI_ARG(up <= up_bytes, "analysis: upload area too small");

        // ---- device-only packed sources and the download area ----
        size_t pix_luma[3], pix_chroma[3], px = 0;
        for (int c = 0; c < 3; c++) pix_luma[c] = px, px = align_up(px + (size_t)2 * units_in[c] * ln[c] * ln[c]);
        for (int c = 0; c < 3; c++) pix_chroma[c] = px, px = align_up(px + (size_t)2 * cunits_in[c] * cn[c] * cn[c]);
        XMI_ARG(px <= pix_bytes, "analysis: packed-source area too small");
        size_t down = 0;
        const size_t off_out = down;
        down = align_up(down + sizeof(LumaOut) * ncu);
        const size_t off_me_res = down;
        down = align_up(down + sizeof(xmi_me_result) * ncu);
        ClassLayout L[3], Cc[3];
        for (int c = 0; c < 3; c++) down = L[c].place(down, ln[c], 2 * units_in[c], true);
        for (int c = 0; c < 3; c++) down = Cc[c].place(down, cn[c], 2 * cunits_in[c], false);
        XMI_ARG(down <= down_bytes, "analysis: download area too small");

        // chroma job tables
        {
            int g = 0, r = 0;
            for (int i = 0; i < nch; i++) {
                const ChromaEval &c = *ch[i];
                const ChromaPlace &pc = cplace[i];
                const int n = cn[pc.cls];
                for (int u = 0; u < pc.units; u++) {
                    const int ox = pc.units == 4 ? (u & 1) * 32 : 0, oy = pc.units == 4 ? (u >> 1) * 32 : 0;   // luma offsets of the unit
                    for (int pl = 0; pl < 2; pl++) {
                        const uint32_t unit = (uint32_t)(pl * cunits_in[pc.cls] + pc.first + u);
                        gjobs[g++] = { pl, (c.x0 + ox) / 2, (c.y0 + oy) / 2, n, (uint32_t)(pix_chroma[pc.cls] + (size_t)unit * n * n) };
                        xmi_inter_pred_request &q = preq[pl][r];
                        q.x0 = c.x0 + ox, q.y0 = c.y0 + oy, q.size = pc.units == 4 ? 32 : c.size;
                        q.mv[0] = c.mv.x, q.mv[1] = c.mv.y;
                        q.out_y = 0;
                        q.out_c = (uint32_t)(Cc[pc.cls].pred + (size_t)unit * n * n);
                    }
                    r++;
                }
            }
        }

        XMI_HIP(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, st));
        int rc;
        if (ncu) {
            PrepareParams pp;
            pp.jobs = reinterpret_cast<const LumaJob *>(d_up + off_jobs);
            pp.out = reinterpret_cast<LumaOut *>(d_down + off_out);
            pp.me_req = reinterpret_cast<xmi_me_request *>(d_up + off_me_req);
            pp.me_res = reinterpret_cast<const xmi_me_result *>(d_down + off_me_res);
            pp.src = d_src[0], pp.planes = d_planes, pp.width = w, pp.stride = stride, pp.rows = rows;
            for (int c = 0; c < 3; c++) pp.c.pix[c] = d_pix + pix_luma[c], pp.c.pred[c] = d_down + L[c].pred;
            hipLaunchKernelGGL(analysis_prepare_kernel, dim3(ncu), dim3(256), 0, st, pp);
            XMI_HIP(hipGetLastError());
            rc = xmi_me_search_batch_dev(d_src[0], w, d_planes, stride, rows, kPad, pp.me_req, ncu, kMeRange,
                                         reinterpret_cast<xmi_me_result *>(d_down + off_me_res), st);
            if (rc != XMI_OK) return rc;
            hipLaunchKernelGGL(analysis_me_pred_kernel, dim3(ncu), dim3(256), 0, st, pp);
            XMI_HIP(hipGetLastError());
            for (int c = 0; c < 3; c++) {
                if (!L[c].units) continue;
                rc = xmi_tu_chain_batch_dev(ln[c], L[c].units, d_pix + pix_luma[c], d_down + L[c].pred, d_qp, d_qp + 2 * qp_cap,
                                            reinterpret_cast<int16_t *>(d_down + L[c].level), d_down + L[c].rec,
                                            reinterpret_cast<uint32_t *>(d_down + L[c].sum), st);
                if (rc != XMI_OK) return rc;
                rc = xmi_block_metric_batch_dev(XMI_METRIC_SSD, ln[c], L[c].units, d_pix + pix_luma[c], d_down + L[c].rec,
                                                reinterpret_cast<uint32_t *>(d_down + L[c].ssd), st);
                if (rc != XMI_OK) return rc;
            }
        }
        if (nch) {
            GatherParams gp = { reinterpret_cast<const GatherJob *>(d_up + off_gjobs), { d_src[1], d_src[2] }, w / 2, d_pix };
            hipLaunchKernelGGL(analysis_gather_kernel, dim3(2 * ctotal), dim3(64), 0, st, gp);
            XMI_HIP(hipGetLastError());
            for (int pl = 0; pl < 2; pl++) {    // the chroma plane goes through the request's "U" slot both times
                rc = xmi_pred_inter_batch_dev(nullptr, stride, d_ref[1 + pl] + (size_t)kPadC * stride_c + kPadC, nullptr, stride_c,
                                              reinterpret_cast<const xmi_inter_pred_request *>(d_up + off_preq[pl]), ctotal, nullptr, d_down, nullptr, st);
                if (rc != XMI_OK) return rc;
            }
            for (int c = 0; c < 3; c++) {
                if (!Cc[c].units) continue;
                rc = xmi_tu_chain_batch_dev(cn[c], Cc[c].units, d_pix + pix_chroma[c], d_down + Cc[c].pred, d_qp + qp_cap, d_qp + 2 * qp_cap,
                                            reinterpret_cast<int16_t *>(d_down + Cc[c].level), d_down + Cc[c].rec,
                                            reinterpret_cast<uint32_t *>(d_down + Cc[c].sum), st);
                if (rc != XMI_OK) return rc;
            }
        }
        XMI_HIP(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, st));
        XMI_HIP(hipStreamSynchronize(st));

        // ---- hand the results out ----
        const LumaOut *lo = reinterpret_cast<const LumaOut *>(h_down + off_out);
        const xmi_me_result *mr = reinterpret_cast<const xmi_me_result *>(h_down + off_me_res);
        uint8_t *ub = untile_b.data();
        int16_t *uw = untile_w.data();
        for (int i = 0; i < ncu; i++) {
            CuEval &q = *cu[i];
            const LumaJob &j = jobs[i];
            const ClassLayout &l = L[j.cls];
            const int n = ln[j.cls];
            const uint32_t *sum = reinterpret_cast<const uint32_t *>(h_down + l.sum), *ssd = reinterpret_cast<const uint32_t *>(h_down + l.ssd);
            const int16_t *lev = reinterpret_cast<const int16_t *>(h_down + l.level);
            const uint8_t *rec = h_down + l.rec, *pred = h_down + l.pred;
            XMI_ARG(mr[i].mvp_idx >= 0, "analysis: the motion search refused the request of the CU at (%d,%d) size %d", q.x0, q.y0, q.size);
            q.best_merge = lo[i].best_merge, q.ssd_skip = lo[i].ssd_skip;
            q.mv_me = { mr[i].mv[0], mr[i].mv[1] };
            q.ssd_merge = q.ssd_me = 0;
            for (int u = 0; u < 4; u++) q.sum_merge[u] = q.sum_me[u] = 0;
            for (int u = 0; u < j.units; u++) {
                q.sum_merge[u] = sum[j.unit_merge + u], q.sum_me[u] = sum[j.unit_me + u];
                q.ssd_merge += ssd[j.unit_merge + u], q.ssd_me += ssd[j.unit_me + u];
            }
            const size_t um = (size_t)j.unit_merge * n * n, ue = (size_t)j.unit_me * n * n;
            if (j.units == 1) {
                q.pred_merge = pred + um, q.rec_merge = rec + um, q.level_merge = lev + um;
                q.rec_me = rec + ue, q.level_me = lev + ue;
            } else {      // four 32x32 units back to rows of 64
                auto rows64 = [](auto *dst, const auto *src) {
                    for (int u = 0; u < 4; u++)
                        for (int y = 0; y < 32; y++) std::memcpy(dst + ((u >> 1) * 32 + y) * 64 + (u & 1) * 32, src + u * 1024 + y * 32, 32 * sizeof(*dst));
                };
                rows64(ub, pred + um), rows64(ub + 4096, rec + um), rows64(ub + 8192, rec + ue);
                rows64(uw, lev + um), rows64(uw + 4096, lev + ue);
                q.pred_merge = ub, q.rec_merge = ub + 4096, q.rec_me = ub + 8192, q.level_merge = uw, q.level_me = uw + 4096;
                ub += 3 * 4096, uw += 2 * 4096;
            }
        }
        size_t need_b = 0, need_w = 0;
        for (int i = 0; i < nch; i++)
            if (cplace[i].units == 4) need_b += 2 * 1024, need_w += 2 * 1024;
        chroma_b.resize(need_b), chroma_w.resize(need_w);
        uint8_t *cb = chroma_b.data();
        int16_t *cw = chroma_w.data();
        for (int i = 0; i < nch; i++) {
            ChromaEval &c = *ch[i];
            const ChromaPlace &pc = cplace[i];
            const ClassLayout &l = Cc[pc.cls];
            const int n = cn[pc.cls], half = cunits_in[pc.cls];
            const uint32_t *sum = reinterpret_cast<const uint32_t *>(h_down + l.sum);
            const int16_t *lev = reinterpret_cast<const int16_t *>(h_down + l.level);
            const uint8_t *rec = h_down + (c.skip ? l.pred : l.rec);      // a skipped CU keeps its prediction
            for (int u = 0; u < 4; u++) c.sum_u[u] = c.sum_v[u] = 0;
            for (int u = 0; u < pc.units; u++) c.sum_u[u] = c.skip ? 0 : sum[pc.first + u], c.sum_v[u] = c.skip ? 0 : sum[half + pc.first + u];
            const size_t uu = (size_t)pc.first * n * n, uv = (size_t)(half + pc.first) * n * n;
            if (pc.units == 1) {
                c.rec_u = rec + uu, c.rec_v = rec + uv, c.level_u = lev + uu, c.level_v = lev + uv;
            } else {      // four 16x16 units back to rows of 32
                auto rows32 = [](auto *dst, const auto *src) {
                    for (int u = 0; u < 4; u++)
                        for (int y = 0; y < 16; y++) std::memcpy(dst + ((u >> 1) * 16 + y) * 32 + (u & 1) * 16, src + u * 256 + y * 16, 16 * sizeof(*dst));
                };
                rows32(cb, rec + uu), rows32(cb + 1024, rec + uv), rows32(cw, lev + uu), rows32(cw + 1024, lev + uv);
                c.rec_u = cb, c.rec_v = cb + 1024, c.level_u = cw, c.level_v = cw + 1024;
                cb += 2048, cw += 2048;
            }
        }
        return XMI_OK;
    }

    std::vector<uint8_t> chroma_b;
    std::vector<int16_t> chroma_w;
};

}  // namespace

struct xmi_analysis {
    Analyser an;
    GpuEvaluator ev;
    PictureResult res;
    xmi_analysis(int w, int h) : an(w, h) {}
};

extern "C" {

xmi_analysis *xmi_analysis_create(int width, int height)
{
    if (xmi::require_device() != XMI_OK) return nullptr;
    if (width <= 0 || height <= 0 || width % 64 || height % 64 || width > 8192 || height > 4352) {
        xmi::set_error("xmi_analysis_create: %dx%d: the coded size must be a multiple of 64, at most 8192x4352", width, height);
        return nullptr;
    }
    xmi_analysis *a = new (std::nothrow) xmi_analysis(width, height);
    if (!a) {
        xmi::set_error("xmi_analysis_create: out of host memory");
        return nullptr;
    }
    if (a->ev.init(width, height) != XMI_OK) {
        delete a;
        return nullptr;
    }
    return a;
}

void xmi_analysis_destroy(xmi_analysis *a) { delete a; }

int xmi_analysis_state(xmi_analysis *a, int set, double *sum_cost, int32_t *count, double *avg_cost, uint64_t *frac_bits)
{
    XMI_ARG(a && sum_cost && count && avg_cost && frac_bits, "xmi_analysis_state: null argument");
    Analyser &an = a->an;
    const size_t n = an.sum_cost.size(), r = an.frac_bits.size();
    if (set) {
        std::memcpy(an.sum_cost.data(), sum_cost, n * 8), std::memcpy(an.count.data(), count, n * 4), std::memcpy(an.avg_cost.data(), avg_cost, n * 8);
        std::memcpy(an.frac_bits.data(), frac_bits, r * 8);
    } else {
        std::memcpy(sum_cost, an.sum_cost.data(), n * 8), std::memcpy(count, an.count.data(), n * 4), std::memcpy(avg_cost, an.avg_cost.data(), n * 8);
        std::memcpy(frac_bits, an.frac_bits.data(), r * 8);
    }
    return XMI_OK;
}

int xmi_analysis_p_picture(xmi_analysis *a, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const uint8_t *ref_y, const uint8_t *ref_u,
                           const uint8_t *ref_v, int pad, int qp, uint16_t *info0, uint8_t *info1, int16_t *mv, int16_t *mvd, uint8_t *mvp_idx,
                           int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint32_t *rd_cost,
                           uint8_t *ctx_in, uint8_t *ctx_out, uint64_t *counters)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(a && src_y && src_u && src_v && ref_y && ref_u && ref_v, "xmi_analysis_p_picture: null picture");
    XMI_ARG(pad == kPad, "xmi_analysis_p_picture: the reference must be padded by %d luma samples (got %d)", kPad, pad);
    XMI_ARG(qp >= 0 && qp <= 51, "xmi_analysis_p_picture: qp %d", qp);
    const uint8_t *src[3] = { src_y, src_u, src_v }, *ref[3] = { ref_y, ref_u, ref_v };
    rc = a->ev.bind_pictures(src, ref, qp, chroma_qp(qp));
    if (rc != XMI_OK) return rc;
    rc = a->an.p_picture(qp, a->ev, a->res);
    if (rc != 0) {
        if (rc == -2) xmi::set_error("xmi_analysis_p_picture: the wavefront stalled");
        else xmi::set_error("xmi_analysis_p_picture: %s", a->ev.err);
        return rc < 0 ? rc : XMI_ERR_HIP;
    }
    const PictureResult &r = a->res;
    auto put = [](void *dst, const auto &v) {
        if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
    };
    put(info0, r.info0), put(info1, r.info1), put(mv, r.mv), put(mvd, r.mvd), put(mvp_idx, r.mvp_idx);
    put(coef_y, r.coef_y), put(coef_u, r.coef_u), put(coef_v, r.coef_v), put(rec_y, r.rec_y), put(rec_u, r.rec_u), put(rec_v, r.rec_v);
    put(rd_cost, r.rd_cost), put(ctx_in, r.ctx_in), put(ctx_out, r.ctx_out);
    if (counters) counters[0] = r.evaluator_calls, counters[1] = r.cu_requests, counters[2] = r.chroma_requests;
    return XMI_OK;
}

}  // extern "C"
