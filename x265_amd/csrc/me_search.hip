// Batched replay of the reference's xMotionSearch (src/encode.cpp:2815-3003) on gfx950.
//
// One wavefront per search request. The control flow is the reference's, statement for statement (MV_CLIP = 1,
// MVD_COST_BY_TABLE = 0, SATD_FOR_FRAC_ME = 0): pick the better of two MV predictors by the SAD of its
// interpolated block; test (0,0) unless the caller already did; test the predictor's integer position; walk a
// small diamond for at most ME_RANGE steps with first-improvement-wins order, stopping when nothing improves
// or the cost drops below 800; clip (x against the Y maximum, as the reference does); then two rounds of 9
// sub-pel points at half and quarter step. Cost = SAD + floor(sqrt(lambda)) * (bits(mvp.x - x) + bits(mvp.y - y)),
// bits() = xGetComponentBits (src/encode.cpp:194-205). Ties keep the earlier candidate (strict <).
// Every SAD is computed by the 64 lanes together (v_sad_u8 on dwords, unaligned reference dwords assembled
// with v_alignbyte_b32, wavefront reduction); sub-pel predictions are lookups in the interpolation planes
// (interp_planes.hip), which hold exactly what xPredInterLumaBlk would compute.
#include "xmi_common.h"

namespace {

typedef unsigned int u32;

struct MeParams {
    const uint8_t *cur;
    const uint8_t *planes;   // 16 planes, each rows*stride bytes; pixel (0,0) of the frame at (pad, pad)
    const xmi_me_request *req;
    xmi_me_result *res;
    int cur_stride, stride, rows, pad, n, me_range;
};

__device__ __forceinline__ u32 mv_bits(int v)
{
    const u32 t = v <= 0 ? ((u32)(-v) << 1) + 1u : (u32)v << 1;
    return 1u + 2u * (31u - (u32)__clz((int)t));
}

// SAD of an n x n block: cur is 4-byte aligned, ref has any byte alignment. All 64 lanes return the sum.
__device__ __forceinline__ u32 wave_sad(const uint8_t *cur, int cs, const uint8_t *ref, int rs, int n, int lane)
{
    const int dw_per_row = n >> 2, total = n * dw_per_row;
    u32 s = 0;
    for (int d = lane; d < total; d += 64) {
        const int row = d / dw_per_row, c4 = d - row * dw_per_row;
        const u32 c = *reinterpret_cast<const u32 *>(cur + (size_t)row * cs + c4 * 4);
        const uintptr_t a = reinterpret_cast<uintptr_t>(ref + (ptrdiff_t)row * rs + c4 * 4);
        const u32 *al = reinterpret_cast<const u32 *>(a & ~(uintptr_t)3);
        const u32 sh = (u32)(a & 3);
        const u32 lo = al[0];
        const u32 hi = sh ? al[1] : 0u;
        s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, sh), c, s);
    }
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) s += (u32)__shfl_xor((int)s, k);
    return s;
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void __launch_bounds__(256) me_search_kernel(const MeParams p)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= p.n) return;
    const xmi_me_request rq = p.req[idx];
    const int n = rq.size;
    const size_t plane_bytes = (size_t)p.rows * p.stride;
    const uint8_t *cur = p.cur + (size_t)rq.y0 * p.cur_stride + rq.x0;
    const uint8_t *ref0 = p.planes + (size_t)(p.pad + rq.y0) * p.stride + p.pad + rq.x0;   // integer plane at the PU origin
    const u32 lam = rq.sqrt_lambda;

    // SAD of the block predicted by quarter-pel MV (mx,my): what xPredInterLumaBlk + xSad_AMP would give
    auto sad_qpel = [&](int mx, int my) {
        const uint8_t *r = ref0 + ((my & 3) * 4 + (mx & 3)) * plane_bytes + (ptrdiff_t)(my >> 2) * p.stride + (mx >> 2);
        return wave_sad(cur, p.cur_stride, r, p.stride, n, lane);
    };
    auto sad_ipel = [&](int x, int y) { return wave_sad(cur, p.cur_stride, ref0 + (ptrdiff_t)y * p.stride + x, p.stride, n, lane); };

    int n_int = 0, n_sub = 0;
    int mvp_idx = -1;
    u32 bsad = 0x7fffffffu, bcost = 0x7fffffffu;
    const int passes = ((abs(rq.mvp[0][0] - rq.mvp[1][0]) >> 2) + (abs(rq.mvp[0][1] - rq.mvp[1][1]) >> 2) > 4) ? 2 : 1;
    for (int i = 0; i < passes; i++) {
        const u32 s = sad_qpel(rq.mvp[i][0], rq.mvp[i][1]);
        n_sub++;
        if (s < bsad) {
            bsad = s;
            mvp_idx = i;
        }
    }
    const int px = rq.mvp[mvp_idx][0], py = rq.mvp[mvp_idx][1];
    auto mvcost = [&](int x, int y) { return lam * (mv_bits(px - x) + mv_bits(py - y)); };
    int bmx = 0, bmy = 0, best_k = -1;
    if (!rq.zero_mv_tested) {
        const u32 c = sad_ipel(0, 0) + mvcost(0, 0);
        n_int++;
        if (c < bcost) {
            bcost = c;
            bmx = bmy = 0;
        }
    }
    int omx = px >> 2, omy = py >> 2;
    {
        const u32 c = sad_ipel(omx, omy) + mvcost(omx * 4, omy * 4);
        n_int++;
        if (c < bcost) {
            bcost = c;
            bmx = omx * 4;
            bmy = omy * 4;
        }
    }
    const int dia4x[4] = { 0, 0, -1, 1 }, dia4y[4] = { -1, 1, 0, 0 };
    for (int i = 0; i < p.me_range; i++) {
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tx = clip3(rq.mv_min_ipel[0], rq.mv_max_ipel[0], omx + dia4x[k]);
            const int ty = clip3(rq.mv_min_ipel[1], rq.mv_max_ipel[1], omy + dia4y[k]);
            const u32 c = sad_ipel(tx, ty) + mvcost(tx * 4, ty * 4);
            n_int++;
            if (c < bcost) {
                changed = true;
                bcost = c;
                best_k = k;
                bmx = tx * 4;
                bmy = ty * 4;
            }
        }
        if (!changed) break;
        omx += best_k == 2 ? -1 : best_k == 3 ? 1 : 0;
        omy += best_k == 0 ? -1 : best_k == 1 ? 1 : 0;
        if (bcost < 800) break;
    }
    {   // src/encode.cpp:2950-2953: x is clipped against the Y maximum (kept)
        const int mnx = 4 * (rq.mv_min_ipel[0] - 4), mny = 4 * (rq.mv_min_ipel[1] - 4), mxy = 4 * (rq.mv_max_ipel[1] - 4);
        bmx = clip3(mnx, mxy, bmx);
        bmy = clip3(mny, mxy, bmy);
    }
    bcost = 0x7fffffffu;
    const int dia9x[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 }, dia9y[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 };
    int step = 2;
    for (int i = 0; i < 2; i++) {
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int tx = (int16_t)(bmx + dia9x[k] * step), ty = (int16_t)(bmy + dia9y[k] * step);
            const u32 c = sad_qpel(tx, ty) + mvcost(tx, ty);
            n_sub++;
            if (c < bcost) {
                changed = true;
                bcost = c;
                best_k = k;
            }
        }
        if (changed) {
            bmx += dia9x[best_k] * step;
            bmy += dia9y[best_k] * step;
        }
        step >>= 1;
    }
    if (lane == 0) {
        xmi_me_result out;
        out.mv[0] = (int16_t)bmx;
        out.mv[1] = (int16_t)bmy;
        out.best_cost = bcost;
        out.best_sad = bsad;
        out.mvp_idx = mvp_idx;
        out.n_int_sads = n_int;
        out.n_sub_sads = n_sub;
        p.res[idx] = out;
    }
}

}  // namespace

extern "C" int xmi_me_search_batch_dev(const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows,
                                       int pad, const xmi_me_request *d_req, int n, int me_range, xmi_me_result *d_res,
                                       void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_me_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_cur && d_planes && d_req && d_res, "xmi_me_search_batch_dev: null argument");
    XMI_ARG(cur_stride % 4 == 0 && ((uintptr_t)d_cur & 3) == 0, "xmi_me_search_batch_dev: current frame must be 4-byte aligned");
    XMI_ARG(stride % 4 == 0 && ((uintptr_t)d_planes & 3) == 0 && pad % 4 == 0, "xmi_me_search_batch_dev: planes, stride and pad must be multiples of 4");
    XMI_ARG(me_range >= 0 && me_range <= 1024, "xmi_me_search_batch_dev: me_range %d out of range", me_range);
    MeParams p = { d_cur, d_planes, d_req, d_res, cur_stride, stride, rows, pad, n, me_range };
    hipLaunchKernelGGL(me_search_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}
