// me_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference encoder with its motion search served by libx265_amd.
//
// Same interposition as me_trace_hook.cpp (xMotionSearch, src/encode.cpp:2815, is reached through the PLT in the
// position-independent build of the reference's sources), but instead of forwarding to the original the
// definition below asks the GPU library: once per picture the source luma and the padded previous reconstruction
// go to the device (xmi_frame_begin builds the quarter-pel planes), then every call becomes one
// xmi_frame_me_search request. The encoder's decisions, and therefore its bitstream, must not change.
// One request per call is the slowest possible use of the batch interface; the point is parity in situ.
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "x265.h"
#include "../include/x265_amd.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

static void die(const char *what)
{
    fprintf(stderr, "me_gpu_hook: %s: %s\n", what, xmi_last_error());
    abort();
}

xMV xMotionSearch(CUInt8 *pucCurY, CUInt8 *pucRefY, Int16 *psTmp, CUInt32 nRefStrideY, CUInt nSizeX, CUInt nSizeY,
                  UInt32 *uiBestRDCostY, xMV *sMVP, Int *pMVPIdx, int bZeroMV, xMV mv_min_ipel, xMV mv_max_ipel,
                  double dLambda_ssd, UInt *pSad_best)
{
    static xmi_frame *fs = nullptr;
    static int last_poc = -1;
    (void)pucCurY;
    (void)psTmp;
    const xFrame *ref = h.pFrameRef, *cur = h.pFrameCur;
    const int pad = PAD_YSIZE, stride = ref->iStrideY;
    if (!fs) {
        if (xmi_init(0) != XMI_OK) die("xmi_init");
        if (xmi_frame_create(h.iWidth, h.iHeight, pad, &fs) != XMI_OK) die("xmi_frame_create");
    }
    if (h.iPoc != last_poc) {
        last_poc = h.iPoc;
        if (xmi_frame_begin(fs, cur->pucY, cur->iStrideY, ref->pucY - (long)pad * stride - pad, stride) != XMI_OK) die("xmi_frame_begin");
    }
    const long off = (long)(pucRefY - ref->pucY);
    xmi_me_request rq;
    rq.x0 = (int)(off % stride);
    rq.y0 = (int)(off / stride);
    rq.size = (int)nSizeX;
    rq.mvp[0][0] = sMVP[0].x;
    rq.mvp[0][1] = sMVP[0].y;
    rq.mvp[1][0] = sMVP[1].x;
    rq.mvp[1][1] = sMVP[1].y;
    rq.zero_mv_tested = bZeroMV;
    rq.mv_min_ipel[0] = mv_min_ipel.x;
    rq.mv_min_ipel[1] = mv_min_ipel.y;
    rq.mv_max_ipel[0] = mv_max_ipel.x;
    rq.mv_max_ipel[1] = mv_max_ipel.y;
    rq.sqrt_lambda = (uint32_t)sqrt(dLambda_ssd);
    xmi_me_result rs;
    if (nSizeX != nSizeY || (int)nRefStrideY != stride || xmi_frame_me_search(fs, &rq, 1, ME_RANGE, &rs) != XMI_OK) die("xmi_frame_me_search");
    *uiBestRDCostY = rs.best_cost;
    *pSad_best = rs.best_sad;
    *pMVPIdx = rs.mvp_idx;
    xMV mv;
    mv.x = rs.mv[0];
    mv.y = rs.mv[1];
    return mv;
}
