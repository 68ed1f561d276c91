// planes_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference encoder with its luma interpolation served from
// GPU-built quarter-pel planes.
//
// xPredInterLumaBlk (src/pixel.cpp:8745) is called from src/encode.cpp through the PLT in the position-independent
// build of the reference's sources, so the definition below replaces it (same mechanism as me_trace_hook.cpp).
// Once per picture the padded previous reconstruction goes to the device, xmi_frame_begin builds the 16 planes
// and xmi_frame_planes_host brings them back; every prediction block is then a strided copy out of the plane of
// its fractional phase. The motion search itself and everything else stay the reference's CPU code.
// The encoder's bitstream must not change.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"
#include "../include/x265_amd.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

static void die(const char *what)
{
    fprintf(stderr, "planes_gpu_hook: %s: %s\n", what, xmi_last_error());
    abort();
}

void xPredInterLumaBlk(CUInt8 *pucRef, UInt8 *pucPred, CUInt32 nRefStride, CUInt nSizeX, CUInt nSizeY, xMV bmv, Int16 *psTmp)
{
    static xmi_frame *fs = nullptr;
    static UInt8 *planes = nullptr;
    static int last_poc = -1;
    (void)psTmp;
    const xFrame *ref = h.pFrameRef, *cur = h.pFrameCur;
    const int pad = PAD_YSIZE, stride = ref->iStrideY, rows = h.iHeight + 2 * pad;
    const size_t plane_bytes = (size_t)rows * stride;
    if (!fs) {
        if (xmi_init(0) != XMI_OK) die("xmi_init");
        if (xmi_frame_create(h.iWidth, h.iHeight, pad, &fs) != XMI_OK) die("xmi_frame_create");
        planes = (UInt8 *)malloc(16 * plane_bytes);
        if (!planes || stride != h.iWidth + 2 * pad) die("plane buffer");
    }
    if (h.iPoc != last_poc) {
        last_poc = h.iPoc;
        if (xmi_frame_begin(fs, cur->pucY, cur->iStrideY, ref->pucY - (long)pad * stride - pad, stride) != XMI_OK) die("xmi_frame_begin");
        if (xmi_frame_planes_host(fs, planes) != XMI_OK) die("xmi_frame_planes_host");
    }
    const long off = (long)(pucRef - ref->pucY);
    if ((int)nRefStride != stride || off < -(long)pad * stride || off >= (long)(h.iHeight + pad) * stride) die("reference pointer outside the reference picture");
    const UInt8 *src = planes + (size_t)((bmv.y & 3) * 4 + (bmv.x & 3)) * plane_bytes + (long)pad * stride + pad + off +
                       (long)(bmv.y >> 2) * stride + (bmv.x >> 2);
    for (UInt y = 0; y < nSizeY; y++) memcpy(pucPred + y * MAX_CU_SIZE, src + (long)y * stride, nSizeX);
}
