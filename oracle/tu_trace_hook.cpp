// tu_trace_hook.cpp - TEST INFRASTRUCTURE: records the transform-unit chains of a reference encode.
//
// xSubDctQuantRec (src/encode.cpp:3005-3021) is static, but the four functions it chains - xSubDct, xQuant, xDeQuant,
// xIDctAdd (src/pixel.cpp:3446, 1054, 1244, 5938) - are external and, in the position-independent build, called
// through the PLT. The definitions below forward to the originals (dlsym RTLD_NEXT) and reassemble each chain:
// source and prediction block, size, DST flag, QP and slice type as the encoder passed them; the levels and their sum
// as xQuant returned them; the reconstruction xIDctAdd wrote (chains whose levels are all zero have none: the caller
// copies the prediction). The intra path calls the same four functions one by one (src/encode.cpp:4083-4140), so its
// chains are recorded too. Records go to the file named by XMI_TU_TRACE:
//   int32 size, dst4, qp, slice, poc, stride, sum, has_rec; pix[size^2]; pred[size^2]; int16 level[size^2]; rec[size^2].
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

typedef void (*sub_fn)(Int16 *, UInt8 *, UInt, UInt8 *, UInt, Int16 *, Int16 *, Int, UInt);
typedef UInt32 (*q_fn)(Int16 *, Int16 *, UInt, UInt, Int, Int, xSliceType);
typedef void (*dq_fn)(Int16 *, Int16 *, UInt, UInt, Int, Int);
typedef void (*idct_fn)(UInt8 *, UInt, Int16 *, UInt, UInt8 *, Int16 *, Int16 *, Int, UInt);

namespace {
FILE *g_f = nullptr;
bool g_opened = false;
struct {
    int stage;             // 0 idle, 1 after xSubDct, 2 after xQuant with levels
    int size, dst4, qp, slice, stride;
    unsigned sum;
    UInt8 pix[32 * 32], pred[32 * 32], rec[32 * 32];
    Int16 level[32 * 32];
} g;

void open_once()
{
    if (g_opened) return;
    g_opened = true;
    if (getenv("XMI_TU_TRACE")) g_f = fopen(getenv("XMI_TU_TRACE"), "wb");
}

void emit(int has_rec)
{
    const int n2 = g.size * g.size;
    const int hdr[8] = { g.size, g.dst4, g.qp, g.slice, h.iPoc, g.stride, (int)g.sum, has_rec };
    fwrite(hdr, sizeof hdr, 1, g_f);
    fwrite(g.pix, 1, n2, g_f);
    fwrite(g.pred, 1, n2, g_f);
    fwrite(g.level, 2, n2, g_f);
    if (!has_rec) memcpy(g.rec, g.pred, n2);
    fwrite(g.rec, 1, n2, g_f);
    g.stage = 0;
}
}  // namespace

void xSubDct(Int16 *pDst, UInt8 *pSrc, UInt nStride, UInt8 *pRef, UInt nStrideRef, Int16 *piTmp0, Int16 *piTmp1, Int iWidth, UInt nMode)
{
    static sub_fn real = (sub_fn)dlsym(RTLD_NEXT, "_Z7xSubDctPsPhjS0_jS_S_ij");
    open_once();
    if (g_f) {
        if (g.stage == 2) emit(0);          // the previous chain was never reconstructed
        g.stage = 1;
        g.size = iWidth;
        g.dst4 = (iWidth == 4 && nMode != MODE_INVALID) ? 1 : 0;
        g.stride = (int)nStride;
        for (int y = 0; y < iWidth; y++) {
            memcpy(g.pix + y * iWidth, pSrc + y * nStride, iWidth);
            memcpy(g.pred + y * iWidth, pRef + y * nStrideRef, iWidth);
        }
    }
    real(pDst, pSrc, nStride, pRef, nStrideRef, piTmp0, piTmp1, iWidth, nMode);
}

UInt32 xQuant(Int16 *pDst, Int16 *pSrc, UInt nStride, UInt nQP, Int iWidth, Int iHeight, xSliceType eSType)
{
    static q_fn real = (q_fn)dlsym(RTLD_NEXT, "_Z6xQuantPsS_jjii10xSliceType");
    open_once();
    const UInt32 sum = real(pDst, pSrc, nStride, nQP, iWidth, iHeight, eSType);
    if (g_f && g.stage == 1 && iWidth == g.size) {
        g.qp = (int)nQP;
        g.slice = (int)eSType;
        g.sum = sum;
        for (int y = 0; y < iWidth; y++) memcpy(g.level + y * iWidth, pDst + y * nStride, iWidth * sizeof(Int16));
        if (sum) g.stage = 2;
        else emit(0);
    } else {
        g.stage = 0;
    }
    return sum;
}

void xDeQuant(Int16 *pDst, Int16 *pSrc, UInt nStride, UInt nQP, Int iWidth, Int iHeight)
{
    static dq_fn real = (dq_fn)dlsym(RTLD_NEXT, "_Z8xDeQuantPsS_jjii");
    real(pDst, pSrc, nStride, nQP, iWidth, iHeight);
}

void xIDctAdd(UInt8 *pDst, UInt nStrideDst, Int16 *pSrc, UInt nStride, UInt8 *pRef, Int16 *piTmp0, Int16 *piTmp1, Int iWidth, UInt nMode)
{
    static idct_fn real = (idct_fn)dlsym(RTLD_NEXT, "_Z8xIDctAddPhjPsjS_S0_S0_ij");
    open_once();
    real(pDst, nStrideDst, pSrc, nStride, pRef, piTmp0, piTmp1, iWidth, nMode);
    if (g_f && g.stage == 2 && iWidth == g.size) {
        for (int y = 0; y < iWidth; y++) memcpy(g.rec + y * iWidth, pDst + y * nStrideDst, iWidth);
        emit(1);
    }
}
