// me_trace_hook.cpp - TEST INFRASTRUCTURE: records every xMotionSearch call of a reference encode.
//
// xMotionSearch (src/encode.cpp:2815) is an ordinary external function, and in a position-independent build the
// call from xEncDecideCU_Inter (src/encode.cpp:3379) goes through the PLT. oracle/Makefile (target trace) links
// the reference's encoder sources, compiled where they lie, into a shared library and its CLI (src/x265.cpp)
// into an executable together with this file: the definition below then takes the place of the library's in
// symbol resolution, forwards to the original (dlsym RTLD_NEXT) and appends inputs and outputs to the file
// named by XMI_ME_TRACE. Nothing of the reference is edited or copied; only its header is included for the types.
//
// Trace format (little endian): per call one record of 16 int32
//   poc, x, y, size, mvp0x, mvp0y, mvp1x, mvp1y, zero_mv, minx, miny, maxx, maxy, out_mvx, out_mvy, out_mvp_idx
// followed by lambda (double), out_cost (uint32), out_sad (uint32). The first call of every picture is preceded
// by a frame record: magic -1, poc, width, height, pad, then the source luma (width x height) and the padded
// reference luma ((width + 2 pad) x (height + 2 pad)).
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

typedef xMV (*me_fn)(CUInt8 *, CUInt8 *, Int16 *, CUInt32, CUInt, CUInt, UInt32 *, xMV *, Int *, int, xMV, xMV, double, UInt *);

xMV xMotionSearch(CUInt8 *pucCurY, CUInt8 *pucRefY, Int16 *psTmp, CUInt32 nRefStrideY, CUInt nSizeX, CUInt nSizeY,
                  UInt32 *uiBestRDCostY, xMV *sMVP, Int *pMVPIdx, int bZeroMV, xMV mv_min_ipel, xMV mv_max_ipel,
                  double dLambda_ssd, UInt *pSad_best)
{
    static me_fn real = (me_fn)dlsym(RTLD_NEXT, "_Z13xMotionSearchPKhS0_PsjjjPjP3xMVPiiS3_S3_dS2_");
    static FILE *f = getenv("XMI_ME_TRACE") ? fopen(getenv("XMI_ME_TRACE"), "wb") : nullptr;
    static int last_poc = -1;
    if (!real) {
        fprintf(stderr, "me_trace_hook: original xMotionSearch not found\n");
        abort();
    }
    const xMV mvp0 = sMVP[0], mvp1 = sMVP[1];
    const xMV mv = real(pucCurY, pucRefY, psTmp, nRefStrideY, nSizeX, nSizeY, uiBestRDCostY, sMVP, pMVPIdx, bZeroMV,
                        mv_min_ipel, mv_max_ipel, dLambda_ssd, pSad_best);
    if (!f) return mv;
    const xFrame *ref = h.pFrameRef, *cur = h.pFrameCur;
    const long off = (long)(pucRefY - ref->pucY);
    const int stride = ref->iStrideY, x = (int)(off % stride), y = (int)(off / stride);
    const int w = h.iWidth, hh = h.iHeight, pad = PAD_YSIZE;
    if (x < 0 || y < 0 || x + (int)nSizeX > w || y + (int)nSizeY > hh || (int)nRefStrideY != stride) {
        fprintf(stderr, "me_trace_hook: reference pointer outside the reference picture\n");
        abort();
    }
    for (unsigned r = 0; r < nSizeY; r++)     // the CTU-local source copy must be the source picture at (x, y)
        if (memcmp(pucCurY + r * MAX_CU_SIZE, cur->pucY + (long)(y + r) * cur->iStrideY + x, nSizeX)) {
            fprintf(stderr, "me_trace_hook: source block differs from the source picture\n");
            abort();
        }
    if (h.iPoc != last_poc) {
        last_poc = h.iPoc;
        const int hdr[5] = { -1, h.iPoc, w, hh, pad };
        fwrite(hdr, sizeof hdr, 1, f);
        for (int r = 0; r < hh; r++) fwrite(cur->pucY + (long)r * cur->iStrideY, 1, w, f);
        for (int r = -pad; r < hh + pad; r++) fwrite(ref->pucY + (long)r * stride - pad, 1, w + 2 * pad, f);
    }
    const int rec[16] = { h.iPoc, x, y, (int)nSizeX, mvp0.x, mvp0.y, mvp1.x, mvp1.y, bZeroMV, mv_min_ipel.x, mv_min_ipel.y,
                          mv_max_ipel.x, mv_max_ipel.y, mv.x, mv.y, *pMVPIdx };
    fwrite(rec, sizeof rec, 1, f);
    fwrite(&dLambda_ssd, sizeof(double), 1, f);
    const unsigned tail[2] = { *uiBestRDCostY, *pSad_best };
    fwrite(tail, sizeof tail, 1, f);
    fflush(f);
    return mv;
}
