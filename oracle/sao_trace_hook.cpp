// sao_trace_hook.cpp - TEST INFRASTRUCTURE: records the SAO statistics of a reference encode.
//
// calcSaoStatsCu (src/encode.cpp:364) is an external function called per CTU and component from rdoSaoUnitAll
// (src/encode.cpp:1428) through the PLT of the position-independent build, after the picture has been deblocked and
// before any offset is applied. The definition below forwards to the original and copies out the two tables it
// filled (h->m_count[partIdx], h->m_offsetOrg[partIdx]); at the first call of a picture it also copies the deblocked
// reconstruction and the source picture. Records go to the file named by XMI_SAO_TRACE:
//   picture record: int32 tag = 1, width, height, poc; rec Y, U, V; source Y, U, V (width x height, then halves)
//   CTU record:     int32 tag = 2, ctu x, ctu y, component; int64 count[5][33]; int64 offsetOrg[5][33]
// processSaoUnitAll (src/encode.cpp:1762), which applies the decided offsets to one component plane in place, is
// recorded the same way:
//   plane record:   int32 tag = 3, component, CTUs, poc; per CTU int32 typeIdx, subTypeIdx, offset[4], mergeLeftFlag,
//                   mergeUpFlag; the plane after the call
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

typedef void (*stats_fn)(X265_t *, UInt32, UInt32, int, int);
typedef void (*apply_fn)(X265_t *, SaoLcuParam *, int);

namespace {
FILE *g_f = nullptr;
bool g_opened = false;
int g_poc = -1;

void planes(const xFrame *f, int w, int ht)
{
    for (int y = 0; y < ht; y++) fwrite(f->pucY + (size_t)y * f->iStrideY, 1, w, g_f);
    for (int y = 0; y < ht / 2; y++) fwrite(f->pucU + (size_t)y * f->iStrideC, 1, w / 2, g_f);
    for (int y = 0; y < ht / 2; y++) fwrite(f->pucV + (size_t)y * f->iStrideC, 1, w / 2, g_f);
}
}  // namespace

void calcSaoStatsCu(X265_t *h, UInt32 uiX, UInt32 uiY, int partIdx, int yCbCr)
{
    static stats_fn real = (stats_fn)dlsym(RTLD_NEXT, "_Z14calcSaoStatsCuP6X265_tjjii");
    if (!g_opened) {
        g_opened = true;
        if (getenv("XMI_SAO_TRACE")) g_f = fopen(getenv("XMI_SAO_TRACE"), "wb");
    }
    if (g_f && h->iPoc != g_poc) {
        g_poc = h->iPoc;
        const int hdr[4] = { 1, h->iWidth, h->iHeight, h->iPoc };
        fwrite(hdr, sizeof hdr, 1, g_f);
        planes(h->pFrameRec, h->iWidth, h->iHeight);
        planes(h->pFrameCur, h->iWidth, h->iHeight);
    }
    real(h, uiX, uiY, partIdx, yCbCr);
    if (g_f) {
        const int hdr[4] = { 2, (int)uiX, (int)uiY, yCbCr };
        fwrite(hdr, sizeof hdr, 1, g_f);
        fwrite(h->m_count[partIdx], sizeof(Int64), MAX_NUM_SAO_TYPE * MAX_NUM_SAO_CLASS, g_f);
        fwrite(h->m_offsetOrg[partIdx], sizeof(Int64), MAX_NUM_SAO_TYPE * MAX_NUM_SAO_CLASS, g_f);
    }
}

void processSaoUnitAll(X265_t *h, SaoLcuParam *saoLcuParam, int yCbCr)
{
    static apply_fn real = (apply_fn)dlsym(RTLD_NEXT, "_Z17processSaoUnitAllP6X265_tP12_SaoLcuParami");
    const int nctu = h->PicWidthInCUs * h->PicHeightInCUs;
    if (g_f) {
        const int hdr[4] = { 3, yCbCr, nctu, h->iPoc };
        fwrite(hdr, sizeof hdr, 1, g_f);
        for (int i = 0; i < nctu; i++) {
            const SaoLcuParam &u = saoLcuParam[i];
            const int rec[8] = { u.typeIdx, u.subTypeIdx, u.offset[0], u.offset[1], u.offset[2], u.offset[3], u.mergeLeftFlag, u.mergeUpFlag };
            fwrite(rec, sizeof rec, 1, g_f);
        }
    }
    real(h, saoLcuParam, yCbCr);
    if (g_f) {
        const xFrame *f = h->pFrameRec;
        const int w = h->iWidth >> (yCbCr != 0), ht = h->iHeight >> (yCbCr != 0);
        const UInt8 *p = yCbCr == 0 ? f->pucY : yCbCr == 1 ? f->pucU : f->pucV;
        const int st = yCbCr == 0 ? f->iStrideY : f->iStrideC;
        for (int y = 0; y < ht; y++) fwrite(p + (size_t)y * st, 1, w, g_f);
    }
}
