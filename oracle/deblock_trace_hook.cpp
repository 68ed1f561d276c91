// deblock_trace_hook.cpp - TEST INFRASTRUCTURE: records the in-loop deblocking of a reference encode.
//
// xDeblockFrame (src/Deblock.cpp:524) is an external function called once per picture from xEncodeFrame
// (src/encode.cpp:5911) through the PLT of the position-independent build. The definition below copies what the filter
// reads - the reconstructed planes, the boundary-strength arrays and the QP table the CTU loop filled, the offsets -
// forwards to the original, and copies the planes again. Records go to the file named by XMI_DEBLOCK_TRACE:
//   int32 width, height, poc, slice, tc_offset, beta_offset, cb_qp_offset, cr_qp_offset;
//   Y, U, V before (width x height, width/2 x height/2 twice); vertical_bs, horizontal_bs (width/4 * height/4 each);
//   qp table (width/4 * height/4, one entry per 4x4 block); Y, U, V after.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

typedef void (*db_fn)(X265_t *, CUInt32, CUInt32, Int32, Int32, Int32, Int32);

namespace {
FILE *g_f = nullptr;
bool g_opened = false;

void planes(const X265_t *h, int w, int ht)
{
    const xFrame *f = h->pFrameRec;
    for (int y = 0; y < ht; y++) fwrite(f->pucY + (size_t)y * f->iStrideY, 1, w, g_f);
    for (int y = 0; y < ht / 2; y++) fwrite(f->pucU + (size_t)y * f->iStrideC, 1, w / 2, g_f);
    for (int y = 0; y < ht / 2; y++) fwrite(f->pucV + (size_t)y * f->iStrideC, 1, w / 2, g_f);
}
}  // namespace

void xDeblockFrame(X265_t *h, CUInt32 nCUWidth, CUInt32 nCUHeight, Int32 offX, Int32 offY, Int32 bsOff, Int32 idxTiles)
{
    static db_fn real = (db_fn)dlsym(RTLD_NEXT, "_Z13xDeblockFrameP6X265_tjjiiii");
    if (!g_opened) {
        g_opened = true;
        if (getenv("XMI_DEBLOCK_TRACE")) g_f = fopen(getenv("XMI_DEBLOCK_TRACE"), "wb");
    }
    const int w = h->sps->pic_width_in_luma_samples, ht = h->sps->pic_height_in_luma_samples;
    if (g_f) {
        const int nctb = h->sps->pic_width_in_ctbs * h->sps->pic_height_in_ctbs;
        for (int i = 1; i < nctb; i++)
            if (h->deblock[i].tc_offset != h->deblock[0].tc_offset || h->deblock[i].beta_offset != h->deblock[0].beta_offset || h->deblock[i].disable) {
                fprintf(stderr, "deblock_trace_hook: per-CTB deblocking parameters differ\n");
                abort();
            }
        const int hdr[8] = { w, ht, h->iPoc, (int)h->eSliceType, h->deblock[0].tc_offset, h->deblock[0].beta_offset, h->pps_cb_qp_offset, h->pps_cr_qp_offset };
        fwrite(hdr, sizeof hdr, 1, g_f);
        planes(h, w, ht);
        fwrite(h->vertical_bs, 1, (size_t)(w / 4) * (ht / 4), g_f);
        fwrite(h->horizontal_bs, 1, (size_t)(w / 4) * (ht / 4), g_f);
        if (h->sps->log2_min_coding_block_size != 2) {
            fprintf(stderr, "deblock_trace_hook: QP table granularity is not 4x4\n");
            abort();
        }
        fwrite(h->qp_y_tab, 1, (size_t)(w / 4) * (ht / 4), g_f);
    }
    real(h, nCUWidth, nCUHeight, offX, offY, bsOff, idxTiles);
    if (g_f) planes(h, w, ht);
}
