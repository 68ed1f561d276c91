// deblock_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference's encoder with its in-loop deblocking done by the GPU.
//
// xDeblockFrame (src/Deblock.cpp:524) is called once per picture through the PLT of the position-independent build;
// this definition hands the reconstructed planes, the boundary strengths and the QP table the CTU loop filled to
// xmi_deblock_frame and never calls the original. The reconstruction feeds the next picture's prediction and the MD5
// SEI of the stream, so an identical bitstream means identical filtered planes for every picture.
#include <cstdio>
#include <cstdlib>

#include "x265.h"
#include "../include/x265_amd.h"

// defined by bs_gpu_hook.cpp when that hook is linked in: derives the boundary strengths of the finished picture on the GPU
extern "C" void xmi_hook_fill_strengths(X265_t *) __attribute__((weak));

namespace {
unsigned long g_pictures = 0;
void report() { fprintf(stderr, "deblock_gpu_hook: %lu pictures deblocked on the GPU\n", g_pictures); }
}  // namespace

void xDeblockFrame(X265_t *h, CUInt32, CUInt32, Int32, Int32, Int32, Int32)
{
    static bool once = false;
    if (!once) {
        once = true;
        if (xmi_init(0) != 0) {
            fprintf(stderr, "deblock_gpu_hook: %s\n", xmi_last_error());
            abort();
        }
        atexit(report);
    }
    const int nctb = h->sps->pic_width_in_ctbs * h->sps->pic_height_in_ctbs;
    for (int i = 0; i < nctb; i++)
        if (h->deblock[i].disable || h->deblock[i].tc_offset != h->deblock[0].tc_offset || h->deblock[i].beta_offset != h->deblock[0].beta_offset ||
            h->sps->log2_min_coding_block_size != 2) {
            fprintf(stderr, "deblock_gpu_hook: per-CTB deblocking parameters are not supported\n");
            abort();
        }
    if (xmi_hook_fill_strengths) xmi_hook_fill_strengths(h);
    xFrame *f = h->pFrameRec;
    xmi_deblock_params p = { h->sps->pic_width_in_luma_samples, h->sps->pic_height_in_luma_samples, f->iStrideY, f->iStrideC,
                             h->deblock[0].tc_offset, h->deblock[0].beta_offset, h->pps_cb_qp_offset, h->pps_cr_qp_offset };
    if (xmi_deblock_frame(f->pucY, f->pucU, f->pucV, &p, h->vertical_bs, h->horizontal_bs, h->qp_y_tab) != 0) {
        fprintf(stderr, "deblock_gpu_hook: %s\n", xmi_last_error());
        abort();
    }
    g_pictures++;
}
