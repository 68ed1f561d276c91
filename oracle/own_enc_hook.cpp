// Test infrastructure (integration harness), not product code.
//
// The reference's outer encoder interface bound to this library's encoder: the three functions of src/encode.cpp that the
// reference's command line (src/x265.cpp:203, 313, 337) and its exported C API (src/interface.cpp:48-66, src/libx265.def:1-7:
// xIEncInit -> xEncInit, xIEncFrame -> xEncodeFrame, xIEncFree -> xEncFree) call, defined on xmi_encoder_* of
// include/x265_amd.h. Linked with the reference's own x265.cpp / interface.cpp / params.cpp / preProcess.cpp / rateCtrl.cpp
// (compiled where they lie, oracle/Makefile target own_enc) and with libx265_amd_dropin.so for the xSadN table that
// sceneChangeDetection (src/preProcess.cpp:132) calls - none of the reference's CTU loop, primitives, filters or entropy
// coder is in the link. This is the binding a maintainer adds to put the GPU encoder behind the reference's API.
//
// Limits, as for xmi_encoder_*: constant QP (the command line's -rc 1 changes h->iQP per picture: refused), no tiles.
#include <stdio.h>
#include <map>
#include <mutex>
#include <vector>

#include "x265.h"
#include "../include/x265_amd.h"

namespace {
struct Binding {
    xmi_encoder *enc = nullptr;
    int qp = 0;
    std::vector<UInt8> packed;      // source planes at stride = width when the caller's xFrame has another stride
};
std::mutex g_mu;
std::map<X265_t *, Binding> g_bind;      // X265_t has no user pointer: one binding per handle

Binding *binding(X265_t *h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_bind.find(h);
    return it == g_bind.end() ? nullptr : &it->second;
}

void fail(const char *what)
{
    fprintf(stderr, "own_enc_hook: %s: %s\n", what, xmi_last_error());
    exit(EXIT_FAILURE);      // the reference's own error path (src/params.cpp:94-97)
}
}      // namespace

// src/encode.cpp:5009-5160. The two padded pictures h->refn[] stay (sceneChangeDetection reads h->pFrameRec, the command line
// clears h->qp_y_tab before every picture, src/x265.cpp:308); everything else the original allocates belongs to its CTU loop.
void xEncInit(X265_t *h)
{
    CUInt32 uiWidthPad = h->iWidth + PAD_YSIZE * 2, uiHeightPad = h->iHeight + PAD_YSIZE * 2;
    CUInt32 uiYSize = uiWidthPad * uiHeightPad;
    for (Int i = 0; i < MAX_REF_NUM + 1; i++) {
        UInt8 *ptr = (UInt8 *)calloc(uiYSize * 3 / 2, 1);
        assert(ptr != NULL);
        h->refn[i].base = ptr;
        h->refn[i].pucY = ptr + PAD_YSIZE * uiWidthPad + PAD_YSIZE;
        h->refn[i].pucU = ptr + uiYSize + PAD_CSIZE * uiWidthPad / 2 + PAD_CSIZE;
        h->refn[i].pucV = ptr + uiYSize * 5 / 4 + PAD_CSIZE * uiWidthPad / 2 + PAD_CSIZE;
        h->refn[i].iStrideY = uiWidthPad;
        h->refn[i].iStrideC = uiWidthPad / 2;
    }
    h->iPoc = -1;
    h->uiFrames = 0;
    h->PicWidthInCUs = h->iWidth / h->ucMaxCUWidth;
    h->PicHeightInCUs = h->iHeight / h->ucMaxCUWidth;
    h->qp_y_tab = (UInt8 *)MALLOC(h->iWidth / h->ucMaxCUDepth * h->iHeight / h->ucMaxCUDepth);
    assert(h->qp_y_tab != NULL);
    if (h->bTiles) {
        fprintf(stderr, "own_enc_hook: tiles are not supported\n");
        exit(EXIT_FAILURE);
    }
    Binding b;
    b.qp = h->iQP;
    b.enc = xmi_encoder_create(h->iWidth, h->iHeight, h->iQP, h->bStrongIntraSmoothing ? 1 : 0);
    if (!b.enc)
        fail("xmi_encoder_create");
    if (getenv("XMI_OWN_LANES") && xmi_encoder_configure(b.enc, atoi(getenv("XMI_OWN_LANES")), 200) != 0)
        fail("xmi_encoder_configure");
    std::lock_guard<std::mutex> lk(g_mu);
    g_bind[h] = b;
}

// src/encode.cpp:5162-5200
void xEncFree(X265_t *h)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_bind.find(h);
        if (it != g_bind.end()) {
            xmi_encoder_destroy(it->second.enc);
            g_bind.erase(it);
        }
    }
    FREE(h->qp_y_tab);
    h->qp_y_tab = NULL;
    for (Int i = 0; i < MAX_REF_NUM + 1; i++)
        FREE(h->refn[i].base);
    memset(h->refn, 0, sizeof(h->refn));
}

// src/encode.cpp:5752-6118: one picture in (h->eSliceType says which kind), the access unit's bytes out.
Int32 xEncodeFrame(X265_t *h, xFrame *pFrame, UInt8 *pucOutBuf, UInt32 uiBufSize)
{
    Binding *b = binding(h);
    if (!b) {
        fprintf(stderr, "own_enc_hook: xEncodeFrame before xEncInit\n");
        exit(EXIT_FAILURE);
    }
    if (h->iQP != b->qp) {
        fprintf(stderr, "own_enc_hook: QP changed from %d to %d between pictures (rate control is not supported)\n", b->qp, h->iQP);
        exit(EXIT_FAILURE);
    }
    CInt32 w = h->iWidth, ht = h->iHeight;
    h->pFrameCur = pFrame;
    h->pFrameRef = &h->refn[0];
    h->pFrameRec = &h->refn[MAX_REF_NUM];
    h->iPoc++;
    if (h->eSliceType == SLICE_I)
        h->iPoc = 0;
    const UInt8 *y = pFrame->pucY, *u = pFrame->pucU, *v = pFrame->pucV;
    if (pFrame->iStrideY != w || pFrame->iStrideC != w / 2) {      // the library takes packed planes
        b->packed.resize((size_t)w * ht * 3 / 2);
        UInt8 *py = b->packed.data(), *pu = py + (size_t)w * ht, *pv = pu + (size_t)w * ht / 4;
        for (Int r = 0; r < ht; r++)
            memcpy(py + (size_t)r * w, pFrame->pucY + (size_t)r * pFrame->iStrideY, w);
        for (Int r = 0; r < ht / 2; r++) {
            memcpy(pu + (size_t)r * (w / 2), pFrame->pucU + (size_t)r * pFrame->iStrideC, w / 2);
            memcpy(pv + (size_t)r * (w / 2), pFrame->pucV + (size_t)r * pFrame->iStrideC, w / 2);
        }
        y = py, u = pu, v = pv;
    }
    int64_t n = xmi_encoder_encode(b->enc, y, u, v, h->eSliceType == SLICE_I ? XMI_SLICE_I : XMI_SLICE_P, pucOutBuf, uiBufSize);
    if (n < 0)
        fail("xmi_encoder_encode");
    // the reconstruction goes where the original leaves it (h->pFrameRec), then the two pictures change places as at
    // src/encode.cpp:6087-6092 - sceneChangeDetection of the next picture reads the slot h->pFrameRec after that exchange
    std::vector<UInt8> rec((size_t)w * ht * 3 / 2);
    if (xmi_encoder_reconstruction(b->enc, rec.data(), rec.data() + (size_t)w * ht, rec.data() + (size_t)w * ht * 5 / 4) != 0)
        fail("xmi_encoder_reconstruction");
    xFrame *r = h->pFrameRec;
    for (Int j = 0; j < ht; j++)
        memcpy(r->pucY + (size_t)j * r->iStrideY, rec.data() + (size_t)j * w, w);
    for (Int j = 0; j < ht / 2; j++) {
        memcpy(r->pucU + (size_t)j * r->iStrideC, rec.data() + (size_t)w * ht + (size_t)j * (w / 2), w / 2);
        memcpy(r->pucV + (size_t)j * r->iStrideC, rec.data() + (size_t)w * ht * 5 / 4 + (size_t)j * (w / 2), w / 2);
    }
    if (h->bWriteRecFlag) {      // -rec, src/encode.cpp:6056-6072
        static FILE *fpx = NULL;
        if (fpx == NULL)
            fpx = fopen("OX.YUV", "wb");
        assert(fpx != NULL);
        fwrite(rec.data(), 1, rec.size(), fpx);
        fflush(fpx);
    }
    xFrame tmp = h->refn[MAX_REF_NUM];
    for (Int i = MAX_REF_NUM; i > 0; i--)
        h->refn[i] = h->refn[i - 1];
    h->refn[0] = tmp;
    h->uiFrames++;
    return (Int32)n;
}
