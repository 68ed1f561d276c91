// ctu_trace_hook.cpp - TEST INFRASTRUCTURE: records what the reference's CTU loop decides, CTU by CTU.
//
// Three external functions of src/encode.cpp / src/Deblock.cpp are interposed in the position-independent build of the
// reference (oracle/Makefile, target trace; calls go through the PLT, the definitions below forward with dlsym RTLD_NEXT):
//   xEncDecideCU_Inter (src/encode.cpp:3080)  depth-0 calls only: the CABAC estimation contexts on entry, the RD cost returned
//   xEncWriteCU        (src/encode.cpp:4547)  called once per CTU after the decision and the chroma pass (:5494): the caches
//                                             with every decision of the CTU, and the coefficient levels
//   xDeblockFrame      (src/Deblock.cpp:524)  called once per picture after the CTU loop (:5911): the source picture, the padded
//                                             reference, the unfiltered reconstruction, and the state that survives pictures
// Nothing of the reference is edited or copied; its headers are included for the types. The trace goes to the file named
// by XMI_CTU_TRACE (little endian), a sequence of records that start with an int32 tag:
//   tag 1, one per CTU:
//     int32 poc, slice, ctu_x, ctu_y, qp, has_cost, 0, 0;  double rd_cost (depth-0 return value, 0 in I slices)
//     uint64 frac_bits (rdoParam[ctu_y].fracBits after the CTU)
//     uint8 ctx_in[256] (estimation contexts the decision started from; P slices), uint8 ctx_out[256] (cabacRdoLast[ctu_y])
//     uint16 info0[16*16], uint8 info1[16*16], int16 mv[8*8][2], int16 mvd[8*8][2], uint8 mvp_idx[8*8]
//     int16 coef_y[64*64], coef_u[32*32], coef_v[32*32]
//   tag 2, one per picture, after its CTUs:
//     int32 poc, slice, width, height, qp, pad, ctu_rows, ctu_cols
//     source Y (width x height), U, V (width/2 x height/2)
//     reference Y ((width + 2 pad) x (height + 2 pad)), U, V (pad/2)      - garbage for the first picture
//     unfiltered reconstruction Y, U, V
//     double sum_cost[rows][cols][5], int32 count[rows][cols][5], double avg_cost[rows][cols][5]   (FAST_MD state, :3556-3568)
//     uint64 frac_bits[rows]            (after the CTU loop)
//     uint64 frac_bits_start[rows]      (P slices: when the picture's first CTU is about to be decided. The last row's counter
//                                        also serves the SAO decision of the previous picture, src/encode.cpp:1354, 1477)
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

namespace {
FILE *g_f = nullptr;
bool g_opened = false;
double g_cost = 0;
int g_has_cost = 0;
unsigned char g_ctx_in[256];
unsigned long long g_frac_start[MAX_HEIGHT_CU];

FILE *out()
{
    if (!g_opened) {
        g_opened = true;
        if (getenv("XMI_CTU_TRACE")) g_f = fopen(getenv("XMI_CTU_TRACE"), "wb");
    }
    return g_f;
}

void plane(const unsigned char *p, int stride, int w, int h, int pad)
{
    for (int y = -pad; y < h + pad; y++) fwrite(p + (long)y * stride - pad, 1, w + 2 * pad, g_f);
}
}  // namespace

typedef UInt32 (*decide_fn)(X265_t *, UInt32 *, UInt8 *, CUInt8 *, UInt8 *, UInt8 *, Int16 *, xMV *, xMV *, UInt8 *, UInt8 *, UInt16 *, UInt8 *,
                            Int16 **, UInt8 **, UInt8 **, Int16 **, Int **, CInt32, CUInt, double, CInt32, CInt, CInt, xMV, xMV, UInt, xCabac **,
                            xRdoParam *, Int, Int);

UInt32 xEncDecideCU_Inter(X265_t *h, UInt32 *pCTUCmd, UInt8 *pucPixY, CUInt8 *pucRefY, UInt8 *pucPredY, UInt8 *pucRecY, Int16 *psCoefY, xMV *pasMV,
                          xMV *pasMVD, UInt8 *paucMVPIdx, UInt8 *pucModeY, UInt16 *pusModeYInfo0, UInt8 *pucModeYInfo1, Int16 *ppsTmp[4],
                          UInt8 *paucPredTmp[MAX_CU_DEPTH], UInt8 *paucRecYTmp[MAX_CU_DEPTH], Int16 *pasCoefYTmp[MAX_CU_DEPTH],
                          Int *paiCbfYTmp[MAX_CU_DEPTH], CInt32 iSize, CUInt nQP, double dLambda, CInt32 iRefStrideY, CInt off_y, CInt off_x,
                          xMV mv_min_ipel, xMV mv_max_ipel, UInt uiDepth, xCabac *ppcCabac[MAX_CU_DEPTH + 1], xRdoParam *rdoParam, Int yCtu, Int xCtu)
{
    static decide_fn real =
        (decide_fn)dlsym(RTLD_NEXT, "_Z18xEncDecideCU_InterP6X265_tPjPhPKhS2_S2_PsP3xMVS7_S2_S2_PtS2_PS5_PS2_SA_S9_PPiijdiiiS6_S6_jPP6xCabacP9xRdoParamii");
    if (!real) {
        fprintf(stderr, "ctu_trace_hook: original xEncDecideCU_Inter not found\n");
        abort();
    }
    if (uiDepth == 0) memcpy(g_ctx_in, ppcCabac[0]->contextModels, 256);
    if (uiDepth == 0 && yCtu == 0 && xCtu == 0)
        for (int y = 0; y < (int)h->PicHeightInCUs; y++) g_frac_start[y] = h->rdoParam[y].fracBits;
    // the reference declares the return type UInt32 but returns a double (src/encode.cpp:3638): the caller at depth 0
    // ignores it, the recursive calls read the truncated value. Only the truncated value exists.
    const UInt32 r = real(h, pCTUCmd, pucPixY, pucRefY, pucPredY, pucRecY, psCoefY, pasMV, pasMVD, paucMVPIdx, pucModeY, pusModeYInfo0, pucModeYInfo1,
                          ppsTmp, paucPredTmp, paucRecYTmp, pasCoefYTmp, paiCbfYTmp, iSize, nQP, dLambda, iRefStrideY, off_y, off_x, mv_min_ipel,
                          mv_max_ipel, uiDepth, ppcCabac, rdoParam, yCtu, xCtu);
    if (uiDepth == 0) {
        g_cost = (double)r;
        g_has_cost = 1;
    }
    return r;
}

typedef UInt32 (*write_fn)(const xSliceType, UInt32 *, UInt8 *, UInt16 *, UInt8 *, xMV *, UInt8 *, Int16 *, Int16 *, Int16 *, CUInt32, CUInt32, X265_t *,
                           xMV *, CUInt16);

UInt32 xEncWriteCU(const xSliceType eSliceType, UInt32 *pCmd, UInt8 *pucModeY, UInt16 *pusModeYInfo0, UInt8 *pucModeYInfo1, xMV *pasMVD,
                   UInt8 *paucMVPIdx, Int16 *psCoefY, Int16 *psCoefU, Int16 *psCoefV, CUInt32 rx, CUInt32 ry, X265_t *h, xMV *pasMV, CUInt16 nWidth)
{
    static write_fn real = (write_fn)dlsym(RTLD_NEXT, "_Z11xEncWriteCU10xSliceTypePjPhPtS1_P3xMVS1_PsS5_S5_jjP6X265_tS4_t");
    if (!real) {
        fprintf(stderr, "ctu_trace_hook: original xEncWriteCU not found\n");
        abort();
    }
    if (out()) {
        const int cx = (int)rx / MAX_CU_SIZE, cy = (int)ry / MAX_CU_SIZE;
        const int hdr[9] = { 1, h->iPoc, (int)eSliceType, cx, cy, h->iQP, g_has_cost, 0, 0 };
        fwrite(hdr, sizeof hdr, 1, g_f);
        fwrite(&g_cost, 8, 1, g_f);
        const unsigned long long fb = h->rdoParam[cy].fracBits;
        fwrite(&fb, 8, 1, g_f);
        fwrite(g_ctx_in, 1, 256, g_f);
        fwrite(h->cabacRdoLast[cy][0].contextModels, 1, 256, g_f);
        fwrite(pusModeYInfo0, 2, MAX_PU_XY * MAX_PU_XY, g_f);
        for (int y = 0; y < MAX_PU_XY; y++) fwrite(pucModeYInfo1 + y * (MAX_PU_XY + 1), 1, MAX_PU_XY, g_f);
        for (int y = 0; y < MAX_MV_XY; y++) fwrite(pasMV + y * (MAX_MV_XY + 2), sizeof(xMV), MAX_MV_XY, g_f);
        fwrite(pasMVD, sizeof(xMV), MAX_MV_XY * MAX_MV_XY, g_f);
        fwrite(paucMVPIdx, 1, MAX_MV_XY * MAX_MV_XY, g_f);
        fwrite(psCoefY, 2, MAX_CU_SIZE * MAX_CU_SIZE, g_f);
        fwrite(psCoefU, 2, MAX_CU_SIZE * MAX_CU_SIZE / 4, g_f);
        fwrite(psCoefV, 2, MAX_CU_SIZE * MAX_CU_SIZE / 4, g_f);
        g_has_cost = 0;
        g_cost = 0;
    }
    return real(eSliceType, pCmd, pucModeY, pusModeYInfo0, pucModeYInfo1, pasMVD, paucMVPIdx, psCoefY, psCoefU, psCoefV, rx, ry, h, pasMV, nWidth);
}

typedef void (*db_fn)(X265_t *, CUInt32, CUInt32, Int32, Int32, Int32, Int32);

void xDeblockFrame(X265_t *h, CUInt32 nCUWidth, CUInt32 nCUHeight, Int32 offX, Int32 offY, Int32 bsOff, Int32 idxTiles)
{
    static db_fn real = (db_fn)dlsym(RTLD_NEXT, "_Z13xDeblockFrameP6X265_tjjiiii");
    if (out()) {
        const int w = h->iWidth, ht = h->iHeight, rows = h->PicHeightInCUs, cols = h->PicWidthInCUs;
        const int hdr[9] = { 2, h->iPoc, (int)h->eSliceType, w, ht, h->iQP, PAD_YSIZE, rows, cols };
        fwrite(hdr, sizeof hdr, 1, g_f);
        const xFrame *c = h->pFrameCur, *r = h->pFrameRef, *o = h->pFrameRec;
        plane(c->pucY, c->iStrideY, w, ht, 0);
        plane(c->pucU, c->iStrideC, w / 2, ht / 2, 0);
        plane(c->pucV, c->iStrideC, w / 2, ht / 2, 0);
        plane(r->pucY, r->iStrideY, w, ht, PAD_YSIZE);
        plane(r->pucU, r->iStrideC, w / 2, ht / 2, PAD_YSIZE / 2);
        plane(r->pucV, r->iStrideC, w / 2, ht / 2, PAD_YSIZE / 2);
        plane(o->pucY, o->iStrideY, w, ht, 0);
        plane(o->pucU, o->iStrideC, w / 2, ht / 2, 0);
        plane(o->pucV, o->iStrideC, w / 2, ht / 2, 0);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) fwrite(h->dSumOfRdCosts[y][x], 8, MAX_CU_DEPTH, g_f);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) fwrite(h->nRdCounts[y][x], 4, MAX_CU_DEPTH, g_f);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) fwrite(h->dAverageRdCosts[y][x], 8, MAX_CU_DEPTH, g_f);
        for (int y = 0; y < rows; y++) {
            const unsigned long long fb = h->rdoParam[y].fracBits;
            fwrite(&fb, 8, 1, g_f);
        }
        fwrite(g_frac_start, 8, rows, g_f);
        fflush(g_f);
    }
    real(h, nCUWidth, nCUHeight, offX, offY, bsOff, idxTiles);
}
