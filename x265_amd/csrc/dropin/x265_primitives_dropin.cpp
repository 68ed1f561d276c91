// libx265_amd_dropin.so - the reference's pixel-primitive interface (C++ linkage, same names, same
// signatures) implemented on libx265_amd.so.
//
// The reference has no registration API: its "plugin table" is a set of writable global arrays of
// function pointers plus free functions: xSubDct, xQuant, xDeQuant, xIDctAdd (inc/x265.h:650-670, defined in
// src/pixel.cpp:529-542, 711-717, 810-816, 833-857, 1054, 1244, 3446, 5938) and the prediction functions
// xPredInterLumaBlk, xPredInterChromaBlk, xEncIntraPred (declared in src/encode.cpp:38, 60-79, defined in
// src/pixel.cpp:8745, 9657, 7338), plus xFillMvpCand of src/encode.cpp:2703. This file defines exactly those symbols, so that
//   (a) an encoder build that links this object INSTEAD of its own definitions uses the GPU, or
//   (b) an unmodified encoder calls xmi_install_primitives() once to overwrite the entries of its own
//       tables (they are non-const globals) and wraps the four free functions at link time.
// Argument meaning and the quirks that callers rely on are kept: xSubDct writes dst with the SOURCE
// stride, xIDctAdd reads pred with the COEFFICIENT stride, xQuant returns the sum of |level|, the
// temporary-buffer arguments are accepted and ignored (the GPU path needs no caller scratch).
// Errors: the reference's primitives cannot fail; here a failing GPU call aborts with the library's
// message rather than returning a wrong value.
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../../include/x265_amd.h"

typedef unsigned char UInt8;
typedef short Int16;
typedef int Int;
typedef int Int32;
typedef unsigned int UInt32;
typedef UInt32 UInt;
typedef const UInt8 CUInt8;
typedef const UInt32 CUInt32;
typedef const UInt CUInt;
typedef UInt8 pixel;
typedef enum { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2, X265_MAX_SLICE_TYPE = 3 } xSliceType;  // inc/x265.h:76-81

#define MAX_CU_DEPTH 5  // inc/config.h:96

typedef UInt32 xSad(CUInt8 *pSrc, CUInt32 nStrideSrc, CUInt8 *pRef, CUInt32 nStrideRef);
typedef Int32 xSatd(pixel *pSrc, intptr_t nStrideSrc, pixel *pRef, intptr_t nStrideRef);
typedef int xSSD(pixel *, intptr_t, pixel *, intptr_t);
typedef UInt32 xME_SAD(CUInt8 *, CUInt, CUInt8 *, CUInt);
typedef UInt32 xME_SATD(CUInt8 *, CUInt, CUInt8 *, CUInt);

static int64_t must(int64_t v, const char *what)
{
    if (v < 0) {
        fprintf(stderr, "x265_amd drop-in: %s failed: %s\n", what, xmi_last_error());
        abort();
    }
    return v;
}

#define SAD_FN(N) \
    static UInt32 gpuSad##N(CUInt8 *s, CUInt32 ss, CUInt8 *r, CUInt32 rs) { return (UInt32)must(xmi_sad(N, s, ss, r, rs), "xSad"); }
SAD_FN(4) SAD_FN(8) SAD_FN(16) SAD_FN(32) SAD_FN(64)
#define SATD_FN(N) \
    static Int32 gpuSatd##N(pixel *s, intptr_t ss, pixel *r, intptr_t rs) { return (Int32)must(xmi_satd(N, s, ss, r, rs), "xSatd"); }
SATD_FN(4) SATD_FN(8) SATD_FN(16) SATD_FN(32)
#define SSD_FN(N) \
    static int gpuSsd##N(pixel *s, intptr_t ss, pixel *r, intptr_t rs) { return (int)must(xmi_ssd(N, s, ss, r, rs), "xSsd"); }
SSD_FN(4) SSD_FN(8) SSD_FN(16) SSD_FN(32) SSD_FN(64)
#define HAD_FN(N) \
    static UInt32 gpuHad##N(CUInt8 *s, CUInt ss, CUInt8 *r, CUInt rs) { return (UInt32)must(xmi_satd_hm(N, s, ss, r, rs), "xCalcHADs"); }
HAD_FN(8) HAD_FN(16) HAD_FN(32) HAD_FN(64)

// ---- the tables, with the reference's names (src/pixel.cpp:529, 711, 810, 833, 846) ----
// The k_* copies are what xmi_install_primitives() hands out: when this library is loaded into an
// encoder that still defines its own xSadN etc., the exported names below resolve to the ENCODER's
// arrays (ELF interposition), the static ones never do.
#define SAD_LIST gpuSad4, gpuSad8, gpuSad16, gpuSad32, gpuSad64
#define SATD_LIST gpuSatd4, gpuSatd8, gpuSatd16, gpuSatd32
#define SSD_LIST gpuSsd4, gpuSsd8, gpuSsd16, gpuSsd32, gpuSsd64
// index = 2*xLog2(w-1) + xLog2(h-1) - 9; the encoder only uses the square entries 0, 3, 6, 9 (src/encode.cpp:2841, 3135)
#define SAD_AMP_LIST gpuSad8, nullptr, nullptr, gpuSad16, nullptr, nullptr, gpuSad32, nullptr, nullptr, gpuSad64
#define SATD_AMP_LIST gpuHad8, nullptr, nullptr, gpuHad16, nullptr, nullptr, gpuHad32, nullptr, nullptr, gpuHad64
static xSad *const k_sadN[MAX_CU_DEPTH] = { SAD_LIST };
static xSatd *const k_satdN[MAX_CU_DEPTH] = { SATD_LIST };
static xSSD *const k_ssdN[MAX_CU_DEPTH] = { SSD_LIST };
static xME_SAD *const k_sadAmp[10] = { SAD_AMP_LIST };
static xME_SATD *const k_satdAmp[10] = { SATD_AMP_LIST };
xSad *xSadN[MAX_CU_DEPTH] = { SAD_LIST };
xSatd *xSatdN[MAX_CU_DEPTH] = { SATD_LIST };
xSSD *xSsdN[MAX_CU_DEPTH] = { SSD_LIST };
xME_SAD *xSad_AMP[10] = { SAD_AMP_LIST };
xME_SATD *xSatd_AMP[10] = { SATD_AMP_LIST };

// ---- the free functions (inc/x265.h:652-655) ----
void xSubDct(Int16 *pDst, UInt8 *pSrc, UInt nStride, UInt8 *pRef, UInt nStrideRef, Int16 *, Int16 *, Int iWidth, UInt nMode)
{
    must(xmi_sub_dct(pDst, pSrc, nStride, pRef, nStrideRef, iWidth, nMode), "xSubDct");
}

void xIDctAdd(UInt8 *pDst, UInt nStrideDst, Int16 *pSrc, UInt nStride, UInt8 *pRef, Int16 *, Int16 *, Int iWidth, UInt nMode)
{
    must(xmi_idct_add(pDst, nStrideDst, pSrc, nStride, pRef, iWidth, nMode), "xIDctAdd");
}

UInt32 xQuant(Int16 *pDst, Int16 *pSrc, UInt nStride, UInt nQP, Int iWidth, Int iHeight, xSliceType eSType)
{
    return (UInt32)must(xmi_quant(pDst, pSrc, nStride, nQP, iWidth, iHeight, (int)eSType), "xQuant");
}

void xDeQuant(Int16 *pDst, Int16 *pSrc, UInt nStride, UInt nQP, Int iWidth, Int iHeight)
{
    must(xmi_dequant(pDst, pSrc, nStride, nQP, iWidth, iHeight), "xDeQuant");
}

// ---- the prediction functions (src/encode.cpp:38, 60-79) ----
struct xMV { Int16 x, y; };   // inc/x265.h:156-159

void xPredInterLumaBlk(CUInt8 *pucRef, UInt8 *pucPred, CUInt32 nRefStride, CUInt nSizeX, CUInt nSizeY, xMV bmv, Int16 *)
{
    must(xmi_pred_inter_luma(pucRef, pucPred, nRefStride, nSizeX, nSizeY, bmv.x, bmv.y), "xPredInterLumaBlk");
}

void xPredInterChromaBlk(CUInt8 *pucRefC, UInt8 *pucPredC, CUInt32 nRefStrideC, CUInt nSizeCX, CUInt nSizeCY, xMV bmv, Int16 *)
{
    must(xmi_pred_inter_chroma(pucRefC, pucPredC, nRefStrideC, nSizeCX, nSizeCY, bmv.x, bmv.y), "xPredInterChromaBlk");
}

// src/encode.cpp:2703: MV predictor and merge candidates of one PU from the CTU-local caches
Int xFillMvpCand(xMV *psMVP, CUInt8 *pucModeY, const xMV *psMV, CUInt nSizeX, CUInt nSizeY, CUInt32 ePUMode, CUInt32 nPUIdx)
{
    return (Int)must(xmi_fill_mvp_cand(reinterpret_cast<int16_t (*)[2]>(psMVP), pucModeY, reinterpret_cast<const int16_t (*)[2]>(psMV), nSizeX, nSizeY,
                                       ePUMode, nPUIdx), "xFillMvpCand");
}

void xEncIntraPred(UInt8 *pucDstY, UInt8 *pucRefY, UInt nStride, UInt nMode, UInt nSize, UInt bIsLuma)
{
    must(xmi_enc_intra_pred(pucDstY, pucRefY, nStride, nMode, nSize, bIsLuma), "xEncIntraPred");
}

// ---- hook for an unmodified encoder: overwrite the entries of ITS tables (passed by address) ----
extern "C" void xmi_install_primitives(xSad **sadN, xME_SAD **sadAmp, xSatd **satdN, xSSD **ssdN, xME_SATD **satdAmp)
{
    if (sadN) for (int i = 0; i < MAX_CU_DEPTH; i++) sadN[i] = k_sadN[i];
    if (sadAmp) for (int i = 0; i < 10; i += 3) sadAmp[i] = k_sadAmp[i];
    if (satdN) for (int i = 0; i < 4; i++) satdN[i] = k_satdN[i];
    if (ssdN) for (int i = 0; i < MAX_CU_DEPTH; i++) ssdN[i] = k_ssdN[i];
    if (satdAmp) for (int i = 0; i < 10; i += 3) satdAmp[i] = k_satdAmp[i];
}
