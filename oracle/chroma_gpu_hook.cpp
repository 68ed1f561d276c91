// chroma_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference's encoder with its intra chroma mode searches served by the GPU.
//
// Same interposition as chroma_trace_hook.cpp (xEncodeChromaCU and xEncIntraPred through the PLT of the position-independent
// build, the SATD through the writable table xSatdN), but instead of recording a search it answers it: when the candidate
// loop (src/encode.cpp:4330-4362) asks for its first SATD, the whole search runs on the GPU (xmi_intra_chroma_search: 5
// candidates, U and V, from the padded samples the loop handed to xEncIntraPred and the CU's source blocks) and the loop's
// ten SATD calls return the GPU's numbers; the reference's SATD code is not executed for them. The decision stays with the
// encoder's loop, so an identical bitstream means identical SATDs in every search.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"
#include "../include/x265_amd.h"

typedef void (*prd_fn)(UInt8 *, UInt8 *, UInt, UInt, UInt, UInt);
typedef int (*satd_fn)(UInt8 *, intptr_t, UInt8 *, intptr_t);
typedef void (*chroma_fn)(UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt16 *, UInt8 *,
                          Int16 *, Int16 *, Int16 **, UInt8 *, UInt);

namespace {
satd_fn g_satd[4];
bool g_installed = false, g_in_chroma = false;
unsigned long g_searches = 0, g_other = 0;
int g_pair = 0, g_pred = 0, g_size = 0;
UInt8 *g_dst[2] = { nullptr, nullptr };
UInt8 g_ref[2][4 * 32 + 1];
UInt8 *g_pix_u = nullptr, *g_pix_v = nullptr, *g_mode_y = nullptr;
xmi_chroma_result g_res;

void die(const char *what)
{
    fprintf(stderr, "chroma_gpu_hook: %s: %s\n", what, xmi_last_error());
    abort();
}

int on_satd(int idx, UInt8 *src, intptr_t ss, UInt8 *pred, intptr_t ps)
{
    const int m = g_pair >> 1, pl = g_pair & 1;
    if (!g_in_chroma || pred != g_dst[pl] || (4 << idx) != g_size) {
        g_other++;
        return g_satd[idx](src, ss, pred, ps);
    }
    if (g_pair == 0) {
        // the U block's place in the CTU's source cache gives the V block and the CU's luma mode (src/encode.cpp:4268-4278)
        const long off = src - g_pix_u;
        const int dy = (int)(off / (2 * MAX_CU_SIZE / 2)), dx = (int)(off % (2 * MAX_CU_SIZE / 2)) / 2;
        if (off < 0 || ss != MAX_CU_SIZE / 2) die("source block outside the CTU cache");
        const int luma = g_mode_y[(MAX_PU_XY + 2) * dy + dx];
        if (xmi_intra_chroma_search(src, g_pix_v + off, (uint32_t)ss, g_ref[0], g_ref[1], g_size, luma, &g_res) != 0) die("xmi_intra_chroma_search");
        g_searches++;
    }
    const int v = (int)(pl ? g_res.satd_v[m] : g_res.satd_u[m]);
    g_dst[pl] = nullptr;
    if (++g_pair == 10) g_pair = g_pred = 0;
    return v;
}
template <int I>
int tramp(UInt8 *a, intptr_t sa, UInt8 *b, intptr_t sb) { return on_satd(I, a, sa, b, sb); }
void report() { fprintf(stderr, "chroma_gpu_hook: %lu searches on the GPU, %lu other SATD calls forwarded\n", g_searches, g_other); }
void install()
{
    if (g_installed) return;
    g_installed = true;
    for (int i = 0; i < 4; i++) g_satd[i] = (satd_fn)xSatdN[i];
    xSatdN[0] = (xSatd *)tramp<0>;
    xSatdN[1] = (xSatd *)tramp<1>;
    xSatdN[2] = (xSatd *)tramp<2>;
    xSatdN[3] = (xSatd *)tramp<3>;
    if (xmi_init(0) != 0) die("xmi_init");
    atexit(report);
}
}  // namespace

void xEncIntraPred(UInt8 *pucDst, UInt8 *pucRef, UInt nStride, UInt nMode, UInt nSize, UInt bIsLuma)
{
    static prd_fn real = (prd_fn)dlsym(RTLD_NEXT, "_Z13xEncIntraPredPhS_jjjj");
    install();
    real(pucDst, pucRef, nStride, nMode, nSize, bIsLuma);
    if (!g_in_chroma) return;
    const int m = g_pred >> 1, pl = g_pred & 1;
    if (g_pred == 0) g_size = (int)nSize;
    if (m == 0) memcpy(g_ref[pl], pucRef, 4 * nSize + 1);
    g_dst[pl] = pucDst;
    g_pred++;
}

void xEncodeChromaCU(UInt8 *a0, UInt8 *a1, UInt8 *a2, UInt8 *a3, UInt8 *a4, UInt8 *a5, UInt8 *a6, UInt8 *a7, UInt8 *a8, UInt8 *a9, UInt8 *pucModeY,
                     UInt16 *pucModeYInfo0, UInt8 *pucModeYInfo1, Int16 *c0, Int16 *c1, Int16 **tmp, UInt8 *predTmp, UInt nQPC)
{
    static chroma_fn real = (chroma_fn)dlsym(RTLD_NEXT, "_Z15xEncodeChromaCUPhS_S_S_S_S_S_S_S_S_S_PtS_PsS1_PS1_S_j");
    install();
    g_in_chroma = true;
    g_pair = g_pred = 0;
    g_pix_u = a0;
    g_pix_v = a1;
    g_mode_y = pucModeY;
    real(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, pucModeY, pucModeYInfo0, pucModeYInfo1, c0, c1, tmp, predTmp, nQPC);
    g_in_chroma = false;
}
