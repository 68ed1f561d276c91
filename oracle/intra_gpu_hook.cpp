// intra_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference's encoder with its intra luma mode searches served by the GPU.
//
// Same interposition as intra_trace_hook.cpp (xPaddingRef, xFilterRef, xEncIntraPred through the PLT of the
// position-independent build, the SATD through the writable table xSatdN), but instead of recording a search it
// answers it: when the 35-mode loop (src/encode.cpp:4005-4046) asks for the SATD of its first mode, the whole search
// runs on the GPU (xmi_intra_search: padding, smoothing, 35 predictions, 35 SATDs from the samples the encoder had
// before xPaddingRef) and the loop's 35 SATD calls return the GPU's numbers; the reference's own SATD code is not
// executed for them. The decision stays with the encoder's loop, so an identical bitstream means identical SATDs for
// every mode of every search.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"
#include "../include/x265_amd.h"

typedef void (*pad_fn)(UInt8 *, CUInt8, CInt32 *);
typedef void (*fil_fn)(UInt8 *, UInt8 *, CInt32, CUInt);
typedef void (*prd_fn)(UInt8 *, UInt8 *, UInt, UInt, UInt, UInt);
typedef int (*satd_fn)(UInt8 *, intptr_t, UInt8 *, intptr_t);

namespace {
satd_fn g_satd[4];
bool g_installed = false;
unsigned long g_searches = 0, g_other = 0;
struct {
    bool have_pad;
    int valids, strong, size, next_mode;
    UInt8 raw[4 * 64 + 1];
    UInt8 *dst;
    xmi_intra_result res;
} g;

int on_satd(int idx, UInt8 *src, intptr_t ss, UInt8 *pred, intptr_t ps)
{
    if (g.next_mode < 0 || pred != g.dst || (4 << idx) != g.size) {
        g_other++;
        return g_satd[idx](src, ss, pred, ps);
    }
    if (g.next_mode == 0) {
        xmi_intra_request rq;
        memset(&rq, 0, sizeof rq);
        rq.size = (uint8_t)g.size;
        rq.valids = (uint8_t)(g.have_pad ? g.valids : 0);
        rq.strong_smoothing = (uint8_t)g.strong;
        rq.mpm[0] = 0;
        rq.mpm[1] = 1;
        rq.mpm[2] = 26;
        rq.sqrt_lambda = 1.0;         // the decision is the encoder's: only the 35 SATDs are used
        if (xmi_intra_search(src, (uint32_t)ss, g.raw, &rq, &g.res) != 0) {
            fprintf(stderr, "intra_gpu_hook: %s\n", xmi_last_error());
            abort();
        }
        g_searches++;
    }
    const int v = (int)g.res.satd[g.next_mode++];
    g.dst = nullptr;
    if (g.next_mode == 35) {
        g.next_mode = -1;
        g.have_pad = false;
    }
    return v;
}
template <int I>
int tramp(UInt8 *a, intptr_t sa, UInt8 *b, intptr_t sb) { return on_satd(I, a, sa, b, sb); }
void report() { fprintf(stderr, "intra_gpu_hook: %lu searches on the GPU, %lu other SATD calls forwarded\n", g_searches, g_other); }
void install()
{
    if (g_installed) return;
    g_installed = true;
    g.next_mode = -1;
    for (int i = 0; i < 4; i++) g_satd[i] = (satd_fn)xSatdN[i];
    xSatdN[0] = (xSatd *)tramp<0>;
    xSatdN[1] = (xSatd *)tramp<1>;
    xSatdN[2] = (xSatd *)tramp<2>;
    xSatdN[3] = (xSatd *)tramp<3>;
    if (xmi_init(0) != 0) {
        fprintf(stderr, "intra_gpu_hook: %s\n", xmi_last_error());
        abort();
    }
    atexit(report);
}
}  // namespace

void xPaddingRef(UInt8 *pucRef, CUInt8 ucValids, CInt32 nBlkOffset[6])
{
    static pad_fn real = (pad_fn)dlsym(RTLD_NEXT, "_Z11xPaddingRefPhhPKi");
    install();
    const int size = nBlkOffset[1];
    memcpy(g.raw, pucRef, 4 * size + 1);
    g.valids = ucValids;
    g.size = size;
    g.have_pad = true;
    real(pucRef, ucValids, nBlkOffset);
}

void xFilterRef(UInt8 *pucDst, UInt8 *pucSrc, CInt32 iSize, CUInt bStrongFilter)
{
    static fil_fn real = (fil_fn)dlsym(RTLD_NEXT, "_Z10xFilterRefPhS_ij");
    install();
    g.strong = (int)bStrongFilter;
    real(pucDst, pucSrc, iSize, bStrongFilter);
}

void xEncIntraPred(UInt8 *pucDstY, UInt8 *pucRefY, UInt nStride, UInt nMode, UInt nSize, UInt bIsLuma)
{
    static prd_fn real = (prd_fn)dlsym(RTLD_NEXT, "_Z13xEncIntraPredPhS_jjjj");
    install();
    real(pucDstY, pucRefY, nStride, nMode, nSize, bIsLuma);
    if (nMode == 0 && nStride == nSize && (!g.have_pad || g.size == (int)nSize)) {   // a search starts with planar into a packed block
        g.next_mode = 0;
        g.size = (int)nSize;
        if (!g.have_pad) memset(g.raw, 0x80, sizeof g.raw);
    } else if (g.next_mode < 0 || (int)nMode != g.next_mode || (int)nSize != g.size) {
        g.next_mode = -1;
        return;
    }
    g.dst = pucDstY;
}
