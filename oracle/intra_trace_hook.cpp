// intra_trace_hook.cpp - TEST INFRASTRUCTURE: records the intra luma mode searches of a reference encode.
//
// The 35-mode loop (src/encode.cpp:4005-4046) is inline in xEncDecideCU_Intra, but everything it calls is
// reachable: xPaddingRef and xFilterRef (src/encode.cpp:3843, 3873) and xEncIntraPred (src/pixel.cpp:7338) are
// external functions called through the PLT in the position-independent build, and the SATD goes through the
// writable table xSatdN. The definitions below forward to the originals (dlsym RTLD_NEXT / saved table entries)
// and reassemble, per PU, what the loop saw: the neighbouring samples before padding, the availability bits, the
// strong-smoothing flag, the source block and the SATD of each of the 35 modes. Records go to the file named by
// XMI_INTRA_TRACE: int32 size, valids, strong, poc; raw samples (4 size + 1 bytes); source block (size^2 bytes);
// 35 uint32 SATDs.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x265.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35); xSatdN is declared in x265.h:663

typedef void (*pad_fn)(UInt8 *, CUInt8, CInt32 *);
typedef void (*fil_fn)(UInt8 *, UInt8 *, CInt32, CUInt);
typedef void (*prd_fn)(UInt8 *, UInt8 *, UInt, UInt, UInt, UInt);
typedef int (*satd_fn)(UInt8 *, intptr_t, UInt8 *, intptr_t);

namespace {
FILE *g_f = nullptr;
satd_fn g_satd[4];
bool g_installed = false;
struct {
    bool have_pad;
    int valids, strong, size, next_mode;
    UInt8 raw[4 * 64 + 1], src[32 * 32];
    UInt8 *dst;
    unsigned satd[35];
} g;

void on_satd(int idx, UInt8 *src, intptr_t ss, UInt8 *pred, unsigned value)
{
    if (g.next_mode < 0 || pred != g.dst || (4 << idx) != g.size) return;
    if (g.next_mode == 0)
        for (int y = 0; y < g.size; y++) memcpy(g.src + y * g.size, src + y * ss, g.size);
    g.satd[g.next_mode++] = value;
    g.dst = nullptr;
    if (g.next_mode == 35) {
        const int hdr[4] = { g.size, g.have_pad ? g.valids : 0, g.strong, h.iPoc };
        if (!g.have_pad) memset(g.raw, 0x80, sizeof g.raw);
        fwrite(hdr, sizeof hdr, 1, g_f);
        fwrite(g.raw, 1, 4 * g.size + 1, g_f);
        fwrite(g.src, 1, g.size * g.size, g_f);
        fwrite(g.satd, 4, 35, g_f);
        g.next_mode = -1;
        g.have_pad = false;
    }
}
template <int I>
int tramp(UInt8 *a, intptr_t sa, UInt8 *b, intptr_t sb)
{
    const int r = g_satd[I](a, sa, b, sb);
    if (g_f) on_satd(I, a, sa, b, (unsigned)r);
    return r;
}
void install()
{
    if (g_installed) return;
    g_installed = true;
    if (getenv("XMI_INTRA_TRACE")) g_f = fopen(getenv("XMI_INTRA_TRACE"), "wb");
    g.next_mode = -1;
    for (int i = 0; i < 4; i++) g_satd[i] = (satd_fn)xSatdN[i];
    xSatdN[0] = (xSatd *)tramp<0>;
    xSatdN[1] = (xSatd *)tramp<1>;
    xSatdN[2] = (xSatd *)tramp<2>;
    xSatdN[3] = (xSatd *)tramp<3>;
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
    if (!g_f) return;
    if (nMode == 0 && nStride == nSize && (!g.have_pad || g.size == (int)nSize)) {   // a search starts with planar into a packed block
        g.next_mode = 0;
        g.size = (int)nSize;
        if (!g.have_pad) g.strong = h.bStrongIntraSmoothing;
    } else if (g.next_mode < 0 || (int)nMode != g.next_mode || (int)nSize != g.size) {
        g.next_mode = -1;
        return;
    }
    g.dst = pucDstY;
}
