// bs_gpu_hook.cpp - TEST INFRASTRUCTURE: the reference's encoder with its deblocking boundary strengths derived on the GPU.
//
// ff_hevc_deblocking_boundary_strengths (src/Deblock.cpp:369) is called from xEncWriteCU once per unit through the PLT
// of the position-independent build. The definition below never calls the original: it only copies what the unit owns
// in the CTU-local caches (unit size and luma cbf per 4x4, motion vector per 8x8) into picture-level maps. When the
// picture is complete, xmi_hook_fill_strengths() hands the maps to xmi_deblock_strengths, which writes the encoder's
// vertical_bs / horizontal_bs arrays in one launch; the deblocking that follows (the reference's, or the GPU's in
// deblock_gpu_hook.cpp, which calls this function when it is linked in) reads them. Built with -DBS_HOOK_STANDALONE the
// file also interposes xDeblockFrame to do that call and forward to the reference's filter.
// The encoder's bitstream must not change.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "x265.h"
#include "../include/x265_amd.h"

namespace {
std::vector<UInt8> g_log2, g_cbf;
std::vector<Int16> g_mv;
int g_w = 0, g_h = 0;
unsigned long g_units = 0, g_pictures = 0;

void report() { fprintf(stderr, "bs_gpu_hook: %lu units recorded, strengths of %lu pictures derived on the GPU\n", g_units, g_pictures); }

void ensure(int w, int ht)
{
    if (g_w) return;
    g_w = w;
    g_h = ht;
    g_log2.assign((size_t)(w / 4) * (ht / 4), 0);
    g_cbf.assign((size_t)(w / 4) * (ht / 4), 0);
    g_mv.assign((size_t)(w / 8) * (ht / 8) * 2, 0);
    if (xmi_init(0) != 0) {
        fprintf(stderr, "bs_gpu_hook: %s\n", xmi_last_error());
        abort();
    }
    atexit(report);
}
}  // namespace

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

void ff_hevc_deblocking_boundary_strengths(UInt8 *, UInt8 *, CUInt16 nWidth, int x0, int y0, int log2_trafo_size, int, xMV *pasMV,
                                           UInt8 *puclModeYInfo1, xSliceType eSliceType)
{
    ensure(nWidth, h.iHeight);
    const int size = 1 << log2_trafo_size;
    if ((int)nWidth != g_w || x0 + size > g_w || y0 + size > g_h) {
        fprintf(stderr, "bs_gpu_hook: unit %d,%d size %d outside the %dx%d maps\n", x0, y0, size, g_w, g_h);
        abort();
    }
    for (int r = 0; r < size / 4; r++)
        for (int c = 0; c < size / 4; c++) {
            const size_t at = (size_t)(y0 / 4 + r) * (g_w / 4) + x0 / 4 + c;
            g_log2[at] = (UInt8)log2_trafo_size;
            g_cbf[at] = (UInt8)!!(puclModeYInfo1[r * (MAX_PU_XY + 1) + c] & FLAG_MASK_CBFY);
        }
    if (eSliceType != SLICE_I)
        for (int r = 0; r < size / 8; r++)
            for (int c = 0; c < size / 8; c++) {
                const size_t at = ((size_t)(y0 / 8 + r) * (g_w / 8) + x0 / 8 + c) * 2;
                g_mv[at] = pasMV[r * (MAX_MV_XY + 2) + c].x;
                g_mv[at + 1] = pasMV[r * (MAX_MV_XY + 2) + c].y;
            }
    g_units++;
}

extern "C" void xmi_hook_fill_strengths(X265_t *ph)
{
    ensure(ph->iWidth, ph->iHeight);
    // the arrays are (width/4)*(height/4) bytes, cleared per picture (src/encode.cpp:5220); the library writes the
    // (width/8)*(height/4) entries the filter reads
    memset(ph->vertical_bs, 0xee, (size_t)(g_w / 8) * (g_h / 4));      // nothing the reference's own calls left may survive
    memset(ph->horizontal_bs, 0xee, (size_t)(g_w / 8) * (g_h / 4));
    if (xmi_deblock_strengths(g_w, g_h, ph->eSliceType == SLICE_I, g_log2.data(), g_cbf.data(), g_mv.data(), ph->vertical_bs, ph->horizontal_bs) != 0) {
        fprintf(stderr, "bs_gpu_hook: %s\n", xmi_last_error());
        abort();
    }
    g_pictures++;
}

#ifdef BS_HOOK_STANDALONE
typedef void (*db_fn)(X265_t *, CUInt32, CUInt32, Int32, Int32, Int32, Int32);
void xDeblockFrame(X265_t *ph, CUInt32 a, CUInt32 b, Int32 c, Int32 d, Int32 e, Int32 f)
{
    static db_fn real = (db_fn)dlsym(RTLD_NEXT, "_Z13xDeblockFrameP6X265_tjjiiii");
    xmi_hook_fill_strengths(ph);
    real(ph, a, b, c, d, e, f);
}
#endif
