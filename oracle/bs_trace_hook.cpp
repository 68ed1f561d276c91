// bs_trace_hook.cpp - TEST INFRASTRUCTURE: records how a reference encode derives its deblocking boundary strengths.
//
// ff_hevc_deblocking_boundary_strengths (src/Deblock.cpp:369) is called from xEncWriteCU (src/encode.cpp:4885, 4904) once
// per transform unit, on the CTU-local caches of motion vectors (one per 8x8, pitch MAX_MV_XY + 2) and mode flags (one
// per 4x4, pitch MAX_PU_XY + 1). The definition below copies what the unit owns in those caches into picture-level maps
// - unit size per 4x4, luma cbf per 4x4, motion vector per 8x8 - and forwards to the original, which fills the
// encoder's vertical_bs / horizontal_bs arrays. At xDeblockFrame (src/Deblock.cpp:524), i.e. once the picture is
// complete, maps and arrays are written to the file named by XMI_BS_TRACE:
//   int32 width, height, poc, slice;  unit log2 size (width/4 * height/4), cbf (same), mv (width/8 * height/8 int16
//   pairs);  vertical_bs, horizontal_bs (width/4 * height/4 each).
// A restatement that derives the arrays from the maps alone must reproduce them.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "x265.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)

typedef void (*bs_fn)(UInt8 *, UInt8 *, CUInt16, int, int, int, int, xMV *, UInt8 *, xSliceType);
typedef void (*db_fn)(X265_t *, CUInt32, CUInt32, Int32, Int32, Int32, Int32);

namespace {
FILE *g_f = nullptr;
bool g_opened = false;
std::vector<UInt8> g_log2, g_cbf;
std::vector<Int16> g_mv;
int g_w = 0, g_h = 0;

void ensure()
{
    if (g_w) return;
    g_w = h.iWidth;
    g_h = h.iHeight;
    g_log2.assign((size_t)(g_w / 4) * (g_h / 4), 0);
    g_cbf.assign((size_t)(g_w / 4) * (g_h / 4), 0);
    g_mv.assign((size_t)(g_w / 8) * (g_h / 8) * 2, 0);
}
}  // namespace

void ff_hevc_deblocking_boundary_strengths(UInt8 *horizontal_bs, UInt8 *vertical_bs, CUInt16 nWidth, int x0, int y0, int log2_trafo_size,
                                           int b64x64Flag, xMV *pasMV, UInt8 *puclModeYInfo1, xSliceType eSliceType)
{
    static bs_fn real = (bs_fn)dlsym(RTLD_NEXT, "_Z37ff_hevc_deblocking_boundary_strengthsPhS_tiiiiP3xMVS_10xSliceType");
    ensure();
    if ((int)nWidth != g_w) {
        fprintf(stderr, "bs_trace_hook: strength arrays are %d wide, the picture %d\n", (int)nWidth, g_w);
        abort();
    }
    const int size = 1 << log2_trafo_size;
    for (int r = 0; r < size / 4; r++)
        for (int c = 0; c < size / 4; c++) {
            const size_t at = (size_t)(y0 / 4 + r) * (g_w / 4) + x0 / 4 + c;
            g_log2[at] = (UInt8)log2_trafo_size;
            g_cbf[at] = (UInt8)!!(puclModeYInfo1[r * (MAX_PU_XY + 1) + c] & FLAG_MASK_CBFY);
        }
    if (eSliceType != SLICE_I)      // I slices never look at motion vectors, and 4x4 units have none of their own
        for (int r = 0; r < size / 8; r++)
            for (int c = 0; c < size / 8; c++) {
                const size_t at = ((size_t)(y0 / 8 + r) * (g_w / 8) + x0 / 8 + c) * 2;
                g_mv[at] = pasMV[r * (MAX_MV_XY + 2) + c].x;
                g_mv[at + 1] = pasMV[r * (MAX_MV_XY + 2) + c].y;
            }
    real(horizontal_bs, vertical_bs, nWidth, x0, y0, log2_trafo_size, b64x64Flag, pasMV, puclModeYInfo1, eSliceType);
}

void xDeblockFrame(X265_t *ph, CUInt32 nCUWidth, CUInt32 nCUHeight, Int32 offX, Int32 offY, Int32 bsOff, Int32 idxTiles)
{
    static db_fn real = (db_fn)dlsym(RTLD_NEXT, "_Z13xDeblockFrameP6X265_tjjiiii");
    if (!g_opened) {
        g_opened = true;
        if (getenv("XMI_BS_TRACE")) g_f = fopen(getenv("XMI_BS_TRACE"), "wb");
    }
    ensure();
    if (g_f) {
        const int hdr[4] = { g_w, g_h, ph->iPoc, (int)ph->eSliceType };
        fwrite(hdr, sizeof hdr, 1, g_f);
        fwrite(g_log2.data(), 1, g_log2.size(), g_f);
        fwrite(g_cbf.data(), 1, g_cbf.size(), g_f);
        fwrite(g_mv.data(), sizeof(Int16), g_mv.size(), g_f);
        fwrite(ph->vertical_bs, 1, (size_t)(g_w / 4) * (g_h / 4), g_f);
        fwrite(ph->horizontal_bs, 1, (size_t)(g_w / 4) * (g_h / 4), g_f);
    }
    real(ph, nCUWidth, nCUHeight, offX, offY, bsOff, idxTiles);
}
