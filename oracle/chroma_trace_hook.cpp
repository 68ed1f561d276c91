// chroma_trace_hook.cpp - TEST INFRASTRUCTURE: records the intra chroma mode searches of a reference encode.
//
// xEncodeChromaCU (src/encode.cpp:4234-4418) tries five modes per CU - planar, vertical, horizontal, DC and the luma
// mode, a candidate equal to the luma mode replaced by 34 (:4326-4334) - predicts U and V with xEncIntraPred (not luma,
// destination stride 64), measures both with xSatdN[log2 size - 2] (:4353-4354) and keeps the first smallest sum. All
// three are reachable in the position-independent build (PLT, writable table). The definitions below forward to the
// originals and reassemble, per CU: size, the five modes, the padded neighbouring samples of U and V, both source
// blocks, the ten SATDs; the mode index the encoder kept is read from its flag array when xEncodeChromaCU returns
// (FLAG_OFF_MODEC, :4412), the CUs taken in the function's own order. Records go to the file named by XMI_CHROMA_TRACE:
//   int32 size, poc, best, modes[5];  ref U, ref V (4 size + 1 bytes each);  source U, V (size^2 each);  uint32 satd U[5], V[5].
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "x265.h"

extern X265_t h;   // the CLI's encoder handle (src/x265.cpp:35)
UInt getInterleaveBits(UInt8 x);   // src/encode.cpp:4220, external

typedef void (*prd_fn)(UInt8 *, UInt8 *, UInt, UInt, UInt, UInt);
typedef int (*satd_fn)(UInt8 *, intptr_t, UInt8 *, intptr_t);
typedef void (*chroma_fn)(UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt8 *, UInt16 *, UInt8 *,
                          Int16 *, Int16 *, Int16 **, UInt8 *, UInt);

namespace {
struct Rec {
    int size, modes[5];
    UInt8 ref[2][4 * 32 + 1], src[2][32 * 32];
    unsigned satd[2][5];
};
FILE *g_f = nullptr;
satd_fn g_satd[4];
bool g_installed = false, g_in_chroma = false;
std::vector<Rec> g_recs;       // searches of the xEncodeChromaCU call in progress
Rec g_cur;
int g_pair = -1;               // next (mode, plane) pair to be measured in the search in progress: 2 * mode index + plane
int g_pred = 0;                // next pair to be predicted (U and V are predicted before either is measured)
UInt8 *g_dst[2] = { nullptr, nullptr };

void on_satd(int idx, UInt8 *src, intptr_t ss, UInt8 *pred, unsigned value)
{
    const int m = g_pair >> 1, pl = g_pair & 1;
    if (!g_in_chroma || pred != g_dst[pl] || (4 << idx) != g_cur.size) return;
    if (m == 0)
        for (int y = 0; y < g_cur.size; y++) memcpy(g_cur.src[pl] + y * g_cur.size, src + y * ss, g_cur.size);
    g_cur.satd[pl][m] = value;
    g_dst[pl] = nullptr;
    if (++g_pair == 10) {
        g_recs.push_back(g_cur);
        g_pair = g_pred = 0;
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
    if (getenv("XMI_CHROMA_TRACE")) g_f = fopen(getenv("XMI_CHROMA_TRACE"), "wb");
    for (int i = 0; i < 4; i++) g_satd[i] = (satd_fn)xSatdN[i];
    xSatdN[0] = (xSatd *)tramp<0>;
    xSatdN[1] = (xSatd *)tramp<1>;
    xSatdN[2] = (xSatd *)tramp<2>;
    xSatdN[3] = (xSatd *)tramp<3>;
}
}  // namespace

void xEncIntraPred(UInt8 *pucDst, UInt8 *pucRef, UInt nStride, UInt nMode, UInt nSize, UInt bIsLuma)
{
    static prd_fn real = (prd_fn)dlsym(RTLD_NEXT, "_Z13xEncIntraPredPhS_jjjj");
    install();
    real(pucDst, pucRef, nStride, nMode, nSize, bIsLuma);
    if (!g_f || !g_in_chroma) return;
    if (bIsLuma || nStride != MAX_CU_SIZE || g_pair < 0) {
        fprintf(stderr, "chroma_trace_hook: unexpected prediction call inside xEncodeChromaCU\n");
        abort();
    }
    const int m = g_pred >> 1, pl = g_pred & 1;
    if (g_pred == 0) g_cur.size = (int)nSize;
    if (pl == 0) g_cur.modes[m] = (int)nMode;
    if ((int)nSize != g_cur.size || (int)nMode != g_cur.modes[m]) {
        fprintf(stderr, "chroma_trace_hook: search out of step\n");
        abort();
    }
    if (m == 0) memcpy(g_cur.ref[pl], pucRef, 4 * nSize + 1);
    g_dst[pl] = pucDst;
    g_pred++;
}

void xEncodeChromaCU(UInt8 *a0, UInt8 *a1, UInt8 *a2, UInt8 *a3, UInt8 *a4, UInt8 *a5, UInt8 *a6, UInt8 *a7, UInt8 *a8, UInt8 *a9, UInt8 *pucModeY,
                     UInt16 *pucModeYInfo0, UInt8 *pucModeYInfo1, Int16 *c0, Int16 *c1, Int16 **tmp, UInt8 *predTmp, UInt nQPC)
{
    static chroma_fn real = (chroma_fn)dlsym(RTLD_NEXT, "_Z15xEncodeChromaCUPhS_S_S_S_S_S_S_S_S_S_PtS_PsS1_PS1_S_j");
    install();
    g_in_chroma = true;
    g_recs.clear();
    g_pair = g_pred = 0;
    real(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, pucModeY, pucModeYInfo0, pucModeYInfo1, c0, c1, tmp, predTmp, nQPC);
    g_in_chroma = false;
    if (!g_f) return;
    if (g_pair != 0) {
        fprintf(stderr, "chroma_trace_hook: incomplete search at the end of a CTU\n");
        abort();
    }
    // the CUs in the function's own order (src/encode.cpp:4265-4417)
    size_t k = 0;
    for (UInt nIdx = 0; nIdx != MAX_PU_XY * MAX_PU_XY; k++) {
        const int dx = getInterleaveBits((UInt8)nIdx), dy = getInterleaveBits((UInt8)(nIdx >> 1));
        int sizeY = 1 << (2 + (pucModeYInfo1[(MAX_PU_XY + 1) * dy + dx] >> FLAG_OFF_SIZE));
        const int size = sizeY == 4 ? 4 : sizeY >> 1;
        const int best = (pucModeYInfo0[dy * MAX_PU_XY + dx] >> FLAG_OFF_MODEC) & 7;
        if (k >= g_recs.size() || g_recs[k].size != size || g_recs[k].modes[4] != (int)pucModeY[(MAX_PU_XY + 2) * dy + dx]) {
            fprintf(stderr, "chroma_trace_hook: CU walk and recorded searches disagree\n");
            abort();
        }
        const Rec &r = g_recs[k];
        const int hdr[8] = { r.size, h.iPoc, best, r.modes[0], r.modes[1], r.modes[2], r.modes[3], r.modes[4] };
        fwrite(hdr, sizeof hdr, 1, g_f);
        fwrite(r.ref[0], 1, 4 * r.size + 1, g_f);
        fwrite(r.ref[1], 1, 4 * r.size + 1, g_f);
        fwrite(r.src[0], 1, r.size * r.size, g_f);
        fwrite(r.src[1], 1, r.size * r.size, g_f);
        fwrite(r.satd[0], 4, 5, g_f);
        fwrite(r.satd[1], 4, 5, g_f);
        if (sizeY == 4) sizeY <<= 1;
        nIdx += sizeY * sizeY / 16;
    }
    if (k != g_recs.size()) {
        fprintf(stderr, "chroma_trace_hook: %zu searches recorded, %zu CUs walked\n", g_recs.size(), k);
        abort();
    }
}
