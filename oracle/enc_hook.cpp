// enc_hook.cpp - integration harness (TEST INFRASTRUCTURE): the one object a maintainer adds to the
// reference encoder's link so that its unmodified CTU loop calls other pixel primitives.
//
// It is linked (oracle/Makefile, target enc) with objects compiled from the reference's OWN sources where they
// lie. The reference's 7 NASM-only symbols cannot be assembled in this image; they are bound to address 0 at
// link time and the table entries that point at them are overwritten here before main() runs. Three builds:
//   HOOK_MODE 0  "refsse" : only the NASM-backed entries (xSatdN[0..2], xSsdN[0..4]) come from the CPU oracle;
//                           every other primitive is the reference's own compiled C++/SSE4 code.
//   HOOK_MODE 1  "oracle" : every table entry and the four free functions come from the CPU oracle.
//   HOOK_MODE 2  "gpu"    : every table entry and the four free functions come from libx265_amd (GPU).
// None of these is "the reference": they share the reference's control flow, and equal bitstreams show
// that the primitives are interchangeable in situ.
#include <cstdint>
#include <cstdio>

typedef unsigned char u8;
typedef unsigned int u32;
typedef short i16;

// the reference's tables (src/pixel.cpp:529, 711, 810, 833, 846); C++ variables are not name-mangled
extern void *xSadN[5], *xSad_AMP[10], *xSatdN[5], *xSsdN[5], *xSatd_AMP[10];

#if HOOK_MODE == 2
#include "../include/x265_amd.h"
extern "C" void xmi_install_primitives(void **, void **, void **, void **, void **);
#else
#include "x265_oracle.h"
template <int N> static u32 oSad(const u8 *a, u32 sa, const u8 *b, u32 sb) { return orc_sad(N, a, (int)sa, b, (int)sb); }
template <int I> static int oSatdTab(u8 *a, intptr_t sa, u8 *b, intptr_t sb) { return (int)orc_satd_table(I, a, (int)sa, b, (int)sb); }
template <int N> static int oSsd(u8 *a, intptr_t sa, u8 *b, intptr_t sb) { return (int)orc_ssd(N, a, (int)sa, b, (int)sb); }
template <int N> static u32 oHad(const u8 *a, u32 sa, const u8 *b, u32 sb) { return orc_satd_hm(N, a, (int)sa, b, (int)sb); }
#endif

static struct Hook {
    Hook()
    {
#if HOOK_MODE == 2
        if (xmi_init(0) != XMI_OK) {
            fprintf(stderr, "enc_hook: %s\n", xmi_last_error());
            __builtin_trap();
        }
        xmi_install_primitives(xSadN, xSad_AMP, xSatdN, xSsdN, xSatd_AMP);
#else
        xSatdN[0] = (void *)oSatdTab<0>;
        xSatdN[1] = (void *)oSatdTab<1>;
        xSatdN[2] = (void *)oSatdTab<2>;
        xSsdN[0] = (void *)oSsd<4>;
        xSsdN[1] = (void *)oSsd<8>;
        xSsdN[2] = (void *)oSsd<16>;
        xSsdN[3] = (void *)oSsd<32>;
        xSsdN[4] = (void *)oSsd<64>;   // the reference's 64x64 wrapper calls the 32x32 NASM symbol
#if HOOK_MODE == 1
        xSatdN[3] = (void *)oSatdTab<3>;
        xSadN[0] = (void *)oSad<4>;
        xSadN[1] = (void *)oSad<8>;
        xSadN[2] = (void *)oSad<16>;
        xSadN[3] = (void *)oSad<32>;
        xSadN[4] = (void *)oSad<64>;
        xSad_AMP[0] = (void *)oSad<8>;
        xSad_AMP[3] = (void *)oSad<16>;
        xSad_AMP[6] = (void *)oSad<32>;
        xSad_AMP[9] = (void *)oSad<64>;
        xSatd_AMP[0] = (void *)oHad<8>;
        xSatd_AMP[3] = (void *)oHad<16>;
        xSatd_AMP[6] = (void *)oHad<32>;
        xSatd_AMP[9] = (void *)oHad<64>;
#endif
#endif
    }
} g_hook;

#if HOOK_MODE != 0
// GNU ld --wrap targets: calls made by the reference's encode.o to xSubDct / xQuant / xDeQuant / xIDctAdd land here
extern "C" {
static void die(const char *what)
{
#if HOOK_MODE == 2
    fprintf(stderr, "enc_hook: %s failed: %s\n", what, xmi_last_error());
#else
    fprintf(stderr, "enc_hook: %s failed\n", what);
#endif
    __builtin_trap();
}
void __wrap__Z7xSubDctPsPhjS0_jS_S_ij(i16 *dst, u8 *src, u32 stride, u8 *pred, u32 spred, i16 *, i16 *, int w, u32 mode)
{
#if HOOK_MODE == 2
    if (xmi_sub_dct(dst, src, stride, pred, spred, w, mode) < 0) die("xSubDct");
#else
    orc_sub_dct(dst, src, (int)stride, pred, (int)spred, w, mode);
#endif
}
u32 __wrap__Z6xQuantPsS_jjii10xSliceType(i16 *dst, i16 *src, u32 stride, u32 qp, int w, int h, int st)
{
#if HOOK_MODE == 2
    const int64_t r = xmi_quant(dst, src, stride, qp, w, h, st);
    if (r < 0) die("xQuant");
    return (u32)r;
#else
    return orc_quant(dst, src, (int)stride, qp, w, h, st);
#endif
}
void __wrap__Z8xDeQuantPsS_jjii(i16 *dst, i16 *src, u32 stride, u32 qp, int w, int h)
{
#if HOOK_MODE == 2
    if (xmi_dequant(dst, src, stride, qp, w, h) < 0) die("xDeQuant");
#else
    orc_dequant(dst, src, (int)stride, qp, w, h);
#endif
}
void __wrap__Z8xIDctAddPhjPsjS_S0_S0_ij(u8 *dst, u32 sdst, i16 *src, u32 stride, u8 *pred, i16 *, i16 *, int w, u32 mode)
{
#if HOOK_MODE == 2
    if (xmi_idct_add(dst, sdst, src, stride, pred, w, mode) < 0) die("xIDctAdd");
#else
    orc_idct_add(dst, (int)sdst, src, (int)stride, pred, w, mode);
#endif
}
}
#endif
