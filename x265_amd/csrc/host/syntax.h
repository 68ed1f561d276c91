// syntax.h - the CABAC syntax elements of the reference's CU coding, written once for every consumer: a "sink" receives the
// bins in coding order. The rate estimate of the decision loops (xEstimateCmdsBit, src/encode.cpp:162) and the arithmetic
// coder of the bitstream writer (xWriteCabacCmds, src/bitstream.cpp:822) are two sinks of the same element coders, as the
// reference's command lists serve both.
//
// A sink provides   void bin(unsigned value, int ctx);      one context-coded bin            (xEncodeBin_pCmd)
//                   void ep(unsigned value, int count);     `count` bypass bins, msb first   (xEncodeBinsEP_pCmd)
// One call of ep() stands for one command of the reference: a command word keeps 16 bits of the value
// (FLAG_OFF_VAL, inc/x265.h:412), which the writing sink reproduces.
#pragma once
#include <cstring>
#include <cstdint>
#include <cstdlib>

namespace xmi {
namespace syntax {

// context layout (inc/x265.h:286-311)
enum {
    kCtxSplit = 0, kCtxSkip = 3, kCtxMergeFlag = 6, kCtxMergeIdx = 7, kCtxPartSize = 8, kCtxPredMode = 13, kCtxIntraPred = 14,
    kCtxChromaPred = 15, kCtxMvd = 22, kCtxQtCbf = 29, kCtxQtRootCbf = 39, kCtxSigCg = 40, kCtxSig = 44, kCtxLastX = 86,
    kCtxLastY = 116, kCtxOne = 146, kCtxAbs = 170, kCtxMvpIdx = 176, kCtxTransSubdiv = 178, kCtxSaoMerge = 182, kCtxSaoType = 183,
    kNumQtCbf = 5, kNumSigCg = 2, kNumSigLuma = 27, kNumLast = 15, kNumOneLuma = 16, kNumAbsLuma = 4, kNumContexts = 184
};
enum { kScanDiag = 0, kScanHor = 1, kScanVer = 2 };

// Coefficient scans (positions y * size + x) of a 4..32 block and the scan of its 4x4 groups (positions gy * side + gx):
// up-right diagonal, horizontal, vertical (src/table.cpp:268-511). log2_size 2..5.
const uint16_t *coef_scan(int scan_idx, int log2_size);
const uint8_t *group_scan(int scan_idx, int log2_size);

// getCoefScanIdx (src/encode.cpp:2120): mode-dependent scans of intra 4x4 / 8x8 luma and 4x4 chroma blocks
inline int coef_scan_idx(int size, bool intra, bool luma, int mode)
{
    if (!intra) return kScanDiag;
    const bool multiple = luma ? (size == 4 || size == 8) : size == 4;      // (a 2x2 chroma block never occurs)
    if (!multiple) return kScanDiag;
    return std::abs(mode - 26) < 5 ? kScanHor : std::abs(mode - 10) < 5 ? kScanVer : kScanDiag;
}

inline int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

// codeLastSignificantXY (src/encode.cpp:2247)
template <class S>
void code_last_xy(S &s, int x, int y, int log2, bool luma, int scan_idx)
{
    static const uint8_t group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
    static const uint8_t min_in_group[10] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 };
    if (scan_idx == kScanVer) {
        const int t = x;
        x = y, y = t;
    }
    const int size = 1 << log2;
    const int cx = kCtxLastX + (luma ? 0 : kNumLast), cy = kCtxLastY + (luma ? 0 : kNumLast);
    const int gx = group_idx[x], gy = group_idx[y], gmax = group_idx[size - 1];
    const int off = luma ? (log2 - 2) * 3 + ((log2 - 1) >> 2) : 0, sh = luma ? (log2 + 1) >> 2 : log2 - 2;
    for (int c = 0; c < gx; c++) s.bin(1, cx + off + (c >> sh));
    if (gx < gmax) s.bin(0, cx + off + (gx >> sh));
    for (int c = 0; c < gy; c++) s.bin(1, cy + off + (c >> sh));
    if (gy < gmax) s.bin(0, cy + off + (gy >> sh));
    if (gx > 3) s.ep((unsigned)(x - min_in_group[gx]), (gx - 2) >> 1);
    if (gy > 3) s.ep((unsigned)(y - min_in_group[gy]), (gy - 2) >> 1);
}

// getSigCtxInc (src/encode.cpp:2187)
inline int sig_ctx(int pattern, int scan_idx, int x, int y, int log2, bool luma)
{
    static const uint8_t map4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
    if (x + y == 0) return 0;
    if (log2 == 2) return map4[4 * y + x];
    const int off = log2 == 3 ? (scan_idx == kScanDiag ? 9 : 15) : (luma ? 21 : 12);
    const int xs = x & 3, ys = y & 3;
    int cnt;
    if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
    else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
    else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
    else cnt = 2;
    return ((luma && ((x >> 2) + (y >> 2)) > 0) ? 3 : 0) + off + cnt;
}

// xWriteCoefRemainExGolomb (src/encode.cpp:2329)
template <class S>
void code_remaining(S &s, unsigned v, unsigned rice)
{
    if (v < (3u << rice)) {
        const unsigned len = v >> rice;
        s.ep((1u << (len + 1)) - 2, (int)len + 1);
        if (rice) s.ep(v & ((1u << rice) - 1), (int)rice);
    } else {
        unsigned len = rice, code = v - (3u << rice);
        while (code >= (1u << len)) {
            code -= 1u << len;
            len++;
        }
        s.ep((1u << (3 + len + 1 - rice)) - 2, (int)(3 + len + 1 - rice));
        if (len) s.ep(code, (int)len);
    }
}

// xWriteEpExGolomb (src/encode.cpp:2298): one command
template <class S>
void code_ep_ex_golomb(S &s, unsigned v, unsigned k)
{
    unsigned bins = 0;
    int n = 0;
    while (v >= (1u << k)) {
        bins = 2 * bins + 1;
        n++;
        v -= 1u << k;
        k++;
    }
    bins = 2 * bins;
    n++;
    bins = (bins << k) | v;
    n += (int)k;
    s.ep(bins, n);
}

// xEncodeCoeffNxN (src/encode.cpp:2385-2553): the residual syntax of one size x size transform block whose levels lie
// `stride` apart; the block holds at least one non-zero level.
template <class S>
void code_coefficients(S &s, const int16_t *coef, int stride, int size, bool luma, int scan_idx)
{
    const int log2 = ilog2(size), side = size >> 2;
    const uint16_t *scan = coef_scan(scan_idx, log2);
    const uint8_t *scan_cg = group_scan(scan_idx, log2);
    // which 4x4 groups hold a coefficient (four 8-byte looks per group), the last of them in scan order, and in it the last
    // coefficient: positions 16 set .. 16 set + 15 of every scan are the samples of group scan_cg[set]
    uint8_t cg_flag[64] = {};
    int top_set = -1;
    for (int set = side * side - 1; set >= 0; set--) {
        const int g = scan_cg[set], gy = g / side, gx = g - gy * side;
        const int16_t *c = coef + (gy * 4) * stride + gx * 4;
        uint64_t any = 0;
        for (int r = 0; r < 4; r++) {
            uint64_t v;
            std::memcpy(&v, c + r * stride, 8);
            any |= v;
        }
        if (any) {
            cg_flag[g] = 1;
            if (top_set < 0) top_set = set;
        }
    }
    if (top_set < 0) top_set = 0;                       // (the callers code blocks with a coefficient only)
    int last = top_set * 16 + 15, lx = 0, ly = 0;
    for (;; last--) {
        const int p = scan[last];
        ly = p >> log2, lx = p - (ly << log2);
        if (coef[ly * stride + lx] || last == 0) break;
    }
    code_last_xy(s, lx, ly, log2, luma, scan_idx);

    const int base_cg = kCtxSigCg + (luma ? 0 : kNumSigCg), base_sig = kCtxSig + (luma ? 0 : kNumSigLuma);
    const int last_set = last >> 4;
    unsigned c1 = 1;
    int pos = last;
    for (int set = last_set; set >= 0; set--) {
        const int first = set << 4;
        int16_t mag[16];
        unsigned signs = 0;
        int n = 0;
        unsigned rice = 0;
        if (pos == last) {
            const int v = coef[ly * stride + lx];
            mag[0] = (int16_t)std::abs(v);
            signs = v < 0;
            n = 1;
            pos--;
        }
        const int g = scan_cg[set], gy = g / side, gx = g - gy * side;
        const bool right = gx < side - 1 && cg_flag[g + 1], below = gy < side - 1 && cg_flag[g + side];
        if (set == last_set || set == 0) cg_flag[g] = 1;
        else s.bin(cg_flag[g] != 0, base_cg + (right || below));
        if (cg_flag[g]) {
            const int pattern = size == 4 ? -1 : (int)right + 2 * (int)below;
            for (; pos >= first; pos--) {
                const int p = scan[pos], y = p >> log2, x = p - (y << log2);
                const int v = coef[y * stride + x];
                if (pos != first || set == 0 || n) s.bin(v != 0, base_sig + sig_ctx(pattern, scan_idx, x, y, log2, luma));
                if (v) {
                    mag[n++] = (int16_t)std::abs(v);
                    signs = (signs << 1) + (v < 0);
                }
            }
        } else {
            pos = first - 1;
        }
        if (n > 0) {
            unsigned ctx_set = (set > 0 && luma) ? 2 : 0;
            if (c1 == 0) ctx_set++;
            c1 = 1;
            int base = kCtxOne + 4 * (int)ctx_set + (luma ? 0 : kNumOneLuma);
            const int n1 = n < 8 ? n : 8;
            int first_g1 = 16;
            for (int i = 0; i < n1; i++) {
                const unsigned g1 = mag[i] > 1;
                s.bin(g1, base + (int)c1);
                if (g1) {
                    c1 = 0;
                    if (first_g1 > i) first_g1 = i;
                } else if (c1 != 0) {
                    c1 = c1 + 1 < 3 ? c1 + 1 : 3;
                }
            }
            if (c1 == 0) {
                base = kCtxAbs + (int)ctx_set + (luma ? 0 : kNumAbsLuma);
                if (first_g1 != 16) s.bin(mag[first_g1] > 2, base);
            }
            s.ep(signs, n);
            int first2 = 1;
            if (c1 == 0 || n > 8) {
                for (int i = 0; i < n; i++) {
                    const int base_level = i < 8 ? 2 + first2 : 1;
                    if (mag[i] >= base_level) {
                        code_remaining(s, (unsigned)(mag[i] - base_level), rice);
                        if (mag[i] > 3 * (1 << rice)) rice = rice + 1 < 4 ? rice + 1 : 4;
                    }
                    if (mag[i] >= 2) first2 = 0;
                }
            }
        }
    }
}

// xWriteMVD (src/encode.cpp:4482)
template <class S>
void code_mvd(S &s, int dx, int dy)
{
    const unsigned ax = (unsigned)std::abs(dx), ay = (unsigned)std::abs(dy);
    s.bin(ax != 0, kCtxMvd);
    s.bin(ay != 0, kCtxMvd);
    if (ax) s.bin(ax > 1, kCtxMvd + 1);
    if (ay) s.bin(ay > 1, kCtxMvd + 1);
    if (ax) {
        if (ax > 1) code_ep_ex_golomb(s, ax - 2, 1);
        s.ep(dx < 0, 1);
    }
    if (ay) {
        if (ay > 1) code_ep_ex_golomb(s, ay - 2, 1);
        s.ep(dy < 0, 1);
    }
}

// xWriteMergeIdx (src/encode.cpp:4525)
template <class S>
void code_merge_idx(S &s, unsigned idx)
{
    s.bin(idx != 0, kCtxMergeIdx);
    if (idx) s.ep((1u << idx) - 2, (int)idx);
}

// xWriteIntraDirLumaAngGroup (src/encode.cpp:4438): the prev_intra_luma_pred flags of the group first, then the indices
template <class S>
void code_intra_dir_luma(S &s, int n, const unsigned pred_idx[4])
{
    for (int i = 0; i < n; i++) s.bin(pred_idx[i] < 3, kCtxIntraPred);
    for (int i = 0; i < n; i++) {
        const unsigned v = pred_idx[i];
        if (v < 3) s.ep(v + (v != 0), 1 + (v != 0));
        else s.ep(v - 3, 5);
    }
}

// xWriteIntraDirModeChroma (src/encode.cpp:4466): mode index 4 = derived from luma
template <class S>
void code_intra_dir_chroma(S &s, unsigned mode_idx)
{
    s.bin(mode_idx != 4, kCtxChromaPred);
    if (mode_idx != 4) s.ep(mode_idx, 2);
}

// SAO parameters of one CTU and component (SaoLcuParam, inc/x265.h:337)
struct SaoUnit {
    int merge_left = 0, merge_up = 0;
    int type_idx = -1;          // -1 off, 0..3 edge classes, 4 band offset
    int sub_type = 0;           // band position, or the edge class once the unit has been coded
    int length = 0;
    int offset[4] = { 0, 0, 0, 0 };
};

// codeSaoMaxUvlc (src/encode.cpp:1944)
template <class S>
void code_sao_max_uvlc(S &s, unsigned code, unsigned max_symbol)
{
    if (max_symbol == 0) return;
    const unsigned code_last = max_symbol > code;
    s.ep(code != 0, 1);
    if (code) s.ep(((1u << (code - 1)) - 1) << code_last, (int)(code - 1 + code_last));
}

// encodeSaoOffset (src/encode.cpp:1985); like the reference it notes the edge class in sub_type of a luma / Cb unit
template <class S>
void code_sao_offset(S &s, SaoUnit &u, int comp)
{
    const unsigned symbol = (unsigned)(u.type_idx + 1);
    if (comp != 2) {
        s.bin(symbol != 0, kCtxSaoType);
        if (symbol) s.ep(symbol <= 4 ? 1 : 0, 1);
    }
    if (!symbol) return;
    if (u.type_idx < 4 && comp != 2) u.sub_type = u.type_idx;
    if (u.type_idx == 4) {
        for (int i = 0; i < u.length; i++) code_sao_max_uvlc(s, (unsigned)std::abs(u.offset[i]), 7);
        for (int i = 0; i < u.length; i++)
            if (u.offset[i] != 0) s.ep(u.offset[i] < 0, 1);
        s.ep((unsigned)u.sub_type, 5);
    } else if (u.type_idx < 4) {
        code_sao_max_uvlc(s, (unsigned)u.offset[0], 7);
        code_sao_max_uvlc(s, (unsigned)u.offset[1], 7);
        code_sao_max_uvlc(s, (unsigned)-u.offset[2], 7);
        code_sao_max_uvlc(s, (unsigned)-u.offset[3], 7);
        if (comp != 2) s.ep((unsigned)u.sub_type, 2);
    }
}

}  // namespace syntax
}  // namespace xmi
