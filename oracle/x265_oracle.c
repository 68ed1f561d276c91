/*
 * x265_oracle.c - CPU restatement of the x265-Milestone pixel primitives (TEST INFRASTRUCTURE).
 * See x265_oracle.h for scope, citations and parity status. Nothing here is shipped or measured as
 * the product; it is the checker for the HIP kernels and the "port" CPU baseline.
 */
#include "x265_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* floor(log2(x))+1, inc/utils.h:34-47; sizes are passed as xLog2(size-1): 4->2 ... 64->6 */
static int xlog2(unsigned x)
{
    int n = 0;
    while (x) { n++; x >>= 1; }
    return n;
}

/* ------------------------------------------------------------------ SAD / SSD ---- */

uint32_t orc_sad_wh(int w, int h, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    uint32_t s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            s += (uint32_t)abs((int)src[y * ss + x] - (int)ref[y * rs + x]);
    return s;
}

uint32_t orc_sad(int n, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    return orc_sad_wh(n, n, src, ss, ref, rs);
}

uint32_t orc_ssd(int n, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    uint32_t s = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int d = (int)src[y * ss + x] - (int)ref[y * rs + x];
            s += (uint32_t)(d * d);
        }
    return s;
}

/* ------------------------------------------------------------------ SATD ---- */

/* sum |H_n * D * H_n| for one n x n tile (n = 4 or 8), natural-order Hadamard. The sum of absolute
 * values does not depend on the row/column order of H, so the reference's H4/H8 orderings
 * (src/pixel.cpp:50-61) and a butterfly give the same number. */
static uint32_t hadamard_abs_sum(int n, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    int m[8][8];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            m[y][x] = (int)src[y * ss + x] - (int)ref[y * rs + x];
    for (int y = 0; y < n; y++)
        for (int len = 1; len < n; len <<= 1)
            for (int i = 0; i < n; i += 2 * len)
                for (int j = i; j < i + len; j++) {
                    int a = m[y][j], b = m[y][j + len];
                    m[y][j] = a + b;
                    m[y][j + len] = a - b;
                }
    for (int x = 0; x < n; x++)
        for (int len = 1; len < n; len <<= 1)
            for (int i = 0; i < n; i += 2 * len)
                for (int j = i; j < i + len; j++) {
                    int a = m[j][x], b = m[j + len][x];
                    m[j][x] = a + b;
                    m[j + len][x] = a - b;
                }
    uint32_t s = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            s += (uint32_t)abs(m[y][x]);
    return s;
}

uint32_t orc_satd_hm(int n, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    if (n == 4)
        return (hadamard_abs_sum(4, src, ss, ref, rs) + 1) >> 1;
    uint32_t s = 0;
    for (int y = 0; y < n; y += 8)
        for (int x = 0; x < n; x += 8)
            s += (hadamard_abs_sum(8, src + y * ss + x, ss, ref + y * rs + x, rs) + 2) >> 2;
    return s;
}

uint32_t orc_satd_x264(int n, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    uint32_t s = 0;
    for (int y = 0; y < n; y += 4)
        for (int x = 0; x < n; x += 4)
            s += hadamard_abs_sum(4, src + y * ss + x, ss, ref + y * rs + x, rs);
    return s >> 1;
}

uint32_t orc_satd_table(int idx, const uint8_t *src, int ss, const uint8_t *ref, int rs)
{
    switch (idx) {
    case 0: return orc_satd_x264(4, src, ss, ref, rs);
    case 1: return orc_satd_x264(8, src, ss, ref, rs);
    case 2: return orc_satd_x264(16, src, ss, ref, rs);
    default: return orc_satd_hm(32, src, ss, ref, rs);
    }
}

/* ------------------------------------------------------------------ transforms ---- */

/* HEVC core transform basis. a[m] is the integer the standard assigns to 64*sqrt(2)*cos(m*pi/64);
 * row k of the N-point matrix is T[k][n] = sign * a[.] of the angle k*(2n+1)*(32/N) (mod 128).
 * Values equal xg_aiT4/8/16/32 in src/table.cpp. */
static const int16_t k_cos64[33] = {
    64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0
};

static int dct_coef(int n_pt, int k, int n)
{
    if (k == 0)
        return 64;
    int m = (k * (32 / n_pt) * (2 * n + 1)) & 127;
    if (m <= 32) return k_cos64[m];
    if (m <= 64) return -k_cos64[64 - m];
    if (m <= 96) return -k_cos64[m - 64];
    return k_cos64[128 - m];
}

static const int16_t k_dst4[4][4] = {
    { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 }
};

static void build_matrix(int n, int use_dst, int m[32][32])
{
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            m[k][j] = use_dst ? k_dst4[k][j] : dct_coef(n, k, j);
}

void orc_sub_dct(int16_t *dst, const uint8_t *src, int stride, const uint8_t *pred, int stride_pred,
                 int width, unsigned mode)
{
    const int n = width;
    const int lg = xlog2((unsigned)n - 1);
    const int use_dst = (n == 4) && (mode != ORC_MODE_INVALID);
    int m[32][32];
    static int16_t resid[32 * 32], tmp[32 * 32]; /* not re-entrant: test infrastructure only */
    build_matrix(n, use_dst, m);

    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            resid[i * n + j] = (int16_t)((int)src[i * stride + j] - (int)pred[i * stride_pred + j]);

    int shift = lg - 1, rnd = 1 << (shift - 1);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) {
            int acc = 0;
            for (int j = 0; j < n; j++)
                acc += m[k][j] * resid[i * n + j];
            tmp[k * n + i] = (int16_t)((acc + rnd) >> shift);
        }
    shift = lg + 6;
    rnd = 1 << (shift - 1);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) {
            int acc = 0;
            for (int j = 0; j < n; j++)
                acc += m[k][j] * tmp[i * n + j];
            dst[k * stride + i] = (int16_t)((acc + rnd) >> shift);
        }
}

static const int k_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 }; /* src/table.cpp:150 */
static const int k_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };               /* src/table.cpp:155 */

uint32_t orc_quant(int16_t *dst, const int16_t *src, int stride, unsigned qp, int width, int height,
                   int slice_type)
{
    const int lg = xlog2((unsigned)width - 1);
    const int q = k_quant_scales[qp % 6];
    const int qbits = 14 + (int)(qp / 6) + (15 - 8 - lg);
    const int rnd = (slice_type == ORC_SLICE_I ? 171 : 85) << (qbits - 9);
    uint32_t sum = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int c = src[y * stride + x];
            int level = (abs(c) * q + rnd) >> qbits;
            sum += (uint32_t)level;
            if (c < 0) level = -level;
            dst[y * stride + x] = (int16_t)clip3(-32768, 32767, level);
        }
    return sum;
}

void orc_dequant(int16_t *dst, const int16_t *src, int stride, unsigned qp, int width, int height)
{
    const int lg = xlog2((unsigned)width - 1);
    const int shift = 20 - 14 - (15 - 8 - lg);
    const int rnd = 1 << (shift - 1);
    const int scale = k_inv_quant_scales[qp % 6] << (qp / 6);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int v = (src[y * stride + x] * scale + rnd) >> shift;
            dst[y * stride + x] = (int16_t)clip3(-32768, 32767, v);
        }
}

void orc_idct_add(uint8_t *dst, int stride_dst, const int16_t *src, int stride, const uint8_t *pred,
                  int width, unsigned mode)
{
    const int n = width;
    const int use_dst = (n == 4) && (mode != ORC_MODE_INVALID);
    int m[32][32];
    static int16_t t0[32 * 32], t1[32 * 32];
    build_matrix(n, use_dst, m);

    /* stage 1 over columns of the coefficient block: t0[i][j] = sum_k M[k][j] * src[k][i] */
    int shift = 7, rnd = 1 << (shift - 1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            int acc = 0;
            for (int k = 0; k < n; k++)
                acc += m[k][j] * src[k * stride + i];
            t0[i * n + j] = (int16_t)clip3(-32768, 32767, (acc + rnd) >> shift);
        }
    shift = 12;
    rnd = 1 << (shift - 1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            int acc = 0;
            for (int k = 0; k < n; k++)
                acc += m[k][j] * t0[k * n + i];
            t1[i * n + j] = (int16_t)clip3(-32768, 32767, (acc + rnd) >> shift);
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            dst[i * stride_dst + j] = (uint8_t)clip3(0, 255, t1[i * n + j] + pred[i * stride + j]);
}

uint32_t orc_tu_chain(const uint8_t *pix, const uint8_t *pred, uint8_t *rec, int16_t *coef,
                      int size, int stride, int stride_rec, unsigned qp, unsigned mode, int slice_type)
{
    static int16_t work[64 * 64];
    orc_sub_dct(work, pix, stride, pred, stride, size, mode);
    uint32_t sum = orc_quant(coef, work, stride, qp, size, size, slice_type);
    if (sum) {
        orc_dequant(work, coef, stride, qp, size, size);
        orc_idct_add(rec, stride_rec, work, stride, pred, size, mode);
    } else {
        for (int i = 0; i < size; i++)
            memcpy(rec + i * stride_rec, pred + i * stride, (size_t)size);
    }
    return sum;
}

/* ------------------------------------------------------------------ SAD surface ---- */

void orc_sad_surface(const uint8_t *cur, int cur_stride, const uint8_t *ref, int ref_stride,
                     int width, int height, int range, int pitch,
                     uint16_t *sad8, uint16_t *sad16, uint32_t *sad32, uint32_t *sad64,
                     int ctu_begin, int ctu_end)
{
    const int wc = width / 64;
    const int nmv = 2 * range + 1;
    const size_t plane = (size_t)nmv * (size_t)pitch;
    (void)height;
    for (int ctu = ctu_begin; ctu < ctu_end; ctu++) {
        const int cx = (ctu % wc) * 64, cy = (ctu / wc) * 64;
        for (int dy = -range; dy <= range; dy++)
            for (int dx = -range; dx <= range; dx++) {
                uint32_t s8[64];
                const size_t at = (size_t)(dy + range) * (size_t)pitch + (size_t)(dx + range);
                for (int b = 0; b < 64; b++) {
                    int bx = (b & 7) * 8, by = (b >> 3) * 8;
                    s8[b] = orc_sad(8, cur + (size_t)(cy + by) * cur_stride + cx + bx, cur_stride,
                                    ref + (ptrdiff_t)(cy + by + dy) * ref_stride + cx + bx + dx, ref_stride);
                    sad8[((size_t)ctu * 64 + b) * plane + at] = (uint16_t)s8[b];
                }
                uint32_t s16[16], s32[4] = { 0, 0, 0, 0 }, s64 = 0;
                for (int b = 0; b < 16; b++) {
                    int bx = (b & 3) * 2, by = (b >> 2) * 2;
                    s16[b] = s8[by * 8 + bx] + s8[by * 8 + bx + 1] + s8[(by + 1) * 8 + bx] + s8[(by + 1) * 8 + bx + 1];
                    sad16[((size_t)ctu * 16 + b) * plane + at] = (uint16_t)s16[b];
                    s32[(b >> 3) * 2 + ((b & 3) >> 1)] += s16[b];
                }
                for (int b = 0; b < 4; b++) {
                    sad32[((size_t)ctu * 4 + b) * plane + at] = s32[b];
                    s64 += s32[b];
                }
                sad64[(size_t)ctu * plane + at] = s64;
            }
    }
}

/* ------------------------------------------------------------------ ME driver ---- */

unsigned orc_mv_component_bits(int val)
{
    unsigned length = 1;
    unsigned t = (val <= 0) ? ((unsigned)(-val) << 1) + 1 : ((unsigned)val << 1);
    while (t != 1) {
        t >>= 1;
        length += 2;
    }
    return length;
}

static const int k_luma_filter[4][8] = { /* src/table.cpp:525-530 */
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 }
};

void orc_pred_inter_luma(const uint8_t *ref, uint8_t *dst, int rs, int w, int h, int mvx, int mvy)
{
    const int dx = mvx >> 2, dy = mvy >> 2, xf = mvx & 3, yf = mvy & 3;
    ref += (ptrdiff_t)dy * rs + dx;
    if (!xf && !yf) {
        for (int i = 0; i < h; i++)
            memcpy(dst + i * 64, ref + (ptrdiff_t)i * rs, (size_t)w);
    } else if (!yf) { /* src/pixel.cpp:7705-7734 */
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 8; t++)
                    s += ref[(ptrdiff_t)i * rs + j + t - 3] * k_luma_filter[xf][t];
                dst[i * 64 + j] = (uint8_t)clip3(0, 255, (s + 32) >> 6);
            }
    } else if (!xf) { /* src/pixel.cpp:8137-8166 */
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 8; t++)
                    s += ref[(ptrdiff_t)(i + t - 3) * rs + j] * k_luma_filter[yf][t];
                dst[i * 64 + j] = (uint8_t)clip3(0, 255, (s + 32) >> 6);
            }
    } else { /* src/pixel.cpp:8690-8741 */
        static int16_t tmp[(64 + 7) * 64];
        for (int i = 0; i < h + 7; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 8; t++)
                    s += ref[(ptrdiff_t)(i - 3) * rs + j + t - 3] * k_luma_filter[xf][t];
                tmp[i * 64 + j] = (int16_t)(s - 8192);
            }
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 8; t++)
                    s += tmp[(i + t) * 64 + j] * k_luma_filter[yf][t];
                dst[i * 64 + j] = (uint8_t)clip3(0, 255, (s + 8192 * 64 + 2048) >> 12);
            }
    }
}

static const int k_chroma_filter[8][4] = { /* xg_chromaFilter, src/table.cpp:532-541 */
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 }
};

void orc_pred_inter_chroma(const uint8_t *ref, uint8_t *dst, int rs, int w, int h, int mvx, int mvy)
{
    const int dx = mvx >> 3, dy = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
    ref += (ptrdiff_t)dy * rs + dx;
    if (!xf && !yf) {
        for (int i = 0; i < h; i++)
            memcpy(dst + i * 32, ref + (ptrdiff_t)i * rs, (size_t)w);
    } else if (!yf) { /* src/pixel.cpp:8982-9007 */
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 4; t++)
                    s += ref[(ptrdiff_t)i * rs + j + t - 1] * k_chroma_filter[xf][t];
                dst[i * 32 + j] = (uint8_t)clip3(0, 255, (s + 32) >> 6);
            }
    } else if (!xf) { /* src/pixel.cpp:9257-9282 */
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 4; t++)
                    s += ref[(ptrdiff_t)(i + t - 1) * rs + j] * k_chroma_filter[yf][t];
                dst[i * 32 + j] = (uint8_t)clip3(0, 255, (s + 32) >> 6);
            }
    } else { /* src/pixel.cpp:9611-9654 */
        int16_t tmp[(32 + 3) * 32];
        for (int i = 0; i < h + 3; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 4; t++)
                    s += ref[(ptrdiff_t)(i - 1) * rs + j + t - 1] * k_chroma_filter[xf][t];
                tmp[i * 32 + j] = (int16_t)(s - 8192);
            }
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                int s = 0;
                for (int t = 0; t < 4; t++)
                    s += tmp[(i + t) * 32 + j] * k_chroma_filter[yf][t];
                dst[i * 32 + j] = (uint8_t)clip3(0, 255, (s + 8192 * 64 + 2048) >> 12);
            }
    }
}

void orc_motion_search(const uint8_t *cur, const uint8_t *ref, int rs, int size,
                       const orc_mv mvp[2], int zero_mv_tested, orc_mv mv_min_ipel, orc_mv mv_max_ipel,
                       double lambda, int me_range, orc_me_result *out)
{
    static const int dia4[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
    static const int dia9[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 },
                                    { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
    const uint32_t ulambda = (uint32_t)sqrt(lambda);
    uint8_t tmp[64 * 64];
    int mvp_idx = -1, bmx = 0, bmy = 0, best_k = -1;
    uint32_t bsad = INT_MAX, bcost = INT_MAX;
    int passes = 1;
    out->n_int_sads = 0;
    out->n_sub_sads = 0;

    if ((abs(mvp[0].x - mvp[1].x) >> 2) + (abs(mvp[0].y - mvp[1].y) >> 2) > 4)
        passes = 2;
    for (int i = 0; i < passes; i++) {
        orc_pred_inter_luma(ref, tmp, rs, size, size, mvp[i].x, mvp[i].y);
        uint32_t s = orc_sad(size, cur, 64, tmp, 64);
        out->n_sub_sads++;
        if (s < bsad) {
            bsad = s;
            mvp_idx = i;
        }
    }
    const int px = mvp[mvp_idx].x, py = mvp[mvp_idx].y;
#define MVCOST(X, Y) (ulambda * (orc_mv_component_bits(px - (X)) + orc_mv_component_bits(py - (Y))))
    if (!zero_mv_tested) {
        uint32_t c = orc_sad(size, cur, 64, ref, rs) + MVCOST(0, 0);
        out->n_int_sads++;
        if (c < bcost) {
            bcost = c;
            bmx = bmy = 0;
        }
    }
    int omx = px >> 2, omy = py >> 2;
    {
        uint32_t c = orc_sad(size, cur, 64, ref + (ptrdiff_t)omy * rs + omx, rs) + MVCOST(omx * 4, omy * 4);
        out->n_int_sads++;
        if (c < bcost) {
            bcost = c;
            bmx = omx * 4;
            bmy = omy * 4;
        }
    }
    for (int i = 0; i < me_range; i++) {
        int changed = 0;
        for (int k = 0; k < 4; k++) {
            int tx = clip3(mv_min_ipel.x, mv_max_ipel.x, omx + dia4[k][0]);
            int ty = clip3(mv_min_ipel.y, mv_max_ipel.y, omy + dia4[k][1]);
            uint32_t c = orc_sad(size, cur, 64, ref + (ptrdiff_t)ty * rs + tx, rs) + MVCOST(tx * 4, ty * 4);
            out->n_int_sads++;
            if (c < bcost) {
                changed = 1;
                bcost = c;
                best_k = k;
                bmx = tx * 4;
                bmy = ty * 4;
            }
        }
        if (!changed)
            break;
        omx += dia4[best_k][0];
        omy += dia4[best_k][1];
        if (bcost < 800)
            break;
    }
    {   /* src/encode.cpp:2950-2953: x is clipped against the y maximum (reference quirk, kept) */
        int mnx = 4 * (mv_min_ipel.x - 4), mny = 4 * (mv_min_ipel.y - 4);
        int mxy = 4 * (mv_max_ipel.y - 4);
        bmx = clip3(mnx, mxy, bmx);
        bmy = clip3(mny, mxy, bmy);
    }
    bcost = INT_MAX;
    int step = 2;
    for (int i = 0; i < 2; i++) {
        int changed = 0;
        for (int k = 0; k < 9; k++) {
            int tx = bmx + dia9[k][0] * step, ty = bmy + dia9[k][1] * step;
            tx = (int16_t)tx; /* tmv is an xMV of Int16 */
            ty = (int16_t)ty;
            orc_pred_inter_luma(ref, tmp, rs, size, size, tx, ty);
            uint32_t c = orc_sad(size, cur, 64, tmp, 64) + MVCOST(tx, ty);
            out->n_sub_sads++;
            if (c < bcost) {
                changed = 1;
                bcost = c;
                best_k = k;
            }
        }
        if (changed) {
            bmx += dia9[best_k][0] * step;
            bmy += dia9[best_k][1] * step;
        }
        step >>= 1;
    }
#undef MVCOST
    out->mv.x = (int16_t)bmx;
    out->mv.y = (int16_t)bmy;
    out->best_cost = bcost;
    out->best_sad = bsad;
    out->mvp_idx = mvp_idx;
}

/* ------------------------------------------------------------------ intra mode search ---- */

static const uint8_t k_intra_filter_type[4][35] = { /* src/table.cpp:32-40, sizes 4..32 */
    { 0 },
    { 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 },
    { 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1 },
    { 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1 },
};
static const int8_t k_intra_angle[35] = { /* src/table.cpp:43-50 */
    0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
    -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int16_t k_intra_inv_angle[35] = { /* src/table.cpp:53-60 */
    0, 0, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315,
    256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

void orc_padding_ref(uint8_t *ref, int size, unsigned valids)
{
    const int off[6] = { 0, size, 2 * size, 2 * size + 1, 3 * size + 1, 4 * size + 1 };
    int n = 0;
    while (n < 5 && !(valids & (1u << n))) n++;
    uint8_t pad = ref[off[n]];
    for (int i = 0; i < off[n]; i++) ref[i] = pad;
    for (; n < 5; n++)
        if (!(valids & (1u << n))) {
            pad = ref[off[n] - 1];
            for (int i = off[n]; i < off[n + 1]; i++) ref[i] = pad;
        }
}

void orc_filter_ref(uint8_t *dst, const uint8_t *src, int size, int strong)
{
    const int bl = src[0], tl = src[2 * size], tr = src[4 * size];
    const int bilinear_left = abs(bl + tl - 2 * src[size]) < 8;
    const int bilinear_top = abs(tl + tr - 2 * src[3 * size]) < 8;
    if (strong && size == 32 && bilinear_left && bilinear_top) {
        dst[0] = src[0];
        dst[2 * size] = src[2 * size];
        dst[4 * size] = src[4 * size];
        for (int i = 1; i < 2 * size; i++) {
            dst[i] = (uint8_t)(((2 * size - i) * bl + i * tl + size) >> 6);
            dst[2 * size + i] = (uint8_t)(((2 * size - i) * tl + i * tr + size) >> 6);
        }
    } else {
        dst[0] = src[0];
        dst[4 * size] = src[4 * size];
        for (int i = 1; i < 4 * size; i++) dst[i] = (uint8_t)((src[i - 1] + 2 * src[i] + src[i + 1] + 2) >> 2);
    }
}

void orc_intra_pred(uint8_t *dst, const uint8_t *ref, int stride, int mode, int size, int is_luma)
{
    const uint8_t *left = ref + 2 * size - 1, *top = ref + 2 * size + 1, *tl = ref + 2 * size;
    const int lg = xlog2((unsigned)size - 1);
    if (mode == 0) { /* planar, src/pixel.cpp:6489-6532 */
        const int bl = left[-size], tr = top[size];
        for (int i = 0; i < size; i++)
            for (int j = 0; j < size; j++)
                dst[i * stride + j] = (uint8_t)(((size - 1 - j) * left[-i] + (j + 1) * tr + (size - 1 - i) * top[j] + (i + 1) * bl + size) >> (lg + 1));
    } else if (mode == 1) { /* DC, src/pixel.cpp:6583-6601, 6704-6730 */
        unsigned sum = 0;
        for (int i = 0; i < size; i++) sum += top[i] + left[-i];
        const int dc = (int)((sum + (unsigned)size) / (unsigned)(2 * size));
        for (int i = 0; i < size; i++) memset(dst + i * stride, dc, (size_t)size);
        if (is_luma) {
            dst[0] = (uint8_t)((top[0] + left[0] + 2 * dc + 2) >> 2);
            for (int i = 1; i < size; i++) {
                dst[i] = (uint8_t)((top[i] + 3 * dc + 2) >> 2);
                dst[i * stride] = (uint8_t)((left[-i] + 3 * dc + 2) >> 2);
            }
        }
    } else { /* angular, src/pixel.cpp:7243-7331 */
        const int hor = mode < 18, angle = k_intra_angle[mode], inv = k_intra_inv_angle[mode];
        uint8_t buf[2 * 64 + 1], *main_ = buf + (angle < 0 ? 64 : 0);
        uint8_t tmp[32 * 32];
        for (int x = 0; x < size + 1; x++) main_[x] = hor ? tl[-x] : tl[x];
        if (angle < 0) {
            int sum = 128;
            for (int x = -1; x > ((size * angle) >> 5); x--) {
                sum += inv;
                main_[x] = tl[hor ? (sum >> 8) : -(sum >> 8)];
            }
        } else {
            for (int x = size + 1; x < 2 * size + 1; x++) main_[x] = hor ? tl[-x] : tl[x];
        }
        int delta = 0;
        for (int k = 0; k < size; k++) {
            delta += angle;
            const int idx = delta >> 5, fact = delta & 31;
            for (int x = 0; x < size; x++)
                tmp[k * size + x] = fact ? (uint8_t)(((32 - fact) * main_[x + idx + 1] + fact * main_[x + idx + 2] + 16) >> 5)
                                         : main_[x + idx + 1];
        }
        if (is_luma && angle == 0) {
            const int o = hor ? 1 : -1;
            for (int x = 0; x < size; x++)
                tmp[x * size] = (uint8_t)clip3(0, 255, tmp[x * size] + ((tl[(x + 1) * o] - tl[0]) >> 1));
        }
        for (int k = 0; k < size; k++)
            for (int x = 0; x < size; x++)
                dst[k * stride + x] = hor ? tmp[x * size + k] : tmp[k * size + x];
    }
}

void orc_intra_search(const uint8_t *src, int src_stride, const uint8_t *ref_raw, int size, unsigned valids,
                      int strong_smoothing, const uint8_t mpm_in[3], double sqrt_lambda, orc_intra_result *out)
{
    uint8_t ref[2][4 * 64 + 1], pred[32 * 32];
    const int lg = xlog2((unsigned)size - 1);
    if (valids == 0) {
        memset(ref, 0x80, sizeof ref);   /* src/encode.cpp:4006-4008: both sample sets */
    } else {
        memcpy(ref[0], ref_raw, (size_t)(4 * size + 1));
        orc_padding_ref(ref[0], size, valids);
        orc_filter_ref(ref[1], ref[0], size, strong_smoothing);
    }
    double best = 1.7976931348623157e308;
    int best_mode = 0;
    for (int mode = 0; mode < 35; mode++) {
        orc_intra_pred(pred, ref[k_intra_filter_type[lg - 2][mode]], size, mode, size, size != 32);
        double cost = (mode == mpm_in[0] ? 1.0 : (mode == mpm_in[1] || mode == mpm_in[2]) ? 2.0 : 3.0) * sqrt_lambda;
        const uint32_t satd = orc_satd_table(lg - 2, src, src_stride, pred, size);
        out->satd[mode] = satd;
        cost += (double)satd;
        if (cost < best) {
            best = cost;
            best_mode = mode;
        }
    }
    uint8_t mpm[3] = { mpm_in[0], mpm_in[1], mpm_in[2] };
    int idx;
    if (best_mode == mpm[0]) idx = 0;
    else if (best_mode == mpm[1]) idx = 1;
    else if (best_mode == mpm[2]) idx = 2;
    else { /* src/encode.cpp:4059-4080 */
        uint8_t t;
        if (mpm[0] > mpm[1]) { t = mpm[0]; mpm[0] = mpm[1]; mpm[1] = t; }
        if (mpm[0] > mpm[2]) { t = mpm[0]; mpm[0] = mpm[2]; mpm[2] = t; }
        if (mpm[1] > mpm[2]) { t = mpm[1]; mpm[1] = mpm[2]; mpm[2] = t; }
        idx = best_mode;
        idx -= idx > mpm[2];
        idx -= idx > mpm[1];
        idx -= idx > mpm[0];
        idx += 3;
    }
    out->best_mode = best_mode;
    out->pred_idx = idx;
    out->best_cost = best;
}

/* ------------------------------------------------------------------ deblocking filter ---- */

static const uint8_t k_tc_table[54] = { /* src/Deblock.cpp:195-200 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static const uint8_t k_beta_table[52] = { /* src/Deblock.cpp:202-207 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30,
    32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };

static uint8_t clip_pixel(int a) { return (uint8_t)(a < 0 ? 0 : a > 255 ? 255 : a); }

/* src/Deblock.cpp:61-145: one 4-line segment of a luma edge. xs = step across the edge, ys = step along it. */
static void deblock_luma_segment(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int beta, int tc)
{
#define PX(i, line) ((int)pix[(i) * xs + (line) * ys])
    const int dp0 = abs(PX(-3, 0) - 2 * PX(-2, 0) + PX(-1, 0)), dq0 = abs(PX(2, 0) - 2 * PX(1, 0) + PX(0, 0));
    const int dp3 = abs(PX(-3, 3) - 2 * PX(-2, 3) + PX(-1, 3)), dq3 = abs(PX(2, 3) - 2 * PX(1, 3) + PX(0, 3));
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (abs(PX(-4, 0) - PX(-1, 0)) + abs(PX(3, 0) - PX(0, 0)) < beta_3 && abs(PX(-1, 0) - PX(0, 0)) < tc25 &&
        abs(PX(-4, 3) - PX(-1, 3)) + abs(PX(3, 3) - PX(0, 3)) < beta_3 && abs(PX(-1, 3) - PX(0, 3)) < tc25 &&
        (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
        const int tc2 = tc << 1;
        for (int d = 0; d < 4; d++, pix += ys) {
            const int p3 = PX(-4, 0), p2 = PX(-3, 0), p1 = PX(-2, 0), p0 = PX(-1, 0), q0 = PX(0, 0), q1 = PX(1, 0), q2 = PX(2, 0), q3 = PX(3, 0);
            pix[-1 * xs] = (uint8_t)clip3(p0 - tc2, p0 + tc2, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * xs] = (uint8_t)clip3(p1 - tc2, p1 + tc2, (p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * xs] = (uint8_t)clip3(p2 - tc2, p2 + tc2, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            pix[0] = (uint8_t)clip3(q0 - tc2, q0 + tc2, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[xs] = (uint8_t)clip3(q1 - tc2, q1 + tc2, (p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * xs] = (uint8_t)clip3(q2 - tc2, q2 + tc2, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        }
    } else {
        const int tc_2 = tc >> 1, side = (beta + (beta >> 1)) >> 3;
        const int nd_p = dp0 + dp3 < side ? 2 : 1, nd_q = dq0 + dq3 < side ? 2 : 1;
        for (int d = 0; d < 4; d++, pix += ys) {
            const int p2 = PX(-3, 0), p1 = PX(-2, 0), p0 = PX(-1, 0), q0 = PX(0, 0), q1 = PX(1, 0), q2 = PX(2, 0);
            int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta0) >= 10 * tc) continue;
            delta0 = clip3(-tc, tc, delta0);
            pix[-1 * xs] = clip_pixel(p0 + delta0);
            pix[0] = clip_pixel(q0 - delta0);
            if (nd_p > 1) pix[-2 * xs] = clip_pixel(p1 + clip3(-tc_2, tc_2, (((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1));
            if (nd_q > 1) pix[xs] = clip_pixel(q1 + clip3(-tc_2, tc_2, (((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1));
        }
    }
#undef PX
}

/* src/Deblock.cpp:147-168 */
static void deblock_chroma_segment(uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int tc)
{
    for (int d = 0; d < 4; d++, pix += ys) {
        const int p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
        const int delta0 = clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
        pix[-xs] = clip_pixel(p0 + delta0);
        pix[0] = clip_pixel(q0 - delta0);
    }
}

/* src/Deblock.cpp:209-234 (qp_bd_offset = 0) */
static int deblock_chroma_tc(int qp_y, int offset, int tc_offset)
{
    static const int qp_c[] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
    const int qp_i = clip3(0, 57, qp_y + offset);
    const int qp = qp_i < 30 ? qp_i : qp_i > 43 ? qp_i - 6 : qp_c[qp_i - 30];
    return k_tc_table[clip3(0, 53, qp + 2 + tc_offset)];
}

void orc_deblock_frame(uint8_t *y, uint8_t *u, uint8_t *v, const orc_deblock_params *p, const uint8_t *vertical_bs,
                       const uint8_t *horizontal_bs, const uint8_t *qp_tab)
{
    const int W = p->width, H = p->height, bsw = W >> 3, qw = W >> 2;   /* QP table: log2_min_coding_block_size = 2 (src/encode.cpp:5060) */
    /* CTB by CTB in raster order, vertical edges of the CTB, then its horizontal edges shifted 8 samples to the left
     * (ff_hevc_deblocking_filter_CTB, src/Deblock.cpp:244-342). xDeblockFrame (:524-559) visits every CTB once when the
     * picture has at least two CTB rows and columns: hls_filters (:512-522) filters CTB (x - 1, y - 1) when called for
     * CTB (x, y) with x, y > 0, the last column and row from their neighbours, and the last CTB after the loop. With one
     * CTB row or one CTB column hls_filters never does anything, and only the last CTB is filtered. */
    const int cols = (W + 63) / 64, rows = (H + 63) / 64, only_last = (rows == 1) != (cols == 1);
    for (int y0 = 0; y0 < H; y0 += 64)
        for (int x0 = 0; x0 < W; x0 += 64) {
            if (only_last && !(x0 == (cols - 1) * 64 && y0 == (rows - 1) * 64)) continue;
            int x_end = x0 + 64 > W ? W : x0 + 64;
            const int y_end = y0 + 64 > H ? H : y0 + 64;
            for (int x = x0 ? x0 : 8; x < x_end; x += 8)
                for (int yy = y0; yy < y_end; yy += 4) {
                    const int bs = vertical_bs[(x >> 3) + (yy >> 2) * bsw];
                    if (!bs) continue;
                    const int qp = (qp_tab[((x - 1) >> 2) + (yy >> 2) * qw] + qp_tab[(x >> 2) + (yy >> 2) * qw] + 1) >> 1;
                    const int beta = k_beta_table[clip3(0, 51, qp + ((p->beta_offset >> 1) << 1))];
                    const int tc = k_tc_table[clip3(0, 53, qp + 2 * (bs - 1) + ((p->tc_offset >> 1) << 1))];
                    deblock_luma_segment(y + (ptrdiff_t)yy * p->stride_y + x, 1, p->stride_y, beta, tc);
                    if ((x & 15) == 0 && (yy & 7) == 0 && bs == 2) {
                        deblock_chroma_segment(u + (ptrdiff_t)(yy / 2) * p->stride_c + x / 2, 1, p->stride_c, deblock_chroma_tc(qp, p->cb_qp_offset, p->tc_offset));
                        deblock_chroma_segment(v + (ptrdiff_t)(yy / 2) * p->stride_c + x / 2, 1, p->stride_c, deblock_chroma_tc(qp, p->cr_qp_offset, p->tc_offset));
                    }
                }
            if (x_end != W) x_end -= 8;
            for (int yy = y0 ? y0 : 8; yy < y_end; yy += 8)
                for (int x = x0 ? x0 - 8 : 0; x < x_end; x += 4) {
                    const int bs = horizontal_bs[(x + yy * bsw) >> 2];
                    if (!bs) continue;
                    const int qp = (qp_tab[(x >> 2) + ((yy - 1) >> 2) * qw] + qp_tab[(x >> 2) + (yy >> 2) * qw] + 1) >> 1;
                    const int beta = k_beta_table[clip3(0, 51, qp + ((p->beta_offset >> 1) << 1))];
                    const int tc = k_tc_table[clip3(0, 53, qp + 2 * (bs - 1) + ((p->tc_offset >> 1) << 1))];
                    deblock_luma_segment(y + (ptrdiff_t)yy * p->stride_y + x, p->stride_y, 1, beta, tc);
                    if ((x & 7) == 0 && (yy & 15) == 0 && bs == 2) {
                        deblock_chroma_segment(u + (ptrdiff_t)(yy / 2) * p->stride_c + x / 2, p->stride_c, 1, deblock_chroma_tc(qp, p->cb_qp_offset, p->tc_offset));
                        deblock_chroma_segment(v + (ptrdiff_t)(yy / 2) * p->stride_c + x / 2, p->stride_c, 1, deblock_chroma_tc(qp, p->cr_qp_offset, p->tc_offset));
                    }
                }
        }
}

/* ------------------------------------------------------------------ SAO statistics ---- */

static int sao_sign(int x) { return (x > 0) - (x < 0); }

void orc_sao_stats_ctu(const uint8_t *rec, int rec_stride, const uint8_t *org, int org_stride, int pic_w, int pic_h,
                       int ctu_x, int ctu_y, int ctu_size, int is_chroma, int64_t count[5][33], int64_t offset_org[5][33])
{
    static const int eo_table[5] = { 1, 2, 0, 3, 4 };      /* g_eoTable, src/encode.cpp:274 */
    const int skip = is_chroma ? 2 : 4, skip_right = is_chroma ? 3 : 5;
    const int lpelx = ctu_x * ctu_size, tpely = ctu_y * ctu_size;
    const int rpelx = lpelx + ctu_size > pic_w ? pic_w : lpelx + ctu_size, bpely = tpely + ctu_size > pic_h ? pic_h : tpely + ctu_size;
    const int lw = rpelx - lpelx, lh = bpely - tpely;
    const int at_right = rpelx == pic_w, at_bottom = bpely == pic_h;
    rec += (ptrdiff_t)tpely * rec_stride + lpelx;
    org += (ptrdiff_t)tpely * org_stride + lpelx;
    memset(count, 0, sizeof(int64_t) * 5 * 33);
    memset(offset_org, 0, sizeof(int64_t) * 5 * 33);
#define REC(x, y) ((int)rec[(ptrdiff_t)(y) * rec_stride + (x)])
#define ADD(type, cls, x, y) (offset_org[type][cls] += (int)org[(ptrdiff_t)(y) * org_stride + (x)] - REC(x, y), count[type][cls]++)
    /* band offset (src/encode.cpp:415-438): class = 1 + (sample >> 3) */
    for (int y = 0; y < (at_bottom ? lh : lh - skip); y++)
        for (int x = 0; x < (at_right ? lw : lw - skip_right); x++) ADD(4, 1 + (REC(x, y) >> 3), x, y);
    const int sx = lpelx == 0 ? 1 : 0, sy = tpely == 0 ? 1 : 0;
    const int ex_in = at_right ? lw - 1 : lw - skip_right;           /* EO_0, EO_2, EO_3 */
    const int ex_v = at_right ? lw : lw - skip_right;                /* EO_1 */
    const int ey = at_bottom ? lh - 1 : lh - skip;
    /* EO_0, horizontal neighbours (:456-480): rows 0 .. lh - skip even in the last CTU row */
    for (int y = 0; y < lh - skip; y++)
        for (int x = sx; x < ex_in; x++)
            ADD(0, eo_table[sao_sign(REC(x, y) - REC(x - 1, y)) + sao_sign(REC(x, y) - REC(x + 1, y)) + 2], x, y);
    /* EO_1, vertical neighbours (:489-524) */
    for (int y = sy; y < ey; y++)
        for (int x = 0; x < ex_v; x++)
            ADD(1, eo_table[sao_sign(REC(x, y) - REC(x, y - 1)) + sao_sign(REC(x, y) - REC(x, y + 1)) + 2], x, y);
    /* EO_2, up-left / down-right (:532-574) */
    for (int y = sy; y < ey; y++)
        for (int x = sx; x < ex_in; x++)
            ADD(2, eo_table[sao_sign(REC(x, y) - REC(x - 1, y - 1)) + sao_sign(REC(x, y) - REC(x + 1, y + 1)) + 2], x, y);
    /* EO_3, up-right / down-left (:582-620) */
    for (int y = sy; y < ey; y++)
        for (int x = sx; x < ex_in; x++)
            ADD(3, eo_table[sao_sign(REC(x, y) - REC(x + 1, y - 1)) + sao_sign(REC(x, y) - REC(x - 1, y + 1)) + 2], x, y);
#undef ADD
#undef REC
}

/* ------------------------------------------------------------------ SAO: applying the offsets ---- */

void orc_sao_apply_plane(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width, int height, int ctu_size,
                         const orc_sao_ctu_param *params)
{
    static const int eo_table[5] = { 1, 2, 0, 3, 4 };      /* g_eoTable, src/encode.cpp:274 */
    static const int nb[4][2][2] = { { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } }, { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } } };
    const int wc = (width + ctu_size - 1) / ctu_size;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const orc_sao_ctu_param *p = &params[(y / ctu_size) * wc + x / ctu_size];
            const int c = src[(ptrdiff_t)y * src_stride + x];
            int v = c;
            if (p->type_idx == 4) {   /* band offset: src/encode.cpp:1844-1862, 1737-1749 */
                const int band = c >> 3, k = (band - p->band_position) & 31;
                if (k < 4) v = clip3(0, 255, c + p->offset[k]);
            } else if (p->type_idx >= 0) {   /* edge offset classes: :1607-1736; the two neighbours are samples before SAO */
                const int (*n)[2] = nb[p->type_idx];
                const int xa = x + n[0][0], ya = y + n[0][1], xb = x + n[1][0], yb = y + n[1][1];
                if (xa >= 0 && xa < width && xb >= 0 && xb < width && ya >= 0 && ya < height && yb >= 0 && yb < height) {
                    const int cat = eo_table[sao_sign(c - src[(ptrdiff_t)ya * src_stride + xa]) + sao_sign(c - src[(ptrdiff_t)yb * src_stride + xb]) + 2];
                    if (cat) v = clip3(0, 255, c + p->offset[cat - 1]);
                }
            }
            dst[(ptrdiff_t)y * dst_stride + x] = (uint8_t)v;
        }
}

/* ---- boundary strengths from picture-level maps (src/Deblock.cpp:344-503) ---- */
static int orc_bs_pair(int intra_slice, const int16_t *cur_mv, int cur_cbf, const int16_t *nb_mv, int nb_cbf, int unit_edge)
{
    if (unit_edge) {                                   /* src/Deblock.cpp:346-353 */
        if (intra_slice) return 2;
        if (cur_cbf || nb_cbf) return 1;
    }
    if (!intra_slice)                                  /* :355-364, one reference picture */
        return abs(nb_mv[0] - cur_mv[0]) >= 4 || abs(nb_mv[1] - cur_mv[1]) >= 4;
    return 0;
}

void orc_deblock_strengths(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf,
                           const int16_t *mv, uint8_t *vertical_bs, uint8_t *horizontal_bs)
{
    const int w4 = width >> 2, w8 = width >> 3, bsw = width >> 3;
    memset(vertical_bs, 0, (size_t)w8 * (height >> 2));
    memset(horizontal_bs, 0, (size_t)w8 * (height >> 2));
    /* the reference's order: one call per unit, each writing the edges on its own top and left side and inside it */
    for (int y4 = 0; y4 < (height >> 2); y4++)
        for (int x4 = 0; x4 < w4; x4++) {
            const int lg = unit_log2[y4 * w4 + x4], size = 1 << lg, x0 = x4 * 4, y0 = y4 * 4;
            if (lg < 2 || (x0 & (size - 1)) || (y0 & (size - 1))) continue;       /* not the top-left 4x4 of a unit */
            const int16_t *cur_mv = mv + 2 * ((y0 >> 3) * w8 + (x0 >> 3));
            const int cur_cbf = cbf[y4 * w4 + x4];
            const int inner = lg > 2 && !intra_slice;                               /* :424, :476 */
            if ((y0 & 7) == 0 && y0 != 0)                                           /* :393-422 */
                for (int i = 0; i < size; i += 4) {
                    const int bs = orc_bs_pair(intra_slice, cur_mv, cur_cbf, mv + 2 * (((y0 >> 3) - 1) * w8 + (x0 >> 3) + (i >> 3)),
                                               cbf[(y4 - 1) * w4 + x4 + (i >> 2)], 1);
                    if (bs) horizontal_bs[((x0 + i) + y0 * bsw) >> 2] = (uint8_t)bs;
                }
            if (inner)                                                              /* :426-447 */
                for (int j = 8; j < size; j += 8)
                    for (int i = 0; i < size; i += 4) {
                        const int bs = orc_bs_pair(intra_slice, cur_mv, cur_cbf, mv + 2 * (((y0 >> 3) + (j >> 3) - 1) * w8 + (x0 >> 3) + (i >> 3)),
                                                   cbf[(y4 + (j >> 2) - 1) * w4 + x4 + (i >> 2)], 0);
                        if (bs) horizontal_bs[((x0 + i) + (y0 + j) * bsw) >> 2] = (uint8_t)bs;
                    }
            if ((x0 & 7) == 0 && x0 != 0)                                           /* :449-474 */
                for (int i = 0; i < size; i += 4) {
                    const int bs = orc_bs_pair(intra_slice, cur_mv, cur_cbf, mv + 2 * (((y0 >> 3) + (i >> 3)) * w8 + (x0 >> 3) - 1),
                                               cbf[(y4 + (i >> 2)) * w4 + x4 - 1], 1);
                    if (bs) vertical_bs[(x0 >> 3) + ((y0 + i) >> 2) * bsw] = (uint8_t)bs;
                }
            if (inner)                                                              /* :478-502 */
                for (int j = 0; j < size; j += 4)
                    for (int i = 8; i < size; i += 8) {
                        const int bs = orc_bs_pair(intra_slice, cur_mv, cur_cbf, mv + 2 * (((y0 >> 3) + (j >> 3)) * w8 + (x0 >> 3) + (i >> 3) - 1),
                                                   cbf[(y4 + (j >> 2)) * w4 + x4 + (i >> 2) - 1], 0);
                        if (bs) vertical_bs[((x0 + i) >> 3) + ((y0 + j) >> 2) * bsw] = (uint8_t)bs;
                    }
        }
}

/* ---- intra chroma mode search (src/encode.cpp:4326-4365) ---- */
void orc_intra_chroma_search(const uint8_t *src_u, const uint8_t *src_v, int src_stride, const uint8_t *ref_u, const uint8_t *ref_v,
                             int size, int luma_mode, orc_chroma_result *out)
{
    const int cand[5] = { 0, 26, 10, 1, luma_mode };            /* nMostModeC, :4260-4263, :4327 */
    const int lg = size == 4 ? 2 : size == 8 ? 3 : size == 16 ? 4 : 5;
    uint8_t pred[2][32 * 32];
    uint32_t best = 0xffffffffu;                                /* MAX_SAD */
    out->best_idx = 0;
    for (int m = 0; m < 5; m++) {
        int mode = cand[m];
        if (m < 4 && mode == luma_mode) mode = 34;               /* :4333 */
        out->modes[m] = mode;
        orc_intra_pred(pred[0], ref_u, size, mode, size, 0);
        orc_intra_pred(pred[1], ref_v, size, mode, size, 0);
        out->satd_u[m] = orc_satd_table(lg - 2, src_u, src_stride, pred[0], size);
        out->satd_v[m] = orc_satd_table(lg - 2, src_v, src_stride, pred[1], size);
        const uint32_t sum = out->satd_u[m] + out->satd_v[m];
        if (sum < best) {                                        /* :4357, strict */
            best = sum;
            out->best_idx = m;
        }
    }
    out->best_mode = out->modes[out->best_idx];
}

/* ---- MV predictor and merge candidates (src/encode.cpp:2703-2809) ---- */
typedef struct { int ok; int16_t x, y; } orc_nb;

static orc_nb orc_mvp_neighbour(int width, int height, const uint8_t *valid, const int16_t *mv, int px, int py)
{
    /* px, py: luma position of the 4x4 block the mode flag is read for; the vector is that of its 8x8 block */
    orc_nb n = { 0, 0, 0 };
    if (px < 0 || py < 0 || px >= width || py >= height) return n;
    n.ok = valid[(py >> 2) * (width >> 2) + (px >> 2)] != 0;
    n.x = mv[2 * ((py >> 3) * (width >> 3) + (px >> 3))];
    n.y = mv[2 * ((py >> 3) * (width >> 3) + (px >> 3)) + 1];
    return n;
}

int orc_fill_mvp_cand(int width, int height, const uint8_t *valid, const int16_t *mv, int x0, int y0, int size_x, int size_y,
                      int pu_mode, int pu_idx, int16_t out[7][2])
{
    const int no_b1 = pu_mode == 1 && pu_idx == 1, no_a1 = pu_mode == 2 && pu_idx == 1;        /* :2713-2714 */
    /* the flag of a neighbour is read at 4x4 precision, its vector at 8x8 precision, both relative to the PU (:2721-2732):
     * vector row size_y/8 (or size_y/8 - 1) of the cache is the 8x8 row of luma line y0 + 8 (size_y/8) (- 8) */
    const orc_nb fbl = orc_mvp_neighbour(width, height, valid, mv, x0 - 1, y0 + size_y);
    const orc_nb fl = orc_mvp_neighbour(width, height, valid, mv, x0 - 1, y0 + size_y - 4);
    const orc_nb ftr = orc_mvp_neighbour(width, height, valid, mv, x0 + size_x, y0 - 1);
    const orc_nb ft = orc_mvp_neighbour(width, height, valid, mv, x0 + size_x - 4, y0 - 1);
    const orc_nb ftl = orc_mvp_neighbour(width, height, valid, mv, x0 - 1, y0 - 1);
    orc_nb bl = orc_mvp_neighbour(width, height, valid, mv, x0 - 1, y0 + 8 * (size_y / 8));
    orc_nb l = orc_mvp_neighbour(width, height, valid, mv, x0 - 1, y0 + 8 * (size_y / 8 - 1));
    orc_nb tr = orc_mvp_neighbour(width, height, valid, mv, x0 + 8 * (size_x / 8), y0 - 1);
    orc_nb t = orc_mvp_neighbour(width, height, valid, mv, x0 + 8 * (size_x / 8 - 1), y0 - 1);
    orc_nb tl = ftl;
    bl.ok = fbl.ok; l.ok = fl.ok; tr.ok = ftr.ok; t.ok = ft.ok;
    memset(out, 0, 7 * 2 * sizeof(int16_t));
    int n = 0;
    const int grp_l = bl.ok || l.ok, grp_t = tr.ok || t.ok || tl.ok;
    if (grp_l) { const orc_nb c = bl.ok ? bl : l; out[n][0] = c.x; out[n][1] = c.y; n++; }
    if (grp_t) { const orc_nb c = tr.ok ? tr : t.ok ? t : tl; out[n][0] = c.x; out[n][1] = c.y; n++; }
    if (grp_l && grp_t && out[0][0] == out[1][0] && out[0][1] == out[1][1]) out[1][0] = out[1][1] = 0;       /* :2748-2752 */
    if (n < 2) out[1][0] = out[1][1] = 0;                                                                      /* :2756-2758 */
    int16_t (*mrg)[2] = out + 2;
    n = 0;
#define ORC_PUSH(c) do { mrg[n][0] = (c).x; mrg[n][1] = (c).y; n++; } while (0)
#define ORC_DIFF(a, b) ((a).x != (b).x || (a).y != (b).y)
    if (!no_a1 && l.ok) ORC_PUSH(l);
    if (!no_b1 && t.ok && (!l.ok || no_a1 || ORC_DIFF(l, t))) ORC_PUSH(t);
    if (tr.ok && (!t.ok || no_b1 || ORC_DIFF(t, tr))) ORC_PUSH(tr);
    if (bl.ok && (!l.ok || no_a1 || ORC_DIFF(l, bl))) ORC_PUSH(bl);
    if (n < 4) {
        if (tl.ok && (!l.ok || no_a1 || ORC_DIFF(l, tl)) && (!t.ok || no_b1 || ORC_DIFF(t, tl))) {
            ORC_PUSH(tl);
        } else {                                               /* a zero vector unless one is in the list already, :2794-2805 */
            int zero = 0;
            for (int i = 0; i < n; i++) zero += mrg[i][0] == 0 && mrg[i][1] == 0;
            if (!zero) { mrg[n][0] = mrg[n][1] = 0; n++; }
        }
    }
#undef ORC_PUSH
#undef ORC_DIFF
    return n;
}
