/*
 * x265_rdoq.c - CPU statement of the rate-distortion optimised quantiser (TEST INFRASTRUCTURE, part of libx265_oracle.so).
 *
 * PARITY UNPINNED BY CONSTRUCTION. The reference has no level decision: src/RdoQ.cpp is compiled out (#if RDOQ, never
 * defined), its only contents are the bit-cost table builders estimateBit / estCBFBit / estSignificantCoeffGroupMapBit /
 * estSignificantMapBit / estSignificantCoefficientsBit (src/RdoQ.cpp:67-305) over the structure estBitsSbacStruct
 * (inc/RdoQ.h:38-49), and the function that would use them (My_xRateDistOptQuant, inc/RdoQ.h:55) has no definition.
 * What is stated here is therefore this build's own algorithm - the level decision of the HEVC test model's
 * xRateDistOptQuant (per coefficient the rounded level or one less or zero, whole 4x4 groups dropped when that is
 * cheaper, the last significant position moved back over coefficients of magnitude one, the empty block) priced with
 * exactly those tables against the residual syntax the reference's entropy coder writes (xEncodeCoeffNxN,
 * src/encode.cpp:2385-2553) - in integer arithmetic throughout, so that the HIP kernel can be held to it bit for bit.
 * It checks the kernel against this statement; it says nothing about any reference behaviour.
 */
#include "x265_oracle.h"

#include <stdlib.h>
#include <string.h>

#define IEP 32768 /* one bypass bin, in the tables' 15-bit fixed point */

/* g_entropyBits (src/RdoQ.cpp:52-62): bits of coding a bin with state s ^ bin, 15-bit fixed point */
static const int32_t k_entropy_bits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb
};

/* context layout of the reference (inc/x265.h:286-311) */
enum { OFF_QT_CBF = 29, OFF_SIG_CG = 40, OFF_SIG = 44, OFF_LAST_X = 86, OFF_LAST_Y = 116, OFF_ONE = 146, OFF_ABS = 170,
       N_SIG_CG = 2, N_SIG_LUMA = 27, N_LAST = 15, N_ONE_LUMA = 16, N_ABS_LUMA = 4, N_QT_CBF = 5 };

static const uint8_t k_group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };

static int ilog2i(int v)
{
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

/* estimateBit (src/RdoQ.cpp:67-305) for one block size and component, from a set of context states. The reference fills only
 * the significance entries the block size uses; here every entry of the component is filled, the used ones with the same values. */
void orc_rdoq_est_bits(const uint8_t *ctx, int log2_size, int is_luma, orc_rdoq_tables *t)
{
    memset(t, 0, sizeof *t);
    for (int c = 0; c < N_SIG_CG; c++) /* estSignificantCoeffGroupMapBit :120-135 */
        for (int b = 0; b < 2; b++) t->sig_cg[c][b] = k_entropy_bits[ctx[OFF_SIG_CG + c + (is_luma ? 0 : N_SIG_CG)] ^ b];
    const int nsig = is_luma ? N_SIG_LUMA : 15; /* estSignificantMapBit :142-197 */
    for (int c = 0; c < nsig; c++)
        for (int b = 0; b < 2; b++) t->sig[c][b] = k_entropy_bits[ctx[OFF_SIG + (is_luma ? 0 : N_SIG_LUMA) + c] ^ b];
    /* :199-226: prefix bins of the last position, cumulated per group index */
    const int off = is_luma ? (log2_size - 2) * 3 + ((log2_size - 1) >> 2) : 0, sh = is_luma ? (log2_size + 1) >> 2 : log2_size - 2;
    const int gmax = k_group_idx[(1 << log2_size) - 1];
    for (int xy = 0; xy < 2; xy++) {
        int32_t *dst = xy ? t->last_y : t->last_x;
        const int base = (xy ? OFF_LAST_Y : OFF_LAST_X) + (is_luma ? 0 : N_LAST);
        int32_t bits = 0;
        int c;
        for (c = 0; c < gmax; c++) {
            const uint8_t s = ctx[base + off + (c >> sh)];
            dst[c] = bits + k_entropy_bits[s ^ 0];
            bits += k_entropy_bits[s ^ 1];
        }
        dst[c] = bits;
    }
    const int none = is_luma ? N_ONE_LUMA : 8, nabs = is_luma ? N_ABS_LUMA : 2; /* estSignificantCoefficientsBit :255-290 */
    for (int c = 0; c < none; c++)
        for (int b = 0; b < 2; b++) t->one[c][b] = k_entropy_bits[ctx[OFF_ONE + (is_luma ? 0 : N_ONE_LUMA) + c] ^ b];
    for (int c = 0; c < nabs; c++)
        for (int b = 0; b < 2; b++) t->abs[c][b] = k_entropy_bits[ctx[OFF_ABS + (is_luma ? 0 : N_ABS_LUMA) + c] ^ b];
    /* estCBFBit :85-113: the flag of the context the reference's coder uses for a transform unit of this component */
    for (int b = 0; b < 2; b++) t->cbf[b] = k_entropy_bits[ctx[OFF_QT_CBF + (is_luma ? 0 : N_QT_CBF)] ^ b];
}

/* The price of a bit. With err = (|coef| * Q - (level << qbits)) >> (qbits - 4), i.e. the quantisation error in sixteenths of
 * a level, the distortion term of a cost is err * err; one bit of rate then weighs lambda * 256 * (Q * 2^tshift / 2^qbits)^2
 * of those units (tshift = 15 - 8 - log2 size, the scaling of the forward transform), i.e. per unit of the 15-bit tables
 * 1 / 32768 of that. The weight is handed on as a mantissa below 2^30 and a shift: rate cost = (mantissa * bits) >> shift. */
void orc_rdoq_lambda(double lambda, unsigned qp, int log2_size, uint32_t *mant, int32_t *shift)
{
    static const int q6[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
    const int qbits = 14 + (int)(qp / 6) + (15 - 8 - log2_size), tshift = 15 - 8 - log2_size;
    double v = lambda * 256.0 / 32768.0;
    const double g = (double)q6[qp % 6];
    v *= g * g;
    for (int i = 0; i < 2 * tshift; i++) v *= 2.0;
    for (int i = 0; i < 2 * qbits; i++) v *= 0.5;
    int s = 0;
    if (v <= 0) {
        *mant = 0, *shift = 0;
        return;
    }
    while (v < (double)(1 << 29) && s < 60) v *= 2.0, s++;
    while (v >= (double)(1 << 30) && s > 0) v *= 0.5, s--;
    if (v >= (double)(1 << 30)) v = (double)((1 << 30) - 1);
    *mant = (uint32_t)(v + 0.5);
    if (*mant >= (1u << 30)) *mant = (1u << 30) - 1;
    *shift = s;
}

static int64_t rc(uint32_t m, int s, int64_t bits) { return (int64_t)(((uint64_t)m * (uint64_t)bits) >> s); }

/* scan orders: up-right diagonal, horizontal, vertical over a 4x4 group, and the same order over the groups of the block
 * (src/table.cpp:268-511) */
static void order4(int kind, int n, uint8_t *o)
{
    int k = 0;
    if (kind == 0) {
        for (int d = 0; d < 2 * n - 1; d++)
            for (int y = d < n ? d : n - 1; y >= 0 && d - y < n; y--) o[k++] = (uint8_t)(y * n + (d - y));
    } else {
        for (int i = 0; i < n * n; i++) o[k++] = (uint8_t)(kind == 1 ? i : (i % n) * n + i / n);
    }
}

static void make_scan(int kind, int log2, uint16_t *scan, uint8_t *groups)
{
    const int n = 1 << log2, g = n / 4;
    uint8_t in4[16];
    order4(kind, 4, in4);
    if (g == 1) groups[0] = 0;
    else order4(kind, g, groups);
    int k = 0;
    for (int gi = 0; gi < g * g; gi++) {
        const int gy = groups[gi] / g, gx = groups[gi] % g;
        for (int i = 0; i < 16; i++) scan[k++] = (uint16_t)((gy * 4 + in4[i] / 4) * n + gx * 4 + in4[i] % 4);
    }
}

/* getSigCtxInc (src/encode.cpp:2187) */
static int sig_ctx(int pattern, int scan_idx, int x, int y, int log2, int luma)
{
    static const uint8_t map4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
    if (x + y == 0) return 0;
    if (log2 == 2) return map4[4 * y + x];
    const int off = log2 == 3 ? (scan_idx == 0 ? 9 : 15) : (luma ? 21 : 12);
    const int xs = x & 3, ys = y & 3;
    int cnt;
    if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
    else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
    else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
    else cnt = 2;
    return ((luma && ((x >> 2) + (y >> 2)) > 0) ? 3 : 0) + off + cnt;
}

/* bits of the magnitude and sign of a level >= 1 as the coefficient coder writes them (greater-than-one flag for the first
 * eight levels of a group, greater-than-two flag for the first of those above one, Golomb-Rice remainder: src/encode.cpp:2497-2551) */
static int32_t level_bits(const orc_rdoq_tables *t, int level, int c1, int ctx_set, int c1_idx, int c2_idx, int rice, int *escaped)
{
    const int base = c1_idx < 8 ? 2 + (c2_idx < 1) : 1;
    int32_t r = IEP;
    *escaped = 0;
    if (level >= base) {
        const unsigned sym = (unsigned)(level - base);
        int len;
        if (sym < (3u << rice)) {
            len = (int)(sym >> rice) + 1 + rice;
        } else {
            unsigned l = (unsigned)rice, code = sym - (3u << rice);
            while (code >= (1u << l)) code -= 1u << l, l++;
            len = (int)(3 + l + 1 - (unsigned)rice) + (int)l;
        }
        r += len * IEP;
        if (c1_idx < 8) {
            r += t->one[4 * ctx_set + c1][1];
            if (c2_idx < 1) r += t->abs[ctx_set][1];
        }
        *escaped = 1;
    } else if (level == 1) {
        r += t->one[4 * ctx_set + c1][0];
    } else { /* level 2 below a base of 3 */
        r += t->one[4 * ctx_set + c1][1] + t->abs[ctx_set][0];
    }
    return r;
}

static int32_t last_bits(const orc_rdoq_tables *t, int x, int y, int scan_idx)
{
    if (scan_idx == 2) {
        const int s = x;
        x = y, y = s;
    }
    const int gx = k_group_idx[x], gy = k_group_idx[y];
    int32_t r = t->last_x[gx] + t->last_y[gy];
    if (gx > 3) r += IEP * ((gx - 2) >> 1);
    if (gy > 3) r += IEP * ((gy - 2) >> 1);
    return r;
}

uint32_t orc_rdoq_quant(int16_t *level, const int16_t *coef, int stride, int n, unsigned qp, int is_luma, int scan_idx,
                        const orc_rdoq_tables *t, uint32_t lam_m, int32_t lam_s)
{
    static const int q6[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
    const int log2 = ilog2i(n), side = n >> 2, nn = n * n;
    const int qbits = 14 + (int)(qp / 6) + (15 - 8 - log2), q = q6[qp % 6], eshift = qbits - 4;
    uint16_t scan[1024];
    uint8_t scan_cg[64];
    make_scan(scan_idx, log2, scan, scan_cg);

    static __thread uint32_t ld[1024], pos_b[1024], lev_b[1024];
    static __thread int16_t lvl[1024];
    int32_t cg_b[64];
    int64_t cg_d[64];
    uint8_t cg_flag[64];
    memset(cg_flag, 0, sizeof cg_flag);
#define D_OF(i, L) (((int64_t)ld[i] - ((int64_t)(L) << qbits)) >> eshift)
    int top = -1;
    for (int i = 0; i < nn; i++) {
        const int p = scan[i], y = p >> log2, x = p - (y << log2);
        ld[i] = (uint32_t)(abs(coef[y * stride + x]) * q);
        uint32_t mx = (uint32_t)(((uint64_t)ld[i] + (1u << (qbits - 1))) >> qbits);
        if (mx > 32767) mx = 32767;
        lvl[i] = (int16_t)mx; /* the rounded level until the decision */
        if (mx) top = i;
    }
    for (int y = 0; y < n; y++) memset(level + y * stride, 0, (size_t)n * 2);
    if (top < 0) return 0;

    int64_t d_all0 = 0, d_beyond = 0;
    for (int i = 0; i < nn; i++) {
        const int64_t e = D_OF(i, 0);
        d_all0 += e * e;
        if (i > top) d_beyond += e * e;
    }

    /* ---- levels, group by group from the last one down ---- */
    int64_t tot_d = 0, tot_b = 0;
    int c1_carry = 1;
    const int last_set = top >> 4;
    for (int set = last_set; set >= 0; set--) {
        const int first = set << 4, hi = set == last_set ? top : first + 15;
        const int g = scan_cg[set], gy = g / side, gx = g - gy * side;
        const int right = gx < side - 1 && cg_flag[g + 1], below = gy < side - 1 && cg_flag[g + side];
        const int pattern = n == 4 ? -1 : right + 2 * below, ctx_cg = right || below;
        const int ctx_set = ((set > 0 && is_luma) ? 2 : 0) + (c1_carry == 0);
        int c1 = 1, c1_idx = 0, c2_idx = 0, rice = 0, nnz = 0;
        int64_t s_d = 0, s_b = 0, z_d = 0;
        for (int pos = hi; pos >= first; pos--) {
            const int p = scan[pos], y = p >> log2, x = p - (y << log2);
            const int is_last = pos == top, mx = lvl[pos];
            const int64_t e0 = D_OF(pos, 0), d0 = e0 * e0;
            z_d += d0;
            int32_t sig0 = 0, sig1 = 0;
            if (!is_last) {
                const int sc = sig_ctx(pattern, scan_idx, x, y, log2, is_luma);
                sig0 = t->sig[sc][0], sig1 = t->sig[sc][1];
            }
            int best_l = 0, best_esc = 0;
            int64_t best_d = d0, best_cost = d0 + rc(lam_m, lam_s, sig0);
            int32_t best_b = sig0, best_lb = 0;
            int have = !is_last; /* the last position holds a level: zero is no choice there */
            for (int l = mx; l >= 1 && l >= mx - 1; l--) {
                int esc;
                const int32_t lb = level_bits(t, l, c1, ctx_set, c1_idx, c2_idx, rice, &esc);
                const int64_t e = D_OF(pos, l), d = e * e, cost = d + rc(lam_m, lam_s, (int64_t)sig1 + lb);
                if (!have || cost < best_cost) {
                    have = 1;
                    best_l = l, best_d = d, best_cost = cost, best_b = sig1 + lb, best_lb = lb, best_esc = esc;
                }
            }
            lvl[pos] = (int16_t)best_l;
            pos_b[pos] = (uint32_t)best_b;
            lev_b[pos] = (uint32_t)best_lb;
            s_d += best_d, s_b += best_b;
            if (best_l > 0) {
                nnz++;
                if (best_esc && best_l > 3 * (1 << rice)) rice = rice + 1 < 4 ? rice + 1 : 4;
                if (c1_idx < 8) {
                    if (best_l > 1) c1 = 0;
                    else if (c1 > 0 && c1 < 3) c1++;
                }
                if (best_l > 1) c2_idx++;
                c1_idx++;
            }
        }
        const int flag_coded = set != last_set && set != 0;
        int keep = nnz > 0;
        if (flag_coded && nnz > 0) {
            const int64_t cost_keep = s_d + rc(lam_m, lam_s, s_b + t->sig_cg[ctx_cg][1]), cost_drop = z_d + rc(lam_m, lam_s, t->sig_cg[ctx_cg][0]);
            if (cost_drop < cost_keep) keep = 0;
        }
        if (keep) {
            cg_flag[g] = 1;
            cg_d[set] = s_d, cg_b[set] = (int32_t)(s_b + (flag_coded ? t->sig_cg[ctx_cg][1] : 0));
            c1_carry = c1;
        } else if (flag_coded) {
            for (int pos = first; pos <= hi; pos++) lvl[pos] = 0, pos_b[pos] = 0;
            cg_flag[g] = 0;
            cg_d[set] = z_d, cg_b[set] = t->sig_cg[ctx_cg][0];
        } else { /* group 0 without a level: its significance flags are coded one by one; the coder takes its group flag as set */
            cg_flag[g] = 1;
            cg_d[set] = s_d, cg_b[set] = (int32_t)s_b;
        }
        tot_d += cg_d[set], tot_b += cg_b[set];
    }

    /* ---- the last position: back over levels of magnitude one while that is cheaper ---- */
    int best_last = -1;
    int64_t best_d = 0, best_b = 0, best_cost = 0;
    {
        int64_t above_d = d_beyond; /* distortion of everything behind the candidate, all zero */
        int64_t below_sets_d = tot_d, below_sets_b = tot_b;
        for (int set = last_set; set >= 0; set--) {
            const int first = set << 4, hi = set == last_set ? top : first + 15;
            const int g = scan_cg[set];
            below_sets_d -= cg_d[set], below_sets_b -= cg_b[set];
            int64_t in_d = 0, in_b = 0; /* the positions of this group in front of the candidate, as decided */
            for (int pos = first; pos <= hi; pos++) {
                const int64_t e = D_OF(pos, lvl[pos]);
                in_d += e * e, in_b += pos_b[pos];
            }
            int stop = 0;
            for (int pos = hi; pos >= first; pos--) {
                const int64_t e = D_OF(pos, lvl[pos]), d = e * e, e0 = D_OF(pos, 0);
                in_d -= d, in_b -= pos_b[pos];
                if (lvl[pos] > 0 && cg_flag[g]) {
                    const int p = scan[pos], y = p >> log2, x = p - (y << log2);
                    const int64_t cd = above_d + d + in_d + below_sets_d;
                    const int64_t cb = (int64_t)lev_b[pos] + in_b + below_sets_b + last_bits(t, x, y, scan_idx);
                    const int64_t cost = cd + rc(lam_m, lam_s, cb);
                    if (best_last < 0 || cost < best_cost) best_last = pos, best_cost = cost, best_d = cd, best_b = cb;
                    if (lvl[pos] > 1) {
                        stop = 1;
                        break;
                    }
                }
                above_d += e0 * e0;
            }
            if (stop) break;
        }
    }
    /* ---- the empty block ---- */
    const int64_t cost_coded = best_d + rc(lam_m, lam_s, best_b + t->cbf[1]), cost_empty = d_all0 + rc(lam_m, lam_s, t->cbf[0]);
    if (best_last < 0 || cost_empty < cost_coded) return 0;
    uint32_t sum = 0;
    for (int i = 0; i <= best_last; i++) {
        if (!lvl[i]) continue;
        const int p = scan[i], y = p >> log2, x = p - (y << log2);
        const int c = coef[y * stride + x];
        level[y * stride + x] = (int16_t)(c < 0 ? -lvl[i] : lvl[i]);
        sum += (uint32_t)lvl[i];
    }
#undef D_OF
    return sum;
}
