// rdoq_device.h - the level decision of rdoq.hip as device functions, shared by its batch kernel and by the transform chain of
// the analysis kernel (each translation unit that includes this file has its own copy of the scan tables and uploads it
// with rdoq_upload_scans()). Include from a .hip file only.
#pragma once
#include "xmi_common.h"
#include "host/syntax.h"

namespace {

constexpr int kIep = 32768;                         // one bypass bin in the tables' 15-bit fixed point
__device__ uint16_t g_rdoq_scan[3][4][1024];        // coefficient scans [diagonal / horizontal / vertical][log2 size - 2]: positions y * size + x
__device__ uint16_t g_rdoq_inv[3][4][1024];         // the inverse: scan index of position y * size + x
__device__ uint8_t g_rdoq_scan_cg[3][4][64];        // the same orders over the 4x4 groups
__constant__ int k_rdoq_q6[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // src/table.cpp:150
__constant__ uint8_t k_group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
__constant__ uint8_t k_sig_map4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

// the tables of every block size (index log2 size - 2) and component (0 luma, 1 chroma) of one picture, and the weight of a bit
struct RdoqSet {
    xmi_rdoq_tables tables[4][2];
    xmi_rdoq_lambda lambda[4][2];
};

inline int rdoq_upload_scans()
{
    static uint16_t scan[3][4][1024], inv[3][4][1024];
    static uint8_t groups[3][4][64];
    for (int k = 0; k < 3; k++)
        for (int l = 2; l <= 5; l++) {
            const int n = 1 << l;
            memcpy(scan[k][l - 2], xmi::syntax::coef_scan(k, l), (size_t)n * n * 2);
            memcpy(groups[k][l - 2], xmi::syntax::group_scan(k, l), (size_t)(n / 4) * (n / 4));
            for (int i = 0; i < n * n; i++) inv[k][l - 2][scan[k][l - 2][i]] = (uint16_t)i;
        }
    XMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rdoq_scan), scan, sizeof scan));
    XMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rdoq_inv), inv, sizeof inv));
    XMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rdoq_scan_cg), groups, sizeof groups));
    return XMI_OK;
}

__device__ __forceinline__ long long rate_cost(uint32_t m, int s, long long bits) { return (long long)(((unsigned long long)m * (unsigned long long)bits) >> s); }

// getSigCtxInc (src/encode.cpp:2187)
__device__ __forceinline__ int sig_ctx(int pattern, int scan_idx, int x, int y, int log2, int luma)
{
    if (x + y == 0) return 0;
    if (log2 == 2) return k_sig_map4[4 * y + x];
    const int off = log2 == 3 ? (scan_idx == 0 ? 9 : 15) : (luma ? 21 : 12);
    const int xs = x & 3, ys = y & 3;
    int cnt;
    if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
    else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
    else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
    else cnt = 2;
    return ((luma && ((x >> 2) + (y >> 2)) > 0) ? 3 : 0) + off + cnt;
}

// magnitude and sign of a level >= 1 as the coder writes them (src/encode.cpp:2497-2551)
__device__ __forceinline__ int level_bits(const xmi_rdoq_tables &t, int level, int c1, int ctx_set, int c1_idx, int c2_idx, int rice, int &escaped)
{
    const int base = c1_idx < 8 ? 2 + (c2_idx < 1) : 1;
    int r = kIep;
    escaped = 0;
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
        r += len * kIep;
        if (c1_idx < 8) {
            r += t.one[4 * ctx_set + c1][1];
            if (c2_idx < 1) r += t.abs[ctx_set][1];
        }
        escaped = 1;
    } else if (level == 1) {
        r += t.one[4 * ctx_set + c1][0];
    } else {
        r += t.one[4 * ctx_set + c1][1] + t.abs[ctx_set][0];
    }
    return r;
}

__device__ __forceinline__ int last_bits(const xmi_rdoq_tables &t, int x, int y, int scan_idx)
{
    if (scan_idx == 2) {
        const int s = x;
        x = y, y = s;
    }
    const int gx = k_group_idx[x], gy = k_group_idx[y];
    int r = t.last_x[gx] + t.last_y[gy];
    if (gx > 3) r += kIep * ((gx - 2) >> 1);
    if (gy > 3) r += kIep * ((gy - 2) >> 1);
    return r;
}

// The decision for one N x N block, on one lane. ld(i): the magnitude of the coefficient at scan index i times the quantiser
// scale; lvl[i]: its rounded level on entry, the decided level on return; top: the last scan index with a rounded level (>= 0);
// d_all / d_beyond: squared error of coding nothing, over the block / behind `top`. pos_b (N*N words), cg_d, cg_b, cg_flag (one
// entry per 4x4 group) are scratch. Returns the scan index of the last coded level - the levels behind it are to be taken as
// zero - or -1 for the empty block.
template <int N, class LdFn>
__device__ int rdoq_decide(LdFn ld, const uint16_t *scan, const uint8_t *scan_cg, int16_t *lvl, uint32_t *pos_b, long long *cg_d, int *cg_b, uint8_t *cg_flag,
                           const xmi_rdoq_tables &t, int top, long long d_all, long long d_beyond, int qbits, uint32_t lam_m, int lam_s, int scan_idx, int is_luma)
{
    constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5, SIDE = N / 4;
    const int eshift = qbits - 4;
    auto d_of = [&](int i, int l) -> long long {
        const long long e = ((long long)ld(i) - ((long long)l << qbits)) >> eshift;
        return e * e;
    };
    // ---- the levels, group by group from the last one down ----
    long long tot_d = 0, tot_b = 0;
    int c1_carry = 1;
    const int last_set = top >> 4;
    for (int set = last_set; set >= 0; set--) {
        const int first = set << 4, hi = set == last_set ? top : first + 15;
        const int g = scan_cg[set], gy = g / SIDE, gx = g - gy * SIDE;
        const int right = gx < SIDE - 1 && cg_flag[g + 1], below = gy < SIDE - 1 && cg_flag[g + SIDE];
        const int pattern = N == 4 ? -1 : right + 2 * below, ctx_cg = right || below;
        const int ctx_set = ((set > 0 && is_luma) ? 2 : 0) + (c1_carry == 0);
        int c1 = 1, c1_idx = 0, c2_idx = 0, rice = 0, nnz = 0;
        long long s_d = 0, s_b = 0, z_d = 0;
        for (int pos = hi; pos >= first; pos--) {
            const int pp = scan[pos], y = pp >> LOG2, x = pp - (y << LOG2);
            const bool is_last = pos == top;
            const int mx = lvl[pos];
            const long long d0 = d_of(pos, 0);
            z_d += d0;
            int sig0 = 0, sig1 = 0;
            if (!is_last) {
                const int sc = sig_ctx(pattern, scan_idx, x, y, LOG2, is_luma);
                sig0 = t.sig[sc][0], sig1 = t.sig[sc][1];
            }
            int best_l = 0, best_esc = 0, best_b = sig0;
            long long best_d = d0, best_cost = d0 + rate_cost(lam_m, lam_s, sig0);
            bool have = !is_last;      // the last position holds a level: zero is no choice there
            for (int l = mx; l >= 1 && l >= mx - 1; l--) {
                int esc;
                const int lb = level_bits(t, l, c1, ctx_set, c1_idx, c2_idx, rice, esc);
                const long long d = d_of(pos, l), cost = d + rate_cost(lam_m, lam_s, (long long)sig1 + lb);
                if (!have || cost < best_cost) {
                    have = true;
                    best_l = l, best_d = d, best_cost = cost, best_b = sig1 + lb, best_esc = esc;
                }
            }
            lvl[pos] = (int16_t)best_l;
            pos_b[pos] = (uint32_t)best_b;
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
        const bool flag_coded = set != last_set && set != 0;
        bool keep = nnz > 0;
        if (flag_coded && nnz > 0) {
            const long long cost_keep = s_d + rate_cost(lam_m, lam_s, s_b + t.sig_cg[ctx_cg][1]), cost_drop = z_d + rate_cost(lam_m, lam_s, t.sig_cg[ctx_cg][0]);
            if (cost_drop < cost_keep) keep = false;
        }
        if (keep) {
            cg_flag[g] = 1;
            cg_d[set] = s_d, cg_b[set] = (int)(s_b + (flag_coded ? t.sig_cg[ctx_cg][1] : 0));
            c1_carry = c1;
        } else if (flag_coded) {
            for (int pos = first; pos <= hi; pos++) lvl[pos] = 0, pos_b[pos] = 0;
            cg_flag[g] = 0;
            cg_d[set] = z_d, cg_b[set] = t.sig_cg[ctx_cg][0];
        } else {      // group 0 without a level: its significance flags are coded one by one; the coder takes its group flag as set
            cg_flag[g] = 1;
            cg_d[set] = s_d, cg_b[set] = (int)s_b;
        }
        tot_d += cg_d[set], tot_b += cg_b[set];
    }
    // ---- the last position: back over levels of magnitude one while that is cheaper ----
    int best_last = -1;
    long long best_d = 0, best_b = 0, best_cost = 0;
    {
        long long above_d = d_beyond, below_sets_d = tot_d, below_sets_b = tot_b;
        for (int set = last_set; set >= 0; set--) {
            const int first = set << 4, hi = set == last_set ? top : first + 15;
            const int g = scan_cg[set], gy = g / SIDE, gx = g - gy * SIDE;
            const int right = gx < SIDE - 1 && cg_flag[g + 1], below = gy < SIDE - 1 && cg_flag[g + SIDE];
            const int pattern = N == 4 ? -1 : right + 2 * below;      // (the groups behind were final when this one was decided)
            below_sets_d -= cg_d[set], below_sets_b -= cg_b[set];
            long long in_d = 0, in_b = 0;
            for (int pos = first; pos <= hi; pos++) in_d += d_of(pos, lvl[pos]), in_b += pos_b[pos];
            bool stop = false;
            for (int pos = hi; pos >= first; pos--) {
                const long long d = d_of(pos, lvl[pos]);
                in_d -= d, in_b -= pos_b[pos];
                if (lvl[pos] > 0 && cg_flag[g]) {
                    const int pp = scan[pos], y = pp >> LOG2, x = pp - (y << LOG2);
                    // bits of the level alone: what the position cost, less the significance flag it carried
                    const int sig1 = pos == top ? 0 : t.sig[sig_ctx(pattern, scan_idx, x, y, LOG2, is_luma)][1];
                    const long long cd = above_d + d + in_d + below_sets_d;
                    const long long cb = (long long)pos_b[pos] - sig1 + in_b + below_sets_b + last_bits(t, x, y, scan_idx);
                    const long long cost = cd + rate_cost(lam_m, lam_s, cb);
                    if (best_last < 0 || cost < best_cost) best_last = pos, best_cost = cost, best_d = cd, best_b = cb;
                    if (lvl[pos] > 1) {
                        stop = true;
                        break;
                    }
                }
                above_d += d_of(pos, 0);
            }
            if (stop) break;
        }
    }
    // ---- the empty block ----
    const long long cost_coded = best_d + rate_cost(lam_m, lam_s, best_b + t.cbf[1]), cost_empty = d_all + rate_cost(lam_m, lam_s, t.cbf[0]);
    return best_last >= 0 && !(cost_empty < cost_coded) ? best_last : -1;
}

}  // namespace
