// rdoq.hip - rate-distortion optimised quantisation of transform units (the "RdoQ level decisions" of the hot path).
//
// No counterpart to match: the reference compiles its RdoQ.cpp out (#if RDOQ, src/RdoQ.cpp:32) and that file holds only the
// bit-cost table builders estimateBit / estCBFBit / estSignificantCoeffGroupMapBit / estSignificantMapBit /
// estSignificantCoefficientsBit (src/RdoQ.cpp:67-305) over estBitsSbacStruct (inc/RdoQ.h:38-49); the level decision it
// announces (My_xRateDistOptQuant, inc/RdoQ.h:55) is defined nowhere. xmi_rdoq_est_bits is those builders; the decision is
// the HEVC test model's, priced with their tables against the residual syntax the reference's coder writes
// (xEncodeCoeffNxN, src/encode.cpp:2385-2553): per coefficient the rounded level, one less, or zero; a 4x4 group dropped
// when its flag alone is cheaper; the last significant position moved back over levels of magnitude one; the empty block.
// Integer arithmetic throughout (costs = squared error in sixteenths of a level + ((mantissa * bits) >> shift)), so the CPU
// checker's statement of the same algorithm is reproduced bit for bit. Using it changes the stream: it is an extension,
// off unless asked for, and never part of a parity claim against the reference.
//
// Mapping: one wavefront per transform unit. The per-coefficient quantities (scaled magnitude, rounded level) are computed
// by all lanes into LDS; the decision itself is a chain through the coder's context state (c1, Rice parameter, the flags of
// the groups behind) and runs on one lane over LDS operands; groups without a rounded level are passed in one step.
#include "xmi_common.h"
#include "host/analysis_core.h"
#include "host/syntax.h"

namespace {

constexpr int kIep = 32768;
__device__ uint16_t g_scan[3][4][1024];     // coefficient scans [diagonal / horizontal / vertical][log2 size - 2]: positions y * size + x
__device__ uint8_t g_scan_cg[3][4][64];     // the same orders over the 4x4 groups
__constant__ int k_rdoq_q6[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };      // src/table.cpp:150
__constant__ uint8_t k_group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
__constant__ uint8_t k_sig_map4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

struct RdoqParams {
    const int16_t *coef;
    const uint8_t *qp, *flags;
    const xmi_rdoq_tables *tables;
    const xmi_rdoq_lambda *lambda;      // [52], by QP
    int16_t *level;
    uint32_t *sum;
    int count, scan_idx, is_luma;
};

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

template <int N>
__global__ void __launch_bounds__(64) rdoq_kernel(const RdoqParams p)
{
    constexpr int NN = N * N, LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5, SIDE = N / 4, SETS = SIDE * SIDE;
    __shared__ uint32_t ld[NN], pos_b[NN], lev_b[NN];
    __shared__ int16_t lvl[NN];
    __shared__ uint16_t scan[NN];
    __shared__ uint8_t scan_cg[SETS], cg_flag[SETS];
    __shared__ int cg_b[SETS];
    __shared__ long long cg_d[SETS];
    __shared__ xmi_rdoq_tables t;
    __shared__ int s_top;
    __shared__ unsigned long long s_dall, s_dbeyond;
    __shared__ uint32_t s_sum;
    const int lane = threadIdx.x;
    const size_t tu = blockIdx.x;
    const int16_t *coef = p.coef + tu * NN;
    int16_t *level = p.level + tu * NN;
    const int qp = p.qp[tu], scan_idx = p.scan_idx, is_luma = p.is_luma;
    const int qbits = 14 + qp / 6 + (15 - 8 - LOG2), q = k_rdoq_q6[qp % 6], eshift = qbits - 4;
    const uint32_t lam_m = p.lambda[qp].mantissa;
    const int lam_s = p.lambda[qp].shift;

    for (int i = lane; i < (int)(sizeof(xmi_rdoq_tables) / 4); i += 64) reinterpret_cast<int *>(&t)[i] = reinterpret_cast<const int *>(p.tables)[i];
    for (int i = lane; i < SETS; i += 64) scan_cg[i] = g_scan_cg[scan_idx][LOG2 - 2][i], cg_flag[i] = 0;
    if (lane == 0) s_top = -1, s_dall = 0, s_dbeyond = 0, s_sum = 0;
    __syncthreads();
    // ---- all lanes: scaled magnitudes and rounded levels in scan order, the last of them, the distortion of zeros ----
    int top_l = -1;
    for (int i = lane; i < NN; i += 64) {
        const int pos = g_scan[scan_idx][LOG2 - 2][i];
        scan[i] = (uint16_t)pos;
        const uint32_t v = (uint32_t)(abs((int)coef[pos]) * q);
        ld[i] = v;
        uint32_t mx = (uint32_t)(((unsigned long long)v + (1u << (qbits - 1))) >> qbits);
        if (mx > 32767u) mx = 32767u;
        lvl[i] = (int16_t)mx;
        if (mx) top_l = i;
        level[pos] = 0;
    }
    atomicMax(&s_top, top_l);
    __syncthreads();
    const int top = s_top;
    if (top < 0) {
        if (lane == 0) p.sum[tu] = 0;
        return;
    }
    {
        unsigned long long all0 = 0, beyond = 0;
        for (int i = lane; i < NN; i += 64) {
            const long long e = (long long)ld[i] >> eshift;
            all0 += (unsigned long long)(e * e);
            if (i > top) beyond += (unsigned long long)(e * e);
        }
        atomicAdd(&s_dall, all0);
        atomicAdd(&s_dbeyond, beyond);
    }
    __syncthreads();
    if (lane != 0) return;

    auto d_of = [&](int i, int l) -> long long {
        const long long e = ((long long)ld[i] - ((long long)l << qbits)) >> eshift;
        return e * e;
    };
    // ---- one lane: the levels, group by group from the last one down ----
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
            int best_l = 0, best_esc = 0, best_b = sig0, best_lb = 0;
            long long best_d = d0, best_cost = d0 + rate_cost(lam_m, lam_s, sig0);
            bool have = !is_last;
            for (int l = mx; l >= 1 && l >= mx - 1; l--) {
                int esc;
                const int lb = level_bits(t, l, c1, ctx_set, c1_idx, c2_idx, rice, esc);
                const long long d = d_of(pos, l), cost = d + rate_cost(lam_m, lam_s, (long long)sig1 + lb);
                if (!have || cost < best_cost) {
                    have = true;
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
        } else {
            cg_flag[g] = 1;
            cg_d[set] = s_d, cg_b[set] = (int)s_b;
        }
        tot_d += cg_d[set], tot_b += cg_b[set];
    }
    // ---- the last position: back over levels of magnitude one while that is cheaper ----
    int best_last = -1;
    long long best_d = 0, best_b = 0, best_cost = 0;
    {
        long long above_d = (long long)s_dbeyond, below_sets_d = tot_d, below_sets_b = tot_b;
        for (int set = last_set; set >= 0; set--) {
            const int first = set << 4, hi = set == last_set ? top : first + 15;
            const int g = scan_cg[set];
            below_sets_d -= cg_d[set], below_sets_b -= cg_b[set];
            long long in_d = 0, in_b = 0;
            for (int pos = first; pos <= hi; pos++) in_d += d_of(pos, lvl[pos]), in_b += pos_b[pos];
            bool stop = false;
            for (int pos = hi; pos >= first; pos--) {
                const long long d = d_of(pos, lvl[pos]);
                in_d -= d, in_b -= pos_b[pos];
                if (lvl[pos] > 0 && cg_flag[g]) {
                    const int pp = scan[pos], y = pp >> LOG2, x = pp - (y << LOG2);
                    const long long cd = above_d + d + in_d + below_sets_d;
                    const long long cb = (long long)lev_b[pos] + in_b + below_sets_b + last_bits(t, x, y, scan_idx);
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
    const long long cost_coded = best_d + rate_cost(lam_m, lam_s, best_b + t.cbf[1]), cost_empty = (long long)s_dall + rate_cost(lam_m, lam_s, t.cbf[0]);
    uint32_t sum = 0;
    if (best_last >= 0 && !(cost_empty < cost_coded)) {
        for (int i = 0; i <= best_last; i++) {
            const int l = lvl[i];
            if (!l) continue;
            const int pos = scan[i];
            level[pos] = (int16_t)(coef[pos] < 0 ? -l : l);
            sum += (uint32_t)l;
        }
    }
    p.sum[tu] = sum;
}

bool size_ok(int n) { return n == 4 || n == 8 || n == 16 || n == 32; }

}  // namespace

// the scan tables of the device that is current: called once per device binding
int xmi::rdoq_prepare_device()
{
    static uint16_t scan[3][4][1024];
    static uint8_t groups[3][4][64];
    for (int k = 0; k < 3; k++)
        for (int l = 2; l <= 5; l++) {
            const int n = 1 << l;
            memcpy(scan[k][l - 2], xmi::syntax::coef_scan(k, l), (size_t)n * n * 2);
            memcpy(groups[k][l - 2], xmi::syntax::group_scan(k, l), (size_t)(n / 4) * (n / 4));
        }
    XMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scan), scan, sizeof scan));
    XMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scan_cg), groups, sizeof groups));
    return XMI_OK;
}

extern "C" {

// estimateBit (src/RdoQ.cpp:67-305) for one block size and component from a set of context states in the reference's layout
// (inc/x265.h:286-311). Host arithmetic; the result goes to the device as one of the batch call's arguments.
int xmi_rdoq_est_bits(const uint8_t *ctx, int log2_size, int is_luma, xmi_rdoq_tables *out)
{
    using namespace xmi::syntax;
    XMI_ARG(ctx && out && log2_size >= 2 && log2_size <= 5, "xmi_rdoq_est_bits: null argument or log2 size %d outside 2..5", log2_size);
    const int32_t *ent = xmi::analysis::entropy_bits_table();
    memset(out, 0, sizeof *out);
    const bool luma = is_luma != 0;
    for (int c = 0; c < kNumSigCg; c++)
        for (int b = 0; b < 2; b++) out->sig_cg[c][b] = ent[ctx[kCtxSigCg + c + (luma ? 0 : kNumSigCg)] ^ b];
    for (int c = 0; c < (luma ? kNumSigLuma : 15); c++)
        for (int b = 0; b < 2; b++) out->sig[c][b] = ent[ctx[kCtxSig + (luma ? 0 : kNumSigLuma) + c] ^ b];
    static const uint8_t group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
    const int off = luma ? (log2_size - 2) * 3 + ((log2_size - 1) >> 2) : 0, sh = luma ? (log2_size + 1) >> 2 : log2_size - 2;
    const int gmax = group_idx[(1 << log2_size) - 1];
    for (int xy = 0; xy < 2; xy++) {
        int32_t *dst = xy ? out->last_y : out->last_x;
        const int base = (xy ? kCtxLastY : kCtxLastX) + (luma ? 0 : kNumLast);
        int32_t bits = 0;
        int c = 0;
        for (; c < gmax; c++) {
            const uint8_t s = ctx[base + off + (c >> sh)];
            dst[c] = bits + ent[s];
            bits += ent[s ^ 1];
        }
        dst[c] = bits;
    }
    for (int c = 0; c < (luma ? kNumOneLuma : 8); c++)
        for (int b = 0; b < 2; b++) out->one[c][b] = ent[ctx[kCtxOne + (luma ? 0 : kNumOneLuma) + c] ^ b];
    for (int c = 0; c < (luma ? kNumAbsLuma : 2); c++)
        for (int b = 0; b < 2; b++) out->abs[c][b] = ent[ctx[kCtxAbs + (luma ? 0 : kNumAbsLuma) + c] ^ b];
    for (int b = 0; b < 2; b++) out->cbf[b] = ent[ctx[kCtxQtCbf + (luma ? 0 : kNumQtCbf)] ^ b];
    return XMI_OK;
}

// The contexts a slice starts from (xCabacInit, src/bitstream.cpp:588): the state the tables are usually built from.
int xmi_rdoq_initial_contexts(int slice_type, int qp, uint8_t ctx[256])
{
    XMI_ARG(ctx && (slice_type == XMI_SLICE_P || slice_type == XMI_SLICE_I) && qp >= 0 && qp <= 51, "xmi_rdoq_initial_contexts: bad argument");
    xmi::analysis::init_contexts(slice_type, qp, ctx);
    return XMI_OK;
}

// The weight of a bit against squared error in sixteenths of a level, as mantissa and shift (see the file header):
// lambda * 256 / 32768 * (Q * 2^tshift / 2^qbits)^2 with the quantiser constants of xQuant (src/pixel.cpp:1060-1089).
int xmi_rdoq_lambda_of(double lambda, int qp, int log2_size, xmi_rdoq_lambda *out)
{
    XMI_ARG(out && qp >= 0 && qp <= 51 && log2_size >= 2 && log2_size <= 5 && lambda >= 0, "xmi_rdoq_lambda_of: bad argument");
    static const int q6[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
    const int qbits = 14 + qp / 6 + (15 - 8 - log2_size), tshift = 15 - 8 - log2_size;
    double v = lambda * 256.0 / 32768.0;
    const double g = (double)q6[qp % 6];
    v *= g * g;
    for (int i = 0; i < 2 * tshift; i++) v *= 2.0;
    for (int i = 0; i < 2 * qbits; i++) v *= 0.5;
    int s = 0;
    out->mantissa = 0, out->shift = 0;
    if (v <= 0) return XMI_OK;
    while (v < (double)(1 << 29) && s < 60) v *= 2.0, s++;
    while (v >= (double)(1 << 30) && s > 0) v *= 0.5, s--;
    if (v >= (double)(1 << 30)) v = (double)((1 << 30) - 1);
    uint32_t m = (uint32_t)(v + 0.5);
    if (m >= (1u << 30)) m = (1u << 30) - 1;
    out->mantissa = m, out->shift = s;
    return XMI_OK;
}

int xmi_rdoq_quant_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_qp, const uint8_t *d_flags, const xmi_rdoq_tables *d_tables,
                             const xmi_rdoq_lambda *d_lambda, int scan_idx, int is_luma, int16_t *d_level, uint32_t *d_sum, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(size_ok(n), "xmi_rdoq_quant_batch_dev: size %d not in {4,8,16,32}", n);
    XMI_ARG(count >= 0 && d_coef && d_qp && d_tables && d_lambda && d_level && d_sum, "xmi_rdoq_quant_batch_dev: null argument or negative count");
    XMI_ARG(scan_idx >= 0 && scan_idx <= 2, "xmi_rdoq_quant_batch_dev: scan %d (0 diagonal, 1 horizontal, 2 vertical)", scan_idx);
    XMI_ARG((((uintptr_t)d_tables | (uintptr_t)d_lambda) & 3) == 0, "xmi_rdoq_quant_batch_dev: tables must be 4-byte aligned");
    (void)d_flags;      // the slice type only sets the rounding of the plain quantiser; kept for a signature like xmi_quant_batch_dev's
    if (count == 0) return XMI_OK;
    const RdoqParams p = { d_coef, d_qp, d_flags, d_tables, d_lambda, d_level, d_sum, count, scan_idx, is_luma != 0 };
    hipStream_t st = xmi::as_stream(stream);
    switch (n) {
    case 4: hipLaunchKernelGGL(rdoq_kernel<4>, dim3(count), dim3(64), 0, st, p); break;
    case 8: hipLaunchKernelGGL(rdoq_kernel<8>, dim3(count), dim3(64), 0, st, p); break;
    case 16: hipLaunchKernelGGL(rdoq_kernel<16>, dim3(count), dim3(64), 0, st, p); break;
    default: hipLaunchKernelGGL(rdoq_kernel<32>, dim3(count), dim3(64), 0, st, p); break;
    }
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

// xSubDctQuantRec (src/encode.cpp:3005-3021) with the level decision in the quantiser's place: xSubDct -> decision -> xDeQuant
// -> xIDctAdd on the caller's stream. d_work: count * n * n int16 the call may overwrite (coefficients, then dequantised values).
int xmi_tu_chain_rdoq_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred, const uint8_t *d_qp, const uint8_t *d_flags,
                                const xmi_rdoq_tables *d_tables, const xmi_rdoq_lambda *d_lambda, int scan_idx, int is_luma, int16_t *d_work,
                                int16_t *d_level, uint8_t *d_rec, uint32_t *d_sum, void *stream)
{
    XMI_ARG(d_work, "xmi_tu_chain_rdoq_batch_dev: null work buffer");
    int rc = xmi_sub_dct_batch_dev(n, count, d_pix, d_pred, d_flags, d_work, stream);
    if (rc != XMI_OK) return rc;
    rc = xmi_rdoq_quant_batch_dev(n, count, d_work, d_qp, d_flags, d_tables, d_lambda, scan_idx, is_luma, d_level, d_sum, stream);
    if (rc != XMI_OK) return rc;
    rc = xmi_dequant_batch_dev(n, count, d_level, d_qp, d_work, stream);
    if (rc != XMI_OK) return rc;
    return xmi_idct_add_batch_dev(n, count, d_work, d_pred, d_flags, d_rec, stream);
}

}  // extern "C"
