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
#include "rdoq_device.h"

namespace {

struct RdoqParams {
    const int16_t *coef;
    const uint8_t *qp, *flags;
    const xmi_rdoq_tables *tables;
    const xmi_rdoq_lambda *lambda;      // [52], by QP
    int16_t *level;
    uint32_t *sum;
    int count, scan_idx, is_luma;
};

template <int N>
__global__ void __launch_bounds__(64) rdoq_kernel(const RdoqParams p)
{
    constexpr int NN = N * N, LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5, SETS = (N / 4) * (N / 4);
    __shared__ uint32_t ld[NN], pos_b[NN];
    __shared__ int16_t lvl[NN];
    __shared__ uint16_t scan[NN];
    __shared__ uint8_t scan_cg[SETS], cg_flag[SETS];
    __shared__ int cg_b[SETS];
    __shared__ long long cg_d[SETS];
    __shared__ xmi_rdoq_tables t;
    __shared__ int s_top;
    __shared__ unsigned long long s_dall, s_dbeyond;
    const int lane = threadIdx.x;
    const size_t tu = blockIdx.x;
    const int16_t *coef = p.coef + tu * NN;
    int16_t *level = p.level + tu * NN;
    const int qp = p.qp[tu], scan_idx = p.scan_idx, is_luma = p.is_luma;
    const int qbits = 14 + qp / 6 + (15 - 8 - LOG2), q = k_rdoq_q6[qp % 6], eshift = qbits - 4;
    const uint32_t lam_m = p.lambda[qp].mantissa;
    const int lam_s = p.lambda[qp].shift;

    for (int i = lane; i < (int)(sizeof(xmi_rdoq_tables) / 4); i += 64) reinterpret_cast<int *>(&t)[i] = reinterpret_cast<const int *>(p.tables)[i];
    for (int i = lane; i < SETS; i += 64) scan_cg[i] = g_rdoq_scan_cg[scan_idx][LOG2 - 2][i], cg_flag[i] = 0;
    if (lane == 0) s_top = -1, s_dall = 0, s_dbeyond = 0;
    __syncthreads();
    // ---- all lanes: scaled magnitudes and rounded levels in scan order, the last of them, the distortion of zeros ----
    int top_l = -1;
    for (int i = lane; i < NN; i += 64) {
        const int pos = g_rdoq_scan[scan_idx][LOG2 - 2][i];
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
    const int best_last = rdoq_decide<N>([&](int i) { return ld[i]; }, scan, scan_cg, lvl, pos_b, cg_d, cg_b, cg_flag, t, top, (long long)s_dall,
                                         (long long)s_dbeyond, qbits, lam_m, lam_s, scan_idx, is_luma);
    uint32_t sum = 0;
    for (int i = 0; i <= best_last; i++) {
        const int l = lvl[i];
        if (!l) continue;
        const int pos = scan[i];
        level[pos] = (int16_t)(coef[pos] < 0 ? -l : l);
        sum += (uint32_t)l;
    }
    p.sum[tu] = sum;
}

bool size_ok(int n) { return n == 4 || n == 8 || n == 16 || n == 32; }

}  // namespace

// the scan tables of the device that is current: called once per device binding
int xmi::rdoq_prepare_device() { return rdoq_upload_scans(); }

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
