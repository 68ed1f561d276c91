// Batched intra luma mode search for gfx950: the 35-mode loop of xEncDecideCU_Intra (src/encode.cpp:4005-4081).
//
// One wavefront per PU request. The wavefront stages the neighbouring samples in LDS, pads them (xPaddingRef,
// src/encode.cpp:3843), smooths them (xFilterRef, :3873) and stages the source block. The 35 predictions are
// never written anywhere: a lane owns one transform tile of one mode (4x4 tiles for the x264-style SATD bound
// for sizes 4..16, 8x8 tiles for xCalcHADs32x32_test at 32), evaluates xEncIntraPred (src/pixel.cpp:7338:
// planar :6489, DC :6704, angular :7243 incl. the edge filters and the transpose of the horizontal modes) pixel
// by pixel for that tile straight from the LDS samples, and runs the Hadamard transform of the residual in
// registers. 64 / tiles-per-block modes are evaluated at once; tile sums meet through a segmented DPP reduction.
// Cost = satd + {1,2,3} * sqrt_lambda in double, strict < in mode order, as the reference's loop does.
//
// xmi_intra_pred_batch_dev / xmi_enc_intra_pred write the prediction block itself (xEncIntraPred, one mode) with the
// same per-sample code: one wavefront per request.
#include "xmi_common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace {

typedef unsigned int u32;

__constant__ int8_t k_angle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                    -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };   // src/table.cpp:43
__constant__ int16_t k_inv_angle[35] = { 0, 0, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315,
                                         256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };   // :53
// xg_aucIntraFilterType (src/table.cpp:32) as one bit per mode, sizes 4, 8, 16, 32
__constant__ unsigned long long k_filter_bits[4] = {
    0ull,
    (1ull << 0) | (1ull << 2) | (1ull << 18) | (1ull << 34),
    0x7FFFFFFFFull & ~((1ull << 1) | (1ull << 9) | (1ull << 10) | (1ull << 11) | (1ull << 25) | (1ull << 26) | (1ull << 27)),
    0x7FFFFFFFFull & ~((1ull << 1) | (1ull << 10) | (1ull << 26)),
};

struct IntraParams {
    const uint8_t *src;
    const uint8_t *refs;
    const xmi_intra_request *req;
    xmi_intra_result *res;
    int src_stride, n;
};

template <int W>
__device__ __forceinline__ int group_sum_to_last_lane(int v)
{
    if (W >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    if (W >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    if (W >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    if (W >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    return v;
}

// one prediction sample of xEncIntraPred(mode, size, is_luma) at row y, column x; ref = sample array, tl = ref + 2*size
template <int N>
__device__ __forceinline__ int pred_sample(const uint8_t *ref, int mode, int y, int x, int dc, bool is_luma)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    const uint8_t *tl = ref + 2 * N;
    if (mode == 0) {         // planar: left[i] = tl[-1-i], top[j] = tl[1+j]
        const int l = tl[-1 - y], t = tl[1 + x], bl = tl[-1 - N], tr = tl[1 + N];
        return ((N - 1 - x) * l + (x + 1) * tr + (N - 1 - y) * t + (y + 1) * bl + N) >> (LG + 1);
    }
    if (mode == 1) {
        if (!is_luma || (x > 0 && y > 0)) return dc;
        if (x == 0 && y == 0) return (tl[1] + tl[-1] + 2 * dc + 2) >> 2;
        return ((y == 0 ? tl[1 + x] : tl[-1 - y]) + 3 * dc + 2) >> 2;
    }
    const bool hor = mode < 18;
    const int angle = k_angle[mode], inv = k_inv_angle[mode];
    const int k = hor ? x : y, xx = hor ? y : x;      // row / column before the transpose of the horizontal modes
    const int delta = (k + 1) * angle, idx = delta >> 5, fact = delta & 31;
    auto main_ref = [&](int p) {
        if (p < 0) p = -((128 - p * inv) >> 8);       // projected side samples
        return (int)tl[hor ? -p : p];
    };
    const int pos = xx + idx + 1;
    int v = main_ref(pos);
    if (fact) v = ((32 - fact) * v + fact * main_ref(pos + 1) + 16) >> 5;
    if (is_luma && angle == 0 && xx == 0) v = min(255, max(0, v + (((int)tl[hor ? k + 1 : -(k + 1)] - (int)tl[0]) >> 1)));
    return v;
}

// WPR wavefronts work on one request: each prepares the samples for itself and takes every WPR-th pass of modes;
// s_best (WPR entries of cost, mode) joins them at the end.
template <int N, int WPR>
__device__ void search(const IntraParams &p, const xmi_intra_request &rq, int idx, uint8_t *s_ref, uint8_t *s_src, int lane, int sub,
                       double *s_best_cost, int *s_best_mode)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    constexpr int L = 4 * N + 1;
    constexpr int TS = N == 32 ? 8 : 4;                 // transform tile size of the SATD
    constexpr int T = (N / TS) * (N / TS);              // tiles per block
    constexpr int MP = 64 / T;                          // modes per pass
    uint8_t *ref0 = s_ref, *ref1 = s_ref + 136;
    const unsigned valids = rq.valids & 0x1fu;

    // ---- neighbouring samples: load, pad, smooth ----
    const uint8_t *raw = p.refs + (size_t)idx * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < L; i += 64) ref0[i] = valids ? raw[i] : (uint8_t)0x80;
    __builtin_amdgcn_wave_barrier();
    if (valids) {
        const int off[6] = { 0, N, 2 * N, 2 * N + 1, 3 * N + 1, 4 * N + 1 };
        int n0 = 0;
        while (!(valids & (1u << n0))) n0++;
        {
            const uint8_t pad = ref0[off[n0]];
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < off[n0]; i += 64) ref0[i] = pad;
            __builtin_amdgcn_wave_barrier();
        }
        for (int s = n0; s < 5; s++)
            if (!(valids & (1u << s))) {
                const uint8_t pad = ref0[off[s] - 1];
                __builtin_amdgcn_wave_barrier();
                for (int i = off[s] + lane; i < off[s + 1]; i += 64) ref0[i] = pad;
                __builtin_amdgcn_wave_barrier();
            }
        const int bl = ref0[0], tlv = ref0[2 * N], tr = ref0[4 * N];
        const bool bilinear = rq.strong_smoothing && N == 32 && abs(bl + tlv - 2 * ref0[N]) < 8 && abs(tlv + tr - 2 * ref0[3 * N]) < 8;
        for (int i = lane; i < L; i += 64) {
            int v;
            if (i == 0 || i == 4 * N || (bilinear && i == 2 * N)) v = ref0[i];
            else if (bilinear) v = i < 2 * N ? ((2 * N - i) * bl + i * tlv + N) >> (LG + 1) : ((4 * N - i) * tlv + (i - 2 * N) * tr + N) >> (LG + 1);
            else v = (ref0[i - 1] + 2 * ref0[i] + ref0[i + 1] + 2) >> 2;
            ref1[i] = (uint8_t)v;
        }
    } else {
        for (int i = lane; i < L; i += 64) ref1[i] = 0x80;
    }
    // ---- source block ----
    const uint8_t *src = p.src + (size_t)rq.y0 * p.src_stride + rq.x0;
    for (int e = lane; e < N * N / 4; e += 64) {
        const int y = e / (N / 4), x4 = e % (N / 4);
        reinterpret_cast<u32 *>(s_src)[e] = *reinterpret_cast<const u32 *>(src + (size_t)y * p.src_stride + 4 * x4);
    }
    __builtin_amdgcn_wave_barrier();

    // DC value of the unfiltered samples (mode 1 never uses the smoothed set): (sum top + sum left + N) / 2N
    int dc;
    {
        int s = 0;
        for (int i = lane; i < 2 * N; i += 64) s += i < N ? ref0[2 * N - 1 - i] : ref0[2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) s += __shfl_xor(s, k);
        dc = (s + N) >> (LG + 1);
    }

    const bool is_luma = N != 32;
    const unsigned long long fbits = k_filter_bits[LG - 2];
    const int tile = lane % T, ty = (tile / (N / TS)) * TS, tx = (tile % (N / TS)) * TS;
    double best_cost = 1.7976931348623157e308;
    int best_mode = 64;
    xmi_intra_result *out = p.res + idx;
    for (int m0 = sub * MP; m0 < 35; m0 += MP * WPR) {
        const int mode = m0 + lane / T;
        int sum = 0;
        if (mode < 35) {
            const uint8_t *ref = ((fbits >> mode) & 1) ? ref1 : ref0;
            xmi::s16x2 d[TS][TS / 2];       // residual as pairs of neighbouring columns
#pragma unroll
            for (int y = 0; y < TS; y++)
#pragma unroll
                for (int x = 0; x < TS; x += 2) {
                    d[y][x / 2].x = (short)((int)s_src[(ty + y) * N + tx + x] - pred_sample<N>(ref, mode, ty + y, tx + x, dc, is_luma));
                    d[y][x / 2].y = (short)((int)s_src[(ty + y) * N + tx + x + 1] - pred_sample<N>(ref, mode, ty + y, tx + x + 1, dc, is_luma));
                }
            sum = (int)xmi::hadamard_abs_sum_packed<TS>(d);
            if (TS == 8) sum = (sum + 2) >> 2;          // xCalcHADs8x8 rounding, per tile
        }
        sum = group_sum_to_last_lane<T>(sum);
        if (TS == 4) sum >>= 1;                          // x264 form: halved once per block
        if (mode < 35 && tile == T - 1) {
            out->satd[mode] = (u32)sum;
            const double bits = mode == rq.mpm[0] ? 1.0 : (mode == rq.mpm[1] || mode == rq.mpm[2]) ? 2.0 : 3.0;
            const double cost = bits * rq.sqrt_lambda + (double)(u32)sum;
            if (cost < best_cost) {
                best_cost = cost;
                best_mode = mode;
            }
        }
    }
    // first minimum in mode order over the lanes that hold results
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const double oc = __shfl_xor(best_cost, k);
        const int om = __shfl_xor(best_mode, k);
        if (oc < best_cost || (oc == best_cost && om < best_mode)) {
            best_cost = oc;
            best_mode = om;
        }
    }
    if (WPR > 1) {
        if (lane == 0) {
            s_best_cost[sub] = best_cost;
            s_best_mode[sub] = best_mode;
        }
        __syncthreads();
        if (sub != 0) return;
#pragma unroll
        for (int w = 1; w < WPR; w++) {
            const double oc = s_best_cost[w];
            const int om = s_best_mode[w];
            if (oc < best_cost || (oc == best_cost && om < best_mode)) {
                best_cost = oc;
                best_mode = om;
            }
        }
    }
    if (lane == 0) {
        int m[3] = { rq.mpm[0], rq.mpm[1], rq.mpm[2] };
        int pi;
        if (best_mode == m[0]) pi = 0;
        else if (best_mode == m[1]) pi = 1;
        else if (best_mode == m[2]) pi = 2;
        else {       // src/encode.cpp:4059-4080
            int t;
            if (m[0] > m[1]) { t = m[0]; m[0] = m[1]; m[1] = t; }
            if (m[0] > m[2]) { t = m[0]; m[0] = m[2]; m[2] = t; }
            if (m[1] > m[2]) { t = m[1]; m[1] = m[2]; m[2] = t; }
            pi = best_mode;
            pi -= pi > m[2];
            pi -= pi > m[1];
            pi -= pi > m[0];
            pi += 3;
        }
        out->best_mode = best_mode;
        out->pred_idx = pi;
        out->best_cost = best_cost;
        out->reserved = 0;
    }
}

// One kernel per PU size, each over the whole request array (wavefronts of other sizes leave at once): the
// register budget - and with it the number of resident wavefronts that hide the LDS and memory latency of this
// serial per-PU work - is then set by the size's own needs instead of by the 32x32 instance.
template <int N>
__global__ void __launch_bounds__(256) intra_search_kernel(const IntraParams p)
{
    constexpr int WPR = 1;       // wavefronts per request (4 was tried for 32x32 and lost: the unrolled code streams through the instruction cache either way)
    __shared__ __align__(16) uint8_t s_ref[4][272];
    __shared__ __align__(16) uint8_t s_src[4][N * N];
    __shared__ double s_best_cost[4];
    __shared__ int s_best_mode[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = WPR == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_intra_request rq = p.req[idx];
    if (rq.size != N) return;                   // another instance's size, or no PU size at all (then no result is written)
    search<N, WPR>(p, rq, idx, s_ref[wave], s_src[wave], lane, WPR == 4 ? wave : 0, s_best_cost, s_best_mode);
}

// ---- prediction blocks: one wavefront per request ----
struct IntraPredParams {
    const uint8_t *refs;
    const xmi_intra_pred_request *req;
    uint8_t *pred;
    int n;
};

template <int N>
__device__ __forceinline__ void predict(const uint8_t *s_ref, uint8_t *dst, int mode, bool is_luma, int lane)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    int dc;
    {
        int s = 0;
        for (int i = lane; i < 2 * N; i += 64) s += i < N ? s_ref[2 * N - 1 - i] : s_ref[2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) s += __shfl_xor(s, k);
        dc = (s + N) >> (LG + 1);
    }
    for (int e = lane; e < N * N; e += 64) dst[e] = (uint8_t)pred_sample<N>(s_ref, mode, e / N, e % N, dc, is_luma);
}

__global__ void __launch_bounds__(256) intra_pred_kernel(const IntraPredParams p)
{
    __shared__ __align__(16) uint8_t s_ref[4][136];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_intra_pred_request rq = p.req[idx];
    if ((rq.size != 4 && rq.size != 8 && rq.size != 16 && rq.size != 32) || rq.mode > 34) return;   // no PU size or no mode: nothing is written
    const uint8_t *raw = p.refs + (size_t)idx * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < 4 * rq.size + 1; i += 64) s_ref[wave][i] = raw[i];
    __builtin_amdgcn_wave_barrier();
    uint8_t *dst = p.pred + rq.out;
    const bool is_luma = rq.is_luma != 0;
    switch (rq.size) {
    case 4: predict<4>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    case 8: predict<8>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    case 16: predict<16>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    default: predict<32>(s_ref[wave], dst, rq.mode, is_luma, lane); break;
    }
}

// ---- intra chroma mode search: one wavefront per CU ----
// xEncodeChromaCU (src/encode.cpp:4326-4365): candidates planar, vertical, horizontal, DC and the luma mode (one of the
// first four that equals the luma mode becomes 34), U and V predicted without the luma edge filters, each measured with
// the x264-form SATD the reference binds for 4x4 .. 16x16, first smallest U + V sum kept. The 10 (candidate, plane)
// blocks are cut into 4x4 transform tiles; a lane takes one tile per pass, tile sums meet through the segmented DPP
// reduction of the luma search.
struct ChromaParams {
    const uint8_t *src_u, *src_v, *refs;
    const xmi_chroma_request *req;
    xmi_chroma_result *res;
    int src_stride, n;
};

template <int N>
__device__ __forceinline__ void chroma_search(const ChromaParams &p, const xmi_chroma_request &rq, int idx, uint8_t (*s_ref)[136], uint8_t (*s_src)[256],
                                              u32 *s_satd, int lane)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : 4;
    constexpr int T = (N / 4) * (N / 4);             // 4x4 tiles per block
    const uint8_t *raw = p.refs + (size_t)idx * 2 * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < 2 * (4 * N + 1); i += 64) {
        const int pl = i >= 4 * N + 1, k = i - pl * (4 * N + 1);
        s_ref[pl][k] = raw[pl * XMI_INTRA_REF_PITCH + k];
    }
    for (int e = lane; e < 2 * N * N / 4; e += 64) {
        const int pl = e >= N * N / 4, k = e - pl * (N * N / 4), y = k / (N / 4), x4 = k % (N / 4);
        const uint8_t *src = (pl ? p.src_v : p.src_u) + (size_t)(rq.y0 + y) * p.src_stride + rq.x0 + 4 * x4;
        reinterpret_cast<u32 *>(s_src[pl])[k] = *reinterpret_cast<const u32 *>(src);
    }
    __builtin_amdgcn_wave_barrier();
    int dc[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        int sum = 0;
        for (int i = lane; i < 2 * N; i += 64) sum += i < N ? s_ref[pl][2 * N - 1 - i] : s_ref[pl][2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) sum += __shfl_xor(sum, k);
        dc[pl] = (sum + N) >> (LG + 1);
    }
    const int luma = rq.luma_mode;
    const int cand[5] = { 0, 26, 10, 1, luma };
    for (int item0 = 0; item0 < 10 * T; item0 += 64) {
        const int item = item0 + lane, combo = item / T, tile = item % T;
        int sum = 0;
        if (combo < 10) {
            const int m = combo >> 1, pl = combo & 1;
            int mode = cand[m];
            if (m < 4 && mode == luma) mode = 34;
            const int ty = (tile / (N / 4)) * 4, tx = (tile % (N / 4)) * 4;
            xmi::s16x2 d[4][2];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x += 2) {
                    d[y][x / 2].x = (short)((int)s_src[pl][(ty + y) * N + tx + x] - pred_sample<N>(s_ref[pl], mode, ty + y, tx + x, dc[pl], false));
                    d[y][x / 2].y = (short)((int)s_src[pl][(ty + y) * N + tx + x + 1] - pred_sample<N>(s_ref[pl], mode, ty + y, tx + x + 1, dc[pl], false));
                }
            sum = (int)xmi::hadamard_abs_sum_packed<4>(d);
        }
        sum = group_sum_to_last_lane<T>(sum);
        if (combo < 10 && tile == T - 1) s_satd[combo] = (u32)sum >> 1;      // x264 form: halved once per block
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        xmi_chroma_result out;
        u32 best = 0xffffffffu;
        out.best_idx = 0;
        for (int m = 0; m < 5; m++) {
            out.satd_u[m] = s_satd[2 * m];
            out.satd_v[m] = s_satd[2 * m + 1];
            const u32 both = out.satd_u[m] + out.satd_v[m];
            if (both < best) {
                best = both;
                out.best_idx = m;
            }
        }
        const int bm = cand[out.best_idx];
        out.best_mode = (out.best_idx < 4 && bm == luma) ? 34 : bm;
        p.res[idx] = out;
    }
}

__global__ void __launch_bounds__(256) intra_chroma_search_kernel(const ChromaParams p)
{
    __shared__ __align__(16) uint8_t s_ref[4][2][136];
    __shared__ __align__(16) uint8_t s_src[4][2][256];
    __shared__ u32 s_satd[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = (int)blockIdx.x * 4 + wave;
    if (idx >= p.n) return;
    const xmi_chroma_request rq = p.req[idx];
    const bool ok = (rq.size == 4 || rq.size == 8 || rq.size == 16) && rq.luma_mode < 35 && rq.x0 >= 0 && rq.y0 >= 0 && (rq.x0 & 3) == 0;
    if (!ok) {                   // marked, so that the caller can tell
        if (lane == 0) {
            xmi_chroma_result out = {};
            out.best_idx = out.best_mode = -1;
            p.res[idx] = out;
        }
        return;
    }
    switch (rq.size) {
    case 4: chroma_search<4>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    case 8: chroma_search<8>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    default: chroma_search<16>(p, rq, idx, s_ref[wave], s_src[wave], s_satd[wave], lane); break;
    }
}

std::mutex g_pred_mu;   // the per-call form shares one staging area

// one kernel per PU size present in `sizes` (bit mask of 4, 8, 16, 32) over the whole request array
int launch_search(const uint8_t *d_src, int src_stride, const uint8_t *d_ref_samples, const xmi_intra_request *d_req, int n,
                  xmi_intra_result *d_res, int sizes, void *stream)
{
    IntraParams p = { d_src, d_ref_samples, d_req, d_res, src_stride, n };
    const dim3 grid((n + 3) / 4), block(256);
    if (sizes & 4) hipLaunchKernelGGL(intra_search_kernel<4>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 8) hipLaunchKernelGGL(intra_search_kernel<8>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 16) hipLaunchKernelGGL(intra_search_kernel<16>, grid, block, 0, xmi::as_stream(stream), p);
    if (sizes & 32) hipLaunchKernelGGL(intra_search_kernel<32>, grid, block, 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

}  // namespace

// ---- neighbour gather -------------------------------------------------------------------------------------------
// One wavefront per PU: the 4 size + 1 neighbouring samples in the order of src/encode.cpp:4010-4015 (left column from
// the bottom up, top-left, top row, top-right) out of the reconstruction plane, and the five availability bits of
// :3965-3974 out of the map of reconstructed 4x4 units. Samples of a segment that is not available are set to 0x80 (the
// reference leaves whatever its cache holds there and xPaddingRef overwrites them).
namespace {
struct GatherParams {
    const uint8_t *rec;
    const uint8_t *coded;
    xmi_intra_request *req;
    uint8_t *refs;
    int stride, width, height, coded_stride, n;
};

__global__ void __launch_bounds__(256) intra_gather_kernel(const GatherParams p)
{
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= p.n) return;
    const xmi_intra_request rq = p.req[idx];
    const int n = rq.size, x0 = rq.x0, y0 = rq.y0;
    auto unit = [&](int x, int y) -> bool {       // is the 4x4 unit that holds sample (x, y) reconstructed?
        return x >= 0 && y >= 0 && x < p.width && y < p.height && p.coded[(size_t)(y >> 2) * p.coded_stride + (x >> 2)] != 0;
    };
    const bool lb = unit(x0 - 1, y0 + n), l = unit(x0 - 1, y0), lt = unit(x0 - 1, y0 - 1), t = unit(x0, y0 - 1), tr = unit(x0 + n, y0 - 1);
    uint8_t *out = p.refs + (size_t)idx * XMI_INTRA_REF_PITCH;
    for (int i = lane; i < 4 * n + 1; i += 64) {
        int x, y;
        bool ok;
        if (i < 2 * n) {
            x = x0 - 1, y = y0 + 2 * n - 1 - i;
            ok = i < n ? lb : l;
        } else {
            x = x0 - 1 + (i - 2 * n), y = y0 - 1;
            ok = i == 2 * n ? lt : i <= 3 * n ? t : tr;
        }
        // a segment is flagged by its first unit; the part of it that leaves the picture reads as 0x80
        ok = ok && x >= 0 && y >= 0 && x < p.width && y < p.height;
        out[i] = ok ? p.rec[(size_t)y * p.stride + x] : (uint8_t)0x80;
    }
    if (lane == 0) p.req[idx].valids = (uint8_t)((lb ? 1 : 0) | (l ? 2 : 0) | (lt ? 4 : 0) | (t ? 8 : 0) | (tr ? 16 : 0));
}
}  // namespace

extern "C" int xmi_intra_gather_dev(const uint8_t *d_rec, int rec_stride, int width, int height, const uint8_t *d_coded, int coded_stride,
                                    xmi_intra_request *d_req, int n, uint8_t *d_ref_samples, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_gather_dev: negative count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_rec && d_coded && d_req && d_ref_samples, "xmi_intra_gather_dev: null argument");
    XMI_ARG(width > 0 && height > 0 && width % 4 == 0 && height % 4 == 0 && rec_stride >= width && coded_stride >= width / 4,
            "xmi_intra_gather_dev: %dx%d, strides %d / %d", width, height, rec_stride, coded_stride);
    GatherParams p = { d_rec, d_coded, d_req, d_ref_samples, rec_stride, width, height, coded_stride, n };
    hipLaunchKernelGGL(intra_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_intra_pred_batch_dev(const uint8_t *d_ref_samples, const xmi_intra_pred_request *d_req, int n, uint8_t *d_pred, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_pred_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_ref_samples && d_req && d_pred, "xmi_intra_pred_batch_dev: null argument");
    IntraPredParams p = { d_ref_samples, d_req, d_pred, n };
    hipLaunchKernelGGL(intra_pred_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_enc_intra_pred(uint8_t *dst, const uint8_t *ref, uint32_t stride, uint32_t mode, uint32_t size, uint32_t is_luma)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(dst && ref, "xmi_enc_intra_pred: null argument");
    XMI_ARG(size == 4 || size == 8 || size == 16 || size == 32, "xmi_enc_intra_pred: size %u not in {4,8,16,32}", size);
    XMI_ARG(mode < 35, "xmi_enc_intra_pred: mode %u > 34", mode);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    const size_t rq_off = 144, out_off = 160, blk = (size_t)size * size;
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, out_off + 1024));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[160];
    memset(host, 0, sizeof host);
    memcpy(host, ref, 4 * size + 1);
    xmi_intra_pred_request rq = { (uint8_t)size, (uint8_t)mode, (uint8_t)(is_luma ? 1 : 0), 0, 0 };
    memcpy(host + rq_off, &rq, sizeof rq);
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    IntraPredParams p = { dev, reinterpret_cast<const xmi_intra_pred_request *>(dev + rq_off), dev + out_off, 1 };
    hipLaunchKernelGGL(intra_pred_kernel, dim3(1), dim3(256), 0, nullptr, p);
    XMI_HIP(hipGetLastError());
    uint8_t out[1024];
    XMI_HIP(hipMemcpy(out, dev + out_off, blk, hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < size; y++) memcpy(dst + (size_t)y * stride, out + (size_t)y * size, size);
    return XMI_OK;
}

extern "C" int xmi_intra_search(const uint8_t *src, uint32_t src_stride, const uint8_t *ref_samples, const xmi_intra_request *req,
                                xmi_intra_result *res)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(src && ref_samples && req && res, "xmi_intra_search: null argument");
    const int n = req->size;
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32, "xmi_intra_search: size %d not in {4,8,16,32}", n);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    // staging: source block packed (rows n apart) | samples | request | result
    const size_t off_ref = 1024, off_req = off_ref + 144, off_res = off_req + 32, total = off_res + sizeof(xmi_intra_result);
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, 2048));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[off_res];
    memset(host, 0, sizeof host);
    for (int y = 0; y < n; y++) memcpy(host + (size_t)y * n, src + (size_t)y * src_stride, (size_t)n);
    memcpy(host + off_ref, ref_samples, (size_t)(4 * n + 1));
    xmi_intra_request rq = *req;
    rq.x0 = rq.y0 = 0;
    memcpy(host + off_req, &rq, sizeof rq);
    static_assert(sizeof(xmi_intra_request) <= 32 && 1024 + 144 + 32 + sizeof(xmi_intra_result) <= 2048, "staging layout");
    (void)total;
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    rc = launch_search(dev, n, dev + off_ref, reinterpret_cast<const xmi_intra_request *>(dev + off_req), 1,
                       reinterpret_cast<xmi_intra_result *>(dev + off_res), n, nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(res, dev + off_res, sizeof(xmi_intra_result), hipMemcpyDeviceToHost));
    return XMI_OK;
}

extern "C" int xmi_intra_search_batch_dev(const uint8_t *d_src, int src_stride, const uint8_t *d_ref_samples,
                                          const xmi_intra_request *d_req, int n, xmi_intra_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_src && d_ref_samples && d_req && d_res, "xmi_intra_search_batch_dev: null argument");
    XMI_ARG(src_stride % 4 == 0 && ((uintptr_t)d_src & 3) == 0, "xmi_intra_search_batch_dev: source frame must be 4-byte aligned");
    return launch_search(d_src, src_stride, d_ref_samples, d_req, n, d_res, 4 | 8 | 16 | 32, stream);
}

extern "C" int xmi_intra_chroma_search_batch_dev(const uint8_t *d_src_u, const uint8_t *d_src_v, int src_stride, const uint8_t *d_ref_samples,
                                                 const xmi_chroma_request *d_req, int n, xmi_chroma_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_intra_chroma_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(d_src_u && d_src_v && d_ref_samples && d_req && d_res, "xmi_intra_chroma_search_batch_dev: null argument");
    XMI_ARG(src_stride > 0 && src_stride % 4 == 0 && (((uintptr_t)d_src_u | (uintptr_t)d_src_v) & 3) == 0,
            "xmi_intra_chroma_search_batch_dev: chroma planes must be 4-byte aligned");
    ChromaParams p = { d_src_u, d_src_v, d_ref_samples, d_req, d_res, src_stride, n };
    hipLaunchKernelGGL(intra_chroma_search_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_intra_chroma_search(const uint8_t *src_u, const uint8_t *src_v, uint32_t src_stride, const uint8_t *ref_u, const uint8_t *ref_v,
                                       int size, int luma_mode, xmi_chroma_result *res)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(src_u && src_v && ref_u && ref_v && res, "xmi_intra_chroma_search: null argument");
    XMI_ARG(size == 4 || size == 8 || size == 16, "xmi_intra_chroma_search: size %d not in {4,8,16}", size);
    XMI_ARG(luma_mode >= 0 && luma_mode < 35, "xmi_intra_chroma_search: luma mode %d", luma_mode);
    std::lock_guard<std::mutex> lk(g_pred_mu);
    // staging: U block | V block (rows `size` apart) | U samples | V samples | request | result
    const size_t blk = (size_t)size * size, off_ref = 512, off_req = off_ref + 2 * XMI_INTRA_REF_PITCH, off_res = off_req + 16;
    static_assert(sizeof(xmi_chroma_request) <= 16 && 512 + 2 * XMI_INTRA_REF_PITCH + 16 + sizeof(xmi_chroma_result) <= 2048, "staging layout");
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(9, 2048));
    if (!dev) return XMI_ERR_HIP;
    uint8_t host[512 + 2 * XMI_INTRA_REF_PITCH + 16];
    memset(host, 0, sizeof host);
    for (int y = 0; y < size; y++) {
        memcpy(host + (size_t)y * size, src_u + (size_t)y * src_stride, (size_t)size);
        memcpy(host + blk + (size_t)y * size, src_v + (size_t)y * src_stride, (size_t)size);
    }
    memcpy(host + off_ref, ref_u, (size_t)(4 * size + 1));
    memcpy(host + off_ref + XMI_INTRA_REF_PITCH, ref_v, (size_t)(4 * size + 1));
    xmi_chroma_request rq = { 0, 0, (uint8_t)size, (uint8_t)luma_mode, { 0, 0 } };
    memcpy(host + off_req, &rq, sizeof rq);
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    rc = xmi_intra_chroma_search_batch_dev(dev, dev + blk, size, dev + off_ref, reinterpret_cast<const xmi_chroma_request *>(dev + off_req), 1,
                                           reinterpret_cast<xmi_chroma_result *>(dev + off_res), nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(res, dev + off_res, sizeof(xmi_chroma_result), hipMemcpyDeviceToHost));
    return XMI_OK;
}
