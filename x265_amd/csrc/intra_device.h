// intra_device.h - device functions of the intra searches that more than one kernel uses: per-sample prediction
// (xEncIntraPred, src/pixel.cpp:7338), padding and smoothing of the neighbouring samples (xPaddingRef / xFilterRef,
// src/encode.cpp:3843, 3873), the 35-mode luma search of one PU on one or four wavefronts and the five-candidate chroma
// search on one wavefront. Include from a .hip file only.
#pragma once
#include "xmi_common.h"

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
// (angle, inv: k_angle[mode], k_inv_angle[mode] - table reads by lane, which a caller with a fixed mode makes once)
template <int N>
__device__ __forceinline__ int pred_sample(const uint8_t *ref, int mode, int angle, int inv, int y, int x, int dc, bool is_luma)
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

template <int N>
__device__ __forceinline__ int pred_sample(const uint8_t *ref, int mode, int y, int x, int dc, bool is_luma)
{
    return pred_sample<N>(ref, mode, mode > 1 ? k_angle[mode] : 0, mode > 1 ? k_inv_angle[mode] : 0, y, x, dc, is_luma);
}

// xPaddingRef (src/encode.cpp:3843) on 4 N + 1 samples in LDS, by one wavefront: the samples below the first available
// segment take its first sample, every later unavailable segment repeats the sample in front of it. valids != 0.
template <int N>
__device__ __forceinline__ void pad_ref(uint8_t *ref0, unsigned valids, int lane)
{
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
}

// WPR wavefronts work on one request: each prepares the samples for itself and takes every WPR-th pass of modes;
// s_best (WPR entries of cost, mode) joins them at the end.
template <int N, int WPR>
__device__ void search(const IntraParams &p, const xmi_intra_request &rq, int idx, uint8_t *s_ref, uint8_t *s_src, int lane, int sub,
                       double *s_best_cost, int *s_best_mode, int *lds_best = nullptr, int part = 0, int parts = 1, double *lds_cost = nullptr)
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
        pad_ref<N>(ref0, valids, lane);
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
    xmi_intra_result *out = p.res ? p.res + idx : nullptr;      // a caller that only wants the winner (lds_best) passes no result array
    // `parts` workgroups may share the modes of one request (part = this one's index): pass i of all goes to wavefront i % (WPR * parts)
    for (int m0 = (part * WPR + sub) * MP; m0 < 35; m0 += MP * WPR * parts) {
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
            if (out) out->satd[mode] = (u32)sum;
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
        if (out) {
            out->best_mode = best_mode;
            out->pred_idx = pi;
            out->best_cost = best_cost;
            out->reserved = 0;
        }
        if (lds_best) *lds_best = best_mode;
        if (lds_cost) *lds_cost = best_cost;
    }
}

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

// the candidate loop on samples and source blocks that are in LDS already (s_ref: padded, U then V; s_src: rows N apart);
// lane 0 returns the winner: index into the candidate list and the mode it stands for
template <int N>
__device__ __forceinline__ void chroma_candidates(int luma, uint8_t (*s_ref)[136], uint8_t (*s_src)[256], u32 *s_satd, int lane, xmi_chroma_result &out)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : 4;
    constexpr int T = (N / 4) * (N / 4);             // 4x4 tiles per block
    int dc[2];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        int sum = 0;
        for (int i = lane; i < 2 * N; i += 64) sum += i < N ? s_ref[pl][2 * N - 1 - i] : s_ref[pl][2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) sum += __shfl_xor(sum, k);
        dc[pl] = (sum + N) >> (LG + 1);
    }
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
    }
}

template <int N>
__device__ __forceinline__ void chroma_search(const ChromaParams &p, const xmi_chroma_request &rq, int idx, uint8_t (*s_ref)[136], uint8_t (*s_src)[256],
                                              u32 *s_satd, int lane)
{
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
    xmi_chroma_result out;
    chroma_candidates<N>(rq.luma_mode, s_ref, s_src, s_satd, lane, out);
    if (lane == 0) p.res[idx] = out;
}

}  // namespace
