// analysis_gpu.hip - the P-picture analysis loop on the GPU: host/analysis_core.cpp (decision recursion, rate estimate,
// wavefront of CTU rows) bound to an evaluator that serves every batch of CU requests with ONE kernel launch, and the C ABI
// around it (xmi_analysis_*).
//
// A batch is a handful of CUs (one per CTU row that is ready), and what the host waits for is its latency, not its
// throughput. So a round trip is: the host writes the job table into pinned memory the GPU reads directly, launches
// analysis_round_kernel, and spins on a word of pinned memory; the kernel writes every result straight into pinned host
// memory and its last workgroup sets that word. No copies are enqueued and nothing waits on an interrupt.
//   luma job, two workgroups (they share no data, so the two candidates of a CU are evaluated side by side):
//     both     SSD of every usable merge candidate against the source - a prediction is a plain copy out of the quarter-pel
//              plane of its phase - and the first smallest (src/encode.cpp:3272-3289)
//     merge    transform chain of the source against that prediction (xPreEncRec), SSD of the reconstruction
//     search   xMotionSearch (me_search_device.h, one wavefront) with bZeroMV from the selection; prediction at the vector
//              found; transform chain; SSD of the reconstruction
//   chroma job (one plane of one transform unit), one workgroup: 4-tap prediction at the CU's vector, transform chain
// The transform chain is the VALU form of tu_chain.hip (LDS operands, v_dot2) for one unit per workgroup; a 64x64 CU runs its
// four 32x32 units one after the other.
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "host/analysis_core.h"
#include "host/encoder_core.h"
#include "intra_device.h"
#include "me_search_device.h"
#include "tu_valu_device.h"
#include "rdoq_device.h"

using namespace xmi::analysis;

namespace {

constexpr int kPad = 80, kPadC = 40;          // PAD_YSIZE / PAD_CSIZE of the reference (inc/config.h:52-53)
constexpr int kMeRange = 32;                  // ME_RANGE (inc/config.h:50)

// ---- job and result records (pinned host memory, read / written by the kernel) ----
struct LumaJob {
    int32_t x0, y0, size;
    int32_t n_merge;
    int16_t merge_mv[5][2];
    uint8_t merge_ok[5];
    uint8_t urgent;               // the host waits for the merge path of this CU before it takes the round's answers
    uint8_t reserved[2];
    int16_t amvp[2][2];
    int16_t mv_min_ipel[2], mv_max_ipel[2];
    uint32_t sqrt_lambda;
    uint32_t out_head;            // byte offsets into the result area: LumaHead, then the blocks
    uint32_t out_pred_merge, out_rec_merge, out_level_merge, out_rec_me, out_level_me;
    uint32_t out_done;            // eight words (path x transform unit) in the part of the result area that holds nothing else
};
struct LumaHead {                 // per transform unit: a 64x64 CU has four, each written by a workgroup of its own
    int32_t best_merge;
    uint32_t ssd_skip, ssd_merge[4], sum_merge[4];
    int16_t mv[2];
    uint32_t ssd_me[4], sum_me[4];
};
struct ChromaJob {
    int32_t plane;                // 0 = U, 1 = V
    int32_t x, y, n;              // unit in plane samples: 4, 8 or 16 square
    int16_t mv[2];                // the CU's luma vector (eighth-pel in chroma)
    int32_t skip;
    uint32_t out_sum, out_rec, out_level;
    int32_t rec_stride;           // samples per row of the CU's result blocks (the unit may be a quarter of them)
};

struct RoundParams {
    const LumaJob *jobs;
    const ChromaJob *cjobs;
    int nluma, nchroma;           // workgroups of the luma part, chroma jobs
    int n64;                      // 64x64 CUs, which are the first jobs of the table
    const uint8_t *src[3];        // source planes, rows width / width/2 apart
    const uint8_t *planes;        // 16 quarter-pel luma planes of the padded reference: rows `stride` bytes apart, phases `phase_pitch`
    unsigned phase_pitch, row_bytes;      // row_bytes: bytes of a row
    const uint8_t *ref_c[2];      // sample (0,0) of the padded chroma reference planes
    int width, stride, rows, stride_c;
    int qp, qpc;
    const RdoqSet *rdoq;          // tables of the level decision that takes the quantiser's place, or null (the reference's quantiser)
    uint8_t *out;                 // result area
    unsigned *counter;            // workgroups finished since the handle was made (device memory)
    unsigned target;              // value of the counter when this round is complete
    unsigned *flag;               // pinned host word: the round's number once every workgroup is through
    unsigned seq;
    unsigned *counter_early;      // the same three for the workgroups whose answers are read first (merge paths of the CUs that
    unsigned target_early;        // asked, chroma): what the host waits for
    unsigned *flag_early;
};

__constant__ int k_chroma_taps[8][4] = { { 0, 64, 0, 0 },   { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                         { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };   // src/table.cpp:532-541

// LDS of one workgroup: transform operands (up to 32x32), the unit's source and prediction bytes, the search window
struct __align__(16) RoundLds {
    int16_t mat[2][32 * 40];      // core matrix and its transpose, rows Cfg<N>::LD apart
    int16_t a[32 * 40], b[32 * 40];
    uint8_t pix[32 * 32], pred[32 * 32];
    uint8_t win[(64 + 16) * (64 + 24)];
    uint32_t red[4];
    uint32_t red5[4][5];
    uint32_t red36[4 * 9];
    uint32_t acc[2];
    xmi_me_result me;
};

__device__ __forceinline__ const uint8_t *plane_block(const RoundParams &p, int x0, int y0, int mvx, int mvy)
{
    const size_t plane = (size_t)((mvy & 3) * 4 + (mvx & 3)) * (size_t)p.phase_pitch;
    return p.planes + plane + (size_t)(kPad + y0 + (mvy >> 2)) * p.stride + kPad + x0 + (mvx >> 2);
}

#ifdef XMI_EXP_ROUND_TIMING
__device__ unsigned long long g_chain_ticks[4][6];      // per unit size 4/8/16/32: set-up, forward, quantiser, inverse, output, count
#define XMI_CSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define XMI_CSTAMP(var)
#endif

// xSubDctQuantRec (src/encode.cpp:3005-3021) of one N x N unit by the whole workgroup; P-slice rounding and DCT unless the
// template arguments say I slice / DST:
// source and prediction come with their own strides (any address space); levels, reconstruction and, when asked for, the
// prediction go out with the stride of the CU's result blocks. Returns sum |level| and the SSD of the reconstruction
// against the source in every thread.
// DST-VII matrix of 4x4 intra luma blocks (xSubDST4, src/pixel.cpp:1567)
__constant__ int8_t k_dst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

// LAZY: the levels go out only when one of them is not zero - the caller reads a unit's levels only when its sum says so, and
// most units of a P picture quantise to nothing (the result area is host memory: every store crosses the bus).
template <int N, bool INTRA = false, bool DST = false, bool LAZY = false>
__device__ void chain_unit(RoundLds &L, const uint8_t *pix, int pix_stride, const uint8_t *pred, int pred_stride, int qp, int16_t *level,
                           uint8_t *rec, uint8_t *pred_out, int out_stride, uint32_t &sum, uint32_t &ssd, uint8_t *rec_copy = nullptr, int copy_stride = 0,
                           const RdoqSet *rq = nullptr, int comp = 0)
{
    constexpr int LD = Cfg<N>::LD, LG = Cfg<N>::LOG2, TPT = Cfg<N>::TPT;
    const int tid = threadIdx.x;
    const bool live = tid < TPT;
    XMI_CSTAMP(c0);
    int16_t *M = L.mat[0], *Mt = L.mat[1];
    // four neighbouring samples per thread and step: the source is dword aligned (picture positions and strides are multiples
    // of 4), the prediction starts at any byte (a block of a quarter-pel plane, or LDS)
    for (int e = tid * 4; e < N * N; e += 1024) {
        const int k = e / N, j = e % N;
        const uint32_t s4 = *reinterpret_cast<const uint32_t *>(pix + (size_t)k * pix_stride + j);
        const uint8_t *pp = pred + (size_t)k * pred_stride + j;
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(pp) & 3);
        const uint32_t *pw = reinterpret_cast<const uint32_t *>(pp - sh);      // (pointer arithmetic keeps the address space known)
        const uint32_t q4 = __builtin_amdgcn_alignbyte(pw[1], pw[0], sh);
        *reinterpret_cast<uint32_t *>(&L.pix[e]) = s4;
        *reinterpret_cast<uint32_t *>(&L.pred[e]) = q4;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int16_t c = (int16_t)(DST ? (int)k_dst4[k & 3][(j + t) & 3] : dct_coef(N, k, j + t));
            M[k * LD + j + t] = c;
            Mt[(j + t) * LD + k] = c;
            L.a[k * LD + j + t] = (int16_t)((int)((s4 >> (8 * t)) & 255) - (int)((q4 >> (8 * t)) & 255));
        }
    }
    __syncthreads();
    XMI_CSTAMP(c1);
    if (live) stage<N, false>(M, L.a, L.b, LG - 1, tid);
    __syncthreads();
    if (live) stage<N, false>(M, L.b, L.a, LG + 6, tid);
    __syncthreads();
    XMI_CSTAMP(c2);
    const int qbits = 14 + qp / 6 + (15 - 8 - LG), q = k_quant_scales[qp % 6], qrnd = (INTRA ? 171 : 85) << (qbits - 9);      // xQuant (src/pixel.cpp:1060)
    const int dshift = 20 - 14 - (15 - 8 - LG), dscale = k_inv_quant_scales[qp % 6] << (qp / 6);
    uint32_t local = 0;
    uint2 held = make_uint2(0u, 0u);      // (a unit has at most 1024 samples: one step per thread)
    static_assert(N <= 32, "one quantiser step per thread");
    // The level decision in the quantiser's place (rq != null; rdoq_device.h): rounded levels in scan order by all threads, the
    // decision on one of them, then the loop below takes its magnitudes from the decided levels. Scratch is what the chain
    // does not need at this point: the search window (costs, levels, tables), the forward matrix (the scan) and L.b (groups).
    int best_last = -1;
    const int16_t *r_lvl = reinterpret_cast<const int16_t *>(L.win + 4096);
    if (rq) {
        constexpr int SETS = (N / 4) * (N / 4);
        uint32_t *pos_b = reinterpret_cast<uint32_t *>(L.win);
        int16_t *lvl = reinterpret_cast<int16_t *>(L.win + 4096);
        xmi_rdoq_tables *tab = reinterpret_cast<xmi_rdoq_tables *>(L.win + 4096 + 2048);
        static_assert(4096 + 2048 + sizeof(xmi_rdoq_tables) <= sizeof L.win, "decision scratch fits the search window");
        uint16_t *scan = reinterpret_cast<uint16_t *>(M);
        unsigned char *gb = reinterpret_cast<unsigned char *>(L.b);
        long long *cg_d = reinterpret_cast<long long *>(gb);
        int *cg_b = reinterpret_cast<int *>(gb + 512);
        uint8_t *cg_flag = gb + 768, *scan_cg = gb + 832;
        unsigned long long *s_d = reinterpret_cast<unsigned long long *>(gb + 896);      // [0] nothing coded, [1] behind the last rounded level
        int *s_int = reinterpret_cast<int *>(gb + 912);                                  // [0] last rounded level, [1] the decision
        const int eshift = qbits - 4;
        for (int i2 = tid; i2 < (int)(sizeof(xmi_rdoq_tables) / 4); i2 += 256) reinterpret_cast<int *>(tab)[i2] = reinterpret_cast<const int *>(&rq->tables[LG - 2][comp])[i2];
        for (int i2 = tid; i2 < SETS; i2 += 256) cg_flag[i2] = 0, scan_cg[i2] = g_rdoq_scan_cg[0][LG - 2][i2];
        if (tid == 0) s_d[0] = s_d[1] = 0, s_int[0] = -1, s_int[1] = -1;
        __syncthreads();
        int top_l = -1;
        for (int i2 = tid; i2 < N * N; i2 += 256) {
            const int pos = g_rdoq_scan[0][LG - 2][i2];
            scan[i2] = (uint16_t)pos;
            const uint32_t v = (uint32_t)(abs((int)L.a[(pos >> LG) * LD + (pos & (N - 1))]) * q);
            uint32_t mx = (uint32_t)(((unsigned long long)v + (1u << (qbits - 1))) >> qbits);
            if (mx > 32767u) mx = 32767u;
            lvl[i2] = (int16_t)mx;
            if (mx) top_l = i2;
        }
        atomicMax(&s_int[0], top_l);
        __syncthreads();
        const int top = s_int[0];
        if (top >= 0) {
            unsigned long long all0 = 0, beyond = 0;
            for (int i2 = tid; i2 < N * N; i2 += 256) {
                const int pos = scan[i2];
                const long long e = (long long)(uint32_t)(abs((int)L.a[(pos >> LG) * LD + (pos & (N - 1))]) * q) >> eshift;
                all0 += (unsigned long long)(e * e);
                if (i2 > top) beyond += (unsigned long long)(e * e);
            }
            atomicAdd(&s_d[0], all0);
            atomicAdd(&s_d[1], beyond);
            __syncthreads();
            if (tid == 0)
                s_int[1] = rdoq_decide<N>([&](int i2) { const int pos = scan[i2]; return (uint32_t)(abs((int)L.a[(pos >> LG) * LD + (pos & (N - 1))]) * q); }, scan, scan_cg,
                                          lvl, pos_b, cg_d, cg_b, cg_flag, *tab, top, (long long)s_d[0], (long long)s_d[1], qbits,
                                          rq->lambda[LG - 2][comp].mantissa, rq->lambda[LG - 2][comp].shift, 0, comp == 0);
            __syncthreads();
            best_last = s_int[1];
        }
        __syncthreads();      // the group arrays in L.b are done with: the loop below fills it
    }
    for (int e = tid * 4; e < N * N; e += 1024) {
        const int k = e / N, i = e % N;
        int lv[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int c = L.a[k * LD + i + t];
            int mag = (abs(c) * q + qrnd) >> qbits;
            if (rq) {
                const int at = g_rdoq_inv[0][LG - 2][k * N + i + t];
                mag = at <= best_last ? r_lvl[at] : 0;
            }
            local += (uint32_t)mag;
            lv[t] = clip16(c < 0 ? -mag : mag);
            L.b[(i + t) * LD + k] = (int16_t)clip16((lv[t] * dscale + (1 << (dshift - 1))) >> dshift);
        }
        held = make_uint2((uint32_t)(lv[0] & 0xffff) | ((uint32_t)lv[1] << 16), (uint32_t)(lv[2] & 0xffff) | ((uint32_t)lv[3] << 16));
        if (!LAZY) *reinterpret_cast<uint2 *>(level + (size_t)k * out_stride + i) = held;
    }
    // (the barrier also tells every thread whether the unit has a level at all: without one its reconstruction is its
    // prediction, and the prediction stays here - the result area is host memory, and the writes to it are what rounds of
    // several handles queue for)
    const bool any_level = __syncthreads_or(local != 0) != 0;
    XMI_CSTAMP(c3);
    if (live) stage<N, true>(Mt, L.b, L.a, 7, tid);
    __syncthreads();
    if (live) stage<N, true>(Mt, L.a, L.b, 12, tid);
    __syncthreads();
    XMI_CSTAMP(c4);
    uint32_t d2 = 0;
    for (int e = tid * 4; e < N * N; e += 1024) {
        const int i = e / N, j = e % N;
        const uint32_t pr4 = *reinterpret_cast<const uint32_t *>(&L.pred[e]), s4 = *reinterpret_cast<const uint32_t *>(&L.pix[e]);
        uint32_t r4 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int v = min(255, max(0, (int)L.b[(j + t) * LD + i] + (int)((pr4 >> (8 * t)) & 255)));
            r4 |= (uint32_t)v << (8 * t);
            const int d = (int)((s4 >> (8 * t)) & 255) - v;
            d2 += (uint32_t)(d * d);
        }
        *reinterpret_cast<uint32_t *>(rec + (size_t)i * out_stride + j) = r4;
        if (rec_copy) *reinterpret_cast<uint32_t *>(rec_copy + i * copy_stride + j) = r4;
        if (pred_out && any_level) *reinterpret_cast<uint32_t *>(pred_out + (size_t)i * out_stride + j) = pr4;
    }
    // the two sums of the unit: one value per wavefront into LDS (256 atomics on one word take longer than the transform)
    local = wave_sum(local);
    d2 = wave_sum(d2);
    if ((tid & 63) == 0) L.red5[tid >> 6][0] = local, L.red5[tid >> 6][1] = d2;
    __syncthreads();
    sum = L.red5[0][0] + L.red5[1][0] + L.red5[2][0] + L.red5[3][0];
    ssd = L.red5[0][1] + L.red5[1][1] + L.red5[2][1] + L.red5[3][1];
    __syncthreads();
    if (LAZY && sum && tid * 4 < N * N) *reinterpret_cast<uint2 *>(level + (size_t)((tid * 4) / N) * out_stride + (tid * 4) % N) = held;
#ifdef XMI_EXP_ROUND_TIMING
    if (tid == 0) {
        const unsigned long long c5 = __builtin_amdgcn_s_memrealtime();
        constexpr int si = N == 4 ? 0 : N == 8 ? 1 : N == 16 ? 2 : 3;
        atomicAdd(&g_chain_ticks[si][0], c1 - c0), atomicAdd(&g_chain_ticks[si][1], c2 - c1), atomicAdd(&g_chain_ticks[si][2], c3 - c2);
        atomicAdd(&g_chain_ticks[si][3], c4 - c3), atomicAdd(&g_chain_ticks[si][4], c5 - c4), atomicAdd(&g_chain_ticks[si][5], 1ull);
    }
#endif
}

#ifdef XMI_EXP_ROUND_TIMING
// measurement build: per CU size (8, 16, 32, 64) and path, 100 MHz ticks spent in merge selection / search / chain, and counts
__device__ unsigned long long g_round_ticks[4][2][4];
// rounds with 64x64 CUs, 100 MHz ticks from the first workgroup's start: to the early flag, to the full flag; rounds; then the
// longest early luma workgroup and the longest chroma workgroup of a round, summed over rounds
__device__ unsigned long long g_span[5];
__device__ unsigned long long g_span_start[1024];
__device__ unsigned g_span_max[1024][2];
// 64x64 merge path: workgroups by their length in steps of 4 us (last bin: 60 us and more), and by the length of their
// merge selection alone in steps of 2 us
__device__ unsigned long long g_merge64_hist[2][16];
#define XMI_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define XMI_STAMP(var)
#endif

template <int N>
__device__ void luma_job(const RoundParams &p, RoundLds &L, const LumaJob &j, int path, int u)
{
    XMI_STAMP(t_begin);
    constexpr int UN = N == 64 ? 32 : N;
    const int tid = threadIdx.x;
    const uint8_t *src = p.src[0] + (size_t)j.y0 * p.width + j.x0;
    // merge candidates: both workgroups of the job come to the same selection. The candidates do not depend on each other:
    // their loads go out together and one reduction serves all of them.
    const uint8_t *cand[5];
    int n_ok = 0, n_zero = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        cand[k] = (k < j.n_merge && j.merge_ok[k]) ? plane_block(p, j.x0, j.y0, j.merge_mv[k][0], j.merge_mv[k][1]) : nullptr;
        if (cand[k]) n_ok++, n_zero += j.merge_mv[k][0] == 0 && j.merge_mv[k][1] == 0;
    }
    // The search path needs the selection for one bit only: whether the winner is the zero vector (bZeroMV, src/encode.cpp:3341).
    // Usable candidates are distinct, so without a zero vector among them the bit is 0, and with the zero vector as the only
    // one it is 1: no distortion has to be measured for that.
    const bool known = path == 1 && (n_zero == 0 || n_ok == 1);
    uint32_t best_ssd = 0xffffffffu;
    int best = -1;
    auto select = [&] {
        uint32_t acc[5] = { 0, 0, 0, 0, 0 };
        // 64x64 (four steps per thread, few CUs, the workgroups a round waits for): no branch inside the loops - a candidate
        // that is left out reads the block of one that is not and its sum is ignored below - so that the loads of all
        // candidates and steps go out together instead of one latency after the other. The smaller sizes have one step and
        // are many: they skip the candidates that are left out.
        const uint8_t *first = nullptr;
#pragma unroll
        for (int k = 4; k >= 0; k--) first = cand[k] ? cand[k] : first;
        if (!first) return;
        const uint8_t *blk[5];
#pragma unroll
        for (int k = 0; k < 5; k++) blk[k] = cand[k] ? cand[k] : first;
#pragma unroll
        for (int i = tid * 4; i < N * N; i += 1024) {
            const int y = i / N, x = i - y * N;
            const uint32_t a = *reinterpret_cast<const uint32_t *>(src + (size_t)y * p.width + x);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                if (N < 64 && !cand[k]) continue;
                // four samples of the candidate from the two aligned dwords that hold them; SSD = a.a + b.b - 2 a.b on bytes
                const uint8_t *bp = blk[k] + (size_t)y * p.stride + x;
                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(bp) & 3);
                const uint32_t *bw = reinterpret_cast<const uint32_t *>(bp - sh);
                const uint32_t b = __builtin_amdgcn_alignbyte(bw[1], bw[0], sh);
                acc[k] += __builtin_amdgcn_udot4(a, a, 0u, false) + __builtin_amdgcn_udot4(b, b, 0u, false) - 2u * __builtin_amdgcn_udot4(a, b, 0u, false);
            }
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        }
        __syncthreads();
        if ((tid & 63) == 0)
#pragma unroll
            for (int k = 0; k < 5; k++) L.red5[tid >> 6][k] = acc[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t ssd = L.red5[0][k] + L.red5[1][k] + L.red5[2][k] + L.red5[3][k];
            if (cand[k] && ssd < best_ssd) best_ssd = ssd, best = k;      // the first smallest (src/encode.cpp:3281)
        }
    };
    // A search on the whole workgroup (32x32, 64x64) asks for the bit only when its first stage shows that it matters, and
    // the selection is made then.
    constexpr bool kLazy = N >= 32;
    if (!known && !(kLazy && path == 1)) select();
    LumaHead *head = reinterpret_cast<LumaHead *>(p.out + j.out_head);
    XMI_STAMP(t_merge);
    int mvx, mvy;
    if (path == 0) {
        if (tid == 0) head->best_merge = best, head->ssd_skip = best_ssd;
        if (best < 0) return;
        mvx = j.merge_mv[best][0], mvy = j.merge_mv[best][1];
    } else {
        xmi_me_request rq;
        rq.x0 = j.x0, rq.y0 = j.y0, rq.size = N;
        rq.mvp[0][0] = j.amvp[0][0], rq.mvp[0][1] = j.amvp[0][1], rq.mvp[1][0] = j.amvp[1][0], rq.mvp[1][1] = j.amvp[1][1];
        auto zero_won = [&] { return best >= 0 && j.merge_mv[best][0] == 0 && j.merge_mv[best][1] == 0; };
        rq.zero_mv_tested = known ? n_zero != 0 : zero_won();
        rq.mv_min_ipel[0] = j.mv_min_ipel[0], rq.mv_min_ipel[1] = j.mv_min_ipel[1];
        rq.mv_max_ipel[0] = j.mv_max_ipel[0], rq.mv_max_ipel[1] = j.mv_max_ipel[1];
        rq.sqrt_lambda = j.sqrt_lambda;
        MeParams mp = { p.src[0], p.planes, nullptr, &L.me, p.width, p.stride, p.rows, kPad, 1, kMeRange, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, p.phase_pitch, p.row_bytes };
        if constexpr (kLazy) {      // the whole workgroup shares the block
            if (known) {
                search<N, false, false, 4>(mp, rq, 0, tid, L.win, L.red36);
            } else {
                search<N, false, false, 4>(mp, rq, 0, tid, L.win, L.red36, [&] {
                    select();
                    return zero_won();
                });
            }
        } else {
            if (tid < 64) search<N, false>(mp, rq, 0, tid, L.win);
        }
        __syncthreads();
        mvx = L.me.mv[0], mvy = L.me.mv[1];
        if (tid == 0) head->mv[0] = (int16_t)mvx, head->mv[1] = (int16_t)mvy;
    }
    XMI_STAMP(t_search);
    const uint8_t *pr = plane_block(p, j.x0, j.y0, mvx, mvy);
    int16_t *level = reinterpret_cast<int16_t *>(p.out + (path ? j.out_level_me : j.out_level_merge));
    uint8_t *rec = p.out + (path ? j.out_rec_me : j.out_rec_merge), *pred_out = path ? nullptr : p.out + j.out_pred_merge;
    const int ux = (u & 1) * 32, uy = (u >> 1) * 32;      // u is 0 unless the CU is 64x64
    const size_t o = (size_t)uy * N + ux;
    uint32_t sum, ssd;
    chain_unit<UN, false, false, true>(L, src + (size_t)uy * p.width + ux, p.width, pr + (size_t)uy * p.stride + ux, p.stride, p.qp, level + o, rec + o,
                   pred_out ? pred_out + o : nullptr, N, sum, ssd, nullptr, 0, p.rdoq, 0);
    if (tid == 0) {
        (path ? head->sum_me : head->sum_merge)[u] = sum;
        (path ? head->ssd_me : head->ssd_merge)[u] = ssd;
#ifdef XMI_EXP_ROUND_TIMING
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        constexpr int si = N == 8 ? 0 : N == 16 ? 1 : N == 32 ? 2 : 3;
        atomicAdd(&g_round_ticks[si][path][0], t_merge - t_begin);
        atomicAdd(&g_round_ticks[si][path][1], t_search - t_merge);
        atomicAdd(&g_round_ticks[si][path][2], t_end - t_search);
        atomicAdd(&g_round_ticks[si][path][3], 1ull);
        if (N == 64 && path == 0) {
            atomicAdd(&g_merge64_hist[0][min(15ull, (t_end - t_begin) / 400)], 1ull);
            atomicAdd(&g_merge64_hist[1][min(15ull, (t_merge - t_begin) / 200)], 1ull);
        }
#endif
    }
}

// one sample of xPredInterChromaBlk (src/pixel.cpp:9657-9703): 4-tap filters at eighth-pel; with both fractions the second
// stage works on the unrounded first-stage sums, (sum + 2048) >> 12 (the reference's -8192 / +8192*64 offsets cancel)
__device__ __forceinline__ int chroma_sample(const uint8_t *ref, int stride, int xf, int yf)
{
    if (!xf && !yf) return ref[0];
    if (!yf) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) s += (int)ref[t - 1] * k_chroma_taps[xf][t];
        return min(255, max(0, (s + 32) >> 6));
    }
    if (!xf) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) s += (int)ref[(ptrdiff_t)(t - 1) * stride] * k_chroma_taps[yf][t];
        return min(255, max(0, (s + 32) >> 6));
    }
    int s = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int hsum = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) hsum += (int)ref[(ptrdiff_t)(r - 1) * stride + t - 1] * k_chroma_taps[xf][t];
        s += hsum * k_chroma_taps[yf][r];
    }
    return min(255, max(0, (s + 2048) >> 12));
}

template <int N>
__device__ void chroma_job(const RoundParams &p, RoundLds &L, const ChromaJob &j, uint8_t *pred_lds)
{
    const int tid = threadIdx.x;
    const uint8_t *ref = p.ref_c[j.plane] + (ptrdiff_t)(j.y + (j.mv[1] >> 3)) * p.stride_c + j.x + (j.mv[0] >> 3);
    const int xf = j.mv[0] & 7, yf = j.mv[1] & 7;
    for (int e = tid; e < N * N; e += 256) {
        const int y = e / N, x = e - y * N;
        pred_lds[e] = (uint8_t)chroma_sample(ref + (ptrdiff_t)y * p.stride_c + x, p.stride_c, xf, yf);
    }
    __syncthreads();
    uint8_t *rec = p.out + j.out_rec;
    if (j.skip) {
        for (int e = tid; e < N * N; e += 256) rec[(size_t)(e / N) * j.rec_stride + e % N] = pred_lds[e];
        return;
    }
    uint32_t sum, ssd;
    chain_unit<N, false, false, true>(L, p.src[1 + j.plane] + (size_t)j.y * (p.width / 2) + j.x, p.width / 2, pred_lds, N, p.qpc,
                  reinterpret_cast<int16_t *>(p.out + j.out_level), rec, nullptr, j.rec_stride, sum, ssd, nullptr, 0, p.rdoq, 1);
    if (tid == 0) *reinterpret_cast<uint32_t *>(p.out + j.out_sum) = sum;
}

// completion of a round: every wavefront waits until its own results have left for host memory, then the workgroup counts
// itself; the last one of the round tells the host
__device__ __forceinline__ void round_done(unsigned *counter, unsigned target, unsigned *flag, unsigned seq)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(counter, 1u) + 1u;
        if (done == target) {
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Completion of a round of a P picture. A row reads the merge results of the CU that asked first and, where a merge candidate
// ends the CU (most CTUs of a picture with coherent motion end so at 64x64), nothing else of what it asked for; the searches
// are the longest workgroups. So every luma workgroup reports in a word of its CU's result header, where the host looks when
// it comes to need that part; the host is told (flag_early) when the workgroups it reads first are through, and `flag` says
// when every workgroup is - before that the tables and the result area must not be used again.
__device__ __forceinline__ void round_done_split(const RoundParams &p, unsigned *done_word, bool early)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every thread has fenced before the barrier: the word needs no release of its own)
        if (done_word) __hip_atomic_store(done_word, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (early && atomicAdd(p.counter_early, 1u) + 1u == p.target_early) {
            __threadfence_system();
            __hip_atomic_store(p.flag_early, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef XMI_EXP_ROUND_TIMING
            if (p.n64 > 0) {
                const unsigned slot = (p.seq * 2u + (unsigned)((reinterpret_cast<uintptr_t>(p.flag) >> 12) & 1u)) & 1023u;
                const unsigned long long t0 = atomicAdd(&g_span_start[slot], 0ull);
                if (t0) atomicAdd(&g_span[0], __builtin_amdgcn_s_memrealtime() - (t0 - 1)), atomicAdd(&g_span[2], 1ull);
                atomicAdd(&g_span[3], (unsigned long long)atomicExch(&g_span_max[slot][0], 0u));
                atomicAdd(&g_span[4], (unsigned long long)atomicExch(&g_span_max[slot][1], 0u));
            }
#endif
        }
        if (atomicAdd(p.counter, 1u) + 1u == p.target) {
            __threadfence_system();
            __hip_atomic_store(p.flag, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef XMI_EXP_ROUND_TIMING
            if (p.n64 > 0) {
                const unsigned slot = (p.seq * 2u + (unsigned)((reinterpret_cast<uintptr_t>(p.flag) >> 12) & 1u)) & 1023u;
                const unsigned long long t0 = atomicExch(&g_span_start[slot], 0ull);
                if (t0) atomicAdd(&g_span[1], __builtin_amdgcn_s_memrealtime() - (t0 - 1));
            }
#endif
        }
    }
}

// A round of a picture with coherent motion brings several hundred workgroups; registers are held to 128 so that four of
// them fit on a CU (the compiler needs no spill for it).
#ifndef XMI_ROUND_WAVES
#define XMI_ROUND_WAVES 4
#endif
#ifdef XMI_EXP_ROUND_TIMING
__device__ unsigned long long g_clock_ticks[2];      // shader clock ticks and 100 MHz ticks, summed over workgroups
static std::atomic<uint64_t> g_round_class[5][3];    // host side: calls of P pictures by their largest CU
#endif
__global__ void __launch_bounds__(256, XMI_ROUND_WAVES) analysis_round_kernel(const RoundParams p)
{
    __shared__ RoundLds L;
    __shared__ uint8_t s_cpred[16 * 16];
    const int b = blockIdx.x;
    unsigned *done_word = nullptr;
    bool early = true;
#ifdef XMI_EXP_ROUND_TIMING
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned slot = (p.seq * 2u + (unsigned)((reinterpret_cast<uintptr_t>(p.flag) >> 12) & 1u)) & 1023u;
    if (p.n64 > 0 && threadIdx.x == 0 && atomicCAS(&g_span_start[slot], 0ull, r0 + 1) != 0ull) atomicMin(&g_span_start[slot], r0 + 1);      // (0 = unset)
#endif
    if (b < p.nluma) {
        // The job table lies in host memory, a read of it is a trip over the bus: which job a workgroup serves follows from its
        // index alone - the 64x64 CUs come first, eight workgroups each (two candidates x four transform units), then two per CU
        const int rest = b - 8 * p.n64;
        const int job = rest < 0 ? b >> 3 : p.n64 + (rest >> 1), path = rest < 0 ? (b >> 2) & 1 : rest & 1, unit = rest < 0 ? b & 3 : 0;
        const LumaJob j = p.jobs[job];
        done_word = reinterpret_cast<unsigned *>(p.out + j.out_done) + path * 4 + unit;
        early = j.urgent && path == 0;
        switch (j.size) {
        case 8: luma_job<8>(p, L, j, path, 0); break;
        case 16: luma_job<16>(p, L, j, path, 0); break;
        case 32: luma_job<32>(p, L, j, path, 0); break;
        case 64: luma_job<64>(p, L, j, path, unit); break;
        default: break;
        }
    } else if (b - p.nluma < p.nchroma) {
        const ChromaJob j = p.cjobs[b - p.nluma];
        switch (j.n) {
        case 4: chroma_job<4>(p, L, j, s_cpred); break;
        case 8: chroma_job<8>(p, L, j, s_cpred); break;
        case 16: chroma_job<16>(p, L, j, s_cpred); break;
        default: break;
        }
    }
#ifdef XMI_EXP_ROUND_TIMING
    if (threadIdx.x == 0) {
        atomicAdd(&g_clock_ticks[0], __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(&g_clock_ticks[1], __builtin_amdgcn_s_memrealtime() - r0);
        if (p.n64 > 0 && early) atomicMax(&g_span_max[slot][done_word ? 0 : 1], (unsigned)(__builtin_amdgcn_s_memrealtime() - r0));
    }
#endif
    if (p.flag_early) round_done_split(p, done_word, early);
    else round_done(p.counter, p.target, p.flag, p.seq);      // the host waits for the whole round and reads no completion words
}

// ---- I pictures: one workgroup per PU request ----
struct IntraJob {
    int32_t x0, y0, size;
    uint8_t valids, strong, mpm[3], parts;       // parts: workgroups that share the PU's mode search
    // An 8x8 PU may leave its right column and mode for requests of the same round that stand for the block to its right
    // (publish: hand-over slot + 1), or be such a request (after: slot + 1 of the block to its left; top_mode: the mode above,
    // from which and the left block's mode the most probable modes follow). 0: neither.
    uint8_t publish, after, top_mode, reserved[3];
    int32_t qp;
    double sqrt_lambda;
    uint8_t ref[136];
    uint32_t out_head, out_rec, out_level;
};
struct IntraHead {
    int32_t best_mode;
    uint32_t sum, ssd;
};
struct IntraChromaJob {
    int32_t x0, y0, size;             // in chroma samples
    uint8_t valids, luma_mode, reserved[2];
    int32_t qpc;
    uint8_t ref[2][136];
    uint32_t out_head, out_rec[2], out_level[2];
};
struct IntraChromaHead {
    int32_t best_idx;
    uint32_t sum[2];
};
// four 4x4 PUs of an 8x8 block back to back (luma), and the same for an 8x8 area of the chroma planes
struct IntraQuadJob {
    int32_t x0, y0;
    uint8_t valids[4], left_mode[2], top_mode[2], strong, publish, after, reserved;      // publish / after: as in IntraJob
    int32_t qp;
    double sqrt_lambda;
    uint8_t ref[40];
    uint32_t out_head, out_rec, out_level;
};
struct IntraQuadHead {
    int32_t best_mode[4];
    uint32_t sum[4], ssd[4];
};
struct IntraChromaQuadJob {
    int32_t x0, y0;
    uint8_t valids[4], luma_mode[4];
    int32_t qpc;
    uint8_t ref[2][40];
    uint32_t out_head, out_rec[2], out_level[2];
};
struct IntraChromaQuadHead {
    int32_t best_idx[4];
    uint32_t sum[2][4];
};
struct IntraGroup {               // what one workgroup of the PU part does
    uint16_t job;
    uint8_t part, reserved;
};
struct IntraPartial {             // a workgroup's best candidate, for the workgroup of the PU that finishes last
    double cost;
    int32_t mode, reserved;
};
constexpr int kIntraParts = 3;    // workgroups per 16x16 / 32x32 PU

struct IntraRoundParams {
    const IntraJob *jobs;
    const IntraGroup *groups;
    IntraPartial *partials;       // [job][part], device memory
    unsigned *arrived;            // [job], device memory, zero between rounds
    const IntraChromaJob *cjobs;
    const IntraQuadJob *qjobs;
    const IntraChromaQuadJob *cqjobs;
    int npu, nchroma, nquad, ncquad;
    // Requests that stand after a block of the same round come last - their workgroups start after the ones they wait for -:
    // PU workgroups npu_first .. npu_first + npu_after - 1 of `groups`, 4x4 groups nquad .. nquad + nquad_after - 1 of `qjobs`.
    int npu_first, npu_after, nquad_after;
    unsigned *handoff;            // [slot][4], device memory: the round's number, right column (rows 0..3, 4..7), modes
    const uint8_t *src[3];
    int width;
    uint8_t *out;
    unsigned *counter;
    unsigned target;
    unsigned *flag;
    unsigned seq;
};
struct __align__(16) IntraLds {
    unsigned hand[4];                 // what the block to the left handed over
    int hand_ok;
    RoundLds r;                       // transform operands; r.win holds the source copies of the search and the prediction
    uint8_t ref[4][272];              // per wavefront: padded samples, then the smoothed set at +136
    double best_cost[4];
    int best_mode[4];
    int best;
    double cost;
    int last;
    uint32_t satd[10];
    xmi_chroma_result cres;
    // quads: the 8x8 area(s) being reconstructed, their outer samples, the samples of the PU in hand, the modes so far
    uint8_t area[2][64], outer[2][40], qref[2][20];
    int qmode[4];
    double qcost[36];                 // quads: cost of every mode of the PU in hand
    uint32_t qpred[36 * 4];           // quads: the four predicted samples of every (mode, row) of the PU in hand
};

// sample at (xx, yy) relative to the origin of an 8x8 area: from its 33 outer samples, or from the area itself
__device__ __forceinline__ uint8_t quad_sample(const uint8_t *outer, const uint8_t *area, int yy, int xx)
{
    if (yy < 0) return outer[16 + 1 + xx];
    if (xx < 0) return outer[15 - yy];
    return xx < 8 && yy < 8 ? area[yy * 8 + xx] : (uint8_t)0x80;
}

// A decided 8x8 block (area: its reconstruction in LDS, rows 8 apart) leaves its right column and the modes of its right-hand
// units for the workgroups of the round that evaluate the block to its right. They run on other CUs, maybe of another XCD:
// everything goes through device-scope atomics, the round's number last.
__device__ __forceinline__ void publish_block(const IntraRoundParams &p, int slot, const uint8_t *area, int mode_top, int mode_bottom)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned lo = 0, hi = 0;
        for (int r = 0; r < 4; r++) lo |= (unsigned)area[r * 8 + 7] << (8 * r), hi |= (unsigned)area[(r + 4) * 8 + 7] << (8 * r);
        unsigned *h = p.handoff + 4 * slot;
        __hip_atomic_store(h + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(h + 2, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(h + 3, (unsigned)mode_top | ((unsigned)mode_bottom << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __hip_atomic_store(h, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The other side: wait for the block to the left (its workgroup was started before this one; the wait is bounded all the same,
// and a request that could not wait says so in its result instead of answering). All threads return the same value.
__device__ __forceinline__ bool await_block(const IntraRoundParams &p, IntraLds &L, int slot)
{
    if (threadIdx.x == 0) {
        const unsigned *h = p.handoff + 4 * slot;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
        int ok = 0;
        for (;;) {
            if (__hip_atomic_load(h, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == p.seq) {
                ok = 1;
                break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 50000ull) break;      // half a millisecond
            __builtin_amdgcn_s_sleep(4);
        }
        if (ok)
            for (int k = 1; k < 4; k++) L.hand[k] = __hip_atomic_load(h + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.hand_ok = ok;
    }
    __syncthreads();
    return L.hand_ok != 0;
}

// most probable modes from the modes to the left and above (src/encode.cpp:3980-4003)
__device__ __forceinline__ void most_probable_modes(int left, int top, uint8_t *mpm)
{
    if (left == top) {
        if (left > 1) mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)(((left + 29) % 32) + 2), mpm[2] = (uint8_t)(((left - 1) % 32) + 2);
        else mpm[0] = 0, mpm[1] = 1, mpm[2] = 26;
    } else {
        mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)top;
        mpm[2] = (left && top) ? 0 : (left + top < 2 ? 26 : 1);
    }
}

// one luma PU (src/encode.cpp:4005-4115): 35-mode search on the four wavefronts, prediction of the winner, transform chain
template <int N>
__device__ void intra_pu(const IntraRoundParams &p, IntraLds &L, const IntraJob &j, int job, int part)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    xmi_intra_request rq;
    rq.x0 = j.x0, rq.y0 = j.y0, rq.size = N, rq.valids = j.valids, rq.strong_smoothing = j.strong;
    rq.mpm[0] = j.mpm[0], rq.mpm[1] = j.mpm[1], rq.mpm[2] = j.mpm[2];
    rq.sqrt_lambda = j.sqrt_lambda;
    const IntraParams ip = { p.src[0], j.ref, nullptr, nullptr, p.width, 1 };
    const int parts = j.parts;
    search<N, 4>(ip, rq, 0, L.ref[wave], L.r.win + wave * 1024, lane, wave, L.best_cost, L.best_mode, &L.best, part, parts, &L.cost);
    __syncthreads();
    if (parts > 1) {
        // the workgroups of the PU leave their candidates in device memory; the one that arrives last picks the winner -
        // smallest cost, smallest mode among equals, i.e. the first minimum in mode order - and goes on alone
        IntraPartial *mine = p.partials + job * kIntraParts;
        if (tid == 0) {
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(&mine[part].cost), __builtin_bit_cast(unsigned long long, L.cost), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[part].mode, L.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            const unsigned before = atomicAdd(p.arrived + job, 1u);
            L.last = before == (unsigned)parts - 1;
            if (L.last) {
                __threadfence();
                double bc = 1.7976931348623157e308;
                int bm = 64;
                for (int g = 0; g < parts; g++) {
                    const double c = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<unsigned long long *>(&mine[g].cost), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const int m = __hip_atomic_load(&mine[g].mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (c < bc || (c == bc && m < bm)) bc = c, bm = m;
                }
                L.best = bm;
                __hip_atomic_store(p.arrived + job, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next round
            }
        }
        __syncthreads();
        if (!L.last) return;
    }
    const int mode = L.best;
    const uint8_t *ref = ((k_filter_bits[LG - 2] >> mode) & 1) ? L.ref[0] + 136 : L.ref[0];
    int dc;
    {
        int s = 0;
        for (int i = lane; i < 2 * N; i += 64) s += i < N ? L.ref[0][2 * N - 1 - i] : L.ref[0][2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) s += __shfl_xor(s, k);
        dc = (s + N) >> (LG + 1);
    }
    uint8_t *pr = L.r.win + 4096;
    for (int e = tid; e < N * N; e += 256) pr[e] = (uint8_t)pred_sample<N>(ref, mode, e / N, e % N, dc, N != 32);
    __syncthreads();
    uint32_t sum, ssd;
    // (the source block is in LDS since the search: wavefront 0's copy, rows N apart - no second trip to memory for it)
    const bool hands_on = N == 8 && j.publish != 0;
    chain_unit<N, true, N == 4>(L.r, L.r.win, N, pr, N, j.qp, reinterpret_cast<int16_t *>(p.out + j.out_level), p.out + j.out_rec, nullptr, N, sum, ssd,
                                hands_on ? L.area[0] : nullptr, 8);
    if (tid == 0) {
        IntraHead *h = reinterpret_cast<IntraHead *>(p.out + j.out_head);
        h->best_mode = mode, h->sum = sum, h->ssd = ssd;
    }
    if (hands_on) publish_block(p, j.publish - 1, L.area[0], mode, mode);
}

// chroma of one PU (src/encode.cpp:4306-4411): candidate search on one wavefront, then prediction and chain of U and V
template <int N>
__device__ void intra_chroma_pu(const IntraRoundParams &p, IntraLds &L, const IntraChromaJob &j)
{
    constexpr int LG = N == 4 ? 2 : N == 8 ? 3 : 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wc = p.width / 2;
    uint8_t (*s_ref)[136] = reinterpret_cast<uint8_t (*)[136]>(&L.ref[0][0]);
    uint8_t (*s_src)[256] = reinterpret_cast<uint8_t (*)[256]>(L.r.win);
    if (wave == 0) {
        for (int i = lane; i < 2 * (4 * N + 1); i += 64) {
            const int pl = i >= 4 * N + 1, k = i - pl * (4 * N + 1);
            s_ref[pl][k] = j.valids ? j.ref[pl][k] : (uint8_t)0x80;
        }
        __builtin_amdgcn_wave_barrier();
        if (j.valids) {
            pad_ref<N>(s_ref[0], j.valids, lane);
            pad_ref<N>(s_ref[1], j.valids, lane);
        }
        for (int e = lane; e < 2 * N * N / 4; e += 64) {
            const int pl = e >= N * N / 4, k = e - pl * (N * N / 4), y = k / (N / 4), x4 = k % (N / 4);
            reinterpret_cast<uint32_t *>(s_src[pl])[k] = *reinterpret_cast<const uint32_t *>(p.src[1 + pl] + (size_t)(j.y0 + y) * wc + j.x0 + 4 * x4);
        }
        __builtin_amdgcn_wave_barrier();
        chroma_candidates<N>(j.luma_mode, s_ref, s_src, L.satd, lane, L.cres);
    }
    __syncthreads();
    const int mode = L.cres.best_mode;
    IntraChromaHead *h = reinterpret_cast<IntraChromaHead *>(p.out + j.out_head);
    uint8_t *pr = L.r.win + 4096;
    for (int pl = 0; pl < 2; pl++) {
        int s = 0;
        for (int i = lane; i < 2 * N; i += 64) s += i < N ? s_ref[pl][2 * N - 1 - i] : s_ref[pl][2 * N + 1 + (i - N)];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) s += __shfl_xor(s, k);
        const int dc = (s + N) >> (LG + 1);
        for (int e = tid; e < N * N; e += 256) pr[e] = (uint8_t)pred_sample<N>(s_ref[pl], mode, e / N, e % N, dc, false);
        __syncthreads();
        uint32_t sum, ssd;
        chain_unit<N, true, false>(L.r, p.src[1 + pl] + (size_t)j.y0 * wc + j.x0, wc, pr, N, j.qpc, reinterpret_cast<int16_t *>(p.out + j.out_level[pl]),
                                   p.out + j.out_rec[pl], nullptr, N, sum, ssd);
        if (tid == 0) h->sum[pl] = sum;
    }
    if (tid == 0) h->best_idx = L.cres.best_idx;
}

// DCT matrix of 4x4 blocks (xSubDCT4, src/pixel.cpp:1415)
__constant__ int8_t k_dct4[4][4] = { { 64, 64, 64, 64 }, { 83, 36, -36, -83 }, { 64, -64, -64, 64 }, { 36, -83, 83, -36 } };

// xSubDctQuantRec of one 4x4 unit on ONE wavefront (lanes 0..15 own a sample / coefficient each, no workgroup barrier), I-slice
// rounding: the stages of chain_unit<4> with the operands in 16-entry LDS arrays. M: the transform matrix [k][j]; src: the
// source block (LDS); pred: 16 packed samples (LDS). Returns sum |level| and the SSD of the reconstruction in every lane.
__device__ __forceinline__ void chain4_wave(const int16_t *M, int16_t *A, int16_t *B, const uint8_t *src, int src_stride, const uint8_t *pred, int qp,
                                            int16_t *level, uint8_t *rec, int out_stride, uint8_t *rec_copy, int copy_stride, uint32_t &sum, uint32_t &ssd, int lane)
{
    const bool on = lane < 16;
    const int a = (lane >> 2) & 3, b = lane & 3;
    const int s = src[a * src_stride + b], p = pred[a * 4 + b];
    if (on) A[a * 4 + b] = (int16_t)(s - p);
    __builtin_amdgcn_wave_barrier();
    int v = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) v += (int)M[a * 4 + c] * (int)A[b * 4 + c];
    if (on) B[a * 4 + b] = (int16_t)((v + 1) >> 1);
    __builtin_amdgcn_wave_barrier();
    v = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) v += (int)M[a * 4 + c] * (int)B[b * 4 + c];
    const int coef = (int16_t)((v + 128) >> 8);      // vertical frequency a, horizontal frequency b
    const int qbits = 19 + qp / 6, q = k_quant_scales[qp % 6], qrnd = 171 << (qbits - 9), dscale = k_inv_quant_scales[qp % 6] << (qp / 6);
    const int mag = (abs(coef) * q + qrnd) >> qbits;
    const int lv = clip16(coef < 0 ? -mag : mag);
    __builtin_amdgcn_wave_barrier();
    if (on) {
        level[a * out_stride + b] = (int16_t)lv;
        A[b * 4 + a] = (int16_t)clip16((lv * dscale + 1) >> 1);      // dequantised, transposed
    }
    __builtin_amdgcn_wave_barrier();
    v = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) v += (int)M[c * 4 + a] * (int)A[b * 4 + c];
    if (on) B[a * 4 + b] = (int16_t)clip16((v + 64) >> 7);
    __builtin_amdgcn_wave_barrier();
    v = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) v += (int)M[c * 4 + a] * (int)B[b * 4 + c];
    if (on) A[a * 4 + b] = (int16_t)clip16((v + 2048) >> 12);           // column a, row b
    __builtin_amdgcn_wave_barrier();
    const int r = min(255, max(0, (int)A[b * 4 + a] + p));                // row a, column b
    if (on) {
        rec[a * out_stride + b] = (uint8_t)r;
        rec_copy[a * copy_stride + b] = (uint8_t)r;
    }
    uint32_t m = on ? (uint32_t)mag : 0u, d2 = on ? (uint32_t)((s - r) * (s - r)) : 0u;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m += __shfl_xor(m, o), d2 += __shfl_xor(d2, o);
    sum = __shfl(m, 0), ssd = __shfl(d2, 0);
    __builtin_amdgcn_wave_barrier();
}

// IntraQuadEval on one wavefront: PU k sees the reconstruction and the modes of the PUs before it. The 8x8 source block and
// everything the four PUs exchange stay in LDS; nothing in the loop waits for memory or for another wavefront.
#ifdef XMI_EXP_ROUND_TIMING
__device__ unsigned long long g_quad_ticks[6];      // set-up, per PU: neighbours, search, prediction, chain, head; count
#define XMI_QSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define XMI_QSTAMP(var)
#endif
__device__ void intra_quad(const IntraRoundParams &p, IntraLds &L, const IntraQuadJob &j)
{
    // The four PUs follow each other on wavefront 0 (neighbours, prediction, transform chain: a few dozen samples each); the
    // 35-mode search, nine tenths of the time when one wavefront does it with a lane per mode, is spread over the whole
    // workgroup: a thread per (mode, row of the block).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    XMI_QSTAMP(q0);
    if (tid < 33) L.outer[0][tid] = j.ref[tid];
    if (tid < 16) {
        reinterpret_cast<uint32_t *>(L.r.pix)[tid] = *reinterpret_cast<const uint32_t *>(p.src[0] + (size_t)(j.y0 + (tid >> 1)) * p.width + j.x0 + (tid & 1) * 4);
        L.r.mat[0][tid] = k_dst4[tid >> 2][tid & 3];
    }
    IntraQuadHead *h = reinterpret_cast<IntraQuadHead *>(p.out + j.out_head);
    int16_t *level = reinterpret_cast<int16_t *>(p.out + j.out_level);
    uint8_t *rec = p.out + j.out_rec;
    __syncthreads();
#ifdef XMI_EXP_ROUND_TIMING
    if (tid == 0) atomicAdd(&g_quad_ticks[0], __builtin_amdgcn_s_memrealtime() - q0), atomicAdd(&g_quad_ticks[5], 1ull);
#endif
    uint8_t *ref = L.ref[0];
    // The thread's share of every search: one row of one mode. Which neighbouring samples an angular mode reads for a position,
    // and with what weight, depends on neither the PU nor the samples (xEncIntraPred: two samples of the main side - or of the
    // other side projected onto it - blended by the fraction of (k + 1) angle): worked out once, as offsets from the top-left
    // sample, for the thread's four positions. Bits 0..7 / 8..15: the two offsets (signed), 16..20: fraction, 21: the edge filter
    // of the pure horizontal / vertical mode applies, 24..31: its sample's offset.
    const int my_mode = tid >> 2, my_row = tid & 3;
    uint32_t tap[4] = { 0, 0, 0, 0 };
    if (my_mode > 1 && my_mode < 35) {
        const int angle = k_angle[my_mode], inv = k_inv_angle[my_mode];
        const bool hor = my_mode < 18;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int kk = hor ? x : my_row, xx = hor ? my_row : x;
            const int delta = (kk + 1) * angle, idx = delta >> 5, fact = delta & 31, pos = xx + idx + 1;
            auto off = [&](int q) {
                if (q < 0) q = -((128 - q * inv) >> 8);       // projected side samples
                return hor ? -q : q;
            };
            const bool edge = angle == 0 && xx == 0;
            const int eo = hor ? kk + 1 : -(kk + 1);
            tap[x] = (uint32_t)(off(pos) & 255) | ((uint32_t)(off(pos + 1) & 255) << 8) | ((uint32_t)fact << 16) | ((uint32_t)edge << 21) | ((uint32_t)(eo & 255) << 24);
        }
    }
    for (int k = 0; k < 4; k++) {
        const int ox = (k & 1) * 4, oy = (k >> 1) * 4;
        XMI_QSTAMP(q1);
        // ---- neighbouring samples of the PU (17, never smoothed at this size), padded; their DC value ----
        if (wave == 0) {
            const unsigned valids = j.valids[k] & 0x1fu;
            if (lane < 8) ref[lane] = valids ? quad_sample(L.outer[0], L.area[0], oy + 7 - lane, ox - 1) : (uint8_t)0x80;
            else if (lane < 17) ref[lane] = valids ? quad_sample(L.outer[0], L.area[0], oy - 1, ox - 1 + (lane - 8)) : (uint8_t)0x80;
            __builtin_amdgcn_wave_barrier();
            if (valids) pad_ref<4>(ref, valids, lane);
            __builtin_amdgcn_wave_barrier();
            int sdc = lane < 8 ? (lane < 4 ? ref[7 - lane] : ref[9 + (lane - 4)]) : 0;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) sdc += __shfl_xor(sdc, m);
            if (lane == 0) L.last = (sdc + 4) >> 3;
        }
        __syncthreads();
        XMI_QSTAMP(q2);
        // ---- the search: SATD (4x4 Hadamard, halved) + {1, 2, 3} sqrt(lambda) by most-probable-mode rank, first smallest ----
        const int dc = L.last;
        {
            const int left = (k & 1) ? L.qmode[k - 1] : j.left_mode[k >> 1], top = (k >> 1) ? L.qmode[k - 2] : j.top_mode[k & 1];
            int mpm0, mpm1, mpm2;      // most probable modes (src/encode.cpp:3980-4003)
            if (left == top) {
                if (left > 1) mpm0 = left, mpm1 = ((left + 29) % 32) + 2, mpm2 = ((left - 1) % 32) + 2;
                else mpm0 = 0, mpm1 = 1, mpm2 = 26;
            } else {
                mpm0 = left, mpm1 = top;
                mpm2 = (left && top) ? 0 : (left + top < 2 ? 26 : 1);
            }
            const int mode = my_mode, row = my_row;
            int hd[4] = { 0, 0, 0, 0 };
            if (mode < 35) {
                int d[4];
                uint32_t pv = 0;
                const uint8_t *tl = ref + 8;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    int v;
                    if (mode < 2) {
                        v = pred_sample<4>(ref, mode, 0, 0, row, x, dc, true);
                    } else {
                        const uint32_t t = tap[x];
                        const int f = (int)((t >> 16) & 31);
                        v = ((32 - f) * (int)tl[(int)(int8_t)(t & 255)] + f * (int)tl[(int)(int8_t)((t >> 8) & 255)] + 16) >> 5;
                        if (t & (1u << 21)) v = min(255, max(0, v + (((int)tl[(int)(int8_t)(t >> 24)] - (int)tl[0]) >> 1)));
                    }
                    d[x] = (int)L.r.pix[(oy + row) * 8 + ox + x] - v;
                    pv |= (uint32_t)v << (8 * x);
                }
                L.qpred[mode * 4 + row] = pv;      // the winner's rows are the PU's prediction
                const int a0 = d[0] + d[1], a1 = d[0] - d[1], a2 = d[2] + d[3], a3 = d[2] - d[3];
                hd[0] = a0 + a2, hd[1] = a1 + a3, hd[2] = a0 - a2, hd[3] = a1 - a3;
            }
            // the other direction across the four threads of the mode (signs and order of the outputs do not matter to the sum)
            int sum = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {      // (the four threads of a mode are a DPP quad)
                int v = hd[c];
                const int o1 = __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
                v = (row & 1) ? o1 - v : v + o1;
                const int o2 = __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, true);      // quad_perm:[2,3,0,1]
                v = (row & 2) ? o2 - v : v + o2;
                sum += abs(v);
            }
            sum += __builtin_amdgcn_mov_dpp(sum, 0xb1, 0xf, 0xf, true);
            sum += __builtin_amdgcn_mov_dpp(sum, 0x4e, 0xf, 0xf, true);
            sum >>= 1;
            if (mode < 35 && row == 0) {
                const double bits = mode == mpm0 ? 1.0 : (mode == mpm1 || mode == mpm2) ? 2.0 : 3.0;
                L.qcost[mode] = bits * j.sqrt_lambda + (double)(uint32_t)sum;
            }
        }
        __syncthreads();
        XMI_QSTAMP(q3);
        if (wave == 0) {
            // first minimum in mode order: lane m holds mode m's cost; butterfly over (cost, mode) pairs
            double best_c = lane < 35 ? L.qcost[lane] : 1.7976931348623157e308;
            int mode = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double oc = __shfl_xor(best_c, o);
                const int om = __shfl_xor(mode, o);
                if (oc < best_c || (oc == best_c && om < mode)) best_c = oc, mode = om;
            }
            mode = __builtin_amdgcn_readfirstlane(mode);
            if (lane < 4) reinterpret_cast<uint32_t *>(L.r.pred)[lane] = L.qpred[mode * 4 + lane];
            __builtin_amdgcn_wave_barrier();
            XMI_QSTAMP(q4);
            uint32_t sum, ssd;
            chain4_wave(L.r.mat[0], L.r.a, L.r.b, L.r.pix + oy * 8 + ox, 8, L.r.pred, j.qp, level + oy * 8 + ox, rec + oy * 8 + ox, 8, L.area[0] + oy * 8 + ox, 8, sum, ssd, lane);
            XMI_QSTAMP(q5);
            if (lane == 0) {
                h->best_mode[k] = mode, h->sum[k] = sum, h->ssd[k] = ssd;
                L.qmode[k] = mode;
            }
            __builtin_amdgcn_wave_barrier();
#ifdef XMI_EXP_ROUND_TIMING
            if (lane == 0) {
                atomicAdd(&g_quad_ticks[1], q2 - q1), atomicAdd(&g_quad_ticks[2], q3 - q2), atomicAdd(&g_quad_ticks[3], q4 - q3);
                atomicAdd(&g_quad_ticks[4], q5 - q4);
            }
#endif
        }
        // (the other wavefronts wait at the next PU's first barrier until wavefront 0 has its neighbours ready)
    }
    if (j.publish) publish_block(p, j.publish - 1, L.area[0], L.qmode[1], L.qmode[3]);
}

// IntraChromaQuadEval on one wavefront
__device__ void intra_chroma_quad(const IntraRoundParams &p, IntraLds &L, const IntraChromaQuadJob &j)
{
    const int lane = threadIdx.x & 63, wc = p.width / 2;
    if (threadIdx.x >= 64) return;
    for (int i = lane; i < 66; i += 64) L.outer[i / 33][i % 33] = j.ref[i / 33][i % 33];
    if (lane < 32) {      // the 8x8 source blocks of U and V
        const int pl = lane >> 4, i = lane & 15;
        reinterpret_cast<uint32_t *>(L.r.pix)[lane] = *reinterpret_cast<const uint32_t *>(p.src[1 + pl] + (size_t)(j.y0 + (i >> 1)) * wc + j.x0 + (i & 1) * 4);
    }
    if (lane < 16) L.r.mat[0][lane] = k_dct4[lane >> 2][lane & 3];
    IntraChromaQuadHead *h = reinterpret_cast<IntraChromaQuadHead *>(p.out + j.out_head);
    uint8_t (*s_ref)[136] = reinterpret_cast<uint8_t (*)[136]>(&L.ref[0][0]);
    uint8_t (*s_src)[256] = reinterpret_cast<uint8_t (*)[256]>(L.r.win);
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < 4; k++) {
        const int ox = (k & 1) * 4, oy = (k >> 1) * 4;
        const unsigned valids = j.valids[k];
        if (lane < 34) {
            const int pl = lane / 17, i = lane % 17;
            const uint8_t v = i < 8 ? quad_sample(L.outer[pl], L.area[pl], oy + 7 - i, ox - 1) : quad_sample(L.outer[pl], L.area[pl], oy - 1, ox - 1 + (i - 8));
            s_ref[pl][i] = valids ? v : (uint8_t)0x80;
        }
        __builtin_amdgcn_wave_barrier();
        if (valids) {
            pad_ref<4>(s_ref[0], valids, lane);
            pad_ref<4>(s_ref[1], valids, lane);
        }
        if (lane < 8) {
            const int pl = lane >> 2, yy = lane & 3;
            reinterpret_cast<uint32_t *>(s_src[pl])[yy] = *reinterpret_cast<const uint32_t *>(L.r.pix + pl * 64 + (oy + yy) * 8 + ox);
        }
        __builtin_amdgcn_wave_barrier();
        chroma_candidates<4>(j.luma_mode[k], s_ref, s_src, L.satd, lane, L.cres);
        __builtin_amdgcn_wave_barrier();
        const int mode = L.cres.best_mode;
        for (int pl = 0; pl < 2; pl++) {
            int s = lane < 8 ? (lane < 4 ? s_ref[pl][7 - lane] : s_ref[pl][9 + (lane - 4)]) : 0;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) s += __shfl_xor(s, m);
            const int dc = (__shfl(s, 0) + 4) >> 3;
            if (lane < 16) L.r.pred[lane] = (uint8_t)pred_sample<4>(s_ref[pl], mode, lane / 4, lane % 4, dc, false);
            __builtin_amdgcn_wave_barrier();
            uint32_t sum, ssd;
            chain4_wave(L.r.mat[0], L.r.a, L.r.b, L.r.pix + pl * 64 + oy * 8 + ox, 8, L.r.pred, j.qpc, reinterpret_cast<int16_t *>(p.out + j.out_level[pl]) + oy * 8 + ox,
                        p.out + j.out_rec[pl] + oy * 8 + ox, 8, L.area[pl] + oy * 8 + ox, 8, sum, ssd, lane);
            if (lane == 0) h->sum[pl][k] = sum;
        }
        if (lane == 0) h->best_idx[k] = L.cres.best_idx;
        __builtin_amdgcn_wave_barrier();
    }
}

// A job record lives in host memory, and every read of it crosses the bus: the workgroup fetches its record once, all lanes
// at a time, into LDS and works from there.
template <class Job>
__device__ __forceinline__ const Job &fetch_job(const Job *src, uint32_t *s_job)
{
    static_assert(sizeof(Job) % 4 == 0, "job records are copied as dwords");
    for (int i = threadIdx.x; i < (int)(sizeof(Job) / 4); i += 256) s_job[i] = reinterpret_cast<const uint32_t *>(src)[i];
    __syncthreads();
    return *reinterpret_cast<const Job *>(s_job);
}

#ifdef XMI_EXP_ROUND_TIMING
// measurement build: per job type (PU 4/8/16/32, chroma 4/8/16, luma group, chroma group) 100 MHz ticks of the workgroup and count
__device__ unsigned long long g_intra_ticks[9][2];
#endif

__global__ void __launch_bounds__(256) intra_round_kernel(const IntraRoundParams p)
{
    __shared__ IntraLds L;
#ifdef XMI_EXP_ROUND_TIMING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    int job_type = -1;
#define XMI_JOB_TYPE(v) job_type = (v)
#else
#define XMI_JOB_TYPE(v)
#endif
    __shared__ __align__(16) uint32_t s_job[(sizeof(IntraJob) > sizeof(IntraChromaJob) ? sizeof(IntraJob) : sizeof(IntraChromaJob)) / 4 + 4];
    static_assert(sizeof(IntraQuadJob) <= sizeof(IntraJob) && sizeof(IntraChromaQuadJob) <= sizeof(IntraChromaJob), "s_job holds every job type");
    const int b = blockIdx.x;
    const int first_after = p.npu + p.nchroma + p.nquad + p.ncquad;
    if (b >= first_after) {
        // a request for the 8x8 block to the right of a block of this round: that block's right column and modes go into the
        // record, then it is a request like any other
        const int k = b - first_after;
        if (k < p.npu_after) {
            const IntraGroup g = p.groups[p.npu_first + k];
            const IntraJob &j = fetch_job(p.jobs + g.job, s_job);
            XMI_JOB_TYPE(1);
            IntraJob &jw = *reinterpret_cast<IntraJob *>(s_job);
            if (j.size == 8 && j.after && await_block(p, L, j.after - 1)) {
                if (threadIdx.x < 8) jw.ref[15 - threadIdx.x] = (uint8_t)(L.hand[1 + (threadIdx.x >> 2)] >> (8 * (threadIdx.x & 3)));
                if (threadIdx.x == 8) most_probable_modes((int)(L.hand[3] & 255u), jw.top_mode, jw.mpm);
                __syncthreads();
                intra_pu<8>(p, L, j, g.job, g.part);
            } else if (threadIdx.x == 0) {
                reinterpret_cast<IntraHead *>(p.out + j.out_head)->best_mode = -1;
            }
        } else if (k - p.npu_after < p.nquad_after) {
            XMI_JOB_TYPE(7);
            const IntraQuadJob &j = fetch_job(p.qjobs + p.nquad + (k - p.npu_after), s_job);
            IntraQuadJob &jw = *reinterpret_cast<IntraQuadJob *>(s_job);
            if (j.after && await_block(p, L, j.after - 1)) {
                if (threadIdx.x < 8) jw.ref[15 - threadIdx.x] = (uint8_t)(L.hand[1 + (threadIdx.x >> 2)] >> (8 * (threadIdx.x & 3)));
                if (threadIdx.x == 8) jw.left_mode[0] = (uint8_t)(L.hand[3] & 255u), jw.left_mode[1] = (uint8_t)((L.hand[3] >> 8) & 255u);
                __syncthreads();
                intra_quad(p, L, j);
            } else if (threadIdx.x == 0) {
                reinterpret_cast<IntraQuadHead *>(p.out + j.out_head)->best_mode[0] = -1;
            }
        }
    } else if (b < p.npu) {
        const IntraGroup g = p.groups[b];
        const IntraJob &j = fetch_job(p.jobs + g.job, s_job);
        XMI_JOB_TYPE(j.size == 4 ? 0 : j.size == 8 ? 1 : j.size == 16 ? 2 : 3);
        switch (j.size) {
        case 4: intra_pu<4>(p, L, j, g.job, g.part); break;
        case 8: intra_pu<8>(p, L, j, g.job, g.part); break;
        case 16: intra_pu<16>(p, L, j, g.job, g.part); break;
        case 32: intra_pu<32>(p, L, j, g.job, g.part); break;
        default: break;
        }
    } else if (b - p.npu < p.nchroma) {
        const IntraChromaJob &j = fetch_job(p.cjobs + (b - p.npu), s_job);
        XMI_JOB_TYPE(j.size == 4 ? 4 : j.size == 8 ? 5 : 6);
        switch (j.size) {
        case 4: intra_chroma_pu<4>(p, L, j); break;
        case 8: intra_chroma_pu<8>(p, L, j); break;
        case 16: intra_chroma_pu<16>(p, L, j); break;
        default: break;
        }
    } else if (b - p.npu - p.nchroma < p.nquad) {
        XMI_JOB_TYPE(7);
        intra_quad(p, L, fetch_job(p.qjobs + (b - p.npu - p.nchroma), s_job));
    } else if (b - p.npu - p.nchroma - p.nquad < p.ncquad) {
        XMI_JOB_TYPE(8);
        intra_chroma_quad(p, L, fetch_job(p.cqjobs + (b - p.npu - p.nchroma - p.nquad), s_job));
    }
#ifdef XMI_EXP_ROUND_TIMING
    __syncthreads();
    if (threadIdx.x == 0 && job_type >= 0) {
        atomicAdd(&g_intra_ticks[job_type][0], __builtin_amdgcn_s_memrealtime() - t_begin);
        atomicAdd(&g_intra_ticks[job_type][1], 1ull);
    }
#endif
    round_done(p.counter, p.target, p.flag, p.seq);
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int kMaxLanes = 8;

class GpuEvaluator : public Evaluator, public xmi::encoder::FrameOps {
public:
    struct ChromaPlace {
        size_t sum, rec[2], level[2];
    };
    // what one lane of the analysis loop owns: its round trips overlap those of the other lanes
    struct LaneRes {
        hipStream_t st = nullptr;
        unsigned *d_counter = nullptr;
        uint8_t *d_intra = nullptr;                        // per PU job of a round: candidates of its workgroups, arrival counter
        uint8_t *h_jobs = nullptr, *h_out = nullptr;       // pinned, mapped; the kernel reads / writes them in place
        unsigned *h_flag = nullptr;
        uint8_t *g_jobs = nullptr, *g_out = nullptr;       // the same buffers as the GPU addresses them
        unsigned *g_flag = nullptr;
        unsigned issued = 0, seq = 0;
        unsigned issued_early = 0;                         // P rounds: workgroups whose answers are read first
        bool late_pending = false;                         // the last round may still have workgroups under way
        std::vector<ChromaPlace> place;
        std::vector<int> order, qorder;
        std::vector<const void *> slot_owner;      // intra rounds: the blocks that hand their right column on, by slot
        // where a round trip's time goes, nanoseconds (xmi_analysis_counters): table set-up, launch call, wait, result read-back
        uint64_t t_prepare = 0, t_launch = 0, t_wait = 0, t_collect = 0, rounds = 0;
        char err[256] = "";
    };
    int device = 0, w = 0, h = 0, stride = 0, rows = 0, stride_c = 0, rows_c = 0, max_cu = 0, max_chroma = 0, lanes = 0, lane_rows = 0;
    uint8_t *d_src[3] = {}, *d_ref[3] = {}, *d_planes = nullptr;
    LaneRes lane[kMaxLanes];
    size_t jobs_bytes = 0, out_bytes = 0, off_groups = 0, off_cjobs = 0, off_done = 0, intra_scratch = 0;
    hipStream_t st = nullptr;                              // picture set-up
    int cur_qp = 0, cur_qpc = 0;
    bool planes_by_row = std::getenv("XMI_EXP_PLANES_CLASSIC") == nullptr;      // layout of d_planes (interp_planes_launch)
    bool rdoq = false, rdoq_active = false;      // the level decision (rdoq.hip) in the P pictures' transform chains: asked for / in force for the bound picture
    RdoqSet h_rdoq, *d_rdoq = nullptr;
    // P rounds: hand the answers out when the merge paths of the asking CUs and the chroma jobs are through (rows then poll
    // for the rest when they get to it) - for the polling configurations, where latency counts; with sleeping waiters (many
    // encoders on one GPU, the host cores are what is short) the coordinator waits for the whole round
    bool early_answers = false;
    // picture-level passes of the encoder (FrameOps): reconstruction before / after the SAO offsets, decision maps, strength
    // tables, statistics and parameters on the device; one pinned staging buffer for the copies in either direction
    uint8_t *d_rec[3] = {}, *d_sao[3] = {}, *d_maps = nullptr, *d_bs = nullptr, *d_qp = nullptr, *h_stage = nullptr;
    int16_t *d_mv = nullptr;
    xmi_sao_stats *d_stats = nullptr;
    xmi_sao_ctu_param *d_params = nullptr;
    size_t stage_bytes = 0;

    ~GpuEvaluator() override { release(); }

    void release()
    {
        for (int i = 0; i < 3; i++) {
            if (d_src[i]) (void)hipFree(d_src[i]);
            if (d_ref[i]) (void)hipFree(d_ref[i]);
            d_src[i] = d_ref[i] = nullptr;
        }
        if (d_planes) (void)hipFree(d_planes), d_planes = nullptr;
        if (d_rdoq) (void)hipFree(d_rdoq), d_rdoq = nullptr;
        for (int i = 0; i < 3; i++) {
            if (d_rec[i]) (void)hipFree(d_rec[i]);
            if (d_sao[i]) (void)hipFree(d_sao[i]);
            d_rec[i] = d_sao[i] = nullptr;
        }
        if (d_maps) (void)hipFree(d_maps), d_maps = nullptr;
        if (d_bs) (void)hipFree(d_bs), d_bs = nullptr;
        if (d_qp) (void)hipFree(d_qp), d_qp = nullptr;
        if (d_mv) (void)hipFree(d_mv), d_mv = nullptr;
        if (d_stats) (void)hipFree(d_stats), d_stats = nullptr;
        if (d_params) (void)hipFree(d_params), d_params = nullptr;
        if (h_stage) (void)hipHostFree(h_stage), h_stage = nullptr;
        release_lanes();
        if (st) (void)hipStreamDestroy(st), st = nullptr;
    }

    void release_lanes()
    {
        for (LaneRes &l : lane) {
            if (l.st) (void)hipStreamSynchronize(l.st);
            if (l.d_counter) (void)hipFree(l.d_counter), l.d_counter = nullptr;
            if (l.d_intra) (void)hipFree(l.d_intra), l.d_intra = nullptr;
            if (l.h_jobs) (void)hipHostFree(l.h_jobs), l.h_jobs = nullptr;
            if (l.h_out) (void)hipHostFree(l.h_out), l.h_out = nullptr;
            if (l.h_flag) (void)hipHostFree(l.h_flag), l.h_flag = nullptr;
            if (l.st) (void)hipStreamDestroy(l.st), l.st = nullptr;
        }
    }

    int init(int width, int height, int nlanes)
    {
        w = width, h = height;
        XMI_HIP(hipGetDevice(&device));
        stride = w + 2 * kPad, rows = h + 2 * kPad, stride_c = w / 2 + 2 * kPadC, rows_c = h / 2 + 2 * kPadC;
        const size_t plane = (size_t)stride * rows, plane_c = (size_t)stride_c * rows_c;
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[0]), (size_t)w * h));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[1]), (size_t)w * h / 4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_src[2]), (size_t)w * h / 4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[0]), plane));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[1]), plane_c));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_ref[2]), plane_c));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_planes), 16 * plane));
        XMI_HIP(hipStreamCreate(&st));
        XMI_HIP(hipMemsetAsync(d_planes, 0, 16 * plane, st));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_rdoq), sizeof(RdoqSet)));
        {      // the scan tables of the level decision, once per process (one device per process)
            static std::mutex mu;
            static bool uploaded = false;
            std::lock_guard<std::mutex> lk(mu);
            if (!uploaded) {
                const int rc = rdoq_upload_scans();
                if (rc != XMI_OK) return rc;
                uploaded = true;
            }
        }
        XMI_HIP(hipStreamSynchronize(st));
        return set_lanes(nlanes);
    }

    // what the picture-level passes need on top of init()
    int init_frame_ops()
    {
        const size_t ny = (size_t)w * h, n4 = ny / 16, n8 = ny / 64, nctu = (size_t)(w / 64) * (h / 64);
        for (int c = 0; c < 3; c++) {
            XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_rec[c]), c ? ny / 4 : ny));
            XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_sao[c]), c ? ny / 4 : ny));
        }
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_maps), 2 * n4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_bs), 2 * n4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_qp), n4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_mv), n8 * 4));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_stats), nctu * 3 * sizeof(xmi_sao_stats)));
        XMI_HIP(hipMalloc(reinterpret_cast<void **>(&d_params), nctu * 3 * sizeof(xmi_sao_ctu_param)));
        stage_bytes = std::max(ny * 3 / 2 + 2 * n4 + n8 * 4, nctu * 3 * sizeof(xmi_sao_stats)) + 4096;
        XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_stage), stage_bytes, hipHostMallocDefault));
        return XMI_OK;
    }

    int deblock_and_stats(int width, int height, int intra_slice, int qp, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                          const uint8_t *const rec[3], const uint8_t *const org[3], xmi::encoder::SaoStats *stats) override
    {
        static_assert(sizeof(xmi::encoder::SaoStats) == sizeof(xmi_sao_stats), "same layout");
        XMI_ARG(width == w && height == h && d_rec[0], "encoder: picture-level passes were set up for %dx%d", w, h);
        (void)org;      // the source planes are on the device since bind()
        const size_t ny = (size_t)w * h, n4 = ny / 16, n8 = ny / 64, nctu = (size_t)(w / 64) * (h / 64);
        // one staging block: Y, U, V, unit sizes, cbf flags, vectors
        uint8_t *sp = h_stage;
        std::memcpy(sp, rec[0], ny), std::memcpy(sp + ny, rec[1], ny / 4), std::memcpy(sp + ny * 5 / 4, rec[2], ny / 4);
        uint8_t *sm = sp + ny * 3 / 2;
        std::memcpy(sm, unit_log2, n4), std::memcpy(sm + n4, cbf, n4);
        if (!intra_slice) std::memcpy(sm + 2 * n4, mv, n8 * 4);
        XMI_HIP(hipMemcpyAsync(d_rec[0], sp, ny, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_rec[1], sp + ny, ny / 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_rec[2], sp + ny * 5 / 4, ny / 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_maps, sm, 2 * n4, hipMemcpyHostToDevice, st));
        if (!intra_slice) XMI_HIP(hipMemcpyAsync(d_mv, sm + 2 * n4, n8 * 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemsetAsync(d_qp, qp, n4, st));
        int rc = xmi_deblock_strengths_dev(w, h, intra_slice, d_maps, d_maps + n4, intra_slice ? nullptr : d_mv, d_bs, d_bs + n4, st);
        if (rc != XMI_OK) return rc;
        const xmi_deblock_params dp = { w, h, w, w / 2, 0, 0, 0, 0 };
        rc = xmi_deblock_frame_dev(d_rec[0], d_rec[1], d_rec[2], &dp, d_bs, d_bs + n4, d_qp, st);
        if (rc != XMI_OK) return rc;
        const xmi_sao_geom g = { w, h, w, w / 2, w, w / 2 };
        rc = xmi_sao_stats_dev(&g, d_rec[0], d_rec[1], d_rec[2], d_src[0], d_src[1], d_src[2], d_stats, st);
        if (rc != XMI_OK) return rc;
        XMI_HIP(hipMemcpyAsync(h_stage, d_stats, nctu * 3 * sizeof(xmi_sao_stats), hipMemcpyDeviceToHost, st));
        XMI_HIP(hipStreamSynchronize(st));
        std::memcpy(stats, h_stage, nctu * 3 * sizeof(xmi_sao_stats));
        return XMI_OK;
    }

    int sao_apply(int width, int height, const xmi::encoder::SaoCtuParam *const params[3], uint8_t *const out[3]) override
    {
        static_assert(sizeof(xmi::encoder::SaoCtuParam) == sizeof(xmi_sao_ctu_param), "same layout");
        XMI_ARG(width == w && height == h && d_rec[0], "encoder: picture-level passes were set up for %dx%d", w, h);
        const size_t ny = (size_t)w * h, nctu = (size_t)(w / 64) * (h / 64);
        xmi_sao_ctu_param *hp = reinterpret_cast<xmi_sao_ctu_param *>(h_stage + ny * 3 / 2);
        for (int c = 0; c < 3; c++)
            if (params[c]) std::memcpy(hp + c * nctu, params[c], nctu * sizeof(xmi_sao_ctu_param));
        XMI_HIP(hipMemcpyAsync(d_params, hp, 3 * nctu * sizeof(xmi_sao_ctu_param), hipMemcpyHostToDevice, st));
        const size_t off[3] = { 0, ny, ny * 5 / 4 };
        for (int c = 0; c < 3; c++) {
            const int pw = c ? w / 2 : w, ph = c ? h / 2 : h;
            const uint8_t *from = d_rec[c];
            if (params[c]) {
                const int rc = xmi_sao_apply_plane_dev(d_rec[c], pw, d_sao[c], pw, pw, ph, c ? 32 : 64, d_params + c * nctu, st);
                if (rc != XMI_OK) return rc;
                from = d_sao[c];
            }
            XMI_HIP(hipMemcpyAsync(h_stage + off[c], from, (size_t)pw * ph, hipMemcpyDeviceToHost, st));
        }
        XMI_HIP(hipStreamSynchronize(st));
        for (int c = 0; c < 3; c++) std::memcpy(out[c], h_stage + off[c], c ? ny / 4 : ny);
        return XMI_OK;
    }

    // (re)creates what the lanes own; rows are dealt to lanes round robin, so a lane serves at most ceil(rows / lanes) of them
    int set_lanes(int nlanes)
    {
        XMI_ARG(nlanes >= 1 && nlanes <= kMaxLanes, "analysis: %d lanes (1..%d)", nlanes, kMaxLanes);
        release_lanes();
        lanes = nlanes;
        lane_rows = h / 64;                   // every lane may be handed the requests of all rows
        max_cu = 85 * lane_rows;              // per CTU row and round at most every CU of a CTU: 64 + 16 + 4 + 1
        max_chroma = lane_rows * 64 * 2;      // per CTU at most 64 CUs of 8x8, two planes
        off_groups = align_up(std::max(sizeof(LumaJob), sizeof(IntraJob)) * max_cu);
        off_cjobs = off_groups + align_up(sizeof(IntraGroup) * max_cu * 8);
        jobs_bytes = off_cjobs + align_up(std::max(sizeof(ChromaJob) * max_chroma, (sizeof(IntraChromaJob) + 3 * sizeof(IntraQuadJob) + sizeof(IntraChromaQuadJob)) * (size_t)lane_rows));
        // per CU: head, prediction, two reconstructions, two level blocks; per CTU of chroma: for each of up to 64 CUs the
        // sums and four blocks, each rounded up to the 256-byte grid of the area
        size_t cu_bytes = 0;
        for (int n = 8, count = 64; n <= 64; n *= 2, count /= 4)
            cu_bytes += (size_t)count * (align_up(sizeof(LumaHead)) + 3 * align_up((size_t)n * n) + 2 * align_up(2 * (size_t)n * n));
        out_bytes = (size_t)lane_rows * cu_bytes + (size_t)lane_rows * (64 * 5 * 256 + 2 * 1024 + 2 * 2048) + 4096;
        // the completion words of the P rounds, eight per CU request, behind the blocks: this part only ever holds numbers of
        // rounds, so a word equal to the current round's number was written in this round and nothing has to be cleared
        off_done = out_bytes;
        out_bytes += align_up(32 * (size_t)max_cu);
        const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
        for (int i = 0; i < lanes; i++) {
            LaneRes &l = lane[i];
            XMI_HIP(hipMalloc(reinterpret_cast<void **>(&l.d_counter), 256));
            intra_scratch = align_up(sizeof(IntraPartial) * kIntraParts * max_cu);
            // partial winners and arrival counters of the PUs that several workgroups search, then the hand-over slots
            const size_t intra_bytes = intra_scratch + align_up(4 * (size_t)max_cu) + align_up(16 * 4 * (size_t)lane_rows);
            XMI_HIP(hipMalloc(reinterpret_cast<void **>(&l.d_intra), intra_bytes));
            XMI_HIP(hipMemsetAsync(l.d_intra, 0, intra_bytes, st));
            XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&l.h_jobs), jobs_bytes, flags));
            XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&l.h_out), out_bytes, flags));
            XMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&l.h_flag), 256, flags));
            XMI_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&l.g_jobs), l.h_jobs, 0));
            XMI_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&l.g_out), l.h_out, 0));
            XMI_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&l.g_flag), l.h_flag, 0));
            std::memset(l.h_out, 0, out_bytes);
            std::memset(l.h_flag, 0, 256);
            XMI_HIP(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
            XMI_HIP(hipMemsetAsync(l.d_counter, 0, 256, st));
            l.issued = l.seq = l.issued_early = 0;
            l.late_pending = false;
            l.err[0] = 0;
        }
        XMI_HIP(hipStreamSynchronize(st));
        return XMI_OK;
    }

    // source planes (w x h, w/2 x h/2) and, for a P picture, padded reference planes from host memory, of which the
    // quarter-pel planes are built
    int bind(const uint8_t *const src[3], const uint8_t *const ref[3], int qp, int qpc)
    {
        for (int i = 0; i < lanes; i++) {      // search paths nobody waited for may still read the planes of the last picture
            const int rc = settle(lane[i]);
            if (rc != XMI_OK) return rc;
        }
        XMI_HIP(hipMemcpyAsync(d_src[0], src[0], (size_t)w * h, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_src[1], src[1], (size_t)w * h / 4, hipMemcpyHostToDevice, st));
        XMI_HIP(hipMemcpyAsync(d_src[2], src[2], (size_t)w * h / 4, hipMemcpyHostToDevice, st));
        if (ref[0]) {
            XMI_HIP(hipMemcpyAsync(d_ref[0], ref[0], (size_t)stride * rows, hipMemcpyHostToDevice, st));
            XMI_HIP(hipMemcpyAsync(d_ref[1], ref[1], (size_t)stride_c * rows_c, hipMemcpyHostToDevice, st));
            XMI_HIP(hipMemcpyAsync(d_ref[2], ref[2], (size_t)stride_c * rows_c, hipMemcpyHostToDevice, st));
            int rc = xmi::interp_planes_launch(d_ref[0], stride, rows, stride, d_planes, planes_by_row, st);
            if (rc != XMI_OK) return rc;
        }
        cur_qp = qp, cur_qpc = qpc;
        rdoq_active = false;
        if (rdoq && ref[0]) {      // the level decision serves the transform units of P pictures
            // tables from the states the coder ended the previous picture with (the contexts the slice starts with when there is
            // none), the Lagrange multiplier of the decision loop (src/encode.cpp:5406)
            uint8_t ctx[256];
            if (have_coder_ctx) std::memcpy(ctx, coder_ctx, 256);
            else xmi::analysis::init_contexts(1, qp, ctx);
            const double lambda = 0.578 * std::pow(2.0, (qp - 12) / 3.0);
            for (int lg = 2; lg <= 5; lg++)
                for (int comp = 0; comp < 2; comp++) {
                    int rc = xmi_rdoq_est_bits(ctx, lg, comp == 0, &h_rdoq.tables[lg - 2][comp]);
                    if (rc == XMI_OK) rc = xmi_rdoq_lambda_of(lambda, comp ? qpc : qp, lg, &h_rdoq.lambda[lg - 2][comp]);
                    if (rc != XMI_OK) return rc;
                }
            XMI_HIP(hipMemcpyAsync(d_rdoq, &h_rdoq, sizeof h_rdoq, hipMemcpyHostToDevice, st));
            rdoq_active = true;
        }
        XMI_HIP(hipStreamSynchronize(st));
        return XMI_OK;
    }

    uint8_t coder_ctx[256];
    bool have_coder_ctx = false;
    void coder_contexts(const uint8_t *ctx) override
    {
        have_coder_ctx = ctx != nullptr;
        if (ctx) std::memcpy(coder_ctx, ctx, 256);
    }

    int bind_pictures(const uint8_t *const src[3], const uint8_t *const ref[3], int pad, int qp) override
    {
        XMI_ARG(pad == kPad, "analysis: the reference must be padded by %d luma samples (got %d)", kPad, pad);
        return bind(src, ref, qp, chroma_qp(qp));
    }

    int evaluate_intra(int ln, const IntraBatch &bt) override
    {
        if (ln < 0 || ln >= lanes) return XMI_ERR_ARG;
        thread_local int bound = -1;
        if (bound != device) {
            if (hipSetDevice(device) != hipSuccess) {
                snprintf(lane[ln].err, sizeof lane[ln].err, "hipSetDevice(%d) failed on a lane thread", device);
                return XMI_ERR_HIP;
            }
            bound = device;
        }
        const int rc = run_intra(lane[ln], bt);
        if (rc != XMI_OK) snprintf(lane[ln].err, sizeof lane[ln].err, "%s", xmi_last_error());
        return rc;
    }

    int evaluate(int ln, CuEval *const *cu, int ncu, ChromaEval *const *ch, int nch) override
    {
        if (ln < 0 || ln >= lanes) return XMI_ERR_ARG;
        // the lanes run on threads of the analysis loop, which start with the default device current
        thread_local int bound = -1;
        if (bound != device) {
            if (hipSetDevice(device) != hipSuccess) {
                snprintf(lane[ln].err, sizeof lane[ln].err, "hipSetDevice(%d) failed on a lane thread", device);
                return XMI_ERR_HIP;
            }
            bound = device;
        }
        const int rc = run(lane[ln], cu, ncu, ch, nch);
        if (rc != XMI_OK) snprintf(lane[ln].err, sizeof lane[ln].err, "%s", xmi_last_error());
        return rc;
    }

    static void take_merge(CuEval &q, const LumaHead &hd)
    {
        q.best_merge = hd.best_merge, q.ssd_skip = hd.ssd_skip;
        q.ssd_merge = 0;
        for (int u = 0; u < 4; u++) {
            const bool used = u < (q.size == 64 ? 4 : 1);
            q.sum_merge[u] = used ? hd.sum_merge[u] : 0;
            q.ssd_merge += used ? hd.ssd_merge[u] : 0;
        }
    }

    static void take_search(CuEval &q, const LumaHead &hd)
    {
        q.mv_me = { hd.mv[0], hd.mv[1] };
        q.ssd_me = 0;
        for (int u = 0; u < 4; u++) {
            const bool used = u < (q.size == 64 ? 4 : 1);
            q.sum_me[u] = used ? hd.sum_me[u] : 0;
            q.ssd_me += used ? hd.ssd_me[u] : 0;
        }
    }

    // waits for the workgroups of one path of a CU (they started with the round: a wait of microseconds at most)
    int await_path(const CuEval &q, int path)
    {
        volatile const uint32_t *done = q.answer_done + 4 * path;
        const uint32_t seq = q.answer_round;
        const int units = q.size == 64 ? 4 : 1;
        auto all = [&] {
            for (int u = 0; u < units; u++)
                if (done[u] != seq) return false;
            return true;
        };
        for (unsigned spins = 0; !all(); spins++) {
            if ((spins & 0xfffff) == 0xfffff) {      // a failed kernel must not hang the row
                (void)hipSetDevice(device);          // (a row's thread has made no HIP call before)
                const hipError_t e = hipStreamQuery(lane[q.answer_lane].st);
                if (e != hipErrorNotReady && !all()) {
                    xmi::set_error("analysis: the workgroups of the CU at (%d,%d) never reported completion (%s)", q.x0, q.y0, hipGetErrorString(e));
                    return XMI_ERR_HIP;
                }
            }
            // (with several encoders on one GPU the wait can be that of a full device, and the core has other threads to run)
            if ((spins & 255) == 255) std::this_thread::yield();
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return XMI_OK;
    }

    // Both are called from the row's thread while the lane's answers are still the row's to read.
    int finish_merge(CuEval &q) override
    {
        if (!q.merge_pending) return XMI_OK;
        const LumaHead &hd = *static_cast<const LumaHead *>(q.merge_pending);
        const int rc = await_path(q, 0);
        if (rc != XMI_OK) return rc;
        take_merge(q, hd);
        q.merge_pending = nullptr;
        return XMI_OK;
    }

    int finish_search(CuEval &q) override
    {
        if (!q.search_pending) return XMI_OK;
        const LumaHead &hd = *static_cast<const LumaHead *>(q.search_pending);
        const int rc = await_path(q, 1);
        if (rc != XMI_OK) return rc;
        take_search(q, hd);
        q.search_pending = nullptr;
        return XMI_OK;
    }

    const char *first_error() const
    {
        for (int i = 0; i < lanes; i++)
            if (lane[i].err[0]) return lane[i].err;
        return "";
    }

private:
    // launch and wait: the host spins on the completion word and looks at the stream now and then, so that a failed launch
    // cannot hang it
    int wait_round(LaneRes &R, unsigned seq, const unsigned *word = nullptr)
    {
        volatile const unsigned *flag = word ? word : R.h_flag;
        for (unsigned spins = 0; *flag != seq; spins++) {
            if ((spins & 0xfffff) == 0xfffff) {
                const hipError_t q = hipStreamQuery(R.st);
                if (q == hipSuccess) {
                    if (*flag == seq) break;
                    xmi::set_error("analysis: the round kernel finished without reporting completion");
                    return XMI_ERR_HIP;
                }
                if (q != hipErrorNotReady) {
                    xmi::set_error("analysis: round kernel failed: %s", hipGetErrorString(q));
                    return XMI_ERR_HIP;
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return XMI_OK;
    }

    // every workgroup of the lane's last round is through: its tables and result area may be written again
    int settle(LaneRes &R)
    {
        if (!R.late_pending) return XMI_OK;
        R.late_pending = false;
        return wait_round(R, R.seq);
    }

    int run_intra(LaneRes &R, const IntraBatch &bt)
    {
        IntraEval *const *pu = bt.pu;
        IntraChromaEval *const *ch = bt.chroma;
        const int npu = bt.npu, nch = bt.nchroma, nq = bt.nquad, ncq = bt.ncquad;
        XMI_ARG(npu <= max_cu && nch <= lane_rows && nq <= 3 * lane_rows && ncq <= lane_rows, "analysis: intra requests (%d, %d, %d, %d) exceed the round capacity", npu, nch, nq,
                ncq);
        const int settled = settle(R);
        if (settled != XMI_OK) return settled;
        const auto t0 = std::chrono::steady_clock::now();
        IntraJob *jobs = reinterpret_cast<IntraJob *>(R.h_jobs);
        IntraGroup *groups = reinterpret_cast<IntraGroup *>(R.h_jobs + off_groups);
        int ngroups_pu = 0;
        IntraChromaJob *cjobs = reinterpret_cast<IntraChromaJob *>(R.h_jobs + off_cjobs);
        IntraQuadJob *qjobs = reinterpret_cast<IntraQuadJob *>(cjobs + lane_rows);
        IntraChromaQuadJob *cqjobs = reinterpret_cast<IntraChromaQuadJob *>(qjobs + 3 * lane_rows);
        // job k serves request order[k]: the requests that stand after a block of this round come last, and the block they
        // stand after gets a hand-over slot
        std::vector<int> &order = R.order, &qorder = R.qorder;
        std::vector<const void *> &owner = R.slot_owner;
        order.clear(), qorder.clear(), owner.clear();
        for (int i = 0; i < npu; i++)
            if (!pu[i]->after_kind) order.push_back(i);
        const int npu_plain = (int)order.size();
        for (int i = 0; i < npu; i++)
            if (pu[i]->after_kind) order.push_back(i);
        for (int i = 0; i < nq; i++)
            if (!bt.quad[i]->after_kind) qorder.push_back(i);
        const int nq_plain = (int)qorder.size();
        for (int i = 0; i < nq; i++)
            if (bt.quad[i]->after_kind) qorder.push_back(i);
        auto slot_of = [&](uint8_t kind, const IntraEval *a, const IntraQuadEval *b) -> int {
            const void *who = kind == 1 ? static_cast<const void *>(a) : static_cast<const void *>(b);
            for (size_t k = 0; k < owner.size(); k++)
                if (owner[k] == who) return (int)k;
            owner.push_back(who);
            return (int)owner.size() - 1;
        };
        auto published = [&](const void *who) -> uint8_t {
            for (size_t k = 0; k < owner.size(); k++)
                if (owner[k] == who) return (uint8_t)(k + 1);
            return 0;
        };
        for (int i = npu_plain; i < npu; i++) {
            const IntraEval &q = *pu[order[i]];
            XMI_ARG((q.after_kind == 1 && q.after_pu) || (q.after_kind == 2 && q.after_quad), "analysis: intra PU after no block");
            (void)slot_of(q.after_kind, q.after_pu, q.after_quad);
        }
        for (int i = nq_plain; i < nq; i++) {
            const IntraQuadEval &q = *bt.quad[qorder[i]];
            XMI_ARG((q.after_kind == 1 && q.after_pu) || (q.after_kind == 2 && q.after_quad), "analysis: intra 4x4 group after no block");
            (void)slot_of(q.after_kind, q.after_pu, q.after_quad);
        }
        XMI_ARG((int)owner.size() <= 4 * lane_rows, "analysis: %zu hand-over slots exceed the round capacity", owner.size());
        size_t n_publishers = 0;
        size_t out = 0;
        int npu_first = 0;
        for (int i = 0; i < npu; i++) {
            const IntraEval &q = *pu[order[i]];
            IntraJob &j = jobs[i];
            if (i == npu_plain) npu_first = ngroups_pu;
            XMI_ARG(q.size == 4 || q.size == 8 || q.size == 16 || q.size == 32, "analysis: intra PU size %d", q.size);
            XMI_ARG(q.x0 >= 0 && q.y0 >= 0 && q.x0 % 4 == 0 && q.y0 % 4 == 0 && q.x0 + q.size <= w && q.y0 + q.size <= h, "analysis: intra PU (%d,%d) size %d outside the picture",
                    q.x0, q.y0, q.size);
            XMI_ARG(q.qp >= 0 && q.qp <= 51 && q.mpm[0] < 35 && q.mpm[1] < 35 && q.mpm[2] < 35 && q.top_mode < 35, "analysis: intra PU parameters out of range");
            XMI_ARG(!q.after_kind || q.size == 8, "analysis: only an 8x8 PU stands after a block");
            j.x0 = q.x0, j.y0 = q.y0, j.size = q.size;
            j.valids = q.valids & 0x1f, j.strong = q.strong;
            j.mpm[0] = q.mpm[0], j.mpm[1] = q.mpm[1], j.mpm[2] = q.mpm[2];
            j.top_mode = q.top_mode;
            j.publish = q.after_kind || q.size != 8 ? 0 : published(&q);
            j.after = q.after_kind ? (uint8_t)(slot_of(q.after_kind, q.after_pu, q.after_quad) + 1) : 0;
            n_publishers += j.publish != 0;
            j.reserved[0] = j.reserved[1] = j.reserved[2] = 0;
            j.qp = q.qp, j.sqrt_lambda = q.sqrt_lambda;
            j.parts = q.size >= 16 ? kIntraParts : 1;      // the 35-mode search of a large PU is spread over several workgroups
            for (int g = 0; g < j.parts; g++) groups[ngroups_pu++] = { (uint16_t)i, (uint8_t)g, 0 };
            std::memcpy(j.ref, q.ref, (size_t)(4 * q.size + 1));
            const size_t px = (size_t)q.size * q.size;
            j.out_head = (uint32_t)out, out = align_up(out + sizeof(IntraHead));
            j.out_rec = (uint32_t)out, out = align_up(out + px);
            j.out_level = (uint32_t)out, out = align_up(out + 2 * px);
        }
        if (npu_plain == npu) npu_first = ngroups_pu;
        const int npu_after_groups = ngroups_pu - npu_first;
        for (int i = 0; i < nch; i++) {
            const IntraChromaEval &c = *ch[i];
            IntraChromaJob &j = cjobs[i];
            XMI_ARG(c.size == 4 || c.size == 8 || c.size == 16, "analysis: intra chroma size %d", c.size);
            XMI_ARG(c.x0 >= 0 && c.y0 >= 0 && c.x0 % 4 == 0 && c.x0 + c.size <= w / 2 && c.y0 + c.size <= h / 2 && c.luma_mode < 35 && c.qpc >= 0 && c.qpc <= 51,
                    "analysis: intra chroma block (%d,%d) size %d outside the picture, or bad parameters", c.x0, c.y0, c.size);
            j.x0 = c.x0, j.y0 = c.y0, j.size = c.size;
            j.valids = c.valids & 0x1f, j.luma_mode = c.luma_mode;
            j.qpc = c.qpc;
            std::memcpy(j.ref[0], c.ref_u, (size_t)(4 * c.size + 1)), std::memcpy(j.ref[1], c.ref_v, (size_t)(4 * c.size + 1));
            const size_t px = (size_t)c.size * c.size;
            j.out_head = (uint32_t)out, out = align_up(out + sizeof(IntraChromaHead));
            for (int k = 0; k < 2; k++) j.out_rec[k] = (uint32_t)out, out = align_up(out + px);
            for (int k = 0; k < 2; k++) j.out_level[k] = (uint32_t)out, out = align_up(out + 2 * px);
        }
        for (int i = 0; i < nq; i++) {
            const IntraQuadEval &q = *bt.quad[qorder[i]];
            IntraQuadJob &j = qjobs[i];
            XMI_ARG(q.x0 >= 0 && q.y0 >= 0 && q.x0 % 8 == 0 && q.y0 % 8 == 0 && q.x0 + 8 <= w && q.y0 + 8 <= h && q.qp >= 0 && q.qp <= 51,
                    "analysis: intra 4x4 group (%d,%d) outside the picture, or bad parameters", q.x0, q.y0);
            j.x0 = q.x0, j.y0 = q.y0;
            for (int k = 0; k < 4; k++) j.valids[k] = q.valids[k] & 0x1f;
            for (int k = 0; k < 2; k++) {
                XMI_ARG(q.left_mode[k] < 35 && q.top_mode[k] < 35, "analysis: intra 4x4 group: neighbour mode out of range");
                j.left_mode[k] = q.left_mode[k], j.top_mode[k] = q.top_mode[k];
            }
            j.publish = q.after_kind ? 0 : published(&q);
            j.after = q.after_kind ? (uint8_t)(slot_of(q.after_kind, q.after_pu, q.after_quad) + 1) : 0;
            n_publishers += j.publish != 0;
            j.reserved = 0;
            j.strong = q.strong, j.qp = q.qp, j.sqrt_lambda = q.sqrt_lambda;
            std::memcpy(j.ref, q.ref, 33);
            j.out_head = (uint32_t)out, out = align_up(out + sizeof(IntraQuadHead));
            j.out_rec = (uint32_t)out, out = align_up(out + 64);
            j.out_level = (uint32_t)out, out = align_up(out + 128);
        }
        // (a block somebody stands after is a request of this round: otherwise its workgroups would wait for nothing)
        XMI_ARG(n_publishers == owner.size(), "analysis: an intra request stands after a block that is not part of the round");
        for (int i = 0; i < ncq; i++) {
            const IntraChromaQuadEval &c = *bt.cquad[i];
            IntraChromaQuadJob &j = cqjobs[i];
            XMI_ARG(c.x0 >= 0 && c.y0 >= 0 && c.x0 % 8 == 0 && c.y0 % 8 == 0 && c.x0 + 8 <= w / 2 && c.y0 + 8 <= h / 2 && c.qpc >= 0 && c.qpc <= 51,
                    "analysis: intra chroma group (%d,%d) outside the picture, or bad parameters", c.x0, c.y0);
            j.x0 = c.x0, j.y0 = c.y0, j.qpc = c.qpc;
            for (int k = 0; k < 4; k++) {
                XMI_ARG(c.luma_mode[k] < 35, "analysis: intra chroma group: luma mode out of range");
                j.valids[k] = c.valids[k] & 0x1f, j.luma_mode[k] = c.luma_mode[k];
            }
            std::memcpy(j.ref[0], c.ref_u, 33), std::memcpy(j.ref[1], c.ref_v, 33);
            j.out_head = (uint32_t)out, out = align_up(out + sizeof(IntraChromaQuadHead));
            for (int k = 0; k < 2; k++) j.out_rec[k] = (uint32_t)out, out = align_up(out + 64);
            for (int k = 0; k < 2; k++) j.out_level[k] = (uint32_t)out, out = align_up(out + 128);
        }
        XMI_ARG(out <= off_done, "analysis: result area too small (%zu of %zu bytes)", out, off_done);
        IntraRoundParams p;
        p.jobs = reinterpret_cast<const IntraJob *>(R.g_jobs);
        p.groups = reinterpret_cast<const IntraGroup *>(R.g_jobs + off_groups);
        p.partials = reinterpret_cast<IntraPartial *>(R.d_intra);
        p.arrived = reinterpret_cast<unsigned *>(R.d_intra + intra_scratch);
        p.cjobs = reinterpret_cast<const IntraChromaJob *>(R.g_jobs + off_cjobs);
        p.qjobs = reinterpret_cast<const IntraQuadJob *>(p.cjobs + lane_rows);
        p.cqjobs = reinterpret_cast<const IntraChromaQuadJob *>(p.qjobs + 3 * lane_rows);
        p.npu = npu_first, p.nchroma = nch, p.nquad = nq_plain, p.ncquad = ncq;
        p.npu_first = npu_first, p.npu_after = npu_after_groups, p.nquad_after = nq - nq_plain;
        p.handoff = reinterpret_cast<unsigned *>(R.d_intra + intra_scratch + align_up(4 * (size_t)max_cu));
        for (int k = 0; k < 3; k++) p.src[k] = d_src[k];
        p.width = w;
        p.out = R.g_out;
        const unsigned ngroups = (unsigned)(ngroups_pu + nch + nq + ncq);
        R.issued += ngroups;
        R.seq++;
        p.counter = R.d_counter, p.target = R.issued, p.flag = R.g_flag, p.seq = R.seq;
        std::atomic_thread_fence(std::memory_order_release);
        const auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(intra_round_kernel, dim3(ngroups), dim3(256), 0, R.st, p);
        XMI_HIP(hipGetLastError());
        const auto t2 = std::chrono::steady_clock::now();
        const int rc = wait_round(R, R.seq);
        if (rc != XMI_OK) return rc;
        const auto t3 = std::chrono::steady_clock::now();
        for (int i = 0; i < npu; i++) {
            IntraEval &q = *pu[order[i]];
            const IntraJob &j = jobs[i];
            const IntraHead &hd = *reinterpret_cast<const IntraHead *>(R.h_out + j.out_head);
            q.best_mode = hd.best_mode, q.sum = hd.sum, q.ssd = hd.ssd;
            q.rec = R.h_out + j.out_rec, q.level = reinterpret_cast<const int16_t *>(R.h_out + j.out_level);
        }
        for (int i = 0; i < nch; i++) {
            IntraChromaEval &c = *ch[i];
            const IntraChromaJob &j = cjobs[i];
            const IntraChromaHead &hd = *reinterpret_cast<const IntraChromaHead *>(R.h_out + j.out_head);
            c.best_idx = hd.best_idx, c.sum_u = hd.sum[0], c.sum_v = hd.sum[1];
            c.rec_u = R.h_out + j.out_rec[0], c.rec_v = R.h_out + j.out_rec[1];
            c.level_u = reinterpret_cast<const int16_t *>(R.h_out + j.out_level[0]), c.level_v = reinterpret_cast<const int16_t *>(R.h_out + j.out_level[1]);
        }
        for (int i = 0; i < nq; i++) {
            IntraQuadEval &q = *bt.quad[qorder[i]];
            const IntraQuadJob &j = qjobs[i];
            const IntraQuadHead &hd = *reinterpret_cast<const IntraQuadHead *>(R.h_out + j.out_head);
            for (int k = 0; k < 4; k++) q.best_mode[k] = hd.best_mode[k], q.sum[k] = hd.sum[k], q.ssd[k] = hd.ssd[k];
            q.rec = R.h_out + j.out_rec, q.level = reinterpret_cast<const int16_t *>(R.h_out + j.out_level);
        }
        // the requests that stood after a block say what they were evaluated with, as the kernel put it in place
        for (int i = npu_plain; i < npu; i++) take_left_block(*pu[order[i]]);
        for (int i = nq_plain; i < nq; i++) take_left_block(*bt.quad[qorder[i]]);
        for (int i = 0; i < ncq; i++) {
            IntraChromaQuadEval &c = *bt.cquad[i];
            const IntraChromaQuadJob &j = cqjobs[i];
            const IntraChromaQuadHead &hd = *reinterpret_cast<const IntraChromaQuadHead *>(R.h_out + j.out_head);
            for (int k = 0; k < 4; k++) c.best_idx[k] = hd.best_idx[k], c.sum_u[k] = hd.sum[0][k], c.sum_v[k] = hd.sum[1][k];
            c.rec_u = R.h_out + j.out_rec[0], c.rec_v = R.h_out + j.out_rec[1];
            c.level_u = reinterpret_cast<const int16_t *>(R.h_out + j.out_level[0]), c.level_v = reinterpret_cast<const int16_t *>(R.h_out + j.out_level[1]);
        }
        const auto t4 = std::chrono::steady_clock::now();
        auto ns = [](auto a, auto b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
        R.t_prepare += ns(t0, t1), R.t_launch += ns(t1, t2), R.t_wait += ns(t2, t3), R.t_collect += ns(t3, t4), R.rounds++;
        return XMI_OK;
    }

    int run(LaneRes &R, CuEval *const *cu, int ncu, ChromaEval *const *ch, int nch)
    {
        XMI_ARG(ncu <= max_cu, "analysis: %d CU requests exceed the round capacity %d", ncu, max_cu);
        const int settled = settle(R);
        if (settled != XMI_OK) return settled;
        const auto t0 = std::chrono::steady_clock::now();
        uint8_t *const h_jobs = R.h_jobs, *const h_out = R.h_out;
        std::vector<ChromaPlace> &place = R.place;
        LumaJob *jobs = reinterpret_cast<LumaJob *>(h_jobs);
        ChromaJob *cjobs = reinterpret_cast<ChromaJob *>(h_jobs + off_cjobs);
        size_t out = 0;
        int nluma = 0, n64 = 0, n_early = 0;
        std::vector<int> &order = R.order;      // job k serves request order[k]: the 64x64 CUs first
        order.resize(ncu);
        for (int i = 0; i < ncu; i++)
            if (cu[i]->size == 64) order[n64++] = i;
        for (int i = 0, k = n64; i < ncu; i++)
            if (cu[i]->size != 64) order[k++] = i;
        for (int i = 0; i < ncu; i++) {
            const CuEval &q = *cu[order[i]];
            LumaJob &j = jobs[i];
            XMI_ARG(q.size == 8 || q.size == 16 || q.size == 32 || q.size == 64, "analysis: CU size %d", q.size);
            XMI_ARG(q.x0 >= 0 && q.y0 >= 0 && q.x0 % 8 == 0 && q.y0 % 8 == 0 && q.x0 + q.size <= w && q.y0 + q.size <= h && q.n_merge >= 0 && q.n_merge <= 5,
                    "analysis: CU (%d,%d) size %d outside the picture", q.x0, q.y0, q.size);
            j.x0 = q.x0, j.y0 = q.y0, j.size = q.size, j.n_merge = q.n_merge;
            for (int k = 0; k < 5; k++) {
                j.merge_mv[k][0] = k < q.n_merge ? q.merge_mv[k].x : 0, j.merge_mv[k][1] = k < q.n_merge ? q.merge_mv[k].y : 0;
                // a candidate the reference's clip would change is left out by the caller; one that would leave the padded
                // planes can only be a caller's mistake and is left out here
                const bool inside = std::abs(j.merge_mv[k][0] >> 2) <= kPad - 8 && std::abs(j.merge_mv[k][1] >> 2) <= kPad - 8;
                j.merge_ok[k] = k < q.n_merge && q.merge_ok[k] && inside;
            }
            j.amvp[0][0] = q.amvp[0].x, j.amvp[0][1] = q.amvp[0].y, j.amvp[1][0] = q.amvp[1].x, j.amvp[1][1] = q.amvp[1].y;
            j.mv_min_ipel[0] = q.mv_min_ipel.x, j.mv_min_ipel[1] = q.mv_min_ipel.y;
            j.mv_max_ipel[0] = q.mv_max_ipel.x, j.mv_max_ipel[1] = q.mv_max_ipel.y;
            j.sqrt_lambda = (uint32_t)std::sqrt(q.lambda);
            // what the batch kernel checks per request (request_in_bounds, me_search.hip): every block the search can read lies in the planes
            xmi_me_request rq = {};
            rq.mvp[0][0] = j.amvp[0][0], rq.mvp[0][1] = j.amvp[0][1], rq.mvp[1][0] = j.amvp[1][0], rq.mvp[1][1] = j.amvp[1][1];
            rq.mv_min_ipel[0] = j.mv_min_ipel[0], rq.mv_min_ipel[1] = j.mv_min_ipel[1], rq.mv_max_ipel[0] = j.mv_max_ipel[0], rq.mv_max_ipel[1] = j.mv_max_ipel[1];
            XMI_ARG(xmi::me_request_reach(rq) + 8 <= kPad, "analysis: the search of the CU at (%d,%d) could leave the padded reference", q.x0, q.y0);
            const size_t px = (size_t)q.size * q.size;
            j.out_head = (uint32_t)out, out = align_up(out + sizeof(LumaHead));
            j.out_done = (uint32_t)(off_done + 32 * (size_t)i);
            j.urgent = q.urgent != 0;
            n_early += j.urgent ? (q.size == 64 ? 4 : 1) : 0;
            j.out_pred_merge = (uint32_t)out, out = align_up(out + px);
            j.out_rec_merge = (uint32_t)out, out = align_up(out + px);
            j.out_rec_me = (uint32_t)out, out = align_up(out + px);
            j.out_level_merge = (uint32_t)out, out = align_up(out + 2 * px);
            j.out_level_me = (uint32_t)out, out = align_up(out + 2 * px);
            nluma += q.size == 64 ? 8 : 2;
        }
        int nc = 0;
        place.resize(nch);
        for (int i = 0; i < nch; i++) {
            const ChromaEval &c = *ch[i];
            XMI_ARG(c.size == 8 || c.size == 16 || c.size == 32 || c.size == 64, "analysis: chroma CU size %d", c.size);
            XMI_ARG(c.x0 >= 0 && c.y0 >= 0 && c.x0 + c.size <= w && c.y0 + c.size <= h, "analysis: chroma CU outside the picture");
            XMI_ARG(std::abs(c.mv.x >> 2) <= kPad - 8 && std::abs(c.mv.y >> 2) <= kPad - 8, "analysis: chroma vector (%d,%d) leaves the padded reference", c.mv.x, c.mv.y);
            const int n = c.size / 2, units = c.size == 64 ? 4 : 1, un = units == 4 ? 16 : n;
            XMI_ARG(nc + 2 * units <= max_chroma, "analysis: chroma units exceed the round capacity %d", max_chroma);
            ChromaPlace &pl = place[i];
            pl.sum = out, out = align_up(out + 4 * 8);
            for (int k = 0; k < 2; k++) pl.rec[k] = out, out = align_up(out + (size_t)n * n);
            for (int k = 0; k < 2; k++) pl.level[k] = out, out = align_up(out + 2 * (size_t)n * n);
            for (int k = 0; k < 2; k++)
                for (int u = 0; u < units; u++) {
                    const int ox = units == 4 ? (u & 1) * 16 : 0, oy = units == 4 ? (u >> 1) * 16 : 0;
                    ChromaJob &j = cjobs[nc++];
                    j.plane = k, j.x = c.x0 / 2 + ox, j.y = c.y0 / 2 + oy, j.n = un;
                    j.mv[0] = c.mv.x, j.mv[1] = c.mv.y;
                    j.skip = c.skip;
                    j.out_sum = (uint32_t)(pl.sum + 4 * (size_t)(k * 4 + u));
                    j.out_rec = (uint32_t)(pl.rec[k] + (size_t)oy * n + ox);
                    j.out_level = (uint32_t)(pl.level[k] + 2 * ((size_t)oy * n + ox));
                    j.rec_stride = n;
                }
        }
        XMI_ARG(out <= off_done, "analysis: result area too small (%zu of %zu bytes)", out, off_done);

        RoundParams p;
        p.jobs = reinterpret_cast<const LumaJob *>(R.g_jobs);
        p.cjobs = reinterpret_cast<const ChromaJob *>(R.g_jobs + off_cjobs);
        p.nluma = nluma, p.nchroma = nc, p.n64 = n64;
        for (int k = 0; k < 3; k++) p.src[k] = d_src[k];
        p.planes = d_planes;
        p.ref_c[0] = d_ref[1] + (size_t)kPadC * stride_c + kPadC, p.ref_c[1] = d_ref[2] + (size_t)kPadC * stride_c + kPadC;
        // the 16 phases of a row lie next to each other: the candidates of a sub-pel stage - up to nine planes - then share
        // their pages and memory rows instead of lying megabytes apart
        p.width = w, p.rows = rows, p.stride_c = stride_c;
        p.stride = planes_by_row ? 16 * stride : stride;
        p.phase_pitch = planes_by_row ? (unsigned)stride : (unsigned)((size_t)stride * rows), p.row_bytes = (unsigned)stride;
        p.qp = cur_qp, p.qpc = cur_qpc;
        p.rdoq = rdoq_active ? d_rdoq : nullptr;
        p.out = R.g_out;
        const unsigned ngroups = (unsigned)(nluma + nc);
        R.issued += ngroups;
        n_early += nc;
        if (early_answers && n_early > 0) R.issued_early += (unsigned)n_early;
        R.seq++;
        const unsigned seq = R.seq;
        p.counter = R.d_counter, p.target = R.issued, p.flag = R.g_flag, p.seq = seq;
        const bool early_return = early_answers && n_early > 0;      // (nothing marked: everything is waited for)
        p.counter_early = R.d_counter + 16, p.target_early = R.issued_early, p.flag_early = early_return ? R.g_flag + 16 : nullptr;
        R.late_pending = early_return;
        std::atomic_thread_fence(std::memory_order_release);       // the job table is complete before the launch
        const auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(analysis_round_kernel, dim3(ngroups), dim3(256), 0, R.st, p);
        XMI_HIP(hipGetLastError());
        const auto t2 = std::chrono::steady_clock::now();
        const int wrc = wait_round(R, seq, early_return ? R.h_flag + 16 : nullptr);
        if (wrc != XMI_OK) return wrc;
        const auto t3 = std::chrono::steady_clock::now();

        for (int i = 0; i < ncu; i++) {
            CuEval &q = *cu[order[i]];
            const LumaJob &j = jobs[i];
            const LumaHead &hd = *reinterpret_cast<const LumaHead *>(h_out + j.out_head);
            // the results themselves come with finish_merge() / finish_search(), when the row gets to them
            q.merge_pending = q.search_pending = &hd, q.answer_round = seq, q.answer_lane = (int)(&R - lane);
            q.answer_done = reinterpret_cast<const uint32_t *>(h_out + j.out_done);
            if (!early_return) take_merge(q, hd), take_search(q, hd), q.merge_pending = q.search_pending = nullptr;
            q.pred_merge = h_out + j.out_pred_merge, q.rec_merge = h_out + j.out_rec_merge, q.rec_me = h_out + j.out_rec_me;
            q.level_merge = reinterpret_cast<const int16_t *>(h_out + j.out_level_merge);
            q.level_me = reinterpret_cast<const int16_t *>(h_out + j.out_level_me);
        }
        for (int i = 0; i < nch; i++) {
            ChromaEval &c = *ch[i];
            const ChromaPlace &pl = place[i];
            const uint32_t *sum = reinterpret_cast<const uint32_t *>(h_out + pl.sum);
            const int units = c.size == 64 ? 4 : 1;
            for (int u = 0; u < 4; u++) c.sum_u[u] = (!c.skip && u < units) ? sum[u] : 0, c.sum_v[u] = (!c.skip && u < units) ? sum[4 + u] : 0;
            c.rec_u = h_out + pl.rec[0], c.rec_v = h_out + pl.rec[1];
            c.level_u = reinterpret_cast<const int16_t *>(h_out + pl.level[0]), c.level_v = reinterpret_cast<const int16_t *>(h_out + pl.level[1]);
        }
        const auto t4 = std::chrono::steady_clock::now();
        auto ns = [](auto a, auto b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
        R.t_prepare += ns(t0, t1), R.t_launch += ns(t1, t2), R.t_wait += ns(t2, t3), R.t_collect += ns(t3, t4), R.rounds++;
#ifdef XMI_EXP_ROUND_TIMING
        {      // calls by their largest CU (8, 16, 32, 64; none = chroma alone): count, time from launch to completion, CU requests
            int cls = 4;
            for (int i = 0; i < ncu; i++) {
                const int c = cu[i]->size == 64 ? 3 : cu[i]->size == 32 ? 2 : cu[i]->size == 16 ? 1 : 0;
                cls = cls == 4 ? c : (c > cls ? c : cls);
            }
            g_round_class[cls][0]++, g_round_class[cls][1] += ns(t1, t3), g_round_class[cls][2] += (uint64_t)ncu;
        }
#endif
        return XMI_OK;
    }
};

}  // namespace

struct xmi_analysis {
    Analyser an;
    GpuEvaluator ev;
    PictureResult res;
    xmi_analysis(int w, int h) : an(w, h) {}
};

extern "C" {

xmi_analysis *xmi_analysis_create(int width, int height)
{
    if (xmi::enter_device() != XMI_OK) return nullptr;
    if (width <= 0 || height <= 0 || width % 64 || height % 64 || width > 8192 || height > 4352) {
        xmi::set_error("xmi_analysis_create: %dx%d: the coded size must be a multiple of 64, at most 8192x4352", width, height);
        return nullptr;
    }
    xmi_analysis *a = new (std::nothrow) xmi_analysis(width, height);
    if (!a) {
        xmi::set_error("xmi_analysis_create: out of host memory");
        return nullptr;
    }
    if (a->ev.init(width, height, a->an.lanes) != XMI_OK) {
        delete a;
        return nullptr;
    }
    return a;
}

void xmi_analysis_destroy(xmi_analysis *a) { delete a; }

int xmi_analysis_workers(xmi_analysis *a, int workers)
{
    XMI_ARG(a, "xmi_analysis_workers: null handle");
    XMI_ARG(workers >= 0 && workers <= 1024, "xmi_analysis_workers: %d threads", workers);
    a->an.workers = workers;
    return XMI_OK;
}

int xmi_analysis_speculate(xmi_analysis *a, int level)
{
    XMI_ARG(a, "xmi_analysis_speculate: null handle");
    XMI_ARG(level >= 0 && level <= 2, "xmi_analysis_speculate: level %d (0..2)", level);
    a->an.speculate = level;
    return XMI_OK;
}

int xmi_analysis_configure(xmi_analysis *a, int lanes, int spin_us)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(a, "xmi_analysis_configure: null handle");
    XMI_ARG(spin_us >= 0 && spin_us <= 1000000, "xmi_analysis_configure: spin_us %d", spin_us);
    if (lanes != a->ev.lanes) {
        rc = a->ev.set_lanes(lanes);
        if (rc == XMI_OK) a->an.lanes = lanes;
    }
    if (rc == XMI_OK) a->an.spin_us = spin_us, a->ev.early_answers = spin_us > 0;
    return rc;
}

#ifdef XMI_EXP_ROUND_TIMING
int xmi_debug_search_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_search_ticks), sizeof(unsigned long long) * 10) == hipSuccess ? 0 : -1; }
int xmi_debug_quad_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_ticks), sizeof(unsigned long long) * 6) == hipSuccess ? 0 : -1; }
int xmi_debug_round_classes(unsigned long long *out)
{
    for (int c = 0; c < 5; c++)
        for (int k = 0; k < 3; k++) out[c * 3 + k] = g_round_class[c][k].exchange(0);
    return 0;
}
int xmi_debug_merge64_hist(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_merge64_hist), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1; }
int xmi_debug_span(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 5) == hipSuccess ? 0 : -1; }
int xmi_debug_clock_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_ticks), sizeof(unsigned long long) * 2) == hipSuccess ? 0 : -1; }
int xmi_debug_round_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_ticks), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1; }
int xmi_debug_intra_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_intra_ticks), sizeof(unsigned long long) * 18) == hipSuccess ? 0 : -1; }
int xmi_debug_chain_ticks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ticks), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : -1; }
#endif

int xmi_analysis_timing(xmi_analysis *a, uint64_t out[7])
{
    XMI_ARG(a && out, "xmi_analysis_timing: null argument");
    for (int k = 0; k < 5; k++) out[k] = 0;
    out[5] = a->res.row_busy_ns, out[6] = a->res.row_requests;      // of the last picture
    for (int i = 0; i < a->ev.lanes; i++) {
        const GpuEvaluator::LaneRes &l = a->ev.lane[i];
        out[0] += l.t_prepare, out[1] += l.t_launch, out[2] += l.t_wait, out[3] += l.t_collect, out[4] += l.rounds;
    }
    return XMI_OK;
}

int xmi_analysis_state(xmi_analysis *a, int set, double *sum_cost, int32_t *count, double *avg_cost, uint64_t *frac_bits)
{
    XMI_ARG(a && sum_cost && count && avg_cost && frac_bits, "xmi_analysis_state: null argument");
    Analyser &an = a->an;
    const size_t n = an.sum_cost.size(), r = an.frac_bits.size();
    if (set) {
        std::memcpy(an.sum_cost.data(), sum_cost, n * 8), std::memcpy(an.count.data(), count, n * 4), std::memcpy(an.avg_cost.data(), avg_cost, n * 8);
        std::memcpy(an.frac_bits.data(), frac_bits, r * 8);
    } else {
        std::memcpy(sum_cost, an.sum_cost.data(), n * 8), std::memcpy(count, an.count.data(), n * 4), std::memcpy(avg_cost, an.avg_cost.data(), n * 8);
        std::memcpy(frac_bits, an.frac_bits.data(), r * 8);
    }
    return XMI_OK;
}

int xmi_analysis_p_picture(xmi_analysis *a, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const uint8_t *ref_y, const uint8_t *ref_u,
                           const uint8_t *ref_v, int pad, int qp, uint16_t *info0, uint8_t *info1, int16_t *mv, int16_t *mvd, uint8_t *mvp_idx,
                           int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint32_t *rd_cost,
                           uint8_t *ctx_in, uint8_t *ctx_out, uint64_t *counters)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(a && src_y && src_u && src_v && ref_y && ref_u && ref_v, "xmi_analysis_p_picture: null picture");
    XMI_ARG(pad == kPad, "xmi_analysis_p_picture: the reference must be padded by %d luma samples (got %d)", kPad, pad);
    XMI_ARG(qp >= 0 && qp <= 51, "xmi_analysis_p_picture: qp %d", qp);
    const uint8_t *src[3] = { src_y, src_u, src_v }, *ref[3] = { ref_y, ref_u, ref_v };
    rc = a->ev.bind(src, ref, qp, chroma_qp(qp));
    if (rc != XMI_OK) return rc;
    rc = a->an.p_picture(qp, a->ev, a->res);
    if (rc != 0) {
        if (rc == -2) xmi::set_error("xmi_analysis_p_picture: the wavefront stalled");
        else xmi::set_error("xmi_analysis_p_picture: %s", a->ev.first_error());
        return rc < 0 ? rc : XMI_ERR_HIP;
    }
    const PictureResult &r = a->res;
    auto put = [](void *dst, const auto &v) {
        if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
    };
    put(info0, r.info0), put(info1, r.info1), put(mv, r.mv), put(mvd, r.mvd), put(mvp_idx, r.mvp_idx);
    put(coef_y, r.coef_y), put(coef_u, r.coef_u), put(coef_v, r.coef_v), put(rec_y, r.rec_y), put(rec_u, r.rec_u), put(rec_v, r.rec_v);
    put(rd_cost, r.rd_cost), put(ctx_in, r.ctx_in), put(ctx_out, r.ctx_out);
    if (counters) counters[0] = r.evaluator_calls, counters[1] = r.cu_requests, counters[2] = r.chroma_requests;
    return XMI_OK;
}

}  // extern "C"

// ---- the whole encoder ----------------------------------------------------------------------------------------------------
struct xmi_encoder {
    xmi::encoder::Encoder enc;
    GpuEvaluator ev;
    uint64_t lane_mark[kMaxLanes][4] = {};      // the lanes' time sums when the last picture began
    xmi_encoder(int w, int h, int qp, int strong) : enc(w, h, qp, strong) {}
};

extern "C" {

xmi_encoder *xmi_encoder_create(int width, int height, int qp, int strong_intra_smoothing)
{
    if (xmi::enter_device() != XMI_OK) return nullptr;
    if (width <= 0 || height <= 0 || width % 64 || height % 64 || width > 8192 || height > 4352 || qp < 0 || qp > 51) {
        xmi::set_error("xmi_encoder_create: %dx%d qp %d: the coded size must be a multiple of 64, at most 8192x4352, qp 0..51", width, height, qp);
        return nullptr;
    }
    xmi_encoder *e = new (std::nothrow) xmi_encoder(width, height, qp, strong_intra_smoothing ? 1 : 0);
    if (!e) {
        xmi::set_error("xmi_encoder_create: out of host memory");
        return nullptr;
    }
    if (e->ev.init(width, height, e->enc.analyser().lanes) != XMI_OK || e->ev.init_frame_ops() != XMI_OK) {
        delete e;
        return nullptr;
    }
    return e;
}

void xmi_encoder_destroy(xmi_encoder *e) { delete e; }

int xmi_encoder_configure(xmi_encoder *e, int lanes, int spin_us)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(e, "xmi_encoder_configure: null handle");
    XMI_ARG(spin_us >= 0 && spin_us <= 1000000, "xmi_encoder_configure: spin_us %d", spin_us);
    if (lanes != e->ev.lanes) {
        rc = e->ev.set_lanes(lanes);
        if (rc == XMI_OK) e->enc.analyser().lanes = lanes;
    }
    if (rc == XMI_OK) e->enc.analyser().spin_us = spin_us, e->ev.early_answers = spin_us > 0;
    return rc;
}

int xmi_encoder_workers(xmi_encoder *e, int workers)
{
    XMI_ARG(e, "xmi_encoder_workers: null handle");
    XMI_ARG(workers >= 0 && workers <= 1024, "xmi_encoder_workers: %d threads", workers);
    e->enc.analyser().workers = workers;
    return XMI_OK;
}

int xmi_encoder_rdoq(xmi_encoder *e, int on)
{
    XMI_ARG(e, "xmi_encoder_rdoq: null handle");
    e->ev.rdoq = on != 0;
    return XMI_OK;
}

int xmi_encoder_picture_hash(xmi_encoder *e, int on)
{
    XMI_ARG(e, "xmi_encoder_picture_hash: null handle");
    e->enc.picture_hash(on != 0);
    return XMI_OK;
}

int xmi_encoder_speculate(xmi_encoder *e, int level)
{
    XMI_ARG(e, "xmi_encoder_speculate: null handle");
    XMI_ARG(level >= 0 && level <= 2, "xmi_encoder_speculate: level %d (0..2)", level);
    e->enc.analyser().speculate = level;
    return XMI_OK;
}

namespace {
void mark_lanes(xmi_encoder *e)
{
    for (int i = 0; i < e->ev.lanes; i++) {
        const GpuEvaluator::LaneRes &l = e->ev.lane[i];
        e->lane_mark[i][0] = l.t_prepare, e->lane_mark[i][1] = l.t_launch, e->lane_mark[i][2] = l.t_wait, e->lane_mark[i][3] = l.t_collect;
    }
}
// the encoder's own error codes as messages and XMI_ERR_* values
int64_t encoder_error(xmi_encoder *e, const char *who, int n, size_t capacity)
{
    if (n == -100) xmi::set_error("%s: the output buffer must hold at least 4096 bytes", who);
    else if (n == -101) xmi::set_error("%s: a P picture needs a picture before it (the first picture, or the one after a failed call, is an I picture)", who);
    else if (n == -102) xmi::set_error("%s: access unit exceeds capacity (%zu bytes; slice data is limited to 3 * width * height + 4096 bytes)", who, capacity);
    else if (n == -103) xmi::set_error("%s: pictures are waiting for xmi_encoder_collect (two at most; none for xmi_encoder_encode)", who);
    else if (n == -104) xmi::set_error("%s: no picture is waiting", who);
    else if (e->ev.first_error()[0]) xmi::set_error("%s: %s", who, e->ev.first_error());
    return n <= -100 && n >= -104 ? XMI_ERR_ARG : (n == XMI_ERR_ARG || n == XMI_ERR_NO_DEVICE ? n : XMI_ERR_HIP);
}
}  // namespace

int64_t xmi_encoder_encode(xmi_encoder *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, uint8_t *out, size_t capacity)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(e && src_y && src_u && src_v && out, "xmi_encoder_encode: null argument");
    XMI_ARG(slice_type == XMI_SLICE_P || slice_type == XMI_SLICE_I, "xmi_encoder_encode: slice type %d", slice_type);
    mark_lanes(e);
    const int n = e->enc.encode(src_y, src_u, src_v, slice_type, e->ev, e->ev, out, capacity);
    return n > 0 ? n : encoder_error(e, "xmi_encoder_encode", n, capacity);
}

int xmi_encoder_submit(xmi_encoder *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(e && src_y && src_u && src_v, "xmi_encoder_submit: null argument");
    XMI_ARG(slice_type == XMI_SLICE_P || slice_type == XMI_SLICE_I, "xmi_encoder_submit: slice type %d", slice_type);
    mark_lanes(e);
    const int n = e->enc.submit(src_y, src_u, src_v, slice_type, e->ev, e->ev);
    return n == 0 ? XMI_OK : (int)encoder_error(e, "xmi_encoder_submit", n, 0);
}

int64_t xmi_encoder_collect(xmi_encoder *e, uint8_t *out, size_t capacity)
{
    XMI_ARG(e && out, "xmi_encoder_collect: null argument");
    const int n = e->enc.collect(out, capacity);
    return n > 0 ? n : encoder_error(e, "xmi_encoder_collect", n, capacity);
}

int xmi_encoder_reconstruction(xmi_encoder *e, uint8_t *y, uint8_t *u, uint8_t *v)
{
    XMI_ARG(e && y && u && v, "xmi_encoder_reconstruction: null argument");
    std::memcpy(y, e->enc.rec_y().data(), e->enc.rec_y().size());
    std::memcpy(u, e->enc.rec_u().data(), e->enc.rec_u().size());
    std::memcpy(v, e->enc.rec_v().data(), e->enc.rec_v().size());
    return XMI_OK;
}

int xmi_encoder_timing(xmi_encoder *e, uint64_t out[10])
{
    XMI_ARG(e && out, "xmi_encoder_timing: null argument");
    for (int k = 0; k < 4; k++) out[k] = e->enc.stage_ns[k];
    out[4] = e->enc.analysis_result().evaluator_calls;
    out[5] = e->enc.analysis_result().cu_requests + e->enc.analysis_result().chroma_requests;
    for (int k = 6; k < 10; k++) out[k] = 0;
    for (int i = 0; i < e->ev.lanes; i++) {
        const GpuEvaluator::LaneRes &l = e->ev.lane[i];
        out[6] += l.t_prepare - e->lane_mark[i][0], out[7] += l.t_launch - e->lane_mark[i][1];
        out[8] += l.t_wait - e->lane_mark[i][2], out[9] += l.t_collect - e->lane_mark[i][3];
    }
    return XMI_OK;
}

}  // extern "C"
