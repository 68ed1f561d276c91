// me_search_device.h - the motion search of one PU as a device function (one wavefront, or a row of 16 lanes for a packed
// 8x8 search): shared by the batch kernels of me_search.hip and by the analysis kernel, which runs a search in the middle
// of a CU's evaluation. Include from a .hip file only.
#pragma once
#include "xmi_common.h"

#include <type_traits>

namespace {

typedef unsigned int u32;
struct __attribute__((aligned(4))) dword_pair { u32 x, y; };      // an 8-byte access that is only dword-aligned

struct MeParams {
    const uint8_t *cur;
    const uint8_t *planes;   // 16 phase planes; sample (y, x) of phase ph at planes[ph * phase_pitch + y * stride + x]; pixel (0,0) of the frame at (pad, pad)
    const xmi_me_request *req;
    xmi_me_result *res;
    int cur_stride, stride, rows, pad, n, me_range;      // stride: bytes between rows of a plane
    // resident SAD surface of the same picture pair (SURF kernels only)
    const uint16_t *sad8, *sad16;
    const u32 *sad32, *sad64;
    int surf_range, surf_wc, surf_nctu;
    u32 phase_pitch, row_bytes;                          // bytes between phases; bytes of a row that may be read
};

__device__ __forceinline__ u32 mv_bits(int v)
{
    const u32 t = v <= 0 ? ((u32)(-v) << 1) + 1u : (u32)v << 1;
    return 1u + 2u * (31u - (u32)__clz((int)t));
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

// wavefront sum, returned as a uniform value
__device__ __forceinline__ u32 wave_sum(u32 x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);    // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);    // row_bcast:31
    return (u32)__builtin_amdgcn_readlane(v, 63);
}

// sum over each aligned group of 16 lanes (a DPP row), returned in every lane of the group: quad butterflies, then the
// half-row and row mirrors. Only lanes of the own row are read, so rows may be in different branches of the search.
__device__ __forceinline__ u32 row_sum(u32 x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true);     // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true);     // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);    // row_mirror
    return (u32)v;
}

// One search. The reference evaluates its candidates one after the other, but inside each stage they do not
// depend on each other (only the comparisons do), so a stage's SADs are computed together - all loads of a
// stage are in flight at once - and the comparisons are then applied in the reference's order.
// PACK (8x8 only): the block's 16 dwords occupy a quarter of the wavefront, so four searches share one - each row of 16
// lanes runs its own request (idx, rq and everything derived from them differ per row; the rows diverge where their
// searches do), and SADs are summed per row.
// WAVES = 4 (32x32 and 64x64 only): the whole workgroup of 256 threads runs one search - every thread of it must call, with
// the same request - and the partial sums of its wavefronts meet in s_red (4 x 9 words of LDS); a 64x64 candidate is then 4
// dwords per thread instead of 16, and a stage's candidates fit in registers together.
// zero_fn (optional): the caller's answer to "was (0,0) tested already?", asked for only when it matters - when (0,0) would
// beat the predictor's integer position - instead of rq.zero_mv_tested; every thread that shares the block calls it together.
// n_int_sads of the result then counts (0,0) only when it was asked.
#ifdef XMI_EXP_ROUND_TIMING
__device__ unsigned long long g_search_ticks[2][5];      // workgroup searches, 32x32 | 64x64: first stage, diamond walk, sub-pel rounds, integer SADs, count
#endif
template <int N, bool SURF, bool PACK = false, int WAVES = 1, class ZeroFn = std::nullptr_t>
__device__ __forceinline__ void search(const MeParams &p, const xmi_me_request &rq, int idx, int wave_lane, uint8_t *s_win, u32 *s_red = nullptr,
                                       ZeroFn zero_fn = nullptr)
{
    static_assert(!PACK || N == 8, "only 8x8 blocks fit four to a wavefront");
    static_assert(WAVES == 1 || (WAVES == 4 && N >= 32 && !PACK), "a workgroup search is for the large blocks");
    constexpr int GT = PACK ? 16 : 64 * WAVES;       // threads that share the block
    constexpr int DW = N * N / 4, DPR = N / 4;       // dwords per block / per block row
    constexpr int J = DW < GT ? 1 : DW / GT;         // block dwords per lane
    // candidates evaluated together, i.e. whose loads are in flight at once (bounds the registers): a 64x64 search on a whole
    // workgroup takes a sub-pel round's nine in one memory round trip
    constexpr int KC = J == 1 ? 9 : J == 4 ? (WAVES == 4 ? 9 : 4) : 1;
    const int lane = PACK ? (wave_lane & 15) : wave_lane;
    const bool live = DW >= GT || lane < DW;
    u32 cur[J];
    // the lane's dwords: d = lane + GT j -> row lane / DPR + j * (GT / DPR), column dword lane % DPR
    constexpr int JR = GT / DPR > 0 ? GT / DPR : 1;  // rows between a lane's dwords
    const int row0 = live ? lane / DPR : 0, c40 = live ? lane % DPR : 0;
    const int off0 = row0 * p.stride + c40 * 4, joff = JR * p.stride;    // byte offsets inside a reference block
#pragma unroll
    for (int j = 0; j < J; j++)
        cur[j] = live ? *reinterpret_cast<const u32 *>(p.cur + (size_t)(rq.y0 + row0 + j * JR) * p.cur_stride + rq.x0 + c40 * 4) : 0u;
    // sums of CNT partial values over the threads that share the block, uniform in all of them
    auto reduce = [&](auto cnt_tag, u32 *v) {
        constexpr int CNT = decltype(cnt_tag)::value;
        if constexpr (WAVES == 1) {
#pragma unroll
            for (int k = 0; k < CNT; k++) v[k] = PACK ? row_sum(v[k]) : wave_sum(v[k]);
        } else {
            const int w = wave_lane >> 6;
#pragma unroll
            for (int k = 0; k < CNT; k++) v[k] = wave_sum(v[k]);
            __syncthreads();                          // the previous sums have been read
            if ((wave_lane & 63) == 0)
#pragma unroll
                for (int k = 0; k < CNT; k++) s_red[w * 9 + k] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CNT; k++) v[k] = s_red[k] + s_red[9 + k] + s_red[18 + k] + s_red[27 + k];
        }
    };
    // SADs of COUNT candidate blocks (top-left pointers, any byte alignment) against the current block
    // (candidates are byte offsets into the plane buffer: 32-bit arithmetic, and loads of the form uniform base + lane offset)
    auto eval = [&](auto count_tag, const u32 *cand, u32 *out) {
        constexpr int COUNT = decltype(count_tag)::value;
#pragma unroll
        for (int k0 = 0; k0 < COUNT; k0 += KC) {
            u32 lo[KC][J], hi[KC][J], sh[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) {
                if (k0 + k >= COUNT) continue;
                const u32 a = cand[k0 + k];
                sh[k] = a & 3u;
                const u32 al = (a & ~3u) + (u32)off0;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const dword_pair w = live ? *reinterpret_cast<const dword_pair *>(p.planes + (al + (u32)(j * joff))) : dword_pair{ 0u, 0u };
                    lo[k][j] = w.x;
                    hi[k][j] = w.y;
                }
            }
            u32 part[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) {
                part[k] = 0;
                if (k0 + k >= COUNT) continue;
#pragma unroll
                for (int j = 0; j < J; j++) part[k] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi[k][j], lo[k][j], sh[k]), cur[j], part[k]);
            }
            reduce(std::integral_constant<int, KC>{}, part);
#pragma unroll
            for (int k = 0; k < KC; k++)
                if (k0 + k < COUNT) out[k0 + k] = part[k];
        }
    };
    const u32 ref_off = (u32)((p.pad + rq.y0) * p.stride + p.pad + rq.x0), plane_u32 = p.phase_pitch;
    auto qpel = [&](int mx, int my) { return ref_off + (u32)((my & 3) * 4 + (mx & 3)) * plane_u32 + (u32)((my >> 2) * p.stride + (mx >> 2)); };
    auto ipel = [&](int x, int y) { return ref_off + (u32)(y * p.stride + x); };
    // ---- reference window in LDS for the diamond walk (kernels without a SAD surface) ----
    // A walk is a chain of up to me_range dependent steps, and read from memory every step is a round trip. The window
    // holds the (N + 2 WR)^2 samples around a centre, enough for WR steps in any direction; steps inside it cost LDS
    // reads only, and the walk reloads it around its current position when it reaches the rim. Rows are WP bytes apart
    // and start at the dword that holds the window's first sample (offset win_sh into it); dwords outside the plane
    // buffer are not read (no candidate lies there).
    constexpr int WR = 8, WH = N + 2 * WR, WP = N + 2 * WR + 8, WPD = WP / 4, GL = GT;
    auto group_barrier = [&] {
        if constexpr (WAVES == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    int win_cx = 0, win_cy = 0, win_sh = 0;
    bool win_ok = false;
    auto win_load = [&](int cx, int cy) {
        const int gy0 = p.pad + rq.y0 + cy - WR, gx = p.pad + rq.x0 + cx - WR, gxb = gx & ~3;
        group_barrier();                      // reads of the previous window are done
        for (int e = lane; e < WH * WPD; e += GL) {
            const int r = e / WPD, c = e - r * WPD, y = gy0 + r, x = gxb + 4 * c;
            u32 v = 0;
            if (y >= 0 && y < p.rows && x >= 0 && x + 4 <= (int)p.row_bytes) v = *reinterpret_cast<const u32 *>(p.planes + (size_t)y * p.stride + x);
            reinterpret_cast<u32 *>(s_win)[e] = v;
        }
        group_barrier();
        win_cx = cx;
        win_cy = cy;
        win_sh = gx & 3;
        win_ok = true;
    };
    auto win_has = [&](int cx, int cy, int x, int y) { return abs(x - cx) <= WR && abs(y - cy) <= WR; };
    auto win_sad = [&](int x, int y) -> u32 {   // this thread's part of the SAD
        const int bo = x - win_cx + WR + win_sh, d0 = (bo >> 2) + c40;
        const u32 sh = (u32)(bo & 3);
        const u32 *row = reinterpret_cast<const u32 *>(s_win) + (y - win_cy + WR + row0) * WPD + d0;
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const u32 *q = row + j * JR * WPD;
            acc = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(q[1], q[0], sh), cur[j], acc);
        }
        if (!live) acc = 0;      // an 8x8 block on a whole wavefront: the lanes beyond its 16 dwords hold no samples
        return acc;
    };
    const u32 lam = rq.sqrt_lambda;
    // surface coordinates of this PU (wave-uniform)
    constexpr int LEVEL = N == 8 ? 0 : N == 16 ? 1 : N == 32 ? 2 : 3;
    const int R = p.surf_range;
    const bool on_tree = SURF && (rq.x0 & (N - 1)) == 0 && (rq.y0 & (N - 1)) == 0;
    const int s_ctu = (rq.y0 >> 6) * p.surf_wc + (rq.x0 >> 6), s_blk = ((rq.y0 & 63) / N) * (64 / N) + (rq.x0 & 63) / N;
    auto in_window = [&](int x, int y) { return x >= -R && x <= R && y >= -R && y <= R; };
    auto surf = [&](int x, int y) -> u32 {
        const size_t e = xmi::surface_index(R, (size_t)p.surf_nctu, LEVEL, s_ctu, s_blk, y, x);
        return LEVEL == 0 ? (u32)p.sad8[e] : LEVEL == 1 ? (u32)p.sad16[e] : LEVEL == 2 ? p.sad32[e] : p.sad64[e];
    };

#ifdef XMI_EXP_ROUND_TIMING
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- predictor choice, (0,0) and the integer start point ----
    const int passes = ((abs(rq.mvp[0][0] - rq.mvp[1][0]) >> 2) + (abs(rq.mvp[0][1] - rq.mvp[1][1]) >> 2) > 4) ? 2 : 1;
    u32 s5[5];
    {
        const int ax = rq.mvp[0][0] >> 2, ay = rq.mvp[0][1] >> 2, bx = rq.mvp[1][0] >> 2, by = rq.mvp[1][1] >> 2;
        const u32 c[5] = { qpel(rq.mvp[0][0], rq.mvp[0][1]), qpel(rq.mvp[1][0], rq.mvp[1][1]), ipel(0, 0), ipel(ax, ay), ipel(bx, by) };
        if (on_tree && in_window(ax, ay) && in_window(bx, by)) {
            eval(std::integral_constant<int, 2>{}, c, s5);
            s5[2] = surf(0, 0);
            s5[3] = surf(ax, ay);
            s5[4] = surf(bx, by);
        } else {
            eval(std::integral_constant<int, 5>{}, c, s5);
        }
    }
    int n_int = 0, n_sub = passes;
    const int mvp_idx = (passes == 2 && s5[1] < s5[0]) ? 1 : 0;
    const u32 bsad = mvp_idx ? s5[1] : s5[0];
    const int px = mvp_idx ? rq.mvp[1][0] : rq.mvp[0][0], py = mvp_idx ? rq.mvp[1][1] : rq.mvp[0][1];
    auto mvcost = [&](int x, int y) { return lam * (mv_bits(px - x) + mv_bits(py - y)); };
    u32 bcost = 0x7fffffffu;
    int bmx = 0, bmy = 0, best_k = -1;
    int omx = px >> 2, omy = py >> 2;
    bool zero_tested = rq.zero_mv_tested != 0;
    if constexpr (!std::is_same<ZeroFn, std::nullptr_t>::value) {
        // (0,0) goes first and the predictor's position replaces it unless it costs more: testing (0,0) changes nothing when
        // it costs more than the predictor's position, or is that position
        const bool matters = s5[2] + mvcost(0, 0) <= (mvp_idx ? s5[4] : s5[3]) + mvcost(omx * 4, omy * 4) && (omx != 0 || omy != 0);
        zero_tested = matters ? zero_fn() : true;
    }
    if (!zero_tested) {
        const u32 c = s5[2] + mvcost(0, 0);
        n_int++;
        if (c < bcost) {
            bcost = c;
            bmx = bmy = 0;
        }
    }
    {
        const u32 c = (mvp_idx ? s5[4] : s5[3]) + mvcost(omx * 4, omy * 4);
        n_int++;
        if (c < bcost) {
            bcost = c;
            bmx = omx * 4;
            bmy = omy * 4;
        }
    }
#ifdef XMI_EXP_ROUND_TIMING
    const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- small diamond ----
    const int dia4x[4] = { 0, 0, -1, 1 }, dia4y[4] = { -1, 1, 0, 0 };
    for (int i = 0; i < p.me_range; i++) {
        int tx[4], ty[4];
        u32 c[4];
        u32 s4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            tx[k] = clip3(rq.mv_min_ipel[0], rq.mv_max_ipel[0], omx + dia4x[k]);
            ty[k] = clip3(rq.mv_min_ipel[1], rq.mv_max_ipel[1], omy + dia4y[k]);
            c[k] = ipel(tx[k], ty[k]);
        }
        if (on_tree && in_window(tx[0], ty[0]) && in_window(tx[1], ty[1]) && in_window(tx[2], ty[2]) && in_window(tx[3], ty[3])) {
#pragma unroll
            for (int k = 0; k < 4; k++) s4[k] = surf(tx[k], ty[k]);
        } else if (SURF) {
            eval(std::integral_constant<int, 4>{}, c, s4);
        } else {
            auto all_in = [&](int cx, int cy) {
                return win_has(cx, cy, tx[0], ty[0]) && win_has(cx, cy, tx[1], ty[1]) && win_has(cx, cy, tx[2], ty[2]) && win_has(cx, cy, tx[3], ty[3]);
            };
            if (!(win_ok && all_in(win_cx, win_cy)) && all_in(omx, omy)) win_load(omx, omy);
            if (win_ok && all_in(win_cx, win_cy)) {
#pragma unroll
                for (int k = 0; k < 4; k++) s4[k] = win_sad(tx[k], ty[k]);
                reduce(std::integral_constant<int, 4>{}, s4);
            } else {      // the clip box keeps the candidates away from the walk's position
                eval(std::integral_constant<int, 4>{}, c, s4);
            }
        }
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 cst = s4[k] + mvcost(tx[k] * 4, ty[k] * 4);
            n_int++;
            if (cst < bcost) {
                changed = true;
                bcost = cst;
                best_k = k;
                bmx = tx[k] * 4;
                bmy = ty[k] * 4;
            }
        }
        if (!changed) break;
        omx += best_k == 2 ? -1 : best_k == 3 ? 1 : 0;
        omy += best_k == 0 ? -1 : best_k == 1 ? 1 : 0;
        if (bcost < 800) break;
    }
    {   // src/encode.cpp:2950-2953: x is clipped against the Y maximum (kept)
        const int mnx = 4 * (rq.mv_min_ipel[0] - 4), mny = 4 * (rq.mv_min_ipel[1] - 4), mxy = 4 * (rq.mv_max_ipel[1] - 4);
        bmx = clip3(mnx, mxy, bmx);
        bmy = clip3(mny, mxy, bmy);
    }
#ifdef XMI_EXP_ROUND_TIMING
    const unsigned long long st2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- half- and quarter-pel rounds ----
    bcost = 0x7fffffffu;
    const int dia9x[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 }, dia9y[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 };
    int step = 2;
    for (int i = 0; i < 2; i++) {
        int tx[9], ty[9];
        u32 c[9];
        u32 s9[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            tx[k] = (int16_t)(bmx + dia9x[k] * step);
            ty[k] = (int16_t)(bmy + dia9y[k] * step);
            c[k] = qpel(tx[k], ty[k]);
        }
        eval(std::integral_constant<int, 9>{}, c, s9);
        int bdx = 0, bdy = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const u32 cst = s9[k] + mvcost(tx[k], ty[k]);
            n_sub++;
            if (cst < bcost) {
                bcost = cst;
                best_k = k;
                bdx = dia9x[k] * step;
                bdy = dia9y[k] * step;
            }
        }
        bmx += bdx;       // nothing improved: the offsets stay 0
        bmy += bdy;
        step >>= 1;
    }
#ifdef XMI_EXP_ROUND_TIMING
    if (WAVES == 4 && wave_lane == 0) {
        const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *t = g_search_ticks[N == 64];
        atomicAdd(&t[0], st1 - st0), atomicAdd(&t[1], st2 - st1), atomicAdd(&t[2], st3 - st2), atomicAdd(&t[3], (unsigned long long)n_int), atomicAdd(&t[4], 1ull);
    }
#endif
    if (wave_lane == 0 || (PACK && lane == 0)) {
        xmi_me_result out;
        out.mv[0] = (int16_t)bmx;
        out.mv[1] = (int16_t)bmy;
        out.best_cost = bcost;
        out.best_sad = bsad;
        out.mvp_idx = mvp_idx;
        out.n_int_sads = n_int;
        out.n_sub_sads = n_sub;
        p.res[idx] = out;
    }
}

}  // namespace
