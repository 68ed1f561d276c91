// Batched replay of the reference's xMotionSearch (src/encode.cpp:2815-3003) on gfx950.
//
// One wavefront per 16x16 .. 64x64 search request, four 8x8 requests per wavefront. The control flow is the reference's, statement for statement (MV_CLIP = 1,
// MVD_COST_BY_TABLE = 0, SATD_FOR_FRAC_ME = 0): pick the better of two MV predictors by the SAD of its
// interpolated block; test (0,0) unless the caller already did; test the predictor's integer position; walk a
// small diamond for at most ME_RANGE steps with first-improvement-wins order, stopping when nothing improves
// or the cost drops below 800; clip (x against the Y maximum, as the reference does); then two rounds of 9
// sub-pel points at half and quarter step. Cost = SAD + floor(sqrt(lambda)) * (bits(mvp.x - x) + bits(mvp.y - y)),
// bits() = xGetComponentBits (src/encode.cpp:194-205). Ties keep the earlier candidate (strict <).
// Every SAD is computed by the 64 lanes together (v_sad_u8 on dwords, unaligned reference dwords assembled
// with v_alignbyte_b32, DPP wavefront reduction); sub-pel predictions are lookups in the interpolation planes
// (interp_planes.hip), which hold exactly what xPredInterLumaBlk would compute. The SADs of one stage (the two
// predictors + (0,0) + both possible start points; the 4 diamond points; the 9 sub-pel points) are independent
// and are evaluated together, so a search is about 8 dependent memory round trips instead of 28; the diamond walk
// reads a window of the reference that it keeps in LDS.
// What a batch costs (measured with the hardware counters on 28 000 8x8 searches): the wavefronts issue vector
// instructions a quarter of the time and the chip holds about one wavefront per SIMD on average - the time goes with
// the cache lines the candidates pull in. A candidate row of 8..64 samples is one 128-byte line, the 9 sub-pel
// candidates of a round lie in up to 9 different planes, so a search touches ~180 lines for 1.5 KB of samples.
// (Interleaving the 16 phases per 4 samples makes a round's candidates share lines, and was measured: the sub-pel
// stages gain 18 %, integer candidates and wide blocks pull 16 times the lines and the batch gets 3 times slower.)
//
// With a resident SAD surface of the picture pair (sad_surface.hip) the integer candidates are not computed at
// all: SAD(PU, integer MV) is element [ctu][blk][dy][dx] of the surface, because the surface's windows are
// relative to each block's own position exactly like the MVs of a search. A request takes that path when its PU
// sits on the CTU quad-tree and the stage's candidates are inside [-R, R]^2; otherwise the stage is computed.
#include "me_search_device.h"

namespace {

// Request tables live in device memory, so the kernel checks what it is about to address: a request is served only
// if its PU lies in the picture and every block the search can read (xmi::me_request_reach), with the dword
// over-read of an unaligned load, lies inside the `rows` x `stride` plane buffers.
template <bool SURF>
__device__ __forceinline__ bool request_in_bounds(const MeParams &p, const xmi_me_request &rq)
{
    const int n = rq.size;
    if (n != 8 && n != 16 && n != 32 && n != 64) return false;
    const int reach = xmi::me_request_reach(rq);
    if (rq.x0 < 0 || rq.y0 < 0 || (rq.x0 & 3) || rq.x0 + n > p.cur_stride || rq.x0 + n + 2 * p.pad > p.stride) return false;
    if (rq.y0 + n + 2 * p.pad > p.rows) return false;
    if (reach + 8 > p.pad) return false;
    if (SURF && (rq.x0 + n > p.surf_wc * 64 || ((rq.y0 + n + 63) >> 6) * p.surf_wc > p.surf_nctu)) return false;
    return true;
}

__device__ __forceinline__ int nth_set_bit(unsigned mask, int k)
{
    for (int i = 0; i < k; i++) mask &= mask - 1;
    return __ffs(mask) - 1;
}

// The grid has two parts. In the first, wavefront i answers request i unless it is an 8x8 search: a refused request
// is marked, a 16x16 .. 64x64 search runs on the whole wavefront. In the second, a workgroup looks at 16 consecutive
// requests and packs their 8x8 searches four to a wavefront (one per row of 16 lanes); wavefronts left without work
// leave at once. A search is a chain of dependent memory round trips, so the time of a batch is set by how many
// wavefronts it keeps busy, and most requests of a picture are 8x8.
template <bool SURF>
__global__ void __launch_bounds__(256) me_search_kernel(const MeParams p)
{
    // reference windows of the diamond walks: one per wavefront, or four small ones (one per row of 16 lanes)
    constexpr int kWinBytes = (64 + 16) * (64 + 24);
    __shared__ __align__(16) uint8_t s_win[SURF ? 1 : 4][SURF ? 16 : kWinBytes];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *win = s_win[SURF ? 0 : wave];
    const int single_blocks = (p.n + 3) >> 2;
    // Workgroup b runs on XCD b % 8, each with an L2 of its own: the grid is dealt so that an XCD serves one contiguous
    // eighth of the requests - neighbouring requests are neighbouring blocks of the picture and read overlapping windows
    // (each part of the grid, rounded up to a multiple of 8 workgroups, for itself)
    const int part1 = (single_blocks + 7) & ~7, packed_blocks = (p.n + 15) >> 4, part2 = (packed_blocks + 7) & ~7;
    const int b = (int)blockIdx.x, in2 = b >= part1, rel = in2 ? b - part1 : b;
    const int block = (rel & 7) * ((in2 ? part2 : part1) >> 3) + (rel >> 3);
    if (block >= (in2 ? packed_blocks : single_blocks)) return;
    if (!in2) {
        const int idx = block * 4 + wave;
        if (idx >= p.n) return;
        const xmi_me_request rq = p.req[idx];
        if (!request_in_bounds<SURF>(p, rq)) {        // refused: marked, so that the caller can tell
            if (lane == 0) {
                xmi_me_result out = {};
                out.best_cost = out.best_sad = 0xffffffffu;
                out.mvp_idx = -1;
                p.res[idx] = out;
            }
            return;
        }
        // (a 32x32 or 64x64 search on the whole workgroup - search<N, SURF, false, 4> - answers sooner, 13 instead of 17 us for
        // 64x64, but a batch is a matter of throughput: four searches side by side finish a batch of them in half the time)
        switch (rq.size) {
        case 8: break;                                // second part of the grid
        case 16: search<16, SURF>(p, rq, idx, lane, win); break;
        case 32: search<32, SURF>(p, rq, idx, lane, win); break;
        default: search<64, SURF>(p, rq, idx, lane, win); break;
        }
        return;
    }
    const int base = block * 16;
    bool is8 = false;
    if (lane < 16 && base + lane < p.n) {
        const xmi_me_request rq = p.req[base + lane];
        is8 = rq.size == 8 && request_in_bounds<SURF>(p, rq);
    }
    const unsigned m8 = (unsigned)__ballot(is8);
    const int n8 = __popc(m8), mine = 4 * wave + (lane >> 4);      // the row's request among the 8x8 ones of the 16
    if (mine < n8) {
        const int idx = base + nth_set_bit(m8, mine);
        const xmi_me_request rq = p.req[idx];
        search<8, SURF, true>(p, rq, idx, lane, win + (SURF ? 0 : (lane >> 4) * ((8 + 16) * (8 + 24))));
    }
}

int check_common(const char *who, const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows, int pad,
                 const xmi_me_request *d_req, int me_range, xmi_me_result *d_res)
{
    XMI_ARG(d_cur && d_planes && d_req && d_res, "%s: null argument", who);
    XMI_ARG(cur_stride % 4 == 0 && ((uintptr_t)d_cur & 3) == 0, "%s: current frame must be 4-byte aligned", who);
    XMI_ARG(stride % 4 == 0 && ((uintptr_t)d_planes & 3) == 0 && pad % 4 == 0, "%s: planes, stride and pad must be multiples of 4", who);
    XMI_ARG(me_range >= 0 && me_range <= 1024, "%s: me_range %d out of range", who, me_range);
    XMI_ARG(rows > 0 && stride > 0 && (unsigned long long)rows * (unsigned long long)stride * 16ull < (1ull << 31),
            "%s: 16 planes of %d x %d bytes exceed the 2 GiB the kernel's 32-bit offsets cover", who, rows, stride);
    return XMI_OK;
}

}  // namespace

extern "C" int xmi_me_search_batch_dev(const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows,
                                       int pad, const xmi_me_request *d_req, int n, int me_range, xmi_me_result *d_res,
                                       void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_me_search_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    rc = check_common("xmi_me_search_batch_dev", d_cur, cur_stride, d_planes, stride, rows, pad, d_req, me_range, d_res);
    if (rc != XMI_OK) return rc;
    MeParams p = { d_cur, d_planes, d_req, d_res, cur_stride, stride, rows, pad, n, me_range, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, (u32)((size_t)rows * stride), (u32)stride };
    hipLaunchKernelGGL(me_search_kernel<false>, dim3(((n + 3) / 4 + 7) / 8 * 8 + ((n + 15) / 16 + 7) / 8 * 8), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_me_search_surface_batch_dev(const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows,
                                               int pad, const xmi_sad_geom *g, const uint16_t *d_sad8, const uint16_t *d_sad16,
                                               const uint32_t *d_sad32, const uint32_t *d_sad64, const xmi_me_request *d_req, int n,
                                               int me_range, xmi_me_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_me_search_surface_batch_dev: negative request count");
    if (n == 0) return XMI_OK;
    rc = check_common("xmi_me_search_surface_batch_dev", d_cur, cur_stride, d_planes, stride, rows, pad, d_req, me_range, d_res);
    if (rc != XMI_OK) return rc;
    XMI_ARG(g && d_sad8 && d_sad16 && d_sad32 && d_sad64, "xmi_me_search_surface_batch_dev: null surface");
    XMI_ARG(g->width > 0 && g->height > 0 && g->width % 64 == 0 && g->height % 64 == 0 &&
                (g->range == 16 || g->range == 32 || g->range == 64 || g->range == 128),
            "xmi_me_search_surface_batch_dev: surface geometry %dx%d range %d is not one xmi_sad_surface_dev produces", g->width,
            g->height, g->range);
    MeParams p = { d_cur, d_planes, d_req, d_res, cur_stride, stride, rows, pad, n, me_range, d_sad8, d_sad16, d_sad32, d_sad64,
                   g->range, g->width / 64, (g->width / 64) * (g->height / 64), (u32)((size_t)rows * stride), (u32)stride };
    hipLaunchKernelGGL(me_search_kernel<true>, dim3(((n + 3) / 4 + 7) / 8 * 8 + ((n + 15) / 16 + 7) / 8 * 8), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}
