// Frame-wide integer-ME SAD surface for gfx950.
//
// For every 64x64 CTU and every integer MV in [-R,R]^2 the kernel produces the 85 SADs of the CU
// quad-tree (64 8x8, 16 16x16, 4 32x32, 1 64x64): exactly the values the reference obtains from
// xSad_AMP[idx](cur, 64, ref + dy*stride + dx, stride) (src/encode.cpp:2885-2931,
// src/pixel.cpp:578-696, 833-844). SAD is an exact integer sum, so any summation order is bit-exact.
//
// Mapping (one workgroup = one CTU x one band of dy rows, R/8 wavefronts):
//   lane = z*4 + gl            z  = 16x16 block of the CTU in z-order (quadrant-major)
//                              gl = one of 4 dx-groups owned by the wavefront; a dx-group is 4
//                                   consecutive dx, produced at once by v_qsad_pk_u16_u8
//   The lane walks down the reference window one row per step. A reference row r meets the 8 rows y
//   of each 8x8 child, contributing to dy = r - y: a ring of 8 packed-u16 accumulators per child
//   holds the 8 dy values in flight; one completes per step and is written out. Reference and
//   current pixels are staged once in LDS; nothing is re-read from HBM.
//   16x16 = sum of the lane's 4 children (still packed u16, max 65280); 32x32 = DPP row-rotate sum
//   over the 4 lanes of a quadrant; 64x64 = cross-row shuffle sum over the 4 quadrants.
//   The dx = +R column (one more than a multiple of 4) is produced by the EDGE instantiation.
//
// Output layout (chosen so that every store instruction of a wavefront writes one contiguous run):
//   main part, dx in [-R, R-1]: per (ctu, dy) a row of R/8 tiles of 16 consecutive dx (rows are
//   multiples of 128 bytes, so every row starts on a cache line):
//     sad8 : [ctu][dy][tile][c*16 + z][dx % 16]   c = (by8&1)*2 + (bx8&1), z = z-order of the parent 16x16
//     sad16: [ctu][dy][tile][z][dx % 16]
//     sad32: [ctu][dy][tile][q][dx % 16]          q = raster index of the 32x32 quadrant
//     sad64: [ctu][dy][tile][dx % 16]
//   edge part, dx = +R: appended after the main part of the same plane: [ctu][dy][block slot].
//   xmi_sad_surface_index() does this arithmetic.
#include "xmi_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

struct SadParams {
    const uint8_t *cur;
    const uint8_t *ref;  // pixel (0,0) of the padded reference
    uint16_t *sad8, *sad16;
    u32 *sad32, *sad64;
    int cur_stride, ref_stride;
    int wc;         // CTUs per frame row
    int seg_len;    // dy rows per workgroup (main) or per lane group (edge)
    int ctu_begin;
    int nctu_total;  // CTUs in the frame: the edge part of every plane starts after nctu_total main blocks
};

constexpr int kCurLdsBytes = 16 * 16 * 16;  // [row in 16x16 block][z][16 bytes]

// blocks per level, and elements per (ctu, dy) row of the main part of each level's plane
__host__ __device__ constexpr int level_blocks(int level) { return level == 0 ? 64 : level == 1 ? 16 : level == 2 ? 4 : 1; }
__host__ __device__ constexpr int row_elems(int R, int level) { return (R / 8) * level_blocks(level) * 16; }

__device__ __forceinline__ u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

template <int CTRL>
__device__ __forceinline__ u32 dpp_add(u32 x)
{
    return x + (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}

template <int R, bool EDGE>
__global__ void __launch_bounds__(EDGE ? 64 : 8 * R, EDGE ? 1 : (R == 128 ? 4 : 4)) sad_surface_kernel(const SadParams p)
{
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char *cur_lds = lds;
    unsigned char *ref_lds = lds + kCurLdsBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int ctu = p.ctu_begin + blockIdx.x;
    const int cx0 = (ctu % p.wc) * 64, cy0 = (ctu / p.wc) * 64;
    constexpr int nmv = 2 * R + 1;

    // window geometry
    const int seg_base = EDGE ? 0 : (int)blockIdx.y * p.seg_len;      // first dy index of this workgroup
    const int ndy_wg = EDGE ? nmv : min(p.seg_len, nmv - seg_base);   // dy rows staged for the workgroup
    const int valid_rows = ndy_wg + 63;
    constexpr int wb_dw = EDGE ? 17 : (R / 2 + 16);                   // window width in dwords
    constexpr int rls = (wb_dw | 1) * 4;                                  // odd dword pitch: no bank conflicts
    constexpr int col_origin = EDGE ? R : -R;

    // ---- stage current CTU: cur_lds[(r16*16 + z)*16 + byte] ----
    for (int idx = tid; idx < 1024; idx += blockDim.x) {
        const int row = idx >> 4, cd = idx & 15;
        const int bx16 = cd >> 2, by16 = row >> 4, r16 = row & 15;
        const int z = ((((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1)));
        const u32 v = *reinterpret_cast<const u32 *>(p.cur + (size_t)(cy0 + row) * p.cur_stride + cx0 + cd * 4);
        reinterpret_cast<u32 *>(cur_lds)[(r16 * 16 + z) * 4 + (cd & 3)] = v;
    }
    // ---- stage reference window rows [cy0 - R + seg_base, +valid_rows) x [cx0 + col_origin, +4*wb_dw) ----
    // flat dword index so that all lanes carry loads; 8 independent loads are in flight per thread
    {
        const uint8_t *src0 = p.ref + (ptrdiff_t)(cy0 - R + seg_base) * p.ref_stride + cx0 + col_origin;
        const int total = valid_rows * wb_dw;
        for (int base = tid; base < total; base += 8 * (int)blockDim.x) {
            u32 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int idx = base + k * (int)blockDim.x;
                const int row = idx / wb_dw, col = idx - row * wb_dw;
                v[k] = idx < total ? *reinterpret_cast<const u32 *>(src0 + (ptrdiff_t)row * p.ref_stride + col * 4) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int idx = base + k * (int)blockDim.x;
                const int row = idx / wb_dw, col = idx - row * wb_dw;
                if (idx < total) *reinterpret_cast<u32 *>(ref_lds + row * rls + col * 4) = v[k];
            }
        }
    }
    __syncthreads();

    // ---- per-lane geometry ----
    const int z = lane >> 2, gl = lane & 3;
    const int bx16 = ((z >> 2) & 1) * 2 + (z & 1), by16 = (z >> 3) * 2 + ((z >> 1) & 1);
    const int g = EDGE ? 0 : wave * 4 + gl;               // dx group (main); edge reads at window column 0
    const int dy_lane0 = EDGE ? gl * p.seg_len : 0;       // first dy row (window-relative) of this lane
    const int ndy_lane = EDGE ? p.seg_len : ndy_wg;
    const int nsteps = (ndy_lane + 7 + 7) & ~7;

    const unsigned char *ref_lane = ref_lds + (by16 * 16 + dy_lane0) * rls + (bx16 * 4 + g) * 4;

    // output offsets (elements): main rows are [tile][slot][16 dx]; the edge rows [slot] follow all main rows
    constexpr int ROW8 = EDGE ? 64 : row_elems(R, 0), ROW16 = EDGE ? 16 : row_elems(R, 1);
    constexpr int ROW32 = EDGE ? 4 : row_elems(R, 2), ROW64 = EDGE ? 1 : row_elems(R, 3);
    const size_t rows_before = (EDGE ? (size_t)p.nctu_total * nmv : 0);
    uint16_t *o8 = p.sad8 + rows_before * row_elems(R, 0) + (size_t)ctu * nmv * ROW8;
    uint16_t *o16 = p.sad16 + rows_before * row_elems(R, 1) + (size_t)ctu * nmv * ROW16;
    u32 *o32 = p.sad32 + rows_before * row_elems(R, 2) + (size_t)ctu * nmv * ROW32;
    u32 *o64 = p.sad64 + rows_before * row_elems(R, 3) + (size_t)ctu * nmv * ROW64;
    const int q = (by16 >> 1) * 2 + (bx16 >> 1);
    const u32 off8 = EDGE ? z : wave * 1024 + z * 16 + gl * 4;       // child c adds c*16 slots
    const u32 off16 = EDGE ? z : wave * 256 + z * 16 + gl * 4;
    // 32x32 / 64x64: after the reductions every lane of a quadrant (of the wavefront) holds the same 4 sums;
    // lane b = z&3 writes the b-th one, so all 64 (16) lanes store one dword each into one contiguous run
    const u32 off32 = EDGE ? q : wave * 64 + q * 16 + gl * 4 + (z & 3);
    const u32 off64 = EDGE ? 0 : wave * 16 + gl * 4 + (z & 3);
    constexpr u32 CH = EDGE ? 16 : 256;                              // element step between children

    u64 acc[4][8];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[c][j] = 0;

    for (int s0 = 0; s0 < nsteps; s0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int s = s0 + u;
            // The current-CTU rows are loop invariant; holding them would cost 64 VGPRs. Launder the LDS
            // address every step so the reads stay inside the step (LDS bandwidth is plentiful).
            u32 cur_off = (u32)z * 16u;
            asm volatile("" : "+v"(cur_off));
            const unsigned char *cl = cur_lds + cur_off;
#ifndef XMI_EXP_NOCOMPUTE
#pragma unroll
            for (int cy = 0; cy < 2; cy++) {
                const u32 *rr = reinterpret_cast<const u32 *>(ref_lane + (cy * 8 + s) * rls);
                const u32 w0 = rr[0], w1 = rr[1], w2 = rr[2], w3 = rr[3], w4 = rr[4];
                const u64 p01 = pack64(w0, w1), p12 = pack64(w1, w2), p23 = pack64(w2, w3), p34 = pack64(w3, w4);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    asm volatile("" ::: "memory");  // bound the live range of the current-pixel registers:
                    __builtin_amdgcn_sched_barrier(0);  // neither the compiler nor the scheduler may hoist these reads
                    uint4 c[4];
#pragma unroll
                    for (int yy = 0; yy < 4; yy++) c[yy] = *reinterpret_cast<const uint4 *>(cl + (cy * 8 + h * 4 + yy) * 256);
#pragma unroll
                    for (int yy = 0; yy < 4; yy++) {
                        const int y = h * 4 + yy;
                        const int j = (u - y) & 7;
                        const u64 a0 = (y == 0) ? 0ull : acc[cy * 2 + 0][j];
                        const u64 a1 = (y == 0) ? 0ull : acc[cy * 2 + 1][j];
                        acc[cy * 2 + 0][j] = __builtin_amdgcn_qsad_pk_u16_u8(p12, c[yy].y, __builtin_amdgcn_qsad_pk_u16_u8(p01, c[yy].x, a0));
                        acc[cy * 2 + 1][j] = __builtin_amdgcn_qsad_pk_u16_u8(p34, c[yy].w, __builtin_amdgcn_qsad_pk_u16_u8(p23, c[yy].z, a1));
                    }
                }
            }
#endif
            // Pin every accumulator here: without it the compiler sinks this step's SADs into later
            // steps (only one accumulator per child is consumed now) and keeps whole rows live.
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("" : "+v"(acc[c][j]));
            // the accumulators that started 7 steps ago are complete: dy index (lane-relative) s-7
            const int jd = (u + 1) & 7;
            const int dyl = s - 7;
            const int dyi = seg_base + dy_lane0 + dyl;
            const u64 v0 = acc[0][jd], v1 = acc[1][jd], v2 = acc[2][jd], v3 = acc[3][jd];
            const u64 v16 = v0 + v1 + v2 + v3;  // packed u16 lanes, each <= 65280: no carry between lanes
            u32 x0 = (u32)v16 & 0xffffu, x1 = (u32)v16 >> 16, x2 = (u32)(v16 >> 32) & 0xffffu, x3 = (u32)(v16 >> 48);
            // 32x32: sum over the quadrant's 4 blocks = lane bits 3:2 (row rotate by 4 and 8 inside 16 lanes)
            x0 = dpp_add<0x128>(dpp_add<0x124>(x0));
            x1 = dpp_add<0x128>(dpp_add<0x124>(x1));
            x2 = dpp_add<0x128>(dpp_add<0x124>(x2));
            x3 = dpp_add<0x128>(dpp_add<0x124>(x3));
            // 64x64: sum over the 4 quadrants = lane bits 5:4
            u32 t0 = x0 + (u32)__shfl_xor((int)x0, 16), t1 = x1 + (u32)__shfl_xor((int)x1, 16);
            u32 t2 = x2 + (u32)__shfl_xor((int)x2, 16), t3 = x3 + (u32)__shfl_xor((int)x3, 16);
            t0 += (u32)__shfl_xor((int)t0, 32);
            t1 += (u32)__shfl_xor((int)t1, 32);
            t2 += (u32)__shfl_xor((int)t2, 32);
            t3 += (u32)__shfl_xor((int)t3, 32);

#ifdef XMI_EXP_NOSTORE
            if (dyl >= 0 && dyl < ndy_lane && dyi < nmv && v16 == 0x1234567812345678ull && t0 == 77u) {
#else
            if (dyl >= 0 && dyl < ndy_lane && dyi < nmv) {
#endif
                if (!EDGE) {
#ifndef XMI_EXP_STORES
#define XMI_EXP_STORES 15
#endif
                    u64 *r8 = reinterpret_cast<u64 *>(o8 + (u32)dyi * ROW8 + off8);
                    if (XMI_EXP_STORES & 1) {
                        r8[0] = v0;
                        r8[CH / 4] = v1;
                        r8[2 * CH / 4] = v2;
                        r8[3 * CH / 4] = v3;
                    }
                    if (XMI_EXP_STORES & 2) *reinterpret_cast<u64 *>(o16 + (u32)dyi * ROW16 + off16) = v16;
                    const int b = z & 3;
                    if (XMI_EXP_STORES & 4) o32[(u32)dyi * ROW32 + off32] = b == 0 ? x0 : b == 1 ? x1 : b == 2 ? x2 : x3;
                    if ((XMI_EXP_STORES & 8) && z < 4) o64[(u32)dyi * ROW64 + off64] = b == 0 ? t0 : b == 1 ? t1 : b == 2 ? t2 : t3;
                } else {
                    uint16_t *r8 = o8 + (u32)dyi * ROW8 + off8;
                    r8[0] = (uint16_t)v0;
                    r8[CH] = (uint16_t)v1;
                    r8[2 * CH] = (uint16_t)v2;
                    r8[3 * CH] = (uint16_t)v3;
                    o16[(u32)dyi * ROW16 + off16] = (uint16_t)v16;
                    if ((z & 3) == 0) o32[(u32)dyi * ROW32 + off32] = x0;
                    if (z == 0) o64[(u32)dyi * ROW64 + off64] = t0;
                }
            }
        }
    }
}

int seg_len_edge(int nmv) { return (nmv + 3) / 4; }

size_t lds_bytes_main(int R, int seg_len)
{
    const int nsteps = (seg_len + 7 + 7) & ~7;
    const int rows = nsteps + 56;
    return kCurLdsBytes + (size_t)rows * ((R / 2 + 16 + 1) * 4);
}

size_t lds_bytes_edge(int nmv)
{
    const int seg = seg_len_edge(nmv);
    const int nsteps = (seg + 7 + 7) & ~7;
    const int rows = 3 * seg + 56 + nsteps;
    const int loaded = nmv + 63;
    return kCurLdsBytes + (size_t)(rows > loaded ? rows : loaded) * (17 * 4);
}

template <int R>
int launch_pair(const SadParams &p, const SadParams &pe, int n, int nseg, size_t lds_main, size_t lds_edge, hipStream_t st)
{
    static bool attr_done = false;  // one flag per R instantiation
    if (!attr_done) {
        XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<R, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<R, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((sad_surface_kernel<R, false>), dim3(n, nseg), dim3(8 * R), lds_main, st, p);
    XMI_HIP(hipGetLastError());
    hipLaunchKernelGGL((sad_surface_kernel<R, true>), dim3(n), dim3(64), lds_edge, st, pe);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

}  // namespace

extern "C" {

static bool range_ok(int R) { return R == 16 || R == 32 || R == 64 || R == 128; }

size_t xmi_sad_surface_elems(const xmi_sad_geom *g, int level)
{
    if (!g || level < 0 || level > 3 || !range_ok(g->range)) return 0;
    const size_t nctu = (size_t)(g->width / 64) * (size_t)(g->height / 64);
    return nctu * (size_t)(2 * g->range + 1) * (size_t)(row_elems(g->range, level) + level_blocks(level));
}

size_t xmi_sad_surface_index(const xmi_sad_geom *g, int level, int ctu, int blk, int dy, int dx)
{
    const int R = g->range;
    const int dxi = dx + R, dyi = dy + R;
    int slot;  // position of the block inside a tile
    if (level == 0) {
        const int by = blk >> 3, bx = blk & 7, by16 = by >> 1, bx16 = bx >> 1;
        const int z = (((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1));
        slot = ((by & 1) * 2 + (bx & 1)) * 16 + z;
    } else if (level == 1) {
        const int by16 = blk >> 2, bx16 = blk & 3;
        slot = (((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1));
    } else {
        slot = level == 2 ? blk : 0;
    }
    const size_t nmv = (size_t)(2 * R + 1), nb = (size_t)level_blocks(level);
    const size_t nctu = (size_t)(g->width / 64) * (size_t)(g->height / 64);
    const size_t row = (size_t)ctu * nmv + (size_t)dyi;
    if (dxi == 2 * R) return nctu * nmv * (size_t)row_elems(R, level) + row * nb + (size_t)slot;
    return row * (size_t)row_elems(R, level) + ((size_t)(dxi >> 4) * nb + (size_t)slot) * 16 + (size_t)(dxi & 15);
}

int xmi_sad_surface_dev(const xmi_sad_geom *g, const uint8_t *d_cur, const uint8_t *d_ref_origin,
                        uint16_t *d_sad8, uint16_t *d_sad16, uint32_t *d_sad32, uint32_t *d_sad64,
                        int ctu_begin, int ctu_end, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(g && d_cur && d_ref_origin && d_sad8 && d_sad16 && d_sad32 && d_sad64, "xmi_sad_surface_dev: null argument");
    XMI_ARG(g->width > 0 && g->height > 0 && g->width % 64 == 0 && g->height % 64 == 0,
            "xmi_sad_surface_dev: width/height (%d x %d) must be positive multiples of 64", g->width, g->height);
    const int R = g->range;
    XMI_ARG(range_ok(R), "xmi_sad_surface_dev: range %d not in {16,32,64,128}", R);
    const int nmv = 2 * R + 1;
    XMI_ARG(g->cur_stride >= g->width && g->cur_stride % 4 == 0 && g->ref_stride >= g->width + 2 * (R + 4) && g->ref_stride % 4 == 0,
            "xmi_sad_surface_dev: strides (cur %d, ref %d) must be multiples of 4 and cover width (+2*(range+4) for ref)",
            g->cur_stride, g->ref_stride);
    XMI_ARG(((uintptr_t)d_cur & 3) == 0 && ((uintptr_t)d_ref_origin & 3) == 0, "xmi_sad_surface_dev: frame pointers must be 4-byte aligned");
    XMI_ARG((((uintptr_t)d_sad8 | (uintptr_t)d_sad16 | (uintptr_t)d_sad32 | (uintptr_t)d_sad64) & 15) == 0,
            "xmi_sad_surface_dev: surface planes must be 16-byte aligned");
    const int nctu = (g->width / 64) * (g->height / 64);
    XMI_ARG(ctu_begin >= 0 && ctu_begin <= ctu_end && ctu_end <= nctu, "xmi_sad_surface_dev: CTU range [%d,%d) outside [0,%d)", ctu_begin, ctu_end, nctu);
    if (ctu_begin == ctu_end) return XMI_OK;

    SadParams p;
    p.cur = d_cur;
    p.ref = d_ref_origin;
    p.sad8 = d_sad8;
    p.sad16 = d_sad16;
    p.sad32 = d_sad32;
    p.sad64 = d_sad64;
    p.cur_stride = g->cur_stride;
    p.ref_stride = g->ref_stride;
    p.wc = g->width / 64;
    p.ctu_begin = ctu_begin;
    p.nctu_total = nctu;

    hipStream_t st = xmi::as_stream(stream);
    const int n = ctu_end - ctu_begin;

    // main: dx in [-R, R-1]; split the dy range into bands when there are too few CTUs to fill 256 CUs
    int nseg = 1;
    while (n * nseg < 256 && nseg < 8) nseg *= 2;
    p.seg_len = (nmv + nseg - 1) / nseg;
    nseg = (nmv + p.seg_len - 1) / p.seg_len;
    const size_t lds_main = lds_bytes_main(R, p.seg_len);
    XMI_ARG(lds_main <= 160 * 1024, "xmi_sad_surface_dev: LDS window %zu exceeds 160 KiB", lds_main);
    const size_t lds_edge = lds_bytes_edge(nmv);
    SadParams pe = p;
    pe.seg_len = seg_len_edge(nmv);
    // main: dx in [-R, R-1], R/8 wavefronts per CTU band.  edge: the dx = +R column, one wavefront per
    // CTU whose 4 lane groups split the dy range.
    switch (R) {
    case 16: rc = launch_pair<16>(p, pe, n, nseg, lds_main, lds_edge, st); break;
    case 32: rc = launch_pair<32>(p, pe, n, nseg, lds_main, lds_edge, st); break;
    case 64: rc = launch_pair<64>(p, pe, n, nseg, lds_main, lds_edge, st); break;
    default: rc = launch_pair<128>(p, pe, n, nseg, lds_main, lds_edge, st); break;
    }
    return rc;
}

int xmi_sad_surface(const xmi_sad_geom *g, const uint8_t *cur, const uint8_t *ref_origin, int pad,
                    uint16_t *sad8, uint16_t *sad16, uint32_t *sad32, uint32_t *sad64)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(g && cur && ref_origin && sad8 && sad16 && sad32 && sad64, "xmi_sad_surface: null argument");
    XMI_ARG(range_ok(g->range), "xmi_sad_surface: range %d not in {16,32,64,128}", g->range);
    XMI_ARG(pad >= g->range + 4 && pad % 4 == 0, "xmi_sad_surface: pad %d must be a multiple of 4 and >= range+4", pad);
    XMI_ARG(g->width > 0 && g->height > 0 && g->cur_stride >= g->width && g->ref_stride >= g->width + 2 * pad,
            "xmi_sad_surface: bad geometry");
    const size_t cur_bytes = (size_t)g->cur_stride * g->height;
    const size_t ref_rows = (size_t)g->height + 2 * (size_t)pad;
    const size_t ref_bytes = (size_t)g->ref_stride * ref_rows;
    const size_t e8 = xmi_sad_surface_elems(g, 0), e16 = xmi_sad_surface_elems(g, 1);
    const size_t e32 = xmi_sad_surface_elems(g, 2), e64 = xmi_sad_surface_elems(g, 3);
    uint8_t *d_cur = (uint8_t *)xmi::scratch(0, cur_bytes);
    uint8_t *d_ref = (uint8_t *)xmi::scratch(1, ref_bytes);
    uint16_t *d8 = (uint16_t *)xmi::scratch(2, e8 * 2);
    uint16_t *d16 = (uint16_t *)xmi::scratch(3, e16 * 2);
    uint32_t *d32 = (uint32_t *)xmi::scratch(4, e32 * 4);
    uint32_t *d64 = (uint32_t *)xmi::scratch(5, e64 * 4);
    if (!d_cur || !d_ref || !d8 || !d16 || !d32 || !d64) return XMI_ERR_HIP;
    const uint8_t *ref_base = ref_origin - (ptrdiff_t)pad * g->ref_stride - pad;
    XMI_HIP(hipMemcpy(d_cur, cur, cur_bytes, hipMemcpyHostToDevice));
    XMI_HIP(hipMemcpy(d_ref, ref_base, ref_bytes, hipMemcpyHostToDevice));
    rc = xmi_sad_surface_dev(g, d_cur, d_ref + (size_t)pad * g->ref_stride + pad, d8, d16, d32, d64, 0,
                             (g->width / 64) * (g->height / 64), nullptr);
    if (rc != XMI_OK) return rc;
    XMI_HIP(hipMemcpy(sad8, d8, e8 * 2, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy(sad16, d16, e16 * 2, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy(sad32, d32, e32 * 4, hipMemcpyDeviceToHost));
    XMI_HIP(hipMemcpy(sad64, d64, e64 * 4, hipMemcpyDeviceToHost));
    return XMI_OK;
}

}  // extern "C"
