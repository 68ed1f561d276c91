// Frame-wide integer-ME SAD surface for gfx950.
//
// For every 64x64 CTU and every integer MV in [-R,R]^2 the kernel produces the 85 SADs of the CU
// quad-tree (64 8x8, 16 16x16, 4 32x32, 1 64x64): exactly the values the reference obtains from
// xSad_AMP[idx](cur, 64, ref + dy*stride + dx, stride) (src/encode.cpp:2885-2931,
// src/pixel.cpp:578-696, 833-844). SAD is an exact integer sum, so any summation order is bit-exact.
//
// Mapping (one workgroup = one CTU x one band of dy rows x one slice of the dx range; four wavefronts of 16 dx each from R = 32 on):
//   lane = q*16 + gl*4 + b     z = q*4 + b = 16x16 block of the CTU in z-order (q = quadrant, b = block in it)
//                              gl = one of 4 dx-groups owned by the wavefront; a dx-group is 4
//                                   consecutive dx, produced at once by v_qsad_pk_u16_u8
//   The lane walks down the reference window one row per step. A reference row r meets the 8 rows y
//   of each 8x8 child, contributing to dy = r - y: a ring of 8 packed-u16 accumulators per child
//   holds the 8 dy values in flight; one completes per step and is written out. Reference and
//   current pixels are staged once in LDS; nothing is re-read from HBM.
//   16x16 = sum of the lane's 4 children (still packed u16, max 65280); 32x32 = DPP quad reduce-scatter
//   over the 4 lanes of a quadrant; 64x64 = swizzle/permute sum over the 4 quadrants.
//   The dx = +R column (one more than a multiple of 4) is produced by a short v_sad_u8 pass at the end.
//
// Output layout (chosen so that every store instruction of a wavefront writes one contiguous run):
//   main part, dx in [-R, R-1]: per (ctu, dy) a row of R/8 tiles of 16 consecutive dx; rows start on cache lines,
//   xmi::surface_row_pitch elements apart (the payload plus a few unwritten lines, see xmi_common.h):
//     sad8 : [ctu][dy][tile][c*16 + z][dx % 16]   c = (by8&1)*2 + (bx8&1), z = z-order of the parent 16x16
//     sad16: [ctu][dy][tile][z][dx % 16]
//     sad32: [ctu][dy][tile][q][dx % 16]          q = raster index of the 32x32 quadrant
//     sad64: [ctu][dy][tile][dx % 16]
//   edge part, dx = +R: appended after the main part of the same plane: [ctu][dy][block slot].
//   xmi_sad_surface_index() does this arithmetic.
//
// Measurement variants (never defined in the library build; tools/build_variant.sh compiles them into separate
// libraries): XMI_EXP_NOSTORE / XMI_EXP_NOCOMPUTE leave out the stores / the SAD instructions to time the other half,
// XMI_EXP_HALFCOMPUTE leaves out half of the SAD instructions, XMI_EXP_NOPIPE runs full-range bands through the plain step
// loop instead of the software pipeline, XMI_EXP_NOPEEL and XMI_EXP_NOSCHEDBAR undo two scheduling measures,
// XMI_EXP_X4 / XMI_EXP_GROUPS0 / XMI_EXP_STORE_AUX=<bits> are the store-width, instruction-order and cache-policy
// experiments, XMI_EXP_TIMELINE records per-workgroup clocks.
#include "xmi_common.h"

#include <type_traits>

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
    int seg_len;    // dy rows per workgroup (band height)
    int ctu_begin;
    int nctu_total;  // CTUs in the frame: the edge part of every plane starts after nctu_total main blocks
};

constexpr int kCurLdsBytes = 16 * 16 * 16;  // [row in 16x16 block][z][16 bytes]

// blocks per level, and elements per (ctu, dy) row of the main part of each level's plane
constexpr int level_blocks(int level) { return xmi::surface_level_blocks(level); }
constexpr int row_pitch(int R, int level) { return xmi::surface_row_pitch(R, level); }

__device__ __forceinline__ u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// Cache-policy bits of the surface stores in the pipelined loop: 2 = nt (non-temporal). The surface is written once and read
// by a later kernel, if at all; streamed past the caches the launch is 3 % shorter (0.354 -> 0.343 ms at 1080p, R = 64, A/B in
// one process per pair on two devices). With both halves of the kernel running the chip holds 2.04 GHz instead of the
// 2.39 GHz it holds for either half alone (rocm-smi sclk and s_memtime / s_memrealtime inside the kernel), so what shortens
// the launch is less energy per launch, not fewer cycles: scope bits (16 = sc1, 17 = sc0 sc1) gained 1 %, 16-byte stores 1 %,
// the order of the SAD instructions nothing.
#ifndef XMI_EXP_STORE_AUX
#define XMI_EXP_STORE_AUX 2
#endif
__device__ __forceinline__ u32x2 as_u32x2(u64 v) { return __builtin_bit_cast(u32x2, v); }

template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 x)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;   // quad_perm [1,0,3,2] and [2,3,0,1]
constexpr int kSwizzleXor16 = (16 << 10) | 0x1F;      // ds_swizzle bit mode: lane ^ 16 inside each half-wave

// Sum a packed pair of u16x2 registers (lo = dx 0,1; hi = dx 2,3) over the 4 lanes of a quad so that
// lane b (= lane & 3) ends up with the total for dx index b (a reduce-scatter: 8 VALU ops).
__device__ __forceinline__ u32 quad_reduce_scatter(u32 lo, u32 hi, int b)
{
    const bool b1 = b & 2, b0 = b & 1;
    const u32 keep = b1 ? hi : lo, send = b1 ? lo : hi;
    const u32 got = dpp_mov<kQuadXor2>(send);
    const u32 ya = (keep & 0xffffu) + (got & 0xffffu), yb = (keep >> 16) + (got >> 16);
    const u32 keep2 = b0 ? yb : ya, send2 = b0 ? ya : yb;
    return keep2 + dpp_mov<kQuadXor1>(send2);
}

#ifdef XMI_EXP_TIMELINE
__device__ unsigned long long g_timeline[4096 * 6];   // per workgroup: s_memtime at start / staged / main loop done / end, s_memrealtime at start / end
#endif

// Workgroup = one CTU x one band of dy rows x one slice of the dx range (blockIdx.z of SPLIT); R/8/SPLIT wavefronts, each
// producing 16 consecutive dx. lane = q*16 + gl*4 + b: q = 32x32 quadrant (raster), b = 16x16 block inside it (raster),
// gl = dx group of the wavefront.
// A lane keeps the 16 rows of its 16x16 block of the current CTU in 64 VGPRs (about 190 in all: two wavefronts per SIMD).
// SPLIT is chosen so that a workgroup is four wavefronts, one per SIMD: two workgroups share a CU, the older one has issue
// priority and runs as if alone, the younger one fills its gaps and the staging / last-column phases of its neighbours.
// Measured at 1080p, R = 64, steady state: rows re-read from LDS in every step (128 VGPRs, 8 wavefronts) 380 us per
// launch; rows in registers, 8 wavefronts (one workgroup per CU at a time, its last wavefronts run alone) 359 us.
template <int R, int SPLIT>
__global__ void __launch_bounds__(8 * R / SPLIT, 2) sad_surface_kernel(const SadParams p)
{
    constexpr bool kCurInRegs = true;
    constexpr int NT = 8 * R / SPLIT;                          // threads of the workgroup
    const int part = SPLIT > 1 ? (int)blockIdx.z : 0;          // dx slice: [-R + part * 2R/SPLIT, ... + 2R/SPLIT)
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char *cur_lds = lds;
    unsigned char *ref_lds = lds + kCurLdsBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef XMI_EXP_TIMELINE
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int nwaves = R / 8 / SPLIT;
    const int wave_g = part * nwaves + wave;                   // index of the wavefront's 16-dx tile in the row
    const int ctu = p.ctu_begin + blockIdx.x;
    const int cx0 = (ctu % p.wc) * 64, cy0 = (ctu / p.wc) * 64;
    constexpr int nmv = 2 * R + 1;

    // window geometry: rows [dy band] + 63, columns [first dx of the slice, last dx + 67] (the last 4 bytes serve the dx = +R column)
    const int seg_base = (int)blockIdx.y * p.seg_len;          // first dy index of this workgroup
    const int ndy = min(p.seg_len, nmv - seg_base);            // dy rows of this workgroup
    const int valid_rows = ndy + 63;
    constexpr int wb_dw = R / 2 / SPLIT + 17;                  // window width in dwords (odd: conflict-free pitch)
    constexpr int rls = wb_dw * 4;

    // ---- stage current CTU: cur_lds[(r16*16 + z)*16 + byte], z = q*4 + b ----
    for (int idx = tid; idx < 1024; idx += NT) {
        const int row = idx >> 4, cd = idx & 15;
        const int bx16 = cd >> 2, by16 = row >> 4, r16 = row & 15;
        const int zz = ((((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1)));
        const u32 v = *reinterpret_cast<const u32 *>(p.cur + (size_t)(cy0 + row) * p.cur_stride + cx0 + cd * 4);
        reinterpret_cast<u32 *>(cur_lds)[(r16 * 16 + zz) * 4 + (cd & 3)] = v;
    }
    // ---- stage the reference window; flat dword index, 8 independent loads in flight per thread ----
    {
        const uint8_t *src0 = p.ref + (ptrdiff_t)(cy0 - R + seg_base) * p.ref_stride + cx0 - R + part * (2 * R / SPLIT);
        const int total = valid_rows * wb_dw;
        for (int base = tid; base < total; base += 8 * NT) {
            u32 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int idx = base + k * NT;
                const int row = idx / wb_dw, col = idx - row * wb_dw;
                v[k] = idx < total ? *reinterpret_cast<const u32 *>(src0 + (ptrdiff_t)row * p.ref_stride + col * 4) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int idx = base + k * NT;
                const int row = idx / wb_dw, col = idx - row * wb_dw;
                if (idx < total) *reinterpret_cast<u32 *>(ref_lds + row * rls + col * 4) = v[k];
            }
        }
    }
    __syncthreads();

    // ---- per-lane geometry ----
    const int b = lane & 3, gl = (lane >> 2) & 3, q = lane >> 4;
    const int z = q * 4 + b;
    const int bx16 = (q & 1) * 2 + (b & 1), by16 = (q >> 1) * 2 + (b >> 1);
    const int g = wave * 4 + gl;                              // dx group inside the slice: dx = first dx of the slice + 4g .. +3
    const int nsteps = (ndy + 7 + 7) & ~7;
    const unsigned char *ref_lane = ref_lds + (by16 * 16) * rls + (bx16 * 4 + g) * 4;
    const u32 xor32_addr = (u32)((lane ^ 32) * 4);            // ds_bpermute address of the lane 32 away

    // output rows (uniform per workgroup and step) and per-lane byte offsets inside a row
    constexpr int ROW8 = row_pitch(R, 0), ROW16 = row_pitch(R, 1), ROW32 = row_pitch(R, 2), ROW64 = row_pitch(R, 3);   // elements between rows
    const size_t row0 = (size_t)ctu * nmv + seg_base;
    char *o8 = reinterpret_cast<char *>(p.sad8 + row0 * ROW8);
    char *o16 = reinterpret_cast<char *>(p.sad16 + row0 * ROW16);
    char *o32 = reinterpret_cast<char *>(p.sad32 + row0 * ROW32);
    char *o64 = reinterpret_cast<char *>(p.sad64 + row0 * ROW64);
    const u32 off8 = (u32)(wave_g * 1024 + z * 16 + gl * 4) * 2u;   // child c adds c*512 bytes
#ifdef XMI_EXP_X4
    const u32 off8x = (u32)(wave_g * 2048 + (z * 4 + gl) * 16);
#endif
    const u32 off16 = (u32)(wave_g * 256 + z * 16 + gl * 4) * 2u;
    const u32 off32 = (u32)(wave_g * 64 + q * 16 + gl * 4 + b) * 4u; // lane b stores the sum for dx = 4*gl + b
    const u32 off64 = (u32)(wave_g * 16 + gl * 4 + b) * 4u;

    // buffer resources over this workgroup's rows of the four planes (at most 257 rows of 33 KiB: far below the 4 GiB range)
    const __amdgpu_buffer_rsrc_t rs8 = __builtin_amdgcn_make_buffer_rsrc(o8, 0, ndy * (ROW8 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(o16, 0, ndy * (ROW16 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs32 = __builtin_amdgcn_make_buffer_rsrc(o32, 0, ndy * (ROW32 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs64 = __builtin_amdgcn_make_buffer_rsrc(o64, 0, ndy * (ROW64 * 4), 0x00020000);

    uint4 creg[kCurInRegs ? 16 : 1];
    if constexpr (kCurInRegs) {
#pragma unroll
        for (int r = 0; r < 16; r++) creg[r] = *reinterpret_cast<const uint4 *>(cur_lds + z * 16 + r * 256);
    }
    u64 acc[4][8];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[c][j] = 0;
#ifdef XMI_EXP_TIMELINE
    const unsigned long long t_staged = __builtin_amdgcn_s_memtime();
#endif

    // One 8-step group (the accumulator ring period). PHASE 0 / 2 are the first / last group of a full-range
    // band (ndy = 1 mod 8): there the row pairs whose dy falls outside the band are skipped at compile time
    // (y > u in the first group, y < u in the last); PHASE 1 is the generic group.
    auto group = [&](auto phase_tag, const int s0) {
        constexpr int PHASE = decltype(phase_tag)::value;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int s = s0 + u;
            // Without kCurInRegs: the current-CTU rows are loop invariant and holding them would cost 64 VGPRs. Launder
            // the LDS address every step so the reads stay inside the step.
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
                    uint4 c[4];
                    if constexpr (kCurInRegs) {
#pragma unroll
                        for (int yy = 0; yy < 4; yy++) c[yy] = creg[cy * 8 + h * 4 + yy];
                    } else {
#ifndef XMI_EXP_NOSCHEDBAR
                        asm volatile("" ::: "memory");  // bound the live range of the current-pixel registers:
                        __builtin_amdgcn_sched_barrier(0);  // neither the compiler nor the scheduler may hoist these reads
#endif
#pragma unroll
                        for (int yy = 0; yy < 4; yy++) c[yy] = *reinterpret_cast<const uint4 *>(cl + (cy * 8 + h * 4 + yy) * 256);
                    }
#pragma unroll
                    for (int yy = 0; yy < 4; yy++) {
                        const int y = h * 4 + yy;
                        if ((PHASE == 0 && y > u) || (PHASE == 2 && y < u)) continue;
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
            // the accumulators that started 7 steps ago are complete: dy index (band-relative) s-7
            const int jd = (u + 1) & 7;
            const int dyl = s - 7;
            const u64 v0 = acc[0][jd], v1 = acc[1][jd], v2 = acc[2][jd], v3 = acc[3][jd];
            const u64 v16 = v0 + v1 + v2 + v3;  // packed u16 lanes, each <= 65280: no carry between lanes
            // 32x32: reduce-scatter over the quadrant's 4 lanes; 64x64: sum over the 4 quadrants (lane bits 5:4)
            const u32 r32 = quad_reduce_scatter((u32)v16, (u32)(v16 >> 32), b);
            u32 t64 = r32 + (u32)__builtin_amdgcn_ds_swizzle((int)r32, kSwizzleXor16);
            t64 += (u32)__builtin_amdgcn_ds_bpermute((int)xor32_addr, (int)t64);
#ifdef XMI_EXP_NOSTORE
            if (dyl >= 0 && dyl < ndy && v16 == 0x1234567812345678ull && t64 == 77u) {
#else
            if (dyl >= 0 && dyl < ndy) {
#endif
                // buffer stores: the row offset is uniform (an SGPR), the lane offset a loop-invariant VGPR: no address arithmetic
                const int dyo = __builtin_amdgcn_readfirstlane(dyl);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v0), rs8, off8, dyo * (ROW8 * 2), 0);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v1), rs8, off8 + 512, dyo * (ROW8 * 2), 0);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v2), rs8, off8 + 1024, dyo * (ROW8 * 2), 0);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v3), rs8, off8 + 1536, dyo * (ROW8 * 2), 0);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v16), rs16, off16, dyo * (ROW16 * 2), 0);
                __builtin_amdgcn_raw_buffer_store_b32(r32, rs32, off32, dyo * (ROW32 * 4), 0);
                if (q == 0) __builtin_amdgcn_raw_buffer_store_b32(t64, rs64, off64, dyo * (ROW64 * 4), 0);
            }
        }
    };
    // The same group as a software pipeline, for full-range bands (ndy = 1 mod 8: every step from the eighth on completes a
    // row, so no store is conditional). A step is four runs of 16 SAD instructions (256 issue cycles each); everything else
    // is placed between the runs and is one step behind or ahead, so that a wavefront never waits for its own memory
    // operations and its stores leave one or two at a time instead of seven in a burst at the step's end:
    //   - the reference dwords of a run are read from LDS two runs earlier;
    //   - the row completed by step s-1 (P0..P3) goes through the pyramid during step s: the 16x16 sum, the quad
    //     reduce-scatter and the ds_swizzle after run 0, the ds_bpermute after run 1, the 64x64 total after run 2, with the
    //     seven stores spread over the three gaps.
    // sched_barrier keeps the compiler from pulling the pieces together again.
    u64 P0 = 0, P1 = 0, P2 = 0, P3 = 0;          // the row completed by the previous step
    u32 DA[5], DB[5];                            // the five dwords of the reference row that meet the upper / lower 8x8 children
    u64 RA[4], RB[4];                            // the same as byte-shifted dword pairs, put together at the run that uses them first
    auto load_row = [&](u32 (&d)[5], const int row) {
        const u32 *rr = reinterpret_cast<const u32 *>(ref_lane + row * rls);
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = rr[k];
    };
    auto make_pairs = [&](u64 (&w)[4], const u32 (&d)[5]) {
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = pack64(d[k], d[k + 1]);
    };
    // A run's SAD instructions have no side effects, so nothing but data dependences would hold them in place: the empty
    // asm statements (ordered among themselves and with the stores) take the run's operands in front of it and its results
    // behind it, which ties the run between the memory operations written before and after it.
    auto run16 = [&](auto phase_tag, const int u, const int cy, const int h, u64 (&w)[4]) {
        constexpr int PHASE = decltype(phase_tag)::value;
#ifndef XMI_EXP_NOCOMPUTE
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(w[k]));
#ifdef XMI_EXP_GROUPS0      // experiment: the four instructions that share a reference operand issue back to back
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int yy = 0; yy < 4; yy++) {
                const int y = h * 4 + yy;
                if ((PHASE == 0 && y > u) || (PHASE == 2 && y < u)) continue;
                const int j = (u - y) & 7;
                const uint4 c = creg[cy * 8 + y];
                u64 &a = acc[cy * 2 + (k >> 1)][j];
                const u32 cd = k == 0 ? c.x : k == 1 ? c.y : k == 2 ? c.z : c.w;
                a = __builtin_amdgcn_qsad_pk_u16_u8(w[k], cd, ((k & 1) == 0 && y == 0) ? 0ull : a);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int yy = 0; yy < 4; yy++) {
            const int y = h * 4 + yy;
            if ((PHASE == 0 && y > u) || (PHASE == 2 && y < u)) continue;
            const int j = (u - y) & 7;
            const uint4 c = creg[cy * 8 + y];
            const u64 a0 = (y == 0) ? 0ull : acc[cy * 2 + 0][j];
            const u64 a1 = (y == 0) ? 0ull : acc[cy * 2 + 1][j];
            acc[cy * 2 + 0][j] = __builtin_amdgcn_qsad_pk_u16_u8(w[1], c.y, __builtin_amdgcn_qsad_pk_u16_u8(w[0], c.x, a0));
            acc[cy * 2 + 1][j] = __builtin_amdgcn_qsad_pk_u16_u8(w[3], c.w, __builtin_amdgcn_qsad_pk_u16_u8(w[2], c.z, a1));
        }
#endif
#pragma unroll
        for (int yy = 0; yy < 4; yy++) {
            const int y = h * 4 + yy;
            if ((PHASE == 0 && y > u) || (PHASE == 2 && y < u)) continue;
            const int j = (u - y) & 7;
            asm volatile("" : "+v"(acc[cy * 2 + 0][j]));
            asm volatile("" : "+v"(acc[cy * 2 + 1][j]));
        }
#endif
    };
    // STORE: the previous step completed a row (dy index s - 8 of the band)
    auto pipe_step = [&](auto phase_tag, auto store_tag, const int s, const int u) {
        constexpr bool STORE = decltype(store_tag)::value;
        const int soff = STORE ? __builtin_amdgcn_readfirstlane(s - 8) : 0;
        u64 v16 = 0;
        u32 r32 = 0, t64 = 0, swz = 0, bp = 0;
        load_row(DB, 8 + s);
        __builtin_amdgcn_sched_barrier(0);
        make_pairs(RA, DA);
        run16(phase_tag, u, 0, 0, RA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STORE) {
            v16 = P0 + P1 + P2 + P3;
            r32 = quad_reduce_scatter((u32)v16, (u32)(v16 >> 32), b);
            swz = (u32)__builtin_amdgcn_ds_swizzle((int)r32, kSwizzleXor16);
#ifdef XMI_EXP_X4      // timing experiment only: the same bytes as two 16-byte stores per lane (not the documented layout)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ (u32)P0, (u32)(P0 >> 32), (u32)P1, (u32)(P1 >> 32) }, rs8, off8x, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
#elif !defined(XMI_EXP_NOSTORE)
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P0), rs8, off8, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P1), rs8, off8 + 512, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        run16(phase_tag, u, 0, 1, RA);
        __builtin_amdgcn_sched_barrier(0);
        load_row(DA, s + 1);
        if constexpr (STORE) {
            t64 = r32 + swz;
            bp = (u32)__builtin_amdgcn_ds_bpermute((int)xor32_addr, (int)t64);
#ifdef XMI_EXP_X4
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ (u32)P2, (u32)(P2 >> 32), (u32)P3, (u32)(P3 >> 32) }, rs8, off8x + 1024, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
#elif !defined(XMI_EXP_NOSTORE)
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P2), rs8, off8 + 1024, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P3), rs8, off8 + 1536, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        make_pairs(RB, DB);
#ifndef XMI_EXP_HALFCOMPUTE
        run16(phase_tag, u, 1, 0, RB);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STORE) {
            t64 += bp;
#ifdef XMI_EXP_NOSTORE
            if (v16 == 0x1234567812345678ull && t64 == 77u)
#endif
            {
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v16), rs16, off16, soff * (ROW16 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32(r32, rs32, off32, soff * (ROW32 * 4), XMI_EXP_STORE_AUX);
                if (q == 0) __builtin_amdgcn_raw_buffer_store_b32(t64, rs64, off64, soff * (ROW64 * 4), XMI_EXP_STORE_AUX);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef XMI_EXP_HALFCOMPUTE
        run16(phase_tag, u, 1, 1, RB);
#endif
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" : "+v"(acc[c][j]));
        const int jd = (u + 1) & 7;
        P0 = acc[0][jd], P1 = acc[1][jd], P2 = acc[2][jd], P3 = acc[3][jd];
        __builtin_amdgcn_sched_barrier(0);
    };
    using Tag0 = std::integral_constant<int, 0>;
    using Tag1 = std::integral_constant<int, 1>;
    using Tag2 = std::integral_constant<int, 2>;
    const bool peel = (ndy & 7) == 1 && nsteps >= 24;
#ifdef XMI_EXP_NOPEEL
    for (int s0 = 0; s0 < nsteps; s0 += 8) group(Tag1{}, s0);
#else
    if (peel) {
#ifdef XMI_EXP_NOPIPE
        group(Tag0{}, 0);
        for (int s0 = 8; s0 < nsteps - 8; s0 += 8) group(Tag1{}, s0);
        group(Tag2{}, nsteps - 8);
#else
        load_row(DA, 0);
#pragma unroll
        for (int u = 0; u < 8; u++) pipe_step(Tag0{}, std::false_type{}, u, u);
        for (int s0 = 8; s0 < nsteps - 8; s0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) pipe_step(Tag1{}, std::true_type{}, s0 + u, u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) pipe_step(Tag2{}, std::true_type{}, nsteps - 8 + u, u);
        {   // the row completed by the last step
            const int soff = __builtin_amdgcn_readfirstlane(nsteps - 8);
            const u64 v16 = P0 + P1 + P2 + P3;
            const u32 r32 = quad_reduce_scatter((u32)v16, (u32)(v16 >> 32), b);
            u32 t64 = r32 + (u32)__builtin_amdgcn_ds_swizzle((int)r32, kSwizzleXor16);
            t64 += (u32)__builtin_amdgcn_ds_bpermute((int)xor32_addr, (int)t64);
#ifdef XMI_EXP_NOSTORE
            if (v16 == 0x1234567812345678ull && t64 == 77u)
#endif
            {
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P0), rs8, off8, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P1), rs8, off8 + 512, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P2), rs8, off8 + 1024, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(P3), rs8, off8 + 1536, soff * (ROW8 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v16), rs16, off16, soff * (ROW16 * 2), XMI_EXP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32(r32, rs32, off32, soff * (ROW32 * 4), XMI_EXP_STORE_AUX);
                if (q == 0) __builtin_amdgcn_raw_buffer_store_b32(t64, rs64, off64, soff * (ROW64 * 4), XMI_EXP_STORE_AUX);
            }
        }
#endif
    } else {
        for (int s0 = 0; s0 < nsteps; s0 += 8) group(Tag1{}, s0);
    }
#endif

#ifdef XMI_EXP_TIMELINE
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    // ---- the dx = +R column: one more than a multiple of 4, so it gets a plain v_sad_u8 pass of its own.
    // lane (z, gl) of wavefront w takes the band rows d = (4w + gl), +4*nwaves, ...; the reference bytes of
    // dx = +R start at dword R/2/SPLIT + 4*bx16 of the last slice's window, already staged.
    if (part == SPLIT - 1) {
        constexpr int nlan = 4 * nwaves;
        const unsigned char *ref_e = ref_lds + (by16 * 16) * rls + (R / 2 / SPLIT + bx16 * 4) * 4;
        const unsigned char *cl = cur_lds + z * 16;
        const size_t rows_main = (size_t)p.nctu_total * nmv;
        uint16_t *e8 = p.sad8 + rows_main * ROW8 + row0 * 64;
        uint16_t *e16 = p.sad16 + rows_main * ROW16 + row0 * 16;
        u32 *e32 = p.sad32 + rows_main * ROW32 + row0 * 4;
        u32 *e64 = p.sad64 + rows_main * ROW64 + row0;
        for (int d0 = 0; d0 < ndy; d0 += nlan) {
            const int d = d0 + wave * 4 + gl;
            const int dr = min(d, ndy - 1);                    // lanes past the band recompute its last row (not stored)
            u32 a[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int cy = 0; cy < 2; cy++)
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const u32 *rr = reinterpret_cast<const u32 *>(ref_e + (cy * 8 + y + dr) * rls);
                    const uint4 c = *reinterpret_cast<const uint4 *>(cl + (cy * 8 + y) * 256);
                    a[cy * 2 + 0] = __builtin_amdgcn_sad_u8(rr[1], c.y, __builtin_amdgcn_sad_u8(rr[0], c.x, a[cy * 2 + 0]));
                    a[cy * 2 + 1] = __builtin_amdgcn_sad_u8(rr[3], c.w, __builtin_amdgcn_sad_u8(rr[2], c.z, a[cy * 2 + 1]));
                }
            const u32 s16 = a[0] + a[1] + a[2] + a[3];
            u32 s32 = s16 + dpp_mov<kQuadXor1>(s16);
            s32 += dpp_mov<kQuadXor2>(s32);
            u32 s64 = s32 + (u32)__builtin_amdgcn_ds_swizzle((int)s32, kSwizzleXor16);
            s64 += (u32)__builtin_amdgcn_ds_bpermute((int)xor32_addr, (int)s64);
            if (d < ndy) {
                uint16_t *r8 = e8 + (size_t)d * 64 + z;
                r8[0] = (uint16_t)a[0];
                r8[16] = (uint16_t)a[1];
                r8[32] = (uint16_t)a[2];
                r8[48] = (uint16_t)a[3];
                e16[(size_t)d * 16 + z] = (uint16_t)s16;
                if (b == 0) e32[(size_t)d * 4 + q] = s32;
                if (z == 0) e64[d] = s64;
            }
        }
    }
#ifdef XMI_EXP_TIMELINE
#ifdef XMI_EXP_TIMELINE_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the end stamps then include this wavefront's outstanding stores
#endif
    if (tid == 0 && blockIdx.x * SPLIT + part < 4096) {
        unsigned long long *t = g_timeline + (blockIdx.x * SPLIT + part) * 6;
        t[0] = t_start;
        t[1] = t_staged;
        t[2] = t_loop;
        t[3] = __builtin_amdgcn_s_memtime();
        t[4] = r_start;
        t[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

constexpr int split_of(int R) { return R >= 64 ? R / 32 : 1; }      // four wavefronts per workgroup from R = 32 on

size_t lds_bytes(int R, int seg_len)
{
    const int nsteps = (seg_len + 7 + 7) & ~7;
    const int rows = nsteps + 56;
    return kCurLdsBytes + (size_t)rows * ((R / 2 / split_of(R) + 17) * 4);
}

template <int R>
int launch_surface(const SadParams &p, int n, int nseg, size_t lds, hipStream_t st)
{
    constexpr int SPLIT = split_of(R);
    hipLaunchKernelGGL((sad_surface_kernel<R, SPLIT>), dim3(n, nseg, SPLIT), dim3(8 * R / SPLIT), lds, st, p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

}  // namespace

// The 160 KiB LDS opt-in of the four instantiations, for the device that is current: called once per device
// binding (under the library mutex), never from a launch path, so no launch touches function attributes.
int xmi::sad_surface_prepare_device()
{
    XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<16, split_of(16)>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<32, split_of(32)>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<64, split_of(64)>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    XMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sad_surface_kernel<128, split_of(128)>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return XMI_OK;
}

extern "C" {

static bool range_ok(int R) { return R == 16 || R == 32 || R == 64 || R == 128; }

size_t xmi_sad_surface_elems(const xmi_sad_geom *g, int level)
{
    if (!g || level < 0 || level > 3 || !range_ok(g->range)) return 0;
    const size_t nctu = (size_t)(g->width / 64) * (size_t)(g->height / 64);
    return nctu * (size_t)(2 * g->range + 1) * (size_t)(row_pitch(g->range, level) + level_blocks(level));
}

#ifdef XMI_EXP_TIMELINE
int xmi_debug_timeline(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)n * 48) == hipSuccess ? 0 : -1;
}
#endif

size_t xmi_sad_surface_index(const xmi_sad_geom *g, int level, int ctu, int blk, int dy, int dx)
{
    const size_t nctu = (size_t)(g->width / 64) * (size_t)(g->height / 64);
    return xmi::surface_index(g->range, nctu, level, ctu, blk, dy, dx);
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

    // split the dy range into bands when there are too few CTUs to give every CU its two workgroups
    int nseg = 1;
    while (n * nseg * split_of(R) < 512 && nseg < 8) nseg *= 2;
    p.seg_len = (nmv + nseg - 1) / nseg;
    nseg = (nmv + p.seg_len - 1) / p.seg_len;
    const size_t lds = lds_bytes(R, p.seg_len);
    XMI_ARG(lds <= 160 * 1024, "xmi_sad_surface_dev: LDS window %zu exceeds 160 KiB", lds);
    switch (R) {
    case 16: return launch_surface<16>(p, n, nseg, lds, st);
    case 32: return launch_surface<32>(p, n, nseg, lds, st);
    case 64: return launch_surface<64>(p, n, nseg, lds, st);
    default: return launch_surface<128>(p, n, nseg, lds, st);
    }
}

int xmi_sad_surface(const xmi_sad_geom *g, const uint8_t *cur, const uint8_t *ref_origin, int pad,
                    uint16_t *sad8, uint16_t *sad16, uint32_t *sad32, uint32_t *sad64)
{
    int rc = xmi::enter_device();
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
    // scratch slots 0-5 are shared by every caller of this entry point: hold the lock until the last copy has returned
    std::lock_guard<std::mutex> lk(xmi::picture_call_mutex());
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
