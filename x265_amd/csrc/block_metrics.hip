// Batched block-distortion metrics for gfx950: SAD, SSD, HM-style SATD and x264-style SATD over packed
// n x n block pairs.
//
//   SAD        xSad4x4..xSad64x64                       src/pixel.cpp:546-696
//   SSD        x265_pixel_ssd_* (exact sum of squares)  src/pixel.cpp:793-816
//   SATD (HM)  xCalcHADs4x4 ((s+1)>>1), xCalcHADs8x8 ((s+2)>>2), 16/32/64 = sum of rounded 8x8 tiles
//                                                        src/pixel.cpp:63-458
//   SATD (x264) x265_pixel_satd_{4x4,8x8,16x16}_sse4 = (sum over 4x4 tiles of sum|H4 D H4|) / 2
//                                                        src/x86/pixel-a.asm:1645, 3455, 1787
//
// SATD: one lane owns one tile (4x4 or 8x8) of one block, keeps the whole residual tile in registers, runs the
// 2-D Hadamard there, and the tiles of a block (consecutive lanes) are summed with a segmented
// wavefront reduction. SAD / SSD of 16-byte aligned arrays: streaming kernel (sum_metric_kernel below).
// HBM traffic is the algorithmic 2*n*n + 4 bytes per pair.
#include "xmi_common.h"

namespace {

typedef unsigned int u32;

enum { M_SAD = XMI_METRIC_SAD, M_SSD = XMI_METRIC_SSD, M_SATD_HM = XMI_METRIC_SATD_HM, M_SATD_X264 = XMI_METRIC_SATD_X264 };

template <int T>
__device__ __forceinline__ void hadamard_1d(int *v)
{
#pragma unroll
    for (int len = 1; len < T; len <<= 1)
#pragma unroll
        for (int i = 0; i < T; i += 2 * len)
#pragma unroll
            for (int j = i; j < i + len; j++) {
                const int a = v[j], b = v[j + len];
                v[j] = a + b;
                v[j + len] = a - b;
            }
}

// value of one T x T tile (T = 4 or 8) at a/b with row stride n
template <int T, int METRIC>
__device__ __forceinline__ u32 tile_value(const uint8_t *a, const uint8_t *b, int n)
{
    int d[T][T];
#pragma unroll
    for (int r = 0; r < T; r++) {
        if (T == 8) {
            const uint2 va = *reinterpret_cast<const uint2 *>(a + r * n), vb = *reinterpret_cast<const uint2 *>(b + r * n);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                d[r][c] = (int)((va.x >> (8 * c)) & 255) - (int)((vb.x >> (8 * c)) & 255);
                d[r][c + 4] = (int)((va.y >> (8 * c)) & 255) - (int)((vb.y >> (8 * c)) & 255);
            }
        } else {
            const u32 va = *reinterpret_cast<const u32 *>(a + r * n), vb = *reinterpret_cast<const u32 *>(b + r * n);
#pragma unroll
            for (int c = 0; c < 4; c++) d[r][c] = (int)((va >> (8 * c)) & 255) - (int)((vb >> (8 * c)) & 255);
        }
    }
    u32 s = 0;
    if (METRIC == M_SAD || METRIC == M_SSD) {
#pragma unroll
        for (int r = 0; r < T; r++)
#pragma unroll
            for (int c = 0; c < T; c++) s += METRIC == M_SAD ? (u32)abs(d[r][c]) : (u32)(d[r][c] * d[r][c]);
        return s;
    }
#pragma unroll
    for (int r = 0; r < T; r++) hadamard_1d<T>(d[r]);
#pragma unroll
    for (int c = 0; c < T; c++) {
        int col[T];
#pragma unroll
        for (int r = 0; r < T; r++) col[r] = d[r][c];
        hadamard_1d<T>(col);
#pragma unroll
        for (int r = 0; r < T; r++) s += (u32)abs(col[r]);
    }
    if (METRIC == M_SATD_HM) return T == 4 ? (s + 1) >> 1 : (s + 2) >> 2;  // each tile rounded on its own
    return s;                                                            // x264 form: halved after the block sum
}

// SATD of one T x T tile with packed 16-bit arithmetic: the residual is unpacked into pairs of neighbouring
// columns and handed to xmi::hadamard_abs_sum_packed. Returns the same value as tile_value<T, M_SATD_*>.
using xmi::s16x2;

// swz: row r is read from a + ((r ^ swz) * n). A Hadamard transform of rows permuted by r -> r ^ swz is the transform of
// the rows in order with some output rows negated (its rows are characters of the group (Z2)^k: h(i, j ^ g) = h(i, j)
// h(i, g)), so the sum of magnitudes does not change; callers use it to spread their LDS reads over the banks.
template <int T, int METRIC>
__device__ __forceinline__ u32 tile_satd_packed(const uint8_t *a, const uint8_t *b, int n, int swz = 0)
{
    constexpr int P = T / 2;                  // column pairs per row
    s16x2 d[T][P];
#pragma unroll
    for (int r = 0; r < T; r++) {
        u32 wa[T / 4], wb[T / 4];
        if (T == 8) {
            const uint2 va = *reinterpret_cast<const uint2 *>(a + (r ^ swz) * n), vb = *reinterpret_cast<const uint2 *>(b + (r ^ swz) * n);
            wa[0] = va.x;
            wa[T / 4 - 1] = va.y;
            wb[0] = vb.x;
            wb[T / 4 - 1] = vb.y;
        } else {
            wa[0] = *reinterpret_cast<const u32 *>(a + r * n);
            wb[0] = *reinterpret_cast<const u32 *>(b + r * n);
        }
#pragma unroll
        for (int w = 0; w < T / 4; w++) {
            const s16x2 a01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wa[w], 0x0c010c00u));
            const s16x2 a23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wa[w], 0x0c030c02u));
            const s16x2 b01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wb[w], 0x0c010c00u));
            const s16x2 b23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, wb[w], 0x0c030c02u));
            d[r][2 * w] = a01 - b01;
            d[r][2 * w + 1] = a23 - b23;
        }
    }
    const u32 s = xmi::hadamard_abs_sum_packed<T>(d);
    if (METRIC == M_SATD_HM) return T == 4 ? (s + 1) >> 1 : (s + 2) >> 2;
    return s;
}

template <int T, int METRIC>
__global__ void __launch_bounds__(256) metric_kernel(const uint8_t *a, const uint8_t *b, u32 *out, int n, int count)
{
    const int tiles_per_row = n / T, tiles = tiles_per_row * tiles_per_row;   // tiles is a power of two <= 64
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long blk = gid / tiles;
    const int tile = (int)(gid % tiles);
    u32 v = 0;
    if (blk < count) {
        const size_t off = (size_t)blk * n * n + (size_t)(tile / tiles_per_row) * T * n + (size_t)(tile % tiles_per_row) * T;
        v = (METRIC == M_SATD_HM || METRIC == M_SATD_X264) ? tile_satd_packed<T, METRIC>(a + off, b + off, n) : tile_value<T, METRIC>(a + off, b + off, n);
    }
    for (int sh = 1; sh < tiles; sh <<= 1) v += (u32)__shfl_xor((int)v, sh);  // tiles of a block sit in consecutive lanes
    if (blk < count && tile == 0) out[blk] = METRIC == M_SATD_X264 ? v >> 1 : v;
}

// SAD and SSD are plain sums over the block, so any assignment of bytes to lanes will do: a lane takes 16
// consecutive bytes of both operands per step (fully coalesced 16-byte loads, 1 KB per wavefront instruction)
// and the n*n/16 lanes of a block meet in a segmented DPP reduction; a 64x64 block is four steps of one
// wavefront. SSD of four byte pairs = a.a + b.b - 2 a.b with three v_dot4_u32_u8.
template <int W>
__device__ __forceinline__ int group_sum_to_last_lane(int v)
{
    if (W >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    if (W >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    if (W >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    if (W >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    if (W >= 32) v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);    // row_bcast:15 into rows 1 and 3
    if (W >= 64) v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);    // row_bcast:31 into rows 2 and 3
    return v;
}

template <int N, int METRIC>
__global__ void __launch_bounds__(256) sum_metric_kernel(const uint8_t *a, const uint8_t *b, u32 *out, int count)
{
    constexpr int LPB = N * N / 16 < 64 ? N * N / 16 : 64;     // lanes per block within one step
    constexpr int STEPS = N * N / 16 / LPB;                    // steps per block (64x64: 4)
    const int lane = threadIdx.x & 63;
    const size_t total = (size_t)count * N * N;                // bytes per operand, a multiple of 16
    const size_t nchunk = (total + 1023) / 1024;               // one chunk = one wavefront step = 1 KB per operand
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    for (size_t c0 = wave0 * STEPS; c0 < nchunk; c0 += nwaves * STEPS) {
        u32 acc = 0;
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            const size_t e = (c0 + st) * 1024 + (size_t)lane * 16;
            if (e < total) {
                const uint4 va = *reinterpret_cast<const uint4 *>(a + e), vb = *reinterpret_cast<const uint4 *>(b + e);
                const u32 wa[4] = { va.x, va.y, va.z, va.w }, wb[4] = { vb.x, vb.y, vb.z, vb.w };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (METRIC == M_SAD) {
                        acc = __builtin_amdgcn_sad_u8(wa[k], wb[k], acc);
                    } else {
                        acc = __builtin_amdgcn_udot4(wa[k], wa[k], acc, false);
                        acc = __builtin_amdgcn_udot4(wb[k], wb[k], acc, false);
                        acc -= 2u * __builtin_amdgcn_udot4(wa[k], wb[k], 0u, false);
                    }
                }
            }
        }
        const u32 v = (u32)group_sum_to_last_lane<LPB>((int)acc);
        const size_t blk = (c0 * 1024 + (size_t)lane * 16) / ((size_t)N * N);
        if ((lane & (LPB - 1)) == LPB - 1 && blk < (size_t)count) out[blk] = v;
    }
}

template <int N, int METRIC>
int launch_sum(const uint8_t *a, const uint8_t *b, u32 *out, int count, hipStream_t st)
{
    constexpr int STEPS = N * N / 16 < 64 ? 1 : N * N / 16 / 64;
    const size_t nchunk = ((size_t)count * N * N + 1023) / 1024, work = (nchunk + STEPS - 1) / STEPS;
    const unsigned blocks = (unsigned)((work + 3) / 4 < 16384 ? (work + 3) / 4 : 16384);
    hipLaunchKernelGGL((sum_metric_kernel<N, METRIC>), dim3(blocks), dim3(256), 0, st, a, b, out, count);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

template <int METRIC>
int launch_sum_n(int n, const uint8_t *a, const uint8_t *b, u32 *out, int count, hipStream_t st)
{
    switch (n) {
    case 4: return launch_sum<4, METRIC>(a, b, out, count, st);
    case 8: return launch_sum<8, METRIC>(a, b, out, count, st);
    case 16: return launch_sum<16, METRIC>(a, b, out, count, st);
    case 32: return launch_sum<32, METRIC>(a, b, out, count, st);
    default: return launch_sum<64, METRIC>(a, b, out, count, st);
    }
}

// SATD with coalesced loads: a wavefront takes 64 transform tiles (whole blocks) per step. Each lane loads 16
// consecutive bytes of both operands per KB of the chunk and scatters them into a tile-major LDS patch; then lane p reads
// the tile at position p and transforms it in registers as metric_kernel does. The tiles of a block occupy the positions
// of that block in some order; the block sum does not care which. Bank layout (8-byte LDS stores are served in groups of
// 16 lanes over 32 banks, 8-byte loads in groups of 16 or 32 lanes over 32 or 64 banks):
//   8x8 tiles: 64 bytes per tile, no padding, so a tile covers half of the banks, the half chosen by the parity of its
//     position. A store instruction writes 8 rows to each of two tiles whose indices differ by 2: tile t sits at position
//     t ^ ((t >> 1) & 1), which gives the two different parities. With N = 64 it writes 4 rows to each of four tiles, and
//     w(t) moves the rows of every other tile pair to slots r ^ 4. Lane p loads slot s ^ f(p) into register row s, with
//     f different among the lanes that share banks. Both are XOR permutations of the rows, which SATD does not see
//     (tile_satd_packed).
//   4x4 tiles of 8x8 blocks: 16 bytes per tile, no padding: a lane's two row pairs go out as two 8-byte stores, a tile
//     comes back as one 16-byte load; tile t sits at position t ^ ((t >> 3) & 1) so that blocks two apart use different
//     banks. 4x4 tiles of 16x16 blocks keep a 20-byte pitch (dword stores and loads, conflict free as they are).
template <int T, int METRIC, int N>
__global__ void __launch_bounds__(256) satd_staged_kernel(const uint8_t *a, const uint8_t *b, u32 *out, int count)
{
    constexpr int PITCH = T == 8 ? 64 : N == 8 ? 16 : 20;
    constexpr int TPR = N / T, TILES = TPR * TPR;              // tiles per block row / per block
    constexpr int SUB = T == 8 ? 1 : 4;                        // patches per loop iteration: eight 16-byte loads in flight per lane either way
    constexpr int CHUNK = 64 * T * T;                          // bytes per operand and patch
    constexpr int PIECES = CHUNK / 1024;                       // 16-byte pieces per lane and patch
    __shared__ __align__(16) uint8_t s_patch[4][2][64 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *pa = s_patch[wave][0], *pb = s_patch[wave][1];
    const size_t total = (size_t)count * N * N;
    const size_t nchunk = (total + CHUNK * SUB - 1) / (CHUNK * SUB);
    auto wswz = [](int t) { return N == 64 ? ((t >> 2) & 1) * 4 : 0; };       // store-side row permutation of 8x8 tiles
    for (size_t c = (size_t)blockIdx.x * 4 + wave; c < nchunk; c += (size_t)gridDim.x * 4) {
        uint4 ra[SUB][PIECES], rb[SUB][PIECES];
#pragma unroll
        for (int u = 0; u < SUB; u++)
#pragma unroll
            for (int k = 0; k < PIECES; k++) {
                const size_t e = (c * SUB + u) * CHUNK + k * 1024 + lane * 16;
                if (e < total) ra[u][k] = *reinterpret_cast<const uint4 *>(a + e), rb[u][k] = *reinterpret_cast<const uint4 *>(b + e);
            }
#pragma unroll
      for (int u = 0; u < SUB; u++) {
        const size_t base = (c * SUB + u) * CHUNK;
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            const int o = k * 1024 + lane * 16;                // byte offset inside the patch
            if (base + o >= total) continue;
            const uint4 va = ra[u][k], vb = rb[u][k];
            const int bi = o / (N * N), ob = o % (N * N), row = ob / N, col = ob % N;
            if (T == 8 && N == 8) {                            // two rows of one tile: the patch is the chunk itself
                *reinterpret_cast<uint4 *>(pa + o) = va;
                *reinterpret_cast<uint4 *>(pb + o) = vb;
            } else if (T == 8) {                               // one row piece of two neighbouring tiles
                const int t = bi * TILES + (row / 8) * TPR + col / 8, flip = (t >> 1) & 1;      // t is even
                const int slot = ((row % 8) ^ wswz(t)) * 8, at0 = (t ^ flip) * PITCH + slot, at1 = ((t + 1) ^ flip) * PITCH + slot;
                *reinterpret_cast<uint2 *>(pa + at0) = make_uint2(va.x, va.y);
                *reinterpret_cast<uint2 *>(pa + at1) = make_uint2(va.z, va.w);
                *reinterpret_cast<uint2 *>(pb + at0) = make_uint2(vb.x, vb.y);
                *reinterpret_cast<uint2 *>(pb + at1) = make_uint2(vb.z, vb.w);
            } else if (N == 8) {                               // 4x4 tiles of an 8x8 block: two rows x two tiles
                const int t = bi * TILES + (row / 4) * TPR, flip = (t >> 3) & 1;                // t is even
                const int at0 = (t ^ flip) * PITCH + (row % 4) * 4, at1 = ((t + 1) ^ flip) * PITCH + (row % 4) * 4;
                *reinterpret_cast<uint2 *>(pa + at0) = make_uint2(va.x, va.z);
                *reinterpret_cast<uint2 *>(pa + at1) = make_uint2(va.y, va.w);
                *reinterpret_cast<uint2 *>(pb + at0) = make_uint2(vb.x, vb.z);
                *reinterpret_cast<uint2 *>(pb + at1) = make_uint2(vb.y, vb.w);
            } else {                                           // 4x4 tiles of a 16x16 block: one row x four tiles
                const int at = (bi * TILES + (row / 4) * TPR) * PITCH + (row % 4) * 4;
                *reinterpret_cast<u32 *>(pa + at) = va.x;
                *reinterpret_cast<u32 *>(pa + at + PITCH) = va.y;
                *reinterpret_cast<u32 *>(pa + at + 2 * PITCH) = va.z;
                *reinterpret_cast<u32 *>(pa + at + 3 * PITCH) = va.w;
                *reinterpret_cast<u32 *>(pb + at) = vb.x;
                *reinterpret_cast<u32 *>(pb + at + PITCH) = vb.y;
                *reinterpret_cast<u32 *>(pb + at + 2 * PITCH) = vb.z;
                *reinterpret_cast<u32 *>(pb + at + 3 * PITCH) = vb.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32 v = tile_satd_packed<T, METRIC>(pa + lane * PITCH, pb + lane * PITCH, T, T == 8 ? ((lane >> 1) & 7) ^ ((lane >> 4) & 1) : 0);
        __builtin_amdgcn_wave_barrier();
        v = (u32)group_sum_to_last_lane<TILES>((int)v);
        const size_t blk = (base + (size_t)lane * T * T) / ((size_t)N * N);
        if ((lane & (TILES - 1)) == TILES - 1 && blk < (size_t)count) out[blk] = METRIC == M_SATD_X264 ? v >> 1 : v;
      }
    }
}

// 4x4 blocks: a tile is 16 consecutive bytes, so a lane's 16-byte load is its tile. Eight tiles per lane and iteration (sixteen
// loads in flight), grid-stride: a one-tile-per-thread launch is a stream of very short wavefronts and reads 15 % slower.
template <int METRIC>
__global__ void __launch_bounds__(256) satd4_kernel(const uint8_t *a, const uint8_t *b, u32 *out, int count)
{
    constexpr int K = 8;            // tiles per lane and iteration
    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)count, nwaves = (size_t)gridDim.x * 4;
    for (size_t t0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * K); t0 < n; t0 += nwaves * (64 * K)) {
        uint4 va[K], vb[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const size_t t = t0 + k * 64 + lane;
            va[k] = vb[k] = make_uint4(0, 0, 0, 0);
            if (t < n) va[k] = *reinterpret_cast<const uint4 *>(a + t * 16), vb[k] = *reinterpret_cast<const uint4 *>(b + t * 16);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const size_t t = t0 + k * 64 + lane;
            const u32 v = tile_satd_packed<4, METRIC>(reinterpret_cast<const uint8_t *>(&va[k]), reinterpret_cast<const uint8_t *>(&vb[k]), 4);
            if (t < n) out[t] = METRIC == M_SATD_X264 ? v >> 1 : v;
        }
    }
}

template <int METRIC>
int launch_satd4(const uint8_t *a, const uint8_t *b, u32 *out, int count, hipStream_t st)
{
    const size_t iters = ((size_t)count + 511) / 512;
    const unsigned blocks = (unsigned)((iters + 3) / 4 < 16384 ? (iters + 3) / 4 : 16384);
    hipLaunchKernelGGL((satd4_kernel<METRIC>), dim3(blocks), dim3(256), 0, st, a, b, out, count);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

template <int T, int METRIC, int N>
int launch_staged(const uint8_t *a, const uint8_t *b, u32 *out, int count, hipStream_t st)
{
    constexpr size_t per_iter = (size_t)64 * T * T * (T == 8 ? 1 : 4);
    const size_t nchunk = ((size_t)count * N * N + per_iter - 1) / per_iter;
    const unsigned blocks = (unsigned)((nchunk + 3) / 4 < 8192 ? (nchunk + 3) / 4 : 8192);
    hipLaunchKernelGGL((satd_staged_kernel<T, METRIC, N>), dim3(blocks), dim3(256), 0, st, a, b, out, count);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

template <int T, int METRIC>
int launch(const uint8_t *a, const uint8_t *b, u32 *out, int n, int count, hipStream_t st)
{
    const long long lanes = (long long)count * (n / T) * (n / T);
    const unsigned blocks = (unsigned)((lanes + 255) / 256);
    hipLaunchKernelGGL((metric_kernel<T, METRIC>), dim3(blocks), dim3(256), 0, st, a, b, out, n, count);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

}  // namespace

extern "C" int xmi_block_metric_batch_dev(int metric, int n, int count, const uint8_t *d_a, const uint8_t *d_b,
                                          uint32_t *d_out, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(count >= 0, "xmi_block_metric_batch_dev: negative count");
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32 || n == 64, "xmi_block_metric_batch_dev: block size %d not in {4,8,16,32,64}", n);
    if (count == 0) return XMI_OK;
    XMI_ARG(d_a && d_b && d_out, "xmi_block_metric_batch_dev: null argument");
    XMI_ARG((((uintptr_t)d_a | (uintptr_t)d_b) & (n == 4 ? 3 : 7)) == 0, "xmi_block_metric_batch_dev: block arrays must be %d-byte aligned", n == 4 ? 4 : 8);
    hipStream_t st = xmi::as_stream(stream);
    if (metric == XMI_METRIC_SATD_TAB) {
        // the table the encoder calls (src/pixel.cpp:529-542): x264 form for 4/8/16, HM 8x8 tiles for 32
        XMI_ARG(n <= 32, "xmi_block_metric_batch_dev: xSatdN has no %dx%d entry", n, n);
        metric = n == 32 ? XMI_METRIC_SATD_HM : XMI_METRIC_SATD_X264;
    }
    if ((metric == XMI_METRIC_SAD || metric == XMI_METRIC_SSD) && (((uintptr_t)d_a | (uintptr_t)d_b) & 15) == 0)
        return metric == XMI_METRIC_SAD ? launch_sum_n<M_SAD>(n, d_a, d_b, d_out, count, st) : launch_sum_n<M_SSD>(n, d_a, d_b, d_out, count, st);
    if ((((uintptr_t)d_a | (uintptr_t)d_b) & 15) == 0) {
        if (metric == XMI_METRIC_SATD_HM && n == 8) return launch_staged<8, M_SATD_HM, 8>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_HM && n == 16) return launch_staged<8, M_SATD_HM, 16>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_HM && n == 32) return launch_staged<8, M_SATD_HM, 32>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_HM && n == 64) return launch_staged<8, M_SATD_HM, 64>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_X264 && n == 4) return launch_satd4<M_SATD_X264>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_HM && n == 4) return launch_satd4<M_SATD_HM>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_X264 && n == 8) return launch_staged<4, M_SATD_X264, 8>(d_a, d_b, d_out, count, st);
        if (metric == XMI_METRIC_SATD_X264 && n == 16) return launch_staged<4, M_SATD_X264, 16>(d_a, d_b, d_out, count, st);
    }
    switch (metric) {
    case XMI_METRIC_SAD: return n == 4 ? launch<4, M_SAD>(d_a, d_b, d_out, n, count, st) : launch<8, M_SAD>(d_a, d_b, d_out, n, count, st);
    case XMI_METRIC_SSD: return n == 4 ? launch<4, M_SSD>(d_a, d_b, d_out, n, count, st) : launch<8, M_SSD>(d_a, d_b, d_out, n, count, st);
    case XMI_METRIC_SATD_HM: return n == 4 ? launch<4, M_SATD_HM>(d_a, d_b, d_out, n, count, st) : launch<8, M_SATD_HM>(d_a, d_b, d_out, n, count, st);
    case XMI_METRIC_SATD_X264:
        XMI_ARG(n <= 16, "xmi_block_metric_batch_dev: the x264 SATD exists for 4, 8 and 16 only");
        return launch<4, M_SATD_X264>(d_a, d_b, d_out, n, count, st);
    default: xmi::set_error("xmi_block_metric_batch_dev: unknown metric %d", metric); return XMI_ERR_ARG;
    }
}
