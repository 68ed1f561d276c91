// Batched transform-unit chain for gfx950: residual -> forward DCT/DST -> quant -> dequant -> inverse -> recon.
//
// Restates, per TU, the reference's xSubDctQuantRec (src/encode.cpp:3005-3021) and its parts xSubDct
// (src/pixel.cpp:3446, C spec :1415/:1567/:1826/:2300/:3149), xQuant (:1060), xDeQuant (:1249), xIDctAdd (:5938,
// C spec :3584/:3725/:4071/:4629/:5573). All arithmetic is integer and bit-exact: 32-bit accumulation of
// int16 x int16 products, the reference's shifts and rounding, int16 clipping after each inverse stage.
//
// Every 1-D stage has the same shape  out[a][b] = f((sum_c Mat[a][c] * in[b][c] + rnd) >> shift)  when
// the forward stages use the core matrix M and the inverse stages its transpose (the buffers are kept
// transposed between stages exactly like the reference's temporaries), so one routine serves all four.
// The inner product runs on v_dot2c_i32_i16 over LDS-resident int16 rows with an odd dword pitch.
#include "tu_valu_device.h"

#include <atomic>
#include <type_traits>

namespace {

__constant__ int8_t k_dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };

enum { MODE_CHAIN = 0, MODE_SUBDCT = 1, MODE_QUANT = 2, MODE_DEQUANT = 3, MODE_IDCT = 4 };

struct TuParams {
    const uint8_t *pix, *pred, *qp, *flags;
    const int16_t *coef_in;   // QUANT: coefficients, DEQUANT: levels, IDCT: coefficients
    int16_t *out16;           // CHAIN/QUANT: levels, SUBDCT/DEQUANT: coefficients
    uint8_t *rec;
    uint32_t *sum;
    int count;
};

template <int N, int MODE>
__global__ void __launch_bounds__(256) tu_kernel(const TuParams p)
{
    constexpr int LD = Cfg<N>::LD, TPT = Cfg<N>::TPT, TUS = Cfg<N>::TUS, LG = Cfg<N>::LOG2;
    __shared__ __align__(16) int16_t s_mat[2][2][N * LD];   // [dct|dst][M|Mt]
    __shared__ __align__(16) int16_t s_a[TUS][N * LD];
    __shared__ __align__(16) int16_t s_b[TUS][N * LD];
    __shared__ uint32_t s_sum[TUS];

    const int tid = threadIdx.x;
    const int slot = tid / TPT, t = tid % TPT;
    const int tu = blockIdx.x * TUS + slot;
    const bool live = tu < p.count;

    // core matrices (and the 4x4 DST) and their transposes
    for (int e = tid; e < N * N; e += 256) {
        const int k = e / N, j = e % N;
        const int16_t c = (int16_t)dct_coef(N, k, j);
        s_mat[0][0][k * LD + j] = c;
        s_mat[0][1][j * LD + k] = c;
        if (N == 4) {
            const int16_t d = k_dst4[k * 4 + j];
            s_mat[1][0][k * LD + j] = d;
            s_mat[1][1][j * LD + k] = d;
        }
    }
    if (tid < TUS) s_sum[tid] = 0;

    const int flags = live && p.flags ? p.flags[tu] : 0;
    const int qp = live && p.qp ? p.qp[tu] : 0;
    const int use_dst = (N == 4) && (flags & 2);
    const int16_t *M = s_mat[use_dst][0], *Mt = s_mat[use_dst][1];
    int16_t *A = s_a[slot], *B = s_b[slot];
    const size_t base = (size_t)tu * N * N;

    if (MODE == MODE_CHAIN || MODE == MODE_SUBDCT) {
        if (live)
            for (int e = t; e < N * N; e += TPT) A[(e / N) * LD + e % N] = (int16_t)((int)p.pix[base + e] - (int)p.pred[base + e]);
        __syncthreads();
        if (live) stage<N, false>(M, A, B, LG - 1, t);   // rows of the residual -> columns of the temporary
        __syncthreads();
        if (live) stage<N, false>(M, B, A, LG + 6, t);   // A[k][i] = coefficient (row k = vertical frequency)
        __syncthreads();
        if (MODE == MODE_SUBDCT) {
            if (live)
                for (int e = t; e < N * N; e += TPT) p.out16[base + e] = A[(e / N) * LD + e % N];
            return;
        }
    }
    if (MODE == MODE_QUANT || MODE == MODE_DEQUANT || MODE == MODE_IDCT) {
        if (live)
            for (int e = t; e < N * N; e += TPT) A[(e / N) * LD + e % N] = p.coef_in[base + e];
        __syncthreads();
    }
    if (MODE == MODE_CHAIN || MODE == MODE_QUANT || MODE == MODE_DEQUANT) {
        // xQuant: level = (|c| * Q + rnd) >> qbits, sign restored, clipped to int16; sum of levels before sign/clip.
        // xDeQuant: (level * (IQ << qp/6) + (1 << (shift-1))) >> shift, clipped to int16.
        const int qbits = 14 + qp / 6 + (15 - 8 - LG);
        const int q = k_quant_scales[qp % 6];
        const int qrnd = ((flags & 1) ? 171 : 85) << (qbits - 9);
        const int dshift = 20 - 14 - (15 - 8 - LG);
        const int dscale = k_inv_quant_scales[qp % 6] << (qp / 6);
        uint32_t local = 0;
        if (live)
            for (int e = t; e < N * N; e += TPT) {
                const int k = e / N, i = e % N;
                int level = A[k * LD + i];
                if (MODE != MODE_DEQUANT) {
                    const int c = level;
                    const int mag = (abs(c) * q + qrnd) >> qbits;
                    local += (uint32_t)mag;
                    level = clip16(c < 0 ? -mag : mag);
                    p.out16[base + e] = (int16_t)level;
                }
                if (MODE != MODE_QUANT) {
                    const int d = clip16((level * dscale + (1 << (dshift - 1))) >> dshift);
                    if (MODE == MODE_DEQUANT) p.out16[base + e] = (int16_t)d;
                    else B[i * LD + k] = (int16_t)d;   // transposed for the inverse stages
                }
            }
        if (MODE != MODE_DEQUANT) {
            if (live && local) atomicAdd(&s_sum[slot], local);
            __syncthreads();
            if (live && t == 0) p.sum[tu] = s_sum[slot];
            if (MODE == MODE_QUANT) return;
        } else {
            return;
        }
        __syncthreads();
    }
    if (MODE == MODE_IDCT) {
        // transpose the coefficient block into B for the inverse stages
        if (live)
            for (int e = t; e < N * N; e += TPT) B[(e % N) * LD + e / N] = A[(e / N) * LD + e % N];
        __syncthreads();
    }
    // inverse: A[j][i] = clip16(sum_k M[k][j] * coef[k][i]) ; B[j][i] = clip16(sum_k M[k][j] * A'[k][i]) = residual[i][j]
    if (live) stage<N, true>(Mt, B, A, 7, t);
    __syncthreads();
    if (live) stage<N, true>(Mt, A, B, 12, t);
    __syncthreads();
    if (live)
        for (int e = t; e < N * N; e += TPT) {
            const int i = e / N, j = e % N;
            const int v = (int)B[j * LD + i] + (int)p.pred[base + e];
            p.rec[base + e] = (uint8_t)min(255, max(0, v));
        }
}

// ---------------------------------------------------------------------------------------------------
// 8x8 / 16x16 / 32x32 chain on the matrix cores: one wavefront per 32x32 tile, the whole chain register to register.
//
// v_mfma_i32_32x32x32_i8 computes D[m][n] = sum_k A[m][k] * B[k][n] (+C) with, for lane l (r = l & 31, h = l >> 5):
//   A fragment byte j = A[r][16h + j],  B fragment byte j = B[16h + j][r],  D register g = D[rho(h,g)][r],
//   rho(h,g) = (g & 3) + 8 (g >> 2) + 4h          (checked with exact integer data, tools/microbench/mfma_i8_probe.hip).
// A result tile therefore has its column on the lane and its rows in the registers, so it can feed the
// next product directly as long as that product contracts over the tile's ROW index; the contraction
// order inside the instruction is then rho(h, .) instead of 16h + ., which is absorbed by permuting the
// constant (matrix) operand the same way. Chain (R = residual, M = core matrix):
//   F1  D1 = R  * M^T      A = residual rows (from memory),          B = M (natural)      -> D1[i][k] = tmp[k][i]
//   F2  D2 = M  * D1       A = M with columns permuted by rho,       B = D1 (registers)   -> D2[k][i] = coef[k][i]
//   I1  D3 = D2^T * M      A = D2 (registers, k permuted by rho),    B = M rows permuted  -> D3[i][j] = t0[i][j]
//   I2  D4 = M^T * D3      A = M rows permuted (same fragment),      B = D3 (registers)   -> D4[j][i] = t1[i][j]
// 16-bit operands are fed as two int8 slices, t = 256 * hi + (lo' + 128) with hi = t >> 8 (signed) and
// lo' = (t & 255) - 128, so sum M t = 256 sum M hi + sum M lo' + 128 sum M: two MFMAs and a constant that is
// folded into the accumulator input together with the rounding offset. The residual of F1 is
// (src - 128) - (pred - 128): two MFMAs against M and -M. Everything is exact 32-bit integer arithmetic.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int rho(int h, int g) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// pack bits 15:8 (signed high byte) and bits 7:0 minus 128 of 16 ints into two 16-byte MFMA fragments.
// SAT: the ints are first clipped to int16 (v_cvt_pk_i16_i32 packs two of them with signed saturation).
template <bool SAT>
__device__ __forceinline__ void slice16(const int (&t)[16], v4i &hi, v4i &lo)
{
    typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned p01, p23;
        if (SAT) {
            p01 = __builtin_bit_cast(unsigned, (s2)__builtin_amdgcn_cvt_pk_i16(t[4 * w + 0], t[4 * w + 1]));
            p23 = __builtin_bit_cast(unsigned, (s2)__builtin_amdgcn_cvt_pk_i16(t[4 * w + 2], t[4 * w + 3]));
        } else {
            p01 = __builtin_amdgcn_perm((unsigned)t[4 * w + 1], (unsigned)t[4 * w + 0], 0x05040100u);
            p23 = __builtin_amdgcn_perm((unsigned)t[4 * w + 3], (unsigned)t[4 * w + 2], 0x05040100u);
        }
        hi[w] = (int)__builtin_amdgcn_perm(p23, p01, 0x07050301u);
        lo[w] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
    }
}

// sum over aligned groups of W consecutive lanes (W = 2 .. 64); the total lands in the last lane of each group
template <int W>
__device__ __forceinline__ int group_sum_to_last_lane(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);                 // row_shr:1
    if (W >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    if (W >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    if (W >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    if (W >= 32) v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);    // row_bcast:15 into rows 1 and 3
    if (W >= 64) v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);    // row_bcast:31 into rows 2 and 3
    return v;
}

// xQuant / xDeQuant on their own are streaming kernels: a wavefront takes 1024 coefficients (whole TUs) per step,
// 8 per lane and 16-byte access, fully coalesced; the per-TU sum of |level| is a segmented DPP reduction.
typedef short v8s __attribute__((ext_vector_type(8)));

template <int N, bool DEQUANT>
__global__ void __launch_bounds__(256) quant_stream_kernel(const TuParams p)
{
    constexpr int LG = Cfg<N>::LOG2, NN = N * N;
    constexpr int LPT = NN / 8 < 64 ? NN / 8 : 64;       // lanes per TU within one 512-coefficient half step
    __shared__ int2 s_qtab[52];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 52) s_qtab[tid] = make_int2(k_quant_scales[tid % 6] | ((14 + tid / 6 + (15 - 8 - LG)) << 16), k_inv_quant_scales[tid % 6] << (tid / 6));
    __syncthreads();
    const size_t total = (size_t)p.count * NN;            // coefficients, a multiple of 16
    const size_t nsteps = (total + 1023) / 1024;
    for (size_t step = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave); step < nsteps; step += (size_t)gridDim.x * 4) {
        int part[2] = { 0, 0 };
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const size_t e0 = step * 1024 + half * 512 + lane * 8;
            if (e0 < total) {
                const int tu = (int)(e0 / NN);
                const v8s c = *reinterpret_cast<const v8s *>(p.coef_in + e0);
                const int2 e = s_qtab[min((int)p.qp[tu], 51)];
                v8s o;
                if (DEQUANT) {
                    const int dscale = e.y;
#pragma unroll
                    for (int k = 0; k < 8; k++) o[k] = (short)clip16((__mul24((int)c[k], dscale) + (1 << (LG - 2))) >> (LG - 1));
                } else {
                    const int q = e.x & 0xffff, qbits = e.x >> 16;
                    const int qrnd = (85 + 86 * (p.flags ? (int)(p.flags[tu] & 1) : 0)) << (qbits - 9);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int v = c[k];
                        const int mag = (int)((__umul24((unsigned)abs(v), (unsigned)q) + (unsigned)qrnd) >> qbits);
                        part[half] += mag;
                        o[k] = (short)clip16(v < 0 ? -mag : mag);
                    }
                }
                *reinterpret_cast<v8s *>(p.out16 + e0) = o;
            }
        }
        if (!DEQUANT) {
            if (N == 32) {          // both halves belong to one TU
                const int v = group_sum_to_last_lane<64>(part[0] + part[1]);
                const size_t tu = step;
                if (lane == 63 && tu < (size_t)p.count) p.sum[tu] = (uint32_t)v;
            } else {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int v = group_sum_to_last_lane<LPT>(part[half]);
                    const size_t tu = (step * 1024 + half * 512 + lane * 8) / NN;
                    if ((lane & (LPT - 1)) == LPT - 1 && tu < (size_t)p.count) p.sum[tu] = (uint32_t)v;
                }
            }
        }
    }
}

template <int N, bool DEQUANT>
int launch_quant_stream(const TuParams &p, hipStream_t st)
{
    const size_t nsteps = ((size_t)p.count * N * N + 1023) / 1024;
    const int blocks = (int)((nsteps + 3) / 4 < 8192 ? (nsteps + 3) / 4 : 8192);
    hipLaunchKernelGGL((quant_stream_kernel<N, DEQUANT>), dim3(blocks), dim3(256), 0, st, p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

template <bool DEQUANT>
int launch_quant_stream_n(int n, const TuParams &p, hipStream_t st)
{
    switch (n) {
    case 4: return launch_quant_stream<4, DEQUANT>(p, st);
    case 8: return launch_quant_stream<8, DEQUANT>(p, st);
    case 16: return launch_quant_stream<16, DEQUANT>(p, st);
    default: return launch_quant_stream<32, DEQUANT>(p, st);
    }
}

// N = 32: one TU per 32x32 tile. N = 16 / 8: a tile holds (32/N)^2 independent TUs side by side and the
// constant operand is the block-diagonal matrix diag(M_N, ..): block (a,b) of M R M^T is then M_N R_ab M_N^T,
// so the same four products transform every TU of the tile at once (TU t of tile s sits at block row a, block
// column b with t = s*S + b*(32/N) + a, so the TUs a lane works on are consecutive).
// A lane holds column r of a result tile and the 16 rows rho(h, 0..15): four groups of four consecutive rows
// (group w = rows 8w+4h .. 8w+4h+3), so it sees G = min(4, 32/N) TUs of its column block; slot(w) names them.
// MODE_CHAIN: the whole chain. MODE_SUBDCT: F1, F2 and the coefficients are stored (xSubDct). MODE_IDCT: the
// coefficients are read, I1, I2 and the reconstruction follow (xIDctAdd).
template <int N, int MODE>
__global__ void __launch_bounds__(256) tu_mfma_kernel(const TuParams p)
{
    constexpr int NB = 32 / N;            // TUs per tile row / column
    constexpr int S = NB * NB;            // TUs per tile
    constexpr int LG = N == 32 ? 5 : N == 16 ? 4 : 3;
    constexpr int G = N == 32 ? 1 : N == 16 ? 2 : 4;
    auto slot = [](int w) { return N == 32 ? 0 : N == 16 ? (w >> 1) : w; };
    __shared__ __align__(16) int s_colsum[32];
    __shared__ int8_t s_m[32 * 33];       // block-diagonal core matrix, built once per workgroup
    __shared__ int2 s_qtab[52];           // per qp: { quant scale | qbits << 16, dequant scale << qp/6 }
    // per wavefront: the 1024 levels / coefficients of a tile in memory order, used where a row of a TU is too short
    // for good stores (8x8) and to turn the row-major coefficient input of MODE_IDCT into columns. TU column b is
    // skewed by 2bN bytes so that the TU columns one LDS instruction touches fall into different banks.
    constexpr bool LVL_LDS = N == 8 && MODE != MODE_IDCT;
    constexpr bool PATCH = LVL_LDS || MODE == MODE_IDCT;
    __shared__ __align__(16) int16_t s_lvl[PATCH ? 4 : 1][PATCH ? 1024 + 64 : 8];
    auto chunk_skew = [](int c) { return N < 32 ? ((c * 8) / (NB * N * N)) * N : 0; };   // chunk c = 8 values in memory order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int e = tid; e < 1024; e += 256) {
        const int k = e >> 5, j = e & 31;
        s_m[k * 33 + j] = (k / N == j / N) ? (int8_t)dct_coef(N, k % N, j % N) : (int8_t)0;
    }
    if (tid < 52) s_qtab[tid] = make_int2(k_quant_scales[tid % 6] | ((14 + tid / 6 + (15 - 8 - LG)) << 16), k_inv_quant_scales[tid % 6] << (tid / 6));
    __syncthreads();
    if (tid < 32) {
        int sum = 0;
        for (int k = 0; k < 32; k++) sum += s_m[k * 33 + tid];
        s_colsum[tid] = sum;
    }
    __syncthreads();

    // constant fragments (built once per wavefront; the wavefront then loops over its tiles)
    v4i m_nat, m_neg, m_colperm, m_rowperm;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned a = 0, b = 0, c = 0, d = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 4 * w + e;
            const int nat = s_m[r * 33 + 16 * h + j];
            a |= (unsigned)(nat & 255) << (8 * e);
            b |= (unsigned)((-nat) & 255) << (8 * e);
            c |= (unsigned)(s_m[r * 33 + rho(h, j)] & 255) << (8 * e);
            d |= (unsigned)(s_m[rho(h, j) * 33 + r] & 255) << (8 * e);
        }
        m_nat[w] = (int)a;
        m_neg[w] = (int)b;
        m_colperm[w] = (int)c;
        m_rowperm[w] = (int)d;
    }
    const int cs_lane = (1 << 6) + 128 * s_colsum[r];
    __syncthreads();
    if (tid < 32) s_colsum[tid] = (1 << 11) + 128 * s_colsum[tid];      // accumulator input of the last product, per output row
    __syncthreads();

    // byte offset of pixel (row, col) of a tile inside the packed TU arrays, relative to the tile's first TU
    auto pix_off = [](int row, int col) { return ((col / N) * NB + row / N) * (N * N) + (row % N) * N + col % N; };
    const int ntiles = p.count / S;       // whole tiles only: the host sends the remainder of a batch to tu_kernel
    const int cb = r / N;                 // this lane's column block in every result tile
    int16_t *lv = s_lvl[PATCH ? wave : 0];
    const int lv_skew_w = N < 32 ? cb * N : 0;
    const int tu_rel0 = cb * NB;          // tile-relative index of the first TU of the lane's column block (slot s: + s)
    // software pipeline: the operands of the next tile are requested before the current one is computed.
    // Nothing is derived from them until the next iteration, so no wait separates the requests from the arithmetic.
    const int t0 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave), t_step = gridDim.x * 4;   // wavefront-uniform
    v4i s_next = { 0, 0, 0, 0 }, p_next = { 0, 0, 0, 0 };
    v4i c_next[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };   // MODE_IDCT: chunks lane and lane + 64 of the coefficient tile
    unsigned qp_next = 0, fl_next = 0;    // G consecutive bytes each
    auto fetch = [&](int tile) {
        const int tu_first = tile * S;
        const size_t b = (size_t)tu_first * (N * N);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned off = (unsigned)pix_off(r, 16 * h + 4 * w);
            if (MODE != MODE_IDCT) s_next[w] = *reinterpret_cast<const int *>(p.pix + b + off);
            p_next[w] = *reinterpret_cast<const int *>(p.pred + b + off);
        }
        if (MODE == MODE_IDCT) {
            c_next[0] = *reinterpret_cast<const v4i *>(p.coef_in + b + lane * 8);
            c_next[1] = *reinterpret_cast<const v4i *>(p.coef_in + b + 512 + lane * 8);
        }
        if (MODE != MODE_CHAIN) {
        } else if (G == 1) {
            qp_next = p.qp[tu_first + tu_rel0];
            fl_next = p.flags[tu_first + tu_rel0];
        } else if (G == 2) {
            qp_next = *reinterpret_cast<const uint16_t *>(p.qp + tu_first + tu_rel0);
            fl_next = *reinterpret_cast<const uint16_t *>(p.flags + tu_first + tu_rel0);
        } else {
            qp_next = *reinterpret_cast<const uint32_t *>(p.qp + tu_first + tu_rel0);
            fl_next = *reinterpret_cast<const uint32_t *>(p.flags + tu_first + tu_rel0);
        }
    };
    auto claim = [&]() {
#pragma unroll
        for (int w = 0; w < 4; w++) asm volatile("" : "+v"(s_next[w]), "+v"(p_next[w]), "+v"(c_next[0][w]), "+v"(c_next[1][w]));
        asm volatile("" : "+v"(qp_next), "+v"(fl_next));
    };
    if (t0 < ntiles) fetch(t0);
    claim();
    for (int tile = t0; tile < ntiles; tile += t_step) {
        const int tu_first = tile * S;
        const size_t base = (size_t)tu_first * (N * N);
        v4i s_frag = s_next, p_frag = p_next;
        const unsigned qp_cur = qp_next, fl_cur = fl_next;
        const v4i p_raw = p_next;
        const v4i c_cur0 = c_next[0], c_cur1 = c_next[1];
        if (tile + t_step < ntiles) fetch(tile + t_step);
        int t[16];
        v4i hi, lo;
        v16i acc_hi, acc_lo;
        int zero = 0, bias1 = cs_lane;
        // Accumulator inputs are rebuilt per tile from a register the compiler cannot see through: kept as loop
        // invariants they would pin 32 VGPRs and cost a wavefront of occupancy.
        asm volatile("" : "+v"(zero), "+v"(bias1));

        if (MODE == MODE_IDCT) {
            // coefficient tile: memory order -> LDS patch -> column r, rows rho(h, g)
            *reinterpret_cast<v4i *>(lv + lane * 8 + chunk_skew(lane)) = c_cur0;
            *reinterpret_cast<v4i *>(lv + (lane + 64) * 8 + chunk_skew(lane + 64)) = c_cur1;
#pragma unroll
            for (int g = 0; g < 16; g++) t[g] = (int)lv[pix_off(rho(h, g), r) + lv_skew_w];
        } else {

        // ---- F1: residual rows x M^T ----
#pragma unroll
        for (int w = 0; w < 4; w++) {
            s_frag[w] ^= (int)0x80808080u;
            p_frag[w] ^= (int)0x80808080u;
        }
        v16i acc;
#pragma unroll
        for (int g = 0; g < 16; g++) acc[g] = 1 << (LG - 2);   // rounding of the >> (log2 N - 1)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_frag, m_nat, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(p_frag, m_neg, acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; g++) t[g] = acc[g] >> (LG - 1);          // tmp[k = r][i = rho(h,g)], fits int16

        // ---- F2: M x D1 ----
        slice16<false>(t, hi, lo);
#pragma unroll
        for (int g = 0; g < 16; g++) {
            acc_hi[g] = 0;
            // rounding of the >> (log2 N + 6) + 128 * rowsum(M[k]) (only the DC row of each block has a non-zero sum)
            acc_lo[g] = zero + ((1 << (LG + 5)) + (rho(h, g) % N == 0 ? 128 * 64 * N : 0));
        }
        acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_colperm, hi, acc_hi, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_colperm, lo, acc_lo, 0, 0, 0);

        // ---- quant / dequant on coef[k = rho(h,g)][i = r]; the TU (and its qp) depends on the row group of g ----
        // xQuant: level = sign(c) * ((|c| * Q + rnd) >> qbits). With floor division by 2^qbits this is
        // (c * Q + rnd) >> qbits for c >= 0 and (c * Q + (2^qbits - 1 - rnd)) >> qbits for c < 0.
        int q = 0, qbits = 0, qadd_pos = 0, qadd_neg = 0, dscale = 0, local = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int sl = slot(w);
            if (MODE == MODE_CHAIN && (w == 0 || slot(w) != slot(w - 1))) {      // parameters of the TU of this row group
                const int2 e = s_qtab[min((qp_cur >> (8 * sl)) & 255u, 51u)];
                q = e.x & 0xffff;
                qbits = e.x >> 16;
                dscale = e.y;
                qadd_pos = (85 + 86 * (int)((fl_cur >> (8 * sl)) & 1u)) << (qbits - 9);   // 171 for I slices, 85 otherwise
                qadd_neg = (1 << qbits) - 1 - qadd_pos;
                local = 0;
            }
            int lvl[4];
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++) {
                const int g = 4 * w + e4;
                const int c = ((acc_hi[g] << 8) + acc_lo[g]) >> (LG + 6);            // |c| < 2^15 for 8-bit video
                if (MODE == MODE_SUBDCT) {
                    lvl[e4] = c;
                    continue;
                }
                const int level = (__mul24(c, q) + (c < 0 ? qadd_neg : qadd_pos)) >> qbits;   // |level| < 2^14: no clipping
                local += abs(level);
                lvl[e4] = level;
                t[g] = (__mul24(level, dscale) + (1 << (LG - 2))) >> (LG - 1);   // xDeQuant; clipped to int16 by the slicing
            }
            if (!LVL_LDS) {
            // Level rows 8w + 4h + e of column r: trading lane halves between registers e and e + 1 leaves rows
            // 8w + e (+ 4) in lanes 0..31 and rows 8w + e + 1 (+ 4) in lanes 32..63, so that one store instruction
            // writes two adjacent 64-byte rows = one full 128-byte line.
#pragma unroll
            for (int e4 = 0; e4 < 4; e4 += 2) {
                const auto x = __builtin_amdgcn_permlane32_swap((unsigned)lvl[e4], (unsigned)lvl[e4 + 1], false, false);
                p.out16[base + pix_off(8 * w + e4 + h, r)] = (int16_t)x[0];
                p.out16[base + pix_off(8 * w + 4 + e4 + h, r)] = (int16_t)x[1];
            }
            } else {
            // 8x8 rows are only 16 bytes: the levels go through the wavefront's LDS patch (memory order) and leave
            // as 16-byte stores, 1 KB per instruction
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++) lv[pix_off(8 * w + 4 * h + e4, r) + lv_skew_w] = (int16_t)lvl[e4];
            }
            if (MODE == MODE_CHAIN && (w == 3 || slot(w + 1) != slot(w))) {      // sum of |level| of this TU
                int v = local + __shfl_xor(local, 32);                          // both lane halves
                v = group_sum_to_last_lane<N>(v);                               // the N lanes of the column block
                if (r % N == N - 1 && h == 0) p.sum[tu_first + tu_rel0 + sl] = (uint32_t)v;
            }
        }

        if (MODE == MODE_SUBDCT) claim();
        if (LVL_LDS) {
#pragma unroll
        for (int c = lane; c < 128; c += 64) {           // chunk c = 8 consecutive levels in memory order
            const v4i d = *reinterpret_cast<const v4i *>(lv + c * 8 + chunk_skew(c));
            *reinterpret_cast<v4i *>(p.out16 + base + c * 8) = d;
        }
        }
        if (MODE == MODE_SUBDCT) continue;
        }   // MODE != MODE_IDCT

        // ---- I1: D2^T x M ----
        slice16<true>(t, hi, lo);
#pragma unroll
        for (int g = 0; g < 16; g++) {
            acc_hi[g] = 0;
            acc_lo[g] = bias1;
        }
        acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, m_rowperm, acc_hi, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, m_rowperm, acc_lo, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; g++) t[g] = ((acc_hi[g] << 8) + acc_lo[g]) >> 7;   // t0[i = rho(h,g)][j = r], clipped by the slicing

        // ---- I2: M^T x D3 ----
        slice16<true>(t, hi, lo);
#pragma unroll
        for (int g = 0; g < 16; g++) acc_hi[g] = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {       // rows rho(h, 4w .. 4w+3) = 8w + 4h .. + 3
            const int4 c4 = *reinterpret_cast<const int4 *>(&s_colsum[8 * w + 4 * h]);
            acc_lo[4 * w + 0] = c4.x;
            acc_lo[4 * w + 1] = c4.y;
            acc_lo[4 * w + 2] = c4.z;
            acc_lo[4 * w + 3] = c4.w;
        }
        acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_rowperm, hi, acc_hi, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_rowperm, lo, acc_lo, 0, 0, 0);

        // The prefetched registers are claimed here, before the last stores of the iteration are issued: the wait
        // the compiler places in front of this statement then covers the loads (requested most of an iteration
        // ago) but none of the stores that follow, and nothing is pending when the next iteration starts.
        claim();

        // ---- reconstruct: residual t1[i = r][j = rho(h,g)]; 4 groups of 4 consecutive pixels per lane. The int16
        // clip of the residual is dropped: a residual beyond +-32767 saturates the 8-bit sum either way. ----
        // prediction bytes of the lane's output pixels (row r, columns 8 qd + 4h .. + 3): half of them sit in the
        // partner lane (r, 1 - h) of the A fragment; v_permlane32_swap trades lanes 32..63 of its first operand
        // against lanes 0..31 of its second
        unsigned pv_cur[4];
        {
            const auto x = __builtin_amdgcn_permlane32_swap((unsigned)p_raw[0], (unsigned)p_raw[1], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap((unsigned)p_raw[2], (unsigned)p_raw[3], false, false);
            pv_cur[0] = x[0];
            pv_cur[2] = x[1];
            pv_cur[1] = y[0];
            pv_cur[3] = y[1];
        }
        unsigned out[4];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const unsigned pv = pv_cur[qd];
            out[qd] = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int g = 4 * qd + e;
                const int res = ((acc_hi[g] << 8) + acc_lo[g]) >> 12;
                const int v = min(255, max(0, res + (int)((pv >> (8 * e)) & 255)));
                out[qd] |= (unsigned)v << (8 * e);
            }
        }
        {   // back to the layout of the loads (row r, columns 16h .. 16h + 15): one 16-byte store per lane
            const auto x = __builtin_amdgcn_permlane32_swap(out[0], out[2], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(out[1], out[3], false, false);
            v4i o = { (int)x[0], (int)x[1], (int)y[0], (int)y[1] };
            if (N >= 16) {
                *reinterpret_cast<v4i *>(p.rec + base + pix_off(r, 16 * h)) = o;
            } else {   // 8x8: the 16 bytes span two TUs
                *reinterpret_cast<int2 *>(p.rec + base + pix_off(r, 16 * h)) = make_int2(o[0], o[1]);
                *reinterpret_cast<int2 *>(p.rec + base + pix_off(r, 16 * h + 8)) = make_int2(o[2], o[3]);
            }
        }
    }
}

// 4x4 chain on the matrix cores: a 32x32 tile holds 64 TUs (8x8 grid, TU t of tile s at block row a, block
// column b with t = 64 s + 8 b + a) and the constant operand is diag(M4, ..) with M4 the DCT or, for TUs with
// flag bit 1, the DST. A tile whose 64 TUs agree on the transform runs the chain once with that matrix; a mixed
// tile runs it twice and each pass stores only its own TUs. Everything that depends on the matrix (operand
// fragments, accumulator inputs) is read from LDS per pass, so nothing of it is pinned in registers.
// Layouts along the chain (lane = (r, h), register group w = rows 8w + 4h .. + 3 of a result tile):
//   loads / reconstruction:  lane row r, columns 16h + 4w .. + 3      -> TU (a = r/4, b = 4h + w)
//   after F1, F2 (quant):     rows 8w + 4h + e, column r                -> TU (a = 2w + h, b = r/4)
//   after I1, I2:             lane row r, columns 8w + 4h + e           -> TU (a = r/4, b = 2w + h)
template <int MODE>
__global__ void __launch_bounds__(256) tu_mfma4_kernel(const TuParams p)
{
    constexpr int S = 64, LG = 2;
    __shared__ __align__(16) v4i s_frag[2][4][64];     // [dct | dst][natural, negated, columns permuted, rows permuted][lane]
    __shared__ __align__(16) int s_bias[2][12];        // [dct | dst]: F2 input by row % 4, I2 input by row % 4, I1 input by column % 4
    __shared__ int2 s_qtab[52];                        // per qp: { quant scale | qbits << 16, dequant scale << qp/6 }
    // per wavefront: levels of a tile, TU (a, b) at lv_at(a, b). A store instruction of the quantiser writes one level row of
    // the eight TUs (a, 0 .. 7) per lane half: 16 int16 apart they would meet in pairs on a bank (stores: dword address mod 32),
    // so TUs 4 .. 7 of a row sit four dwords further on; a row takes 72 dwords, which spreads the 16-byte reads of the way out.
    __shared__ __align__(16) int16_t s_lvl[4][8 * 144];
    __shared__ uint32_t s_sum[4][64];                  // per wavefront: sum of |level| of the 64 TUs, memory order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // 4-point DCT / DST-VII matrices (src/pixel.cpp:1415, 1567) with each row packed into a word, so that building the
    // operand fragments needs no table loads
    auto mat4 = [](int type, int k, int j) {
        const unsigned row = type ? (k == 0 ? 0x544a371du : k == 1 ? 0xb6004a4au : k == 2 ? 0x37b6e354u : 0xe34aac37u)
                                  : (k == 0 ? 0x40404040u : k == 1 ? 0xaddc2453u : k == 2 ? 0x40c0c040u : 0xdc53ad24u);
        return (int)(int8_t)(row >> (8 * j));
    };
    auto mat = [&](int type, int k, int j) { return (k >> 2) == (j >> 2) ? mat4(type, k & 3, j & 3) : 0; };
    for (int e = tid; e < 512; e += 256) {
        const int type = e >> 8, kind = (e >> 6) & 3, rr = e & 31, hh = (e >> 5) & 1;
        v4i f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned a = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int j = 4 * w + b;
                const int v = kind == 0 ? mat(type, rr, 16 * hh + j) : kind == 1 ? -mat(type, rr, 16 * hh + j)
                            : kind == 2 ? mat(type, rr, rho(hh, j)) : mat(type, rho(hh, j), rr);
                a |= (unsigned)(v & 255) << (8 * b);
            }
            f[w] = (int)a;
        }
        s_frag[type][kind][e & 63] = f;
    }
    if (tid < 24) {
        const int type = tid / 12, i = tid % 12, c = i & 3;
        int rowsum = 0, colsum = 0;
        for (int k = 0; k < 4; k++) {
            rowsum += mat4(type, c, k);
            colsum += mat4(type, k, c);
        }
        s_bias[type][i] = i < 4 ? (1 << (LG + 5)) + 128 * rowsum : i < 8 ? (1 << 11) + 128 * colsum : (1 << 6) + 128 * colsum;
    }
    if (tid < 52) s_qtab[tid] = make_int2(k_quant_scales[tid % 6] | ((14 + tid / 6 + (15 - 8 - LG)) << 16), k_inv_quant_scales[tid % 6] << (tid / 6));
    __syncthreads();

    auto pix_off = [](int row, int col) { return ((col >> 2) * 8 + (row >> 2)) * 16 + (row & 3) * 4 + (col & 3); };
    auto lv_at = [](int a, int b) { return a * 144 + b * 16 + (b >> 2) * 8; };
    const int ntiles = p.count / S;
    int16_t *lv = s_lvl[wave];
    uint32_t *sm = s_sum[wave];
    const int t0 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave), t_step = gridDim.x * 4;
    v4i s_next = { 0, 0, 0, 0 }, p_next = { 0, 0, 0, 0 };
    v4i c_next[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }; // MODE_IDCT: chunks lane and lane + 64 of the coefficient tile
    uint2 qp_next = { 0, 0 }, fl_next = { 0, 0 };       // the 8 TUs of block column b = r/4, byte a
    auto fetch = [&](int tile) {
        const int tu_first = tile * S;
        const size_t b = (size_t)tu_first * 16;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned off = (unsigned)pix_off(r, 16 * h + 4 * w);
            if (MODE != MODE_IDCT) s_next[w] = *reinterpret_cast<const int *>(p.pix + b + off);
            p_next[w] = *reinterpret_cast<const int *>(p.pred + b + off);
        }
        if (MODE == MODE_IDCT) {
            c_next[0] = *reinterpret_cast<const v4i *>(p.coef_in + b + lane * 8);
            c_next[1] = *reinterpret_cast<const v4i *>(p.coef_in + b + 512 + lane * 8);
        }
        if (MODE == MODE_CHAIN) qp_next = *reinterpret_cast<const uint2 *>(p.qp + tu_first + (r >> 2) * 8);
        if (p.flags) fl_next = *reinterpret_cast<const uint2 *>(p.flags + tu_first + (r >> 2) * 8);
    };
    auto claim = [&]() {
#pragma unroll
        for (int w = 0; w < 4; w++) asm volatile("" : "+v"(s_next[w]), "+v"(p_next[w]), "+v"(c_next[0][w]), "+v"(c_next[1][w]));
        asm volatile("" : "+v"(qp_next.x), "+v"(qp_next.y), "+v"(fl_next.x), "+v"(fl_next.y));
    };
    if (t0 < ntiles) fetch(t0);
    claim();
    // The operands of a transform: the single stages keep them in registers from tile to tile until a tile asks for the other
    // transform; the chain, which has no registers to spare for that, reads them per pass.
    constexpr bool KEEP = MODE != MODE_CHAIN;
    int have_type = -1;
    v4i m_nat = { 0, 0, 0, 0 }, m_neg = { 0, 0, 0, 0 }, m_colperm = { 0, 0, 0, 0 }, m_rowperm = { 0, 0, 0, 0 };
    int4 b_f2 = { 0, 0, 0, 0 }, b_i2 = { 0, 0, 0, 0 };
    int b_i1 = 0;
    for (int tile = t0; tile < ntiles; tile += t_step) {
        const int tu_first = tile * S;
        const size_t base = (size_t)tu_first * 16;
        const v4i s_raw = s_next, p_raw = p_next;
        const uint2 qp_cur = qp_next, fl_cur = fl_next;
        const v4i c_cur0 = c_next[0], c_cur1 = c_next[1];
        if (tile + t_step < ntiles) fetch(tile + t_step);
        if (MODE == MODE_IDCT) {
            // coefficient tile: memory order (TU t = 8b + a at t * 16) -> LDS patch (TU (a, b) at (8a + b) * 16)
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int c = lane + 64 * half, tu = c >> 1;
                *reinterpret_cast<v4i *>(lv + lv_at(tu & 7, tu >> 3) + (c & 1) * 8) = half ? c_cur1 : c_cur0;
            }
        }

        // prediction bytes in the reconstruction layout (row r, columns 8 qd + 4h .. + 3), see tu_mfma_kernel
        unsigned pv_cur[4];
        {
            const auto x = __builtin_amdgcn_permlane32_swap((unsigned)p_raw[0], (unsigned)p_raw[1], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap((unsigned)p_raw[2], (unsigned)p_raw[3], false, false);
            pv_cur[0] = x[0];
            pv_cur[2] = x[1];
            pv_cur[1] = y[0];
            pv_cur[3] = y[1];
        }

        // uniform tile: one pass with its transform; mixed tile: a DCT pass and a DST pass, each storing its own TUs
        const bool dst_lane = ((fl_cur.x & fl_cur.y) & 0x02020202u) == 0x02020202u;
        const bool dct_lane = ((fl_cur.x | fl_cur.y) & 0x02020202u) == 0u;
        const bool all_dst = __all(dst_lane), all_dct = __all(dct_lane);
        auto pass = [&](auto mixed_c, const int type) {
            constexpr bool MIXED = decltype(mixed_c)::value;
            if (!KEEP || type != have_type) {
                m_nat = s_frag[type][0][lane], m_neg = s_frag[type][1][lane];
                m_colperm = s_frag[type][2][lane], m_rowperm = s_frag[type][3][lane];
                b_f2 = *reinterpret_cast<const int4 *>(&s_bias[type][0]);
                b_i2 = *reinterpret_cast<const int4 *>(&s_bias[type][4]);
                b_i1 = s_bias[type][8 + (r & 3)];
                have_type = type;
            }

            int t[16];
            v4i hi, lo;
            v16i acc_hi, acc_lo;
            if (MODE == MODE_IDCT) {
#pragma unroll
                for (int g = 0; g < 16; g++)       // coef[k = 8w + 4h + e][i = r] of TU (a = 2w + h, b = r/4)
                    t[g] = (int)lv[lv_at(2 * (g >> 2) + h, r >> 2) + (g & 3) * 4 + (r & 3)];
            } else {
            // ---- F1 ----
            v4i s_frag4, p_frag4;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                s_frag4[w] = s_raw[w] ^ (int)0x80808080u;
                p_frag4[w] = p_raw[w] ^ (int)0x80808080u;
            }
            v16i acc;
#pragma unroll
            for (int g = 0; g < 16; g++) acc[g] = 1 << (LG - 2);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_frag4, m_nat, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(p_frag4, m_neg, acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 16; g++) t[g] = acc[g] >> (LG - 1);

            // ---- F2 ----
            slice16<false>(t, hi, lo);
#pragma unroll
            for (int g = 0; g < 16; g++) {
                acc_hi[g] = 0;
                acc_lo[g] = (g & 3) == 0 ? b_f2.x : (g & 3) == 1 ? b_f2.y : (g & 3) == 2 ? b_f2.z : b_f2.w;
            }
            acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_colperm, hi, acc_hi, 0, 0, 0);
            acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_colperm, lo, acc_lo, 0, 0, 0);

            // ---- quant / dequant: register group w belongs to TU (a = 2w + h, b = r/4) ----
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned sh = 16 * (w & 1) + 8 * h;
                const unsigned qpb = ((w < 2 ? qp_cur.x : qp_cur.y) >> sh) & 255u, flb = ((w < 2 ? fl_cur.x : fl_cur.y) >> sh) & 255u;
                const bool mine = !MIXED || (int)((flb >> 1) & 1u) == type;
                if (MODE == MODE_SUBDCT) {
#pragma unroll
                    for (int e4 = 0; e4 < 4; e4++) {
                        const int g = 4 * w + e4;
                        const int c = ((acc_hi[g] << 8) + acc_lo[g]) >> (LG + 6);
                        if (mine) lv[lv_at(2 * w + h, r >> 2) + e4 * 4 + (r & 3)] = (int16_t)c;
                    }
                    continue;
                }
                const int2 e = s_qtab[min(qpb, 51u)];
                const int q = e.x & 0xffff, qbits = e.x >> 16, dscale = e.y;
                const int qadd_pos = (85 + 86 * (int)(flb & 1u)) << (qbits - 9);
                const int qadd_neg = (1 << qbits) - 1 - qadd_pos;
                int local = 0;
                const int a = 2 * w + h, b = r >> 2;
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    const int g = 4 * w + e4;
                    const int c = ((acc_hi[g] << 8) + acc_lo[g]) >> (LG + 6);
                    const int level = (__mul24(c, q) + (c < 0 ? qadd_neg : qadd_pos)) >> qbits;
                    local += abs(level);
                    if (mine) lv[lv_at(a, b) + e4 * 4 + (r & 3)] = (int16_t)level;
                    t[g] = (__mul24(level, dscale) + (1 << (LG - 2))) >> (LG - 1);
                }
                local += __builtin_amdgcn_update_dpp(0, local, 0x111, 0xf, 0xf, true);   // row_shr:1
                local += __builtin_amdgcn_update_dpp(0, local, 0x112, 0xf, 0xf, true);   // row_shr:2
                if ((r & 3) == 3 && mine) sm[b * 8 + a] = (uint32_t)local;
            }
            if (MODE == MODE_SUBDCT) return;
            }   // MODE != MODE_IDCT

            // ---- I1 ----
            slice16<true>(t, hi, lo);
#pragma unroll
            for (int g = 0; g < 16; g++) {
                acc_hi[g] = 0;
                acc_lo[g] = b_i1;
            }
            acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, m_rowperm, acc_hi, 0, 0, 0);
            acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, m_rowperm, acc_lo, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 16; g++) t[g] = ((acc_hi[g] << 8) + acc_lo[g]) >> 7;

            // ---- I2 ----
            slice16<true>(t, hi, lo);
#pragma unroll
            for (int g = 0; g < 16; g++) {
                acc_hi[g] = 0;
                acc_lo[g] = (g & 3) == 0 ? b_i2.x : (g & 3) == 1 ? b_i2.y : (g & 3) == 2 ? b_i2.z : b_i2.w;
            }
            acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_rowperm, hi, acc_hi, 0, 0, 0);
            acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(m_rowperm, lo, acc_lo, 0, 0, 0);

            // ---- reconstruct (row r, columns 8 qd + 4h + e), then back to the layout of the loads ----
            unsigned out[4];
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const unsigned pv = pv_cur[qd];
                out[qd] = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int g = 4 * qd + e;
                    const int res = ((acc_hi[g] << 8) + acc_lo[g]) >> 12;
                    const int v = min(255, max(0, res + (int)((pv >> (8 * e)) & 255)));
                    out[qd] |= (unsigned)v << (8 * e);
                }
            }
            const auto x = __builtin_amdgcn_permlane32_swap(out[0], out[2], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(out[1], out[3], false, false);
            const unsigned o[4] = { x[0], x[1], y[0], y[1] };
#pragma unroll
            for (int w = 0; w < 4; w++) {      // TU (a = r/4, b = 4h + w)
                bool mine = true;
                if (MIXED) mine = (int)((p.flags[tu_first + (4 * h + w) * 8 + (r >> 2)] >> 1) & 1) == type;   // p.flags != null when mixed
                if (mine) *reinterpret_cast<unsigned *>(p.rec + base + pix_off(r, 16 * h + 4 * w)) = o[w];
            }
        };
        if (all_dct || all_dst) {
            pass(std::false_type{}, all_dst ? 1 : 0);
        } else {
            for (int type = 0; type < 2; type++) pass(std::true_type{}, type);
        }
        claim();

        // levels: LDS patch -> memory order (TU t = 8b + a at t * 16), 16 bytes = two TU rows per lane and step
        if (MODE != MODE_IDCT) {
#pragma unroll
        for (int c = lane; c < 128; c += 64) {
            const int tu = c >> 1, a = tu & 7, b = tu >> 3;
            const v4i d = *reinterpret_cast<const v4i *>(lv + lv_at(a, b) + (c & 1) * 8);
            *reinterpret_cast<v4i *>(p.out16 + base + c * 8) = d;
        }
        }
        if (MODE == MODE_CHAIN) p.sum[tu_first + lane] = sm[lane];
    }
}

// Workgroups of a kernel that the device holds at the same time (the wavefronts loop over tiles, so a grid of exactly
// that size has no second, partly filled wave of workgroups): register use decides it, so it is asked per kernel.
template <auto Kernel>
int resident_blocks()
{
    static std::atomic<int> cached{ 0 };
    int v = cached.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kernel, 256, 0) != hipSuccess || cus <= 0 || per_cu <= 0)
            v = 2048;
        else
            v = cus * per_cu;
        cached.store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int MODE>
int launch_mfma4(const TuParams &p, hipStream_t st)
{
    constexpr int S = 64;
    const int tiles = p.count / S;
    if (tiles > 0) {
        const int cap = resident_blocks<tu_mfma4_kernel<MODE>>();
        const int blocks = (tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap;
        hipLaunchKernelGGL((tu_mfma4_kernel<MODE>), dim3(blocks), dim3(256), 0, st, p);
        XMI_HIP(hipGetLastError());
    }
    const int rest = p.count - tiles * S;
    if (rest > 0) {
        const size_t tu0 = (size_t)tiles * S, px0 = tu0 * 16;
        auto at = [](auto *ptr, size_t off) { return ptr ? ptr + off : ptr; };
        TuParams q = { at(p.pix, px0), at(p.pred, px0), at(p.qp, tu0), at(p.flags, tu0), at(p.coef_in, px0), at(p.out16, px0),
                       at(p.rec, px0), at(p.sum, tu0), rest };
        hipLaunchKernelGGL((tu_kernel<4, MODE>), dim3((rest + Cfg<4>::TUS - 1) / Cfg<4>::TUS), dim3(256), 0, st, q);
        XMI_HIP(hipGetLastError());
    }
    return XMI_OK;
}

template <int N, int MODE>
int launch_mfma(const TuParams &p, hipStream_t st)
{
    constexpr int S = (32 / N) * (32 / N);
    const int tiles = p.count / S;
    if (tiles > 0) {
        const int cap = resident_blocks<tu_mfma_kernel<N, MODE>>();
        const int blocks = (tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap;   // wavefronts loop over tiles
        hipLaunchKernelGGL((tu_mfma_kernel<N, MODE>), dim3(blocks), dim3(256), 0, st, p);
        XMI_HIP(hipGetLastError());
    }
    const int rest = p.count - tiles * S;                                     // < S TUs: one partial tile, on the VALU kernel
    if (rest > 0) {
        const size_t tu0 = (size_t)tiles * S, px0 = tu0 * N * N;
        auto at = [](auto *ptr, size_t off) { return ptr ? ptr + off : ptr; };
        TuParams q = { at(p.pix, px0), at(p.pred, px0), at(p.qp, tu0), at(p.flags, tu0), at(p.coef_in, px0), at(p.out16, px0),
                       at(p.rec, px0), at(p.sum, tu0), rest };
        hipLaunchKernelGGL((tu_kernel<N, MODE>), dim3((rest + Cfg<N>::TUS - 1) / Cfg<N>::TUS), dim3(256), 0, st, q);
        XMI_HIP(hipGetLastError());
    }
    return XMI_OK;
}

template <int MODE>
int launch_mfma_n(int n, const TuParams &p, hipStream_t st)
{
    if (n == 32) return launch_mfma<32, MODE>(p, st);
    if (n == 16) return launch_mfma<16, MODE>(p, st);
    return launch_mfma<8, MODE>(p, st);
}

template <int MODE>
int launch(int n, const TuParams &p, hipStream_t st)
{
    if (p.count == 0) return XMI_OK;
    switch (n) {
    case 4: hipLaunchKernelGGL((tu_kernel<4, MODE>), dim3((p.count + Cfg<4>::TUS - 1) / Cfg<4>::TUS), dim3(256), 0, st, p); break;
    case 8: hipLaunchKernelGGL((tu_kernel<8, MODE>), dim3((p.count + Cfg<8>::TUS - 1) / Cfg<8>::TUS), dim3(256), 0, st, p); break;
    case 16: hipLaunchKernelGGL((tu_kernel<16, MODE>), dim3(p.count), dim3(256), 0, st, p); break;
    case 32: hipLaunchKernelGGL((tu_kernel<32, MODE>), dim3(p.count), dim3(256), 0, st, p); break;
    default: xmi::set_error("transform size %d not in {4,8,16,32}", n); return XMI_ERR_ARG;
    }
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

int check_common(const char *who, int n, int count)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n == 4 || n == 8 || n == 16 || n == 32, "%s: transform size %d not in {4,8,16,32}", who, n);
    XMI_ARG(count >= 0, "%s: negative count", who);
    return XMI_OK;
}

}  // namespace

extern "C" {

int xmi_tu_chain_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred, const uint8_t *d_qp,
                           const uint8_t *d_flags, int16_t *d_level, uint8_t *d_rec, uint32_t *d_sum, void *stream)
{
    int rc = check_common("xmi_tu_chain_batch_dev", n, count);
    if (rc != XMI_OK) return rc;
    XMI_ARG(count == 0 || (d_pix && d_pred && d_qp && d_flags && d_level && d_rec && d_sum), "xmi_tu_chain_batch_dev: null argument");
    TuParams p = { d_pix, d_pred, d_qp, d_flags, nullptr, d_level, d_rec, d_sum, count };
#ifndef XMI_EXP_TU_VALU
    // the matrix-core kernels want 16-byte aligned pixel/level arrays and word-aligned per-TU arrays; anything else
    // (and the last partial tile of a batch) takes the VALU kernel
    if (count > 0 && (((uintptr_t)d_pix | (uintptr_t)d_pred | (uintptr_t)d_rec | (uintptr_t)d_level) & 15) == 0 &&
        (((uintptr_t)d_qp | (uintptr_t)d_flags | (uintptr_t)d_sum) & 7) == 0) {
        if (n == 4) return launch_mfma4<MODE_CHAIN>(p, xmi::as_stream(stream));
        return launch_mfma_n<MODE_CHAIN>(n, p, xmi::as_stream(stream));
    }
#endif
    return launch<MODE_CHAIN>(n, p, xmi::as_stream(stream));
}

int xmi_sub_dct_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred, const uint8_t *d_flags,
                          int16_t *d_coef, void *stream)
{
    int rc = check_common("xmi_sub_dct_batch_dev", n, count);
    if (rc != XMI_OK) return rc;
    XMI_ARG(count == 0 || (d_pix && d_pred && d_coef), "xmi_sub_dct_batch_dev: null argument");
    TuParams p = { d_pix, d_pred, nullptr, d_flags, nullptr, d_coef, nullptr, nullptr, count };
    if (count > 0 && (((uintptr_t)d_pix | (uintptr_t)d_pred | (uintptr_t)d_coef) & 15) == 0 && ((uintptr_t)d_flags & 7) == 0)
        return n == 4 ? launch_mfma4<MODE_SUBDCT>(p, xmi::as_stream(stream)) : launch_mfma_n<MODE_SUBDCT>(n, p, xmi::as_stream(stream));
    return launch<MODE_SUBDCT>(n, p, xmi::as_stream(stream));
}

int xmi_quant_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_qp, const uint8_t *d_flags,
                        int16_t *d_level, uint32_t *d_sum, void *stream)
{
    int rc = check_common("xmi_quant_batch_dev", n, count);
    if (rc != XMI_OK) return rc;
    XMI_ARG(count == 0 || (d_coef && d_qp && d_level && d_sum), "xmi_quant_batch_dev: null argument");
    TuParams p = { nullptr, nullptr, d_qp, d_flags, d_coef, d_level, nullptr, d_sum, count };
    if (count > 0 && (((uintptr_t)d_coef | (uintptr_t)d_level) & 15) == 0) return launch_quant_stream_n<false>(n, p, xmi::as_stream(stream));
    return launch<MODE_QUANT>(n, p, xmi::as_stream(stream));
}

int xmi_dequant_batch_dev(int n, int count, const int16_t *d_level, const uint8_t *d_qp, int16_t *d_coef, void *stream)
{
    int rc = check_common("xmi_dequant_batch_dev", n, count);
    if (rc != XMI_OK) return rc;
    XMI_ARG(count == 0 || (d_level && d_qp && d_coef), "xmi_dequant_batch_dev: null argument");
    TuParams p = { nullptr, nullptr, d_qp, nullptr, d_level, d_coef, nullptr, nullptr, count };
    if (count > 0 && (((uintptr_t)d_level | (uintptr_t)d_coef) & 15) == 0) return launch_quant_stream_n<true>(n, p, xmi::as_stream(stream));
    return launch<MODE_DEQUANT>(n, p, xmi::as_stream(stream));
}

int xmi_idct_add_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_pred, const uint8_t *d_flags,
                           uint8_t *d_rec, void *stream)
{
    int rc = check_common("xmi_idct_add_batch_dev", n, count);
    if (rc != XMI_OK) return rc;
    XMI_ARG(count == 0 || (d_coef && d_pred && d_rec), "xmi_idct_add_batch_dev: null argument");
    TuParams p = { nullptr, d_pred, nullptr, d_flags, d_coef, nullptr, d_rec, nullptr, count };
    if (count > 0 && (((uintptr_t)d_coef | (uintptr_t)d_pred | (uintptr_t)d_rec) & 15) == 0 && ((uintptr_t)d_flags & 7) == 0)
        return n == 4 ? launch_mfma4<MODE_IDCT>(p, xmi::as_stream(stream)) : launch_mfma_n<MODE_IDCT>(n, p, xmi::as_stream(stream));
    return launch<MODE_IDCT>(n, p, xmi::as_stream(stream));
}

}  // extern "C"
