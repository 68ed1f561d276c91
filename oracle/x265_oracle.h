/*
 * x265_oracle.h - CPU restatement of the x265-Milestone pixel primitives (TEST INFRASTRUCTURE).
 *
 * This is the checker the HIP kernels are held against. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load it.
 * Every function cites the reference file:line it restates (paths relative to the
 * reference tree). Plain C, scalar, no intrinsics: readable first, fast second.
 *
 * Parity status: pinned against the reference's own C++/SSE4 code compiled from
 * src/pixel.cpp + src/table.cpp (oracle/Makefile -> oracle/_ref/libx265ref_pixel.so) through the
 * golden vectors in tests/golden/ (generator: oracle/gen_golden.py). The reference ships no
 * test vectors of its own. Two groups are NOT pinned by reference-compiled code because the
 * reference implements them only in NASM, which cannot be assembled in this image:
 *   - orc_satd_x264 (x265_pixel_satd_{4x4,8x8,16x16}_sse4)   -> "parity unpinned" except for
 *     the 4x4 case, which must equal the reference's own xCalcHADs4x4 (checked).
 *   - orc_ssd       (x265_pixel_ssd_*)                        -> exact sum of squares.
 */
#ifndef X265_ORACLE_H
#define X265_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SLICE_P 1      /* inc/x265.h:79 */
#define ORC_SLICE_I 2      /* inc/x265.h:80 */
#define ORC_MODE_INVALID 255 /* inc/x265.h:163 */

/* A1: src/pixel.cpp:546-696 (xSad4x4..xSad64x64).  n in {4,8,16,32,64}. */
uint32_t orc_sad(int n, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);
/* rectangular entries of xSad_AMP, src/pixel.cpp:719-791 */
uint32_t orc_sad_wh(int w, int h, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);

/* A11: src/pixel.cpp:793-816 binds NASM x265_pixel_ssd_*; exact sum of squared differences. */
uint32_t orc_ssd(int n, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);

/* A3: HM-style SATD. src/pixel.cpp:63-158 (4x4, (sum+1)>>1), :160-330 (8x8, (sum+2)>>2),
 * :332-458 (16/32/64 = sum of independently rounded 8x8 tiles). n in {4,8,16,32,64}. */
uint32_t orc_satd_hm(int n, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);

/* A4: x264-style SATD bound at src/pixel.cpp:531-533; defined by src/x86/pixel-a.asm:1645 (4x4),
 * :3455 (8x8), :1787/:3282 (16x16): sum over 4x4 tiles of sum|H4*D*H4|, halved. n in {4,8,16}.
 * 32-bit accumulation (the asm's 16-bit lane overflow for adversarial 16x16 input is not
 * reproduced). */
uint32_t orc_satd_x264(int n, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);

/* The table the encoder actually calls, src/pixel.cpp:529-542 with SATD_USE_TRUE_ASM=1:
 * idx 0..2 -> x264 4x4/8x8/16x16, idx 3 -> xCalcHADs32x32_test (HM 8x8 tiles). */
uint32_t orc_satd_table(int idx, const uint8_t *src, int stride_src, const uint8_t *ref, int stride_ref);

/* A5: xSubDct, src/pixel.cpp:3446-3458 (C spec :1415 DCT4, :1567 DST4, :1826 DCT8, :2300 DCT16,
 * :3149 DCT32). dst rows use the SAME stride as src (pDst[k*nStride+i]). DST when width==4 and
 * mode != MODE_INVALID. */
void orc_sub_dct(int16_t *dst, const uint8_t *src, int stride, const uint8_t *pred, int stride_pred,
                 int width, unsigned mode);

/* A6: xQuant, src/pixel.cpp:1060-1089 (C spec). Returns sum of |level| (before sign, before clip). */
uint32_t orc_quant(int16_t *dst, const int16_t *src, int stride, unsigned qp, int width, int height,
                   int slice_type);

/* A7: xDeQuant, src/pixel.cpp:1249-1267 (C spec). */
void orc_dequant(int16_t *dst, const int16_t *src, int stride, unsigned qp, int width, int height);

/* A8: xIDctAdd, src/pixel.cpp:5938-5950 (C spec :3584 IDST4, :3725 IDCT4, :4071 IDCT8, :4629 IDCT16,
 * :5573 IDCT32). pred is read with `stride` (the coefficient stride), dst with stride_dst. */
void orc_idct_add(uint8_t *dst, int stride_dst, const int16_t *src, int stride, const uint8_t *pred,
                  int width, unsigned mode);

/* A10: xSubDctQuantRec, src/encode.cpp:3005-3021. One TU: sub+DCT -> quant -> (dequant -> IDCT+add |
 * copy pred). pix/pred/coef share `stride`, rec uses stride_rec. Returns sum|level|. */
uint32_t orc_tu_chain(const uint8_t *pix, const uint8_t *pred, uint8_t *rec, int16_t *coef,
                      int size, int stride, int stride_rec, unsigned qp, unsigned mode, int slice_type);

/*
 * Frame-wide SAD surface: for every 64x64 CTU, every integer MV (dx,dy) in [-range,range]^2 and
 * every square PU of the quad-tree (64 8x8, 16 16x16, 4 32x32, 1 64x64) the value the
 * reference's xSad_AMP[idx](cur,64,ref+dy*stride+dx,stride) would return (src/encode.cpp:2885-2931).
 * A superset of the candidates xMotionSearch visits; each value is exact.
 *
 * cur:  width x height luma (multiples of 64), stride cur_stride.
 * ref:  pointer to pixel (0,0) of a frame padded by at least `range` on every side, stride ref_stride.
 * Output planes, all indexed [ctu][blk][dy+range][dx+range] with row pitch `pitch` elements
 * (pitch >= 2*range+1) and (2*range+1) rows per block; blk is raster order inside the CTU
 * (8x8: by*8+bx, 16x16: by*4+bx, 32x32: by*2+bx); ctu is raster order in the frame.
 * Only CTUs [ctu_begin, ctu_end) are written (at their absolute index).
 */
void orc_sad_surface(const uint8_t *cur, int cur_stride, const uint8_t *ref, int ref_stride,
                     int width, int height, int range, int pitch,
                     uint16_t *sad8, uint16_t *sad16, uint32_t *sad32, uint32_t *sad64,
                     int ctu_begin, int ctu_end);

/* xGetComponentBits, src/encode.cpp:194-205. */
unsigned orc_mv_component_bits(int val);

/* N1: xPredInterLumaBlk, src/pixel.cpp:8745-8795: HEVC 8-tap luma interpolation of a w x h block
 * at quarter-pel MV (mvx,mvy) from ref (pointer at the block's integer origin). dst stride 64. */
void orc_pred_inter_luma(const uint8_t *ref, uint8_t *dst, int ref_stride, int w, int h, int mvx, int mvy);

/* xPredInterChromaBlk, src/pixel.cpp:9657-9703: 4-tap chroma prediction of one block; (mvx, mvy) are the luma
 * quarter-pel MV numbers read as eighth-pel chroma units (src/encode.cpp:3721-3723). dst rows are 32 bytes apart
 * (MAX_CU_SIZE/2). H and V: (sum + 32) >> 6; HV: first stage sum - 8192 in 16 bits, second stage
 * (sum + 8192*64 + 2048) >> 12 (C spec :8982, :9257, :9611; filters src/table.cpp:532-541). */
void orc_pred_inter_chroma(const uint8_t *ref, uint8_t *dst, int ref_stride, int w, int h, int mvx, int mvy);

/* A2: xMotionSearch, src/encode.cpp:2815-3003 (MV_CLIP=1, MVD_COST_BY_TABLE=0, SATD_FOR_FRAC_ME=0).
 * cur: stride 64. ref: pointer at the PU origin in the padded reference. mvp[2]: quarter-pel
 * predictors {x,y}. Outputs mirror the reference's out-parameters. Returns packed mv: x | y<<16. */
typedef struct { int16_t x, y; } orc_mv;
typedef struct {
    orc_mv mv;           /* returned quarter-pel MV */
    uint32_t best_cost;  /* *uiBestRDCostY */
    uint32_t best_sad;   /* *pSad_best (SAD of the chosen MVP) */
    int mvp_idx;         /* *pMVPIdx */
    int n_int_sads;      /* integer-pel SAD evaluations performed */
    int n_sub_sads;      /* sub-pel SAD evaluations performed */
} orc_me_result;
void orc_motion_search(const uint8_t *cur, const uint8_t *ref, int ref_stride, int size,
                       const orc_mv mvp[2], int zero_mv_tested, orc_mv mv_min_ipel, orc_mv mv_max_ipel,
                       double lambda, int me_range, orc_me_result *out);

/* N3: intra luma mode search, src/encode.cpp:4005-4081 with its helpers.
 * Reference sample array layout (both functions below and the search): 4*size+1 bytes,
 * [0 .. 2*size-1] = left column bottom-up (bottom-left first), [2*size] = top-left, then the top row and top-right.
 * orc_padding_ref : xPaddingRef, src/encode.cpp:3843 (valids: bit 0 = bottom-left, 1 = left, 2 = top-left, 3 = top,
 *                   4 = top-right; at least one bit set).
 * orc_filter_ref  : xFilterRef,  src/encode.cpp:3873 ([1 2 1] smoothing or, for 32x32 with strong smoothing, bilinear).
 * orc_intra_pred  : xEncIntraPred, src/pixel.cpp:7338 -> planar (:6489), DC (:6704), angular (:7243); is_luma enables the
 *                   DC / horizontal / vertical edge filters (the reference passes size != 32).
 * orc_intra_search: the 35-mode loop: prediction from the unfiltered or filtered samples (xg_aucIntraFilterType,
 *                   src/table.cpp:32), cost = satd + {1,2,3} * sqrt_lambda in double (most probable modes), strict <
 *                   in mode order; satd = the entry the reference binds in xSatdN[log2(size)-2] (x264 SATD for 4..16,
 *                   xCalcHADs32x32_test for 32); then the mapping of the best mode to its coded index. */
void orc_padding_ref(uint8_t *ref, int size, unsigned valids);
void orc_filter_ref(uint8_t *dst, const uint8_t *src, int size, int strong);
void orc_intra_pred(uint8_t *dst, const uint8_t *ref, int stride, int mode, int size, int is_luma);
typedef struct {
    int best_mode;        /* nBestModeY */
    int pred_idx;         /* iPredIdx: 0..2 = most probable mode hit, else 3 + rank among the remaining 32 */
    double best_cost;     /* dBestSATDCost */
    uint32_t satd[35];    /* SATD of every mode (before the mode-bit term) */
} orc_intra_result;
void orc_intra_search(const uint8_t *src, int src_stride, const uint8_t *ref_raw, int size, unsigned valids,
                      int strong_smoothing, const uint8_t mpm[3], double sqrt_lambda, orc_intra_result *out);

/* xFillMvpCand (src/encode.cpp:2703-2809) on picture-level maps: the two AMVP candidates and the merge candidates of one
 * PU (origin x0,y0, size_x x size_y luma samples, pu_mode 0 = 2Nx2N, 1 = 2NxN, 2 = Nx2N, pu_idx the PU's index in its CU).
 * valid: one byte per 4x4 block, non-zero where the reference's mode cache would not read MODE_INVALID (coded already);
 * mv: one vector per 8x8 block (x then y); width/4 resp. width/8 entries per row; everything outside the picture is
 * invalid. Neighbours: below-left and left beside the PU's bottom-left corner, above-right, above and above-left at its
 * top edge, exactly the cache entries the reference addresses (:2721-2732). out: amvp[2], then up to 5 merge candidates
 * (entries past the returned count are zeroed here; the reference leaves them untouched). */
int orc_fill_mvp_cand(int width, int height, const uint8_t *valid, const int16_t *mv, int x0, int y0, int size_x, int size_y,
                      int pu_mode, int pu_idx, int16_t out[7][2]);

/* Intra chroma mode search of xEncodeChromaCU (src/encode.cpp:4326-4365) for one CU: candidates planar, vertical (26),
 * horizontal (10), DC (1) and the luma mode, a candidate among the first four that equals the luma mode replaced by 34;
 * U and V predicted without the luma edge filters from their padded neighbouring samples (ref_u, ref_v: 4*size+1 bytes
 * each, layout as above), measured with the entry bound in xSatdN[log2(size)-2]; the first candidate with the smallest
 * U + V sum wins. size = 4, 8 or 16 (32 for the chroma of a 64x64 CU). */
typedef struct {
    int best_idx;         /* nBestModeC: index into the candidate list */
    int best_mode;        /* the prediction mode it stands for (realModeC) */
    int modes[5];         /* the candidate modes tried */
    uint32_t satd_u[5], satd_v[5];
} orc_chroma_result;
void orc_intra_chroma_search(const uint8_t *src_u, const uint8_t *src_v, int src_stride, const uint8_t *ref_u, const uint8_t *ref_v,
                             int size, int luma_mode, orc_chroma_result *out);

/* In-loop deblocking of one reconstructed picture, xDeblockFrame (src/Deblock.cpp:524-559): every 64x64 CTB once in
 * raster order through ff_hevc_deblocking_filter_CTB (:244-342) - its vertical edges, then its horizontal edges
 * shifted 8 samples to the left - with the luma / chroma segment filters of :61-168, tc and beta tables :195-207,
 * chroma tc :209-234. Boundary strengths (0, 1, 2) come from the CTU loop: vertical_bs[(x>>3) + (y>>2)*(width>>3)],
 * horizontal_bs[(x + y*(width>>3)) >> 2]; qp_tab holds one QP per 4x4 block, (width>>2) per row. In place. */
typedef struct {
    int width, height;            /* luma picture size */
    int stride_y, stride_c;
    int tc_offset, beta_offset;   /* DBParams of the picture (inc/x265.h:329; the reference leaves them 0) */
    int cb_qp_offset, cr_qp_offset;
} orc_deblock_params;
void orc_deblock_frame(uint8_t *y, uint8_t *u, uint8_t *v, const orc_deblock_params *p, const uint8_t *vertical_bs,
                       const uint8_t *horizontal_bs, const uint8_t *qp_tab);

/* Boundary strengths of a whole picture from picture-level maps: what the per-unit calls of
 * ff_hevc_deblocking_boundary_strengths (src/Deblock.cpp:369-503, boundary_strength :344-366; called from xEncWriteCU,
 * src/encode.cpp:4885, 4904) leave in vertical_bs / horizontal_bs once every CTU is coded. A unit is the square the
 * reference makes one call for (a CU, or one 32x32 quarter of a coded 64x64 CU); unit_log2 holds its log2 size and cbf
 * its luma cbf flag at every 4x4 position (width/4 per row), mv the motion vector of every 8x8 block (width/8 per row,
 * x then y). "Current" is always the unit's top-left entry, as in the reference. Unit edges on the 8x8 grid: 2 in I
 * slices, else 1 if either side has coefficients, else 1 if the vectors differ by 4 or more in a component; edges inside
 * a unit (every 8 samples, P slices only): the vector test alone; nothing on the picture's top and left border.
 * Both arrays are written in full: (width/8) * (height/4) entries each, indexed as orc_deblock_frame reads them. */
void orc_deblock_strengths(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf,
                           const int16_t *mv, uint8_t *vertical_bs, uint8_t *horizontal_bs);

/* calcSaoStatsCu (src/encode.cpp:364-622) for one CTU of one component plane: per SAO type (0..3 = edge offset
 * classes EO_0..EO_3, 4 = band offset) and category, the number of samples and the sum of (original - reconstructed)
 * over the CTU's window: the bottom 4 (chroma 2) rows and the right 5 (chroma 3) columns are left out unless the CTU
 * touches that picture border, edge classes also skip the picture's first / last row or column where a neighbour is
 * missing (EO_0 leaves the bottom rows out even in the last CTU row). rec / org point at the plane's sample (0,0);
 * pic_w, pic_h and ctu_size are in samples of that plane (chroma: halved). */
void orc_sao_stats_ctu(const uint8_t *rec, int rec_stride, const uint8_t *org, int org_stride, int pic_w, int pic_h,
                       int ctu_x, int ctu_y, int ctu_size, int is_chroma, int64_t count[5][33], int64_t offset_org[5][33]);

/* processSaoUnitAll / processSaoCu (src/encode.cpp:1762-1904, 1524-1760) for one component plane: every sample gets
 * the offset of its category under its CTU's SAO parameters - edge classes 0..3 from the signs against the two
 * neighbours of the class taken BEFORE any offset was applied (the reference keeps line and column buffers for
 * that), samples without both neighbours inside the picture unchanged; band offset for the four bands starting at
 * band_position (bands of 8 levels, wrapping at 32). type_idx -1 = off. The reference works in place; src and dst are
 * separate here so that the unmodified neighbours are simply src. params: one entry per CTU in raster order, taken
 * from SaoLcuParam after merging (typeIdx, subTypeIdx, offset[4], inc/x265.h:337). */
typedef struct {
    int8_t type_idx;          /* -1 off, 0..3 edge classes EO_0..EO_3, 4 band offset */
    uint8_t band_position;    /* subTypeIdx of a band offset */
    int8_t offset[4];
    uint8_t reserved[2];
} orc_sao_ctu_param;
void orc_sao_apply_plane(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width, int height, int ctu_size,
                         const orc_sao_ctu_param *params);

/* A9 "RdoQ": rate-distortion optimised quantisation (x265_rdoq.c). PARITY UNPINNED BY CONSTRUCTION: the reference holds
 * only bit-cost table builders (src/RdoQ.cpp:67-305, inc/RdoQ.h:38-49) and no level decision. orc_rdoq_est_bits restates
 * those builders; orc_rdoq_quant is this build's own decision priced with their tables (see the file header). */
typedef struct orc_rdoq_tables {
    int32_t sig_cg[2][2];     /* significantCoeffGroupBits */
    int32_t sig[27][2];       /* significantBits, indexed by getSigCtxInc */
    int32_t last_x[12], last_y[12]; /* lastXBits / lastYBits by group index */
    int32_t one[16][2];       /* greaterOneBits, 4 * context set + c1 */
    int32_t abs[4][2];        /* levelAbsBits by context set */
    int32_t cbf[2];           /* blockCbpBits of the component's first context */
} orc_rdoq_tables;
void orc_rdoq_est_bits(const uint8_t *ctx, int log2_size, int is_luma, orc_rdoq_tables *t);
void orc_rdoq_lambda(double lambda, unsigned qp, int log2_size, uint32_t *mant, int32_t *shift);
/* coef -> level (both with `stride`), returns sum |level|; scan_idx 0 diagonal, 1 horizontal, 2 vertical */
uint32_t orc_rdoq_quant(int16_t *level, const int16_t *coef, int stride, int n, unsigned qp, int is_luma, int scan_idx,
                        const orc_rdoq_tables *t, uint32_t lam_m, int32_t lam_s);

#ifdef __cplusplus
}
#endif
#endif
