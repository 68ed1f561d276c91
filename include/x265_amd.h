/*
 * x265_amd.h - C ABI of libx265_amd.so: the MI355X (gfx950) implementation of the x265-Milestone
 * pixel-primitive hot path (integer-ME SAD, SATD, SSD, forward/inverse transform + quantisation) and of the
 * data-parallel stages on either side of it (interpolation, motion and intra mode search, prediction blocks,
 * in-loop deblocking, SAO statistics and offsets, picture padding).
 *
 * Conventions
 *   - plain C, no C++ or torch types; every buffer is caller-owned.
 *   - *_dev entry points take DEVICE pointers and a hipStream_t passed as void* (NULL = default
 *     stream); they enqueue work and return without synchronising.
 *   - entry points without the _dev suffix take HOST pointers, run on the GPU and return when the
 *     result is in host memory. They are drop-ins for the reference's per-call primitives.
 *   - return value: >= 0 success (or the primitive's value), XMI_ERR_* (< 0) on failure;
 *     xmi_last_error() gives the message for the calling thread. There is NO CPU fallback: without a
 *     usable GPU every compute entry point fails with XMI_ERR_NO_DEVICE.
 *   - threads: *_dev entry points keep no state of their own (one stream per caller); host-pointer entry points
 *     share library-owned staging buffers and serialise on internal mutexes; a frame session belongs to one thread.
 *
 * Citations name the reference interface each entry point replaces (paths relative to the
 * reference tree).
 */
#ifndef X265_AMD_H
#define X265_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMI_OK             0
#define XMI_ERR_ARG       -1   /* bad argument (size, alignment, range, null pointer) */
#define XMI_ERR_HIP       -2   /* a HIP runtime call failed */
#define XMI_ERR_NO_DEVICE -3   /* no usable gfx950 device */

#define XMI_SLICE_P 1          /* xSliceType, inc/x265.h:79-80 */
#define XMI_SLICE_I 2
#define XMI_MODE_INVALID 255   /* inc/x265.h:163: "not an intra luma mode" -> DCT, not DST, for 4x4 */

/* ---- library ------------------------------------------------------------------------------ */
/* Bind the calling process to `device` and make it the calling thread's current device; idempotent. Without this call the
 * first entry point binds the calling thread's current device (HIP's default: device 0). One process, one device: run one
 * process per GPU. After binding,
 *   - entry points that take the caller's device pointers and a stream (the *_dev forms) never change a thread's current
 *     device: called from a thread whose current device is not the bound one they fail with XMI_ERR_ARG;
 *   - entry points that take host pointers or a handle of this library (per-call primitives, xmi_frame_*, xmi_analysis_*,
 *     xmi_encoder_*, the host-plane forms of the picture filters) make the bound device current for the calling thread, so
 *     they may be called from any host thread, also a fresh one. */
int         xmi_init(int device);
void        xmi_shutdown(void);        /* free library-owned device memory */
const char *xmi_last_error(void);
int         xmi_device_count(void);
const char *xmi_version(void);

/* ---- frame-wide integer-ME SAD surface ------------------------------------------------------
 * Replaces the xSad_AMP[nSadIdx](cur, 64, ref + tY*stride + tX, stride) evaluations of
 * xMotionSearch (src/encode.cpp:2885, 2907, 2931; table src/pixel.cpp:833-844; xSad8x8..64x64
 * src/pixel.cpp:578-696): for every 64x64 CTU, every integer MV (dx,dy) in [-range,range]^2 (the
 * box xCuMvClip allows, src/encode.cpp:3794-3812) and every square PU of the quad-tree
 * (64 8x8 + 16 16x16 + 4 32x32 + 1 64x64 = 85 per MV) the exact SAD.
 *
 * Output planes: sad8/sad16 are uint16 (max 65280), sad32/sad64 uint32, 180 bytes of results per (CTU, MV) in total. The
 * layout is tiled so that the GPU writes whole cache lines: main part, dx in [-R, R-1]: per (ctu, dy) one row of R/8
 * tiles of 16 consecutive dx, each tile holding all blocks of the level ([block slot][dx % 16]), followed by a few
 * cache lines of padding that are never written (3 % of the plane; rows a power of two apart stream to memory 15 %
 * slower); the dx = +R column of
 * every (ctu, dy) follows after the main part of the same plane ([ctu][dy][block slot]). ctu is raster order in the
 * frame. Use xmi_sad_surface_index() for the element index of (level, ctu, blk, dy, dx) with blk in raster order
 * inside the CTU (8x8: by*8+bx; 16x16: by*4+bx; 32x32: by*2+bx; 64x64: 0).
 * Requirements: width,height multiples of 64; range in {16,32,64,128}; the reference frame must be
 * readable range+4 pixels beyond every edge (the reference encoder pads by 80, inc/config.h:52);
 * d_cur, d_ref_origin and both strides 4-byte aligned; output planes 16-byte aligned. */
typedef struct xmi_sad_geom {
    int width, height;   /* coded luma size */
    int cur_stride;      /* bytes between rows of the current frame */
    int ref_stride;      /* bytes between rows of the padded reference frame */
    int range;           /* R */
} xmi_sad_geom;

/* elements in one surface plane set: level 0 = 8x8, 1 = 16x16, 2 = 32x32, 3 = 64x64 */
size_t xmi_sad_surface_elems(const xmi_sad_geom *g, int level);
/* element index inside the level's plane; dy,dx in [-range, range] */
size_t xmi_sad_surface_index(const xmi_sad_geom *g, int level, int ctu, int blk, int dy, int dx);

int xmi_sad_surface_dev(const xmi_sad_geom *g, const uint8_t *d_cur, const uint8_t *d_ref_origin,
                        uint16_t *d_sad8, uint16_t *d_sad16, uint32_t *d_sad32, uint32_t *d_sad64,
                        int ctu_begin, int ctu_end, void *stream);

/* host-pointer convenience form: uploads both frames, runs, downloads the four planes.
 * ref_origin points at pixel (0,0) of the padded reference; `pad` is its border in pixels. */
int xmi_sad_surface(const xmi_sad_geom *g, const uint8_t *cur, const uint8_t *ref_origin, int pad,
                    uint16_t *sad8, uint16_t *sad16, uint32_t *sad32, uint32_t *sad64);

/* ---- quarter-pel luma interpolation planes ---------------------------------------------------
 * Replaces the per-block filtering of xPredInterLumaBlk (src/pixel.cpp:8745-8795; filters src/pixel.cpp:7705,
 * 8137, 8690; taps src/table.cpp:525-530) by one pass per reference frame: plane[(yFrac*4 + xFrac)] holds, at
 * every position of the padded reference buffer, the sample xPredInterLumaBlk would produce there for that
 * phase, so   pred(x0,y0,mv)[i][j] = plane[(mv.y&3)*4 + (mv.x&3)][y0 + (mv.y>>2) + i][x0 + (mv.x>>2) + j].
 * d_ref_base: top-left of the padded buffer (`rows` x `stride` bytes, `cols` valid per row); d_planes: 16
 * planes of rows*stride bytes each. Samples closer than 4 to the buffer edge are not written. */
int xmi_interp_planes_dev(const uint8_t *d_ref_base, int stride, int rows, int cols, uint8_t *d_planes, void *stream);

/* ---- batched motion search ------------------------------------------------------------------
 * Replays xMotionSearch (src/encode.cpp:2815-3003) for `n` independent PUs: same candidate order, same cost,
 * same tie-breaking and early exits, same clip quirk - so MV, cost, SAD of the chosen predictor and predictor
 * index equal the reference's out-parameters. Sub-pel candidates read the interpolation planes made by
 * xmi_interp_planes_dev from the previous reconstruction (pixel (0,0) of the frame at (pad, pad) of every
 * plane); integer candidates read plane 0. The caller must guarantee what the reference's frame padding
 * guarantees: every block a candidate can address lies at least 4 samples inside the padded buffer.
 * Requests: PU origin and size (8/16/32/64; x0, y0 multiples of 4), the two MV predictors (quarter-pel), bZeroMV,
 * the integer clip box of xCuMvClip (src/encode.cpp:3794) and (UInt32)sqrt(lambda). me_range = ME_RANGE (32).
 * The request table is device memory, so the kernel checks each request before it reads: a size that is no PU
 * size, a PU outside the picture (rows - 2 pad lines of stride - 2 pad samples), or predictors / a clip box that
 * reach closer than 8 samples to the edge of the padded buffer (max(|mvp >> 2| + 2, |box| + 6) + 8 > pad) are not
 * searched; their result is mvp_idx = -1, best_cost = best_sad = 0xffffffff. With the reference's geometry
 * (pad 80, |MV| <= 64 + 4) no request is refused. */
typedef struct xmi_me_request {
    int32_t x0, y0, size;
    int16_t mvp[2][2];            /* [candidate][x,y] */
    int32_t zero_mv_tested;
    int16_t mv_min_ipel[2], mv_max_ipel[2];
    uint32_t sqrt_lambda;
} xmi_me_request;
typedef struct xmi_me_result {
    int16_t mv[2];                /* returned quarter-pel MV */
    uint32_t best_cost;           /* *uiBestRDCostY */
    uint32_t best_sad;            /* *pSad_best */
    int32_t mvp_idx;              /* *pMVPIdx */
    int32_t n_int_sads, n_sub_sads;
} xmi_me_result;
int xmi_me_search_batch_dev(const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows, int pad,
                            const xmi_me_request *d_req, int n, int me_range, xmi_me_result *d_res, void *stream);
/* Same search, same results, but the integer candidates (the SAD calls of src/encode.cpp:2885-2931) are read
 * from the resident SAD surface that xmi_sad_surface_dev produced for this picture pair (g: its width, height
 * and range; strides unused) instead of being computed. Applies per request when the PU lies on the CTU
 * quad-tree (x0, y0 multiples of size) and per stage when its candidates are inside [-range, range]^2; anything
 * else is computed as above. d_cur / plane 0 must be the frames the surface was made from. */
int xmi_me_search_surface_batch_dev(const uint8_t *d_cur, int cur_stride, const uint8_t *d_planes, int stride, int rows, int pad,
                                    const xmi_sad_geom *g, const uint16_t *d_sad8, const uint16_t *d_sad16,
                                    const uint32_t *d_sad32, const uint32_t *d_sad64, const xmi_me_request *d_req, int n,
                                    int me_range, xmi_me_result *d_res, void *stream);

/* ---- MV predictor and merge candidates ----------------------------------------------------------
 * xFillMvpCand (src/encode.cpp:2703-2809) for a batch of PUs, from picture-level maps instead of the CTU-local caches:
 * d_valid one byte per 4x4 block, non-zero where the reference's mode cache would not read MODE_INVALID (the block is
 * coded), d_mv one vector per 8x8 block (x then y, quarter-pel); width/4 resp. width/8 entries per row; everything outside
 * the picture counts as not coded. Request: PU origin and size in luma samples (multiples of 8, 8..64), pu_mode 0 = 2Nx2N,
 * 1 = 2NxN, 2 = Nx2N, pu_idx the PU's index in its CU. Result: the two AMVP candidates (the sMVP of xMotionSearch and
 * the mvp of xmi_me_request), the merge candidates and their number (entries past it are zero; -1 marks a request that
 * is no PU of the picture). */
typedef struct xmi_mvp_request {
    int32_t x0, y0;
    uint8_t size_x, size_y, pu_mode, pu_idx;
} xmi_mvp_request;
typedef struct xmi_mvp_result {
    int16_t amvp[2][2];
    int16_t merge[5][2];
    int32_t merge_count;
} xmi_mvp_result;
int xmi_mvp_candidates_dev(int width, int height, const uint8_t *d_valid, const int16_t *d_mv, const xmi_mvp_request *d_req, int n,
                           xmi_mvp_result *d_res, void *stream);

/* Search requests from candidate lists, on the device: d_req[i] gets PU i's origin and size (square PUs), its AMVP pair as
 * mvp, d_zero_mv_tested[i] (0 when the array is null), the clip box of its CTU (as xmi_cu_mv_clip) and sqrt_lambda - what
 * xEncDecideCU_Inter assembles before it calls xMotionSearch (src/encode.cpp:3379). A PU that xmi_mvp_candidates_dev
 * refused or that is not square becomes a request with size 0, which the search refuses in turn. */
int xmi_me_requests_from_candidates_dev(int width, int height, const xmi_mvp_request *d_pu, const xmi_mvp_result *d_cand,
                                        const uint8_t *d_zero_mv_tested, int n, uint32_t sqrt_lambda, xmi_me_request *d_req, void *stream);

/* The way back: the vectors of searches (d_res[i].mv of PU d_pu[i]; PUs on the 8x8 grid, results with mvp_idx < 0 skipped)
 * stored into the picture-level maps - every 8x8 vector entry and 4x4 coded flag the PU covers -, as the reference's CU
 * write-back does for its caches. PUs of one call must not overlap. */
int xmi_mv_field_store_dev(int width, int height, const xmi_mvp_request *d_pu, const xmi_me_result *d_res, int n, uint8_t *d_valid,
                           int16_t *d_mv, void *stream);

/* xFillMvpCand with the reference's own arguments (src/encode.cpp:2703-2711): mode_y and mv point at the PU's entries of
 * the CTU-local caches (MAX_PU_XY + 2 = 18 flags per row with MODE_INVALID = 255, MAX_MV_XY + 2 = 10 vectors per row).
 * Writes mvp[0], mvp[1] and mvp[2 .. 2 + count) and returns count (< 0 on error); runs the batch kernel on one request. */
int xmi_fill_mvp_cand(int16_t (*mvp)[2], const uint8_t *mode_y, const int16_t (*mv)[2], uint32_t size_x, uint32_t size_y, uint32_t pu_mode,
                      uint32_t pu_idx);

/* xCuMvClip (src/encode.cpp:3794-3812): the integer clip box (mv_min_ipel / mv_max_ipel of xmi_me_request) of the searches
 * in CTU (ctu_x, ctu_y) of a picture of ctus_w x ctus_h CTUs. Host arithmetic; needs no device. */
int xmi_cu_mv_clip(int ctu_x, int ctu_y, int ctus_w, int ctus_h, int16_t mv_min_ipel[2], int16_t mv_max_ipel[2]);

/* ---- intra luma mode search ---------------------------------------------------------------------
 * The 35-mode SATD search of xEncDecideCU_Intra (src/encode.cpp:4005-4081) for a batch of PUs, one request per
 * PU, with what the reference has at that point of its CTU loop: the PU's source pixels (taken from the
 * source frame at x0,y0), the 4*size+1 neighbouring reconstruction samples as the reference gathers them
 * (src/encode.cpp:4010-4015: left column bottom-up, top-left, top row, top-right; 132 bytes per request), the
 * neighbour availability bits, the three most probable modes and sqrt(lambda). The kernel restates xPaddingRef
 * (:3843), xFilterRef (:3873), xEncIntraPred (src/pixel.cpp:7338) and the SATD entry the reference binds in
 * xSatdN[log2(size)-2]; the cost is satd + {1,2,3}*sqrt_lambda in double, first minimum in mode order.
 * No counterpart call exists in the reference: the loop is inline in xEncDecideCU_Intra. */
#define XMI_INTRA_REF_PITCH 132
typedef struct xmi_intra_request {
    int32_t x0, y0;               /* PU origin in the source frame, multiples of 4 */
    uint8_t size;                 /* 4, 8, 16 or 32 */
    uint8_t valids;               /* ucValids: bit 0 bottom-left, 1 left, 2 top-left, 3 top, 4 top-right */
    uint8_t strong_smoothing;     /* bStrongIntraSmoothing */
    uint8_t mpm[3];               /* aucMostModeY */
    uint8_t reserved[2];
    double sqrt_lambda;           /* dSqrtLambda */
} xmi_intra_request;
typedef struct xmi_intra_result {
    int32_t best_mode;            /* nBestModeY */
    int32_t pred_idx;             /* iPredIdx */
    double best_cost;             /* dBestSATDCost */
    uint32_t satd[35];            /* SATD of every mode */
    uint32_t reserved;
} xmi_intra_result;
int xmi_intra_search_batch_dev(const uint8_t *d_src, int src_stride, const uint8_t *d_ref_samples,
                               const xmi_intra_request *d_req, int n, xmi_intra_result *d_res, void *stream);
/* The step in front of the search, src/encode.cpp:3965-3974 (availability) and :4010-4015 (samples): for every request
 * (x0, y0, size filled in) the 4*size+1 neighbouring samples out of the reconstruction plane d_rec (width x height, rows
 * rec_stride apart) into row i of d_ref_samples, and the availability bits into the request's valids, from d_coded: one
 * byte per 4x4 unit of the picture (rows coded_stride apart), nonzero where the unit is reconstructed - what the
 * reference asks of its mode cache (aucModeY != MODE_INVALID). A segment is available when its first unit is (bottom-left:
 * the unit under the left column; top-right: the unit right of the top row), as in the reference; units outside the
 * picture are not. Samples of unavailable segments are written as 0x80; the search pads over them (xPaddingRef). */
int xmi_intra_gather_dev(const uint8_t *d_rec, int rec_stride, int width, int height, const uint8_t *d_coded, int coded_stride,
                         xmi_intra_request *d_req, int n, uint8_t *d_ref_samples, void *stream);

/* ---- intra chroma mode search --------------------------------------------------------------------
 * The candidate loop of xEncodeChromaCU (src/encode.cpp:4326-4365) for a batch of CUs: candidates planar, vertical (26),
 * horizontal (10), DC (1) and the luma mode - one of the first four that equals the luma mode is replaced by 34 -, U and
 * V predicted by xEncIntraPred without the luma edge filters, each measured with the SATD the reference binds in
 * xSatdN[log2(size) - 2] (src/encode.cpp:4353), the first candidate with the smallest U + V sum kept. Request i: the
 * CU's origin in the chroma planes (x0 a multiple of 4), its chroma size (4, 8 or 16) and luma mode; its neighbouring
 * samples AFTER xPaddingRef (src/encode.cpp:4322), layout as for the luma search, U at d_ref_samples + 2 i *
 * XMI_INTRA_REF_PITCH and V one pitch further. A request the kernel cannot serve (other sizes, mode above 34, bad
 * origin) gets best_idx = best_mode = -1. */
typedef struct xmi_chroma_request {
    int32_t x0, y0;
    uint8_t size, luma_mode, reserved[2];
} xmi_chroma_request;
typedef struct xmi_chroma_result {
    int32_t best_idx;             /* nBestModeC: index into the candidate list */
    int32_t best_mode;            /* the prediction mode it stands for */
    uint32_t satd_u[5], satd_v[5];
} xmi_chroma_result;
int xmi_intra_chroma_search_batch_dev(const uint8_t *d_src_u, const uint8_t *d_src_v, int src_stride, const uint8_t *d_ref_samples,
                                      const xmi_chroma_request *d_req, int n, xmi_chroma_result *d_res, void *stream);
/* one CU with host pointers: the U and V source blocks (rows src_stride apart) and their padded samples */
int xmi_intra_chroma_search(const uint8_t *src_u, const uint8_t *src_v, uint32_t src_stride, const uint8_t *ref_u, const uint8_t *ref_v,
                            int size, int luma_mode, xmi_chroma_result *res);

/* ---- prediction blocks -------------------------------------------------------------------------------
 * Inter: what xPredInterLumaBlk (src/pixel.cpp:8745-8795, 8-tap, quarter-pel) and xPredInterChromaBlk
 * (src/pixel.cpp:9657-9703, 4-tap, the same MV numbers read as eighth-pel) write for one PU, for a batch of PUs
 * straight from the padded reconstruction on the device (d_ref_* = pixel (0,0) of each plane). Blocks are written
 * packed (rows `size` resp. `size`/2 bytes apart) at the byte offsets the request names, ready to be the `pred`
 * input of xmi_tu_chain_batch_dev. A null d_pred_y skips luma, null d_pred_u / d_pred_v skip that chroma plane.
 * A block's filter window is the samples the reference's filters read (none beyond the block when a fraction is 0);
 * with strides that are multiples of 4 it is fetched as the aligned dwords that contain it, i.e. up to 3 bytes more on
 * either side of every window row. */
typedef struct xmi_inter_pred_request {
    int32_t x0, y0, size;         /* luma PU origin and size (8/16/32/64); chroma blocks are size/2 at (x0/2, y0/2) */
    int16_t mv[2];                /* quarter-pel luma MV */
    uint32_t out_y, out_c;        /* byte offsets of the luma block in d_pred_y and of the chroma blocks in d_pred_u / d_pred_v */
} xmi_inter_pred_request;
int xmi_pred_inter_batch_dev(const uint8_t *d_ref_y, int stride_y, const uint8_t *d_ref_u, const uint8_t *d_ref_v, int stride_c,
                             const xmi_inter_pred_request *d_req, int n, uint8_t *d_pred_y, uint8_t *d_pred_u, uint8_t *d_pred_v,
                             void *stream);
/* Intra: xEncIntraPred (src/pixel.cpp:7338) for a batch: request i predicts a size x size block in `mode` (0 planar,
 * 1 DC, 2..34 angular; is_luma switches the DC / horizontal / vertical edge filters) from its 4*size+1 neighbouring
 * samples at d_ref_samples + i * XMI_INTRA_REF_PITCH (bottom-left first, as the reference's aucRef rows; already padded
 * and, where the caller wants the smoothed set, filtered) and writes it packed at d_pred + out. */
typedef struct xmi_intra_pred_request {
    uint8_t size, mode, is_luma, reserved;
    uint32_t out;
} xmi_intra_pred_request;
int xmi_intra_pred_batch_dev(const uint8_t *d_ref_samples, const xmi_intra_pred_request *d_req, int n, uint8_t *d_pred, void *stream);

/* ---- picture-border padding --------------------------------------------------------------------------
 * xFramePadding (src/encode.cpp:2043-2092) for one plane: `pad` samples of border replication on every side (PAD_YSIZE =
 * 80 for luma, PAD_CSIZE = 40 for chroma in the reference, inc/config.h:52). plane points at sample (0,0); the buffer
 * holds pad rows above and below and pad columns left and right; stride >= width + 2 pad. */
int xmi_frame_padding_dev(uint8_t *d_plane, int stride, int width, int height, int pad, void *stream);
int xmi_frame_padding(uint8_t *plane, int stride, int width, int height, int pad);

/* ---- in-loop deblocking -----------------------------------------------------------------------------
 * xDeblockFrame (src/Deblock.cpp:524-559) for one reconstructed 4:2:0 picture, in place: every vertical edge on the
 * 8-sample grid, then every horizontal edge, each 4-sample segment filtered when the boundary strength the CTU loop
 * stored for it (ff_hevc_deblocking_boundary_strengths, src/Deblock.cpp:369) is 1 or 2; chroma on the 16-sample luma
 * grid for strength 2. Tables as the reference keeps them: vertical_bs[(x>>3) + (y>>2)*(width>>3)],
 * horizontal_bs[(x + y*(width>>3)) >> 2] (both (width/8)*(height/4) bytes are read, the size xmi_deblock_strengths
 * writes; the reference allocates more), qp_tab one QP per 4x4 block, width/4 per row, (width/4)*(height/4) bytes.
 * tc_offset / beta_offset are the picture's DBParams (inc/x265.h:329), cb/cr_qp_offset the PPS offsets. A picture of one
 * 64-sample CTB row or column (but not both) has only its last CTB filtered, as the reference's loop gets to no other
 * (hls_filters, src/Deblock.cpp:512-522). */
typedef struct xmi_deblock_params {
    int32_t width, height;            /* luma picture size, multiples of 8 */
    int32_t stride_y, stride_c;
    int32_t tc_offset, beta_offset;
    int32_t cb_qp_offset, cr_qp_offset;
} xmi_deblock_params;
int xmi_deblock_frame_dev(uint8_t *d_y, uint8_t *d_u, uint8_t *d_v, const xmi_deblock_params *p, const uint8_t *d_vertical_bs,
                          const uint8_t *d_horizontal_bs, const uint8_t *d_qp_tab, void *stream);
/* the same with host pointers: planes and tables are copied to the device, filtered, and the planes copied back */
int xmi_deblock_frame(uint8_t *y, uint8_t *u, uint8_t *v, const xmi_deblock_params *p, const uint8_t *vertical_bs,
                      const uint8_t *horizontal_bs, const uint8_t *qp_tab);

/* The boundary strengths themselves, for a whole picture at once, from picture-level maps of the decisions: replaces the
 * per-unit calls of ff_hevc_deblocking_boundary_strengths (src/Deblock.cpp:369-503; called from xEncWriteCU,
 * src/encode.cpp:4885, 4904) and its boundary_strength (:344-366). A unit is the square the reference makes one call for
 * (a CU, or one 32x32 quarter of a coded 64x64 CU). unit_log2: log2 of the unit's size, and cbf: its luma cbf flag, at
 * every 4x4 position (width/4 per row); mv: the motion vector of every 8x8 block (width/8 per row, x then y, quarter-pel;
 * may be null for an I slice). Both tables are written in full, (width/8)*(height/4) entries each, indexed as
 * xmi_deblock_frame_dev reads them. */
int xmi_deblock_strengths_dev(int width, int height, int intra_slice, const uint8_t *d_unit_log2, const uint8_t *d_cbf, const int16_t *d_mv,
                              uint8_t *d_vertical_bs, uint8_t *d_horizontal_bs, void *stream);
int xmi_deblock_strengths(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                          uint8_t *vertical_bs, uint8_t *horizontal_bs);

/* xmi_deblock_strengths for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc): same arguments, same tables, but the maps and tables are staged in an array
 * that is part of the library's device image (2.25 MiB per GPU the library runs on, held until the process ends) instead
 * of a buffer the library allocates, so the call performs no hipMalloc / hipFree. It has no counterpart in the reference.
 * Limit: width * height <= 4096 * 2304 luma samples, larger pictures are refused with XMI_ERR_ARG. Needs xmi_init() (or the
 * implicit binding of a one-device process) on the device it runs on. All calls on one GPU share that GPU's array: calls
 * of one process take turns on an internal mutex; the array keeps the last call's contents, and every byte a call reads
 * is written by that call first. The call copies synchronously and launches on the null stream: it is not for use while
 * a stream capture is in progress in the process. */
int xmi_deblock_strengths_static(int width, int height, int intra_slice, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                                 uint8_t *vertical_bs, uint8_t *horizontal_bs);

/* ---- SAO statistics -----------------------------------------------------------------------------------
 * calcSaoStatsCu (src/encode.cpp:364-622) for every 64x64 CTU and component of a deblocked 4:2:0 picture in one launch:
 * stats[ctu * 3 + component] holds, per SAO type (0..3 = edge classes EO_0..EO_3, 4 = band offset) and category, the
 * number of samples and the sum of (source - reconstruction) - what the reference leaves in h->m_count[component] and
 * h->m_offsetOrg[component] (inc/x265.h:545) after the call for that CTU, with its windows (bottom 4 / 2 rows and right
 * 5 / 3 columns left out inside the picture, picture-border samples without a neighbour left out of the edge classes). */
typedef struct xmi_sao_stats {
    int32_t count[5][33];
    int32_t offset_org[5][33];
} xmi_sao_stats;
typedef struct xmi_sao_geom {
    int32_t width, height;                       /* luma picture size (even) */
    int32_t rec_stride_y, rec_stride_c, org_stride_y, org_stride_c;
} xmi_sao_geom;
int xmi_sao_stats_dev(const xmi_sao_geom *g, const uint8_t *d_rec_y, const uint8_t *d_rec_u, const uint8_t *d_rec_v,
                      const uint8_t *d_org_y, const uint8_t *d_org_u, const uint8_t *d_org_v, xmi_sao_stats *d_stats, void *stream);
/* the same with host pointers (planes are copied to the device, the tables copied back) */
int xmi_sao_stats_frame(const xmi_sao_geom *g, const uint8_t *rec_y, const uint8_t *rec_u, const uint8_t *rec_v, const uint8_t *org_y,
                  const uint8_t *org_u, const uint8_t *org_v, xmi_sao_stats *stats);

/* processSaoUnitAll (src/encode.cpp:1762-1904) for one component plane: every sample gets the offset of its category
 * under its CTU's decided parameters (SaoLcuParam after merging, inc/x265.h:337: typeIdx -1 off / 0..3 edge classes / 4
 * band offset, subTypeIdx = first band, offset[4]); categories are derived from the samples BEFORE any offset, samples
 * of an edge class without both neighbours inside the picture stay. d_src and d_dst are different planes; the
 * host-pointer form filters in place. ctu_size: 64 for luma, 32 for 4:2:0 chroma; params in CTU raster order. */
typedef struct xmi_sao_ctu_param {
    int8_t type_idx;
    uint8_t band_position;
    int8_t offset[4];
    uint8_t reserved[2];
} xmi_sao_ctu_param;
int xmi_sao_apply_plane_dev(const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int width, int height, int ctu_size,
                            const xmi_sao_ctu_param *d_params, void *stream);
int xmi_sao_apply_plane(uint8_t *plane, int stride, int width, int height, int ctu_size, const xmi_sao_ctu_param *params);

/* ---- frame session: host-pointer batch interface --------------------------------------------------
 * For a host CTU loop without device pointers of its own. The session holds one picture pair on the device (source
 * luma, padded previous reconstruction, its 16 quarter-pel planes) and stages request / result tables.
 * xmi_frame_begin replaces, per P picture, every later xPredInterLumaBlk (src/pixel.cpp:8745) of the motion search;
 * xmi_frame_me_search a batch of xMotionSearch calls (src/encode.cpp:2815); xmi_frame_intra_search a batch of
 * the 35-mode loops (src/encode.cpp:4005-4081); xmi_frame_tu_chain a batch of xSubDctQuantRec (src/encode.cpp:3005)
 * with packed n x n blocks in host memory. All calls are synchronous; one thread per session.
 * ref_padded points at the first byte of the padded reconstruction ((height + 2 pad) rows of width + 2 pad). */
typedef struct xmi_frame xmi_frame;
int xmi_frame_create(int width, int height, int pad, xmi_frame **out);
int xmi_frame_destroy(xmi_frame *f);
int xmi_frame_begin(xmi_frame *f, const uint8_t *cur, int cur_stride, const uint8_t *ref_padded, int ref_stride);
/* Optional, after xmi_frame_begin: build the SAD surface of the picture pair (xmi_sad_surface_dev, integer MVs in
 * [-range, range]^2; width and height must be multiples of 64 and pad >= range + 4). Later xmi_frame_me_search
 * calls for this picture then read their integer candidates from it. The next xmi_frame_begin drops it. */
int xmi_frame_sad_surface(xmi_frame *f, int range);
/* copies the 16 quarter-pel planes (plane yFrac*4 + xFrac, each (height + 2 pad) x (width + 2 pad) bytes) to host memory */
int xmi_frame_planes_host(xmi_frame *f, uint8_t *planes);
int xmi_frame_me_search(xmi_frame *f, const xmi_me_request *req, int n, int me_range, xmi_me_result *res);
int xmi_frame_intra_search(xmi_frame *f, const uint8_t *ref_samples, const xmi_intra_request *req, int n, xmi_intra_result *res);
int xmi_frame_tu_chain(xmi_frame *f, int n, int count, const uint8_t *pix, const uint8_t *pred, const uint8_t *qp,
                       const uint8_t *flags, int16_t *level, uint8_t *rec, uint32_t *sum);

/* ---- P-picture analysis ----------------------------------------------------------------------------------
 * The reference's CTU loop for inter pictures - xEncodeCUs (src/encode.cpp:5203-5591) with xEncDecideCU_Inter (:3080-3639),
 * xPreEncRec (:3023-3078), the bit estimate of xEncodeCoeffNxN (:2385-2553) / xEstimateCmdsBit (:162) and the chroma pass
 * xEncInterCTU (:3641-3791) - as one call per picture: mode decision and reconstruction of every CTU, with the decisions,
 * costs and contexts the reference's loop arrives at. The serial part (decision recursion, CABAC rate estimate) runs on
 * the host, one thread per CTU row in the reference's wavefront order; every pixel operation of a CU - merge candidates,
 * motion search, prediction, transform chains, distortions - is batched over the rows that are ready and runs on the GPU.
 *
 * xmi_analysis_create: width, height = coded size (multiples of 64). NULL on failure (xmi_last_error()).
 * xmi_analysis_configure: lanes (1..8) = coordinators, each with a stream and buffers of its own. All of them serve all CTU
 *   rows: one takes the requests that are posted when it looks and makes a round trip to the GPU with them; with more than
 *   one, the requests of rows that were still busy on the host go out with the next coordinator while the first round trip
 *   is under way. spin_us: how long the host threads of the loop poll for a result before they go to
 *   sleep (a round trip takes about 30 us, waking a sleeping thread 10 more); 0, the default, sleeps at once and is the
 *   choice when several handles share the host's cores. With spin_us > 0 a round trip of a P picture also hands its answers
 *   out as soon as the parts that are read first are complete (the merge results of the CUs that asked, chroma), and the
 *   rows poll for the rest when they come to need it. Results depend on neither. Not while a picture is in flight.
 *   (The reference has no counterpart: its -t option sets the number of row threads, src/encode.cpp:5800-5860.)
 * xmi_analysis_state : the state the reference carries from picture to picture and never resets - per CTU and depth the
 *   FAST_MD cost statistics (dSumOfRdCosts, nRdCounts, dAverageRdCosts, inc/x265.h:379-381: (height/64) x (width/64) x 5
 *   entries each) and per CTU row the fractional bits of its rate counter (rdoParam[y].fracBits, inc/x265.h:359). set != 0
 *   loads it, set == 0 reads it back. A new handle starts from zeros, as the reference does.
 * xmi_analysis_p_picture: one P picture. src_*: source planes, packed (width x height, width/2 x height/2); ref_*: first
 *   byte of the padded previous reconstruction ((width + 2 pad) x (height + 2 pad), chroma with pad/2), pad = PAD_YSIZE = 80
 *   (inc/config.h:52); qp: slice QP. Outputs (any may be NULL), picture-level arrays of what the reference keeps per CTU:
 *   info0 / info1 = ausModeYInfo0 / aucModeYInfo1 per 4x4 block (inc/x265.h:220, 228), mv / mvd = vectors and vector
 *   differences per 8x8 block (x, y), mvp_idx per 8x8 block, coef_* = levels, rec_* = unfiltered reconstruction, rd_cost =
 *   the depth-0 return value of xEncDecideCU_Inter per CTU, ctx_in / ctx_out = 256 estimation contexts per CTU before and
 *   after it (cabacRdo[y][0], cabacRdoLast[y]), counters = { evaluator round trips, CU requests, chroma requests }. */
typedef struct xmi_analysis xmi_analysis;
xmi_analysis *xmi_analysis_create(int width, int height);
void xmi_analysis_destroy(xmi_analysis *a);
int xmi_analysis_configure(xmi_analysis *a, int lanes, int spin_us);
/* How far a CU's request to the GPU reaches ahead (no counterpart in the reference; results do not depend on it). 0: one
 * request per CU. 1: a CU's request takes its first child, that child's first child, ... along (same corner, same
 * neighbours). 2 (default): it takes every CU that may still be decided in the CTU, under the guess that all of them end up
 * with the vector of the asking CU's first merge candidate; a CU whose real neighbours are the guessed ones finds its answer
 * waiting. Fewer round trips where motion is coherent, more pixel work. */
int xmi_analysis_speculate(xmi_analysis *a, int level);
/* Host threads that run the CTU rows of a picture (each row is a fiber of one of them; no counterpart in the reference, results
 * do not depend on it). 0 (default): as many as the process has cores for, at most one per row - the choice for one picture
 * stream. A caller that runs several handles at once on one host (GOP segments) gives each a share of the cores, so that
 * their threads can poll (spin_us) instead of sleeping. */
int xmi_analysis_workers(xmi_analysis *a, int workers);
/* Measurement aid (no counterpart in the reference): where the round trips of this handle have spent their time so far,
 * summed over its lanes: out = { ns writing job tables, ns in the launch call, ns waiting for the GPU, ns reading results
 * back, round trips, and for the last picture: ns the CTU-row threads spent on decision work between their requests,
 * requests they made }. What is left of a picture's wall time is thread hand-over. */
int xmi_analysis_timing(xmi_analysis *a, uint64_t out[7]);
int xmi_analysis_state(xmi_analysis *a, int set, double *sum_cost, int32_t *count, double *avg_cost, uint64_t *frac_bits);
int xmi_analysis_p_picture(xmi_analysis *a, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const uint8_t *ref_y,
                           const uint8_t *ref_u, const uint8_t *ref_v, int pad, int qp, uint16_t *info0, uint8_t *info1, int16_t *mv,
                           int16_t *mvd, uint8_t *mvp_idx, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint8_t *rec_y, uint8_t *rec_u,
                           uint8_t *rec_v, uint32_t *rd_cost, uint8_t *ctx_in, uint8_t *ctx_out, uint64_t *counters);

/* ---- the encoder ------------------------------------------------------------------------------------------
 * xEncodeFrame (src/encode.cpp:5752-6118) behind the reference's own outer interface, xIEncFrame(X265_t *, xFrame *, out, size)
 * -> bytes (src/interface.cpp:62): one picture in, its access unit out - VPS / SPS / PPS in front of an I picture, the slice
 * (one CABAC substream per CTU row with entry points, as the reference writes them), the picture-hash SEI of the final
 * reconstruction. A picture runs the analysis loop above (for I pictures xEncDecideCU_Intra :3906 and xEncodeChromaCU :4234:
 * per PU the 35-mode search, prediction, transform chain and distortion as one GPU request), deblocking (boundary strengths
 * from the decision maps, xDeblockFrame), the SAO statistics, decision (rdoSaoUnitAll :1348, host) and offsets, and the
 * entropy coder (xEncWriteCU :4547, xCompressCU :5594; host). Constant QP, one reference picture, no rate control - the
 * configuration of every published figure of the reference. The stream equals the reference's byte for byte when the
 * caller feeds the pictures its command line would (src/x265.cpp:227-252: picture 0 and every (-ip)th picture as I slices;
 * sizes that are no multiple of 64 padded by xFrameAligning, src/preProcess.cpp:33).
 *
 * xmi_encoder_create   : width, height = coded size (multiples of 64), qp 0..51, strong_intra_smoothing = the -sis switch.
 * xmi_encoder_configure: lanes / spin_us of the analysis loop (xmi_analysis_configure).
 * xmi_encoder_encode   : src_* packed source planes at coded size; slice_type XMI_SLICE_I or XMI_SLICE_P (a P picture needs a
 *                        picture before it); out / capacity: at least 4096 bytes. Returns the size of the access unit or
 *                        XMI_ERR_*. Nothing is ever written beyond out + capacity: an access unit that does not fit fails with
 *                        XMI_ERR_ARG ("access unit exceeds capacity"); the slice data itself is limited to 3 * width * height
 *                        + 4096 bytes, so a capacity of 3 * width * height + 8192 never is the limit. (The reference stops
 *                        at 16 384 CABAC commands per CTU, inc/x265.h:416, long before that.) After a failed call the handle
 *                        holds no reference picture: the next picture must be an I picture, and since the statistics
 *                        carried from picture to picture may have advanced, streams no longer equal the reference's.
 * xmi_encoder_submit / xmi_encoder_collect: xmi_encoder_encode in two steps, for callers that can take the access unit of
 *                        picture n after handing in picture n + 1 (no counterpart in the reference, whose xIEncFrame returns the
 *                        picture's bytes). submit does everything up to the slice data and the reference picture of the next
 *                        call, and leaves the picture hash - MD5 over the final reconstruction, a serial pass of 2.6 ms for a
 *                        1920x1088 luma plane - to two threads of its own; collect waits for them, appends the hash unit and
 *                        returns the oldest waiting access unit (same bytes as xmi_encoder_encode's). Two pictures may wait
 *                        at most; a third submit, an xmi_encoder_encode while one waits, and a collect without one fail with
 *                        XMI_ERR_ARG; a collect whose buffer is too small fails and the picture keeps waiting.
 * xmi_encoder_reconstruction: the last picture's reconstruction after the in-loop filters, packed planes.
 * xmi_encoder_timing   : of the last picture, out = { ns analysis, ns deblocking + SAO statistics, ns SAO decision + offsets, ns
 *                        writing, GPU round trips, CU / PU requests, and of the round trips: ns writing job tables, ns in the
 *                        launch call, ns waiting for the GPU, ns reading results } (no counterpart in the reference). */
typedef struct xmi_encoder xmi_encoder;
xmi_encoder *xmi_encoder_create(int width, int height, int qp, int strong_intra_smoothing);
void xmi_encoder_destroy(xmi_encoder *e);
int xmi_encoder_configure(xmi_encoder *e, int lanes, int spin_us);
int xmi_encoder_speculate(xmi_encoder *e, int level);      /* xmi_analysis_speculate for the encoder's analysis loop */
int xmi_encoder_workers(xmi_encoder *e, int workers);      /* xmi_analysis_workers for the encoder's analysis loop */
/* Extension without a counterpart in the reference (its RdoQ.cpp is compiled out): from the next picture on, the transform units
 * of P pictures (luma and chroma) take their levels from the decision of xmi_rdoq_quant_batch_dev instead of xQuant - tables
 * from the contexts the slice starts with, the decision loop's own Lagrange multiplier, diagonal scan. The stream stays a valid
 * stream whose picture hash matches the encoder's reconstruction, and is no longer the reference's. Off by default. */
int xmi_encoder_rdoq(xmi_encoder *e, int on);
/* The picture-hash SEI behind every slice (MD5 of the reconstruction, src/encode.cpp:6012-6053) is the reference's default
 * build (CMake option CHECK_SEI, src/CMakeLists.txt:37-42: on); on = 0 gives the stream of its build without that option - no
 * hash unit, and none of the 2.6 ms (1920x1088) or 11.8 ms (3840x2176) that hashing a picture takes. From the next picture on. */
int xmi_encoder_picture_hash(xmi_encoder *e, int on);
int64_t xmi_encoder_encode(xmi_encoder *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, uint8_t *out,
                           size_t capacity);
int xmi_encoder_submit(xmi_encoder *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type);
int64_t xmi_encoder_collect(xmi_encoder *e, uint8_t *out, size_t capacity);
int xmi_encoder_reconstruction(xmi_encoder *e, uint8_t *y, uint8_t *u, uint8_t *v);
int xmi_encoder_timing(xmi_encoder *e, uint64_t out[10]);

/* ---- batched block metrics -------------------------------------------------------------------
 * count pairs of n x n blocks, packed back to back (block i at offset i*n*n), one uint32 per pair.
 *   XMI_METRIC_SAD       xSadN[log2(n)-2]                        src/pixel.cpp:546-717
 *   XMI_METRIC_SSD       xSsdN[log2(n)-2]                        src/pixel.cpp:793-816
 *   XMI_METRIC_SATD_HM   xCalcHADs4x4/8x8/16x16/32x32/64x64      src/pixel.cpp:63-458
 *   XMI_METRIC_SATD_X264 x265_pixel_satd_{4x4,8x8,16x16}_sse4    src/x86/pixel-a.asm:1645,3455,1787
 *   XMI_METRIC_SATD_TAB  the table the encoder calls, xSatdN[]   src/pixel.cpp:529-542
 *                        (n=4,8,16 -> x264 form, n=32 -> xCalcHADs32x32_test) */
#define XMI_METRIC_SAD       0
#define XMI_METRIC_SSD       1
#define XMI_METRIC_SATD_HM   2
#define XMI_METRIC_SATD_X264 3
#define XMI_METRIC_SATD_TAB  4

int xmi_block_metric_batch_dev(int metric, int n, int count, const uint8_t *d_a, const uint8_t *d_b,
                               uint32_t *d_out, void *stream);

/* ---- batched transform-unit chain ------------------------------------------------------------
 * Replaces xSubDctQuantRec (src/encode.cpp:3005-3021) = xSubDct -> xQuant -> (xDeQuant -> xIDctAdd |
 * copy pred) for `count` independent n x n TUs, n in {4,8,16,32}. Blocks are packed back to back.
 * Per TU i: qp[i] in 0..51, flags[i] bit0 = slice is I (rounding 171/512, else 85/512),
 * bit1 = use the 4x4 DST (reference: width==4 && mode != MODE_INVALID).
 * Outputs: level (int16 n*n), rec (uint8 n*n), sum[i] = sum |level| (the reference's return value). */
int xmi_tu_chain_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred,
                           const uint8_t *d_qp, const uint8_t *d_flags,
                           int16_t *d_level, uint8_t *d_rec, uint32_t *d_sum, void *stream);

/* the four stages on their own, same packing (for callers that interleave host logic) */
int xmi_sub_dct_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred,
                          const uint8_t *d_flags, int16_t *d_coef, void *stream);
int xmi_quant_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_qp,
                        const uint8_t *d_flags, int16_t *d_level, uint32_t *d_sum, void *stream);
int xmi_dequant_batch_dev(int n, int count, const int16_t *d_level, const uint8_t *d_qp,
                          int16_t *d_coef, void *stream);
int xmi_idct_add_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_pred,
                           const uint8_t *d_flags, uint8_t *d_rec, void *stream);

/* ---- "RdoQ": rate-distortion optimised quantisation --------------------------------------------------
 * What the reference holds under this name is compiled out (#if RDOQ, src/RdoQ.cpp:32) and consists of the bit-cost table
 * builders estimateBit / estCBFBit / estSignificantCoeffGroupMapBit / estSignificantMapBit / estSignificantCoefficientsBit
 * (src/RdoQ.cpp:67-305) over estBitsSbacStruct (inc/RdoQ.h:38-49); the level decision (My_xRateDistOptQuant, inc/RdoQ.h:55)
 * is declared and never defined. So there is NO reference behaviour to match: the bit-exact quantiser of the reference is
 * xQuant (xmi_quant_*), and these entry points are an extension that changes the levels - validated against the CPU
 * checker's statement of the same algorithm, never part of a parity claim.
 *   xmi_rdoq_tables          estBitsSbacStruct for one block size and component, costs in 1/32768 bit.
 *   xmi_rdoq_est_bits        the table builders: from context states in the reference's layout (inc/x265.h:286-311). Host.
 *   xmi_rdoq_initial_contexts the states a slice starts from (xCabacInit, src/bitstream.cpp:588). Host.
 *   xmi_rdoq_lambda_of       the weight of a bit for one QP and size from the Lagrange multiplier the reference's decision
 *                            loops use (0.578 * 2^((QP-12)/3) in P slices, 0.57 * ... in I slices, src/encode.cpp:5348, 5406). Host.
 *   xmi_rdoq_quant_batch_dev xmi_quant_batch_dev with the decision: per coefficient the rounded level, one less or zero
 *                            (cost = squared error + weight * bits of the reference's residual syntax, src/encode.cpp:2385-2553),
 *                            4x4 groups dropped when their flag alone is cheaper, the last position moved back over levels of
 *                            magnitude one, the empty block. d_tables: one table set; d_lambda: 52 entries, by QP; scan_idx
 *                            0 diagonal / 1 horizontal / 2 vertical; d_flags is accepted and not read.
 *   xmi_tu_chain_rdoq_batch_dev  xSubDct -> decision -> xDeQuant -> xIDctAdd for packed TUs (xmi_tu_chain_batch_dev with the
 *                            decision in the quantiser's place); d_work: count * n * n int16 of scratch the caller owns. */
typedef struct xmi_rdoq_tables {
    int32_t sig_cg[2][2];            /* significantCoeffGroupBits[ctx][bin] */
    int32_t sig[27][2];              /* significantBits[getSigCtxInc][bin] */
    int32_t last_x[12], last_y[12];  /* lastXBits / lastYBits by group index */
    int32_t one[16][2];              /* greaterOneBits[4 * context set + c1][bin] */
    int32_t abs[4][2];               /* levelAbsBits[context set][bin] */
    int32_t cbf[2];                  /* blockCbpBits of the component's first context */
} xmi_rdoq_tables;
typedef struct xmi_rdoq_lambda {
    uint32_t mantissa;               /* rate cost = (mantissa * bits) >> shift, in squared sixteenths of a level */
    int32_t shift;
} xmi_rdoq_lambda;
int xmi_rdoq_est_bits(const uint8_t *ctx, int log2_size, int is_luma, xmi_rdoq_tables *out);
int xmi_rdoq_initial_contexts(int slice_type, int qp, uint8_t ctx[256]);
int xmi_rdoq_lambda_of(double lambda, int qp, int log2_size, xmi_rdoq_lambda *out);
int xmi_rdoq_quant_batch_dev(int n, int count, const int16_t *d_coef, const uint8_t *d_qp, const uint8_t *d_flags, const xmi_rdoq_tables *d_tables,
                             const xmi_rdoq_lambda *d_lambda, int scan_idx, int is_luma, int16_t *d_level, uint32_t *d_sum, void *stream);
int xmi_tu_chain_rdoq_batch_dev(int n, int count, const uint8_t *d_pix, const uint8_t *d_pred, const uint8_t *d_qp, const uint8_t *d_flags,
                                const xmi_rdoq_tables *d_tables, const xmi_rdoq_lambda *d_lambda, int scan_idx, int is_luma, int16_t *d_work,
                                int16_t *d_level, uint8_t *d_rec, uint32_t *d_sum, void *stream);

/* ---- per-call drop-ins (HOST pointers, strided, synchronous) ---------------------------------
 * Same argument meaning as the reference primitives; each call stages its operands to the GPU,
 * runs the batch kernel with count = 1 and copies the result back. Correct and slow: they exist so
 * an unmodified caller can be re-pointed at the GPU library; throughput comes from the batch forms. */
int64_t xmi_sad(int n, const uint8_t *src, uint32_t stride_src, const uint8_t *ref, uint32_t stride_ref);  /* xSadN / xSad_AMP */
int64_t xmi_ssd(int n, const uint8_t *src, intptr_t stride_src, const uint8_t *ref, intptr_t stride_ref);  /* xSsdN */
int64_t xmi_satd(int n, const uint8_t *src, intptr_t stride_src, const uint8_t *ref, intptr_t stride_ref); /* xSatdN */
int64_t xmi_satd_hm(int n, const uint8_t *src, uint32_t stride_src, const uint8_t *ref, uint32_t stride_ref); /* xCalcHADs* / xSatd_AMP */
/* xSubDct, src/pixel.cpp:3446 (dst uses `stride`, like the reference) */
int xmi_sub_dct(int16_t *dst, const uint8_t *src, uint32_t stride, const uint8_t *pred, uint32_t stride_pred,
                int width, uint32_t mode);
/* xQuant, src/pixel.cpp:1054; returns sum|level| */
int64_t xmi_quant(int16_t *dst, const int16_t *src, uint32_t stride, uint32_t qp, int width, int height, int slice_type);
/* xDeQuant, src/pixel.cpp:1244 */
int xmi_dequant(int16_t *dst, const int16_t *src, uint32_t stride, uint32_t qp, int width, int height);
/* xIDctAdd, src/pixel.cpp:5938 (pred is read with `stride`, like the reference) */
int xmi_idct_add(uint8_t *dst, uint32_t stride_dst, const int16_t *src, uint32_t stride, const uint8_t *pred,
                 int width, uint32_t mode);
/* xPredInterLumaBlk (src/pixel.cpp:8745) / xPredInterChromaBlk (:9657) / xEncIntraPred (:7338) with host pointers and
 * the reference's argument meaning: ref points at the block's integer origin inside the caller's padded plane, pred rows
 * are MAX_CU_SIZE = 64 (luma) or 32 (chroma) bytes apart, the psTmp argument is not needed. Square blocks only (the
 * reference never asks for others): luma 8..64, chroma 4..32. Only the samples the reference's filters read are touched. */
int xmi_pred_inter_luma(const uint8_t *ref, uint8_t *pred, uint32_t ref_stride, uint32_t size_x, uint32_t size_y, int mvx, int mvy);
int xmi_pred_inter_chroma(const uint8_t *ref, uint8_t *pred, uint32_t ref_stride, uint32_t size_x, uint32_t size_y, int mvx, int mvy);
int xmi_enc_intra_pred(uint8_t *dst, const uint8_t *ref, uint32_t stride, uint32_t mode, uint32_t size, uint32_t is_luma);
/* One 35-mode intra search (the loop of src/encode.cpp:4005-4081) with host pointers: src = the PU's source block
 * (rows src_stride apart, as in the encoder's CTU cache), ref_samples = the 4*size+1 neighbouring samples before
 * xPaddingRef; the other fields of the request as in xmi_intra_search_batch_dev (x0, y0 are ignored). */
int xmi_intra_search(const uint8_t *src, uint32_t src_stride, const uint8_t *ref_samples, const xmi_intra_request *req,
                     xmi_intra_result *res);
/* xSubDctQuantRec, src/encode.cpp:3005 */
int64_t xmi_tu_chain(const uint8_t *pix, const uint8_t *pred, uint8_t *rec, int16_t *coef, int size,
                     int stride, int stride_rec, uint32_t qp, uint32_t mode, int slice_type);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_H */
