// analysis_core.h - host side of the P-picture analysis loop: the reference's CU decision for inter pictures
// (xEncodeCUs / xEncDecideCU_Inter / xPreEncRec / xEncInterCTU, src/encode.cpp:5203-5591, 3080-3639, 3023-3078, 3641-3791)
// restated so that every pixel operation of a CU - candidate predictions, distortions, the motion search, the transform
// chains - is ONE request to an evaluator, and the requests of all CTU rows that are ready (the reference's wavefront
// order) are handed over together. This file knows nothing about where the evaluator runs: libx265_amd.so binds it to
// the GPU kernels; the test harness under oracle/ binds it to the CPU checker.
#pragma once
#include <cstdint>
#include <vector>

namespace xmi {
namespace analysis {

struct Mv {
    int16_t x, y;
};

// One square inter CU, everything xEncDecideCU_Inter asks of the pixel primitives for it (src/encode.cpp:3272-3404):
//   - the prediction of every usable merge candidate and its SSD against the source; the first smallest wins (:3272-3289);
//   - transform chain of the source against that prediction (xPreEncRec, :3297), SSD of its reconstruction (:3321);
//   - xMotionSearch (:3379) with bZeroMV set when that candidate is the zero vector (:3341), prediction at the vector
//     found (:3385), transform chain (:3389), SSD of its reconstruction (:3404).
// A 64x64 CU is transformed as four 32x32 units (:3047-3064): sum_*[j*2+i] belongs to the unit at (32 i, 32 j); smaller
// CUs use sum_*[0]. Blocks are size x size, rows `size` apart. The evaluator owns the output blocks; they stay valid until
// its next call.
struct CuEval {
    int x0, y0, size;                 // luma position in the picture, 8..64
    int n_merge;                      // entries of merge_mv
    Mv merge_mv[5];
    uint8_t merge_ok[5];              // 0: candidate left out (its clipped form differs, :3275)
    Mv amvp[2];                       // sMVP[0], sMVP[1] of xFillMvpCand
    Mv mv_min_ipel, mv_max_ipel;      // xCuMvClip box (:3794)
    double lambda;                    // dLambda; the search uses (UInt32)sqrt(lambda) (:2836)
    int qp;
    // results
    int best_merge;                   // index into merge_mv, -1 when no candidate was usable
    uint32_t ssd_skip;                // SSD(source, prediction of best_merge)
    uint32_t ssd_merge;               // SSD(source, reconstruction of best_merge)
    uint32_t sum_merge[4];            // sum |level| per transform unit; the levels of a unit are defined only where it is not zero
    const uint8_t *pred_merge, *rec_merge;      // pred_merge: defined only in the units whose sum_merge is not zero (elsewhere rec_merge is the prediction)
    const int16_t *level_merge;
    Mv mv_me;
    uint32_t ssd_me, sum_me[4];
    const uint8_t *rec_me;
    const int16_t *level_me;
    // An evaluator may return before all results are in place: then merge_pending / search_pending are not null, the merge
    // results (best_merge .. level_merge) resp. the search results (mv_me .. level_me) not yet valid, and
    // Evaluator::finish_merge() / finish_search() bring them. `urgent` (set by the caller) marks the requests whose merge
    // results it reads first: an evaluator that returns early must have those under way to completion at least. The other
    // fields are for the evaluator's own use.
    uint8_t urgent;
    const void *merge_pending, *search_pending;
    uint32_t answer_round;
    int answer_lane;
    const uint32_t *answer_done;
};

// Chroma of one decided CU (xEncInterCTU, src/encode.cpp:3719-3788): prediction of U and V at the CU's vector; unless
// the CU is skipped, their transform chains at the chroma QP. Blocks are size/2 x size/2; a 64x64 CU again as four units.
// For a skipped CU the reconstruction is the prediction and the levels are zero.
struct ChromaEval {
    int x0, y0, size;                 // luma position and size of the CU
    Mv mv;
    int skip;
    int qpc;
    const uint8_t *rec_u, *rec_v;
    const int16_t *level_u, *level_v;
    uint32_t sum_u[4], sum_v[4];      // (levels of a unit are defined only where its sum is not zero)
};

// One square intra PU of an I picture, everything xEncDecideCU_Intra asks of the pixel primitives for it
// (src/encode.cpp:4005-4115): padding and smoothing of the neighbouring samples (xPaddingRef :3843, xFilterRef :3873), the
// 35-mode SATD search (cost = SATD + {1,2,3} sqrt(lambda) by most-probable-mode rank, first smallest), the prediction of
// the winner, its transform chain (DST for 4x4, I-slice rounding) and the SSD of the reconstruction. The neighbouring
// samples are what the reference gathers (:4010-4015: left column from the bottom up, top-left, top row and top-right,
// 4 size + 1 bytes) with the availability bits of :3965-3974; samples of unavailable segments are arbitrary.
struct IntraQuadEval;

// most probable modes of a luma PU from the modes to its left and above it (src/encode.cpp:3980-4003)
inline void most_probable_modes(int left, int top, uint8_t mpm[3])
{
    if (left == top) {
        if (left > 1) mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)(((left + 29) % 32) + 2), mpm[2] = (uint8_t)(((left - 1) % 32) + 2);
        else mpm[0] = 0, mpm[1] = 1, mpm[2] = 26;
    } else {
        mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)top;
        mpm[2] = (left && top) ? 0 : (left + top < 2 ? 26 : 1);
    }
}

struct IntraEval {
    int x0, y0, size;                 // luma position in the picture, 4..32
    uint8_t valids;                   // bit 0 below-left, 1 left, 2 top-left, 3 top, 4 top-right
    uint8_t strong;                   // bStrongIntraSmoothing
    uint8_t mpm[3];
    uint8_t left_mode = 1, top_mode = 1;      // what mpm was made from (DC where there is no neighbour)
    uint8_t ref[132];
    double sqrt_lambda;
    int qp;
    // A request of size 8 may stand for the 8x8 block to the right of a block that is decided in the same call: its left
    // column and the mode to its left are then those of `after_pu` (after_kind 1: the block to the left as one 8x8 PU) or of
    // `after_quad` (2: as four 4x4 PUs), which the evaluator puts in place - ref[8..15], left_mode, mpm - before it evaluates
    // the request (take_left_block). The caller asks for both and uses the one whose premise came true.
    uint8_t after_kind = 0;
    const IntraEval *after_pu = nullptr;
    const IntraQuadEval *after_quad = nullptr;
    // results; the blocks are size x size, rows `size` apart, valid until the next call of the lane
    int best_mode;
    uint32_t ssd, sum;
    const uint8_t *rec;
    const int16_t *level;
};

// Chroma of one intra PU (xEncodeChromaCU, src/encode.cpp:4306-4411): padding of the U and V neighbours, the five-candidate
// SATD search (planar, vertical, horizontal, DC, the luma mode; a duplicate of the luma mode becomes mode 34), prediction of
// the winner, both transform chains at the chroma QP.
struct IntraChromaEval {
    int x0, y0, size;                 // position and size in the chroma planes, 4..16
    uint8_t valids, luma_mode;
    uint8_t ref_u[68], ref_v[68];
    int qpc;
    int best_idx;                     // nBestModeC, index into the candidate list
    uint32_t sum_u, sum_v;
    const uint8_t *rec_u, *rec_v;
    const int16_t *level_u, *level_v;
};

// The four 4x4 PUs of an 8x8 luma block, evaluated one after the other in coding order, each as an IntraEval of size 4. A 4x4
// PU is always kept (src/encode.cpp:4133: bReplace at the minimum size), so what a PU sees of the PUs before it - their
// reconstruction as neighbouring samples, their modes as most probable modes - depends on no cost comparison and the four
// evaluations can run back to back where the samples are. ref: the 33 neighbouring samples of the 8x8 block (as an 8x8 PU's),
// which hold every sample the four PUs need from outside the block.
struct IntraQuadEval {
    int x0, y0;                       // luma position of the 8x8 block
    uint8_t ref[36];
    uint8_t valids[4];                // availability bits of each PU (PUs before it in the block count as available)
    uint8_t left_mode[2], top_mode[2];   // effective neighbour modes from outside the block (DC substituted as :3977-3978
                                         // does): left of PU 0 / PU 2, above PU 0 / PU 1
    uint8_t strong;
    double sqrt_lambda;
    int qp;
    uint8_t after_kind = 0;           // as IntraEval::after_kind: ref[8..15] and left_mode[] come from the block to the left
    const IntraEval *after_pu = nullptr;
    const IntraQuadEval *after_quad = nullptr;
    // results; rec and level are the 8x8 block, rows 8 apart
    int best_mode[4];
    uint32_t ssd[4], sum[4];
    const uint8_t *rec;
    const int16_t *level;
};

// What an evaluator does to a request that stands after a block of the same call, once that block's results are there: the
// block's right column becomes the request's left column (ref[15 - row], rows 0..7), the modes of the block's right-hand
// units the modes to the left.
inline void left_block_of(uint8_t after_kind, const IntraEval *pu, const IntraQuadEval *quad, uint8_t col[8], uint8_t mode[2])
{
    const uint8_t *rec = after_kind == 1 ? pu->rec : quad->rec;      // 8x8, rows 8 apart
    for (int r = 0; r < 8; r++) col[r] = rec[r * 8 + 7];
    mode[0] = (uint8_t)(after_kind == 1 ? pu->best_mode : quad->best_mode[1]);
    mode[1] = (uint8_t)(after_kind == 1 ? pu->best_mode : quad->best_mode[3]);
}
inline void take_left_block(IntraEval &q)
{
    uint8_t col[8], mode[2];
    left_block_of(q.after_kind, q.after_pu, q.after_quad, col, mode);
    for (int r = 0; r < 8; r++) q.ref[15 - r] = col[r];
    q.left_mode = mode[0];
    most_probable_modes(q.left_mode, q.top_mode, q.mpm);
}
inline void take_left_block(IntraQuadEval &q)
{
    uint8_t col[8], mode[2];
    left_block_of(q.after_kind, q.after_pu, q.after_quad, col, mode);
    for (int r = 0; r < 8; r++) q.ref[15 - r] = col[r];
    q.left_mode[0] = mode[0], q.left_mode[1] = mode[1];
}

// The chroma of four neighbouring 8x8 luma blocks (an 8x8 area of the chroma planes): four IntraChromaEval of size 4 back
// to back; nothing in xEncodeChromaCU depends on a cost comparison. ref_*: the 33 neighbouring samples of the area.
struct IntraChromaQuadEval {
    int x0, y0;                       // position of the area in the chroma planes
    uint8_t ref_u[36], ref_v[36];
    uint8_t valids[4], luma_mode[4];
    int qpc;
    int best_idx[4];
    uint32_t sum_u[4], sum_v[4];
    const uint8_t *rec_u, *rec_v;     // 8x8 areas, rows 8 apart
    const int16_t *level_u, *level_v;
};

struct IntraBatch {
    IntraEval *const *pu = nullptr;
    int npu = 0;
    IntraChromaEval *const *chroma = nullptr;
    int nchroma = 0;
    IntraQuadEval *const *quad = nullptr;
    int nquad = 0;
    IntraChromaQuadEval *const *cquad = nullptr;
    int ncquad = 0;
};

class Evaluator {
public:
    virtual ~Evaluator() {}
    // Binds the evaluator to the pictures of the next call of Analyser::p_picture / i_picture, for owners that do not bind it
    // by other means: packed source planes (width x height, width/2 x height/2) and, for a P picture, the first byte of
    // each padded plane of the previous reconstruction (`pad` luma samples of border, pad/2 chroma; null for an I picture).
    virtual int bind_pictures(const uint8_t *const src[3], const uint8_t *const ref[3], int pad, int qp)
    {
        (void)src, (void)ref, (void)pad, (void)qp;
        return -1;
    }
    // Context states the entropy coder ended the previous picture with (256 bytes), handed over before bind_pictures(): what an
    // evaluator that prices levels (the level decision that may take the quantiser's place) builds its bit-cost tables from.
    // Null before the first picture.
    virtual void coder_contexts(const uint8_t *ctx) { (void)ctx; }
    // the requests of an I picture; same calling rules as evaluate()
    virtual int evaluate_intra(int lane, const IntraBatch &batch)
    {
        (void)lane, (void)batch;
        return -1;
    }
    // Fills the result fields of every request; returns 0, or an error code that ends the picture. Calls with different
    // `lane` values (0 .. Analyser::lanes - 1) come from different threads and may overlap; calls of one lane never do,
    // and the blocks a call hands out stay valid until the next call of the same lane.
    virtual int evaluate(int lane, CuEval *const *cu, int ncu, ChromaEval *const *chroma, int nchroma) = 0;
    // Complete a request whose merge_pending / search_pending is set (see CuEval); called from the thread of the CU's row,
    // before the lane that answered is called again. Return 0 or an error code that ends the picture.
    virtual int finish_merge(CuEval &q)
    {
        q.merge_pending = nullptr;
        return 0;
    }
    virtual int finish_search(CuEval &q)
    {
        q.search_pending = nullptr;
        return 0;
    }
};

// What a picture leaves behind, picture-level arrays (CTU caches of the reference laid side by side):
//   info0 (ausModeYInfo0) and info1 (aucModeYInfo1) per 4x4 block, width/4 per row; mv, mvd, mvp_idx per 8x8 block,
//   width/8 per row; levels and reconstruction per sample; per CTU the RD cost xEncDecideCU_Inter returned at depth 0
//   and the estimation contexts before and after the CTU (256 bytes each, cabacRdo[y][0] / cabacRdoLast[y]).
struct PictureResult {
    std::vector<uint16_t> info0;
    std::vector<uint8_t> info1;
    std::vector<uint8_t> mode;        // aucModeY per 4x4 block: the intra luma mode in I pictures, 0 (MODE_VALID) in P pictures
    std::vector<Mv> mv, mvd;
    std::vector<uint8_t> mvp_idx;
    std::vector<int16_t> coef_y, coef_u, coef_v;
    std::vector<uint8_t> rec_y, rec_u, rec_v;
    std::vector<uint32_t> rd_cost;
    std::vector<uint8_t> ctx_in, ctx_out;
    uint64_t evaluator_calls = 0, cu_requests = 0, chroma_requests = 0;
    uint64_t row_busy_ns = 0, row_requests = 0;      // host time of the row threads between their requests, and how many they made
};

class Analyser {
public:
    Analyser(int width, int height);          // coded size, multiples of 64
    int width() const { return w_; }
    int height() const { return h_; }
    int rows() const { return rows_; }
    int cols() const { return cols_; }

    // State the reference carries from picture to picture and never resets: the FAST_MD cost statistics per CTU and depth
    // (dSumOfRdCosts, nRdCounts, dAverageRdCosts, src/encode.cpp:3556-3568) and the fractional bits of every row's rate
    // counter (rdoParam[y].fracBits, :189).
    std::vector<double> sum_cost, avg_cost;   // [row][col][5]
    std::vector<int32_t> count;
    std::vector<uint64_t> frac_bits;          // [row]

    // Coordinators, each with evaluator resources of its own (lane index of Evaluator::evaluate). All of them serve all rows:
    // with more than one, the requests of rows that were still busy when one call left go out in the next coordinator's call
    // while the first is under way. Results do not depend on it.
    int lanes = 1;
    // (polling configurations, lanes > 1) how long a coordinator waits for the rows that are still busy on the host before it
    // goes without them, microseconds
#ifndef XMI_GRACE_US
#define XMI_GRACE_US 6
#endif
    int grace_us = XMI_GRACE_US;
    // Rows waiting for results, and the coordinators waiting for their rows, poll for this many microseconds before they
    // block: a round trip is shorter than the time a sleeping thread needs to come back. 0 = block at once (the choice when
    // many analysers share the host's cores).
    int spin_us = 0;
    // Threads that run the CTU rows (each row a fiber of one of them); 0 = as many as the host has cores for, at most one per
    // row. Results do not depend on it.
    int workers = 0;
    // 1: a CU's request also asks for its first child, that child's first child, ... (same corner, same neighbours). 2: it asks
    // for every CU that may still be decided in the CTU, under the guess that all of them end up with the vector of the
    // asking CU's first merge candidate; a CU whose neighbours turn out as guessed finds its answer waiting. Fewer round
    // trips, more pixel work, some of it for CUs the decision never comes to. 0: one request per CU. Results do not
    // depend on it.
    int speculate = 2;

    // One P picture: decisions and reconstruction of every CTU, rows advancing as a wavefront (a CTU starts when its
    // left neighbour and the CTU above and to the right are done). The evaluator is bound to the picture pair (source,
    // padded previous reconstruction) by its owner before the call. Returns 0 or the evaluator's error.
    int p_picture(int qp, Evaluator &ev, PictureResult &out);
    // One I picture (xEncodeCUs with xEncDecideCU_Intra, src/encode.cpp:3906-4215, and xEncodeChromaCU, :4234-4418): same
    // wavefront, same outputs; info0 / info1 in the I-slice layout (prediction index, chroma mode and cbf; availability bits,
    // luma cbf, size), mv / mvd / mvp_idx zero, rd_cost the depth-0 cost truncated to 32 bits.
    int i_picture(int qp, int strong_intra_smoothing, Evaluator &ev, PictureResult &out);

private:
    int picture(int qp, bool intra, int strong, Evaluator &ev, PictureResult &out);
    int w_, h_, rows_, cols_;
};

// the tables the rate estimate walks; exposed for the tests that pin them against the reference's
const uint16_t *diag_scan(int log2_size);               // coefficient scan of a 4..32 block, positions y * size + x
const uint8_t *next_state_table();                       // [state][bin] -> state, 128 x 2
const int32_t *entropy_bits_table();                     // [state ^ bin] -> cost of the bin in 1/32768 bit (g_entropyBits, src/encode.cpp:131)
void init_contexts(int slice_type, int qp, uint8_t ctx[256]);   // xCabacInit (src/bitstream.cpp:588), slice 1 = P, 2 = I
int chroma_qp(int qp);                                   // xg_aucChromaScale (src/table.cpp:160)

}  // namespace analysis
}  // namespace xmi
