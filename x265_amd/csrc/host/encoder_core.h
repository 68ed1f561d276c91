// encoder_core.h - the reference's picture loop (xEncodeFrame, src/encode.cpp:5752-6118) on top of the analysis loop:
// decisions and reconstruction of every CTU (analysis_core), in-loop deblocking, the SAO decision (rdoSaoUnitAll, :1348)
// and offsets, the CABAC slice data with one substream per CTU row (xEncWriteCU :4547, xCompressCU :5594), parameter sets,
// slice header and the picture-hash SEI: one call per picture, the access unit's bytes out.
//
// Everything serial - syntax, arithmetic coding, the SAO parameter search on per-CTU statistics - is host code here; every
// pass over samples is a request to the Evaluator (CU / PU level) or to FrameOps (picture level). libx265_amd.so binds both
// to the GPU; the test harness under oracle/ binds them to the CPU checker.
#pragma once
#include <cstdint>
#include <thread>
#include <vector>

#include "analysis_core.h"
#include "bitstream.h"
#include "syntax.h"

namespace xmi {
namespace encoder {

// layout of xmi_sao_stats / xmi_sao_ctu_param (include/x265_amd.h)
struct SaoStats {
    int32_t count[5][33];
    int32_t offset_org[5][33];
};
struct SaoCtuParam {
    int8_t type_idx;
    uint8_t band_position;
    int8_t offset[4];
    uint8_t reserved[2];
};

// Picture-level sample passes. Planes are packed: width x height luma, width/2 x height/2 chroma.
class FrameOps {
public:
    virtual ~FrameOps() {}
    // In-loop deblocking of the unfiltered reconstruction `rec` - boundary strengths from the decision maps
    // (ff_hevc_deblocking_boundary_strengths, src/Deblock.cpp:369: unit size and luma cbf per 4x4 block, vector per 8x8
    // block), then xDeblockFrame (src/Deblock.cpp:524) - and, on the result, calcSaoStatsCu (src/encode.cpp:364) against the
    // source `org` for every CTU and component: stats[ctu * 3 + component]. The deblocked picture stays with the
    // implementation until sao_apply().
    virtual int deblock_and_stats(int width, int height, int intra_slice, int qp, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                                  const uint8_t *const rec[3], const uint8_t *const org[3], SaoStats *stats) = 0;
    // processSaoUnitAll (src/encode.cpp:1762) on the deblocked picture of the last deblock_and_stats(): params[c] holds
    // component c's parameters per CTU in raster order, or is null when SAO is off for it. The final reconstruction goes
    // to out[].
    virtual int sao_apply(int width, int height, const SaoCtuParam *const params[3], uint8_t *const out[3]) = 0;
};

class Encoder {
public:
    Encoder(int width, int height, int qp, int strong_intra_smoothing);      // coded size, multiples of 64
    ~Encoder();
    Encoder(const Encoder &) = delete;
    Encoder &operator=(const Encoder &) = delete;
    analysis::Analyser &analyser() { return an_; }

    // One picture (xEncodeFrame): src_* are the source planes at coded size (packed), slice_type 1 = P, 2 = I. The access
    // unit - parameter sets in front of an I picture, slice, picture-hash SEI - is written to out (capacity cap); returns its
    // size, or a negative error of the evaluator / frame operations (-100: out smaller than 4096 bytes, -101: P picture without
    // reference, -102: the access unit does not fit, -103: pictures submitted and not collected).
    int encode(const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, analysis::Evaluator &ev, FrameOps &ops, uint8_t *out,
               size_t cap);
    // The same in two steps, for callers that accept the access unit of picture n after they have handed in picture n + 1:
    // submit() does everything up to the slice data and the reference picture of the next call and leaves the picture hash
    // (MD5 of the final reconstruction: a serial pass, 2.6 ms for a 1920x1088 luma plane) to two threads; collect() waits for
    // them, appends the hash unit and hands out the oldest access unit. At most two pictures may be waiting (-103);
    // collect() without one is -104, and with a buffer that is too small -102 (the picture keeps waiting).
    // encode() = submit() + collect().
    int submit(const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, analysis::Evaluator &ev, FrameOps &ops);
    int collect(uint8_t *out, size_t cap);
    int waiting() const { return pending_; }
    // The picture-hash unit behind every slice is the reference's default build (CMake option CHECK_SEI, src/CMakeLists.txt:37,
    // on); off = its build without it: no hashing, no such unit. From the next picture submitted on.
    void picture_hash(bool on) { hash_on_ = on; }

    // the last picture's final reconstruction (after deblocking and SAO), packed planes
    const std::vector<uint8_t> &rec_y() const { return rec_[0]; }
    const std::vector<uint8_t> &rec_u() const { return rec_[1]; }
    const std::vector<uint8_t> &rec_v() const { return rec_[2]; }
    // the last picture's analysis result and SAO parameters ([component][ctu]), for tests
    const analysis::PictureResult &analysis_result() const { return res_; }
    const std::vector<syntax::SaoUnit> &sao_units(int comp) const { return sao_[comp]; }
    const std::vector<SaoStats> &sao_statistics() const { return stats_; }
    int poc() const { return poc_; }
    // wall time of the last picture's stages in nanoseconds: analysis, deblocking + SAO statistics, SAO decision + offsets, writing
    uint64_t stage_ns[4] = { 0, 0, 0, 0 };

private:
    void sao_decide(int slice_type, const SaoStats *stats);

    int w_, h_, rows_, cols_, qp_, strong_;
    analysis::Analyser an_;
    analysis::PictureResult res_;
    std::vector<uint8_t> rec_[3];            // reconstruction of the current picture, packed
    std::vector<uint8_t> ref_[3];            // padded previous reconstruction (PAD_YSIZE 80 / PAD_CSIZE 40)
    bool have_ref_ = false;
    bool have_ctx_ = false;
    bool hash_on_ = true;
    uint8_t last_ctx_[256] = {};             // the coder's context states at the end of the last picture written
    int poc_ = -1;
    std::vector<syntax::SaoUnit> sao_[3];
    bool sao_on_[2] = { true, true };
    std::vector<uint8_t> unit_log2_, unit_cbf_;
    std::vector<SaoStats> stats_;
    std::vector<SaoCtuParam> sao_params_;
    std::vector<uint8_t> sub_;               // slice data under construction
    struct Pending {                         // an access unit that lacks its picture hash
        std::vector<uint8_t> au;             // two bytes the writer may look back over, then the access unit
        bitstream::BitWriter bs;             // where it goes on
        std::thread hash_y, hash_c;
        uint8_t digest[3][16];
        int len = 0;
        bool sei_done = false;
        bool hashed = true;                  // the picture was submitted with the picture hash on
        void join();
    } pend_[2];
    int first_ = 0, pending_ = 0;            // the oldest waiting picture's slot, pictures waiting
};

}  // namespace encoder
}  // namespace xmi
