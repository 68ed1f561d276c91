// encoder_core.cpp - see encoder_core.h. Host logic only: syntax, arithmetic coding, parameter search on statistics.
#include "encoder_core.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "bitstream.h"

namespace xmi {
namespace encoder {

using namespace xmi::syntax;
using analysis::PictureResult;
using bitstream::BitWriter;
using bitstream::Cabac;

namespace {

constexpr int kPad = 80, kPadC = 40;      // PAD_YSIZE / PAD_CSIZE (inc/config.h:52-53)

inline int interleave(unsigned v) { return (int)((v & 1) | ((v >> 1) & 2) | ((v >> 2) & 4) | ((v >> 3) & 8)); }   // getInterleaveBits (src/encode.cpp:4220)

// The rate counter of the SAO decision: xGetWrittenCmdRate (src/encode.cpp:187) on the last CTU row's estimation
// contexts, whose next-state table exists only when that row is row 0 (src/encode.cpp:5052).
struct SaoRate {
    uint64_t frac = 0;
    bool transitions = false;
    uint8_t *ctx = nullptr;
    const int32_t *bits_table = nullptr;
    const uint8_t *next = nullptr;
    void bin(unsigned b, int c)
    {
        const uint8_t s = ctx[c];
        frac += (uint64_t)bits_table[s ^ b];
        ctx[c] = transitions ? next[s * 2 + b] : 0;
    }
    void ep(unsigned, int n) { frac += 32768ull * (unsigned)n; }
    void begin() { frac &= 32767; }
    uint32_t bits() const { return (uint32_t)(frac >> 15); }
};

// a sink that drops everything: where the reference builds a command list only to throw it away, the side effects of the
// element coders on their arguments still happen
struct NullSink {
    void bin(unsigned, int) {}
    void ep(unsigned, int) {}
};

inline int64_t sao_dist(int64_t count, int64_t offset, int64_t offset_org) { return count * offset * offset - offset_org * offset * 2; }   // estSaoDist, shift 0

}  // namespace

Encoder::Encoder(int width, int height, int qp, int strong_intra_smoothing)
    : w_(width), h_(height), rows_(height / 64), cols_(width / 64), qp_(qp), strong_(strong_intra_smoothing), an_(width, height)
{
    const size_t ny = (size_t)w_ * h_;
    rec_[0].assign(ny, 0), rec_[1].assign(ny / 4, 0), rec_[2].assign(ny / 4, 0);
    ref_[0].assign((size_t)(w_ + 2 * kPad) * (h_ + 2 * kPad), 0);
    ref_[1].assign((size_t)(w_ / 2 + 2 * kPadC) * (h_ / 2 + 2 * kPadC), 0);
    ref_[2] = ref_[1];
    for (int c = 0; c < 3; c++) sao_[c].assign((size_t)rows_ * cols_, SaoUnit());
    unit_log2_.assign(ny / 16, 0), unit_cbf_.assign(ny / 16, 0);
    stats_.assign((size_t)rows_ * cols_ * 3, SaoStats());
}

Encoder::~Encoder()
{
    for (Pending &q : pend_) q.join();
}

void Encoder::Pending::join()
{
    if (hash_y.joinable()) hash_y.join();
    if (hash_c.joinable()) hash_c.join();
}

namespace {

// The syntax of one CTU (xEncWriteCU, src/encode.cpp:4547-4924) from the picture-level decision arrays. The reference reads
// CTU-local caches with borders; the accessors below give what those borders hold.
struct CtuView {
    const PictureResult &r;
    int x, y, cols, w4, w8, w;
    int info1(int rr, int cc) const
    {
        if (rr < 0) return y > 0 ? r.info1[(size_t)(y * 16 - 1) * w4 + x * 16 + cc] : 0xff;     // top line buffer, 0xff before the first row
        if (cc < 0) return x > 0 ? r.info1[(size_t)(y * 16 + rr) * w4 + x * 16 - 1] : 0;
        return r.info1[(size_t)(y * 16 + rr) * w4 + x * 16 + cc];
    }
    int mode(int rr, int cc) const
    {
        if (rr < 0) {
            if (y == 0) return 255;
            if (cc < 0) return x > 0 ? 0 : 255;
            if (cc > 15) return x < cols - 1 ? 0 : 255;
            return 0;
        }
        if (cc < 0) return x > 0 ? r.mode[(size_t)(y * 16 + rr) * w4 + x * 16 - 1] : 255;
        return r.mode[(size_t)(y * 16 + rr) * w4 + x * 16 + cc];
    }
    int info0(int rr, int cc) const { return r.info0[(size_t)(y * 16 + rr) * w4 + x * 16 + cc]; }
    const int16_t *coef_y(int rr, int cc) const { return &r.coef_y[(size_t)(y * 64 + rr * 4) * w + x * 64 + cc * 4]; }
    const int16_t *coef_c(const std::vector<int16_t> &p, int rr, int cc) const { return &p[(size_t)(y * 32 + rr * 2) * (w / 2) + x * 32 + cc * 2]; }
};

template <class S>
void write_ctu(S &s, const CtuView &v, bool intra_slice)
{
    static const uint8_t chroma_cand[5] = { 0, 26, 10, 1, 36 };
    const int inter0 = (v.info0(0, 0) >> 7) & 1;
    for (unsigned idx = 0; idx != 256;) {
        const int dx = interleave(idx), dy = interleave(idx >> 1);
        const int f1 = v.info1(dy, dx), valids = f1 & 0x1f;
        int split = f1 >> 6;
        int split_left = (valids & 2) ? v.info1(dy, dx - 1) >> 6 : 7, split_top = (valids & 8) ? v.info1(dy - 1, dx) >> 6 : 7;
        const int b64 = (f1 >> 4) & 1 & inter0;
        if (b64) split = 4;
        if ((v.info1(dy, dx - 1) >> 4) & 1 & inter0) split_left = 4;
        if ((v.info1(dy - 1, dx) >> 4) & 1 & inter0) split_top = 4;
        const int log2 = 2 + split, size = 1 << log2, b4 = size / 4;
        int size_c = size >> 1;
        int mode_y = v.mode(dy, dx);
        const int i0 = v.info0(dy, dx);
        unsigned mode_idx_c = (unsigned)(i0 >> 13) & 7;
        int cbf_y = (f1 >> 5) & 1, cbf_u = (i0 >> 8) & 1, cbf_v = (i0 >> 9) & 1;
        const int is_inter = (i0 >> 7) & 1, merge_idx = i0 & 0x3f, skipped = f1 & 1;
        const int skipped_left = (v.info1(dy, dx - 1) & 1) && v.mode(dy + b4 - 1, dx - 1) != 255;
        const int skipped_above = (v.info1(dy - 1, dx) & 1) && v.mode(dy - 1, dx + b4 - 1) != 255;
        const size_t at8 = (size_t)(v.y * 8 + dy / 2) * v.w8 + v.x * 8 + dx / 2;
        const int root_cbf_c = cbf_u | cbf_v;
        int root_cbf = cbf_y | root_cbf_c;
        const int16_t *coef_u = v.coef_c(v.r.coef_u, dy, dx), *coef_v = v.coef_c(v.r.coef_v, dy, dx);
        const int cstride = v.w / 2;

        if (split <= 4 && idx == 0) s.bin(split < 4, kCtxSplit + (split_left < 4) + (split_top < 4));
        if (split <= 3 && (idx & 63) == 0) s.bin(split < 3, kCtxSplit + (split_left < 3) + (split_top < 3));
        if (split <= 2 && (idx & 15) == 0) s.bin(split < 2, kCtxSplit + (split_left < 2) + (split_top < 2));
        bool done = false;
        if (!intra_slice) {
            s.bin((unsigned)skipped, kCtxSkip + skipped_left + skipped_above);
            if (skipped) {
                code_merge_idx(s, (unsigned)(merge_idx - 1));
                done = true;
            } else {
                s.bin(0, kCtxPredMode);
            }
        }
        if (!done && (idx & 3) == 0) {
            if (intra_slice) {
                if (split <= 1) s.bin(split != 0, kCtxPartSize);
                unsigned pi[4] = { (unsigned)(i0 & 0x3f), 0, 0, 0 };
                if (split == 0) {      // four 4x4 PUs
                    pi[1] = (unsigned)v.info0(dy, dx + 1) & 0x3f, pi[2] = (unsigned)v.info0(dy + 1, dx) & 0x3f, pi[3] = (unsigned)v.info0(dy + 1, dx + 1) & 0x3f;
                }
                code_intra_dir_luma(s, split == 0 ? 4 : 1, pi);
                code_intra_dir_chroma(s, mode_idx_c);
            } else {
                s.bin(1, kCtxPartSize);
                s.bin(merge_idx != 0, kCtxMergeFlag);
                if (merge_idx) {
                    code_merge_idx(s, (unsigned)(merge_idx - 1));
                } else {
                    code_mvd(s, v.r.mvd[at8].x, v.r.mvd[at8].y);
                    s.bin(v.r.mvp_idx[at8] ? 1 : 0, kCtxMvpIdx);
                }
                if (b64) {
                    cbf_y = cbf_u = cbf_v = 0;
                    for (int j = 0; j < 2; j++)
                        for (int i = 0; i < 2; i++) {
                            cbf_y |= (v.info1(dy + j * 8, dx + i * 8) >> 5) & 1;
                            cbf_u |= (v.info0(dy + j * 8, dx + i * 8) >> 8) & 1;
                            cbf_v |= (v.info0(dy + j * 8, dx + i * 8) >> 9) & 1;
                        }
                    root_cbf = cbf_y | cbf_u | cbf_v;
                    if (!merge_idx) s.bin((unsigned)root_cbf, kCtxQtRootCbf);
                } else {
                    if (!merge_idx) s.bin((unsigned)root_cbf, kCtxQtRootCbf);
                    if (root_cbf) s.bin(0, kCtxTransSubdiv + 5 - log2);
                }
            }
            if (!is_inter || (root_cbf && !b64)) {
                s.bin((unsigned)cbf_u, kCtxQtCbf + kNumQtCbf);
                s.bin((unsigned)cbf_v, kCtxQtCbf + kNumQtCbf);
            }
        }
        if (!done) {
            if (b64) {
                if (root_cbf) {
                    s.bin((unsigned)cbf_u, kCtxQtCbf + kNumQtCbf);
                    s.bin((unsigned)cbf_v, kCtxQtCbf + kNumQtCbf);
                    const int root_u = cbf_u, root_v = cbf_v;
                    for (int j = 0; j < 2; j++)
                        for (int i = 0; i < 2; i++) {
                            const int qy = (v.info1(dy + j * 8, dx + i * 8) >> 5) & 1;
                            const int qu = (v.info0(dy + j * 8, dx + i * 8) >> 8) & 1, qv = (v.info0(dy + j * 8, dx + i * 8) >> 9) & 1;
                            if (root_u) s.bin((unsigned)qu, kCtxQtCbf + kNumQtCbf + 1);
                            if (root_v) s.bin((unsigned)qv, kCtxQtCbf + kNumQtCbf + 1);
                            s.bin((unsigned)qy, kCtxQtCbf);
                            if (qy) code_coefficients(s, v.coef_y(dy + j * 8, dx + i * 8), v.w, 32, true, kScanDiag);
                            if (qu) code_coefficients(s, v.coef_c(v.r.coef_u, dy + j * 8, dx + i * 8), cstride, 16, false, kScanDiag);
                            if (qv) code_coefficients(s, v.coef_c(v.r.coef_v, dy + j * 8, dx + i * 8), cstride, 16, false, kScanDiag);
                        }
                }
            } else {
                if (!is_inter || root_cbf_c) s.bin((unsigned)cbf_y, kCtxQtCbf + (size != 4 ? 1 : 0));
                if (cbf_y) code_coefficients(s, v.coef_y(dy, dx), v.w, size, true, coef_scan_idx(size, !is_inter, true, mode_y));
                if (split != 0 || (idx & 3) == 3) {
                    if (split == 0) {      // the last of four 4x4 PUs carries the chroma of their 8x8: signalled with the first one's data
                        const int lt = v.info0(dy - 1, dx - 1);
                        mode_idx_c = (unsigned)(lt >> 13) & 7;
                        mode_y = v.mode(dy - 1, dx - 1);
                        cbf_u = (lt >> 8) & 1, cbf_v = (lt >> 9) & 1;
                        coef_u = v.coef_c(v.r.coef_u, dy - 1, dx - 1), coef_v = v.coef_c(v.r.coef_v, dy - 1, dx - 1);
                        size_c = 4;
                    }
                    int real_c = mode_y;
                    if (mode_idx_c != 4) {
                        real_c = chroma_cand[mode_idx_c];
                        if (real_c == mode_y) real_c = 34;
                    }
                    if (cbf_u) code_coefficients(s, coef_u, cstride, size_c, false, coef_scan_idx(size_c, !is_inter, false, real_c));
                    if (cbf_v) code_coefficients(s, coef_v, cstride, size_c, false, coef_scan_idx(size_c, !is_inter, false, real_c));
                }
            }
        }
        idx += 1u << (((2 + split) << 1) - 4);
    }
}

// The transform units the reference reports to ff_hevc_deblocking_boundary_strengths while it writes a CTU
// (src/encode.cpp:4873-4917): unit size and luma cbf per 4x4 block. A 64x64 CU with residual is four 32x32 units.
void ctu_units(const CtuView &v, uint8_t *unit_log2, uint8_t *unit_cbf)
{
    const int inter0 = (v.info0(0, 0) >> 7) & 1;
    for (unsigned idx = 0; idx != 256;) {
        const int dx = interleave(idx), dy = interleave(idx >> 1);
        const int f1 = v.info1(dy, dx);
        int split = f1 >> 6;
        const int b64 = (f1 >> 4) & 1 & inter0;
        if (b64) split = 4;
        int log2 = 2 + split;
        if (b64) {
            int root = 0;
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++)
                    root |= ((v.info1(dy + j * 8, dx + i * 8) >> 5) & 1) | ((v.info0(dy + j * 8, dx + i * 8) >> 8) & 3);
            if (root && !(f1 & 1)) log2 = 5;
        }
        const int b4 = 1 << (2 + split - 2);
        for (int rr = 0; rr < b4; rr++)
            for (int cc = 0; cc < b4; cc++) {
                const size_t at = (size_t)(v.y * 16 + dy + rr) * v.w4 + v.x * 16 + dx + cc;
                unit_log2[at] = (uint8_t)log2;
                unit_cbf[at] = (uint8_t)((v.info1(dy + rr, dx + cc) >> 5) & 1);
            }
        idx += 1u << (((2 + split) << 1) - 4);
    }
}

// xFramePadding (src/encode.cpp:2043): border replication of a plane held inside a padded buffer
void pad_plane(uint8_t *buf, int width, int height, int pad)
{
    const int stride = width + 2 * pad;
    uint8_t *p0 = buf + (size_t)pad * stride + pad;
    for (int i = 0; i < height; i++) {
        uint8_t *row = p0 + (size_t)i * stride;
        std::memset(row - pad, row[0], (size_t)pad);
        std::memset(row + width, row[width - 1], (size_t)pad);
    }
    uint8_t *first = p0 - pad, *last = first + (size_t)(height - 1) * stride;
    for (int i = 0; i < pad; i++) {
        std::memcpy(first - (size_t)(i + 1) * stride, first, (size_t)stride);
        std::memcpy(last + (size_t)(i + 1) * stride, last, (size_t)stride);
    }
}

}  // namespace

// rdoSaoUnitAll (src/encode.cpp:1348-1522) with estSaoTypeDist (:998), saoComponentParamDist (:1045) and sao2ChromaParamDist
// (:1176): per CTU in raster order the best of "new parameters" (per component the cheapest of off / four edge classes / band
// offset, offsets refined by estIterOffset :959), "merge left" and "merge up", by distortion estimate / lambda + bits.
void Encoder::sao_decide(int slice_type, const SaoStats *stats)
{
    const int qpc = analysis::chroma_qp(qp_);
    double luma_lambda = 0.578 * std::pow(2.0, (qp_ - 12) / 3.0);
    if (slice_type == 2) luma_lambda *= 0.3;
    const double chroma_lambda = luma_lambda * std::pow(2.0, (qpc - qp_) / 3.0);

    // the estimate runs on the rate counter and contexts the analysis of the last CTU row ended with
    uint8_t ctx[256];
    std::memcpy(ctx, &res_.ctx_out[((size_t)rows_ * cols_ - 1) * 256], 256);
    SaoRate rate;
    rate.frac = an_.frac_bits[rows_ - 1];
    rate.transitions = rows_ == 1;
    rate.ctx = ctx;
    rate.bits_table = analysis::entropy_bits_table();
    rate.next = analysis::next_state_table();
    NullSink null;

    int64_t count[3][5][33], org[3][5][33], off[3][5][33];
    const int n_class[5] = { 4, 4, 4, 4, 4 };
    int no_sao[2] = { 0, 0 };

    // estSaoTypeDist: offsets of one type from the statistics; band offsets also fill the per-band tables
    auto type_dist = [&](int comp, int type, double lambda, int *dist_bo, double *cost_bo) -> int64_t {
        int64_t est = 0;
        const int classes = type < 4 ? 4 : 32;
        for (int k = 1; k <= classes; k++) {
            if (type == 4) dist_bo[k - 1] = 0, cost_bo[k - 1] = lambda;
            if (count[comp][type][k]) {
                const double q = (double)org[comp][type][k] / (double)count[comp][type][k];
                int64_t o = (int64_t)(double)(q >= 0 ? (int)(q + 0.5) : (int)(q - 0.5));
                o = o < -7 ? -7 : o > 7 ? 7 : o;
                if (type < 4) {
                    if (o < 0 && k < 3) o = 0;
                    if (o > 0 && k >= 3) o = 0;
                }
                // estIterOffset: from the rounded offset towards zero, the cheapest of distortion + lambda * bits
                int64_t it = o, out = 0;
                double min_cost = lambda;
                while (it != 0) {
                    int64_t bits = type == 4 ? std::llabs(it) + 2 : std::llabs(it) + 1;
                    if (std::llabs(it) == 7) bits--;
                    const int64_t dist = sao_dist(count[comp][type][k], it, org[comp][type][k]);
                    const double cost = (double)dist + lambda * (double)bits;
                    if (cost < min_cost) {
                        min_cost = cost;
                        out = it;
                        if (type == 4) dist_bo[k - 1] = (int)dist, cost_bo[k - 1] = cost;
                    }
                    it = it > 0 ? it - 1 : it + 1;
                }
                off[comp][type][k] = out;
            } else {
                org[comp][type][k] = 0;
                off[comp][type][k] = 0;
            }
            if (type != 4) est += sao_dist(count[comp][type][k], off[comp][type][k], org[comp][type][k]);
        }
        return est;
    };
    auto best_band = [&](const double *cost_bo) {
        double best_cost = 1.7e+308;
        int best = 0;
        for (int i = 0; i < 32 - 4 + 1; i++) {
            double c = 0.0;
            for (int j = i; j < i + 4; j++) c += cost_bo[j];
            if (c < best_cost) best_cost = c, best = i;
        }
        return best;
    };
    auto candidate = [&](int comp, int type, int band) {
        SaoUnit u;
        u.length = n_class[type], u.type_idx = type, u.sub_type = type == 4 ? band : 0;
        for (int k = 0; k < u.length; k++) u.offset[k] = (int)off[comp][type][k + u.sub_type + 1];
        return u;
    };
    auto merge_dist = [&](int comp, const SaoUnit &nb) -> int64_t {
        int64_t est = 0;
        if (nb.type_idx >= 0) {
            const int band = nb.type_idx == 4 ? nb.sub_type : 0;
            for (int k = 0; k < n_class[nb.type_idx]; k++)
                est += sao_dist(count[comp][nb.type_idx][k + band + 1], nb.offset[k], org[comp][nb.type_idx][k + band + 1]);
        }
        return est;
    };

    for (int cy = 0; cy < rows_; cy++)
        for (int cx = 0; cx < cols_; cx++) {
            const int addr = cy * cols_ + cx, addr_up = cy > 0 ? addr - cols_ : -1, addr_left = cx > 0 ? addr - 1 : -1;
            const bool allow_left = cx > 0, allow_up = cy > 0;
            double dist[3] = { 0, 0, 0 };
            SaoUnit merge[3][2];
            for (int comp = 0; comp < 3; comp++) {
                const SaoStats &st = stats[(size_t)addr * 3 + comp];
                for (int t = 0; t < 5; t++)
                    for (int k = 0; k < 33; k++) {
                        off[comp][t][k] = 0;
                        const bool on = comp == 0 ? sao_on_[0] : sao_on_[1];
                        count[comp][t][k] = on ? st.count[t][k] : 0;
                        org[comp][t][k] = on ? st.offset_org[t][k] : 0;
                    }
                sao_[comp][addr] = SaoUnit();
            }
            int dist_bo[33];
            double cost_bo[33];
            // ---- luma: saoComponentParamDist ----
            {
                const double lambda = luma_lambda;
                SaoUnit rdo, &best = sao_[0][addr];
                rate.begin();
                code_sao_offset(rate, rdo, 0);
                double best_cost = rate.bits() * lambda;
                best = rdo;
                int64_t best_dist = 0;
                for (int type = 0; type < 5; type++) {
                    int64_t est = type_dist(0, type, lambda, dist_bo, cost_bo);
                    int band = 0;
                    if (type == 4) {
                        band = best_band(cost_bo);
                        est = 0;
                        for (int k = band; k < band + 4; k++) est += dist_bo[k];
                    }
                    rdo = candidate(0, type, band);
                    rate.begin();
                    code_sao_offset(rate, rdo, 0);
                    const double cost = (double)est + lambda * (double)rate.bits();
                    if (cost < best_cost) best_cost = cost, best = rdo, best_dist = est;
                }
                dist[0] += (double)best_dist / lambda;
                for (int nb = 0; nb < 2; nb++) {
                    const int a = nb == 0 ? (allow_left ? addr_left : -1) : (allow_up ? addr_up : -1);
                    if (a < 0) continue;
                    const SaoUnit &n = sao_[0][a];
                    merge[0][nb] = n;
                    merge[0][nb].merge_up = nb, merge[0][nb].merge_left = !nb;
                    dist[nb + 1] += (double)merge_dist(0, n) / lambda;
                }
            }
            // ---- chroma, both components together: sao2ChromaParamDist ----
            {
                const double lambda = chroma_lambda;
                SaoUnit rdo[2], *best[2] = { &sao_[1][addr], &sao_[2][addr] };
                rate.begin();
                code_sao_offset(rate, rdo[0], 1);
                code_sao_offset(rate, rdo[1], 2);
                double best_cost = rate.bits() * lambda;
                *best[0] = rdo[0], *best[1] = rdo[1];
                int64_t best_dist = 0;
                for (int type = 0; type < 5; type++) {
                    int64_t est[2];
                    int band[2] = { 0, 0 };
                    for (int c = 0; c < 2; c++) {
                        est[c] = type_dist(c + 1, type, lambda, dist_bo, cost_bo);
                        if (type == 4) {
                            band[c] = best_band(cost_bo);
                            est[c] = 0;
                            for (int k = band[c]; k < band[c] + 4; k++) est[c] += dist_bo[k];
                        }
                    }
                    rate.begin();
                    for (int c = 0; c < 2; c++) {
                        rdo[c] = candidate(c + 1, type, band[c]);
                        code_sao_offset(rate, rdo[c], c + 1);
                    }
                    const double cost = (double)(est[0] + est[1]) + lambda * (double)rate.bits();
                    if (cost < best_cost) best_cost = cost, *best[0] = rdo[0], *best[1] = rdo[1], best_dist = est[0] + est[1];
                }
                dist[0] += (double)best_dist / lambda;
                code_sao_offset(null, *best[0], 1);      // (the reference codes the winners once more into a list it drops)
                code_sao_offset(null, *best[1], 2);
                for (int nb = 0; nb < 2; nb++)
                    for (int c = 0; c < 2; c++) {
                        const int a = nb == 0 ? (allow_left ? addr_left : -1) : (allow_up ? addr_up : -1);
                        if (a < 0) continue;
                        const SaoUnit &n = sao_[c + 1][a];
                        merge[c + 1][nb] = n;
                        merge[c + 1][nb].merge_up = nb, merge[c + 1][nb].merge_left = !nb;
                        dist[nb + 1] += (double)merge_dist(c + 1, n) / lambda;
                    }
            }
            if (sao_on_[0] || sao_on_[1]) {
                rate.begin();
                if (allow_left) rate.bin(0, kCtxSaoMerge);
                if (allow_up) rate.bin(0, kCtxSaoMerge);
                for (int comp = 0; comp < 3; comp++)
                    if (comp == 0 ? sao_on_[0] : sao_on_[1]) code_sao_offset(rate, sao_[comp][addr], comp);
                double best_cost = dist[0] + (double)rate.bits();
                for (int up = 0; up < 2; up++) {
                    if (!((allow_left && up == 0) || (allow_up && up == 1))) continue;
                    rate.begin();
                    if (allow_left) rate.bin((unsigned)(1 - up), kCtxSaoMerge);
                    if (allow_up && up == 1) rate.bin(1, kCtxSaoMerge);
                    const double cost = dist[up + 1] + (double)rate.bits();
                    if (cost < best_cost) {
                        best_cost = cost;
                        for (int comp = 0; comp < 3; comp++) {
                            merge[comp][up].merge_left = 1 - up, merge[comp][up].merge_up = up;
                            if (comp == 0 ? sao_on_[0] : sao_on_[1]) sao_[comp][addr] = merge[comp][up];
                        }
                    }
                }
                if (sao_[0][addr].type_idx == -1) no_sao[0]++;
                if (sao_[1][addr].type_idx == -1) no_sao[1] += 2;
            }
        }
    an_.frac_bits[rows_ - 1] = rate.frac;
    (void)no_sao;      // rdoSaoUnitRowEnd stores the off-rates at depth 1, rdoSaoUnitRowInit reads depth 0: SAO stays enabled
}

int Encoder::encode(const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, analysis::Evaluator &ev, FrameOps &ops, uint8_t *out,
                    size_t cap)
{
    if (slice_type != 2 && !have_ref_) return -101;
    if (cap < 4096) return -100;               // parameter sets, slice header and the picture hash need a few hundred bytes at most
    if (pending_) return -103;
    int rc = submit(src_y, src_u, src_v, slice_type, ev, ops);
    if (rc) return rc;
    const auto t = std::chrono::steady_clock::now();
    rc = collect(out, cap);
    stage_ns[3] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count();
    if (rc < 0) {      // the access unit does not fit: the picture is given up and the sequence cannot go on
        pend_[first_].join();
        first_ ^= 1, pending_--;
        have_ref_ = false;
    }
    return rc;
}

int Encoder::submit(const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, int slice_type, analysis::Evaluator &ev, FrameOps &ops)
{
    using clock = std::chrono::steady_clock;
    auto ns = [](clock::time_point a, clock::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
    const bool intra = slice_type == 2;
    if (!intra && !have_ref_) return -101;
    if (pending_ == 2) return -103;
    const size_t ny = (size_t)w_ * h_;
    Pending &slot = pend_[(first_ + pending_) & 1], &other = pend_[(first_ + pending_ + 1) & 1];
    // parameter sets, slice header, entry points (4 bytes each at most), slice data (at most the substream buffer), hash unit;
    // two readable bytes in front for the writer's look-back
    slot.au.resize(2 + ny * 3 + 4096 + 4096 + 8 * (size_t)rows_ + 256);
    slot.au[0] = slot.au[1] = 0xff;
    uint8_t *const out = slot.au.data() + 2;
    const size_t cap = slot.au.size() - 2;
    // What the handle carries from picture to picture is committed when the slice is complete; an error on the way leaves
    // the handle without a reference picture (its analysis statistics may have advanced: the sequence cannot go on).
    struct Commit {
        Encoder &e;
        bool ok = false;
        int poc;
        ~Commit()
        {
            if (ok) e.poc_ = poc, e.have_ref_ = true;
            else e.have_ref_ = false;
        }
    } commit{ *this, false, intra ? 0 : poc_ + 1 };
    const int poc = commit.poc;
    const int wc = w_ / 2, hc = h_ / 2;
    if (!intra) {
        pad_plane(ref_[0].data(), w_, h_, kPad);
        pad_plane(ref_[1].data(), wc, hc, kPadC);
        pad_plane(ref_[2].data(), wc, hc, kPadC);
    }

    BitWriter bs;
    bs.init(out, out + cap);
    auto start_code = [&](uint8_t nal_type_byte) {
        bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(1), bs.raw_byte(nal_type_byte);
    };
    if (intra) {
        const bitstream::StreamParams sp = { w_, h_, strong_ };
        bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(1);
        bs.put(0x4001, 16);
        bitstream::write_vps(bs);
        bs.flush();
        bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(1);
        bs.put(0x4201, 16);
        bitstream::write_sps(bs, sp);
        bs.flush();
        bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(1);
        bs.put(0x4401, 16);
        bitstream::write_pps(bs);
        bs.flush();
        start_code(0x26);
    } else {
        bs.raw_byte(0);
        start_code(0x02);
    }
    bs.put(0x01, 8);

    // ---- decisions and reconstruction of every CTU ----
    const auto t0 = clock::now();
    const uint8_t *srcp[3] = { src_y, src_u, src_v }, *refp[3] = { nullptr, nullptr, nullptr };
    if (!intra) refp[0] = ref_[0].data(), refp[1] = ref_[1].data(), refp[2] = ref_[2].data();
    ev.coder_contexts(have_ctx_ ? last_ctx_ : nullptr);
    int rc = ev.bind_pictures(srcp, refp, kPad, qp_);
    if (rc) return rc < 0 ? rc : -rc;
    rc = intra ? an_.i_picture(qp_, strong_, ev, res_) : an_.p_picture(qp_, ev, res_);
    if (rc) return rc < 0 ? rc : -rc;
    const auto t1 = clock::now();

    // ---- in-loop filters ----
    const int w4 = w_ / 4, w8 = w_ / 8;
    for (int y = 0; y < rows_; y++)
        for (int x = 0; x < cols_; x++) {
            const CtuView v = { res_, x, y, cols_, w4, w8, w_ };
            ctu_units(v, unit_log2_.data(), unit_cbf_.data());
        }
    sao_on_[0] = sao_on_[1] = true;      // rdoSaoUnitRowInit (src/encode.cpp:305): the rates it tests are never written
    const uint8_t *recp[3] = { res_.rec_y.data(), res_.rec_u.data(), res_.rec_v.data() }, *orgp[3] = { src_y, src_u, src_v };
    rc = ops.deblock_and_stats(w_, h_, intra, qp_, unit_log2_.data(), unit_cbf_.data(), reinterpret_cast<const int16_t *>(res_.mv.data()), recp, orgp,
                               stats_.data());
    if (rc) return rc < 0 ? rc : -rc;
    const auto t2 = clock::now();
    sao_decide(slice_type, stats_.data());
    {
        const size_t nctu = (size_t)rows_ * cols_;
        sao_params_.resize(3 * nctu);
        const SaoCtuParam *pp[3] = { nullptr, nullptr, nullptr };
        for (int comp = 0; comp < 3; comp++) {
            if (!(comp == 0 ? sao_on_[0] : sao_on_[1])) continue;
            SaoCtuParam *params = &sao_params_[comp * nctu];
            for (size_t i = 0; i < nctu; i++) {
                const SaoUnit &u = sao_[comp][i];
                params[i].type_idx = (int8_t)u.type_idx, params[i].band_position = (uint8_t)u.sub_type;
                for (int k = 0; k < 4; k++) params[i].offset[k] = (int8_t)u.offset[k];
                params[i].reserved[0] = params[i].reserved[1] = 0;
            }
            pp[comp] = params;
        }
        other.join(), slot.join();      // the hash of the picture before reads rec_
        uint8_t *outp[3] = { rec_[0].data(), rec_[1].data(), rec_[2].data() };
        rc = ops.sao_apply(w_, h_, pp, outp);
        if (rc) return rc < 0 ? rc : -rc;
    }
    const auto t3 = clock::now();

    // The picture hash of the final reconstruction (src/encode.cpp:6012-6053) follows the slice in the access unit and does not
    // depend on it: MD5 runs at about 0.8 GB/s and cannot be split within a plane, so luma and the two chroma planes are
    // hashed on two threads while this one writes the slice.
    // They are joined by collect(), by the next picture before it replaces rec_, or by the destructor.
    sub_.resize(ny * 3 + 4096);
    slot.join();
    slot.hashed = hash_on_;
    if (hash_on_) {
        Pending *const q = &slot;
        const uint8_t *const py = rec_[0].data(), *const pu = rec_[1].data(), *const pv = rec_[2].data();
        const size_t w = (size_t)w_, h = (size_t)h_;
        slot.hash_y = std::thread([=] { bitstream::md5_plane(py, w, w, h, q->digest[0]); });
        slot.hash_c = std::thread([=] {
            bitstream::md5_plane(pu, w / 2, w / 2, h / 2, q->digest[1]);
            bitstream::md5_plane(pv, w / 2, w / 2, h / 2, q->digest[2]);
        });
    }

    // ---- slice header, then the slice data: one substream per CTU row (xCompressCU, src/encode.cpp:5594-5750) ----
    bitstream::write_slice_header(bs, slice_type, poc, sao_on_[0], sao_on_[1], qp_);
    sub_[0] = sub_[1] = 0xff;
    BitWriter sb;
    sb.init(sub_.data() + 2, sub_.data() + sub_.size());
    Cabac cabac;
    cabac.out = &sb;
    analysis::init_contexts(slice_type, qp_, cabac.ctx);
    cabac.reset();
    uint8_t saved[256] = {};      // (a picture one CTU wide never stores them: the reference then starts every further row from its zero-filled set)
    std::vector<int32_t> bytes0((size_t)rows_ + 2, 0), entry((size_t)rows_ + 1, 0);
    for (int y = 0; y < rows_; y++)
        for (int x = 0; x < cols_; x++) {
            if (x == 0 && y > 0) {
                cabac.flush();
                sb.trailing_bits();
                bytes0[y] = sb.flush();
                entry[y - 1] = bytes0[y] - bytes0[y - 1] - (int32_t)sb.escapes;
                sb.escapes = 0;
                cabac.reset();
                std::memcpy(cabac.ctx, saved, 256);
            }
            // SAO parameters of the CTU (src/encode.cpp:5952-5996)
            if (sao_on_[0] || sao_on_[1]) {
                const int addr = y * cols_ + x;
                int merge_left = sao_[0][addr].merge_left, merge_up = sao_[0][addr].merge_up;
                if (x > 0) cabac.bin(merge_left != 0, kCtxSaoMerge);
                else merge_left = 0;
                if (!merge_left) {
                    if (y > 0) cabac.bin(merge_up != 0, kCtxSaoMerge);
                    else merge_up = 0;
                    if (!merge_up)
                        for (int comp = 0; comp < 3; comp++)
                            if (comp == 0 ? sao_on_[0] : sao_on_[1]) code_sao_offset(cabac, sao_[comp][addr], comp);
                }
            }
            const CtuView v = { res_, x, y, cols_, w4, w8, w_ };
            write_ctu(cabac, v, intra);
            // end_of_slice_segment_flag; a CTU row that is not the last ends with end_of_subset_one_bit (src/encode.cpp:5517-5532)
            if (x == cols_ - 1) {
                if (y != rows_ - 1) cabac.terminate(0);
                cabac.terminate(1);
            } else {
                cabac.terminate(0);
            }
            if (x == 1 && y != rows_ - 1) std::memcpy(saved, cabac.ctx, 256);
        }
    const int n_entry = rows_ - 1;
    std::memcpy(last_ctx_, cabac.ctx, 256);
    have_ctx_ = true;
    cabac.flush();
    sb.trailing_bits();
    bytes0[n_entry + 1] = sb.flush();
    entry[n_entry] = bytes0[n_entry + 1] - bytes0[n_entry] - (int32_t)sb.escapes;
    sb.escapes = 0;
    int32_t max_offset = 0;
    for (int i = 0; i < n_entry; i++) max_offset = entry[i] > max_offset ? entry[i] : max_offset;
    int offset_len_minus1 = 0;
    while (max_offset >= (1 << (offset_len_minus1 + 1))) offset_len_minus1++;
    const int total = sb.flush();
    bitstream::write_entry_points(bs, n_entry, offset_len_minus1, entry.data());
    bs.flush();
    bs.raw_bytes(sub_.data() + 2, (size_t)total);
    cabac.out = &bs;
    cabac.flush();      // the reference flushes the coder a second time, into the access unit (src/encode.cpp:5749)
    bs.trailing_bits();
    // slice data beyond the substream buffer (more than 3 bytes per luma sample): nothing was written past its end; the call
    // fails. 64 bytes cover the picture-hash unit that follows, escapes included.
    if (sb.overflowed() || bs.overflowed() || bs.room() < 64 + 8) {
        slot.join();
        return -102;
    }

    // ---- the reconstruction becomes the reference of the next picture ----
    for (int c = 0; c < 3; c++) {
        const int pw = c ? wc : w_, ph = c ? hc : h_, pad = c ? kPadC : kPad, stride = pw + 2 * pad;
        for (int i = 0; i < ph; i++) std::memcpy(&ref_[c][(size_t)(pad + i) * stride + pad], &rec_[c][(size_t)i * pw], (size_t)pw);
    }

    slot.bs = bs, slot.sei_done = false, slot.len = 0;
    pending_++;
    commit.ok = true;
    const auto t4 = clock::now();
    stage_ns[0] = ns(t0, t1), stage_ns[1] = ns(t1, t2), stage_ns[2] = ns(t2, t3), stage_ns[3] = ns(t3, t4);
    return 0;
}

int Encoder::collect(uint8_t *out, size_t cap)
{
    if (!pending_) return -104;
    Pending &slot = pend_[first_];
    if (!slot.sei_done) {
        // ---- picture hash SEI ----
        slot.join();
        BitWriter &bs = slot.bs;
        bs.flush();
        if (slot.hashed) {
            bs.raw_byte(0), bs.raw_byte(0), bs.raw_byte(1), bs.raw_byte(0x50);
            bs.put(0x01, 8), bs.put(0x84, 8), bs.put(0x31, 8), bs.put(0x00, 8);
            for (int c = 0; c < 3; c++)
                for (int j = 0; j < 16; j++) bs.put(slot.digest[c][j], 8);
            bs.trailing_bits();
            bs.flush();
        }
        int len = bs.flush();
        uint8_t *au = slot.au.data() + 2;
        if (au[len - 1] == 0) au[len++] = 0x03;
        slot.len = len, slot.sei_done = true;
    }
    if ((size_t)slot.len > cap) return -102;      // stays pending: a larger buffer can still take it
    std::memcpy(out, slot.au.data() + 2, (size_t)slot.len);
    first_ ^= 1, pending_--;
    return slot.len;
}

}  // namespace encoder
}  // namespace xmi
