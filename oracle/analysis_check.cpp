// analysis_check.cpp - TEST INFRASTRUCTURE: the host logic of the P-picture analysis loop (x265_amd/csrc/host/analysis_core.cpp,
// the same source libx265_amd.so is built from) bound to a CPU evaluator made of the oracle's primitives, so that the
// decision logic can be held against traces of the reference's encoder without a GPU (tests/test_analysis_cpu.py). Built
// into oracle/libx265_analysis_check.so by `make -C oracle analysis_check`. Nothing in the product links or loads this.
#include <cstring>
#include <mutex>
#include <vector>

#include "../x265_amd/csrc/host/analysis_core.h"
#include "x265_oracle.h"

using namespace xmi::analysis;

namespace {

struct Pictures {
    const uint8_t *src_y, *src_u, *src_v;   // width x height, width/2 x height/2
    const uint8_t *ref_y, *ref_u, *ref_v;   // sample (0,0) of the padded planes
    int w, h, ref_stride_y, ref_stride_c;
};

class CpuEvaluator : public Evaluator {
public:
    Pictures p;
    std::mutex mu;
    std::vector<uint8_t> lane_bytes[8];
    std::vector<int16_t> lane_words[8];

    int evaluate(int lane, CuEval *const *cu, int ncu, ChromaEval *const *ch, int nch) override
    {
        std::lock_guard<std::mutex> lk(mu);     // the oracle's transform chain keeps a static work buffer
        std::vector<uint8_t> &bytes = lane_bytes[lane & 7];
        std::vector<int16_t> &words = lane_words[lane & 7];
        size_t nb = 0, nw = 0;
        for (int i = 0; i < ncu; i++) nb += 3 * (size_t)cu[i]->size * cu[i]->size, nw += 2 * (size_t)cu[i]->size * cu[i]->size;
        for (int i = 0; i < nch; i++) nb += (size_t)ch[i]->size * ch[i]->size / 2, nw += (size_t)ch[i]->size * ch[i]->size / 2;
        bytes.assign(nb, 0);
        words.assign(nw, 0);
        uint8_t *b = bytes.data();
        int16_t *w = words.data();
        for (int i = 0; i < ncu; i++) one_cu(*cu[i], b, w);
        for (int i = 0; i < nch; i++) one_chroma(*ch[i], b, w);
        return 0;
    }

private:
    // transform chain(s) of a size x size block held with 64-sample rows; results packed with `size`-sample rows
    static uint32_t chains(const uint8_t *pix, const uint8_t *pred, int size, int qp, uint32_t sum[4], uint8_t *rec_out, int16_t *lev_out, int stride)
    {
        uint8_t rec[64 * 80];
        int16_t lev[64 * 64];
        std::memset(lev, 0, sizeof lev);
        sum[0] = sum[1] = sum[2] = sum[3] = 0;
        const int full = stride == 64 ? 64 : 32;     // size at which a block is cut into four units
        if (size == full) {
            const int hsz = size / 2;
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++) {
                    const int o = j * hsz * stride + i * hsz, orec = j * hsz * 80 + i * hsz;
                    sum[j * 2 + i] = orc_tu_chain(pix + o, pred + o, rec + orec, lev + o, hsz, stride, 80, (unsigned)qp, ORC_MODE_INVALID, ORC_SLICE_P);
                }
        } else {
            sum[0] = orc_tu_chain(pix, pred, rec, lev, size, stride, 80, (unsigned)qp, ORC_MODE_INVALID, ORC_SLICE_P);
        }
        for (int r = 0; r < size; r++) {
            std::memcpy(rec_out + r * size, rec + r * 80, (size_t)size);
            std::memcpy(lev_out + r * size, lev + r * stride, (size_t)size * 2);
        }
        return orc_ssd(size, pix, stride, rec, 80);
    }

    void one_cu(CuEval &q, uint8_t *&b, int16_t *&w)
    {
        const int n = q.size;
        uint8_t cur[64 * 64], pred[64 * 64], best_pred[64 * 64];
        for (int r = 0; r < n; r++) std::memcpy(cur + r * 64, p.src_y + (size_t)(q.y0 + r) * p.w + q.x0, (size_t)n);
        const uint8_t *ref = p.ref_y + (size_t)q.y0 * p.ref_stride_y + q.x0;
        uint8_t *pred_m = b, *rec_m = b + n * n, *rec_e = b + 2 * n * n;
        int16_t *lev_m = w, *lev_e = w + n * n;
        b += 3 * n * n, w += 2 * n * n;
        q.best_merge = -1;
        q.ssd_skip = 0xffffffffu;
        for (int k = 0; k < q.n_merge; k++) {
            if (!q.merge_ok[k]) continue;
            orc_pred_inter_luma(ref, pred, p.ref_stride_y, n, n, q.merge_mv[k].x, q.merge_mv[k].y);
            const uint32_t s = orc_ssd(n, cur, 64, pred, 64);
            if (s < q.ssd_skip) {
                q.ssd_skip = s;
                q.best_merge = k;
                std::memcpy(best_pred, pred, sizeof pred);
            }
        }
        int zero_tested = 0;
        q.pred_merge = pred_m, q.rec_merge = rec_m, q.level_merge = lev_m;
        if (q.best_merge >= 0) {
            for (int r = 0; r < n; r++) std::memcpy(pred_m + r * n, best_pred + r * 64, (size_t)n);
            q.ssd_merge = chains(cur, best_pred, n, q.qp, q.sum_merge, rec_m, lev_m, 64);
            zero_tested = q.merge_mv[q.best_merge].x == 0 && q.merge_mv[q.best_merge].y == 0;
        }
        const orc_mv mvp[2] = { { q.amvp[0].x, q.amvp[0].y }, { q.amvp[1].x, q.amvp[1].y } };
        const orc_mv lo = { q.mv_min_ipel.x, q.mv_min_ipel.y }, hi = { q.mv_max_ipel.x, q.mv_max_ipel.y };
        orc_me_result me;
        orc_motion_search(cur, ref, p.ref_stride_y, n, mvp, zero_tested, lo, hi, q.lambda, 32, &me);
        q.mv_me = { me.mv.x, me.mv.y };
        orc_pred_inter_luma(ref, pred, p.ref_stride_y, n, n, me.mv.x, me.mv.y);
        q.rec_me = rec_e, q.level_me = lev_e;
        q.ssd_me = chains(cur, pred, n, q.qp, q.sum_me, rec_e, lev_e, 64);
    }

    void one_chroma(ChromaEval &c, uint8_t *&b, int16_t *&w)
    {
        const int n = c.size / 2, x = c.x0 / 2, y = c.y0 / 2, wc = p.w / 2;
        uint8_t *rec[2] = { b, b + n * n };
        int16_t *lev[2] = { w, w + n * n };
        b += 2 * n * n, w += 2 * n * n;
        c.rec_u = rec[0], c.rec_v = rec[1], c.level_u = lev[0], c.level_v = lev[1];
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t *src = pl ? p.src_v : p.src_u, *ref = (pl ? p.ref_v : p.ref_u) + (size_t)y * p.ref_stride_c + x;
            uint8_t cur[32 * 32], pred[32 * 32];
            for (int r = 0; r < n; r++) std::memcpy(cur + r * 32, src + (size_t)(y + r) * wc + x, (size_t)n);
            orc_pred_inter_chroma(ref, pred, p.ref_stride_c, n, n, c.mv.x, c.mv.y);
            uint32_t *sum = pl ? c.sum_v : c.sum_u;
            if (c.skip) {
                sum[0] = sum[1] = sum[2] = sum[3] = 0;
                for (int r = 0; r < n; r++) std::memcpy(rec[pl] + r * n, pred + r * 32, (size_t)n);
            } else {
                chains(cur, pred, n, c.qpc, sum, rec[pl], lev[pl], 32);
            }
        }
    }
};

struct Handle {
    Analyser an;
    CpuEvaluator ev;
    PictureResult res;
    Handle(int w, int h) : an(w, h) {}
};

}  // namespace

extern "C" {

void *orc_analysis_create(int width, int height) { return new Handle(width, height); }
int orc_analysis_configure(void *h, int lanes, int spin_us)
{
    if (lanes < 1 || lanes > 8 || spin_us < 0) return -1;
    static_cast<Handle *>(h)->an.lanes = lanes;
    static_cast<Handle *>(h)->an.spin_us = spin_us;
    return 0;
}
void orc_analysis_destroy(void *h) { delete static_cast<Handle *>(h); }

// set != 0: load the carried state; else read it back. Arrays: rows * cols * 5 (sum, count, avg), rows (frac).
void orc_analysis_state(void *hv, int set, double *sum, int32_t *count, double *avg, uint64_t *frac)
{
    Analyser &a = static_cast<Handle *>(hv)->an;
    const size_t n = a.sum_cost.size();
    if (set) {
        std::memcpy(a.sum_cost.data(), sum, n * 8), std::memcpy(a.count.data(), count, n * 4), std::memcpy(a.avg_cost.data(), avg, n * 8);
        std::memcpy(a.frac_bits.data(), frac, a.frac_bits.size() * 8);
    } else {
        std::memcpy(sum, a.sum_cost.data(), n * 8), std::memcpy(count, a.count.data(), n * 4), std::memcpy(avg, a.avg_cost.data(), n * 8);
        std::memcpy(frac, a.frac_bits.data(), a.frac_bits.size() * 8);
    }
}

// ref_*: first byte of the padded planes ((w + 2 pad) x (h + 2 pad), chroma with pad / 2)
int orc_analysis_p_picture(void *hv, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const uint8_t *ref_y, const uint8_t *ref_u,
                           const uint8_t *ref_v, int pad, int qp, uint16_t *info0, uint8_t *info1, int16_t *mv, int16_t *mvd, uint8_t *mvp_idx,
                           int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, uint32_t *rd_cost,
                           uint8_t *ctx_in, uint8_t *ctx_out, uint64_t *counters)
{
    Handle &H = *static_cast<Handle *>(hv);
    const int w = H.an.width(), h = H.an.height(), pc = pad / 2;
    Pictures &p = H.ev.p;
    p.w = w, p.h = h;
    p.src_y = src_y, p.src_u = src_u, p.src_v = src_v;
    p.ref_stride_y = w + 2 * pad, p.ref_stride_c = w / 2 + 2 * pc;
    p.ref_y = ref_y + (size_t)pad * p.ref_stride_y + pad;
    p.ref_u = ref_u + (size_t)pc * p.ref_stride_c + pc;
    p.ref_v = ref_v + (size_t)pc * p.ref_stride_c + pc;
    const int rc = H.an.p_picture(qp, H.ev, H.res);
    if (rc) return rc;
    const PictureResult &r = H.res;
    std::memcpy(info0, r.info0.data(), r.info0.size() * 2), std::memcpy(info1, r.info1.data(), r.info1.size());
    std::memcpy(mv, r.mv.data(), r.mv.size() * 4), std::memcpy(mvd, r.mvd.data(), r.mvd.size() * 4), std::memcpy(mvp_idx, r.mvp_idx.data(), r.mvp_idx.size());
    std::memcpy(coef_y, r.coef_y.data(), r.coef_y.size() * 2), std::memcpy(coef_u, r.coef_u.data(), r.coef_u.size() * 2);
    std::memcpy(coef_v, r.coef_v.data(), r.coef_v.size() * 2);
    std::memcpy(rec_y, r.rec_y.data(), r.rec_y.size()), std::memcpy(rec_u, r.rec_u.data(), r.rec_u.size()), std::memcpy(rec_v, r.rec_v.data(), r.rec_v.size());
    std::memcpy(rd_cost, r.rd_cost.data(), r.rd_cost.size() * 4);
    std::memcpy(ctx_in, r.ctx_in.data(), r.ctx_in.size()), std::memcpy(ctx_out, r.ctx_out.data(), r.ctx_out.size());
    counters[0] = r.evaluator_calls, counters[1] = r.cu_requests, counters[2] = r.chroma_requests;
    return 0;
}

const uint16_t *orc_analysis_scan(int log2_size) { return diag_scan(log2_size); }
const uint8_t *orc_analysis_next_state(void) { return next_state_table(); }
void orc_analysis_init_contexts(int slice_type, int qp, uint8_t *ctx) { init_contexts(slice_type, qp, ctx); }

}  // extern "C"
