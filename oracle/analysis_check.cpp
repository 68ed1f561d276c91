// analysis_check.cpp - TEST INFRASTRUCTURE: the host logic of the P-picture analysis loop (x265_amd/csrc/host/analysis_core.cpp,
// the same source libx265_amd.so is built from) bound to a CPU evaluator made of the oracle's primitives, so that the
// decision logic can be held against traces of the reference's encoder without a GPU (tests/test_analysis_cpu.py). Built
// into oracle/libx265_analysis_check.so by `make -C oracle analysis_check`. Nothing in the product links or loads this.
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../x265_amd/csrc/host/analysis_core.h"
#include "../x265_amd/csrc/host/encoder_core.h"
#include "x265_oracle.h"

using namespace xmi::analysis;

namespace {

struct Pictures {
    const uint8_t *src_y, *src_u, *src_v;   // width x height, width/2 x height/2
    const uint8_t *ref_y, *ref_u, *ref_v;   // sample (0,0) of the padded planes
    int w, h, ref_stride_y, ref_stride_c;
};

// the checker's transform, prediction and chain functions keep static work buffers (oracle/x265_oracle.c): one lock for every
// handle of the process
std::mutex &oracle_mutex()
{
    static std::mutex m;
    return m;
}

class CpuEvaluator : public Evaluator {
public:
    Pictures p;
    std::mutex &mu = oracle_mutex();
    std::vector<uint8_t> lane_bytes[8];
    std::vector<int16_t> lane_words[8];

    int w = 0, h = 0;
    // the level decision of x265_rdoq.c in the transform chains of P pictures, as xmi_encoder_rdoq switches it on in the library
    bool rdoq = false, rdoq_active = false;
    orc_rdoq_tables rdoq_tables[4][2];
    uint32_t rdoq_m[4][2];
    int32_t rdoq_s[4][2];

    uint8_t coder_ctx[256];
    bool have_coder_ctx = false;
    void coder_contexts(const uint8_t *ctx) override
    {
        have_coder_ctx = ctx != nullptr;
        if (ctx) std::memcpy(coder_ctx, ctx, 256);
    }

    int bind_pictures(const uint8_t *const src[3], const uint8_t *const ref[3], int pad, int qp) override
    {
        rdoq_active = rdoq && ref[0];
        if (rdoq_active) {
            uint8_t ctx[256];
            if (have_coder_ctx) std::memcpy(ctx, coder_ctx, 256);
            else xmi::analysis::init_contexts(1, qp, ctx);
            const double lambda = 0.578 * std::pow(2.0, (qp - 12) / 3.0);
            for (int lg = 2; lg <= 5; lg++)
                for (int comp = 0; comp < 2; comp++) {
                    orc_rdoq_est_bits(ctx, lg, comp == 0, &rdoq_tables[lg - 2][comp]);
                    orc_rdoq_lambda(lambda, (unsigned)(comp ? xmi::analysis::chroma_qp(qp) : qp), lg, &rdoq_m[lg - 2][comp], &rdoq_s[lg - 2][comp]);
                }
        }
        const int pc = pad / 2;
        p.w = w, p.h = h;
        p.src_y = src[0], p.src_u = src[1], p.src_v = src[2];
        p.ref_stride_y = w + 2 * pad, p.ref_stride_c = w / 2 + 2 * pc;
        p.ref_y = ref[0] ? ref[0] + (size_t)pad * p.ref_stride_y + pad : nullptr;
        p.ref_u = ref[1] ? ref[1] + (size_t)pc * p.ref_stride_c + pc : nullptr;
        p.ref_v = ref[2] ? ref[2] + (size_t)pc * p.ref_stride_c + pc : nullptr;
        return 0;
    }

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
        if (defer)      // the results are held back as an evaluator with answers still under way would: finish_*() brings them
            for (int i = 0; i < ncu; i++) {
                CuEval &q = *cu[i];
                held[&q] = q;
                q.best_merge = 0x7bad, q.ssd_skip = q.ssd_merge = q.ssd_me = 0x7bad7badu;
                q.mv_me = { 0x7bad, 0x7bad };
                for (int u = 0; u < 4; u++) q.sum_merge[u] = q.sum_me[u] = 0x7bad7badu;
                q.merge_pending = q.search_pending = &q;
            }
        return 0;
    }

    // test switches: hold the results of the CU requests back until finish_merge() / finish_search() ask for them (anything
    // read before that is poison); let the n-th finish call fail
    bool defer = false;
    int fail_at = 0, finish_calls = 0;
    int refuse_every = 0, refuse_count = 0;      // I pictures: every n-th request that stands after a block gets no result
    std::unordered_map<const CuEval *, CuEval> held;

    int finish_merge(CuEval &q) override
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!q.merge_pending) return 0;
        const CuEval &s = held.at(&q);
        q.best_merge = s.best_merge, q.ssd_skip = s.ssd_skip, q.ssd_merge = s.ssd_merge;
        for (int u = 0; u < 4; u++) q.sum_merge[u] = s.sum_merge[u];
        q.merge_pending = nullptr;
        return ++finish_calls == fail_at ? -77 : 0;
    }

    int finish_search(CuEval &q) override
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!q.search_pending) return 0;
        const CuEval &s = held.at(&q);
        q.mv_me = s.mv_me, q.ssd_me = s.ssd_me;
        for (int u = 0; u < 4; u++) q.sum_me[u] = s.sum_me[u];
        q.search_pending = nullptr;
        return ++finish_calls == fail_at ? -77 : 0;
    }

    int evaluate_intra(int lane, const IntraBatch &bt) override
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<uint8_t> &bytes = lane_bytes[lane & 7];
        std::vector<int16_t> &words = lane_words[lane & 7];
        size_t n = 64 * (size_t)bt.nquad + 128 * (size_t)bt.ncquad;
        for (int i = 0; i < bt.npu; i++) n += (size_t)bt.pu[i]->size * bt.pu[i]->size;
        for (int i = 0; i < bt.nchroma; i++) n += 2 * (size_t)bt.chroma[i]->size * bt.chroma[i]->size;
        bytes.assign(n, 0);
        words.assign(n, 0);
        uint8_t *b = bytes.data();
        int16_t *w = words.data();
        for (int i = 0; i < bt.npu; i++)
            if (!bt.pu[i]->after_kind) one_intra(*bt.pu[i], b, w);
        for (int i = 0; i < bt.nchroma; i++) one_intra_chroma(*bt.chroma[i], b, w);
        for (int i = 0; i < bt.nquad; i++)
            if (!bt.quad[i]->after_kind) one_quad(*bt.quad[i], b, w);
        for (int i = 0; i < bt.ncquad; i++) one_chroma_quad(*bt.cquad[i], b, w);
        // requests that stand after a block of this call: that block's results are there now. (Test switch: every n-th of them
        // is answered as by an evaluator that could not wait for the block - no result -, and the caller has to ask again.)
        for (int i = 0; i < bt.npu; i++)
            if (bt.pu[i]->after_kind) {
                take_left_block(*bt.pu[i]), one_intra(*bt.pu[i], b, w);
                if (refuse_every && ++refuse_count % refuse_every == 0) bt.pu[i]->best_mode = -1;
            }
        for (int i = 0; i < bt.nquad; i++)
            if (bt.quad[i]->after_kind) {
                take_left_block(*bt.quad[i]), one_quad(*bt.quad[i], b, w);
                if (refuse_every && ++refuse_count % refuse_every == 0) bt.quad[i]->best_mode[0] = -1;
            }
        return 0;
    }

private:
    // xg_aucIntraFilterType (src/table.cpp:32): which modes predict from the smoothed samples
    static bool smoothed(int size, int mode)
    {
        if (size == 8) return mode == 0 || mode == 2 || mode == 18 || mode == 34;
        if (size == 16) return !(mode == 1 || (mode >= 9 && mode <= 11) || (mode >= 25 && mode <= 27));
        if (size == 32) return !(mode == 1 || mode == 10 || mode == 26);
        return false;
    }

    void one_intra(IntraEval &q, uint8_t *&b, int16_t *&w)
    {
        const int n = q.size;
        uint8_t cur[32 * 64], pred[32 * 64], rec[32 * 80], raw[132], flt[132];
        int16_t lev[32 * 64];
        for (int r = 0; r < n; r++) std::memcpy(cur + r * 64, p.src_y + (size_t)(q.y0 + r) * p.w + q.x0, (size_t)n);
        orc_intra_result res;
        orc_intra_search(cur, 64, q.ref, n, q.valids, q.strong, q.mpm, q.sqrt_lambda, &res);
        q.best_mode = res.best_mode;
        if (q.valids) {
            std::memcpy(raw, q.ref, (size_t)(4 * n + 1));
            orc_padding_ref(raw, n, q.valids);
            orc_filter_ref(flt, raw, n, q.strong);
        } else {
            std::memset(raw, 0x80, sizeof raw), std::memset(flt, 0x80, sizeof flt);
        }
        orc_intra_pred(pred, smoothed(n, res.best_mode) ? flt : raw, 64, res.best_mode, n, n != 32);
        std::memset(lev, 0, sizeof lev);
        q.sum = orc_tu_chain(cur, pred, rec, lev, n, 64, 80, (unsigned)q.qp, (unsigned)res.best_mode, ORC_SLICE_I);
        q.ssd = orc_ssd(n, cur, 64, rec, 80);
        for (int r = 0; r < n; r++) {
            std::memcpy(b + r * n, rec + r * 80, (size_t)n);
            std::memcpy(w + r * n, lev + r * 64, (size_t)n * 2);
        }
        q.rec = b, q.level = w;
        b += n * n, w += n * n;
    }

    // the 17 neighbouring samples of the 4x4 PU k of an 8x8 area: from the area's own 33 samples outside it, from the
    // reconstruction of the PUs before it inside (area: 8x8, rows 8 apart)
    static void quad_neighbours(uint8_t *ref, const uint8_t *outer, const uint8_t *area, int k)
    {
        const int ox = (k & 1) * 4, oy = (k >> 1) * 4;
        auto at = [&](int yy, int xx) -> uint8_t {      // relative to the area's origin
            if (yy < 0) return outer[16 + 1 + xx];
            if (xx < 0) return outer[15 - yy];
            return xx < 8 && yy < 8 ? area[yy * 8 + xx] : (uint8_t)0x80;
        };
        for (int i = 0; i < 8; i++) ref[i] = at(oy + 7 - i, ox - 1);
        for (int i = 0; i <= 8; i++) ref[8 + i] = at(oy - 1, ox - 1 + i);
    }

    // four IntraEval of size 4 back to back
    void one_quad(IntraQuadEval &q, uint8_t *&b, int16_t *&w)
    {
        uint8_t *area = b;
        int16_t *lev = w;
        b += 64, w += 64;
        for (int k = 0; k < 4; k++) {
            IntraEval e;
            e.x0 = q.x0 + (k & 1) * 4, e.y0 = q.y0 + (k >> 1) * 4, e.size = 4;
            e.valids = q.valids[k], e.strong = q.strong, e.sqrt_lambda = q.sqrt_lambda, e.qp = q.qp;
            const int left = (k & 1) ? q.best_mode[k - 1] : q.left_mode[k >> 1], top = (k >> 1) ? q.best_mode[k - 2] : q.top_mode[k & 1];
            if (left == top) {
                if (left > 1) e.mpm[0] = (uint8_t)left, e.mpm[1] = (uint8_t)(((left + 29) % 32) + 2), e.mpm[2] = (uint8_t)(((left - 1) % 32) + 2);
                else e.mpm[0] = 0, e.mpm[1] = 1, e.mpm[2] = 26;
            } else {
                e.mpm[0] = (uint8_t)left, e.mpm[1] = (uint8_t)top;
                e.mpm[2] = (left && top) ? 0 : (left + top < 2 ? 26 : 1);
            }
            quad_neighbours(e.ref, q.ref, area, k);
            uint8_t tmp_b[16], *tb = tmp_b;
            int16_t tmp_w[16], *tw = tmp_w;
            one_intra(e, tb, tw);
            q.best_mode[k] = e.best_mode, q.ssd[k] = e.ssd, q.sum[k] = e.sum;
            for (int r = 0; r < 4; r++) {
                std::memcpy(area + ((k >> 1) * 4 + r) * 8 + (k & 1) * 4, tmp_b + r * 4, 4);
                std::memcpy(lev + ((k >> 1) * 4 + r) * 8 + (k & 1) * 4, tmp_w + r * 4, 8);
            }
        }
        q.rec = area, q.level = lev;
    }

    // four IntraChromaEval of size 4 back to back
    void one_chroma_quad(IntraChromaQuadEval &c, uint8_t *&b, int16_t *&w)
    {
        uint8_t *area[2] = { b, b + 64 };
        int16_t *lev[2] = { w, w + 64 };
        b += 128, w += 128;
        for (int k = 0; k < 4; k++) {
            IntraChromaEval e;
            e.x0 = c.x0 + (k & 1) * 4, e.y0 = c.y0 + (k >> 1) * 4, e.size = 4;
            e.valids = c.valids[k], e.luma_mode = c.luma_mode[k], e.qpc = c.qpc;
            quad_neighbours(e.ref_u, c.ref_u, area[0], k);
            quad_neighbours(e.ref_v, c.ref_v, area[1], k);
            uint8_t tmp_b[32], *tb = tmp_b;
            int16_t tmp_w[32], *tw = tmp_w;
            one_intra_chroma(e, tb, tw);
            c.best_idx[k] = e.best_idx, c.sum_u[k] = e.sum_u, c.sum_v[k] = e.sum_v;
            for (int pl = 0; pl < 2; pl++)
                for (int r = 0; r < 4; r++) {
                    std::memcpy(area[pl] + ((k >> 1) * 4 + r) * 8 + (k & 1) * 4, (pl ? e.rec_v : e.rec_u) + r * 4, 4);
                    std::memcpy(lev[pl] + ((k >> 1) * 4 + r) * 8 + (k & 1) * 4, (pl ? e.level_v : e.level_u) + r * 4, 8);
                }
        }
        c.rec_u = area[0], c.rec_v = area[1], c.level_u = lev[0], c.level_v = lev[1];
    }

    void one_intra_chroma(IntraChromaEval &c, uint8_t *&b, int16_t *&w)
    {
        const int n = c.size, wc = p.w / 2;
        uint8_t cur[2][16 * 32], pred[16 * 32], rec[16 * 40], ref[2][68];
        int16_t lev[16 * 32];
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t *src = pl ? p.src_v : p.src_u;
            for (int r = 0; r < n; r++) std::memcpy(cur[pl] + r * 32, src + (size_t)(c.y0 + r) * wc + c.x0, (size_t)n);
            if (c.valids) {
                std::memcpy(ref[pl], pl ? c.ref_v : c.ref_u, (size_t)(4 * n + 1));
                orc_padding_ref(ref[pl], n, c.valids);
            } else {
                std::memset(ref[pl], 0x80, sizeof ref[pl]);
            }
        }
        orc_chroma_result res;
        orc_intra_chroma_search(cur[0], cur[1], 32, ref[0], ref[1], n, c.luma_mode, &res);
        c.best_idx = res.best_idx;
        for (int pl = 0; pl < 2; pl++) {
            orc_intra_pred(pred, ref[pl], 32, res.best_mode, n, 0);
            std::memset(lev, 0, sizeof lev);
            const uint32_t sum = orc_tu_chain(cur[pl], pred, rec, lev, n, 32, 40, (unsigned)c.qpc, ORC_MODE_INVALID, ORC_SLICE_I);
            for (int r = 0; r < n; r++) {
                std::memcpy(b + r * n, rec + r * 40, (size_t)n);
                std::memcpy(w + r * n, lev + r * 32, (size_t)n * 2);
            }
            (pl ? c.sum_v : c.sum_u) = sum;
            (pl ? c.rec_v : c.rec_u) = b, (pl ? c.level_v : c.level_u) = w;
            b += n * n, w += n * n;
        }
    }

    // xSubDctQuantRec of one inter unit (rows `stride` apart, reconstruction rows 80 apart): the reference's quantiser, or the
    // level decision in its place
    uint32_t unit_chain(const uint8_t *pix, const uint8_t *pred, uint8_t *rec, int16_t *lev, int n, int stride, int qp) const
    {
        if (!rdoq_active) return orc_tu_chain(pix, pred, rec, lev, n, stride, 80, (unsigned)qp, ORC_MODE_INVALID, ORC_SLICE_P);
        const int lg = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5, comp = stride == 64 ? 0 : 1;
        int16_t work[32 * 64];
        orc_sub_dct(work, pix, stride, pred, stride, n, ORC_MODE_INVALID);
        const uint32_t sum = orc_rdoq_quant(lev, work, stride, n, (unsigned)qp, comp == 0, 0, &rdoq_tables[lg - 2][comp], rdoq_m[lg - 2][comp], rdoq_s[lg - 2][comp]);
        if (sum) {
            orc_dequant(work, lev, stride, (unsigned)qp, n, n);
            orc_idct_add(rec, 80, work, stride, pred, n, ORC_MODE_INVALID);
        } else {
            for (int r = 0; r < n; r++) std::memcpy(rec + r * 80, pred + r * stride, (size_t)n);
        }
        return sum;
    }

    // transform chain(s) of a size x size block held with 64-sample rows; results packed with `size`-sample rows
    uint32_t chains(const uint8_t *pix, const uint8_t *pred, int size, int qp, uint32_t sum[4], uint8_t *rec_out, int16_t *lev_out, int stride) const
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
                    sum[j * 2 + i] = unit_chain(pix + o, pred + o, rec + orec, lev + o, hsz, stride, qp);
                }
        } else {
            sum[0] = unit_chain(pix, pred, rec, lev, size, stride, qp);
        }
        for (int r = 0; r < size; r++) {
            std::memcpy(rec_out + r * size, rec + r * 80, (size_t)size);
            std::memcpy(lev_out + r * size, lev + r * stride, (size_t)size * 2);
        }
        // the evaluator contract leaves the levels of a unit undefined when their sum is zero (the GPU does not store them):
        // poisoned here, so that host code that reads them anyway fails the parity tests on the CPU already
        const int units = size == full ? 4 : 1, un = units == 4 ? size / 2 : size;
        for (int u = 0; u < units; u++)
            if (!sum[u])
                for (int r = 0; r < un; r++)
                    for (int c = 0; c < un; c++) lev_out[((units == 4 ? (u >> 1) * un : 0) + r) * size + (units == 4 ? (u & 1) * un : 0) + c] = 0x2bad;
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

// the picture-level passes of the encoder on the checker's primitives
class CpuFrameOps : public xmi::encoder::FrameOps {
public:
    std::vector<uint8_t> plane[3];      // the deblocked picture

    int deblock_and_stats(int width, int height, int intra_slice, int qp, const uint8_t *unit_log2, const uint8_t *cbf, const int16_t *mv,
                          const uint8_t *const rec[3], const uint8_t *const org[3], xmi::encoder::SaoStats *stats) override
    {
        const size_t n = (size_t)(width / 8) * (height / 4), ny = (size_t)width * height;
        std::vector<uint8_t> vbs(n), hbs(n), qp_tab((size_t)(width / 4) * (height / 4), (uint8_t)qp);
        orc_deblock_strengths(width, height, intra_slice, unit_log2, cbf, mv, vbs.data(), hbs.data());
        for (int c = 0; c < 3; c++) plane[c].assign(rec[c], rec[c] + (c ? ny / 4 : ny));
        const orc_deblock_params dp = { width, height, width, width / 2, 0, 0, 0, 0 };
        orc_deblock_frame(plane[0].data(), plane[1].data(), plane[2].data(), &dp, vbs.data(), hbs.data(), qp_tab.data());
        const int cols = width / 64, rows = height / 64;
        for (int cy = 0; cy < rows; cy++)
            for (int cx = 0; cx < cols; cx++)
                for (int c = 0; c < 3; c++) {
                    int64_t count[5][33], off[5][33];
                    const int pw = c ? width / 2 : width, ph = c ? height / 2 : height;
                    orc_sao_stats_ctu(plane[c].data(), pw, org[c], pw, pw, ph, cx, cy, c ? 32 : 64, c != 0, count, off);
                    xmi::encoder::SaoStats &s = stats[(size_t)(cy * cols + cx) * 3 + c];
                    for (int t = 0; t < 5; t++)
                        for (int k = 0; k < 33; k++) s.count[t][k] = (int32_t)count[t][k], s.offset_org[t][k] = (int32_t)off[t][k];
                }
        return 0;
    }
    int sao_apply(int width, int height, const xmi::encoder::SaoCtuParam *const params[3], uint8_t *const out[3]) override
    {
        static_assert(sizeof(orc_sao_ctu_param) == sizeof(xmi::encoder::SaoCtuParam), "same layout");
        for (int c = 0; c < 3; c++) {
            const int pw = c ? width / 2 : width, ph = c ? height / 2 : height;
            if (params[c]) orc_sao_apply_plane(plane[c].data(), pw, out[c], pw, pw, ph, c ? 32 : 64, reinterpret_cast<const orc_sao_ctu_param *>(params[c]));
            else std::memcpy(out[c], plane[c].data(), plane[c].size());
        }
        return 0;
    }
};

struct EncHandle {
    xmi::encoder::Encoder enc;
    CpuEvaluator ev;
    CpuFrameOps ops;
    EncHandle(int w, int h, int qp, int strong) : enc(w, h, qp, strong) { ev.w = w, ev.h = h; }
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
// how far a CU's request reaches ahead (Analyser::speculate): 0 none, 1 first children, 2 the rest of the CTU
int orc_analysis_speculate(void *h, int level)
{
    if (level < 0 || level > 2) return -1;
    static_cast<Handle *>(h)->an.speculate = level;
    return 0;
}
// worker threads the CTU rows are dealt to as fibers (Analyser::workers; 0 = by the host's cores)
int orc_analysis_workers(void *h, int n)
{
    if (n < 0) return -1;
    static_cast<Handle *>(h)->an.workers = n;
    return 0;
}
// test switches of the harness's evaluator: results held back until the rows ask for them; the n-th such request fails (0: none)
int orc_analysis_defer(void *h, int on, int fail_at)
{
    CpuEvaluator &ev = static_cast<Handle *>(h)->ev;
    ev.defer = on != 0, ev.fail_at = fail_at, ev.finish_calls = 0;
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

// ---- the whole encoder on the checker's primitives ----
void *orc_encoder_create(int width, int height, int qp, int strong_intra_smoothing) { return new EncHandle(width, height, qp, strong_intra_smoothing); }
void orc_encoder_destroy(void *h) { delete static_cast<EncHandle *>(h); }
int orc_encoder_configure(void *h, int lanes, int spin_us)
{
    static_cast<EncHandle *>(h)->enc.analyser().lanes = lanes;
    static_cast<EncHandle *>(h)->enc.analyser().spin_us = spin_us;
    return 0;
}
int orc_encoder_workers(void *h, int workers)
{
    static_cast<EncHandle *>(h)->enc.analyser().workers = workers;
    return 0;
}
int orc_encoder_picture_hash(void *h, int on)
{
    static_cast<EncHandle *>(h)->enc.picture_hash(on != 0);
    return 0;
}
int orc_encoder_rdoq(void *h, int on)
{
    static_cast<EncHandle *>(h)->ev.rdoq = on != 0;
    return 0;
}
int orc_encoder_refuse_pairs(void *h, int every)
{
    static_cast<EncHandle *>(h)->ev.refuse_every = every;
    return 0;
}
int orc_encoder_speculate(void *h, int level)
{
    if (level < 0 || level > 2) return -1;
    static_cast<EncHandle *>(h)->enc.analyser().speculate = level;
    return 0;
}
// one picture: packed source planes at coded size, slice_type 1 = P / 2 = I; returns the access unit's size
int orc_encoder_encode(void *hv, const uint8_t *y, const uint8_t *u, const uint8_t *v, int slice_type, uint8_t *out, size_t cap)
{
    EncHandle &H = *static_cast<EncHandle *>(hv);
    return H.enc.encode(y, u, v, slice_type, H.ev, H.ops, out, cap);
}
// the same in two steps (Encoder::submit / collect)
int orc_encoder_submit(void *hv, const uint8_t *y, const uint8_t *u, const uint8_t *v, int slice_type)
{
    EncHandle &H = *static_cast<EncHandle *>(hv);
    return H.enc.submit(y, u, v, slice_type, H.ev, H.ops);
}
int orc_encoder_collect(void *hv, uint8_t *out, size_t cap) { return static_cast<EncHandle *>(hv)->enc.collect(out, cap); }
// the last picture's decisions: arrays as orc_analysis_p_picture hands them out, plus the mode map and the rate counters
void orc_encoder_analysis(void *hv, uint16_t *info0, uint8_t *info1, uint8_t *mode, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint8_t *rec_y,
                          uint8_t *rec_u, uint8_t *rec_v, uint8_t *ctx_out, uint64_t *frac)
{
    EncHandle &H = *static_cast<EncHandle *>(hv);
    const PictureResult &r = H.enc.analysis_result();
    std::memcpy(info0, r.info0.data(), r.info0.size() * 2), std::memcpy(info1, r.info1.data(), r.info1.size()), std::memcpy(mode, r.mode.data(), r.mode.size());
    std::memcpy(coef_y, r.coef_y.data(), r.coef_y.size() * 2), std::memcpy(coef_u, r.coef_u.data(), r.coef_u.size() * 2);
    std::memcpy(coef_v, r.coef_v.data(), r.coef_v.size() * 2);
    std::memcpy(rec_y, r.rec_y.data(), r.rec_y.size()), std::memcpy(rec_u, r.rec_u.data(), r.rec_u.size()), std::memcpy(rec_v, r.rec_v.data(), r.rec_v.size());
    std::memcpy(ctx_out, r.ctx_out.data(), r.ctx_out.size());
    std::memcpy(frac, H.enc.analyser().frac_bits.data(), H.enc.analyser().frac_bits.size() * 8);
}

// the last picture's SAO parameters: out[component][ctu][8] = typeIdx, subTypeIdx, offset[4], mergeLeftFlag, mergeUpFlag
void orc_encoder_sao(void *hv, int32_t *out)
{
    const xmi::encoder::Encoder &e = static_cast<EncHandle *>(hv)->enc;
    for (int c = 0; c < 3; c++)
        for (const xmi::syntax::SaoUnit &u : e.sao_units(c)) {
            const int32_t rec[8] = { u.type_idx, u.sub_type, u.offset[0], u.offset[1], u.offset[2], u.offset[3], u.merge_left, u.merge_up };
            std::memcpy(out, rec, sizeof rec);
            out += 8;
        }
}

void orc_encoder_sao_stats(void *hv, int32_t *out)
{
    const xmi::encoder::Encoder &e = static_cast<EncHandle *>(hv)->enc;
    std::memcpy(out, e.sao_statistics().data(), e.sao_statistics().size() * sizeof(xmi::encoder::SaoStats));
}

// the last picture after deblocking, before the SAO offsets (what the SAO statistics were taken on)
void orc_encoder_deblocked(void *hv, uint8_t *y, uint8_t *u, uint8_t *v)
{
    const CpuFrameOps &o = static_cast<EncHandle *>(hv)->ops;
    std::memcpy(y, o.plane[0].data(), o.plane[0].size()), std::memcpy(u, o.plane[1].data(), o.plane[1].size()), std::memcpy(v, o.plane[2].data(), o.plane[2].size());
}

void orc_encoder_reconstruction(void *hv, uint8_t *y, uint8_t *u, uint8_t *v)
{
    const xmi::encoder::Encoder &e = static_cast<EncHandle *>(hv)->enc;
    std::memcpy(y, e.rec_y().data(), e.rec_y().size()), std::memcpy(u, e.rec_u().data(), e.rec_u().size()), std::memcpy(v, e.rec_v().data(), e.rec_v().size());
}

const uint16_t *orc_analysis_scan(int log2_size) { return diag_scan(log2_size); }
const uint16_t *orc_syntax_scan(int scan_idx, int log2_size) { return xmi::syntax::coef_scan(scan_idx, log2_size); }
const uint8_t *orc_syntax_group_scan(int scan_idx, int log2_size) { return xmi::syntax::group_scan(scan_idx, log2_size); }
const uint8_t *orc_analysis_next_state(void) { return next_state_table(); }
void orc_analysis_init_contexts(int slice_type, int qp, uint8_t *ctx) { init_contexts(slice_type, qp, ctx); }

}  // extern "C"
