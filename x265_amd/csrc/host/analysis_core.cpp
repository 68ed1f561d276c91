// analysis_core.cpp - see analysis_core.h. Host logic only: no pixel arithmetic, no device code.
//
// Structure: one host thread per CTU row runs the decision recursion in plain sequential form and blocks whenever it
// needs pixels (a CuEval or the ChromaEval list of a finished CTU) or a neighbour (the CTU above and to the right).
// When every row thread is blocked, the calling thread hands all pending requests to the evaluator in one call, then
// wakes the rows. The number of evaluator calls per picture is therefore the length of the critical path through the
// wavefront, not the number of CUs.
#include "analysis_core.h"
#include "syntax.h"

#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <fstream>
#include <memory>
#include <string>
#include <thread>

#include <sched.h>
#include <ucontext.h>
#if defined(__SANITIZE_THREAD__)
// ThreadSanitizer follows hand-made context switches only when told: its fiber interface (sanitizer/tsan_interface.h)
extern "C" {
void *__tsan_get_current_fiber(void);
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);
}
#define XMI_TSAN_FIBER(stmt) stmt
#else
#define XMI_TSAN_FIBER(stmt)
#endif

namespace xmi {
namespace analysis {
namespace {

using namespace xmi::syntax;

constexpr int kInvalid = 255, kValid = 0;                 // MODE_INVALID / MODE_VALID (inc/x265.h:163)
constexpr int kFlagInter = 0x80, kOffCbfC = 8;            // ausModeYInfo0 bits (inc/x265.h:216-217)
constexpr int kOffSize = 6, kOffCbfY = 5, kOff64 = 4, kMaskSkip = 1, kValidT = 3, kValidL = 1;   // aucModeYInfo1 bits
constexpr double kDoubleMax = 1.7976931348623158e+308;

// Context initialisation values of P and I slices in context order (src/bitstream.cpp:371-564, rows [1] and [2] of each
// table; 154 = "not used"). xCabacInit turns them into states for the slice QP.
const uint8_t kInitP[184] = {
    107, 139, 126, 197, 185, 201, 110, 122, 154, 139, 154, 154, 154, 149, 154, 152, 139, 95, 79, 63, 31, 31, 140, 198, 153, 153, 154, 154, 154,
    153, 111, 154, 154, 154, 149, 107, 167, 154, 154, 79, 121, 140, 61, 154,
    155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123,
    107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
    125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
    125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
    154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
    107, 167, 91, 122, 107, 167, 168, 154, 124, 138, 94, 139, 153, 185
};
const uint8_t kInitI[184] = {
    139, 141, 157, 154, 154, 154, 154, 154, 184, 154, 154, 154, 154, 154, 184, 63, 139, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
    111, 141, 154, 154, 154, 94, 138, 182, 154, 154, 154, 91, 171, 134, 141,
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 140, 139, 182, 182,
    152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
    138, 153, 136, 167, 152, 152, 154, 154, 153, 138, 138, 139, 153, 200
};

// Fractional bit cost (15 fractional bits) of coding a bin in a given state: index = state ^ bin (src/encode.cpp:131-142,
// the table of the HEVC test model).
const int32_t kEntropyBits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb
};

// CABAC state transitions (state = 2 * probability index + MPS): the standard's transIdxLps table; the MPS step is +1
// up to probability index 62.
const uint8_t kLpsNext[64] = { 0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                               18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                               31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

struct Tables {
    uint8_t next[128][2];
    uint16_t scan[3][4][1024];     // [diagonal / horizontal / vertical][log2 size 2..5]
    uint8_t group[3][4][64];       // the same orders over the 4x4 groups of the block
    Tables()
    {
        for (int s = 0; s < 128; s++) {
            const int p = s >> 1, mps = s & 1;
            const int up = p < 62 ? p + 1 : p;
            next[s][mps] = (uint8_t)(2 * up + mps);
            next[s][mps ^ 1] = p == 0 ? (uint8_t)(mps ^ 1) : (uint8_t)(2 * kLpsNext[p] + mps);
        }
        next[126][0] = next[126][1] = 126, next[127][0] = next[127][1] = 127;   // the terminating state does not move
        // order of an n x n square: up-right diagonal (along every anti-diagonal from bottom-left to top-right), row by row,
        // column by column
        auto order = [](int kind, int n, uint8_t *o) {
            int k = 0;
            if (kind == kScanDiag) {
                for (int d = 0; d < 2 * n - 1; d++)
                    for (int y = d < n ? d : n - 1; y >= 0 && d - y < n; y--) o[k++] = (uint8_t)(y * n + (d - y));
            } else {
                for (int i = 0; i < n * n; i++) o[k++] = (uint8_t)(kind == kScanHor ? i : (i % n) * n + i / n);
            }
        };
        for (int kind = 0; kind < 3; kind++) {
            uint8_t in4[16];
            order(kind, 4, in4);
            for (int l = 2; l <= 5; l++) {
                const int n = 1 << l, g = n / 4;
                uint8_t *go = group[kind][l - 2];
                if (g == 1) go[0] = 0;
                else order(kind, g, go);
                int k = 0;
                for (int gi = 0; gi < g * g; gi++) {
                    const int gy = go[gi] / g, gx = go[gi] % g;
                    for (int i = 0; i < 16; i++) scan[kind][l - 2][k++] = (uint16_t)((gy * 4 + in4[i] / 4) * n + gx * 4 + in4[i] % 4);
                }
            }
        }
    }
};
const Tables &tables()
{
    static const Tables t;
    return t;
}

struct Ctx {
    uint8_t m[256];
};

// Rate counter of one CTU row (xRdoParam, inc/x265.h:354): the fraction of a bit left over from the previous estimate is
// kept (frac &= 32767 at the start of every estimate, src/encode.cpp:189), and the working contexts are cabacRdo[y][0] -
// the same memory as the depth-0 contexts (rdoParam[y].pCabacRdo = cabacRdo[y], src/encode.cpp:5263).
struct Rate {
    uint64_t frac = 0;
    bool transitions = false;   // only row 0 has a next-state table (xBuildNextStateTable(h->rdoParam), src/encode.cpp:5052); the
                                // other rows' tables are zero-filled, so every context they touch falls to state 0
    Ctx *w = nullptr;
    void bin(unsigned b, int c)
    {
        const uint8_t s = w->m[c];
        frac += (uint64_t)kEntropyBits[s ^ b];
        w->m[c] = transitions ? tables().next[s][b] : 0;
    }
    void ep(unsigned, int n) { frac += 32768ull * (unsigned)n; }
    void begin() { frac &= 32767; }
    uint32_t bits() const { return (uint32_t)(frac >> 15); }
};

// Bits of the luma part of one CU as xPreEncRec counts them (src/encode.cpp:3038-3077): prediction mode, part size, the
// coded-block flag of each transform unit and its coefficients, and the root flag of a 64x64 CU.
uint32_t rate_cu_luma(Rate &r, const int16_t *level, int size, const uint32_t sum[4], int *cbf)
{
    r.begin();
    r.bin(0, kCtxPredMode);
    r.bin(1, kCtxPartSize);
    if (size == 64) {
        *cbf = 0;
        uint32_t total = 0;
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) {
                const uint32_t s = sum[j * 2 + i];
                *cbf |= (s ? 1 : 0) << (j * 2 + i);
                total += s;
                r.bin(s != 0, kCtxQtCbf);
                if (s) code_coefficients(r, level + j * 32 * 64 + i * 32, 64, 32, true, kScanDiag);
            }
        r.bin(total != 0, kCtxQtRootCbf);
    } else {
        *cbf = sum[0] ? 1 : 0;
        r.bin(sum[0] != 0, kCtxQtCbf);
        if (sum[0]) code_coefficients(r, level, size, size, true, kScanDiag);
    }
    return r.bits();
}

// motion vector difference and predictor index (xWriteMVD, xWriteUnaryMaxSymbol with one candidate beyond the first;
// src/encode.cpp:4482-4523, 2357-2382)
void rate_mvd(Rate &r, int dx, int dy, int mvp_idx)
{
    code_mvd(r, dx, dy);
    r.bin(mvp_idx ? 1 : 0, kCtxMvpIdx);
}

struct Shared;

// ---- one CTU row ----
struct Row {
    Shared *sh = nullptr;
    int y = 0;
    // CTU-local caches with the reference's borders: validity (MAX_PU_XY + 2)^2 with a border on every side, info1
    // (MAX_PU_XY + 1)^2 with a top / left border, vectors (MAX_MV_XY + 2)^2 (inc/x265.h:209-261)
    uint8_t mode[18 * 18];
    uint16_t info0[16 * 16];
    uint8_t info1[17 * 17];
    Mv mv[10 * 10], mvd[8 * 8];
    uint8_t mvp_idx[8 * 8];
    uint8_t rec_y[64 * 64], rec_u[32 * 32], rec_v[32 * 32];
    int16_t coef_y[64 * 64], coef_u[32 * 32], coef_v[32 * 32];
    // best candidate of the CU being decided, per CU size 8/16/32/64 (paucRecYTmp, pasCoefYTmp, paiCbfYTmp)
    uint8_t best_rec[4][64 * 64];
    int16_t best_coef[4][64 * 64];
    int best_cbf[4];
    Ctx cab[5];                 // cabacRdo[y][0..4]; [0] doubles as the working set
    Rate rate;
    Mv min_ipel, max_ipel;
    double lambda = 0;
    int qp = 0, ctu_x = 0;
    // Requests per CU size 8 / 16 / 32 / 64. A CU's request takes the requests of its first child, that child's first child
    // and so on along: they sit at the same corner, so their neighbours are the CU's own and nothing the CU decides
    // changes what they ask for. A child that finds its answer waiting (spec_ok) saves the round trip.
    // With tree speculation the request also takes every CU that may still be decided in this CTU - the rest of the CU's
    // own quadtree and of its parents' - asked for as if every CU decided from here on ended up with one guessed vector
    // (the first merge candidate of the CU that has to ask): their neighbours are then known. Whoever later finds that its
    // real neighbours are the guessed ones (same_request) has its answer waiting; where motion is coherent a CTU takes one
    // round trip. Answers live until the row's next call.
    CuEval tree[85];            // 64 CUs of 8x8, 16 of 16x16, 4 of 32x32, the 64x64 CU
    bool tree_ok[85] = {};
    std::vector<CuEval *> spec_list;
    static int node(int li, int px, int py)
    {
        static const int first[4] = { 0, 64, 80, 84 };
        const int log2 = li + 3;
        return first[li] + (py >> log2) * (64 >> log2) + (px >> log2);
    }
    void fill_request(CuEval &r, const Mv *cnd, int nm, int px, int py, int sz);
    void spec_node(int px, int py, int size);
    void spec_subtree(int px, int py, int size, Mv g, int floor);
    void mark_decided(int px, int py, int size, Mv g);
    bool guessed = false;       // the row's last call carried requests made under a guess ...
    int guess_hits = 0;         // ... and so many of them have been of use since
    int barren = 0;             // calls in a row whose guess helped nobody
    int asked_px = -1, asked_py = -1;
    // host time of the row between two requests (nanoseconds) and the number of requests: what a round trip waits for
    // besides the evaluator
    uint64_t busy_ns = 0, requests = 0;
#ifdef XMI_EXP_HOST_TIMING
    uint64_t cat[4] = {};
    static uint64_t tick() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif
    std::chrono::steady_clock::time_point t_resume = std::chrono::steady_clock::now();
    std::vector<ChromaEval> chroma_req;
    std::vector<int> chroma_list;      // the CUs chroma_req stands for: px | py << 8 | size << 16
    int chroma_waiting = -1;           // CTU whose chroma requests travel with the next CTU's first luma request, or -1
    // I pictures: the unfiltered reconstruction next to the CTU - the row above from x0 - 1 to x0 + 127 and the column to the
    // left (the reference's aucTopPix line buffers and the left column of its CTU cache) -, the requests per PU size
    // 4 / 8 / 16 / 32 (a PU's request carries those of its chain of first children, as for inter CUs) and the chroma request
    uint8_t top_y[1 + 128], left_y[64], top_u[1 + 64], left_u[32], top_v[1 + 64], left_v[32];
    IntraEval in_spec[4];
    bool in_ok[4] = { false, false, false, false };
    IntraChromaEval ichroma;
    IntraQuadEval quad;             // the four 4x4 PUs under the 8x8 PU of the last chain
    bool quad_ok = false;
    uint8_t quad_rec[16];           // the PU of them that is being decided, packed
    int16_t quad_level[16];
    IntraChromaQuadEval cquad;
    // The 8x8 block to the right of the one a request ends at is asked for in the same call, twice: as it will be when the
    // block to its left is kept as one 8x8 PU ([0]) and as four 4x4 PUs ([1]); the evaluator takes that block's right column
    // and modes from the call's own results (IntraEval::after_kind). Which premise holds is this row's cost comparison; the
    // answers wait until the block comes up, and its round trip is saved.
    IntraEval pair_in[2];
    IntraQuadEval pair_quad[2];
    bool pair_ok = false, req_pair = false;
    uint64_t pairs_taken = 0;
    int pair_x = 0, pair_y = 0;
    void fill_pair(int px, int py);
    bool take_pair(IntraEval &q, int px, int py);
    // I pictures: the chroma PUs of a decided CTU are a chain of requests of their own - nothing in it depends on a cost, and
    // nothing the luma decisions of the next CTU read depends on it (the estimation contexts pass through luma syntax only) -,
    // so its links travel with the luma requests of the CTU to the right. What the chain reads of its CTU is kept here.
    int ic_ctu = -1;                // CTU whose chroma chain is under way, or -1
    int ic_idx = 0;                 // next unit of it in coding order (4x4 units, 256 = through)
    bool ic_begun = false;          // neighbouring samples gathered (the row above has come far enough)
    int ic_dx = 0, ic_dy = 0, ic_size_y = 0;
    uint8_t ic_info1[16 * 16], ic_mode[16 * 16];
    bool chroma_link_fill();        // the next request of the chain into ichroma / cquad; false when it cannot go yet
    void chroma_link_take();
    void chroma_chain_arm(int xp);
    void chroma_chain_drain();
    void wait_row_above(int ctus, bool whole);
    void luma_done(int xp);
    IntraEval *req_in[4] = {};
    int n_req_in = 0;
    bool req_ichroma = false, req_quad = false, req_cquad = false;
    int strong = 0;
    // The request hand-over takes no lock: the row fills its own slots, raises `pending` and counts itself out of its
    // lane's `running`; the lane's coordinator reads the slots of the rows it finds pending, and lowers `pending` when the
    // answers are in place. Only a waiter whose polling time has run out takes the mutex, to sleep.
    std::atomic<bool> pending{ false };
    std::atomic<bool> in_flight{ false };      // ... that a coordinator has taken into the call it is making
    int served_by = -1;         // lane whose buffers hold the answers the row is working on
    void release_answers();
    CuEval *req_cu[85] = {};
    int n_req_cu = 0;
    bool req_chroma = false;     // ... that is part of the call its lane is making right now
    std::atomic<int> need{ -1 };     // CTUs of the row above this row waits for, or -1 (changed under the mutex, polled without)
    bool need_whole = true;          // ... complete (I pictures: with their chroma), or their luma alone
    // A row runs as a fiber of a worker thread (rows y, y + workers, ... share one): a picture has more CTU rows than a host
    // has cores to poll on, and a row that waits - for answers, for the row above - hands its thread to the next one.
    enum Wait { kReady, kAnswers, kRowAbove };
    ucontext_t fiber, *home = nullptr;
    void *san_fiber = nullptr, *san_home = nullptr;      // the sanitizer's handles of the two contexts (ThreadSanitizer builds only)
    std::unique_ptr<char[]> stack;
    Wait wait = kReady;
    bool started = false, done = false;
    bool runnable() const
    {
        return wait == kReady || (wait == kAnswers && !pending.load(std::memory_order_acquire)) ||
               (wait == kRowAbove && need.load(std::memory_order_acquire) < 0);
    }
    void block(Wait why)      // back to the worker until the reason has gone
    {
        wait = why;
        XMI_TSAN_FIBER(__tsan_switch_to_fiber(san_home, 0));
        swapcontext(&fiber, home);
        wait = kReady;
    }

    uint8_t &M(int r, int c) { return mode[(r + 1) * 18 + c + 1]; }
    uint8_t &I1(int r, int c) { return info1[(r + 1) * 17 + c + 1]; }
    uint16_t &I0(int r, int c) { return info0[r * 16 + c]; }
    Mv &V(int r, int c) { return mv[(r + 1) * 10 + c + 1]; }

    int fill_candidates(Mv out[7], int r4, int c4, int r8, int c8, int size);
    uint32_t decide(int px, int py, int size, int depth);
    void chroma_prepare();
    void chroma_apply(int xp);
    void ctu_done(int xp);
    int sample_y(int yy, int xx) const
    {
        if (yy < 0) return top_y[xx + 1];
        if (xx < 0) return yy < 64 ? left_y[yy] : 0x80;
        return xx < 64 && yy < 64 ? rec_y[yy * 64 + xx] : 0x80;
    }
    int sample_c(const uint8_t *top, const uint8_t *left, const uint8_t *rec, int yy, int xx) const
    {
        if (yy < 0) return top[xx + 1];
        if (xx < 0) return yy < 32 ? left[yy] : 0x80;
        return xx < 32 && yy < 32 ? rec[yy * 32 + xx] : 0x80;
    }
    void fill_intra(IntraEval &q, int px, int py, int size);
    double decide_intra(int px, int py, int size, int depth);

    void collect_cus(int px, int py, int size, std::vector<int> &list);
    void run();
    void evaluate_cus(CuEval *const *list, int n);
    void evaluate_chroma();
    void evaluate_intra(IntraEval *const *list, int n, bool with_quad, bool with_pair = false);
    void fill_quad(IntraQuadEval &q, int px, int py);
    void post_and_wait();
    static void fiber_main(unsigned lo, unsigned hi);
    void make_fiber();
};

// A lane is a coordinator with evaluator resources of its own. All lanes serve all rows: a coordinator takes the requests that
// are posted when it looks, sends them to the evaluator in one call and hands the answers back; with more than one, the rows
// that were still busy on the host when one call left go out with the next coordinator's call while the first is under
// way, so a row's cycle is the evaluator's latency plus its own host work, not the slowest row's. The counters of the
// hand-over are lane 0's. The answers of a call live in the lane's buffers until its next call: a coordinator does not
// call again while a row still reads (readers).
struct Lane {
    std::condition_variable cv_main;
    std::atomic<int> running{ 0 };      // rows that are busy on the host
    std::atomic<int> posted{ 0 };       // rows with a request no coordinator has picked up yet
    std::atomic<int> finished{ 0 };
    std::atomic<int> readers{ 0 };      // rows still working on the answers of this lane's last call
    int rows = 0;
};

struct Shared {
    std::mutex m;
    std::condition_variable cv_rows;    // workers none of whose rows could go on for their whole polling time sleep here
    std::atomic<int> sleepers{ 0 };     // workers asleep on cv_rows: a round that finds none wakes nobody (no system call)
    std::atomic<int> error{ 0 };
    int spin_us = 0;                // how long a waiter polls before it blocks
    int grace_us = 6;               // how long a polling coordinator waits for the rows still busy before it goes without them
    int speculate = 2;              // 1: a CU's request carries those of its chain of first children; 2: of the rest of the CTU
    bool defer_chroma = true;       // a CTU's chroma requests travel with the next CTU's first luma request
    uint16_t *ctu_requests = nullptr;      // measurement (XMI_EXP_CTU_REQUESTS): requests a row made while it decided each CTU
    bool pairs = true;              // I pictures: an 8x8 block's request takes the block to its right along (Row::fill_pair)
    std::deque<Lane> lanes;
    std::vector<int> done;          // CTUs finished per row (under the mutex)
    std::vector<int> done_luma;     // I pictures: CTUs whose luma is decided (their chroma may still be under way)
    std::unique_ptr<std::atomic<int>[]> done_seen;      // `done`, for a look without the mutex
    std::deque<Row> rows;
    std::vector<Ctx> saved;         // contexts after CTU 1 of a row: where the row below starts (cabacRdoSaved, src/encode.cpp:5481)
    Ctx start;                      // contexts of the slice start
    Analyser *an = nullptr;
    PictureResult *out = nullptr;
    int w = 0, h = 0, cols = 0, nrows = 0, qp = 0;
    bool intra = false;             // I picture
    int strong = 0;                 // its bStrongIntraSmoothing
    Evaluator *ev = nullptr;
    void fail(int rc);              // an error that a row found ends the picture like one that a coordinator found
};

struct Abort {};

void Shared::fail(int rc)
{
    std::lock_guard<std::mutex> lk(m);
    if (!error) error = rc;
    Lane &ln = lanes[0];
    for (Row &r : rows)
        if (r.need.load() >= 0) ln.running.fetch_add(1), r.need.store(-1);
    ln.cv_main.notify_all();
    cv_rows.notify_all();
}

// xFillMvpCand (src/encode.cpp:2703-2809) for a square 2Nx2N PU: out[0..1] the two AMVP predictors, out[2..] the merge
// candidates; returns their number.
int Row::fill_candidates(Mv out[7], int r4, int c4, int r8, int c8, int size)
{
    const int b4 = size / 4, b8 = size / 8;
    const bool bl = M(r4 + b4, c4 - 1) != kInvalid, l = M(r4 + b4 - 1, c4 - 1) != kInvalid;
    const bool tr = M(r4 - 1, c4 + b4) != kInvalid, t = M(r4 - 1, c4 + b4 - 1) != kInvalid, tl = M(r4 - 1, c4 - 1) != kInvalid;
    const Mv vbl = V(r8 + b8, c8 - 1), vl = V(r8 + b8 - 1, c8 - 1), vtr = V(r8 - 1, c8 + b8), vt = V(r8 - 1, c8 + b8 - 1), vtl = V(r8 - 1, c8 - 1);
    auto same = [](Mv a, Mv b) { return a.x == b.x && a.y == b.y; };
    const Mv zero = { 0, 0 };
    int n = 0;
    out[0] = zero;
    if (bl || l) out[n++] = bl ? vbl : vl;
    if (tr || t || tl) out[n++] = tr ? vtr : t ? vt : vtl;
    if ((bl || l) && (tr || t || tl) && same(out[0], out[1])) out[1] = zero;
    if (n < 2) out[1] = zero;
    n = 0;
    Mv *mg = out + 2;
    if (l) mg[n++] = vl;
    if (t && (!l || !same(vl, vt))) mg[n++] = vt;
    if (tr && (!t || !same(vt, vtr))) mg[n++] = vtr;
    if (bl && (!l || !same(vl, vbl))) mg[n++] = vbl;
    if (n < 4) {
        if (tl && (!l || !same(vl, vtl)) && (!t || !same(vt, vtl))) {
            mg[n++] = vtl;
        } else {
            bool has_zero = false;
            for (int i = 0; i < n; i++) has_zero |= same(mg[i], zero);
            if (!has_zero) mg[n++] = zero;
        }
    }
    return n;
}

void block_in(void *dst, int dst_pitch, const void *src, int src_pitch, int rows, int bytes)
{
    for (int i = 0; i < rows; i++) std::memcpy((char *)dst + (size_t)i * dst_pitch, (const char *)src + (size_t)i * src_pitch, bytes);
}

#ifdef XMI_EXP_MISS_STATS
std::atomic<long> g_miss[4][2], g_miss_what[8], g_hit[4];
#endif
void Row::fill_request(CuEval &r, const Mv *cnd, int nm, int px, int py, int sz)
{
    const int qmin_x = 4 * (min_ipel.x - 4), qmin_y = 4 * (min_ipel.y - 4), qmax_x = 4 * (max_ipel.x - 4), qmax_y = 4 * (max_ipel.y - 4);
    r.x0 = ctu_x * 64 + px, r.y0 = y * 64 + py, r.size = sz;
    r.n_merge = nm;
    for (int k = 0; k < nm; k++) {
        r.merge_mv[k] = cnd[2 + k];
        r.merge_ok[k] = cnd[2 + k].x >= qmin_x && cnd[2 + k].x <= qmax_x && cnd[2 + k].y >= qmin_y && cnd[2 + k].y <= qmax_y;
        // a vector that an earlier usable candidate already has cannot win (the first smallest is taken): not evaluated
        for (int e = 0; e < k && r.merge_ok[k]; e++)
            if (r.merge_ok[e] && cnd[2 + e].x == cnd[2 + k].x && cnd[2 + e].y == cnd[2 + k].y) r.merge_ok[k] = 0;
    }
    r.amvp[0] = cnd[0], r.amvp[1] = cnd[1];
    r.mv_min_ipel = min_ipel, r.mv_max_ipel = max_ipel;
    r.lambda = lambda, r.qp = qp;
    r.urgent = 0;
    r.merge_pending = r.search_pending = nullptr;
}

// the request of the CU at (px, py) as the caches stand
void Row::spec_node(int px, int py, int size)
{
    Mv c[7];
    const int nm = fill_candidates(c, py / 4, px / 4, py / 8, px / 8, size);
    const int idx = node(ilog2(size) - 3, px, py);
    fill_request(tree[idx], c, nm, px, py, size);
    tree_ok[idx] = true;
    spec_list.push_back(&tree[idx]);
}

// requests for the CUs below the one at (px, py) down to size `floor`, in coding order, each seeing the CUs before it decided
// with vector g; afterwards the CU's own area stands in the caches as decided with g
void Row::spec_subtree(int px, int py, int size, Mv g, int floor)
{
    if (size / 2 >= floor && size > 8) {
        const int half = size / 2;
        for (int part = 0; part < 4; part++) {
            const int cx = px + (part & 1) * half, cy = py + (part >> 1) * half;
            spec_node(cx, cy, half);
            spec_subtree(cx, cy, half, g, floor);
        }
        return;      // the four children have covered the area
    }
    mark_decided(px, py, size, g);
}

void Row::mark_decided(int px, int py, int size, Mv g)
{
    for (int i = 0; i < size / 4; i++)
        for (int j = 0; j < size / 4; j++) M(py / 4 + i, px / 4 + j) = kValid;
    for (int i = 0; i < size / 8; i++)
        for (int j = 0; j < size / 8; j++) V(py / 8 + i, px / 8 + j) = g;
}

// Levels of an n x n block whose transform units are `units` (1, or the 4 quadrants): an evaluator leaves the levels of a
// unit undefined when their sum is zero (CuEval, ChromaEval), so those are not read.
void take_levels(int16_t *dst, int dst_stride, const int16_t *src, int n, const uint32_t *sums, int units)
{
    const int un = units == 4 ? n / 2 : n;
    for (int u = 0; u < units; u++) {
        const int ox = units == 4 ? (u & 1) * un : 0, oy = units == 4 ? (u >> 1) * un : 0;
        for (int i = 0; i < un; i++) {
            int16_t *d = dst + (size_t)(oy + i) * dst_stride + ox;
            if (sums[u]) std::memcpy(d, src + (size_t)(oy + i) * n + ox, (size_t)un * 2);
            else std::memset(d, 0, (size_t)un * 2);
        }
    }
}

// xEncDecideCU_Inter (src/encode.cpp:3080-3639), FAST_MD build: returns the RD cost as the reference's callers see it,
// truncated to 32 bits by the function's declared return type.
uint32_t Row::decide(int px, int py, int size, int depth)
{
    const int log2 = ilog2(size), b4 = size / 4, b8 = size / 8, r4 = py / 4, c4 = px / 4, r8 = py / 8, c8 = px / 8, li = log2 - 3;
    const Ctx entry = cab[depth];
    Ctx &work = cab[0];
    Mv cand[7];
    const int nmerge = fill_candidates(cand, r4, c4, r8, c8, size);
    const Mv *mg = cand + 2;

    const int idx = node(li, px, py);
    CuEval &q = tree[idx];
    auto same_request = [&](const CuEval &r) {      // was the waiting answer asked for with what this CU sees now?
        if (r.x0 != ctu_x * 64 + px || r.y0 != y * 64 + py || r.size != size || r.n_merge != nmerge) return false;
        for (int k = 0; k < nmerge; k++)
            if (r.merge_mv[k].x != mg[k].x || r.merge_mv[k].y != mg[k].y) return false;
        return r.amvp[0].x == cand[0].x && r.amvp[0].y == cand[0].y && r.amvp[1].x == cand[1].x && r.amvp[1].y == cand[1].y;
    };
    if (!(tree_ok[idx] && same_request(q))) {
#ifdef XMI_EXP_MISS_STATS
        g_miss[li][tree_ok[idx] ? 1 : 0]++;      // asked under a guess that did not come true | not asked for at all
        if (tree_ok[idx]) {
            int what = 0;      // which part of the request differs: 1 merge count, 2 a merge vector, 4 the AMVP pair
            if (q.n_merge != nmerge) what |= 1;
            for (int k = 0; k < nmerge && k < q.n_merge; k++)
                if (q.merge_mv[k].x != mg[k].x || q.merge_mv[k].y != mg[k].y) what |= 2;
            if (q.amvp[0].x != cand[0].x || q.amvp[0].y != cand[0].y || q.amvp[1].x != cand[1].x || q.amvp[1].y != cand[1].y) what |= 4;
            g_miss_what[what]++;
        }
#endif
#ifdef XMI_EXP_HOST_TIMING
        const uint64_t ts0 = tick();
#endif
        std::memset(tree_ok, 0, sizeof tree_ok);      // a new call: the answers of the last one are gone
        spec_list.clear();
        fill_request(q, cand, nmerge, px, py, size);
        q.urgent = 1;      // the answer this row reads first
        spec_list.push_back(&q);
        if (sh->speculate == 1) {
            for (int sz = size / 2; sz >= 8; sz /= 2) spec_node(px, py, sz);      // same corner: the CU's own neighbours
        } else if (sh->speculate >= 2) {
            uint8_t keep_mode[sizeof mode];
            Mv keep_mv[sizeof mv / sizeof mv[0]];
            std::memcpy(keep_mode, mode, sizeof mode), std::memcpy(keep_mv, mv, sizeof mv);
            const Mv g = nmerge ? mg[0] : cand[0];
            // A call whose guess helped nobody is followed by a modest one: the CU's chain of first children and the later
            // children of its parent with theirs. Where a guess comes true again, the next call takes the rest of the CTU.
            if (guessed) barren = guess_hits > 0 ? 0 : barren + 1;
            const bool wide = barren < 2 || size == 64;
            // Most CTUs are decided at 64x64 .. 16x16: 8x8 CUs are asked for ahead of time only next to a CU of 16x16 or
            // 8x8 that has to ask itself
            const int near_floor = size <= 16 ? 8 : 16;
            if (wide) {
                spec_subtree(px, py, size, g, near_floor);
            } else {
                for (int sz = size / 2; sz >= 8; sz /= 2) spec_node(px, py, sz);
                mark_decided(px, py, size, g);
            }
            for (int cx = px, cy = py, sz = size; sz < 64; sz *= 2) {      // the later children of every parent
                const int ox = cx & ~(2 * sz - 1), oy = cy & ~(2 * sz - 1);
                const int floor = sz <= 16 && size <= 16 ? 8 : 16;
                for (int part = ((cy - oy) / sz) * 2 + (cx - ox) / sz + 1; part < 4; part++) {
                    const int nx = ox + (part & 1) * sz, ny = oy + (part >> 1) * sz;
                    spec_node(nx, ny, sz);
                    if (wide) {
                        spec_subtree(nx, ny, sz, g, floor);
                    } else {
                        for (int s2 = sz / 2; s2 >= 8; s2 /= 2) spec_node(nx, ny, s2);
                        mark_decided(nx, ny, sz, g);
                    }
                }
                if (!wide) break;
                cx = ox, cy = oy;
            }
            std::memcpy(mode, keep_mode, sizeof mode), std::memcpy(mv, keep_mv, sizeof mv);
            guessed = spec_list.size() > 1;
            guess_hits = 0;
            asked_px = px, asked_py = py;
        }
#ifdef XMI_EXP_HOST_TIMING
        cat[1] += tick() - ts0;
#endif
        evaluate_cus(spec_list.data(), (int)spec_list.size());
    } else if (px != asked_px || py != asked_py) {
        guess_hits++;      // not one of the asking CU's first children, whose neighbours were never in question
#ifdef XMI_EXP_MISS_STATS
        g_hit[li]++;
#endif
    }
    tree_ok[idx] = false;
    if (q.merge_pending) {
        const int rc = sh->ev->finish_merge(q);
        if (rc) {
            sh->fail(rc);
            throw Abort();
        }
    }
    // read after the evaluation: the call may have brought the chroma of the CTU to the left, which sets its skip flags
    const bool skip_left = (I1(r4, c4 - 1) & kMaskSkip) && M(r4 + b4 - 1, c4 - 1) != kInvalid;
    const bool skip_above = (I1(r4 - 1, c4) & kMaskSkip) && M(r4 - 1, c4 + b4 - 1) != kInvalid;
    const int ctx_skip = kCtxSkip + (skip_left ? 1 : 0) + (skip_above ? 1 : 0);

    double best = kDoubleMax;
    int merge_idx = 0, skip = 0;
    Mv bmv = { 0, 0 };
    bool go_me = true;
    if (q.best_merge >= 0) {
        const int k = q.best_merge;
        work = entry;
        int cbf = 0;
        const uint32_t coeff_bits = rate_cu_luma(rate, q.level_merge, size, q.sum_merge, &cbf);
        rate.begin();
        rate.bin(1, kCtxMergeFlag);
        rate.bin(k != 0, kCtxMergeIdx);
        if (k) rate.ep(0, k);
        const uint32_t merge_bits = rate.bits();
        double cost = q.ssd_merge + lambda * (coeff_bits + merge_bits);
        const double cost_skip = q.ssd_skip + lambda * merge_bits;
        int is_skip = 0;
        if (cost_skip < cost) {
            is_skip = 1;
            cost = cost_skip;
            work = entry;
        }
        if (cost < best) {
            best = cost;
            merge_idx = k + 1;
            skip = is_skip;
            cab[depth] = work;
            bmv = mg[k];
            if (skip) {
                // the prediction: per transform unit, the evaluator's copy of it where the unit has levels and the
                // reconstruction - the same samples - where it has none
                if (size == 64) {
                    for (int u = 0; u < 4; u++) {
                        const uint8_t *from = q.sum_merge[u] ? q.pred_merge : q.rec_merge;
                        const size_t o = (size_t)(u >> 1) * 32 * 64 + (u & 1) * 32;
                        for (int r = 0; r < 32; r++) std::memcpy(best_rec[li] + o + (size_t)r * 64, from + o + (size_t)r * 64, 32);
                    }
                } else {
                    std::memcpy(best_rec[li], q.sum_merge[0] ? q.pred_merge : q.rec_merge, (size_t)size * size);
                }
                std::memset(best_coef[li], 0, (size_t)size * size * 2);
                best_cbf[li] = 0;
            } else {
                std::memcpy(best_rec[li], q.rec_merge, (size_t)size * size);
                take_levels(best_coef[li], size, q.level_merge, size, q.sum_merge, size == 64 ? 4 : 1);
                best_cbf[li] = cbf;
            }
        }
        if (skip || cbf == 0) go_me = false;      // FAST_MD: a skipped or residual-free merge ends the CU's own search (:3370-3374)
    }
    if (go_me) {
        if (q.search_pending) {
            const int rc = sh->ev->finish_search(q);
            if (rc) {
                sh->fail(rc);
                throw Abort();
            }
        }
        const Mv m = q.mv_me;
        work = entry;
        int cbf = 0;
        const uint32_t coeff_bits = rate_cu_luma(rate, q.level_me, size, q.sum_me, &cbf);
        int idx = 0;
        while (idx < nmerge && !(m.x == mg[idx].x && m.y == mg[idx].y)) idx++;
        const int merge_found = idx != nmerge ? idx + 1 : 0;
        uint32_t mvd_bits = 0, merge_bits = 0;
        if (merge_found) {
            merge_bits = idx == 4 ? idx : idx + 1;
        } else {
            int pi = 0;
            if (std::abs(cand[0].x - cand[1].x) + std::abs(cand[0].y - cand[1].y) > 0) {
                const unsigned d0 = std::abs(cand[0].x - m.x) + std::abs(cand[0].y - m.y);
                const unsigned d1 = std::abs(cand[1].x - m.x) + std::abs(cand[1].y - m.y) + 1;
                pi = d0 <= d1 ? 0 : 1;
            }
            rate.begin();
            rate_mvd(rate, (int16_t)(cand[pi].x - m.x), (int16_t)(cand[pi].y - m.y), pi);
            mvd_bits = rate.bits();
        }
        const double cost = q.ssd_me + lambda * (mvd_bits + merge_bits + coeff_bits);
        if (cost < best) {
            best = cost;
            merge_idx = merge_found;
            skip = 0;
            bmv = m;
            rate.bin(0, ctx_skip);                 // the skip flag joins the estimate here without a fresh start of the counter (:3452)
            cab[depth] = work;
            std::memcpy(best_rec[li], q.rec_me, (size_t)size * size);
            take_levels(best_coef[li], size, q.level_me, size, q.sum_me, size == 64 ? 4 : 1);
            best_cbf[li] = cbf;
        }
    }

    bool replace = size == 8 || skip;
    if (size > 8 && !skip) {
        Analyser &an = *sh->an;
        const int cols = sh->cols;
        auto at = [&](int ry, int rx, int d) { return ((size_t)ry * cols + rx) * 5 + d; };
        bool descend = true;
        if (depth != 0) {
            // cost statistics of this CTU and of its left / upper neighbours at this depth: a CU already cheaper than their
            // weighted average is not split (:3476-3508)
            const double own_cost = an.avg_cost[at(y, ctu_x, depth)], own_n = an.count[at(y, ctu_x, depth)];
            double nb_cost = 0;
            int nb_n = 0;
            auto add = [&](int ry, int rx) { nb_cost += an.avg_cost[at(ry, rx, depth)], nb_n += an.count[at(ry, rx, depth)]; };
            if (y > 0) add(y - 1, ctu_x);
            if (y > 0 && ctu_x > 0) add(y - 1, ctu_x - 1);
            if (y > 0 && ctu_x < cols - 1) add(y - 1, ctu_x + 1);
            if (ctu_x > 0) add(y, ctu_x - 1);
            double avg = -1.0;
            if (own_n + nb_n) avg = (0.6 * own_cost + 0.4 * nb_cost) / (0.6 * own_n + 0.4 * nb_n);
            if (best < avg && avg > 0) {
                replace = true;
                descend = false;
            }
        }
        if (descend) {
            double sub = 0;
            const int half = size / 2;
            for (int part = 0; part < 4; part++) {
                const uint32_t c = decide(px + (part & 1) * half, py + (part >> 1) * half, half, depth + 1);
                sub += (double)c;
                const size_t i = at(y, ctu_x, depth + 1);
                an.sum_cost[i] += (double)c;
                an.count[i] += 1;
                an.avg_cost[i] = an.sum_cost[i] / an.count[i];
            }
            if (depth == 0) {
                const size_t i = at(y, ctu_x, 0);
                an.sum_cost[i] += best;
                an.count[i] += 1;
                an.avg_cost[i] = an.sum_cost[i] / an.count[i];
            }
            if (sub >= best) {
                replace = true;
                for (int d = depth + 1; d < 5; d++) cab[d] = cab[depth];
            } else {
                best = sub;
                cab[depth] = cab[depth + 1];
            }
        }
    }

    if (replace) {
        block_in(rec_y + py * 64 + px, 64, best_rec[li], size, size, size);
        block_in(coef_y + py * 64 + px, 128, best_coef[li], size * 2, size, size * 2);
        int pi = 0;
        if (std::abs(cand[0].x - cand[1].x) + std::abs(cand[0].y - cand[1].y) > 0) {
            const unsigned d0 = std::abs(cand[0].x - bmv.x) + std::abs(cand[0].y - bmv.y);
            const unsigned d1 = std::abs(cand[1].x - bmv.x) + std::abs(cand[1].y - bmv.y) + 1;
            pi = d0 <= d1 ? 0 : 1;
        }
        for (int i = 0; i < b8; i++)
            for (int j = 0; j < b8; j++) {
                V(r8 + i, c8 + j) = bmv;
                mvp_idx[(r8 + i) * 8 + c8 + j] = (uint8_t)pi;
                mvd[(r8 + i) * 8 + c8 + j] = { (int16_t)(bmv.x - cand[pi].x), (int16_t)(bmv.y - cand[pi].y) };
            }
        const int above = y * 64 + py > 0, left = ctu_x * 64 + px > 0;
        for (int i = 0; i < b4; i++)
            for (int j = 0; j < b4; j++) {
                M(r4 + i, c4 + j) = kValid;
                I0(r4 + i, c4 + j) = (uint16_t)(kFlagInter | merge_idx);
                const int cbf = size == 64 ? (best_cbf[li] >> ((i >> 3) * 2 + (j >> 3))) & 1 : best_cbf[li] & 1;
                I1(r4 + i, c4 + j) = (uint8_t)(((log2 - 2) << kOffSize) | ((size == 64) << kOff64) | (above << kValidT) | (left << kValidL) |
                                               (cbf << kOffCbfY) | skip);
            }
    }
    return (uint32_t)best;
}

// the decided CUs of the CTU in coding order: a CU is recognised by the size its first 4x4 entry records
void Row::collect_cus(int px, int py, int size, std::vector<int> &list)
{
    const uint8_t f = I1(py / 4, px / 4);
    const int is64 = (f >> kOff64) & 1 & (I0(0, 0) >> 7);
    const int cu = is64 ? 64 : 4 << (f >> kOffSize);
    if (cu >= size) {
        list.push_back(px | (py << 8) | (size << 16));
        return;
    }
    const int half = size / 2;
    for (int part = 0; part < 4; part++) collect_cus(px + (part & 1) * half, py + (part >> 1) * half, half, list);
}

// xEncInterCTU (src/encode.cpp:3641-3791) in two halves. Nothing the luma decisions of the next CTU ask the evaluator for depends
// on this CTU's chroma (what depends on it - the skip flags of the left neighbours - is read when the answers are back), so
// the chroma requests of a CTU travel with the first luma request of the CTU to its right instead of in a round trip of
// their own; the CTU counts as done, for the row below, once they are applied.
void Row::chroma_prepare()
{
    chroma_list.clear();
    collect_cus(0, 0, 64, chroma_list);
    chroma_req.resize(chroma_list.size());
    for (size_t i = 0; i < chroma_list.size(); i++) {
        const int px = chroma_list[i] & 255, py = (chroma_list[i] >> 8) & 255, size = chroma_list[i] >> 16;
        ChromaEval &c = chroma_req[i];
        c.x0 = ctu_x * 64 + px, c.y0 = y * 64 + py, c.size = size;
        c.mv = V(py / 8, px / 8);
        c.skip = I1(py / 4, px / 4) & 1 & (I0(0, 0) >> 7);
        c.qpc = chroma_qp(qp);
    }
}

// the answers of chroma_prepare()'s requests go into the picture-level arrays of CTU xp, whose luma part is there already
void Row::chroma_apply(int xp)
{
    PictureResult &out = *sh->out;
    const int w4 = sh->w / 4, wc = sh->w / 2;
    auto G0 = [&](int r, int c) -> uint16_t & { return out.info0[(size_t)(y * 16 + r) * w4 + xp * 16 + c]; };
    auto G1 = [&](int r, int c) -> uint8_t & { return out.info1[(size_t)(y * 16 + r) * w4 + xp * 16 + c]; };
    uint8_t *const gu = &out.rec_u[(size_t)(y * 32) * wc + xp * 32], *const gv = &out.rec_v[(size_t)(y * 32) * wc + xp * 32];
    int16_t *const cu = &out.coef_u[(size_t)(y * 32) * wc + xp * 32], *const cv = &out.coef_v[(size_t)(y * 32) * wc + xp * 32];
    for (size_t n = 0; n < chroma_list.size(); n++) {
        const int px = chroma_list[n] & 255, py = (chroma_list[n] >> 8) & 255, size = chroma_list[n] >> 16, r4 = py / 4, c4 = px / 4, b4 = size / 4, sc = size / 2;
        const ChromaEval &c = chroma_req[n];
        block_in(gu + (size_t)(py / 2) * wc + px / 2, wc, c.rec_u, sc, sc, sc);
        block_in(gv + (size_t)(py / 2) * wc + px / 2, wc, c.rec_v, sc, sc, sc);
        if (c.skip) continue;                      // the levels stay zero
        take_levels(cu + (size_t)(py / 2) * wc + px / 2, wc, c.level_u, sc, c.sum_u, size == 64 ? 4 : 1);
        take_levels(cv + (size_t)(py / 2) * wc + px / 2, wc, c.level_v, sc, c.sum_v, size == 64 ? 4 : 1);
        const int merge_idx = G0(r4, c4) & 0x3f;
        if (size == 64) {
            int any = 0;
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++) {
                    const int cbf_c = (c.sum_v[j * 2 + i] ? 2 : 0) | (c.sum_u[j * 2 + i] ? 1 : 0);
                    const int cbf_y = (G1(r4 + j * 8, c4 + i * 8) >> kOffCbfY) & 1;
                    any |= cbf_y | cbf_c;
                    const uint16_t first = G0(r4 + j * 8, c4 + i * 8);
                    for (int ii = 0; ii < 8; ii++)
                        for (int jj = 0; jj < 8; jj++) G0(r4 + j * 8 + ii, c4 + i * 8 + jj) = (uint16_t)(first | (cbf_c << kOffCbfC));
                }
            if (merge_idx && any == 0)
                for (int ii = 0; ii < 16; ii++)
                    for (int jj = 0; jj < 16; jj++) G1(r4 + ii, c4 + jj) |= kMaskSkip;
        } else {
            const int cbf_c = (c.sum_v[0] ? 2 : 0) | (c.sum_u[0] ? 1 : 0), cbf_y = (G1(r4, c4) >> kOffCbfY) & 1;
            const bool skipped = merge_idx && (cbf_y | cbf_c) == 0;
            const uint16_t first0 = G0(r4, c4);
            const uint8_t first1 = G1(r4, c4);
            for (int ii = 0; ii < b4; ii++)
                for (int jj = 0; jj < b4; jj++) {
                    G0(r4 + ii, c4 + jj) = (uint16_t)(first0 | kFlagInter | (cbf_c << kOffCbfC));
                    G1(r4 + ii, c4 + jj) = (uint8_t)(first1 | (skipped ? kMaskSkip : 0));
                }
        }
    }
}

// CTU xp of this row is complete: the row below may go on
void Row::ctu_done(int xp)
{
    Shared &S = *sh;
    std::lock_guard<std::mutex> lk(S.m);
    S.done[y] = xp + 1;
    S.done_seen[y].store(xp + 1, std::memory_order_release);
    if (S.done_luma[y] < xp + 1) S.done_luma[y] = xp + 1;
    if (y + 1 < S.nrows) {
        Row &below = S.rows[y + 1];
        const int wanted = below.need.load(std::memory_order_relaxed);
        if (wanted >= 0 && (below.need_whole ? S.done[y] : S.done_luma[y]) >= wanted) {
            S.lanes[0].running++;      // on the waiter's behalf, before it can ask for anything
            below.need.store(-1, std::memory_order_release);
            if (S.sleepers.load(std::memory_order_relaxed) > 0) S.cv_rows.notify_all();
        }
    }
}

// I pictures: the luma of CTU xp is decided and in the picture's arrays
void Row::luma_done(int xp)
{
    Shared &S = *sh;
    std::lock_guard<std::mutex> lk(S.m);
    S.done_luma[y] = xp + 1;
    if (y + 1 < S.nrows) {
        Row &below = S.rows[y + 1];
        const int wanted = below.need.load(std::memory_order_relaxed);
        if (wanted >= 0 && !below.need_whole && S.done_luma[y] >= wanted) {
            S.lanes[0].running++;
            below.need.store(-1, std::memory_order_release);
            if (S.sleepers.load(std::memory_order_relaxed) > 0) S.cv_rows.notify_all();
        }
    }
}

// until the row above has finished `ctus` CTUs (whole: with their chroma)
void Row::wait_row_above(int ctus, bool whole)
{
    Shared &S = *sh;
    std::unique_lock<std::mutex> lk(S.m);
    if ((whole ? S.done[y - 1] : S.done_luma[y - 1]) >= ctus) return;
    need_whole = whole;
    need.store(ctus, std::memory_order_relaxed);
    release_answers();
    Lane &ln = S.lanes[0];
    if (ln.running.fetch_sub(1) == 1) ln.cv_main.notify_all();
    lk.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    block(kRowAbove);
    t_resume += std::chrono::steady_clock::now() - t0;      // waiting for the row above is not work
    if (S.error) throw Abort();
}

void Row::make_fiber()
{
    constexpr size_t kStack = 512 * 1024;
    stack.reset(new char[kStack]);
    getcontext(&fiber);
    fiber.uc_stack.ss_sp = stack.get();
    fiber.uc_stack.ss_size = kStack;
    fiber.uc_link = nullptr;
    const uintptr_t a = reinterpret_cast<uintptr_t>(this);
    makecontext(&fiber, reinterpret_cast<void (*)()>(&Row::fiber_main), 2, (unsigned)(a & 0xffffffffu), (unsigned)(a >> 32));
    XMI_TSAN_FIBER(san_fiber = __tsan_create_fiber(0));
    started = true;
}

void Row::fiber_main(unsigned lo, unsigned hi)
{
    Row *r = reinterpret_cast<Row *>((uintptr_t)lo | ((uintptr_t)hi << 32));
    r->run();
    r->done = true;
    XMI_TSAN_FIBER(__tsan_switch_to_fiber(r->san_home, 0));
    swapcontext(&r->fiber, r->home);      // for good
}

// cores this process may use: the hardware's, the ones its affinity mask names (a rank of a multi-GPU job that was given a
// slice of the host), or the CPU time its control group grants - whichever is least
int host_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int allowed = CPU_COUNT(&set);
        if (allowed >= 1 && allowed < n) n = allowed;
    }
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    long long quota = 0, period = 0;
    std::string q;
    if (f >> q >> period && q != "max" && period > 0) {
        quota = std::atoll(q.c_str());
        const int granted = (int)((quota + period - 1) / period);
        if (granted >= 1 && granted < n) n = granted;
    }
    return n;
}

// A round trip takes tens of microseconds, and a thread that sleeps on a condition variable needs about ten more to come
// back (longer when its core has gone idle): waiters poll for spin_us first and block only when nothing came.
static inline bool poll_until(int spin_us, const std::function<bool()> &ready)
{
    if (spin_us <= 0) return ready();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; i++) {
            if (ready()) return true;
            __builtin_ia32_pause();
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) return ready();
    }
}

void Row::release_answers()
{
    if (served_by >= 0) sh->lanes[served_by].readers.fetch_sub(1, std::memory_order_acq_rel);
    served_by = -1;
}

void Row::post_and_wait()
{
    busy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_resume).count();
    requests++;
    release_answers();
    Lane &ln = sh->lanes[0];
    // `pending` goes up last: a coordinator that sees it finds the two counters already moved
    ln.posted.fetch_add(1, std::memory_order_acq_rel);
    const bool last = ln.running.fetch_sub(1, std::memory_order_acq_rel) == 1;
    pending.store(true, std::memory_order_release);
    if (last) {      // the last busy row: coordinators may be asleep
        { std::lock_guard<std::mutex> lk(sh->m); }
        ln.cv_main.notify_all();
    }
    block(kAnswers);
    t_resume = std::chrono::steady_clock::now();
    if (sh->error) throw Abort();
}

void Row::evaluate_cus(CuEval *const *list, int n)
{
    for (int i = 0; i < n; i++) req_cu[i] = list[i];
    n_req_cu = n;
    req_chroma = chroma_waiting >= 0;
    post_and_wait();
    n_req_cu = 0;
    if (req_chroma) {
        req_chroma = false;
        const int xp = chroma_waiting;
        chroma_waiting = -1;
#ifdef XMI_EXP_HOST_TIMING
        const uint64_t tc0 = tick();
#endif
        chroma_apply(xp);
        ctu_done(xp);
#ifdef XMI_EXP_HOST_TIMING
        cat[3] += tick() - tc0;
#endif
        // the left neighbours of the CTU being decided are CTU xp's last column
        const PictureResult &out = *sh->out;
        const int w4 = sh->w / 4;
        for (int r = 0; r < 16; r++) I1(r, -1) = out.info1[(size_t)(y * 16 + r) * w4 + ctu_x * 16 - 1];
    }
}

void Row::evaluate_chroma()
{
    n_req_cu = 0;
    req_chroma = true;
    post_and_wait();
    req_chroma = false;
}

void Row::evaluate_intra(IntraEval *const *list, int n, bool with_quad, bool with_pair)
{
    for (int i = 0; i < n; i++) req_in[i] = list[i];
    n_req_in = n, n_req_cu = 0;
    req_chroma = req_ichroma = req_cquad = false;
    req_quad = with_quad, req_pair = with_pair;
    const bool link = ic_ctu >= 0 && chroma_link_fill();
    post_and_wait();
    n_req_in = 0;
    req_quad = req_pair = false;
    if (link) chroma_link_take();
}

// request for the four 4x4 PUs of the 8x8 block at (px, py) of the CTU, nothing of which is decided yet
void Row::fill_quad(IntraQuadEval &q, int px, int py)
{
    const int r4 = py / 4, c4 = px / 4;
    q.x0 = ctu_x * 64 + px, q.y0 = y * 64 + py;
    for (int i = 0; i < 16; i++) q.ref[i] = (uint8_t)sample_y(py + 15 - i, px - 1);
    for (int i = 0; i <= 16; i++) q.ref[16 + i] = (uint8_t)sample_y(py - 1, px - 1 + i);
    auto ok = [&](int r, int c) { return M(r, c) != kInvalid; };
    // PU k at (c4 + (k & 1), r4 + (k >> 1)): neighbours inside the block are the PUs before it
    q.valids[0] = (uint8_t)((ok(r4 + 1, c4 - 1) ? 1 : 0) | (ok(r4, c4 - 1) ? 2 : 0) | (ok(r4 - 1, c4 - 1) ? 4 : 0) | (ok(r4 - 1, c4) ? 8 : 0) | (ok(r4 - 1, c4 + 1) ? 16 : 0));
    q.valids[1] = (uint8_t)(0 | 2 | (ok(r4 - 1, c4) ? 4 : 0) | (ok(r4 - 1, c4 + 1) ? 8 : 0) | (ok(r4 - 1, c4 + 2) ? 16 : 0));
    q.valids[2] = (uint8_t)((ok(r4 + 2, c4 - 1) ? 1 : 0) | (ok(r4 + 1, c4 - 1) ? 2 : 0) | (ok(r4, c4 - 1) ? 4 : 0) | 8 | 16);
    q.valids[3] = (uint8_t)((ok(r4 + 2, c4) ? 1 : 0) | 2 | 4 | 8 | (ok(r4, c4 + 2) ? 16 : 0));
    q.left_mode[0] = ok(r4, c4 - 1) ? M(r4, c4 - 1) : 1, q.left_mode[1] = ok(r4 + 1, c4 - 1) ? M(r4 + 1, c4 - 1) : 1;
    q.top_mode[0] = (ok(r4 - 1, c4) && py) ? M(r4 - 1, c4) : 1, q.top_mode[1] = (ok(r4 - 1, c4 + 1) && py) ? M(r4 - 1, c4 + 1) : 1;
    q.strong = (uint8_t)strong;
    q.sqrt_lambda = std::sqrt(lambda);
    q.qp = qp;
    q.after_kind = 0, q.after_pu = nullptr, q.after_quad = nullptr;
}

// Requests for the 8x8 block to the right of the block at (px, py), which this call decides: both premises. The block's units
// count as available; what the premises differ in - the left column and the modes to the left - comes from the evaluator.
void Row::fill_pair(int px, int py)
{
    const int r4 = py / 4, c4 = px / 4;
    uint8_t keep[4];
    for (int k = 0; k < 4; k++) keep[k] = M(r4 + (k >> 1), c4 + (k & 1)), M(r4 + (k >> 1), c4 + (k & 1)) = 1;
    for (int h = 0; h < 2; h++) {
        fill_intra(pair_in[h], px + 8, py, 8);
        fill_quad(pair_quad[h], px + 8, py);
        pair_in[h].after_kind = pair_quad[h].after_kind = (uint8_t)(h + 1);
        pair_in[h].after_pu = pair_quad[h].after_pu = &in_spec[1];
        pair_in[h].after_quad = pair_quad[h].after_quad = &quad;
    }
    for (int k = 0; k < 4; k++) M(r4 + (k >> 1), c4 + (k & 1)) = keep[k];
    pair_ok = true;
    pair_x = ctu_x * 64 + px + 8, pair_y = y * 64 + py;
}

// The block at (px, py) was asked for with the block to its left: the answers of the premise that came true, if the request
// they answer is the one the block makes now. q: the block's own request, filled in.
bool Row::take_pair(IntraEval &q, int px, int py)
{
    pair_ok = false;
    const int r4 = py / 4, c4 = px / 4;
    const int h = (I1(r4, c4 - 1) >> kOffSize) == 0 ? 1 : 0;      // the block to the left was kept as four 4x4 PUs, or as one PU
    const IntraEval &a = pair_in[h];
    const IntraQuadEval &b = pair_quad[h];
    if (a.best_mode < 0 || b.best_mode[0] < 0) return false;      // the evaluator could not wait for the block to the left
    fill_intra(q, px, py, 8);
    IntraQuadEval now;
    fill_quad(now, px, py);
    const bool same = q.valids == a.valids && std::memcmp(q.ref, a.ref, 33) == 0 && std::memcmp(q.mpm, a.mpm, 3) == 0 && std::memcmp(now.ref, b.ref, 33) == 0 &&
                      std::memcmp(now.valids, b.valids, 4) == 0 && std::memcmp(now.left_mode, b.left_mode, 2) == 0 && std::memcmp(now.top_mode, b.top_mode, 2) == 0;
    if (!same) {      // (cannot happen: the premise is what was decided)
        sh->error = -3;
        throw Abort();
    }
    q.best_mode = a.best_mode, q.ssd = a.ssd, q.sum = a.sum, q.rec = a.rec, q.level = a.level;
    pairs_taken++;
    quad = b;
    quad.after_kind = 0;
    quad_ok = true;
    return true;
}

inline int interleave4(int v) { return (v & 1) | ((v >> 1) & 2) | ((v >> 2) & 4) | ((v >> 3) & 8); }      // getInterleaveBits (src/encode.cpp:4220)

// request of the PU at (px, py) of the CTU: availability, most probable modes, neighbouring samples (src/encode.cpp:3964-4015)
void Row::fill_intra(IntraEval &q, int px, int py, int size)
{
    const int r4 = py / 4, c4 = px / 4, b4 = size / 4;
    q.x0 = ctu_x * 64 + px, q.y0 = y * 64 + py, q.size = size;
    const bool lb = M(r4 + b4, c4 - 1) != kInvalid, l = M(r4, c4 - 1) != kInvalid, lt = M(r4 - 1, c4 - 1) != kInvalid;
    const bool t = M(r4 - 1, c4) != kInvalid, tr = M(r4 - 1, c4 + b4) != kInvalid;
    q.valids = (uint8_t)((lb ? 1 : 0) | (l ? 2 : 0) | (lt ? 4 : 0) | (t ? 8 : 0) | (tr ? 16 : 0));
    q.strong = (uint8_t)strong;
    const int left = l ? M(r4, c4 - 1) : 1, top = (t && py) ? M(r4 - 1, c4) : 1;      // DC where there is no neighbour, and above the CTU
    q.left_mode = (uint8_t)left, q.top_mode = (uint8_t)top;
    most_probable_modes(left, top, q.mpm);
    q.after_kind = 0, q.after_pu = nullptr, q.after_quad = nullptr;
    for (int i = 0; i < 2 * size; i++) q.ref[i] = (uint8_t)sample_y(py + 2 * size - 1 - i, px - 1);
    for (int i = 0; i <= 2 * size; i++) q.ref[2 * size + i] = (uint8_t)sample_y(py - 1, px - 1 + i);
    q.sqrt_lambda = std::sqrt(lambda);
    q.qp = qp;
}

// xEncDecideCU_Intra (src/encode.cpp:3906-4215)
double Row::decide_intra(int px, int py, int size, int depth)
{
    const int log2 = ilog2(size), li = log2 - 2, b4 = size / 4, r4 = py / 4, c4 = px / 4;
    Ctx &work = cab[0];
    work = cab[depth];
    double best = kDoubleMax;
    int best_mode = 0, pred_idx = 0, valids = 0;
    if (size < 64) {
        IntraEval &q = in_spec[li];
        const int gx = ctu_x * 64 + px, gy = y * 64 + py;
        if (size == 4 && quad_ok && (gx & ~7) == quad.x0 && (gy & ~7) == quad.y0) {
            // one of the four PUs that were evaluated with their 8x8 block: the request as this PU sees it now must be
            // the one the evaluator derived, then its results are taken over
            const int k = ((gy >> 2) & 1) * 2 + ((gx >> 2) & 1);
            fill_intra(q, px, py, 4);
            if (q.valids != quad.valids[k]) {      // (cannot happen: both follow the coding order)
                sh->error = -3;
                throw Abort();
            }
            q.best_mode = quad.best_mode[k], q.ssd = quad.ssd[k], q.sum = quad.sum[k];
            block_in(quad_rec, 4, quad.rec + (k >> 1) * 32 + (k & 1) * 4, 8, 4, 4);
            block_in(quad_level, 8, quad.level + (k >> 1) * 32 + (k & 1) * 4, 16, 4, 8);
            q.rec = quad_rec, q.level = quad_level;
            if (k == 3) quad_ok = false;
        } else if (size == 8 && pair_ok && pair_x == gx && pair_y == gy && take_pair(q, px, py)) {
            // (asked for with the block to the left)
        } else if (!(in_ok[li] && q.x0 == gx && q.y0 == gy && q.size == size)) {
            IntraEval *list[4];
            int n = 0;
            bool with_quad = false, with_pair = false;
            fill_intra(q, px, py, size);
            list[n++] = &q;
            for (int sz = size / 2, l = li - 1; sh->speculate && sz >= 4; sz /= 2, l--) {
                if (sz == 4) {      // below the 8x8 PU: all four 4x4 PUs
                    fill_quad(quad, px, py);
                    with_quad = quad_ok = true;
                    break;
                }
                fill_intra(in_spec[l], px, py, sz);
                in_ok[l] = true;
                list[n++] = &in_spec[l];
            }
            pair_ok = false;
            if (with_quad && sh->pairs && (px & 8) == 0) fill_pair(px, py), with_pair = true;
            evaluate_intra(list, n, with_quad, with_pair);
        }
        in_ok[li] = false;
        best_mode = q.best_mode, valids = q.valids;
        // the coded index of the mode: its rank among the most probable modes, else its rank among the other 32 (:4048-4081)
        int m[3] = { q.mpm[0], q.mpm[1], q.mpm[2] };
        if (best_mode == m[0]) pred_idx = 0;
        else if (best_mode == m[1]) pred_idx = 1;
        else if (best_mode == m[2]) pred_idx = 2;
        else {
            if (m[0] > m[1]) std::swap(m[0], m[1]);
            if (m[0] > m[2]) std::swap(m[0], m[2]);
            if (m[1] > m[2]) std::swap(m[1], m[2]);
            pred_idx = best_mode;
            pred_idx -= pred_idx > m[2];
            pred_idx -= pred_idx > m[1];
            pred_idx -= pred_idx > m[0];
            pred_idx += 3;
        }
        int cbf = 0;
        const uint32_t sum[4] = { q.sum, 0, 0, 0 };
        const uint32_t coeff_bits = rate_cu_luma(rate, q.level, size, sum, &cbf);
        rate.begin();
        if (size <= 8) rate.bin(size == 8, kCtxPartSize);
        const unsigned pi[4] = { (unsigned)pred_idx, 0, 0, 0 };
        code_intra_dir_luma(rate, 1, pi);
        const uint32_t mode_bits = rate.bits();
        best = (int)q.ssd + lambda * (double)(coeff_bits + mode_bits);
        cab[depth] = work;
        std::memcpy(best_rec[li], q.rec, (size_t)size * size);
        std::memcpy(best_coef[li], q.level, (size_t)size * size * 2);
        best_cbf[li] = cbf;
    }
    bool replace = size == 4;
    double sub = kDoubleMax;
    if (size > 4 && best >= 0) {
        const int half = size / 2;
        double c[4];
        for (int part = 0; part < 4; part++) c[part] = decide_intra(px + (part & 1) * half, py + (part >> 1) * half, half, depth + 1);
        sub = c[0] + c[1] + c[2] + c[3] + 4;
        if (sub >= best) {
            replace = true;
            for (int d = depth + 1; d < 5; d++) cab[d] = cab[depth];
        } else {
            cab[depth] = cab[depth + 1];
        }
    }
    if (replace) {
        block_in(rec_y + py * 64 + px, 64, best_rec[li], size, size, size);
        block_in(coef_y + py * 64 + px, 128, best_coef[li], size * 2, size, size * 2);
        I0(r4, c4) = (uint16_t)pred_idx;
        const uint8_t f = (uint8_t)((valids & 0x3f) | (best_cbf[li] ? 1 << kOffCbfY : 0) | ((log2 - 2) << kOffSize));
        for (int i = 0; i < b4; i++)
            for (int j = 0; j < b4; j++) {
                M(r4 + i, c4 + j) = (uint8_t)best_mode;
                I1(r4 + i, c4 + j) = f;
            }
    }
    return best < sub ? best : sub;
}

// xEncodeChromaCU (src/encode.cpp:4234-4418): the PUs of the decided CTU in coding order. Luma PUs of 16x16 and more have one
// chroma PU each; the chroma of the 8x8 luma blocks of a 16x16 block - four 4x4 chroma PUs - goes out as one request.
// The chain of CTU xp starts when the luma of the CTU is in the picture's arrays.
void Row::chroma_chain_arm(int xp)
{
    ic_ctu = xp, ic_idx = 0, ic_begun = false;
    for (int r = 0; r < 16; r++) {
        std::memcpy(&ic_info1[r * 16], &I1(r, 0), 16);
        std::memcpy(&ic_mode[r * 16], &M(r, 0), 16);
    }
}

bool Row::chroma_link_fill()
{
    Shared &S = *sh;
    const PictureResult &out = *S.out;
    const int xp = ic_ctu;
    if (!ic_begun) {
        // the unfiltered chroma above (as far as the CTU above and to the right) and to the left of the CTU
        const int wanted = xp + 2 < S.cols ? xp + 2 : S.cols;
        if (y > 0 && S.done_seen[y - 1].load(std::memory_order_acquire) < wanted) return false;
        const int wc = S.w / 2;
        std::memset(top_u, 0x80, sizeof top_u), std::memset(top_v, 0x80, sizeof top_v);
        std::memset(left_u, 0x80, sizeof left_u), std::memset(left_v, 0x80, sizeof left_v);
        if (y > 0) {
            for (int i = 0; i <= 64; i++) {
                const int gx = xp * 32 - 1 + i;
                if (gx >= 0 && gx < wc) top_u[i] = out.rec_u[(size_t)(y * 32 - 1) * wc + gx], top_v[i] = out.rec_v[(size_t)(y * 32 - 1) * wc + gx];
            }
        }
        if (xp > 0) {
            for (int r = 0; r < 32; r++) {
                left_u[r] = out.rec_u[(size_t)(y * 32 + r) * wc + xp * 32 - 1];
                left_v[r] = out.rec_v[(size_t)(y * 32 + r) * wc + xp * 32 - 1];
            }
        }
        ic_begun = true;
    }
    auto info = [&](int dy, int dx) { return ic_info1[dy * 16 + dx]; };
    auto unit_valids = [&](int dy, int dx) {
        int valids = info(dy, dx) & 0x1f;
        if ((info(dy, dx) >> kOffSize) == 0) {      // the chroma block of four 4x4 luma PUs: below-left from the PU below, top-right from the PU to the right
            valids &= ~0x11;
            valids |= info(dy + 1, dx) & 0x01;
            valids |= info(dy, dx + 1) & 0x10;
        }
        return valids;
    };
    auto gather = [&](uint8_t *ru, uint8_t *rv, int cx, int cy, int size) {
        for (int i = 0; i < 2 * size; i++) {
            ru[i] = (uint8_t)sample_c(top_u, left_u, rec_u, cy + 2 * size - 1 - i, cx - 1);
            rv[i] = (uint8_t)sample_c(top_v, left_v, rec_v, cy + 2 * size - 1 - i, cx - 1);
        }
        for (int i = 0; i <= 2 * size; i++) {
            ru[2 * size + i] = (uint8_t)sample_c(top_u, left_u, rec_u, cy - 1, cx - 1 + i);
            rv[2 * size + i] = (uint8_t)sample_c(top_v, left_v, rec_v, cy - 1, cx - 1 + i);
        }
    };
    const int dx = ic_dx = interleave4(ic_idx), dy = ic_dy = interleave4(ic_idx >> 1);
    const int size_y = ic_size_y = 1 << (2 + (info(dy, dx) >> kOffSize));
    const int cx = dx * 2, cy = dy * 2;
    if (size_y >= 16 || !S.speculate) {
        const int size = size_y == 4 ? 4 : size_y >> 1;
        IntraChromaEval &c = ichroma;
        c.x0 = xp * 32 + cx, c.y0 = y * 32 + cy, c.size = size;
        c.valids = (uint8_t)unit_valids(dy, dx), c.luma_mode = ic_mode[dy * 16 + dx];
        c.qpc = chroma_qp(qp);
        gather(c.ref_u, c.ref_v, cx, cy, size);
        req_ichroma = true;
        return true;
    }
    // a 16x16 luma block of smaller PUs (the unit index is a multiple of 16): the 8x8 area of its chroma
    IntraChromaQuadEval &c = cquad;
    c.x0 = xp * 32 + cx, c.y0 = y * 32 + cy;
    c.qpc = chroma_qp(qp);
    for (int k = 0; k < 4; k++) {
        const int uy = dy + (k >> 1) * 2, ux = dx + (k & 1) * 2;
        c.valids[k] = (uint8_t)unit_valids(uy, ux), c.luma_mode[k] = ic_mode[uy * 16 + ux];
    }
    gather(c.ref_u, c.ref_v, cx, cy, 8);
    req_cquad = true;
    return true;
}

void Row::chroma_link_take()
{
    Shared &S = *sh;
    PictureResult &out = *S.out;
    const int xp = ic_ctu, w4 = S.w / 4, dx = ic_dx, dy = ic_dy, cx = dx * 2, cy = dy * 2, size_y = ic_size_y;
    auto flags = [&](int r, int c) -> uint16_t & { return out.info0[(size_t)(y * 16 + r) * w4 + xp * 16 + c]; };
    if (req_ichroma) {
        req_ichroma = false;
        const IntraChromaEval &c = ichroma;
        const int size = c.size;
        block_in(rec_u + cy * 32 + cx, 32, c.rec_u, size, size, size);
        block_in(rec_v + cy * 32 + cx, 32, c.rec_v, size, size, size);
        block_in(coef_u + cy * 32 + cx, 64, c.level_u, size * 2, size, size * 2);
        block_in(coef_v + cy * 32 + cx, 64, c.level_v, size * 2, size, size * 2);
        const int cbf_c = (c.sum_v ? 2 : 0) | (c.sum_u ? 1 : 0);
        flags(dy, dx) |= (uint16_t)((c.best_idx << 13) | (cbf_c << kOffCbfC));
        ic_idx += size_y < 8 ? 4 : size_y * size_y / 16;
    } else {
        req_cquad = false;
        const IntraChromaQuadEval &c = cquad;
        block_in(rec_u + cy * 32 + cx, 32, c.rec_u, 8, 8, 8);
        block_in(rec_v + cy * 32 + cx, 32, c.rec_v, 8, 8, 8);
        block_in(coef_u + cy * 32 + cx, 64, c.level_u, 16, 8, 16);
        block_in(coef_v + cy * 32 + cx, 64, c.level_v, 16, 8, 16);
        for (int k = 0; k < 4; k++) {
            const int cbf_c = (c.sum_v[k] ? 2 : 0) | (c.sum_u[k] ? 1 : 0);
            flags(dy + (k >> 1) * 2, dx + (k & 1) * 2) |= (uint16_t)((c.best_idx[k] << 13) | (cbf_c << kOffCbfC));
        }
        ic_idx += 16;
    }
    if (ic_idx == 256) {      // the CTU is complete
        const int wc = S.w / 2;
        block_in(&out.rec_u[(size_t)(y * 32) * wc + xp * 32], wc, rec_u, 32, 32, 32);
        block_in(&out.rec_v[(size_t)(y * 32) * wc + xp * 32], wc, rec_v, 32, 32, 32);
        block_in(&out.coef_u[(size_t)(y * 32) * wc + xp * 32], S.w, coef_u, 64, 32, 64);
        block_in(&out.coef_v[(size_t)(y * 32) * wc + xp * 32], S.w, coef_v, 64, 32, 64);
        ic_ctu = -1;
        ctu_done(xp);
    }
}

// what is left of the chain, one request after the other
void Row::chroma_chain_drain()
{
    while (ic_ctu >= 0) {
        if (!ic_begun && y > 0) wait_row_above(ic_ctu + 2 < sh->cols ? ic_ctu + 2 : sh->cols, true);
        n_req_in = n_req_cu = 0;
        req_chroma = req_quad = req_ichroma = req_cquad = false;
        if (!chroma_link_fill()) {      // (cannot happen after the wait)
            sh->error = -3;
            throw Abort();
        }
        post_and_wait();
        chroma_link_take();
    }
}

// the row loop of xEncodeCUs (src/encode.cpp:5233-5590)
void Row::run()
{
    Shared &S = *sh;
    PictureResult &out = *S.out;
    const int cols = S.cols, w4 = S.w / 4, w8 = S.w / 8;
    qp = S.qp;
    lambda = (S.intra ? 0.57 : 0.578) * std::pow(2.0, (qp - 12) / 3.0);      // src/encode.cpp:5348, 5406
    strong = S.strong;
    rate.frac = S.an->frac_bits[y];
    rate.transitions = y == 0;
    rate.w = &cab[0];
    Ctx last = S.start;
    auto clip64 = [](int v) { return v < -64 ? -64 : v > 64 ? 64 : v; };
    try {
        for (int x = 0; x < cols; x++) {
            ctu_x = x;
            std::memset(tree_ok, 0, sizeof tree_ok);
            in_ok[0] = in_ok[1] = in_ok[2] = in_ok[3] = false;
            quad_ok = pair_ok = false;
            if (y > 0) wait_row_above(x + 2 < cols ? x + 2 : cols, !S.intra);      // top-right neighbour done (or the whole row above)
#ifdef XMI_EXP_HOST_TIMING
            const uint64_t tp0 = tick();
#endif
            // caches: neighbours from the picture-level arrays of the CTUs already done
            std::memset(mode, kInvalid, sizeof mode);
            std::memset(info1, 0, sizeof info1);
            std::memset(info0, 0, sizeof info0);
            std::memset(mv, 0, sizeof mv);
            if (y > 0) {
                for (int c = 0; c < 16; c++) M(-1, c) = kValid;
                if (x > 0) M(-1, -1) = kValid;
                if (x < cols - 1) M(-1, 16) = kValid;
                for (int c = 0; c < 16; c++) I1(-1, c) = out.info1[(size_t)(y * 16 - 1) * w4 + x * 16 + c];
                for (int c = -1; c <= 8; c++) {
                    const int gx = x * 8 + c;
                    if (gx >= 0 && gx < w8) V(-1, c) = out.mv[(size_t)(y * 8 - 1) * w8 + gx];
                }
            }
            if (x > 0) {
                for (int r = 0; r < 16; r++) {
                    M(r, -1) = out.mode[(size_t)(y * 16 + r) * w4 + x * 16 - 1];      // MODE_VALID in P pictures
                    I1(r, -1) = out.info1[(size_t)(y * 16 + r) * w4 + x * 16 - 1];
                }
                for (int r = 0; r < 8; r++) V(r, -1) = out.mv[(size_t)(y * 8 + r) * w8 + x * 8 - 1];
            }
            const uint64_t requests_before = requests;
            const Ctx &init = (y == 0 || x > 0) ? last : S.saved[y - 1];
            for (int d = 0; d < 5; d++) cab[d] = init;
            std::memcpy(&out.ctx_in[((size_t)y * cols + x) * 256], cab[0].m, 256);
            // xCuMvClip (:3794-3812)
            min_ipel.x = (int16_t)clip64(-64 * x - 64), max_ipel.x = (int16_t)clip64(64 * (cols - x));
            min_ipel.y = (int16_t)clip64(-64 * y - 64), max_ipel.y = (int16_t)clip64(64 * (S.nrows - y));

#ifdef XMI_EXP_HOST_TIMING
            cat[0] += tick() - tp0;
#endif
            if (S.intra) {
                // the unfiltered reconstruction above and to the left of the CTU, as far as the picture has it
                std::memset(top_y, 0x80, sizeof top_y);
                std::memset(left_y, 0x80, sizeof left_y);
                if (y > 0) {
                    for (int i = 0; i <= 128; i++) {
                        const int gx = x * 64 - 1 + i;
                        if (gx >= 0 && gx < S.w) top_y[i] = out.rec_y[(size_t)(y * 64 - 1) * S.w + gx];
                    }
                }
                if (x > 0)
                    for (int r = 0; r < 64; r++) left_y[r] = out.rec_y[(size_t)(y * 64 + r) * S.w + x * 64 - 1];
                std::memset(mvd, 0, sizeof mvd), std::memset(mvp_idx, 0, sizeof mvp_idx);
                out.rd_cost[(size_t)y * cols + x] = (uint32_t)decide_intra(0, 0, 64, 0);
            } else {
                out.rd_cost[(size_t)y * cols + x] = decide(0, 0, 64, 0);
                chroma_prepare();
            }

#ifdef XMI_EXP_HOST_TIMING
            const uint64_t te0 = tick();
#endif
            last = cab[0];
            if (S.ctu_requests) S.ctu_requests[(size_t)y * cols + x] = (uint16_t)(requests - requests_before);
            std::memcpy(&out.ctx_out[((size_t)y * cols + x) * 256], cab[0].m, 256);
            for (int r = 0; r < 16; r++) {
                std::memcpy(&out.info0[(size_t)(y * 16 + r) * w4 + x * 16], &I0(r, 0), 32);
                std::memcpy(&out.info1[(size_t)(y * 16 + r) * w4 + x * 16], &I1(r, 0), 16);
                std::memcpy(&out.mode[(size_t)(y * 16 + r) * w4 + x * 16], &M(r, 0), 16);
            }
            for (int r = 0; r < 8; r++) {
                std::memcpy(&out.mv[(size_t)(y * 8 + r) * w8 + x * 8], &V(r, 0), 8 * sizeof(Mv));
                std::memcpy(&out.mvd[(size_t)(y * 8 + r) * w8 + x * 8], &mvd[r * 8], 8 * sizeof(Mv));
                std::memcpy(&out.mvp_idx[(size_t)(y * 8 + r) * w8 + x * 8], &mvp_idx[r * 8], 8);
            }
            block_in(&out.rec_y[(size_t)(y * 64) * S.w + x * 64], S.w, rec_y, 64, 64, 64);
            block_in(&out.coef_y[(size_t)(y * 64) * S.w + x * 64], S.w * 2, coef_y, 128, 64, 128);
            if (x == 1 && y != S.nrows - 1) {
                std::lock_guard<std::mutex> lk(S.m);
                S.saved[y] = cab[0];
            }
            if (S.intra) {
                // the chroma chain of the CTU before this one has had the luma requests of this CTU to travel with
                luma_done(x);
                chroma_chain_drain();
                chroma_chain_arm(x);
                if (x + 1 == cols || !S.defer_chroma) chroma_chain_drain();
            } else if (x + 1 < cols && S.defer_chroma) {
                chroma_waiting = x;                 // with the next CTU's first request
#ifdef XMI_EXP_HOST_TIMING
                cat[2] += tick() - te0;
#endif
            } else {
                evaluate_chroma();
                chroma_apply(x);
                ctu_done(x);
            }
        }
    } catch (const Abort &) {
    }
    S.an->frac_bits[y] = rate.frac;
    std::lock_guard<std::mutex> lk(S.m);
    release_answers();
    Lane &ln = S.lanes[0];
    ln.finished.fetch_add(1);
    ln.running.fetch_sub(1);
    ln.cv_main.notify_all();
}

}  // namespace

static_assert(sizeof kInitP == 184 && sizeof kInitI == 184, "one initialisation value per context");

int chroma_qp(int qp)
{
    static const uint8_t t[58] = { 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                                   29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
    return qp < 0 ? 0 : qp > 57 ? 51 : t[qp];
}

const uint8_t *next_state_table() { return &tables().next[0][0]; }
const int32_t *entropy_bits_table() { return kEntropyBits; }

const uint16_t *diag_scan(int log2_size) { return log2_size >= 2 && log2_size <= 5 ? tables().scan[0][log2_size - 2] : nullptr; }

void init_contexts(int slice_type, int qp, uint8_t ctx[256])
{
    std::memset(ctx, 0, 256);
    const uint8_t *v = slice_type == 2 ? kInitI : kInitP;
    for (int i = 0; i < 184; i++) {
        const int m = (v[i] >> 4) * 5 - 45, n = ((v[i] & 15) << 3) - 16;
        int s = ((m * qp) >> 4) + n;
        s = s < 1 ? 1 : s > 126 ? 126 : s;
        const int mps = s >= 64;
        ctx[i] = (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
    }
}

Analyser::Analyser(int width, int height) : w_(width), h_(height), rows_(height / 64), cols_(width / 64)
{
    const size_t n = (size_t)rows_ * cols_ * 5;
    sum_cost.assign(n, 0.0);
    avg_cost.assign(n, 0.0);
    count.assign(n, 0);
    frac_bits.assign(rows_, 0);
}

int Analyser::p_picture(int qp, Evaluator &ev, PictureResult &out) { return picture(qp, false, 0, ev, out); }
int Analyser::i_picture(int qp, int strong_intra_smoothing, Evaluator &ev, PictureResult &out) { return picture(qp, true, strong_intra_smoothing, ev, out); }

int Analyser::picture(int qp, bool intra, int strong, Evaluator &ev, PictureResult &out)
{
    if (w_ <= 0 || h_ <= 0 || w_ % 64 || h_ % 64 || qp < 0 || qp > 51) return -1;
    const size_t n4 = (size_t)(w_ / 4) * (h_ / 4), n8 = (size_t)(w_ / 8) * (h_ / 8), ny = (size_t)w_ * h_, nctu = (size_t)rows_ * cols_;
    out.info0.assign(n4, 0), out.info1.assign(n4, 0), out.mode.assign(n4, 0);
    out.mv.assign(n8, Mv{ 0, 0 }), out.mvd.assign(n8, Mv{ 0, 0 }), out.mvp_idx.assign(n8, 0);
    out.coef_y.assign(ny, 0), out.coef_u.assign(ny / 4, 0), out.coef_v.assign(ny / 4, 0);
    out.rec_y.assign(ny, 0), out.rec_u.assign(ny / 4, 0), out.rec_v.assign(ny / 4, 0);
    out.rd_cost.assign(nctu, 0);
    out.ctx_in.assign(nctu * 256, 0), out.ctx_out.assign(nctu * 256, 0);
    out.evaluator_calls = out.cu_requests = out.chroma_requests = 0;

    Shared S;
    S.an = this, S.out = &out;
    S.w = w_, S.h = h_, S.cols = cols_, S.nrows = rows_, S.qp = qp;
    S.done.assign(rows_, 0);
    S.done_luma.assign(rows_, 0);
    S.done_seen.reset(new std::atomic<int>[rows_]);
    for (int y = 0; y < rows_; y++) S.done_seen[y].store(0, std::memory_order_relaxed);
    S.saved.resize(rows_);
    S.intra = intra, S.strong = strong;
    init_contexts(intra ? 2 : 1, qp, S.start.m);
    S.rows = std::deque<Row>(rows_);
    S.spin_us = spin_us;
    S.grace_us = grace_us;
    S.ev = &ev;
    S.speculate = speculate;
    S.defer_chroma = speculate != 0;
    S.pairs = speculate != 0 && std::getenv("XMI_EXP_NO_PAIRS") == nullptr;
    std::vector<uint16_t> ctu_requests;
    if (std::getenv("XMI_EXP_CTU_REQUESTS")) ctu_requests.assign(nctu, 0), S.ctu_requests = ctu_requests.data();
    const int nl = lanes < 1 ? 1 : lanes > rows_ ? rows_ : lanes;
    S.lanes = std::deque<Lane>(nl);
    for (int y = 0; y < rows_; y++) {
        S.rows[y].sh = &S, S.rows[y].y = y;
        S.lanes[0].rows++, S.lanes[0].running++;
    }
    // Worker threads: one per CTU row while the host has cores for them, fewer otherwise (a core is left to each coordinator).
    // Worker w runs the rows w, w + workers, ... as fibers, going round those that can go on; when none can for the polling
    // time, it sleeps.
    int nw = workers > 0 ? workers : host_cores() - nl;
    nw = nw < 1 ? 1 : nw > rows_ ? rows_ : nw;
    auto work = [&S, nw, this](int w) {
        ucontext_t home;
        std::vector<Row *> mine;
        for (int y = w; y < rows_; y += nw) mine.push_back(&S.rows[y]);
        size_t left = mine.size();
        auto any = [&] {
            for (Row *r : mine)
                if (!r->done && r->runnable()) return true;
            return false;
        };
        while (left) {
            bool ran = false;
            for (Row *r : mine) {
                if (r->done || !r->runnable()) continue;
                r->home = &home;
                if (!r->started) r->make_fiber();
                XMI_TSAN_FIBER(r->san_home = __tsan_get_current_fiber());
                XMI_TSAN_FIBER(__tsan_switch_to_fiber(r->san_fiber, 0));
                swapcontext(&home, &r->fiber);
                ran = true;
                if (r->done) {
                    left--;
                    XMI_TSAN_FIBER(__tsan_destroy_fiber(r->san_fiber));
                }
            }
            if (!ran && left && !poll_until(S.spin_us, any)) {
                std::unique_lock<std::mutex> lk(S.m);
                S.sleepers.fetch_add(1, std::memory_order_relaxed);
                S.cv_rows.wait(lk, any);
                S.sleepers.fetch_sub(1, std::memory_order_relaxed);
            }
        }
    };
    std::vector<std::thread> threads;
    threads.reserve(nw + nl);
    for (int w = 0; w < nw; w++) threads.emplace_back(work, w);

    // A coordinator goes when something is posted and nobody reads its last answers any more - at once when no row is busy on
    // the host, else (polling configurations only: a sleeping coordinator is woken when the last busy row posts) after a short
    // wait for the rows that are about to post, so that they share the call.
    auto coordinate = [&S, &ev, &out, nl](int lane) {
        Lane &ln = S.lanes[0], &mine = S.lanes[lane];
        std::vector<CuEval *> cu;
        std::vector<ChromaEval *> chroma;
        std::vector<IntraEval *> pu;
        std::vector<IntraChromaEval *> ichroma;
        std::vector<IntraQuadEval *> quads;
        std::vector<IntraChromaQuadEval *> cquads;
        std::vector<Row *> served;
        auto over = [&] { return ln.finished.load() == ln.rows || S.error.load() != 0; };
        auto idle = [&] { return ln.running.load(std::memory_order_acquire) == 0; };
        auto work = [&] { return ln.posted.load(std::memory_order_acquire) > 0 && mine.readers.load(std::memory_order_acquire) == 0; };
        auto ready = [&] { return idle() && (work() || over()); };
        for (;;) {
            bool go = false;
            if (S.spin_us > 0 && nl > 1) {
                const auto t0 = std::chrono::steady_clock::now();
                auto seen = t0;
                bool waiting = false;
                for (;;) {
                    if (ready()) { go = true; break; }
                    const auto now = std::chrono::steady_clock::now();
                    if (work()) {
                        if (!waiting) waiting = true, seen = now;
                        if (now - seen > std::chrono::microseconds(S.grace_us)) { go = true; break; }
                    } else {
                        waiting = false;
                    }
                    if (now - t0 > std::chrono::microseconds(S.spin_us)) break;
                    __builtin_ia32_pause();
                }
            } else {
                go = poll_until(S.spin_us, ready);
            }
            if (!go) {
                std::unique_lock<std::mutex> lk(S.m);
                ln.cv_main.wait(lk, ready);
            }
            if (ln.finished.load() == ln.rows) break;
            // the rows this call serves: a row that another lane's progress wakes during the call asks for the next one
            cu.clear(), chroma.clear(), pu.clear(), ichroma.clear(), quads.clear(), cquads.clear(), served.clear();
            for (Row &r : S.rows) {
                if (!r.pending.load(std::memory_order_acquire) || r.in_flight.exchange(true, std::memory_order_acq_rel)) continue;
                if (!r.pending.load(std::memory_order_acquire)) {      // answered by another coordinator between the two looks
                    r.in_flight.store(false, std::memory_order_release);
                    continue;
                }
                ln.posted.fetch_sub(1, std::memory_order_acq_rel);
                served.push_back(&r);
                for (int i = 0; i < r.n_req_cu; i++) cu.push_back(r.req_cu[i]);
                if (r.req_chroma)
                    for (ChromaEval &c : r.chroma_req) chroma.push_back(&c);
                for (int i = 0; i < r.n_req_in; i++) pu.push_back(r.req_in[i]);
                if (r.req_ichroma) ichroma.push_back(&r.ichroma);
                if (r.req_quad) quads.push_back(&r.quad);
                if (r.req_pair)
                    for (int h2 = 0; h2 < 2; h2++) pu.push_back(&r.pair_in[h2]), quads.push_back(&r.pair_quad[h2]);
                if (r.req_cquad) cquads.push_back(&r.cquad);
            }
            if (served.empty() && !S.error) continue;      // a row is between counting itself out and raising its flag
            int rc = 0;
            if (!S.error && (!cu.empty() || !chroma.empty())) rc = ev.evaluate(lane, cu.data(), (int)cu.size(), chroma.data(), (int)chroma.size());
            if (!S.error && !rc && (!pu.empty() || !ichroma.empty() || !quads.empty() || !cquads.empty())) {
                IntraBatch b;
                b.pu = pu.data(), b.npu = (int)pu.size(), b.chroma = ichroma.data(), b.nchroma = (int)ichroma.size();
                b.quad = quads.data(), b.nquad = (int)quads.size(), b.cquad = cquads.data(), b.ncquad = (int)cquads.size();
                rc = ev.evaluate_intra(lane, b);
            }
            if (rc && !S.error) S.error = rc;
            for (Row *r : served) {
                ln.running.fetch_add(1, std::memory_order_acq_rel);
                r->served_by = lane;
                mine.readers.fetch_add(1, std::memory_order_acq_rel);
                // `pending` falls before the claim: a coordinator that looks in between finds the row claimed and leaves it
                r->pending.store(false, std::memory_order_release);
                r->in_flight.store(false, std::memory_order_release);
            }
            {
                std::lock_guard<std::mutex> lk(S.m);      // also orders the stores above before the wake-up of a sleeping row
                out.evaluator_calls++, out.cu_requests += cu.size() + pu.size() + quads.size(), out.chroma_requests += chroma.size() + ichroma.size() + cquads.size();
                if (S.error) {       // an error ends the picture: rows waiting for a neighbour leave as well
                    for (Row &r : S.rows)
                        if (r.need.load() >= 0) ln.running.fetch_add(1), r.need.store(-1);
                    ln.cv_main.notify_all();
                    S.cv_rows.notify_all();
                }
            }
            if (S.sleepers.load(std::memory_order_relaxed) > 0 || S.error) S.cv_rows.notify_all();
        }
    };
    for (int l = 1; l < nl; l++) threads.emplace_back(coordinate, l);
    coordinate(0);
    for (std::thread &t : threads) t.join();
    out.row_busy_ns = out.row_requests = 0;
    for (const Row &r : S.rows) out.row_busy_ns += r.busy_ns, out.row_requests += r.requests;
    if (S.ctu_requests) {      // one line per picture: type, columns, then the counts row by row
        std::string line = intra ? "CTU_REQUESTS I " : "CTU_REQUESTS P ";
        line += std::to_string(cols_);
        for (size_t i = 0; i < nctu; i++) line += " " + std::to_string(ctu_requests[i]);
        std::fprintf(stderr, "%s\n", line.c_str());
    }
    if (intra && std::getenv("XMI_EXP_PAIR_STATS")) {
        uint64_t taken = 0;
        for (const Row &r : S.rows) taken += r.pairs_taken;
        std::fprintf(stderr, "I picture: %llu requests of rows, %llu calls, %llu blocks answered with the block to their left\n", (unsigned long long)out.row_requests,
                     (unsigned long long)out.evaluator_calls, (unsigned long long)taken);
    }
#ifdef XMI_EXP_MISS_STATS
    if (!intra) {
        std::fprintf(stderr, "requests that had to be made, by CU size 8/16/32/64 (not asked ahead | guess failed | answers used): ");
        for (int l = 0; l < 4; l++) std::fprintf(stderr, "%ld|%ld|%ld  ", g_miss[l][0].exchange(0), g_miss[l][1].exchange(0), g_hit[l].exchange(0));
        std::fprintf(stderr, "; failed guesses by what differed (1 count, 2 merge vector, 4 AMVP): ");
        for (int k = 1; k < 8; k++) std::fprintf(stderr, "%d:%ld ", k, g_miss_what[k].exchange(0));
        std::fprintf(stderr, "\n");
    }
#endif
#ifdef XMI_EXP_HOST_TIMING
    {
        uint64_t c[4] = {};
        for (const Row &r : S.rows)
            for (int k = 0; k < 4; k++) c[k] += r.cat[k];
        std::fprintf(stderr, "host us per CTU: busy %.2f, prologue %.2f, speculation %.2f, epilogue %.2f, chroma apply %.2f; calls %llu\n", out.row_busy_ns / 1e3 / nctu,
                     c[0] / 1e3 / nctu, c[1] / 1e3 / nctu, c[2] / 1e3 / nctu, c[3] / 1e3 / nctu, (unsigned long long)out.row_requests);
    }
#endif
    return S.error.load();
}

const uint16_t *scan_table(int scan_idx, int log2_size) { return tables().scan[scan_idx][log2_size - 2]; }
const uint8_t *group_table(int scan_idx, int log2_size) { return tables().group[scan_idx][log2_size - 2]; }

}  // namespace analysis

namespace syntax {
const uint16_t *coef_scan(int scan_idx, int log2_size) { return analysis::scan_table(scan_idx, log2_size); }
const uint8_t *group_scan(int scan_idx, int log2_size) { return analysis::group_table(scan_idx, log2_size); }
}  // namespace syntax
}  // namespace xmi
