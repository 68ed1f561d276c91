// MV predictor and merge candidate lists for gfx950: xFillMvpCand (src/encode.cpp:2703-2809) for a batch of PUs.
//
// The reference derives, per PU, two AMVP candidates (what xMotionSearch is given as sMVP) and up to five merge
// candidates from five neighbours - below-left, left, above-right, above, above-left - whose mode flags (one per 4x4
// block, MODE_INVALID = not coded or outside) and motion vectors (one per 8x8 block) it reads from the CTU-local caches.
// Here the flags and vectors are picture-level maps in device memory, so the PUs of a whole anti-diagonal of CTUs (or any
// other set whose neighbours are final) are served by one launch: one thread per request, the reference's statements in
// order. Entries of the merge list past the returned count are zero.
#include "xmi_common.h"

#include <cstring>
#include <mutex>

namespace {

struct MvpParams {
    const uint8_t *valid;
    const short2 *mv;
    const xmi_mvp_request *req;
    xmi_mvp_result *res;
    int width, height, n;
};

struct Nb {
    bool ok;
    short2 v;
};

// flag of the 4x4 block at luma position (fx, fy), vector of the 8x8 block at (vx, vy); both outside the picture: invalid
__device__ __forceinline__ Nb neighbour(const MvpParams &p, int fx, int fy, int vx, int vy)
{
    Nb n = { false, make_short2(0, 0) };
    if (fx >= 0 && fy >= 0 && fx < p.width && fy < p.height) n.ok = p.valid[(fy >> 2) * (p.width >> 2) + (fx >> 2)] != 0;
    if (vx >= 0 && vy >= 0 && vx < p.width && vy < p.height) n.v = p.mv[(vy >> 3) * (p.width >> 3) + (vx >> 3)];
    return n;
}

__device__ __forceinline__ bool differ(const Nb &a, const Nb &b) { return a.v.x != b.v.x || a.v.y != b.v.y; }

__global__ void __launch_bounds__(256) mvp_cand_kernel(const MvpParams p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const xmi_mvp_request rq = p.req[i];
    const int x0 = rq.x0, y0 = rq.y0, sx = rq.size_x, sy = rq.size_y;
    xmi_mvp_result out = {};
    if (x0 < 0 || y0 < 0 || sx < 8 || sy < 8 || sx > 64 || sy > 64 || x0 + sx > p.width || y0 + sy > p.height) {
        out.merge_count = -1;                          // no PU of this picture: marked, so that the caller can tell
        p.res[i] = out;
        return;
    }
    const bool no_b1 = rq.pu_mode == 1 && rq.pu_idx == 1, no_a1 = rq.pu_mode == 2 && rq.pu_idx == 1;      // :2713-2714
    // the cache entries of :2721-2732: flags at 4x4, vectors at 8x8 granularity relative to the PU
    const Nb bl = neighbour(p, x0 - 1, y0 + sy, x0 - 1, y0 + 8 * (sy / 8));
    const Nb l = neighbour(p, x0 - 1, y0 + sy - 4, x0 - 1, y0 + 8 * (sy / 8 - 1));
    const Nb tr = neighbour(p, x0 + sx, y0 - 1, x0 + 8 * (sx / 8), y0 - 1);
    const Nb t = neighbour(p, x0 + sx - 4, y0 - 1, x0 + 8 * (sx / 8 - 1), y0 - 1);
    const Nb tl = neighbour(p, x0 - 1, y0 - 1, x0 - 1, y0 - 1);
    int n = 0;
    const bool grp_l = bl.ok || l.ok, grp_t = tr.ok || t.ok || tl.ok;
    short2 a[2] = { make_short2(0, 0), make_short2(0, 0) };
    if (grp_l) a[n++] = bl.ok ? bl.v : l.v;
    if (grp_t) {
        const short2 c = tr.ok ? tr.v : t.ok ? t.v : tl.v;
        if (n == 0) a[0] = c; else a[1] = c;
        n++;
    }
    if (grp_l && grp_t && a[0].x == a[1].x && a[0].y == a[1].y) a[1] = make_short2(0, 0);      // :2748-2752
    if (n < 2) a[1] = make_short2(0, 0);                                                         // :2756-2758
    out.amvp[0][0] = a[0].x; out.amvp[0][1] = a[0].y;
    out.amvp[1][0] = a[1].x; out.amvp[1][1] = a[1].y;
    n = 0;
    auto push = [&](short2 v) {
        out.merge[n][0] = v.x;
        out.merge[n][1] = v.y;
        n++;
    };
    if (!no_a1 && l.ok) push(l.v);
    if (!no_b1 && t.ok && (!l.ok || no_a1 || differ(l, t))) push(t.v);
    if (tr.ok && (!t.ok || no_b1 || differ(t, tr))) push(tr.v);
    if (bl.ok && (!l.ok || no_a1 || differ(l, bl))) push(bl.v);
    if (n < 4) {
        if (tl.ok && (!l.ok || no_a1 || differ(l, tl)) && (!t.ok || no_b1 || differ(t, tl))) {
            push(tl.v);
        } else {                                       // a zero vector unless the list holds one already, :2794-2805
            bool zero = false;
            for (int k = 0; k < n; k++) zero |= out.merge[k][0] == 0 && out.merge[k][1] == 0;
            if (!zero) push(make_short2(0, 0));
        }
    }
    out.merge_count = n;
    p.res[i] = out;
}

}  // namespace

extern "C" int xmi_mvp_candidates_dev(int width, int height, const uint8_t *d_valid, const int16_t *d_mv, const xmi_mvp_request *d_req, int n,
                                      xmi_mvp_result *d_res, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_mvp_candidates_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(width >= 8 && height >= 8 && width % 8 == 0 && height % 8 == 0, "xmi_mvp_candidates_dev: picture %dx%d must be a multiple of 8", width, height);
    XMI_ARG(d_valid && d_mv && d_req && d_res, "xmi_mvp_candidates_dev: null argument");
    XMI_ARG(((uintptr_t)d_mv & 3) == 0, "xmi_mvp_candidates_dev: motion vectors must be 4-byte aligned");
    MvpParams p = { d_valid, reinterpret_cast<const short2 *>(d_mv), d_req, d_res, width, height, n };
    hipLaunchKernelGGL(mvp_cand_kernel, dim3((n + 255) / 256), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

// Search requests from candidate lists, on the device: request i = PU i's origin and size (square PUs), its AMVP pair, the
// clip box of its CTU (xCuMvClip, src/encode.cpp:3794-3812) and the caller's lambda term; what xEncDecideCU_Inter puts
// together before it calls xMotionSearch (src/encode.cpp:3379). With it the chain maps -> candidates -> requests ->
// searches needs no host table in between.
struct BuildParams {
    const xmi_mvp_request *pu;
    const xmi_mvp_result *cand;
    const uint8_t *zero_tested;
    xmi_me_request *req;
    int n, ctus_w, ctus_h;
    uint32_t sqrt_lambda;
};

__global__ void __launch_bounds__(256) me_request_kernel(const BuildParams p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const xmi_mvp_request pu = p.pu[i];
    const xmi_mvp_result c = p.cand[i];
    xmi_me_request rq = {};
    rq.x0 = pu.x0;
    rq.y0 = pu.y0;
    // a PU the candidate kernel refused, or one that is not square, becomes a request the search kernel refuses (size 0)
    rq.size = (c.merge_count >= 0 && pu.size_x == pu.size_y) ? pu.size_x : 0;
    rq.mvp[0][0] = c.amvp[0][0]; rq.mvp[0][1] = c.amvp[0][1];
    rq.mvp[1][0] = c.amvp[1][0]; rq.mvp[1][1] = c.amvp[1][1];
    rq.zero_mv_tested = p.zero_tested ? p.zero_tested[i] != 0 : 0;
    const int cx = pu.x0 >> 6, cy = pu.y0 >> 6;
    auto clip = [](int v) { return v < -64 ? -64 : v > 64 ? 64 : v; };
    rq.mv_min_ipel[0] = (int16_t)clip(-64 * cx - 64);
    rq.mv_max_ipel[0] = (int16_t)clip(64 * (p.ctus_w - cx));
    rq.mv_min_ipel[1] = (int16_t)clip(-64 * cy - 64);
    rq.mv_max_ipel[1] = (int16_t)clip(64 * (p.ctus_h - cy));
    rq.sqrt_lambda = p.sqrt_lambda;
    p.req[i] = rq;
}

// The way back: the vectors the searches returned (or any final decision in the same form) stored into the picture-level
// maps - the PU's area of 8x8 vector entries and of 4x4 coded flags -, which is what the reference's CU write-back does
// for its caches (xEncCacheStoreCU / the MV cache updates of xEncDecideCU_Inter). One thread per (PU, 8x8 cell).
struct StoreParams {
    const xmi_mvp_request *pu;
    const xmi_me_result *res;
    uint8_t *valid;
    short2 *mv;
    int n, width, height;
};

__global__ void __launch_bounds__(256) mv_store_kernel(const StoreParams p)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), cell = threadIdx.x & 63;
    if (i >= p.n) return;
    const xmi_mvp_request pu = p.pu[i];
    if (pu.x0 < 0 || pu.y0 < 0 || (pu.x0 & 7) || (pu.y0 & 7) || pu.size_x < 8 || pu.size_y < 8 || pu.size_x > 64 || pu.size_y > 64 ||
        pu.x0 + pu.size_x > p.width || pu.y0 + pu.size_y > p.height)
        return;
    const xmi_me_result r = p.res[i];
    if (r.mvp_idx < 0) return;                               // a refused search has no vector
    const int cw = pu.size_x >> 3, ch = pu.size_y >> 3;
    if (cell >= cw * ch) return;
    const int cx = (pu.x0 >> 3) + cell % cw, cy = (pu.y0 >> 3) + cell / cw;
    p.mv[cy * (p.width >> 3) + cx] = make_short2(r.mv[0], r.mv[1]);
    uint8_t *v = p.valid + (2 * cy) * (p.width >> 2) + 2 * cx;
    v[0] = v[1] = 1;
    v[p.width >> 2] = v[(p.width >> 2) + 1] = 1;
}

// Per-call form with the reference's own arguments (src/encode.cpp:2703-2711): mode_y and mv point at the PU's entries of the
// CTU-local caches (pitch 18 flags per 4x4 row, MODE_INVALID = 255; pitch 10 vectors per 8x8 row). The five flag and
// five vector entries the function addresses are copied into a small picture of their own - the PU at (8, 8), one 4x4
// block of border around it - and the batch kernel runs on one request. Returns the merge count; writes mvp[0], mvp[1]
// and mvp[2 .. 2 + count), nothing beyond, as the reference.
extern "C" int xmi_fill_mvp_cand(int16_t (*mvp)[2], const uint8_t *mode_y, const int16_t (*mv)[2], uint32_t size_x, uint32_t size_y, uint32_t pu_mode,
                                 uint32_t pu_idx)
{
    int rc = xmi::enter_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(mvp && mode_y && mv, "xmi_fill_mvp_cand: null argument");
    XMI_ARG(size_x >= 8 && size_y >= 8 && size_x <= 64 && size_y <= 64 && size_x % 8 == 0 && size_y % 8 == 0, "xmi_fill_mvp_cand: PU %ux%u", size_x, size_y);
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    constexpr int W = 8 + 64 + 8, W4 = W / 4, W8 = W / 8;         // PU at (8, 8) of an 80 x 80 picture
    const int bx = (int)size_x / 4, by = (int)size_y / 4, mx = (int)size_x / 8, my = (int)size_y / 8;
    // staging: valid map | vector map | request | result
    const size_t off_mv = 512, off_req = off_mv + (size_t)W8 * W8 * 4, off_res = off_req + 16, total = off_res + sizeof(xmi_mvp_result);
    static_assert((size_t)W4 * W4 <= 512 && sizeof(xmi_mvp_request) <= 16, "staging layout");
    uint8_t host[512 + W8 * W8 * 4 + 16];
    memset(host, 0, sizeof host);
    uint8_t *valid = host;
    int16_t *vec = reinterpret_cast<int16_t *>(host + off_mv);
    auto flag = [&](int r, int c) { valid[(2 + r) * W4 + 2 + c] = mode_y[r * 18 + c] != 255; };        // 4x4 units relative to the PU
    auto vect = [&](int r, int c) {                                                                   // 8x8 units relative to the PU
        vec[2 * ((1 + r) * W8 + 1 + c)] = mv[r * 10 + c][0];
        vec[2 * ((1 + r) * W8 + 1 + c) + 1] = mv[r * 10 + c][1];
    };
    flag(by, -1); flag(by - 1, -1); flag(-1, bx); flag(-1, bx - 1); flag(-1, -1);
    vect(my, -1); vect(my - 1, -1); vect(-1, mx); vect(-1, mx - 1); vect(-1, -1);
    xmi_mvp_request rq = { 8, 8, (uint8_t)size_x, (uint8_t)size_y, (uint8_t)pu_mode, (uint8_t)pu_idx };
    memcpy(host + off_req, &rq, sizeof rq);
    uint8_t *dev = static_cast<uint8_t *>(xmi::scratch(6, total));
    if (!dev) return XMI_ERR_HIP;
    XMI_HIP(hipMemcpy(dev, host, sizeof host, hipMemcpyHostToDevice));
    // the picture is 80 x 80 only when the PU is 64 x 64; a smaller PU leaves the rest of the maps unused (all zero, i.e. not coded)
    rc = xmi_mvp_candidates_dev(W, W, dev, reinterpret_cast<const int16_t *>(dev + off_mv), reinterpret_cast<const xmi_mvp_request *>(dev + off_req), 1,
                                reinterpret_cast<xmi_mvp_result *>(dev + off_res), nullptr);
    if (rc != XMI_OK) return rc;
    xmi_mvp_result res;
    XMI_HIP(hipMemcpy(&res, dev + off_res, sizeof res, hipMemcpyDeviceToHost));
    XMI_ARG(res.merge_count >= 0, "xmi_fill_mvp_cand: request refused");
    memcpy(mvp, res.amvp, sizeof res.amvp);
    memcpy(mvp + 2, res.merge, (size_t)res.merge_count * 4);
    return res.merge_count;
}

extern "C" int xmi_mv_field_store_dev(int width, int height, const xmi_mvp_request *d_pu, const xmi_me_result *d_res, int n, uint8_t *d_valid,
                                      int16_t *d_mv, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_mv_field_store_dev: negative count");
    if (n == 0) return XMI_OK;
    XMI_ARG(width >= 8 && height >= 8 && width % 8 == 0 && height % 8 == 0, "xmi_mv_field_store_dev: picture %dx%d must be a multiple of 8", width, height);
    XMI_ARG(d_pu && d_res && d_valid && d_mv && ((uintptr_t)d_mv & 3) == 0, "xmi_mv_field_store_dev: null or unaligned argument");
    StoreParams p = { d_pu, d_res, d_valid, reinterpret_cast<short2 *>(d_mv), n, width, height };
    hipLaunchKernelGGL(mv_store_kernel, dim3((n + 3) / 4), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

extern "C" int xmi_me_requests_from_candidates_dev(int width, int height, const xmi_mvp_request *d_pu, const xmi_mvp_result *d_cand,
                                                   const uint8_t *d_zero_mv_tested, int n, uint32_t sqrt_lambda, xmi_me_request *d_req, void *stream)
{
    int rc = xmi::require_device();
    if (rc != XMI_OK) return rc;
    XMI_ARG(n >= 0, "xmi_me_requests_from_candidates_dev: negative request count");
    if (n == 0) return XMI_OK;
    XMI_ARG(width >= 64 && height >= 64 && width % 64 == 0 && height % 64 == 0, "xmi_me_requests_from_candidates_dev: picture %dx%d is not a multiple of the 64x64 CTU",
            width, height);
    XMI_ARG(d_pu && d_cand && d_req, "xmi_me_requests_from_candidates_dev: null argument");
    BuildParams p = { d_pu, d_cand, d_zero_mv_tested, d_req, n, width / 64, height / 64, sqrt_lambda };
    hipLaunchKernelGGL(me_request_kernel, dim3((n + 255) / 256), dim3(256), 0, xmi::as_stream(stream), p);
    XMI_HIP(hipGetLastError());
    return XMI_OK;
}

// xCuMvClip (src/encode.cpp:3794-3812): the integer clip box of every search in CTU (ctu_x, ctu_y) of a picture of
// ctus_w x ctus_h CTUs, with the reference's constants (MAX_IMV_VAL 64, MAX_CU_SIZE 64, PAD_YSIZE 80). The reference
// refreshes the y range only at ctu_x == 0, which gives every CTU of a row the row's range. Host arithmetic, no device.
extern "C" int xmi_cu_mv_clip(int ctu_x, int ctu_y, int ctus_w, int ctus_h, int16_t mv_min_ipel[2], int16_t mv_max_ipel[2])
{
    XMI_ARG(mv_min_ipel && mv_max_ipel, "xmi_cu_mv_clip: null argument");
    XMI_ARG(ctus_w > 0 && ctus_h > 0 && ctu_x >= 0 && ctu_y >= 0 && ctu_x < ctus_w && ctu_y < ctus_h, "xmi_cu_mv_clip: CTU %d,%d of %dx%d", ctu_x, ctu_y,
            ctus_w, ctus_h);
    auto clip = [](int v) { return v < -64 ? -64 : v > 64 ? 64 : v; };
    mv_min_ipel[0] = (int16_t)clip(-64 * ctu_x - (80 - 16));
    mv_max_ipel[0] = (int16_t)clip(64 * (ctus_w - ctu_x) + (80 - 64) - 16);
    mv_min_ipel[1] = (int16_t)clip(-64 * ctu_y - (80 - 16));
    mv_max_ipel[1] = (int16_t)clip(64 * (ctus_h - ctu_y) + (80 - 64) - 16);
    return XMI_OK;
}
