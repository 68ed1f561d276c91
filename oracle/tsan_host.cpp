// Test infrastructure, not product code.
//
// The host side of the encoder (x265_amd/csrc/host/*.cpp: CTU rows as fibers of worker threads, coordinators, lock-free
// hand-over, hashing threads) under ThreadSanitizer, bound to the CPU evaluator of analysis_check.cpp: several encoder handles
// encode the same small clip at the same time from threads of their own - what one GPU does for a rank's GOP segments - in
// the one- and two-step form, with one and two coordinators, sleeping and polling waiters. Every stream must equal the first; the sanitizer reports data
// races on the way (oracle/Makefile target tsan builds everything with -fsanitize=thread; TSAN_OPTIONS=halt_on_error=1).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
void *orc_encoder_create(int width, int height, int qp, int strong_intra_smoothing);
void orc_encoder_destroy(void *h);
int orc_encoder_configure(void *h, int lanes, int spin_us);
int orc_encoder_workers(void *h, int workers);
int orc_encoder_encode(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int slice_type, uint8_t *out, size_t cap);
int orc_encoder_submit(void *h, const uint8_t *y, const uint8_t *u, const uint8_t *v, int slice_type);
int orc_encoder_collect(void *h, uint8_t *out, size_t cap);
}

namespace {
constexpr int W = 256, H = 192, PICTURES = 3;

struct Picture {
    std::vector<uint8_t> y, u, v;
};

// a textured picture that moves by (2, 1) samples per picture, with a little noise
std::vector<Picture> clip()
{
    std::vector<Picture> pics(PICTURES);
    uint32_t s = 12345;
    auto rnd = [&] { return (s = s * 1664525u + 1013904223u) >> 24; };
    for (int t = 0; t < PICTURES; t++) {
        Picture &p = pics[t];
        p.y.resize((size_t)W * H), p.u.resize((size_t)W * H / 4), p.v.resize((size_t)W * H / 4);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int xx = x + 2 * t, yy = y + t;
                p.y[(size_t)y * W + x] = (uint8_t)(96 + ((xx * 7 + yy * 3) & 63) + ((xx / 16 + yy / 16) & 1) * 40 + (rnd() & 3));
            }
        for (size_t i = 0; i < p.u.size(); i++) p.u[i] = (uint8_t)(120 + (rnd() & 7)), p.v[i] = (uint8_t)(130 + (rnd() & 7));
    }
    return pics;
}

std::vector<uint8_t> encode(const std::vector<Picture> &pics, int lanes, int workers, bool two_step, int spin_us)
{
    void *e = orc_encoder_create(W, H, 32, 0);
    orc_encoder_configure(e, lanes, spin_us);
    orc_encoder_workers(e, workers);
    std::vector<uint8_t> out((size_t)W * H * 3 + 8192), stream;
    auto take = [&](int n) {
        if (n <= 0) std::fprintf(stderr, "encode failed: %d\n", n), std::exit(2);
        stream.insert(stream.end(), out.begin(), out.begin() + n);
    };
    for (int t = 0; t < PICTURES; t++) {
        const Picture &p = pics[t];
        if (two_step) {
            if (orc_encoder_submit(e, p.y.data(), p.u.data(), p.v.data(), t == 0 ? 2 : 1) != 0) std::fprintf(stderr, "submit failed\n"), std::exit(2);
            if (t) take(orc_encoder_collect(e, out.data(), out.size()));
        } else {
            take(orc_encoder_encode(e, p.y.data(), p.u.data(), p.v.data(), t == 0 ? 2 : 1, out.data(), out.size()));
        }
    }
    if (two_step) take(orc_encoder_collect(e, out.data(), out.size()));
    orc_encoder_destroy(e);
    return stream;
}
}      // namespace

int main()
{
    const std::vector<Picture> pics = clip();
    const std::vector<uint8_t> first = encode(pics, 1, 1, false, 0);
    constexpr int N = 6;
    std::vector<std::vector<uint8_t>> got(N);
    std::vector<std::thread> th;
    for (int i = 0; i < N; i++) th.emplace_back([&, i] { got[i] = encode(pics, 1 + (i & 1), 1 + i % 3, (i & 2) != 0, i >= 3 ? 40 : 0); });
    for (std::thread &t : th) t.join();
    for (int i = 0; i < N; i++)
        if (got[i] != first) {
            std::fprintf(stderr, "handle %d wrote another stream (%zu bytes, first %zu)\n", i, got[i].size(), first.size());
            return 1;
        }
    std::printf("ok: %d concurrent handles, %zu bytes each\n", N, first.size());
    return 0;
}
