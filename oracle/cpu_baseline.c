/*
 * cpu_baseline.c - times the CPU side of the SAD-surface workload (TEST/BENCH INFRASTRUCTURE).
 *
 * Reads two raw luma frames (cur: W x H, ref: (W+2*pad) x (H+2*pad)) from a file and evaluates, for
 * the first `nctu` CTUs and every integer MV in [-R,R]^2, the 85 square-PU SADs - the same candidate
 * list the GPU kernel covers - by calling the per-block SAD primitive once per candidate, exactly as
 * the reference's xMotionSearch does (src/encode.cpp:2885-2931).
 *
 *   kind "reference": the SAD functions are the reference's own compiled xSad_AMP[] table
 *                     (src/pixel.cpp:833-844) loaded from oracle/_ref/libx265ref_pixel.so.
 *   kind "port":      this repo's restatement (orc_sad), used when the reference library is absent.
 *
 * Output: one line  "kind=<k> cands=<n> seconds=<t> checksum=<c>".
 * usage: cpu_baseline <frames.bin> W H pad R nctu [ref_lib.so]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "x265_oracle.h"

typedef uint32_t (*sad_fn)(const uint8_t *, uint32_t, const uint8_t *, uint32_t);

static uint32_t port8(const uint8_t *a, uint32_t sa, const uint8_t *b, uint32_t sb) { return orc_sad(8, a, (int)sa, b, (int)sb); }
static uint32_t port16(const uint8_t *a, uint32_t sa, const uint8_t *b, uint32_t sb) { return orc_sad(16, a, (int)sa, b, (int)sb); }
static uint32_t port32(const uint8_t *a, uint32_t sa, const uint8_t *b, uint32_t sb) { return orc_sad(32, a, (int)sa, b, (int)sb); }
static uint32_t port64(const uint8_t *a, uint32_t sa, const uint8_t *b, uint32_t sb) { return orc_sad(64, a, (int)sa, b, (int)sb); }

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s frames.bin W H pad R nctu [ref_lib.so]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]), pad = atoi(argv[4]), R = atoi(argv[5]);
    int nctu = atoi(argv[6]);
    const int rs = W + 2 * pad;
    const size_t cur_bytes = (size_t)W * H, ref_bytes = (size_t)rs * (H + 2 * pad);
    uint8_t *cur = NULL, *refb = NULL;
    if (posix_memalign((void **)&cur, 64, cur_bytes) || posix_memalign((void **)&refb, 64, ref_bytes)) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(cur, 1, cur_bytes, f) != cur_bytes || fread(refb, 1, ref_bytes, f) != ref_bytes) {
        fprintf(stderr, "cannot read frames from %s\n", argv[1]);
        return 1;
    }
    fclose(f);
    const uint8_t *ref = refb + (size_t)pad * rs + pad;

    sad_fn fn[4] = { port8, port16, port32, port64 };
    const char *kind = "port";
    if (argc > 7) {
        void *h = dlopen(argv[7], RTLD_NOW);
        void **tab = h ? (void **)dlsym(h, "xSad_AMP") : NULL;
        if (tab && tab[0] && tab[3] && tab[6] && tab[9]) {
            /* index = 2*log2(w) + log2(h) - 9 in the reference's convention: 0, 3, 6, 9 are the squares */
            fn[0] = (sad_fn)tab[0];
            fn[1] = (sad_fn)tab[3];
            fn[2] = (sad_fn)tab[6];
            fn[3] = (sad_fn)tab[9];
            kind = "reference";
        }
    }
    const int wc = W / 64, total = wc * (H / 64);
    if (nctu > total) nctu = total;
    /* the reference's ME reads the current CTU from a 64-stride, 16-byte aligned cache (xCache.aucPixY) */
    uint8_t *cache = NULL;
    if (posix_memalign((void **)&cache, 64, 64 * 64)) return 1;

    struct timespec t0, t1;
    uint64_t checksum = 0, cands = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int c = 0; c < nctu; c++) {
        /* spread the sample over the frame */
        const int ctu = (int)(((long long)c * total) / nctu);
        const int cx = (ctu % wc) * 64, cy = (ctu / wc) * 64;
        for (int y = 0; y < 64; y++)
            for (int x = 0; x < 64; x++) cache[y * 64 + x] = cur[(size_t)(cy + y) * W + cx + x];
        for (int dy = -R; dy <= R; dy++)
            for (int dx = -R; dx <= R; dx++) {
                const uint8_t *r0 = ref + (ptrdiff_t)(cy + dy) * rs + cx + dx;
                for (int lvl = 0; lvl < 4; lvl++) {
                    const int n = 8 << lvl;
                    for (int by = 0; by < 64; by += n)
                        for (int bx = 0; bx < 64; bx += n)
                            checksum += fn[lvl](cache + by * 64 + bx, 64, r0 + (ptrdiff_t)by * rs + bx, (uint32_t)rs);
                }
                cands += 85;
            }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("kind=%s cands=%llu seconds=%.6f checksum=%llu\n", kind, (unsigned long long)cands, sec,
           (unsigned long long)checksum);
    return 0;
}
