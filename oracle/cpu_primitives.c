/*
 * cpu_primitives.c - times the CPU side of the per-block primitives (TEST/BENCH INFRASTRUCTURE).
 *
 * For each primitive and block size it runs the reference's own compiled function
 * (oracle/_ref/libx265ref_pixel.so: src/pixel.cpp + src/table.cpp) over a pool of pseudo-random blocks
 * with the argument pattern of the reference's callers: 64-stride source/prediction/coefficient buffers
 * and an 80-stride reconstruction (src/encode.cpp:3005-3021), one call per block, one thread.
 *
 *   kind "reference": xSadN[] (src/pixel.cpp:711), the HM SATD functions xCalcHADs* (src/pixel.cpp:63-527),
 *                     xSubDct / xQuant / xDeQuant / xIDctAdd (src/pixel.cpp:3446, 1060, 1249, 5938).
 *   kind "port":      this repo's restatement, used where the reference binds NASM routines that cannot be
 *                     assembled here (x264-form SATD 4/8/16, SSD) or when the reference library is absent.
 *
 * Output: one JSON line per (primitive, size):
 *   {"cpu": "<name> NxN", "kind": "...", "Mblk_per_s": x, "GBps": y, "seconds": t}
 * GBps uses the same algorithmic bytes per block as the GPU figures (2N^2+4 for a metric pair, 5N^2+4 for a
 * transform-unit chain, 4N^2(+4) for a single transform stage).
 *
 * usage: cpu_primitives [ref_lib.so] [seconds_per_case]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "x265_oracle.h"

enum { POOL = 512, S = 64, SLICE_P = 1, MODE_NONE = 255 };

typedef uint32_t (*metric_fn)(const uint8_t *, uint32_t, const uint8_t *, uint32_t);
typedef void (*subdct_fn)(int16_t *, uint8_t *, unsigned, uint8_t *, unsigned, int16_t *, int16_t *, int, unsigned);
typedef uint32_t (*quant_fn)(int16_t *, int16_t *, unsigned, unsigned, int, int, int);
typedef void (*dequant_fn)(int16_t *, int16_t *, unsigned, unsigned, int, int);
typedef void (*idct_fn)(uint8_t *, unsigned, int16_t *, unsigned, uint8_t *, int16_t *, int16_t *, int, unsigned);

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }

static void *amalloc(size_t n)
{
    void *p = NULL;
    if (posix_memalign(&p, 64, n)) exit(1);
    memset(p, 0, n);
    return p;
}

/* port trampolines with the table signature */
#define PORT(name, call) \
    static uint32_t name(const uint8_t *a, uint32_t sa, const uint8_t *b, uint32_t sb) { return call; }
PORT(p_sad4, orc_sad(4, a, (int)sa, b, (int)sb))
PORT(p_sad8, orc_sad(8, a, (int)sa, b, (int)sb))
PORT(p_sad16, orc_sad(16, a, (int)sa, b, (int)sb))
PORT(p_sad32, orc_sad(32, a, (int)sa, b, (int)sb))
PORT(p_sad64, orc_sad(64, a, (int)sa, b, (int)sb))
PORT(p_ssd4, orc_ssd(4, a, (int)sa, b, (int)sb))
PORT(p_ssd8, orc_ssd(8, a, (int)sa, b, (int)sb))
PORT(p_ssd16, orc_ssd(16, a, (int)sa, b, (int)sb))
PORT(p_ssd32, orc_ssd(32, a, (int)sa, b, (int)sb))
PORT(p_ssd64, orc_ssd(64, a, (int)sa, b, (int)sb))
PORT(p_hm4, orc_satd_hm(4, a, (int)sa, b, (int)sb))
PORT(p_hm8, orc_satd_hm(8, a, (int)sa, b, (int)sb))
PORT(p_hm16, orc_satd_hm(16, a, (int)sa, b, (int)sb))
PORT(p_hm32, orc_satd_hm(32, a, (int)sa, b, (int)sb))
PORT(p_hm64, orc_satd_hm(64, a, (int)sa, b, (int)sb))
PORT(p_x4, orc_satd_x264(4, a, (int)sa, b, (int)sb))
PORT(p_x8, orc_satd_x264(8, a, (int)sa, b, (int)sb))
PORT(p_x16, orc_satd_x264(16, a, (int)sa, b, (int)sb))

static void p_subdct(int16_t *d, uint8_t *s, unsigned st, uint8_t *p, unsigned sp, int16_t *t0, int16_t *t1, int w, unsigned m)
{
    (void)t0;
    (void)t1;
    orc_sub_dct(d, s, (int)st, p, (int)sp, w, m);
}
static uint32_t p_quant(int16_t *d, int16_t *s, unsigned st, unsigned qp, int w, int h, int sl) { return orc_quant(d, s, (int)st, qp, w, h, sl); }
static void p_dequant(int16_t *d, int16_t *s, unsigned st, unsigned qp, int w, int h) { orc_dequant(d, s, (int)st, qp, w, h); }
static void p_idct(uint8_t *d, unsigned sd, int16_t *s, unsigned st, uint8_t *p, int16_t *t0, int16_t *t1, int w, unsigned m)
{
    (void)t0;
    (void)t1;
    orc_idct_add(d, (int)sd, s, (int)st, p, w, m);
}

static uint8_t *g_pix, *g_pred;   /* POOL blocks of 64x64, stride 64 */
static double g_budget = 0.4;
static volatile uint64_t g_sink;

static void report(const char *name, int n, const char *kind, uint64_t blocks, double sec, double bytes_per_block)
{
    printf("{\"cpu\": \"%s %dx%d\", \"kind\": \"%s\", \"Mblk_per_s\": %.3f, \"GBps\": %.3f, \"seconds\": %.2f}\n", name, n, n, kind,
           (double)blocks / sec / 1e6, (double)blocks * bytes_per_block / sec / 1e9, sec);
    fflush(stdout);
}

static void time_metric(const char *name, int n, metric_fn fn, const char *kind)
{
    uint64_t acc = 0, blocks = 0;
    const double t0 = now();
    double t1;
    do {
        for (int i = 0; i < POOL; i++) acc += fn(g_pix + (size_t)i * S * S, S, g_pred + (size_t)i * S * S, S);
        blocks += POOL;
        t1 = now();
    } while (t1 - t0 < g_budget);
    g_sink += acc;
    report(name, n, kind, blocks, t1 - t0, 2.0 * n * n + 4);
}

int main(int argc, char **argv)
{
    void *h = argc > 1 ? dlopen(argv[1], RTLD_NOW) : NULL;
    if (argc > 2) g_budget = atof(argv[2]);

    g_pix = amalloc((size_t)POOL * S * S);
    g_pred = amalloc((size_t)POOL * S * S);
    uint32_t seed = 12345u;
    for (size_t i = 0; i < (size_t)POOL * S * S; i++) {
        const int v = (int)(lcg(&seed) >> 24), d = (int)(lcg(&seed) >> 24) % 25 - 12;   /* residual in [-12,12] */
        const int q = v + d;
        g_pix[i] = (uint8_t)v;
        g_pred[i] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
    }

    /* ---- block metrics ---- */
    static const metric_fn port_sad[5] = { p_sad4, p_sad8, p_sad16, p_sad32, p_sad64 };
    static const metric_fn port_ssd[5] = { p_ssd4, p_ssd8, p_ssd16, p_ssd32, p_ssd64 };
    static const metric_fn port_hm[5] = { p_hm4, p_hm8, p_hm16, p_hm32, p_hm64 };
    static const metric_fn port_x[3] = { p_x4, p_x8, p_x16 };
    static const char *hm_sym[5] = { "_Z12xCalcHADs4x4PKhjS0_j", "_Z12xCalcHADs8x8PKhjS0_j", "_Z14xCalcHADs16x16PKhjS0_j",
                                     "_Z14xCalcHADs32x32PKhjS0_j", "_Z14xCalcHADs64x64PKhjS0_j" };
    void **sad_tab = h ? (void **)dlsym(h, "xSadN") : NULL;
    for (int i = 0; i < 5; i++) {
        const int n = 4 << i;
        const int ref = sad_tab && sad_tab[i];
        time_metric("sad", n, ref ? (metric_fn)sad_tab[i] : port_sad[i], ref ? "reference" : "port");
    }
    for (int i = 0; i < 5; i++) time_metric("ssd", 4 << i, port_ssd[i], "port");
    for (int i = 0; i < 5; i++) {
        void *f = h ? dlsym(h, hm_sym[i]) : NULL;
        time_metric("satd_hm", 4 << i, f ? (metric_fn)f : port_hm[i], f ? "reference" : "port");
    }
    for (int i = 0; i < 3; i++) time_metric("satd_x264", 4 << i, port_x[i], "port");

    /* ---- transform units ---- */
    subdct_fn fsub = h ? (subdct_fn)dlsym(h, "_Z7xSubDctPsPhjS0_jS_S_ij") : NULL;
    quant_fn fq = h ? (quant_fn)dlsym(h, "_Z6xQuantPsS_jjii10xSliceType") : NULL;
    dequant_fn fdq = h ? (dequant_fn)dlsym(h, "_Z8xDeQuantPsS_jjii") : NULL;
    idct_fn fid = h ? (idct_fn)dlsym(h, "_Z8xIDctAddPhjPsjS_S0_S0_ij") : NULL;
    const char *kind = (fsub && fq && fdq && fid) ? "reference" : "port";
    if (!(fsub && fq && fdq && fid)) {
        fsub = p_subdct;
        fq = p_quant;
        fdq = p_dequant;
        fid = p_idct;
    }
    int16_t *t0 = amalloc((64 + 16) * 64 * sizeof(int16_t)), *t1 = amalloc((64 + 16) * 64 * sizeof(int16_t));
    int16_t *coef = amalloc((size_t)POOL * S * S * sizeof(int16_t));   /* transform output per pool block */
    int16_t *lev = amalloc((size_t)POOL * S * S * sizeof(int16_t));    /* levels per pool block */
    uint8_t *rec = amalloc(80 * 64);
    const unsigned qp = 27;
    for (int n = 4; n <= 32; n *= 2) {
        uint64_t blocks, acc = 0;
        double a, b;
        /* xSubDct: dst and tmp0 alias (src/encode.cpp:3007) */
        blocks = 0;
        a = now();
        do {
            for (int i = 0; i < POOL; i++) {
                int16_t *c = coef + (size_t)i * S * S;
                fsub(c, g_pix + (size_t)i * S * S, S, g_pred + (size_t)i * S * S, S, c, t1, n, MODE_NONE);
            }
            blocks += POOL;
            b = now();
        } while (b - a < g_budget);
        report("sub_dct", n, kind, blocks, b - a, 4.0 * n * n);
        blocks = 0;
        a = now();
        do {
            for (int i = 0; i < POOL; i++) acc += fq(lev + (size_t)i * S * S, coef + (size_t)i * S * S, S, qp, n, n, SLICE_P);
            blocks += POOL;
            b = now();
        } while (b - a < g_budget);
        report("quant", n, kind, blocks, b - a, 4.0 * n * n + 4);
        blocks = 0;
        a = now();
        do {
            for (int i = 0; i < POOL; i++) fdq(coef + (size_t)i * S * S, lev + (size_t)i * S * S, S, qp, n, n);
            blocks += POOL;
            b = now();
        } while (b - a < g_budget);
        report("dequant", n, kind, blocks, b - a, 4.0 * n * n);
        blocks = 0;
        a = now();
        do {
            for (int i = 0; i < POOL; i++) {
                memcpy(t0, coef + (size_t)i * S * S, (size_t)n * S * sizeof(int16_t));   /* the inverse transform overwrites its source rows */
                fid(rec, 80, t0, S, g_pred + (size_t)i * S * S, t1, t0, n, MODE_NONE);
            }
            blocks += POOL;
            b = now();
        } while (b - a < g_budget);
        report("idct_add", n, kind, blocks, b - a, 4.0 * n * n);
        /* the chain of xSubDctQuantRec (src/encode.cpp:3005-3021): inverse path only when a level is non-zero */
        blocks = 0;
        a = now();
        do {
            for (int i = 0; i < POOL; i++) {
                uint8_t *px = g_pix + (size_t)i * S * S, *pr = g_pred + (size_t)i * S * S;
                int16_t *lv = lev + (size_t)i * S * S;
                fsub(t0, px, S, pr, S, t0, t1, n, MODE_NONE);
                const uint32_t sum = fq(lv, t0, S, qp, n, n, SLICE_P);
                if (sum) {
                    fdq(t0, lv, S, qp, n, n);
                    fid(rec, 80, t0, S, pr, t1, t0, n, MODE_NONE);
                } else {
                    for (int y = 0; y < n; y++) memcpy(rec + y * 80, pr + y * S, (size_t)n);
                }
                acc += sum;
            }
            blocks += POOL;
            b = now();
        } while (b - a < g_budget);
        report("tu_chain", n, kind, blocks, b - a, 5.0 * n * n + 4);
        g_sink += acc;
    }
    return 0;
}
