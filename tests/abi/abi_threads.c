/*
 * abi_threads.c -- thread-safety of libpolyhip.so as cgo meets it: calls arrive on arbitrary OS threads, several at
 * once (SURVEY.md 8b "Threading").  8 pthreads run a mix of host-pointer entry points -- mash sketch, distance matrix,
 * SmithWaterman with strings through ONE scoring handle shared by all threads, NeedlemanWunsch, SantaLucia, least
 * rotation, seqhash -- for several rounds, each on its own inputs, and every result must equal the one the main thread
 * computed serially before the threads started.  Plain C, links libpolyhip.so only (no Python, no PyTorch).
 *
 *     abi_threads [NTHREADS [ROUNDS]]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "polyhip.h"

#define NOPS 7
#define MAXT 32
#define NREADS 48
#define RLEN 1500
#define SK 64

static uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct {
    /* inputs (per thread) */
    uint8_t *seqs;          /* NREADS x RLEN DNA */
    uint64_t offs[NREADS + 1];
    uint8_t ref[600];
    uint8_t reads[NREADS * 120];
    uint64_t roffs[NREADS + 1];
    /* outputs: [0] = serial, [1] = concurrent */
    uint32_t sketch[2][NREADS * SK];
    uint16_t counts[2][NREADS * NREADS];
    double dist[2][NREADS * NREADS];
    int64_t score[2][NREADS], nwscore[2][NREADS];
    uint32_t endA[2][NREADS], endB[2][NREADS], err[2][NREADS], alen[2][NREADS], nwerr[2][NREADS], nwlen[2][NREADS];
    uint8_t *alnA[2], *alnB[2], *nwA[2], *nwB[2];
    uint32_t stride, nwstride;
    double tm[2][NREADS], dH[2][NREADS], dS[2][NREADS];
    uint64_t rot[2][NREADS];
    uint8_t *rotated[2];
    char hash[2][NREADS * 72];
    uint32_t herr[2][NREADS];
} Work;

static polyhip_scoring *g_sc;
static Work g_w[MAXT];
static int g_rounds = 6;

static int run_op(Work *w, int op, int which)
{
    switch (op) {
    case 0:
        memset(w->sketch[which], 0, sizeof w->sketch[which]);
        return polyhip_mash_sketch_batch(w->seqs, w->offs, NREADS, 21, SK, w->sketch[which]);
    case 1:
        return polyhip_mash_distance_matrix(w->sketch[0], NREADS, SK, w->sketch[0], NREADS, SK, w->counts[which], w->dist[which]);
    case 2:
        memset(w->alnA[which], 0, (size_t)NREADS * w->stride);
        memset(w->alnB[which], 0, (size_t)NREADS * w->stride);
        return polyhip_sw_align_batch(g_sc, w->reads, w->roffs, NREADS, w->ref, NULL, sizeof w->ref, w->score[which], w->endA[which],
                                      w->endB[which], w->err[which], w->alnA[which], w->alnB[which], w->alen[which], w->stride);
    case 3:
        memset(w->nwA[which], 0, (size_t)NREADS * w->nwstride);
        memset(w->nwB[which], 0, (size_t)NREADS * w->nwstride);
        return polyhip_nw_align_batch(g_sc, w->reads, w->roffs, NREADS, w->reads, w->roffs, 0, w->nwscore[which], w->nwerr[which],
                                      w->nwA[which], w->nwB[which], w->nwlen[which], w->nwstride);
    case 4:
        return polyhip_santalucia_batch(w->reads, w->roffs, NREADS, 500e-9, 50e-3, 0.0, w->tm[which], w->dH[which], w->dS[which]);
    case 5:
        return polyhip_least_rotation_batch(w->seqs, w->offs, NREADS, w->rot[which], w->rotated[which]);
    default:
        return polyhip_seqhash_batch(w->seqs, w->offs, NREADS, 0, 1, 1, w->hash[which], w->herr[which]);
    }
}

/* the strings are the LAST len[p] bytes of each slot; what lies in front of them is not part of the result */
static int same_strings(const uint8_t *x, const uint8_t *y, const uint32_t *len, uint32_t stride)
{
    for (int p = 0; p < NREADS; ++p)
        if (len[p] > stride || memcmp(x + (size_t)(p + 1) * stride - len[p], y + (size_t)(p + 1) * stride - len[p], len[p]))
            return 0;
    return 1;
}

static int same(const Work *w, int op)
{
    switch (op) {
    case 0: return !memcmp(w->sketch[0], w->sketch[1], sizeof w->sketch[0]);
    case 1: return !memcmp(w->counts[0], w->counts[1], sizeof w->counts[0]) && !memcmp(w->dist[0], w->dist[1], sizeof w->dist[0]);
    case 2:
        return !memcmp(w->score[0], w->score[1], sizeof w->score[0]) && !memcmp(w->endA[0], w->endA[1], sizeof w->endA[0]) &&
               !memcmp(w->endB[0], w->endB[1], sizeof w->endB[0]) && !memcmp(w->err[0], w->err[1], sizeof w->err[0]) &&
               !memcmp(w->alen[0], w->alen[1], sizeof w->alen[0]) && same_strings(w->alnA[0], w->alnA[1], w->alen[0], w->stride) &&
               same_strings(w->alnB[0], w->alnB[1], w->alen[0], w->stride);
    case 3:
        return !memcmp(w->nwscore[0], w->nwscore[1], sizeof w->nwscore[0]) && !memcmp(w->nwlen[0], w->nwlen[1], sizeof w->nwlen[0]) &&
               !memcmp(w->nwerr[0], w->nwerr[1], sizeof w->nwerr[0]) && same_strings(w->nwA[0], w->nwA[1], w->nwlen[0], w->nwstride) &&
               same_strings(w->nwB[0], w->nwB[1], w->nwlen[0], w->nwstride);
    case 4:
        return !memcmp(w->tm[0], w->tm[1], sizeof w->tm[0]) && !memcmp(w->dH[0], w->dH[1], sizeof w->dH[0]) &&
               !memcmp(w->dS[0], w->dS[1], sizeof w->dS[0]);
    case 5: return !memcmp(w->rot[0], w->rot[1], sizeof w->rot[0]) && !memcmp(w->rotated[0], w->rotated[1], (size_t)NREADS * RLEN);
    default: return !memcmp(w->hash[0], w->hash[1], sizeof w->hash[0]) && !memcmp(w->herr[0], w->herr[1], sizeof w->herr[0]);
    }
}

static void *thread_main(void *arg)
{
    const long t = (long)arg;
    Work *w = &g_w[t];
    long bad = 0;
    if (polyhip_set_device(0) != POLYHIP_OK)
        return (void *)1L;
    for (int r = 0; r < g_rounds; ++r)
        for (int k = 0; k < NOPS; ++k) {
            const int op = (int)((k + t) % NOPS); /* neighbouring threads are in different entry points */
            const int rc = run_op(w, op, 1);
            if (rc != POLYHIP_OK) {
                fprintf(stderr, "thread %ld op %d: status %d (%s)\n", t, op, rc, polyhip_last_error());
                ++bad;
            } else if (!same(w, op)) {
                fprintf(stderr, "thread %ld op %d round %d: result differs from the serial run\n", t, op, r);
                ++bad;
            }
        }
    return (void *)bad;
}

int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 8;
    if (argc > 2)
        g_rounds = atoi(argv[2]);
    if (nt < 1 || nt > MAXT)
        return 2;
    /* a device list from the environment (POLYHIP_DEVICES=0,0: two fan-out workers sharing the GPU) applies to the
     * THREADS' calls only: the serial answers below are one-device answers */
    int dev_ids[64];
    const int ndev_list = polyhip_get_devices(dev_ids, 64);
    if (ndev_list < 0 || polyhip_set_devices(NULL, 0) != POLYHIP_OK) {
        fprintf(stderr, "device list: %s\n", polyhip_last_error());
        return 1;
    }
    /* NUC_4-like scoring through the public flatten contract: +5 / -4, gap -2, alphabet ACGT */
    static int32_t lut[65536];
    uint8_t va[256] = {0}, vb[256] = {0};
    const char *al = "ACGT";
    for (int i = 0; i < 4; ++i) {
        va[(uint8_t)al[i]] = vb[(uint8_t)al[i]] = 1;
        for (int j = 0; j < 4; ++j)
            lut[(uint8_t)al[i] * 256 + (uint8_t)al[j]] = i == j ? 5 : -4;
    }
    if (polyhip_scoring_create(lut, va, vb, -2, &g_sc) != POLYHIP_OK) {
        fprintf(stderr, "scoring_create: %s\n", polyhip_last_error());
        return 1;
    }
    for (int t = 0; t < nt; ++t) {
        Work *w = &g_w[t];
        uint64_t st = 0x7777ull + (uint64_t)t * 1315423911ull;
        w->seqs = (uint8_t *)malloc((size_t)NREADS * RLEN);
        for (size_t i = 0; i < (size_t)NREADS * RLEN; ++i)
            w->seqs[i] = (uint8_t)al[splitmix(&st) & 3];
        for (int i = 0; i <= NREADS; ++i)
            w->offs[i] = (uint64_t)i * RLEN;
        for (size_t i = 0; i < sizeof w->ref; ++i)
            w->ref[i] = (uint8_t)al[splitmix(&st) & 3];
        uint64_t o = 0;
        for (int i = 0; i < NREADS; ++i) { /* ragged reads: windows of the reference with substitutions */
            const int len = 60 + (int)(splitmix(&st) % 60), a = (int)(splitmix(&st) % (sizeof w->ref - 120));
            w->roffs[i] = o;
            for (int j = 0; j < len; ++j)
                w->reads[o + j] = (splitmix(&st) % 100) < 6 ? (uint8_t)al[splitmix(&st) & 3] : w->ref[a + j];
            o += (uint64_t)len;
        }
        w->roffs[NREADS] = o;
        w->stride = polyhip_sw_traceback_stride(g_sc, 120, sizeof w->ref);
        w->nwstride = 240;
        for (int q = 0; q < 2; ++q) {
            w->alnA[q] = (uint8_t *)calloc((size_t)NREADS * w->stride + 1, 1);
            w->alnB[q] = (uint8_t *)calloc((size_t)NREADS * w->stride + 1, 1);
            w->nwA[q] = (uint8_t *)calloc((size_t)NREADS * w->nwstride + 1, 1);
            w->nwB[q] = (uint8_t *)calloc((size_t)NREADS * w->nwstride + 1, 1);
            w->rotated[q] = (uint8_t *)calloc((size_t)NREADS * RLEN + 1, 1);
        }
        for (int op = 0; op < NOPS; ++op) /* the serial answers */
            if (run_op(w, op, 0) != POLYHIP_OK) {
                fprintf(stderr, "serial op %d: %s\n", op, polyhip_last_error());
                return 1;
            }
    }
    if (polyhip_set_devices(dev_ids, ndev_list > 64 ? 64 : ndev_list) != POLYHIP_OK) {
        fprintf(stderr, "polyhip_set_devices: %s\n", polyhip_last_error());
        return 1;
    }
    printf("device list: %d\n", ndev_list);
    pthread_t th[MAXT];
    for (long t = 0; t < nt; ++t)
        pthread_create(&th[t], NULL, thread_main, (void *)t);
    long bad = 0;
    for (int t = 0; t < nt; ++t) {
        void *r = NULL;
        pthread_join(th[t], &r);
        bad += (long)r;
    }
    polyhip_scoring_destroy(g_sc);
    if (bad) {
        printf("abi_threads FAILED: %ld mismatches\n", bad);
        return 1;
    }
    printf("abi_threads ok: %d threads x %d rounds x %d entry points, every result equals the serial run\n", nt, g_rounds, NOPS);
    return 0;
}
