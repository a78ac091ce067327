/*
 * abi_smoke.c -- a torch-free, Python-free consumer of libpolyhip's C ABI: what the cgo shim does.
 *
 * Links only against libpolyhip.so (which brings /opt/rocm's libamdhip64, not the copy bundled with a
 * PyTorch wheel), calls the host-pointer entry points on the reference's own test inputs and compares
 * with the values those tests (and the pinned oracle) hold:
 *   search/mash/mash_test.go:10-61        sketches of the two 62-mers, distances 0 / 1 / 0.19999999999999996
 *   search/align/align_test.go:138-194    SmithWaterman 13 GTT-AC / GTTGAC and 17 A-CACACTA / AGCACAC-A
 *   search/align/example_test.go:46       NeedlemanWunsch
 *   primers/primers_test.go:24-81         MarmurDoty 31, SantaLucia 62.3169.. / 47.4285.., MeltingTemp 52.6338..
 *   seqhash/seqhash_test.go:36-91         RotateSequence, one protein seqhash, pUC19's least rotation (argv[1])
 * Built by __graft_entry__.build() (gcc) and run by tests/test_abi_gpu.py in a subprocess whose
 * environment has no torch on any library path.  Exit status 0 = all checks passed.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "polyhip.h"

static int failures = 0;

#define CHECK(cond, ...)                                                        \
    do {                                                                        \
        if (!(cond)) {                                                          \
            ++failures;                                                         \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);                \
            fprintf(stderr, __VA_ARGS__);                                       \
            fprintf(stderr, "\n");                                              \
        }                                                                       \
    } while (0)

#define OK(call)                                                                \
    do {                                                                        \
        int rc_ = (call);                                                       \
        CHECK(rc_ == POLYHIP_OK, "%s -> %d (%s)", #call, rc_, polyhip_last_error()); \
    } while (0)

static void sketch(const char *seq, uint32_t k, uint32_t s, uint32_t *out)
{
    uint64_t off[2] = {0, strlen(seq)};
    OK(polyhip_mash_sketch_batch((const uint8_t *)seq, off, 1, k, s, out));
}

static double distance(const uint32_t *a, uint32_t sa, const uint32_t *b, uint32_t sb)
{
    double d = -1;
    OK(polyhip_mash_distance_matrix(a, 1, sa, b, 1, sb, NULL, &d));
    return d;
}

static void test_mash(void)
{
    const char *s1 = "ATGCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGA";
    const char *s2 = "ATCGATCGATCGATCGATCGATCGATCGATCGATCGAATGCGATCGATCGATCGATCGATCG";
    uint32_t f1[10] = {0}, f2[10] = {0}, sp[10] = {0};
    sketch(s1, 17, 10, f1);
    for (int i = 0; i < 10; ++i)
        CHECK(f1[i] == 0x096698deu, "sketch[%d] = %08x", i, f1[i]);
    sketch(s1, 17, 9, f2);
    CHECK(distance(f1, 10, f2, 9) == 0.0, "distance(10, 9)");   /* mash_test.go:17 */
    CHECK(distance(f2, 9, f1, 10) == 0.0, "distance(9, 10)");   /* :22 */
    CHECK(distance(f1, 10, sp, 10) == 1.0, "spoofed 10");        /* :30 */
    CHECK(distance(f1, 10, sp, 9) == 1.0, "spoofed 9");          /* :37 */
    uint32_t g[5] = {0};
    sketch(s2, 17, 5, g);
    CHECK(g[0] == 0x08f7dc27u && g[1] == 0x096698deu && g[4] == 0x096698deu, "second 62-mer: %08x %08x", g[0], g[1]);
    CHECK(distance(f1, 10, g, 5) == 0.19999999999999996, "distance 0.2: %.17g", distance(f1, 10, g, 5)); /* :48 */
    uint32_t h1[10] = {0}, h2[5] = {0};
    sketch(s2, 17, 10, h1);
    sketch(s1, 17, 5, h2);
    CHECK(distance(h1, 10, h2, 5) == 0.0, "last TestMash distance"); /* :59 */
    /* s == 1: the reference indexes Sketches[-1] (mash.go:98) iff a later window hashes below the first one.  A
     * homopolymer never does (every window hashes alike: murmur3("A" x 17) = 0x295dfd60, this repo's restatement) and
     * leaves that hash in Sketches[0]; both 62-mers above have a smaller hash further on and panic */
    uint64_t off[2] = {0, 62}, offa[2] = {0, 40};
    uint32_t one = 0;
    CHECK(polyhip_mash_sketch_batch((const uint8_t *)"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", offa, 1, 17, 1, &one) == POLYHIP_OK &&
              one == 0x295dfd60u,
          "s = 1 on a homopolymer: %08x (%s)", one, polyhip_last_error());
    CHECK(polyhip_mash_sketch_batch((const uint8_t *)s1, off, 1, 17, 1, &one) == POLYHIP_ERR_PANIC, "s = 1 must be a panic here");
    CHECK(polyhip_mash_sketch_batch((const uint8_t *)s2, off, 1, 17, 1, &one) == POLYHIP_ERR_PANIC, "s = 1 must be a panic here");
    CHECK(polyhip_mash_sketch_batch((const uint8_t *)s2, off, 1, 17, 0, &one) == POLYHIP_ERR_PANIC, "s = 0 with a window must be a panic");
}

static polyhip_scoring *nuc_scoring(int match, int mismatch, int gap)
{
    static int32_t lut[65536];
    uint8_t va[256] = {0}, vb[256] = {0};
    const char *alpha = "-ACGT";
    memset(lut, 0, sizeof lut);
    for (const char *a = alpha; *a; ++a) {
        va[(unsigned char)*a] = vb[(unsigned char)*a] = 1;
        for (const char *b = alpha; *b; ++b)
            lut[(unsigned char)*a * 256 + (unsigned char)*b] = (*a == '-' || *b == '-') ? 0 : (*a == *b ? match : mismatch);
    }
    polyhip_scoring *sc = NULL;
    OK(polyhip_scoring_create(lut, va, vb, gap, &sc));
    return sc;
}

static void sw(polyhip_scoring *sc, const char *a, const char *b, long want, const char *wa, const char *wb)
{
    uint64_t offA[2] = {0, strlen(a)};
    int64_t score = -1;
    uint32_t ea, eb, err, len;
    uint8_t alnA[64], alnB[64];
    OK(polyhip_sw_align_batch(sc, (const uint8_t *)a, offA, 1, (const uint8_t *)b, NULL, strlen(b), &score, &ea, &eb, &err,
                              alnA, alnB, &len, 64));
    CHECK(score == want && err == 0, "SW(%s, %s) score %ld err %u", a, b, (long)score, err);
    CHECK(len == strlen(wa) && !memcmp(alnA + 64 - len, wa, len) && !memcmp(alnB + 64 - len, wb, len),
          "SW(%s, %s) strings %.*s / %.*s", a, b, (int)len, alnA + 64 - len, (int)len, alnB + 64 - len);
    /* the packed-strings flavour the Go binding calls: same answer, strings at alnOff[0] .. alnOff[1] */
    uint64_t off[2] = {7, 7};
    uint8_t pa[64], pb[64];
    score = -1;
    OK(polyhip_sw_align_batch_packed(sc, (const uint8_t *)a, offA, 1, (const uint8_t *)b, NULL, strlen(b), &score, &ea, &eb, &err, pa, pb,
                                     off, 64));
    CHECK(score == want && off[0] == 0 && off[1] == strlen(wa) && !memcmp(pa, wa, off[1]) && !memcmp(pb, wb, off[1]),
          "SW packed(%s, %s): score %ld, %llu bytes", a, b, (long)score, (unsigned long long)off[1]);
    if (strlen(wa) > 2) { /* a buffer that is too small is an error that says what is needed */
        CHECK(polyhip_sw_align_batch_packed(sc, (const uint8_t *)a, offA, 1, (const uint8_t *)b, NULL, strlen(b), &score, &ea, &eb, &err, pa,
                                            pb, off, 2) == POLYHIP_ERR_INVALID && off[1] == strlen(wa) && score == want,
              "SW packed with a 2-byte buffer: off[1] = %llu", (unsigned long long)off[1]);
    }
}

static void test_align(void)
{
    polyhip_scoring *sc = nuc_scoring(3, -3, -2); /* align_test.go:139-151 */
    sw(sc, "TGTTACGG", "GGTTGACTA", 13, "GTT-AC", "GTTGAC");
    sw(sc, "ACACACTA", "AGCACACA", 17, "A-CACACTA", "AGCACAC-A");
    sw(sc, "", "GAT", 0, "", ""); /* :199-215: empty input */
    /* unknown symbol: "Symbol X not in alphabet" names a[0] first (align.go:189-191) */
    {
        uint64_t offA[2] = {0, 4};
        int64_t score;
        uint32_t ea, eb, err = 0;
        OK(polyhip_sw_batch(sc, (const uint8_t *)"XCGT", offA, 1, (const uint8_t *)"ACGT", NULL, 4, &score, &ea, &eb, &err));
        CHECK(err == ((1u << 8) | 'X') && score == 0, "bad symbol: err %x score %ld", err, (long)score);
    }
    /* NeedlemanWunsch, align/example_test.go:12-46 (its U written as T: this table has no U) */
    polyhip_scoring *d = nuc_scoring(1, -1, -1);
    {
        const char *a = "GATTACA";
        uint64_t offA[2] = {0, 7};
        int64_t score = 99;
        uint32_t err, len;
        uint8_t alnA[32], alnB[32];
        OK(polyhip_nw_align_batch(d, (const uint8_t *)a, offA, 1, (const uint8_t *)"GCATGCT", NULL, 7, &score, &err, alnA, alnB,
                                  &len, 32));
        CHECK(score == 0 && err == 0, "NW GATTACA/GCATGCT score %ld", (long)score); /* example_test.go:46: score 0 */
        CHECK(len == 8 && !memcmp(alnA + 24, "G-ATTACA", 8) && !memcmp(alnB + 24, "GCA-TGCT", 8), "NW strings %.*s / %.*s",
              (int)len, alnA + 32 - len, (int)len, alnB + 32 - len);
    }
    OK(polyhip_scoring_destroy(d));
    OK(polyhip_scoring_destroy(sc));
}

static void test_primers(void)
{
    double tm, dH, dS;
    uint64_t off[2] = {0, 18};
    OK(polyhip_santalucia_batch((const uint8_t *)"ACGATGGCAGTAGCATGC", off, 1, 0.1e-6, 350e-3, 0.0, &tm, &dH, &dS));
    CHECK(tm == 62.31695672635385 && dH == -144.0 && dS == -394.46768721086363, "SantaLucia %.17g %.17g %.17g", tm, dH, dS);
    off[1] = 14;
    OK(polyhip_santalucia_batch((const uint8_t *)"ACGTAGATCTACGT", off, 1, 0.1e-6, 350e-3, 0.0, &tm, &dH, &dS));
    CHECK(tm == 47.42851359405711, "palindrome SantaLucia %.17g", tm);
    off[1] = 17;
    OK(polyhip_santalucia_batch((const uint8_t *)"GTAAAACGACGGCCAGT", off, 1, 500e-9, 50e-3, 0.0, &tm, &dH, &dS));
    CHECK(tm == 52.63382276100299, "MeltingTemp %.17g", tm);
    off[1] = 12;
    OK(polyhip_marmurdoty_batch((const uint8_t *)"ACGTCCGGACTT", off, 1, &tm));
    CHECK(tm == 31.0, "MarmurDoty %.17g", tm);
    /* scan: the (L = 17, start 0) cell of a scan over M13 fwd + tail equals the single call */
    const char *g = "GTAAAACGACGGCCAGTACGT";
    const uint64_t n = strlen(g), ns = n - 16 + 1;
    double *pl = malloc(3 * 3 * ns * sizeof(double));
    OK(polyhip_santalucia_scan((const uint8_t *)g, n, 16, 18, 500e-9, 50e-3, 0.0, pl, pl + 3 * ns, pl + 6 * ns));
    CHECK(pl[1 * ns + 0] == 52.63382276100299, "scan cell %.17g", pl[1 * ns]);
    CHECK(isnan(pl[2 * ns + (ns - 1)]), "window past the end must be NaN");
    free(pl);
}

static void test_seqhash(const char *puc19_path)
{
    uint64_t off[2] = {0, 9}, rot = 99;
    uint8_t out[16] = {0};
    OK(polyhip_least_rotation_batch((const uint8_t *)"TTAGCCCAT", off, 1, &rot, out));
    CHECK(rot == 2 && !memcmp(out, "AGCCCATTT", 9), "RotateSequence -> %llu %.9s", (unsigned long long)rot, out);
    char hash[72] = {0};
    uint32_t err = 1;
    off[1] = 4;
    OK(polyhip_seqhash_batch((const uint8_t *)"MGC*", off, 1, 2, 0, 0, hash, &err));
    CHECK(err == 0 && !strcmp(hash, "v1_PLS_922ec11f5227ce77a42f07f565a7a1a479772b5cf3f1f6e93afc5ecbc0fd5955"), "seqhash %s", hash);
    if (puc19_path) {
        FILE *f = fopen(puc19_path, "rb");
        CHECK(f != NULL, "cannot open %s", puc19_path);
        if (!f)
            return;
        static uint8_t seq[8192], rotated[8192];
        size_t n = fread(seq, 1, sizeof seq, f);
        fclose(f);
        while (n && (seq[n - 1] == '\n' || seq[n - 1] == '\r'))
            --n;
        uint64_t o2[2] = {0, n};
        OK(polyhip_least_rotation_batch(seq, o2, 1, &rot, rotated));
        CHECK(n == 2686 && rot == 2356 && !memcmp(rotated, "aaaaaaaccaccgctaccagcggtggtttg", 30),
              "pUC19: n %zu rot %llu", n, (unsigned long long)rot);
        /* every rotation of the plasmid has the same least rotation (seqhash_test.go:68-91), 64 of them in one call */
        static uint8_t many[64 * 2686], rots[64 * 2686];
        uint64_t offs[65], idx[64];
        for (int r = 0; r < 64; ++r) {
            const size_t sh = (size_t)r * 41 % n;
            memcpy(many + r * n, seq + sh, n - sh);
            memcpy(many + r * n + (n - sh), seq, sh);
            offs[r] = r * n;
        }
        offs[64] = 64 * n;
        OK(polyhip_least_rotation_batch(many, offs, 64, idx, rots));
        for (int r = 0; r < 64; ++r)
            CHECK(!memcmp(rots + r * n, rotated, n), "rotation %d of pUC19 differs", r);
    }
}

/* which HIP runtime did the dynamic linker give us?  (the point of this harness: not a PyTorch wheel's copy) */
static void print_hip_runtime(void)
{
    FILE *f = fopen("/proc/self/maps", "r");
    char line[1024];
    while (f && fgets(line, sizeof line, f)) {
        char *p = strstr(line, "libamdhip64");
        if (p) {
            char *path = strchr(line, '/');
            if (path) {
                path[strcspn(path, "\n")] = 0;
                printf("abi_smoke: HIP runtime %s\n", path);
                break;
            }
        }
    }
    if (f)
        fclose(f);
}

int main(int argc, char **argv)
{
    print_hip_runtime();
    CHECK(polyhip_abi_version() == POLYHIP_ABI_VERSION, "ABI version %d", polyhip_abi_version());
    const int ndev = polyhip_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "abi_smoke: no HIP device (%d: %s) -- the library has no CPU fallback\n", ndev, polyhip_last_error());
        return 77;
    }
    OK(polyhip_set_device(0));
    char arch[64] = "";
    OK(polyhip_device_arch(arch, sizeof arch));
    test_mash();
    test_align();
    test_primers();
    test_seqhash(argc > 1 ? argv[1] : NULL);
    printf("abi_smoke: %s on %s (%d device(s)), %d failure(s)\n", failures ? "FAILED" : "ok", arch, ndev, failures);
    return failures ? 1 : 0;
}
