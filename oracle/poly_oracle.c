/*
 * poly_oracle.c -- CPU restatement of bebop/poly's search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see poly_oracle.h).  Plain C, single thread,
 * deliberately shaped like the Go code it restates so it can be read side
 * by side with the reference.  Citations are relative to /root/reference.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; Go on amd64 never
 * fuses multiply-add, so neither may this file).
 */
#include "poly_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* synthetic inputs -- our own generator, not the reference's
 * (random.DNASequence uses Go's math/rand, random/random.go:52-63, which
 * cannot be reproduced outside Go).                                        */
/* ====================================================================== */

uint64_t orc_splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_dna(uint64_t seed, uint8_t *out, size_t n)
{
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    uint64_t st = seed, x = 0;
    for (size_t i = 0; i < n; i++) {
        if ((i & 31) == 0)
            x = orc_splitmix64(&st);
        out[i] = (uint8_t)acgt[(x >> (2 * (i & 31))) & 3];
    }
}

/* ====================================================================== */
/* search/mash                                                             */
/* ====================================================================== */

static inline uint32_t rotl32(uint32_t x, int r)
{
    return (x << r) | (x >> (32 - r));
}

/* github.com/spaolacci/murmur3 v1.1.0 (go.mod:13), murmur32.go Sum32 ->
 * MurmurHash3_x86_32: little-endian 4-byte blocks, 1..3 byte tail, fmix32.
 * Call site: mash.go:76 (seed 0). */
uint32_t orc_murmur3_32(const uint8_t *data, size_t len, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h = seed;
    size_t nblocks = len / 4;
    for (size_t i = 0; i < nblocks; i++) {
        uint32_t k = (uint32_t)data[4 * i] | ((uint32_t)data[4 * i + 1] << 8) |
                     ((uint32_t)data[4 * i + 2] << 16) |
                     ((uint32_t)data[4 * i + 3] << 24);
        k *= c1;
        k = rotl32(k, 15);
        k *= c2;
        h ^= k;
        h = rotl32(h, 13);
        h = h * 5 + 0xe6546b64u;
    }
    const uint8_t *tail = data + nblocks * 4;
    uint32_t k1 = 0;
    switch (len & 3) {
    case 3:
        k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2:
        k1 ^= (uint32_t)tail[1] << 8; /* fallthrough */
    case 1:
        k1 ^= tail[0];
        k1 *= c1;
        k1 = rotl32(k1, 15);
        k1 *= c2;
        h ^= k1;
    }
    h ^= (uint32_t)len;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* (*Mash).Sketch, mash.go:68-104 -- same control flow, same quirks:
 *  - loop bound len-k, NOT len-k+1 (mash.go:73): the last k-mer is skipped;
 *  - positional fill while kmerStart < s-1 (mash.go:81-84), so a sequence
 *    with fewer than s windows leaves an unsorted prefix and an untouched
 *    tail;
 *  - duplicates are kept; equal-to-max hashes are not inserted (strict '>'
 *    at mash.go:96), which leaves the same multiset;
 *  - s == 0 / s == 1 index Sketches[-1] (mash.go:96 / :98) -> Go panics. */
int orc_mash_sketch(const uint8_t *seq, size_t n, int k, int s,
                    uint32_t *sketches, int faithful)
{
    long maxShifted = (long)s - 1;
    long nwin = (long)n - (long)k; /* may be negative: loop does not run */
    for (long start = 0; start < nwin; start++) {
        uint32_t hash = orc_murmur3_32(seq + start, (size_t)k, 0);
        if (start < maxShifted) {
            sketches[start] = hash;
            continue;
        }
        if (start == maxShifted) {
            sketches[maxShifted] = hash;
            qsort(sketches, (size_t)s, sizeof(uint32_t), cmp_u32);
            continue;
        }
        /* start > maxShifted */
        if (maxShifted < 0)
            return -1; /* s == 0: Sketches[-1], mash.go:96 */
        if (sketches[maxShifted] > hash) {
            sketches[maxShifted] = hash;
            if (maxShifted - 1 < 0)
                return -1; /* s == 1: Sketches[-1], mash.go:98 */
            if (hash < sketches[maxShifted - 1]) {
                if (faithful) {
                    qsort(sketches, (size_t)s, sizeof(uint32_t), cmp_u32);
                } else {
                    /* slice is sorted except for its last element: one
                     * insertion equals the full sort */
                    long j = maxShifted;
                    while (j > 0 && sketches[j - 1] > hash) {
                        sketches[j] = sketches[j - 1];
                        j--;
                    }
                    sketches[j] = hash;
                }
            }
        }
    }
    return 0;
}

/* a loop of orc_mash_sketch over a packed batch: one C call per shard of reads, so that bench.py's
 * all-cores CPU baseline runs one thread per shard without the interpreter between reads */
int orc_mash_sketch_batch(const uint8_t *seqs, const uint64_t *offsets, size_t n, int k, int s,
                          uint32_t *out, int faithful)
{
    for (size_t i = 0; i < n; i++)
        if (orc_mash_sketch(seqs + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), k, s, out + i * (size_t)s,
                            faithful) != 0)
            return -1;
    return 0;
}

/* mash.go:107-135.  Receiver a is "larger" unless a.SketchSize <
 * b.SketchSize (mash.go:109-115). */
static int mash_shared_core(const uint32_t *a, int sa, const uint32_t *b,
                            int sb, int *smaller_size)
{
    const uint32_t *larger = a, *smaller = b;
    int nl = sa, ns = sb;
    if (sa < sb) {
        larger = b;
        nl = sb;
        smaller = a;
        ns = sa;
    }
    *smaller_size = ns;
    /* mash.go:117 range-disjoint early-out (reads [size-1] and [0] whatever
     * state the slices are in) */
    if (larger[nl - 1] < smaller[0] || smaller[ns - 1] < larger[0])
        return 0;
    int same = 0, si = 0, li = 0;
    while (si < ns && li < nl) {
        if (smaller[si] == larger[li]) {
            same++;
            si++;
            li++;
        } else if (smaller[si] < larger[li]) {
            si++;
        } else {
            li++;
        }
    }
    return same;
}

int orc_mash_shared(const uint32_t *a, int sa, const uint32_t *b, int sb)
{
    int ns;
    return mash_shared_core(a, sa, b, sb, &ns);
}

double orc_mash_similarity(const uint32_t *a, int sa, const uint32_t *b, int sb)
{
    int ns;
    int same = mash_shared_core(a, sa, b, sb, &ns);
    return (double)same / (double)ns; /* mash.go:134 */
}

double orc_mash_distance(const uint32_t *a, int sa, const uint32_t *b, int sb)
{
    return 1 - orc_mash_similarity(a, sa, b, sb); /* mash.go:139 */
}

/* ====================================================================== */
/* search/align (+ matrix, alphabet)                                       */
/* ====================================================================== */

/* Alphabet.Encode over single-byte string keys (alphabet.go:25-41): the map
 * is filled in index order so a repeated symbol keeps its LAST index. */
static int alpha_encode(const char *sym, int n, uint8_t c)
{
    int idx = -1;
    for (int i = 0; i < n; i++)
        if ((uint8_t)sym[i] == c)
            idx = i;
    return idx;
}

int orc_submat_score(const orc_submat *m, uint8_t a, uint8_t b, int *out)
{
    /* Scoring.Score does string(byte) (align.go:90): a byte >= 0x80 becomes
     * a two-byte UTF-8 string and can never equal a one-byte symbol. */
    int ia = a < 0x80 ? alpha_encode(m->symA, m->na, a) : -1;
    if (ia < 0)
        return 1; /* first alphabet checked first, matrix.go:29-32 */
    int ib = b < 0x80 ? alpha_encode(m->symB, m->nb, b) : -1;
    if (ib < 0)
        return 2;
    *out = m->scores[ia * m->nb + ib];
    return 0;
}

void orc_submat_flatten(const orc_submat *m, int32_t *lut, uint8_t *validA,
                        uint8_t *validB)
{
    for (int a = 0; a < 256; a++)
        validA[a] = a < 0x80 && alpha_encode(m->symA, m->na, (uint8_t)a) >= 0;
    for (int b = 0; b < 256; b++)
        validB[b] = b < 0x80 && alpha_encode(m->symB, m->nb, (uint8_t)b) >= 0;
    for (int a = 0; a < 256; a++)
        for (int b = 0; b < 256; b++) {
            int v = 0;
            if (orc_submat_score(m, (uint8_t)a, (uint8_t)b, &v) != 0)
                v = 0;
            lut[a * 256 + b] = v;
        }
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* prepend one byte to a right-aligned buffer (alignA = string(x) + alignA,
 * align.go:216-226) */
typedef struct {
    char *buf;
    size_t cap, pos;
} revbuf;
static void rb_init(revbuf *r, char *buf, size_t cap)
{
    r->buf = buf;
    r->cap = cap;
    r->pos = cap;
}
static void rb_push(revbuf *r, char c) { r->buf[--r->pos] = c; }
static void rb_finish(revbuf *r)
{
    size_t len = r->cap - r->pos;
    memmove(r->buf, r->buf + r->pos, len);
    r->buf[len] = 0;
}

/* SmithWaterman, align.go:171-232. */
int orc_smith_waterman(const uint8_t *a, size_t m, const uint8_t *b, size_t n,
                       const orc_submat *mat, int gap, int64_t *score,
                       char *alignA, char *alignB, uint32_t *endA,
                       uint32_t *endB, uint8_t *err_sym)
{
    *score = 0;
    alignA[0] = alignB[0] = 0;
    if (endA)
        *endA = 0;
    if (endB)
        *endB = 0;
    size_t W = n + 1;
    int64_t *H = (int64_t *)calloc((m + 1) * W, sizeof(int64_t)); /* :175-178 */
    int64_t maxScore = 0;
    size_t maxRow = 0, maxCol = 0;
    for (size_t i = 1; i <= m; i++) {         /* columnM, align.go:186 */
        for (size_t j = 1; j <= n; j++) {     /* rowN,    align.go:187 */
            int s;
            int e = orc_submat_score(mat, a[i - 1], b[j - 1], &s);
            if (e) { /* align.go:189-191 */
                *err_sym = e == 1 ? a[i - 1] : b[j - 1];
                free(H);
                return e;
            }
            int64_t diag = H[(i - 1) * W + (j - 1)] + s;
            int64_t up = H[(i - 1) * W + j] + gap;
            int64_t left = H[i * W + (j - 1)] + gap;
            int64_t h = max64(0, max64(diag, max64(up, left)));
            H[i * W + j] = h;
            if (h > maxScore) { /* strict: first max in row-major order */
                maxScore = h;
                maxRow = i;
                maxCol = j;
            }
        }
    }
    revbuf ra, rb;
    rb_init(&ra, alignA, m + n);
    rb_init(&rb, alignB, m + n);
    size_t i = maxRow, j = maxCol;
    while (H[i * W + j] > 0) { /* align.go:210 */
        int s;
        int e = orc_submat_score(mat, a[i - 1], b[j - 1], &s);
        if (e) {
            *err_sym = e == 1 ? a[i - 1] : b[j - 1];
            free(H);
            alignA[0] = alignB[0] = 0;
            return e;
        }
        int64_t h = H[i * W + j];
        if (h == H[(i - 1) * W + (j - 1)] + s) { /* diag first, :215 */
            rb_push(&ra, (char)a[i - 1]);
            rb_push(&rb, (char)b[j - 1]);
            i--;
            j--;
        } else if (h == H[(i - 1) * W + j] + gap) { /* then up, :220 */
            rb_push(&ra, (char)a[i - 1]);
            rb_push(&rb, '-');
            i--;
        } else if (h == H[i * W + (j - 1)] + gap) { /* then left, :224 */
            rb_push(&ra, '-');
            rb_push(&rb, (char)b[j - 1]);
            j--;
        } else {
            break; /* unreachable: h>0 always equals one of the three */
        }
    }
    rb_finish(&ra);
    rb_finish(&rb);
    *score = maxScore;
    if (endA)
        *endA = (uint32_t)maxRow;
    if (endB)
        *endB = (uint32_t)maxCol;
    free(H);
    return 0;
}

/* NeedlemanWunsch, align.go:100-166.  Quirk kept: the traceback stops as
 * soon as EITHER index reaches 0 (align.go:141), so leading residues of the
 * longer remainder are dropped. */
int orc_needleman_wunsch(const uint8_t *a, size_t m, const uint8_t *b,
                         size_t n, const orc_submat *mat, int gap,
                         int64_t *score, char *alignA, char *alignB,
                         uint8_t *err_sym)
{
    *score = 0;
    alignA[0] = alignB[0] = 0;
    size_t W = n + 1;
    int64_t *H = (int64_t *)calloc((m + 1) * W, sizeof(int64_t));
    for (size_t i = 1; i <= m; i++)
        H[i * W] = H[(i - 1) * W] + gap; /* :113-115 */
    for (size_t j = 1; j <= n; j++)
        H[j] = H[j - 1] + gap; /* :118-120 */
    for (size_t i = 1; i <= m; i++)
        for (size_t j = 1; j <= n; j++) {
            int s;
            int e = orc_submat_score(mat, a[i - 1], b[j - 1], &s);
            if (e) {
                *err_sym = e == 1 ? a[i - 1] : b[j - 1];
                free(H);
                return e;
            }
            H[i * W + j] =
                max64(H[(i - 1) * W + (j - 1)] + s,
                      max64(H[(i - 1) * W + j] + gap, H[i * W + (j - 1)] + gap));
        }
    revbuf ra, rb;
    rb_init(&ra, alignA, m + n);
    rb_init(&rb, alignB, m + n);
    size_t i = m, j = n;
    while (i > 0 && j > 0) { /* :141 */
        int s;
        int e = orc_submat_score(mat, a[i - 1], b[j - 1], &s);
        if (e) {
            *err_sym = e == 1 ? a[i - 1] : b[j - 1];
            free(H);
            alignA[0] = alignB[0] = 0;
            return e;
        }
        int64_t h = H[i * W + j];
        if (h == H[(i - 1) * W + (j - 1)] + s) {
            rb_push(&ra, (char)a[i - 1]);
            rb_push(&rb, (char)b[j - 1]);
            i--;
            j--;
        } else if (h == H[(i - 1) * W + j] + gap) {
            rb_push(&ra, (char)a[i - 1]);
            rb_push(&rb, '-');
            i--;
        } else { /* unconditional else, :155-159 */
            rb_push(&ra, '-');
            rb_push(&rb, (char)b[j - 1]);
            j--;
        }
    }
    rb_finish(&ra);
    rb_finish(&rb);
    *score = H[m * W + n];
    free(H);
    return 0;
}

/* ====================================================================== */
/* transform                                                               */
/* ====================================================================== */

/* complementTable, transform.go:78-109: unmapped bytes -> 0x00 */
static uint8_t comp_table[256];
static int comp_ready;
static void comp_init(void)
{
    static const char from[] = "ABCDGHKMNRSTVWY";
    static const char to[] = "TVGHCDMKNYSABWR";
    memset(comp_table, 0, sizeof comp_table);
    for (int i = 0; from[i]; i++) {
        comp_table[(uint8_t)from[i]] = (uint8_t)to[i];
        comp_table[(uint8_t)(from[i] + 32)] = (uint8_t)(to[i] + 32);
    }
    comp_ready = 1;
}

void orc_reverse_complement(const uint8_t *seq, size_t n, uint8_t *out)
{
    if (!comp_ready)
        comp_init();
    for (size_t i = 0; i < n; i++)
        out[i] = comp_table[seq[n - i - 1]]; /* transform.go:19 */
}

/* strings.ToUpper on an ASCII string; bytes >= 0x80 are outside what the
 * build supports (Go would treat them as UTF-8) and are passed through. */
static inline uint8_t ascii_upper(uint8_t c)
{
    return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
}

/* ====================================================================== */
/* primers                                                                 */
/* ====================================================================== */

/* Go src/math/log.go (pure Go on amd64): FreeBSD e_log.c. */
double orc_go_log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01,
                 Ln2Lo = 1.90821492927058770002e-10,
                 L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                 L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (isnan(x) || (isinf(x) && x > 0))
        return x;
    if (x < 0)
        return NAN;
    if (x == 0)
        return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 0.70710678118654752440 /* Sqrt2/2 */) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

/* nearestNeighborsThermodynamics, primers.go:42-59.  A dinucleotide that is
 * not one of the 16 keys reads the map's zero value {0,0} (primers.go:98). */
static void nn_lookup(uint8_t x, uint8_t y, double *H, double *S)
{
    static const struct {
        char k[3];
        double H, S;
    } tab[16] = {
        {"AA", -7.6, -21.3}, {"TT", -7.6, -21.3}, {"AT", -7.2, -20.4},
        {"TA", -7.2, -21.3}, {"CA", -8.5, -22.7}, {"TG", -8.5, -22.7},
        {"GT", -8.4, -22.4}, {"AC", -8.4, -22.4}, {"CT", -7.8, -21.0},
        {"AG", -7.8, -21.0}, {"GA", -8.2, -22.2}, {"TC", -8.2, -22.2},
        {"CG", -10.6, -27.2}, {"GC", -9.8, -24.4}, {"GG", -8.0, -19.9},
        {"CC", -8.0, -19.9},
    };
    *H = 0;
    *S = 0;
    for (int i = 0; i < 16; i++)
        if ((uint8_t)tab[i].k[0] == x && (uint8_t)tab[i].k[1] == y) {
            *H = tab[i].H;
            *S = tab[i].S;
            return;
        }
}

/* SantaLucia, primers.go:70-105 -- additions in the reference's order. */
void orc_santalucia(const uint8_t *seq_in, size_t n, double primer_conc,
                    double salt_conc, double mg_conc, double *tm, double *dHo,
                    double *dSo)
{
    uint8_t *seq = (uint8_t *)calloc(n ? n : 1, 1);
    uint8_t *rc = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++)
        seq[i] = ascii_upper(seq_in[i]); /* :71 */
    const double gasConstant = 1.9872;
    double symmetryFactor;
    double dH = 0, dS = 0;
    dH += 0.2;  /* initialThermodynamicPenalty, :61,:78 */
    dS += -5.7; /* :79 */
    orc_reverse_complement(seq, n, rc);
    if (memcmp(seq, rc, n) == 0) { /* :81 */
        dH += 0;                   /* symmetryThermodynamicPenalty, :62 */
        dS += -1.4;
        symmetryFactor = 1;
    } else {
        symmetryFactor = 4;
    }
    if (seq[n - 1] == 'A' || seq[n - 1] == 'T') { /* :89, 3' end only */
        dH += 2.2;                                /* :63 */
        dS += 6.9;
    }
    double saltEffect = salt_conc + (mg_conc * 140);                /* :94 */
    dS += (0.368 * (double)((long)n - 1) * orc_go_log(saltEffect)); /* :95 */
    for (size_t i = 0; i + 1 < n; i++) {                            /* :97 */
        double h, s;
        nn_lookup(seq[i], seq[i + 1], &h, &s);
        dH += h;
        dS += s;
    }
    *tm = dH * 1000 / (dS + gasConstant * orc_go_log(primer_conc / symmetryFactor)) -
          273.15; /* :103 */
    *dHo = dH;
    *dSo = dS;
    free(seq);
    free(rc);
}

/* MarmurDoty, primers.go:108-118 */
double orc_marmur_doty(const uint8_t *seq, size_t n)
{
    double a = 0, t = 0, c = 0, g = 0;
    for (size_t i = 0; i < n; i++) {
        switch (ascii_upper(seq[i])) {
        case 'A': a += 1; break;
        case 'T': t += 1; break;
        case 'C': c += 1; break;
        case 'G': g += 1; break;
        }
    }
    return 2 * (a + t) + 4 * (c + g) - 7.0;
}

/* MeltingTemp, primers.go:121-128 */
double orc_melting_temp(const uint8_t *seq, size_t n)
{
    double tm, dH, dS;
    orc_santalucia(seq, n, 500e-9, 50e-3, 0.0, &tm, &dH, &dS);
    return tm;
}

/* What a caller of the reference does for BASELINE configs[4]: primers.SantaLucia on every substring
 * genome[i:i+L], L = Lmin..Lmax (primers.go:70-105 per call).  Output layout as polyhip_santalucia_scan:
 * plane (L - Lmin) holds n - L + 1 values; planes are nwin0 = n - Lmin + 1 apart.  Bench/test helper. */
void orc_santalucia_scan(const uint8_t *genome, size_t n, int Lmin, int Lmax, double primer_conc,
                         double salt_conc, double mg_conc, double *tm, double *dH, double *dS)
{
    if (Lmin < 1 || (size_t)Lmin > n)
        return;
    const size_t stride = n - (size_t)Lmin + 1;
    for (int L = Lmin; L <= Lmax && (size_t)L <= n; L++)
        for (size_t i = 0; i + (size_t)L <= n; i++) {
            const size_t o = (size_t)(L - Lmin) * stride + i;
            orc_santalucia(genome + i, (size_t)L, primer_conc, salt_conc, mg_conc, &tm[o], &dH[o], &dS[o]);
        }
}

/* (*Mash).Distance (mash.go:107-140) for every ordered pair of two sets of sorted sketches of size s. */
void orc_mash_distance_matrix(const uint32_t *X, size_t nx, const uint32_t *Y, size_t ny, int s, double *out)
{
    for (size_t i = 0; i < nx; i++)
        for (size_t j = 0; j < ny; j++)
            out[i * ny + j] = orc_mash_distance(X + i * (size_t)s, s, Y + j * (size_t)s, s);
}

/* ====================================================================== */
/* seqhash                                                                 */
/* ====================================================================== */

/* boothLeastRotation, seqhash.go:78-124 -- same variable roles. */
size_t orc_booth_least_rotation(const uint8_t *seq, size_t n)
{
    if (n == 0)
        return 0;
    size_t n2 = 2 * n;
    long *failure = (long *)malloc(n2 * sizeof(long));
    for (size_t i = 0; i < n2; i++)
        failure[i] = -1;
#define S2(i) (seq[(i) < n ? (i) : (i)-n]) /* sequence += sequence, :83 */
    long least = 0;
    for (long ci = 1; ci < (long)n2; ci++) {
        uint8_t ch = S2((size_t)ci);
        long f = failure[ci - least - 1];
        while (f != -1 && ch != S2((size_t)(least + f + 1))) {
            if (ch < S2((size_t)(least + f + 1)))
                least = ci - f - 1;
            f = failure[f];
        }
        if (ch != S2((size_t)(least + f + 1))) {
            if (ch < S2((size_t)least))
                least = ci;
            failure[ci - least] = -1;
        } else {
            failure[ci - least] = f + 1;
        }
    }
#undef S2
    free(failure);
    return (size_t)least;
}

/* RotateSequence, seqhash.go:127-138 */
void orc_rotate_sequence(const uint8_t *seq, size_t n, uint8_t *out)
{
    size_t r = orc_booth_least_rotation(seq, n);
    for (size_t i = 0; i < n; i++) {
        size_t j = r + i;
        out[i] = seq[j < n ? j : j - n];
    }
}

/* ---- BLAKE3 (lukechampine.com/blake3 v1.1.5 Sum256; go.mod:15) --------
 * Restated from the published BLAKE3 specification (hash mode, 32-byte
 * output).  Pinned by the seven digests in seqhash/seqhash_test.go:36-65 and
 * seqhash/example_test.go:19 (all single-block inputs) plus the spec's
 * empty-input digest; the multi-chunk tree is exercised by tests only for
 * self-consistency (no external vector is available offline). */
static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u,
                                  0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu,
                                  0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13,
                                    1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

static inline uint32_t rotr32(uint32_t x, int r)
{
    return (x >> r) | (x << (32 - r));
}
static inline void b3_g(uint32_t *v, int a, int b, int c, int d, uint32_t mx,
                        uint32_t my)
{
    v[a] = v[a] + v[b] + mx;
    v[d] = rotr32(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + my;
    v[d] = rotr32(v[d] ^ v[a], 8);
    v[c] = v[c] + v[d];
    v[b] = rotr32(v[b] ^ v[c], 7);
}
static void b3_compress(const uint32_t cv[8], const uint32_t block[16],
                        uint64_t counter, uint32_t block_len, uint32_t flags,
                        uint32_t out[16])
{
    uint32_t v[16], m[16], t[16];
    memcpy(v, cv, 32);
    memcpy(v + 8, B3_IV, 16);
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    memcpy(m, block, 64);
    for (int r = 0; r < 7; r++) {
        b3_g(v, 0, 4, 8, 12, m[0], m[1]);
        b3_g(v, 1, 5, 9, 13, m[2], m[3]);
        b3_g(v, 2, 6, 10, 14, m[4], m[5]);
        b3_g(v, 3, 7, 11, 15, m[6], m[7]);
        b3_g(v, 0, 5, 10, 15, m[8], m[9]);
        b3_g(v, 1, 6, 11, 12, m[10], m[11]);
        b3_g(v, 2, 7, 8, 13, m[12], m[13]);
        b3_g(v, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++)
            t[i] = m[B3_PERM[i]];
        memcpy(m, t, 64);
    }
    for (int i = 0; i < 8; i++) {
        out[i] = v[i] ^ v[i + 8];
        out[i + 8] = v[i + 8] ^ cv[i];
    }
}
static void b3_load_block(const uint8_t *p, size_t len, uint32_t w[16])
{
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
               ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}
/* chaining value of one chunk (<= 1024 bytes); if root, emits the final
 * 16-word output of the last block with ROOT set. */
static void b3_chunk(const uint8_t *p, size_t len, uint64_t chunk_idx, int root,
                     uint32_t out16[16])
{
    uint32_t cv[8], w[16], o[16];
    memcpy(cv, B3_IV, 32);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t bi = 0; bi < nblocks; bi++) {
        size_t off = bi * 64;
        size_t bl = len - off < 64 ? len - off : 64;
        uint32_t flags = 0;
        if (bi == 0)
            flags |= B3_CHUNK_START;
        if (bi == nblocks - 1) {
            flags |= B3_CHUNK_END;
            if (root)
                flags |= B3_ROOT;
        }
        b3_load_block(p + off, bl, w);
        b3_compress(cv, w, chunk_idx, (uint32_t)bl, flags, o);
        memcpy(cv, o, 32);
    }
    memcpy(out16, o, 64);
}
/* cv of the subtree covering [p, p+len) starting at chunk index c0 */
static void b3_subtree(const uint8_t *p, size_t len, uint64_t c0, int root,
                       uint32_t out16[16])
{
    if (len <= 1024) {
        b3_chunk(p, len, c0, root, out16);
        return;
    }
    /* left = largest power-of-two number of chunks strictly less than total */
    size_t nchunks = (len + 1023) / 1024;
    size_t left = 1;
    while (left * 2 < nchunks)
        left *= 2;
    uint32_t l[16], r[16], block[16];
    b3_subtree(p, left * 1024, c0, 0, l);
    b3_subtree(p + left * 1024, len - left * 1024, c0 + left, 0, r);
    memcpy(block, l, 32);
    memcpy(block + 8, r, 32);
    b3_compress(B3_IV, block, 0, 64, B3_PARENT | (root ? B3_ROOT : 0), out16);
}
void orc_blake3_256(const uint8_t *data, size_t n, uint8_t out[32])
{
    uint32_t o[16];
    b3_subtree(data, n, 0, 1, o);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)o[i];
        out[4 * i + 1] = (uint8_t)(o[i] >> 8);
        out[4 * i + 2] = (uint8_t)(o[i] >> 16);
        out[4 * i + 3] = (uint8_t)(o[i] >> 24);
    }
}

static int lex_less(const uint8_t *a, const uint8_t *b, size_t n)
{
    return memcmp(a, b, n) < 0; /* sort.Strings on equal-length strings */
}

/* Hash, seqhash.go:141-224 */
int orc_seqhash(const uint8_t *seq_in, size_t n, const char *type, int circular,
                int double_stranded, char *out, uint8_t *err_char)
{
    int is_dna = strcmp(type, "DNA") == 0, is_rna = strcmp(type, "RNA") == 0,
        is_prot = strcmp(type, "PROTEIN") == 0;
    uint8_t *seq = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        uint8_t c = ascii_upper(seq_in[i]); /* :143 */
        if (is_rna && c == 'U')
            c = 'T'; /* :146-148 */
        seq[i] = c;
    }
    out[0] = 0;
    if (!is_dna && !is_rna && !is_prot) { /* :151-153 */
        free(seq);
        return 1;
    }
    if (is_dna || is_rna) { /* :154-160 */
        for (size_t i = 0; i < n; i++)
            if (!strchr("ATUGCYRSWKMBDHVNZ", seq[i]) || seq[i] == 0) {
                *err_char = seq[i];
                free(seq);
                return 2;
            }
    }
    if (is_prot) { /* :161-171 */
        for (size_t i = 0; i < n; i++)
            if (!strchr("ACDEFGHIKLMNPQRSTVWYUO*BXZ", seq[i]) || seq[i] == 0) {
                *err_char = seq[i];
                free(seq);
                return 3;
            }
    }
    if (is_prot && double_stranded) { /* :174-176 */
        free(seq);
        return 4;
    }
    uint8_t *cand0 = (uint8_t *)malloc(n ? n : 1);
    uint8_t *cand1 = (uint8_t *)malloc(n ? n : 1);
    uint8_t *rc = (uint8_t *)malloc(n ? n : 1);
    const uint8_t *det;
    if (circular && double_stranded) { /* :181-184 */
        orc_rotate_sequence(seq, n, cand0);
        orc_reverse_complement(seq, n, rc);
        orc_rotate_sequence(rc, n, cand1);
        det = lex_less(cand1, cand0, n) ? cand1 : cand0;
    } else if (circular) { /* :185-186 */
        orc_rotate_sequence(seq, n, cand0);
        det = cand0;
    } else if (double_stranded) { /* :187-190 */
        orc_reverse_complement(seq, n, rc);
        det = lex_less(rc, seq, n) ? rc : seq;
    } else {
        det = seq; /* :191-192 */
    }
    uint8_t dig[32];
    orc_blake3_256(det, n, dig);
    static const char hexd[] = "0123456789abcdef";
    char *o = out;
    *o++ = 'v';
    *o++ = '1';
    *o++ = '_';
    *o++ = is_dna ? 'D' : is_rna ? 'R' : 'P';
    *o++ = circular ? 'C' : 'L';
    *o++ = double_stranded ? 'D' : 'S';
    *o++ = '_';
    for (int i = 0; i < 32; i++) {
        *o++ = hexd[dig[i] >> 4];
        *o++ = hexd[dig[i] & 15];
    }
    *o = 0;
    free(seq);
    free(cand0);
    free(cand1);
    free(rc);
    return 0;
}
