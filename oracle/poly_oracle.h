/*
 * poly_oracle.h -- CPU restatement of bebop/poly's search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (poly_amd/, include/polyhip.h) never links, imports or calls it.
 *
 * Parity status: the Go reference cannot be executed in the authoring
 * container (no Go toolchain), so every function here is a restatement that
 * is pinned against the reference's own test expectations
 * (tests/test_oracle_golden.py lists them one by one).  Raw murmur3 values
 * are pinned against the canonical MurmurHash3_x86_32 vectors because the
 * reference's tests pin them only indirectly (see DESIGN.md "Oracle").
 *
 * All citations are relative to /root/reference.
 */
#ifndef POLY_ORACLE_H
#define POLY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs (not from the reference) ---------- */
uint64_t orc_splitmix64(uint64_t *state);
/* base i = "ACGT"[(x >> (2*(i&31))) & 3], x = (i/32)-th splitmix64 output of
 * a stream whose state starts at `seed`.  Position-addressable so a GPU
 * generator can reproduce it: see orc_synth_dna_at(). */
void orc_synth_dna(uint64_t seed, uint8_t *out, size_t n);

/* ---- search/mash ------------------------------------------------------ */
/* spaolacci/murmur3 v1.1.0 Sum32 == MurmurHash3_x86_32(seed 0); mash.go:76 */
uint32_t orc_murmur3_32(const uint8_t *data, size_t len, uint32_t seed);

/* (*Mash).Sketch, mash.go:68-104.  `sketches` is the caller's Mash.Sketches
 * (length s, prior state preserved where the reference preserves it).
 * faithful != 0 re-sorts the whole slice on every accepted hash exactly like
 * mash.go:90,99; faithful == 0 does the equivalent single insertion.
 * Returns 0, or -1 where the Go code would panic (index out of range). */
int orc_mash_sketch(const uint8_t *seq, size_t n, int k, int s,
                    uint32_t *sketches, int faithful);
/* the same for every sequence of a packed batch (out: n x s, in/out like Sketches) */
int orc_mash_sketch_batch(const uint8_t *seqs, const uint64_t *offsets, size_t n, int k, int s,
                          uint32_t *out, int faithful);

/* (*Mash).Similarity / Distance, mash.go:107-140 (receiver = a). */
double orc_mash_similarity(const uint32_t *a, int sa, const uint32_t *b, int sb);
double orc_mash_distance(const uint32_t *a, int sa, const uint32_t *b, int sb);
/* the integer the similarity is made of (sameHashes, mash.go:121-132),
 * after the early-out of mash.go:117 (which yields 0). */
int orc_mash_shared(const uint32_t *a, int sa, const uint32_t *b, int sb);
void orc_mash_distance_matrix(const uint32_t *X, size_t nx, const uint32_t *Y, size_t ny, int s, double *out);

/* ---- search/align + matrix + alphabet --------------------------------- */
/* A substitution matrix as NewSubstitutionMatrix builds it (matrix.go:20):
 * two alphabets of single-byte symbols and a row-major na x nb score table.
 * Duplicate symbols: last index wins, as in alphabet.go:27-30. */
typedef struct {
    int na, nb;
    const char *symA; /* na bytes */
    const char *symB; /* nb bytes */
    const int *scores; /* na*nb */
} orc_submat;

/* SubstitutionMatrix.Score (matrix.go:28-38): returns 0 and *out, or
 * 1 = first symbol not in FirstAlphabet, 2 = second not in SecondAlphabet. */
int orc_submat_score(const orc_submat *m, uint8_t a, uint8_t b, int *out);

/* SmithWaterman, align.go:171-232.  alignA/alignB must hold m+n+1 bytes.
 * Returns 0, or (1|2) as orc_submat_score with *err_sym = offending byte,
 * in which case score 0 / empty strings are returned like align.go:189-191.
 * endA/endB = (maxScoreRow, maxScoreCol), 1-based, 0 when score is 0. */
int orc_smith_waterman(const uint8_t *a, size_t m, const uint8_t *b, size_t n,
                       const orc_submat *mat, int gap, int64_t *score,
                       char *alignA, char *alignB, uint32_t *endA,
                       uint32_t *endB, uint8_t *err_sym);

/* NeedlemanWunsch, align.go:100-166 (same conventions). */
int orc_needleman_wunsch(const uint8_t *a, size_t m, const uint8_t *b,
                         size_t n, const orc_submat *mat, int gap,
                         int64_t *score, char *alignA, char *alignB,
                         uint8_t *err_sym);

/* Flatten Score() over all byte pairs: lut[a*256+b], valid masks.  This is
 * what the Go wrapper does through the public Score(). */
void orc_submat_flatten(const orc_submat *m, int32_t *lut256x256,
                        uint8_t *validA256, uint8_t *validB256);

/* ---- transform -------------------------------------------------------- */
/* transform.ReverseComplement, transform.go:15-23,78-109 */
void orc_reverse_complement(const uint8_t *seq, size_t n, uint8_t *out);

/* ---- primers ---------------------------------------------------------- */
/* Go's pure-Go math.Log (src/math/log.go, FreeBSD e_log.c); amd64 has no
 * assembly stub for Log, so this is what primers.go:95,103 evaluate. */
double orc_go_log(double x);
/* SantaLucia, primers.go:70-105.  n must be >= 1 (Go panics on ""). */
void orc_santalucia(const uint8_t *seq, size_t n, double primer_conc,
                    double salt_conc, double mg_conc, double *tm, double *dH,
                    double *dS);
/* MarmurDoty, primers.go:108-118; MeltingTemp, primers.go:121-128 */
void orc_santalucia_scan(const uint8_t *genome, size_t n, int Lmin, int Lmax, double primer_conc,
                         double salt_conc, double mg_conc, double *tm, double *dH, double *dS);
double orc_marmur_doty(const uint8_t *seq, size_t n);
double orc_melting_temp(const uint8_t *seq, size_t n);

/* ---- seqhash ---------------------------------------------------------- */
/* boothLeastRotation, seqhash.go:78-124 (index) */
size_t orc_booth_least_rotation(const uint8_t *seq, size_t n);
/* RotateSequence, seqhash.go:127-138 */
void orc_rotate_sequence(const uint8_t *seq, size_t n, uint8_t *out);
/* BLAKE3-256 (lukechampine.com/blake3 v1.1.5 Sum256; seqhash.go:221) */
void orc_blake3_256(const uint8_t *data, size_t n, uint8_t out[32]);
/* Hash, seqhash.go:141-224.  type: "DNA" | "RNA" | "PROTEIN" (anything else
 * errors).  out must hold 72 bytes.  Returns 0, or 1..4 = the error at
 * seqhash.go:152 / :157 / :169 / :175 with *err_char = offending letter. */
int orc_seqhash(const uint8_t *seq, size_t n, const char *type, int circular,
                int double_stranded, char *out, uint8_t *err_char);

#ifdef __cplusplus
}
#endif
#endif
