/*
 * polyhip.h -- C ABI of libpolyhip.so: the MI355X (gfx950) implementation of
 * bebop/poly's search hot path.
 *
 * The reference (pure Go, no FFI of its own) exposes this path as the
 * exported API of four packages; a drop-in keeps those Go signatures and
 * binds the entry points below through cgo (stubs: INTEGRATION.md, go/).
 * Each entry point names the reference function it replaces; citations are
 * relative to the reference checkout.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.
 *  - Every function returns POLYHIP_OK (0) or a negative polyhip_status; the
 *    message of the last failure on the calling thread is
 *    polyhip_last_error().
 *  - Two flavours per operation:
 *      NAME      host pointers (what cgo hands over).  Synchronous: stages
 *                through device memory on the current device and returns
 *                when the outputs are written.
 *      NAME_dev  device pointers + a hipStream_t (passed as void*; NULL = the
 *                null stream).  Asynchronous: enqueues on the stream and
 *                returns; inputs/outputs stay resident in HBM.
 *  - Batches are packed: one contiguous byte buffer + (n+1) uint64 offsets,
 *    sequence i = bytes [offsets[i], offsets[i+1]).  The library never keeps
 *    a caller pointer after returning (cgo pointer rule).
 *  - Thread safe: no unsynchronised globals; the current HIP device of the
 *    calling thread is used (polyhip_set_device is a thin hipSetDevice).
 *  - There is NO CPU fallback: without a usable HIP device every compute
 *    entry point fails with POLYHIP_ERR_HIP.
 *  - Sequence bytes must be ASCII (< 0x80) wherever the reference would
 *    case-fold or map them through string(byte) (Go treats bytes >= 0x80 as
 *    UTF-8 there); such input is rejected with POLYHIP_ERR_INVALID rather
 *    than silently diverging.
 */
#ifndef POLYHIP_H
#define POLYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POLYHIP_ABI_VERSION 1

typedef enum {
    POLYHIP_OK = 0,
    POLYHIP_ERR_INVALID = -1,     /* bad argument (null pointer, unsorted offsets, ...) */
    POLYHIP_ERR_HIP = -2,         /* HIP runtime / device failure (message has the hipError) */
    POLYHIP_ERR_UNSUPPORTED = -3, /* outside the implemented range (documented per call) */
    POLYHIP_ERR_PANIC = -4,       /* the reference would panic on these arguments */
    POLYHIP_ERR_SYMBOL = -5       /* align: "Symbol X not in alphabet" (see polyhip_sw_*) */
} polyhip_status;

typedef void *polyhip_stream_t; /* hipStream_t */

/* ---- runtime --------------------------------------------------------- */
int polyhip_abi_version(void);
const char *polyhip_last_error(void); /* thread-local, never NULL */
int polyhip_device_count(void);       /* >= 0, or a negative status */
int polyhip_set_device(int device);
/* name of the current device's gcnArch (e.g. "gfx950:sramecc+:xnack-") */
int polyhip_device_arch(char *buf, size_t buflen);

/* ---- one host call over several GPUs  (polyhip_init(n_devices); 8e) ---- */
/*
 * The reference is single-threaded Go (search/mash/mash.go:68-140, search/align/align.go:171-232,
 * primers/primers.go:70-128, seqhash/seqhash.go:127-224 are plain loops); a Go host that keeps that API has ONE call
 * per batch, so the node's GPUs have to be reached from inside that call.  The library keeps one device list per
 * process.  Empty (the default): every host-pointer entry point runs on the calling thread's current device.  With n
 * entries the host-pointer entry points
 *     polyhip_mash_sketch_batch, polyhip_mash_distance_matrix, polyhip_mash_sketch_distance_matrix,
 *     polyhip_sw_batch, polyhip_sw_align_batch, polyhip_sw_align_batch_packed, polyhip_nw_align_batch,
 *     polyhip_santalucia_scan, polyhip_santalucia_scan_first, polyhip_santalucia_batch, polyhip_marmurdoty_batch,
 *     polyhip_least_rotation_batch, polyhip_seqhash_batch
 * cut their batch into n contiguous shards balanced by bytes (reads, pairs, window starts, matrix rows: SURVEY 8e's
 * partitioning; no data-path collective) and run shard q on a worker thread that lives on device ids[q], each with
 * its own streams and two-slot upload / compute / download pipeline, results written straight into the caller's
 * buffers.  Results, error codes and messages are those of the one-device call (the status reported is the lowest
 * failing shard's, i.e. the first failure in batch order; positions in messages are positions of the whole batch).
 * An id may appear more than once ("0,0,0"): the shards then share that GPU (testing on a one-GPU box).  The _dev
 * entry points, the feeders and polyhip_scoring_create are not affected (a scoring handle is copied to the other
 * devices of the list on first use).  polyhip_sw_last_path and friends then describe the first non-empty shard's kernels.
 *   polyhip_set_devices(ids, n)  n = 0 clears the list.  Calls in flight finish on the list they started with.
 *   polyhip_get_devices          -> the list's length (ids filled up to `capacity`).
 *   polyhip_init(n)              = polyhip_set_devices({0 .. n-1}); n <= 0: every visible device.
 *   polyhip_shutdown()           = polyhip_set_devices(NULL, 0).
 *   POLYHIP_DEVICES=0,1,2 | all  in the environment: the list a process starts with (read once, at the first
 *                                host-pointer call, unless polyhip_set_devices came first).
 */
int polyhip_set_devices(const int *ids, int n);
int polyhip_get_devices(int *ids, int capacity);
int polyhip_init(int n_devices);
int polyhip_shutdown(void);

/* ---- synthetic inputs (bench/test plumbing) ------------- */
/* d_out[i] = "ACGT"[(x >> 2*(i&31)) & 3], x = splitmix64 output number
 * (first + i)/32 + 1 of the stream seeded with `seed`; `first` must be a
 * multiple of 32 (lets ranks generate disjoint slices of one stream). */
int polyhip_synth_dna_dev(uint64_t seed, uint64_t first, uint8_t *d_out,
                          uint64_t n, polyhip_stream_t stream);

/* ---- K1: search/mash (*Mash).Sketch  (search/mash/mash.go:68-104) ------ */
/*
 * For every sequence i: hash each of the (len_i - k) windows (the reference
 * skips the last k-mer, mash.go:73) with MurmurHash3_x86_32 seed 0
 * (murmur3.Sum32, mash.go:76) over the raw bytes, and leave in
 * out[i*s .. i*s+s) exactly what Sketch leaves in Mash.Sketches:
 *   len_i - k >= s : the s smallest hashes, ascending, duplicates kept;
 *   0 < len_i - k < s : out[i*s + j] = hash of window j for j < len_i - k,
 *                       remaining entries NOT written (caller's prior state
 *                       survives, as in the reference);
 *   len_i - k <= 0 : nothing written.
 * So `out` is in/out: pass the current Sketches (zeros after mash.New).
 * Range: any k; s <= 2^24.  SketchSize up to 8192 with KmerSize up to 4096 run the LDS-resident kernels (k = 17 / 21 / 31
 * specialised); beyond that (mash.New(21, 10000) is ordinary usage) a kernel that hashes every window straight from
 * global memory and keeps its candidates in a stream-ordered scratch allocation (hipMallocAsync / hipFreeAsync on
 * `stream`) -- same results, about an order of magnitude slower per k-mer.
 * s < 2 is the reference's behaviour READ BY READ (mash.go:96,98 index Sketches[-1]): with s == 0 a sequence panics iff it
 * has a window (len > k); with s == 1 window 0 fills Sketches[0] and the sequence panics iff a LATER window hashes below
 * it.  If any sequence of the batch would panic the call returns POLYHIP_ERR_PANIC naming the first one (rows of the
 * sequences that do not panic are written as the reference leaves them, the others are left alone); otherwise
 * POLYHIP_OK.  This one case synchronises `stream` (the verdict comes from the device).
 */
int polyhip_mash_sketch_batch(const uint8_t *seqs, const uint64_t *offsets,
                              uint64_t n, uint32_t k, uint32_t s,
                              uint32_t *out);
int polyhip_mash_sketch_batch_dev(const uint8_t *d_seqs,
                                  const uint64_t *d_offsets, uint64_t n,
                                  uint32_t k, uint32_t s, uint32_t *d_out,
                                  polyhip_stream_t stream);

/* ---- K2: search/mash (*Mash).Similarity / Distance  (search/mash/mash.go:107-140) */
/*
 * For two sets of sketches X (nx x sx, the receivers) and Y (ny x sy):
 *     d_counts[i * ld + j] = sameHashes of X_i.Similarity(Y_j)   (mash.go:108-132)
 * i.e. the reference's result before the division, INCLUDING its behaviour on
 * sketches that are not ascending (a sequence with fewer than SketchSize
 * windows leaves a positional, zero/stale-padded sketch, mash.go:81-84): those
 * pairs run the reference's own range early-out + merge loop.  All-vs-all on
 * one GPU: X == Y.  Sharded over ranks: Y = the all-gathered sketches, X = this
 * rank's row block of Y, d_counts = its row block of the matrix.
 * Similarity = counts / min(sx, sy); Distance = 1 - that (next call).
 * SketchSize 0 -> POLYHIP_ERR_PANIC (mash.go:117 indexes Sketches[-1]).
 * Range: sx, sy <= 65535; nx, ny < 2^31.  A Y set of 2^32 hashes or more is joined in column stripes of fewer than
 * that, one index after the other in the same workspace (polyhip_mash_shared_counts_dev only; the index_build / reuse
 * pair keeps ONE index and so ny*sy < 2^32).  POLYHIP_K2_MAX_ITEMS=<n> lowers the stripe bound (testing aid).
 * d_work: polyhip_mash_shared_counts_workspace_bytes(...) bytes of scratch.
 */
size_t polyhip_mash_shared_counts_workspace_bytes(uint64_t nx, uint32_t sx,
                                                  uint64_t ny, uint32_t sy);
int polyhip_mash_shared_counts_dev(const uint32_t *d_X, uint64_t nx,
                                   uint32_t sx, const uint32_t *d_Y,
                                   uint64_t ny, uint32_t sy,
                                   uint16_t *d_counts, uint64_t ld,
                                   void *d_work, size_t work_bytes,
                                   polyhip_stream_t stream);
/*
 * The same in two steps, for callers that put several X against one Y (row blocks of one matrix, queries against a
 * resident sketch database): polyhip_mash_index_build_dev builds Y's inverted index into d_work (a third of a
 * 12,500 x 100,000 block's time), polyhip_mash_shared_counts_reuse_dev joins an X against the index that an earlier
 * polyhip_mash_index_build_dev / polyhip_mash_shared_counts_dev call with the SAME d_Y, ny, sy left in the SAME
 * d_work (the Y side sits at the front of the workspace, wherever nx puts the rest).  Workspace:
 * polyhip_mash_shared_counts_workspace_bytes(largest nx, ...); index_build alone needs ..._workspace_bytes(0, ...).
 */
int polyhip_mash_index_build_dev(const uint32_t *d_Y, uint64_t ny, uint32_t sy,
                                 void *d_work, size_t work_bytes,
                                 polyhip_stream_t stream);
int polyhip_mash_shared_counts_reuse_dev(const uint32_t *d_X, uint64_t nx,
                                         uint32_t sx, const uint32_t *d_Y,
                                         uint64_t ny, uint32_t sy,
                                         uint16_t *d_counts, uint64_t ld,
                                         void *d_work, size_t work_bytes,
                                         polyhip_stream_t stream);
/*
 * The index built in PARTS (multi-rank all-vs-all: SURVEY 8e, BASELINE configs[2]).  Every rank holds the same
 * gathered Y; rank r calls polyhip_mash_index_build_part_dev(part = r, nparts = nranks): it runs the cheap whole-set
 * steps (ascending check, coarse histogram) and then sorts only ITS share of the value range -- coarse buckets chosen from
 * the histogram so that every part holds about the same number of items -- writing its items and bucket starts at their
 * FINAL offsets in d_work.  polyhip_mash_index_allgather_dev then exchanges the parts in place (two ragged RCCL
 * all-gathers, polyhip_allgatherv_dev) and finishes the header; after it polyhip_mash_shared_counts_reuse_dev works as
 * after polyhip_mash_index_build_dev.  Both calls synchronise `stream` once (the part bounds are read from the device's
 * histogram).  Building parts 0 .. nparts-1 one after the other into ONE workspace, then polyhip_mash_index_finalize_dev,
 * gives the same index as polyhip_mash_index_build_dev (same bucket starts; the same items in every bucket, in whatever
 * order the atomics put them) -- how the parts are tested on one GPU.  polyhip_mash_index_part_spans reports where the
 * parts sit: item_spans / start_spans get nparts + 1 byte offsets into d_work each (part p = [spans[p], spans[p+1])).
 */
struct polyhip_comm;
int polyhip_mash_index_build_part_dev(const uint32_t *d_Y, uint64_t ny, uint32_t sy,
                                      uint32_t part, uint32_t nparts,
                                      void *d_work, size_t work_bytes,
                                      polyhip_stream_t stream);
int polyhip_mash_index_part_spans(uint64_t ny, uint32_t sy, uint32_t nparts,
                                  const void *d_work, size_t work_bytes,
                                  uint64_t *item_spans, uint64_t *start_spans,
                                  polyhip_stream_t stream);
int polyhip_mash_index_finalize_dev(uint64_t ny, uint32_t sy, void *d_work,
                                    size_t work_bytes, polyhip_stream_t stream);
/* Bytes per item of the index in d_work (synchronous read-back; tests, profiling, sizing an exchange): 8 = (value, sketch
 * id | occurrence number); 4 = the compact form the build picks ON THE DEVICE when the join to come is the one-stripe dense
 * join (up to ~113k columns of 10-bit counters) and the value's bits below its bucket plus the largest multiplicity of a
 * hash inside one sketch fit 11 bits: the item then carries the LDS counter it bumps (dword and field), so the join's
 * inner step is subtract, compare, two shifts, and, ds_add.  An index built on its own assumes X sets of Y's SketchSize;
 * a join that does not fit that assumption rebuilds the index with 8-byte items first.  POLYHIP_K2_COMPACT=0 keeps the
 * 8-byte items (testing aid). */
int polyhip_mash_index_format_dev(const void *d_work, uint32_t *item_bytes);
/* How the index in d_work was built (synchronous read-back; tests and profiling).  info[0]: 0 = the two-level build on
 * 8-byte intermediate items, 1 = the sliced build on 4-byte intermediate items (the default where its conditions hold:
 * compact items, SketchSize <= 1024, <= 131,072 sketches, 2^16 <= largest hash < 2^30, 4 <= bucket shift <= 10; the device
 * decides the last three), 2 = the sliced build was planned and called off on the device (a sketch repeats a hash more
 * often than a compact item numbers, or more than 65,536 repeated hashes): the two-level build ran.  For the sliced
 * build info[1] = coarse buckets (hash >> 16), info[2] = parts of the value range, info[3] = coarse buckets per part,
 * info[4] = coarse buckets level 2 could not hold in registers (two passes), info[5] = repeated hashes it numbered.
 * POLYHIP_K2_B4=0 keeps the two-level build (testing aid; the parts API and the in-process item exchange always use it). */
int polyhip_mash_index_build_info_dev(const void *d_work, uint32_t info[6]);
int polyhip_mash_index_allgather_dev(struct polyhip_comm *c, uint64_t ny,
                                     uint32_t sy, void *d_work,
                                     size_t work_bytes, polyhip_stream_t stream);
/* What the last polyhip_mash_shared_counts_dev call on this workspace did
 * (synchronous read-back; tests and profiling): mode 0 = hash join, 1 = the
 * reference's merge for every pair; the number of non-ascending sketches on
 * each side; the rows the join handed to the merge (more related sketches
 * than its LDS table holds); the index's self-join size sum_b |Y_b|^2.
 * Any pointer may be NULL. */
int polyhip_mash_shared_counts_mode_dev(const void *d_work, uint32_t *mode,
                                        uint32_t *n_irregular_x,
                                        uint32_t *n_irregular_y,
                                        uint32_t *n_overflow_rows,
                                        uint64_t *join_estimate);
/* d_dist[i * ld_dist + j] = 1 - float64(counts[i][j]) / float64(min(sx, sy))
 * (mash.go:134,139; so 8 shared of 10 gives 0.19999999999999996). */
int polyhip_mash_distance_from_counts_dev(const uint16_t *d_counts,
                                          uint64_t nx, uint64_t ny,
                                          uint64_t ld_counts, uint32_t sx,
                                          uint32_t sy, double *d_dist,
                                          uint64_t ld_dist,
                                          polyhip_stream_t stream);
/* host flavour: counts (nx*ny u16) and/or dist (nx*ny f64) may be NULL. */
int polyhip_mash_distance_matrix(const uint32_t *X, uint64_t nx, uint32_t sx,
                                 const uint32_t *Y, uint64_t ny, uint32_t sy,
                                 uint16_t *counts, double *dist);
/*
 * BASELINE configs[2] in one host call: mash.New(k, s).Sketch(seq_i) for every sequence of a packed batch
 * (mash.go:59-104), then X_i.Similarity(X_j) / X_i.Distance(X_j) for every ordered pair (mash.go:107-140) -- the two
 * nested loops a caller of the reference writes.  The sketches stay in HBM between the two steps.
 *   sketches  n * s in/out like polyhip_mash_sketch_batch's `out` (prior Sketches in, new ones out), or NULL: zeros in
 *             (= mash.New), nothing out.
 *   counts    n * n sameHashes (u16), and/or  dist  n * n float64 Distance; either may be NULL (both NULL: sketch only).
 * On a device list (polyhip_set_devices) this is SURVEY 8e's flow inside one process: the reads shard by bytes, every
 * device sketches its shard, and the devices build ONE index of all n sketches together without gathering the sketches
 * (round 4): each runs the index's first level on its own rows, the 8-byte items travel by value range
 * (hipMemcpyPeerAsync -- no RCCL, no process per device), the second level runs on 1/N of the range per device, the
 * finished parts are exchanged; each device then joins the rows it sketched, which go straight into the caller's
 * matrix.  A set with an irregular sketch (a read with fewer than s windows: its pairs take the reference's merge,
 * which reads raw sketches), a matrix too wide for the dense join, or POLYHIP_K2_EXCHANGE=0 take the gather instead:
 * the devices pull each other's sketches and each builds the whole index.  polyhip_mash_sketch_distance_matrix_last_path
 * says which ran.  Range: s <= 65535, n < 2^31.  SketchSize < 2: the status of
 * polyhip_mash_sketch_batch (the reference panics in Sketch); SketchSize 0 with a matrix asked for: POLYHIP_ERR_PANIC
 * (mash.go:117).
 */
int polyhip_mash_sketch_distance_matrix(const uint8_t *seqs,
                                        const uint64_t *offsets, uint64_t n,
                                        uint32_t k, uint32_t s,
                                        uint32_t *sketches, uint16_t *counts,
                                        double *dist);
/* the calling thread's last polyhip_mash_sketch_distance_matrix: 0 = one device, 1 = a device list with the item exchange,
 * 2 = a device list with the sketch gather (tests) */
int polyhip_mash_sketch_distance_matrix_last_path(void);
/* What the calling thread's last polyhip_mash_sketch_distance_matrix did (round 5: so that the first run on more than one
 * physical GPU explains itself).  path as above; devices = entries of the device list (1 without one); the device-to-device
 * copies of the call -- rows of sketches on the gather path, index items and finished index parts on the exchange path --
 * counted by TRANSPORT: peer (hipDeviceCanAccessPeer said yes and peer access is on: xGMI), staged (it said no: the runtime
 * bounces the copy through host memory -- correct, and several times slower) and local (both ends on one device: a list
 * that names a device twice); the bytes they moved; wall milliseconds of the call's rounds as the calling thread saw them
 * (each round ends when its slowest device does): sketching, the index (exchange path: its five rounds; gather path: 0,
 * the index is built inside the join), and the join incl. the rows' way back to the host. */
typedef struct polyhip_matrix_info {
    int32_t path, devices;
    int32_t peer_copies, staged_copies, local_copies, reserved;
    uint64_t bytes_peer, bytes_staged, bytes_local;
    double ms_sketch, ms_index, ms_join;
} polyhip_matrix_info;
int polyhip_mash_sketch_distance_matrix_last_info(polyhip_matrix_info *info);

/* ---- K3: search/align SmithWaterman  (search/align/align.go:171-232) ---- */
/*
 * align.Scoring{SubstitutionMatrix, GapPenalty} (align.go:73-95) flattened
 * through the matrix's public Score() (its score table is unexported,
 * matrix.go:13-17):  lut[a*256 + b] = Score(string(byte a), string(byte b)),
 * validA[a] != 0 iff byte a is a symbol of FirstAlphabet, validB likewise
 * for SecondAlphabet (bytes >= 0x80 are never valid: string(byte) is a
 * two-byte UTF-8 string).  All three tables are HOST pointers (parameters,
 * not data) and are copied.  The handle owns small device tables on the HIP
 * device that is current at creation and must be used on that device.
 * Range: |gap| and |scores| such that  max|score| * (lenA + lenB) < 2^31.
 */
typedef struct polyhip_scoring polyhip_scoring;
int polyhip_scoring_create(const int32_t *lut256x256, const uint8_t *validA256,
                           const uint8_t *validB256, int64_t gap,
                           polyhip_scoring **out);
int polyhip_scoring_destroy(polyhip_scoring *sc);

/*
 * Score pass of SmithWaterman for a batch of pairs (A_p, B_p):
 *   H[i][j] = max(0, H[i-1][j-1] + S(a_i, b_j), H[i-1][j] + gap, H[i][j-1] + gap)
 * (align.go:192-195) with the reference's argmax: first maximum in row-major
 * order, i over A outer, j over B inner (strict '>' at align.go:197).
 * Outputs per pair p:
 *   score[p]        maxScore
 *   endA[p],endB[p] (maxScoreRow, maxScoreCol), 1-based; 0,0 when score == 0
 *   err[p]          0, or (which << 8) | symbol for the reference's
 *                   "Symbol X not in alphabet" error (align.go:189-191):
 *                   which = 1 (A / FirstAlphabet) or 2 (B / SecondAlphabet),
 *                   symbol = the byte the reference would name -- a[0] if
 *                   invalid, else the first invalid b[j], else the first
 *                   invalid a[i]; never set when either string is empty.
 *                   score/endA/endB are 0 for such pairs.
 * A is a packed batch (d_A, d_offA).  B is either ONE shared sequence
 * (d_offB == NULL, d_B[0..lenB)) or a packed batch (d_offB != NULL, lenB =
 * the maximum B length).  max_lenA >= every A length (the Go wrapper knows it
 * from packing; a longer A sets err[p] = 0xFFFFFFFF).
 * d_work: polyhip_sw_workspace_bytes(...) bytes of device scratch.
 */
size_t polyhip_sw_workspace_bytes(const polyhip_scoring *sc, uint64_t npairs,
                                  uint32_t max_lenA, uint64_t lenB,
                                  int shared_B);
int polyhip_sw_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A,
                         const uint64_t *d_offA, uint64_t npairs,
                         uint32_t max_lenA, const uint8_t *d_B,
                         const uint64_t *d_offB, uint64_t lenB,
                         int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB,
                         uint32_t *d_err, void *d_work, size_t work_bytes,
                         polyhip_stream_t stream);
/* Host-pointer flavour (cgo): same outputs in host memory. offB == NULL ->
 * shared B of length lenB. */
int polyhip_sw_batch(const polyhip_scoring *sc, const uint8_t *A,
                     const uint64_t *offA, uint64_t npairs, const uint8_t *B,
                     const uint64_t *offB, uint64_t lenB, int64_t *score,
                     uint32_t *endA, uint32_t *endB, uint32_t *err);
/*
 * Traceback of SmithWaterman (align.go:205-229) for the same batch, from the
 * score pass's outputs: walk back from (endA, endB) while H > 0 preferring
 * diagonal, then up (alignB gets '-'), then left (alignA gets '-').
 * Pair p's strings are the LAST d_alnLen[p] bytes of its aln_stride-byte slots:
 *     alignA_p = d_alnA[p*aln_stride + aln_stride - len .. p*aln_stride + aln_stride)
 * (the reference builds them by prepending).  Pairs with err != 0 or score 0
 * get length 0, as the reference returns "".
 * d_score (the score pass's output; may be NULL) lets each pair shrink its window:
 * a read that aligns well needs far fewer columns than the batch-wide bound.
 * aln_stride >= polyhip_sw_traceback_stride(sc, max_lenA, lenB).
 * d_work: any size >= 256 pairs' worth; polyhip_sw_traceback_workspace_bytes
 * returns enough for all pairs at once (capped at 8 GiB); smaller workspaces
 * make the call loop over chunks of pairs.  With several chunks the lane-per-pair
 * kernels run them through the two halves of the workspace on two streams -- `stream`
 * and one the library keeps per calling thread, joined to `stream` by events before
 * and after -- so that the end of a chunk overlaps the start of the next; everything
 * is ordered on `stream` as if it had run there (POLYHIP_TB_OVERLAP=0: it does).
 */
uint32_t polyhip_sw_traceback_stride(const polyhip_scoring *sc,
                                     uint32_t max_lenA, uint64_t lenB);
size_t polyhip_sw_traceback_workspace_bytes(const polyhip_scoring *sc,
                                            uint64_t npairs, uint32_t max_lenA,
                                            uint64_t lenB);
int polyhip_sw_traceback_dev(const polyhip_scoring *sc, const uint8_t *d_A,
                             const uint64_t *d_offA, uint64_t npairs,
                             uint32_t max_lenA, const uint8_t *d_B,
                             const uint64_t *d_offB, uint64_t lenB,
                             const uint32_t *d_endA, const uint32_t *d_endB,
                             const uint32_t *d_err, const int64_t *d_score,
                             uint8_t *d_alnA,
                             uint8_t *d_alnB, uint32_t *d_alnLen,
                             uint32_t aln_stride, void *d_work,
                             size_t work_bytes, polyhip_stream_t stream);
/*
 * The whole SmithWaterman on device pointers in one call: polyhip_sw_batch_dev + polyhip_sw_traceback_dev, same
 * outputs (d_work / d_tb_work sized as for those two).  For batches that take the packed score pass and the
 * byte-profile traceback (BASELINE config 4's shape: >= 48k reads of <= 152 symbols against one reference) the
 * score pass skips its locate step -- a second DP over the columns around each pair's maximum -- and the traceback
 * kernel, which sweeps those columns anyway, finds the row-major-first maximum in its last block: about 7 % less
 * time than the two calls.  The same holds for reads of 257..1024 symbols that take the packed multi-lane pass and the
 * one-wave-per-pair traceback on a byte profile of the pair (polyhip_sw_last_path 7 / polyhip_sw_traceback_last_path 7:
 * 80k reads of 1 kb, 92 -> 82 ms).  POLYHIP_SW_FUSE=0 in the environment keeps the two passes separate (testing aid).
 */
int polyhip_sw_align_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A,
                               const uint64_t *d_offA, uint64_t npairs,
                               uint32_t max_lenA, const uint8_t *d_B,
                               const uint64_t *d_offB, uint64_t lenB,
                               int64_t *d_score, uint32_t *d_endA,
                               uint32_t *d_endB, uint32_t *d_err,
                               uint8_t *d_alnA, uint8_t *d_alnB,
                               uint32_t *d_alnLen, uint32_t aln_stride,
                               void *d_work, size_t work_bytes,
                               void *d_tb_work, size_t tb_work_bytes,
                               polyhip_stream_t stream);
/* Host-pointer flavour of the whole SmithWaterman: score pass + traceback.  With one shared reference and more than
 * ~200 MB of string slots the pairs go through two device slots in chunks of 262,144 (at most eight chunks): the strings
 * of one chunk cross PCIe while the next chunk is aligned.  POLYHIP_SW_HOST_CHUNKS=1..8 sets the chunk count (testing
 * aid; 1 = single shot). */
int polyhip_sw_align_batch(const polyhip_scoring *sc, const uint8_t *A,
                           const uint64_t *offA, uint64_t npairs,
                           const uint8_t *B, const uint64_t *offB,
                           uint64_t lenB, int64_t *score, uint32_t *endA,
                           uint32_t *endB, uint32_t *err, uint8_t *alnA,
                           uint8_t *alnB, uint32_t *alnLen,
                           uint32_t aln_stride);
/* The same with PACKED strings -- what a cgo caller wants (Go strings are made from slices, not from fixed-stride slots):
 * alignA_p = alnA[alnOff[p] .. alnOff[p+1]), alignB_p = the same range of alnB (the two strings of a pair have one
 * length); alnOff has npairs + 1 entries.  A pair's strings are a few hundred of its slot's bytes (151 of 525 at BASELINE
 * config 4), so compacting them on the device cuts the PCIe traffic of 1M reads from 1.05 GB to 0.3 GB.  aln_capacity =
 * bytes each of alnA / alnB holds; if the strings need more, the call returns POLYHIP_ERR_INVALID after filling score,
 * endA, endB, err and alnOff (alnOff[npairs] = the bytes needed).  Chunked through two slots like the call above. */
int polyhip_sw_align_batch_packed(const polyhip_scoring *sc, const uint8_t *A,
                                  const uint64_t *offA, uint64_t npairs,
                                  const uint8_t *B, const uint64_t *offB,
                                  uint64_t lenB, int64_t *score, uint32_t *endA,
                                  uint32_t *endB, uint32_t *err, uint8_t *alnA,
                                  uint8_t *alnB, uint64_t *alnOff,
                                  uint64_t aln_capacity);
/* ---- search/align NeedlemanWunsch  (search/align/align.go:100-166) -------------- */
/*
 * Global alignment of every pair (A_p, B_p) (B shared when d_offB == NULL): score
 * = H[lenA][lenB] with the gap-penalty boundary (:112-120), aligned strings from the
 * reference's traceback, which stops when EITHER index reaches 0 (:141) -- so
 * ("", "GAT") gives score 3*gap and two empty strings.  err as polyhip_sw_batch.
 * Strings: last d_alnLen[p] bytes of the aln_stride-byte slots, aln_stride >=
 * max_lenA + lenB.  Workspace: polyhip_nw_workspace_bytes (whole direction matrix
 * per pair; the call loops over chunks of pairs if given less, >= 256 pairs' worth).
 */
size_t polyhip_nw_workspace_bytes(uint64_t npairs, uint32_t max_lenA,
                                  uint64_t max_lenB);
int polyhip_nw_align_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A,
                               const uint64_t *d_offA, uint64_t npairs,
                               uint32_t max_lenA, const uint8_t *d_B,
                               const uint64_t *d_offB, uint64_t lenB,
                               int64_t *d_score, uint32_t *d_err,
                               uint8_t *d_alnA, uint8_t *d_alnB,
                               uint32_t *d_alnLen, uint32_t aln_stride,
                               void *d_work, size_t work_bytes,
                               polyhip_stream_t stream);
int polyhip_nw_align_batch(const polyhip_scoring *sc, const uint8_t *A,
                           const uint64_t *offA, uint64_t npairs,
                           const uint8_t *B, const uint64_t *offB,
                           uint64_t lenB, int64_t *score, uint32_t *err,
                           uint8_t *alnA, uint8_t *alnB, uint32_t *alnLen,
                           uint32_t aln_stride);

/* which kernel family the last polyhip_sw_batch*_dev call on this thread used (tests):
 * 1 = lane-per-pair register-tiled shared-B kernel, 2 = generic kernel, 3 = packed two-pairs-per-lane
 * pass + locate + one-wave-per-pair kernel for its ties, 4 = one-wave-per-pair kernel (small batches),
 * 5 = register-tiled kernel for per-pair B, 6 = one-wave-per-pair kernel for what those cannot take (reads of
 * 257..4096 symbols, gap >= 0, scores beyond int8; shared or per-pair B), 7 = reads of 257..2048 symbols
 * against one reference, enough of them to fill the chip: packed pass with 2..16 lanes per pair + the
 * one-wave-per-pair kernel over the columns that can reach the maximum.  POLYHIP_SW_WAVE=0 /
 * POLYHIP_SW_PACKED=0 / POLYHIP_SW_PAIR=0 in the environment switch 4, 6 and 7 / 3 and 7 / 5 off (testing aids). */
int polyhip_sw_last_path(void);
/* 1 when that call's packed pass (paths 3 and 7) ran the half-float cell of gfx950 (v_pk_maximum3_f16: three
 * instructions per cell pair instead of four) -- taken when every H stays below 2048, i.e. smax * min(max_lenA, lenB)
 * <= 2047 and smax + |gap| <= 2048; same integers, bit for bit (halves scaled by 2^-11, every sum exact).
 * POLYHIP_SW_F16=0 keeps the int16 cell (testing aid). */
int polyhip_sw_last_packed_half(void);
/* ... and over how many lanes that packed pass spread the rows of a lane's two read pairs (0: no packed pass): 1 = one
 * lane holds all rows (up to 64 rows; POLYHIP_SW_PK1X2=0), 2 = sw_pk1x2_kernel (65..152 rows: two lanes, four waves per
 * SIMD), 2..16 = sw_pkb_kernel's lanes per pair above 152 rows (64 rows per lane; POLYHIP_SW_TILE64=0: 128 / 152). */
int polyhip_sw_last_packed_lanes(void);
/* ... and the last polyhip_sw_traceback_dev call: 1 = byte-profile kernel (shared B, score given,
 * the reference's profile fits LDS), 2 = register-tiled table kernel, 3 = generic kernel, 4 = one-wave-per-pair
 * kernel for reads of 153..4096 symbols (tests; POLYHIP_TB_WAVE=0 switches 4 off), 5 = the half-float byte-profile
 * kernel with TWO LANES per pair (four bands of 64 rows) for reads of 153..256 symbols against one reference under the
 * half-float condition below (POLYHIP_TB_HALF2=0 or POLYHIP_TB_F16=0: path 4 instead; testing aids), 6 = the half-float
 * kernel for EVERY PAIR ITS OWN B (reads against reads): reads of at most 152 symbols, at most six symbol codes, score
 * given, the half-float condition -- a lane builds its own profile from its pair's B symbols (tb_pair16_kernel;
 * POLYHIP_TB_PAIR16=0 or POLYHIP_TB_F16=0: path 2 instead; testing aids), 7 = the one-wave-per-pair kernel for reads of
 * 257..1024 symbols with its sweep on a BYTE PROFILE of the pair in LDS (eight instructions per cell instead of sixteen):
 * gap <= -1, smax - gap <= 127, smin - gap >= -128, the planes of four pairs fit 64 KB (shared or per-pair B;
 * POLYHIP_TB_WAVE8=0: path 4 instead; testing aid).  Paths 4 and 7 walk out of the wave's registers on the scalar unit
 * (reads of at most 1024 symbols; POLYHIP_TB_WALKREG=0: a global load per step as before; testing aid). */
int polyhip_sw_traceback_last_path(void);
/* 1 when that call's byte-profile kernel (path 1) ran in its half-float form (gfx950: packed halves, two bands of rows
 * per lane, nine instructions per cell pair instead of eighteen) -- taken under the packed score pass's condition
 * (every H < 2048) for reads of <= 152 symbols while the table of halves fits twice into a CU's LDS.
 * POLYHIP_TB_F16=0 keeps the 32-bit form (testing aid). */
int polyhip_sw_traceback_last_half(void);
/* ... and the last polyhip_nw_align_batch_dev call: 1 = register-tiled kernel (lenA <= 256), 2 = generic kernel
 * (POLYHIP_NW_GENERIC=1 forces it), 3 = one-wave-per-pair kernel (lenA 257..4096); tests. */
int polyhip_nw_last_path(void);

/* ---- K4: primers SantaLucia / MarmurDoty / MeltingTemp  (primers/primers.go:70-128) */
/*
 * SCAN: SantaLucia(seq[i : i+L], primer_conc, salt_conc, mg_conc)
 * (primers.go:70-105) for every start i in [start0, start0 + nstarts) and every
 * length L in [Lmin, Lmax] of ONE sequence of `len` bytes (the caller of
 * primers/pcr's grow-until-Tm loops, pcr.go:47-53, reads its answers from this
 * table).  Output planes, one per length:
 *     d_tm[(L - Lmin) * ld + (i - start0)]   (likewise d_dH, d_dS),  ld >= nstarts
 * A window that runs past the end of the sequence (i + L > len) gets quiet
 * NaNs.  start0/nstarts let each rank of a multi-GPU job scan its own slice of
 * the starts while reading its (Lmax - 1)-byte halo from the same buffer.
 * Results are bit-identical to the Go code: same fp64 operation order, no FMA
 * contraction, Go's math.Log algorithm for the two logarithms.
 * Lmin == 0 -> POLYHIP_ERR_PANIC (SantaLucia("") panics, primers.go:89).
 * Range: Lmax <= 1024.  MeltingTemp (primers.go:121-128) is this call with
 * (500e-9, 50e-3, 0).
 */
int polyhip_santalucia_scan_dev(const uint8_t *d_seq, uint64_t len,
                                uint64_t start0, uint64_t nstarts,
                                uint32_t Lmin, uint32_t Lmax,
                                double primer_conc, double salt_conc,
                                double mg_conc, double *d_tm, double *d_dH,
                                double *d_dS, uint64_t ld,
                                polyhip_stream_t stream);
/* host flavour: all starts 0 .. len - Lmin, ld = len - Lmin + 1; outputs hold
 * (Lmax - Lmin + 1) * ld doubles each.  Non-ASCII bytes -> POLYHIP_ERR_INVALID. */
int polyhip_santalucia_scan(const uint8_t *seq, uint64_t len, uint32_t Lmin,
                            uint32_t Lmax, double primer_conc,
                            double salt_conc, double mg_conc, double *tm,
                            double *dH, double *dS);
/*
 * SCAN, reduced on the chip: for every start the FIRST length L in [Lmin, Lmax] whose SantaLucia Tm is not below
 * target_tm -- the grow loop of primers/pcr (pcr.go:47-53: lengthen the primer while MeltingTemp < targetTm; the
 * same comparison, so a NaN Tm stops it too) for every position of a sequence at once.  d_first_len[i - start0] = that L
 * (0: no length up to Lmax reaches the target, or no window fits), d_first_tm (may be NULL) its Tm (a NaN where no
 * length was found).  2 + 8 bytes per start leave the chip instead of 24 per window: the host flavour of the
 * full scan is bound by PCIe (1.56 GB for a 5 Mb genome), this one is not.
 */
int polyhip_santalucia_scan_first_dev(const uint8_t *d_seq, uint64_t len,
                                      uint64_t start0, uint64_t nstarts,
                                      uint32_t Lmin, uint32_t Lmax,
                                      double primer_conc, double salt_conc,
                                      double mg_conc, double target_tm,
                                      uint16_t *d_first_len, double *d_first_tm,
                                      polyhip_stream_t stream);
int polyhip_santalucia_scan_first(const uint8_t *seq, uint64_t len, uint32_t Lmin,
                                  uint32_t Lmax, double primer_conc,
                                  double salt_conc, double mg_conc,
                                  double target_tm, uint16_t *first_len,
                                  double *first_tm);
/* BATCH: one SantaLucia call per packed sequence.  An empty sequence is
 * POLYHIP_ERR_PANIC in the host flavour (quiet NaN outputs in the _dev one). */
int polyhip_santalucia_batch_dev(const uint8_t *d_seqs,
                                 const uint64_t *d_offsets, uint64_t n,
                                 double primer_conc, double salt_conc,
                                 double mg_conc, double *d_tm, double *d_dH,
                                 double *d_dS, polyhip_stream_t stream);
int polyhip_santalucia_batch(const uint8_t *seqs, const uint64_t *offsets,
                             uint64_t n, double primer_conc, double salt_conc,
                             double mg_conc, double *tm, double *dH,
                             double *dS);
/* MarmurDoty (primers.go:108-118) per packed sequence. */
int polyhip_marmurdoty_batch_dev(const uint8_t *d_seqs,
                                 const uint64_t *d_offsets, uint64_t n,
                                 double *d_tm, polyhip_stream_t stream);
int polyhip_marmurdoty_batch(const uint8_t *seqs, const uint64_t *offsets,
                             uint64_t n, double *tm);

/* ---- K5: seqhash RotateSequence  (seqhash/seqhash.go:78-138) ------------------ */
/*
 * For every packed sequence: d_rot_index[i] = boothLeastRotation(seq_i)
 * (seqhash.go:78-124: the smallest index of the lexicographically least
 * rotation, byte order, no case folding), and -- if d_rotated != NULL, same
 * packed layout as the input -- RotateSequence(seq_i) = (seq_i + seq_i)[r : r+n]
 * (seqhash.go:127-138).  Empty and one-byte sequences give index 0.
 * max_len >= every sequence length (sizes the LDS staging).
 */
int polyhip_least_rotation_batch_dev(const uint8_t *d_seqs,
                                     const uint64_t *d_offsets, uint64_t n,
                                     uint64_t max_len, uint64_t *d_rot_index,
                                     uint8_t *d_rotated,
                                     polyhip_stream_t stream);
/* host flavour; rotated may be NULL, else it is indexed by the same offsets. */
int polyhip_least_rotation_batch(const uint8_t *seqs, const uint64_t *offsets,
                                 uint64_t n, uint64_t *rot_index,
                                 uint8_t *rotated);

/* ---- S2: seqhash.Hash  (seqhash/seqhash.go:141-224) --------------------------- */
/*
 * Hash(seq_i, sequenceType, circular, doubleStranded) for every packed sequence,
 * one (type, circular, doubleStranded) triple per call: upper-case, RNA U->T,
 * alphabet check, least rotation / reverse complement / bytewise-smaller choice,
 * BLAKE3-256, "v1_" + {D,R,P}{C,L}{D,S} + "_" + 64 hex digits.
 * seq_type: 0 DNA, 1 RNA, 2 PROTEIN; anything else -> POLYHIP_ERR_INVALID with the
 * reference's message (seqhash.go:152); PROTEIN + double_stranded likewise (:175).
 * d_out: n slots of 72 bytes (71 characters + NUL; empty string on error).
 * d_err[i]: 0, or (2 << 8) | letter for seqhash.go:157 ("Only letters
 * ATUGCYRSWKMBDHVNZ are allowed for DNA/RNA. Got letter: X"), (3 << 8) | letter
 * for seqhash.go:169 (proteins) -- the first offending letter, as the reference.
 * d_offsets[0] must be 0 (the normalised copy in the workspace is addressed by the
 * batch's own offsets; the host flavour rebases); total_bytes = d_offsets[n];
 * max_len >= every sequence length.
 */
size_t polyhip_seqhash_workspace_bytes(uint64_t n, uint64_t total_bytes,
                                       int circular, int double_stranded);
int polyhip_seqhash_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets,
                              uint64_t n, uint64_t total_bytes,
                              uint64_t max_len, int seq_type, int circular,
                              int double_stranded, char *d_out,
                              uint32_t *d_err, void *d_work, size_t work_bytes,
                              polyhip_stream_t stream);
int polyhip_seqhash_batch(const uint8_t *seqs, const uint64_t *offsets,
                          uint64_t n, int seq_type, int circular,
                          int double_stranded, char *out, uint32_t *err);

/* ---- read feeder: io/fastq (*Parser).ParseNext / ParseN  (io/fastq/fastq.go:84-216) ---- */
/*
 * A FASTQ file image (d_file, nbytes) becomes the packed batch the kernels above take:
 * d_seqs = the Sequence of every record back to back, d_offsets[0..n] (n+1 entries),
 * d_rec_start[i] (optional) = byte offset of record i's identifier line, for the host to
 * slice identifiers lazily.  Records are four '\n'-terminated lines; a '\r' is not stripped;
 * parsing stops at the first bad record and the records before it are kept (ParseN).
 * d_result[0] = n records, [1] = error code (0 none; 1 no '@' (fastq.go:203), 2 empty
 * sequence (:176), 3 empty quality (:197), 4 unexpected EOF inside a record / last line
 * without '\n' (:142-148), 5 empty identifier line and 6 identifier field without '='
 * (the reference PANICS there, :156 and :163), 7 more records than max_records),
 * [2] = the line the reference's message names, [3] = total sequence bytes.
 * Capacities: d_seqs nbytes, d_offsets / d_rec_start nbytes/7 + 2 entries (or max_records + 1;
 * the shortest record is the 7 bytes "@\nA\n\nI\n": the third line is read unseen, fastq.go:182).
 */
size_t polyhip_fastq_workspace_bytes(uint64_t nbytes);
int polyhip_fastq_pack_dev(const uint8_t *d_file, uint64_t nbytes,
                           uint8_t *d_seqs, uint64_t *d_offsets,
                           uint64_t *d_rec_start, uint64_t max_records,
                           uint64_t *d_result, void *d_work, size_t work_bytes,
                           polyhip_stream_t stream);
int polyhip_fastq_pack(const uint8_t *file, uint64_t nbytes, uint8_t *seqs,
                       uint64_t *offsets, uint64_t *rec_start,
                       uint64_t max_records, uint64_t *result);

/* ---- read feeder: io/fasta (*Parser).ParseNext / ParseN  (io/fasta/fasta.go:102-238) ---- */
/*
 * Same contract as polyhip_fastq_pack for a FASTA image, multi-line records included, with
 * the reference's rules: empty lines and ';' lines are skipped, lines before the first '>'
 * are skipped, a '>' line directly after a header is SEQUENCE (fasta.go:197-204 looks at the
 * next line only after a line has been read), a last record ending in an unterminated line is
 * dropped (ParseNext returns it with io.EOF).  d_rec_start[i] = byte offset of record i's
 * header line.  d_result[0] = n records, [1] = error code (0 none; 1 no '>' in a non-empty
 * file (:223), 2 a header without sequence (:227) -- records before it are kept; 7 more
 * records than max_records), [2] = total sequence bytes, [3] = header lines seen.
 * Capacities: d_seqs nbytes; d_offsets / d_rec_start nbytes/2 + 3 entries.
 */
size_t polyhip_fasta_workspace_bytes(uint64_t nbytes);
int polyhip_fasta_pack_dev(const uint8_t *d_file, uint64_t nbytes,
                           uint8_t *d_seqs, uint64_t *d_offsets,
                           uint64_t *d_rec_start, uint64_t max_records,
                           uint64_t *d_result, void *d_work, size_t work_bytes,
                           polyhip_stream_t stream);
int polyhip_fasta_pack(const uint8_t *file, uint64_t nbytes, uint8_t *seqs,
                       uint64_t *offsets, uint64_t *rec_start,
                       uint64_t max_records, uint64_t *result);

/* ---- R1: the path's one collective -- all-gather of per-rank sketches (RCCL over xGMI) ---- */
/*
 * For hosts without torch.distributed (the Go/cgo drop-in); one process per GPU.  RCCL is
 * resolved at run time (dlopen librccl.so.1), so libpolyhip has no link-time dependency on it.
 * Rank 0 obtains the 128-byte id and passes it to the other ranks by its own channel; every
 * rank then creates its communicator on its current HIP device.  d_all receives
 * nranks * n_local sketches in rank order; the call enqueues one ncclAllGather on `stream`.
 * EVERY rank must pass the SAME n_local (ncclAllGather's contract; the library cannot check it
 * across processes): with ragged shards, pad each rank's block to the largest shard and trim
 * after the gather, as poly_amd/sharding.py::gather_sketches does.  n_local == 0 still enters
 * the collective (all ranks then contribute nothing).  Then each rank runs
 * polyhip_mash_shared_counts_dev(X = its block of d_all, Y = d_all).
 */
typedef struct polyhip_comm polyhip_comm;
int polyhip_comm_unique_id(uint8_t id[128]);
int polyhip_comm_init_rank(const uint8_t id[128], int rank, int nranks,
                           polyhip_comm **out);
int polyhip_comm_destroy(polyhip_comm *c);
int polyhip_comm_rank(const polyhip_comm *c);
int polyhip_comm_size(const polyhip_comm *c);
int polyhip_allgather_sketches_dev(polyhip_comm *c, const uint32_t *d_local,
                                   uint64_t n_local, uint32_t s,
                                   uint32_t *d_all, polyhip_stream_t stream);
/* Ragged all-gather IN PLACE: rank r owns bytes [offsets[r], offsets[r+1]) of d_buf (nranks + 1 ascending offsets, the
 * same on every rank) and every rank ends up with all segments -- one grouped call of nranks ncclBroadcasts, each rank
 * the root of its own segment.  What polyhip_mash_index_allgather_dev moves the parts of the index with. */
int polyhip_allgatherv_dev(polyhip_comm *c, void *d_buf, const uint64_t *offsets,
                           polyhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* POLYHIP_H */
