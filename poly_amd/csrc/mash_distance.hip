// mash_distance.hip -- K2: all-pairs (*Mash).Similarity / Distance for gfx950.
//
// Replaces the merge loop of search/mash/mash.go:107-135 (and Distance,
// :138-140) for every pair (X_i, Y_j) of two sets of sketches:
//     counts[i][j] = sameHashes of X_i.Similarity(Y_j)          (u16)
//     dist[i][j]   = 1 - float64(counts[i][j]) / float64(min(sx, sy))
// (all-vs-all: Y = every gathered sketch, X = this rank's row block of Y.)
//
// The reference merges two sorted lists, <= sx+sy serial steps per pair; doing
// that for 1e10 pairs is 2e13 dependent steps.  Sketches are sorted, so the
// same number falls out of a JOIN ON THE HASH VALUE instead:
//     sameHashes(X_i, Y_j) = sum over values v of min(mult_Xi(v), mult_Yj(v))
// which is a sparse product (sketch x value incidence) * (value x sketch), and
// the work becomes proportional to the number of shared hashes, not to the
// number of pairs:
//
//   check    every sketch: ascending?  (a sketch of a sequence with fewer than
//            s windows is positional/unsorted, mash.go:81-84)  -> per-sketch
//            "irregular" flag; largest last element of Y -> bucket shift; the
//            same pass counts the regular sketches' items per coarse bucket
//   index    the Y side becomes an inverted index: items (value, sketch id,
//            occurrence number among equal values of that sketch) partitioned
//            into NBK buckets by value >> shift (histogram, scan, scatter)
//   rowjoin  one workgroup per X row (Gustavson with an LDS hash accumulator):
//            for every distinct value v of the row (multiplicity a) read v's
//            bucket -- one coalesced load -- and for every item (v, j, occ) with
//            occ < a bump the row's LDS hash table at key j (LDS atomics, no
//            global atomics); then flush the table's entries to counts[i][*].
//            "occ < a" makes a value that is a times in X_i and b times in Y_j
//            count min(a, b) times: what the reference's two-pointer merge counts.
//   dense    rows whose hash table overflows (thousands of related sketches): the
//            same bucket walk with DENSE 16-bit counters in LDS, one per column of
//            a 65,536-column stripe (rowjoin_dense_kernel)
//   generic  the reference's own loop, one pair per lane, for every pair that
//            involves an irregular sketch (the range early-out of mash.go:117
//            and the merge both read unsorted data there), and for ALL pairs when
//            the index says the input is dense (huge buckets: many near-identical
//            sketches), where merging is cheaper than joining.
//
// No MFMA (integer compares).  HBM-bound part: the counts/distance matrix
// itself (2 or 8 B per pair) -- see DESIGN.md, K2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "host_pipeline.h"

namespace polyhip {
namespace k2 {

constexpr int THREADS = 256;
// An index item is (value, sketch id | occurrence number << id_bits): id_bits = ceil(log2(ny)) (<= ID_BITS_MAX),
// the rest of the dword counts the copies of the value inside one sketch, so that the join can apply the merge's
// multiset semantics (occ < multiplicity in the row).  With 100k sketches that leaves 15 bits: a sketch made of
// one repeated hash (a homopolymer read) is an ordinary citizen, not an "irregular" one that drags every pair it
// is part of through the reference's merge loop.
constexpr uint32_t ID_BITS_MAX = 24;
constexpr uint32_t TAB = 2048;        // LDS hash accumulator slots per row
constexpr uint32_t TAB_LIMIT = 1536;  // distinct columns a row may hit before it goes to the merge
constexpr uint32_t S_MAX = 8192;      // X SketchSize rowjoin stages in LDS
constexpr int JOIN_U = 8;             // buckets a wave keeps in flight
#ifndef PH_K2_NCLOG
#define PH_K2_NCLOG 10 // log2 of the most coarse buckets of the index build
#endif

enum { H_MAXVAL = 0, H_SHIFT, H_MODE, H_NIRRX, H_NIRRY, H_NREGX, H_NOVF, H_MAXMULT, H_EST_LO, H_EST_HI, H_FMT,
       H_B4, H_B4_NC, H_B4_R, H_B4_CPP, H_B4_MAGIC, H_B4_OVER, H_B4_NDUP, H_WORDS = 32 };
// H_B4: which index build runs (decided on the device, plan4_kernel / lists_kernel): 0 = the two-level build on 8-byte
// intermediate items (round 2-3), 1 = the sliced build on 4-byte intermediate items (round 5, below), 2 = the sliced build
// was planned but the items cannot be compact after all (a sketch repeats a hash too often): the two-level build takes over.
// H_B4_NC coarse buckets (value >> 16; a power of two), H_B4_R parts of the value range, H_B4_CPP coarse buckets per part,
// H_B4_MAGIC = ceil(2^32 / CPP), H_B4_OVER = coarse buckets that took level 2's two-pass path (diagnostics), H_B4_NDUP =
// records in the list of repeated hashes (below).
// H_FMT: 0 = 8-byte items (value, sketch id | occurrence number << id_bits); 1 = COMPACT 4-byte items, written when the
// join is known to be the one-stripe dense join and the bits fit (decided on the device, lists_kernel):
//     [ value's bits below the bucket : shift | occurrence number : 11 - shift | counter dword : 16 | field shift : 5 ]
// (occurrence number stored + 1, so that the all-zero word is "no item") -- the LDS counter a shared hash bumps (dword =
// column % ndw, field = column / ndw) is worked out ONCE, when the index is built, instead of once per visit: the join's
// inner step becomes subtract, compare, shift, shift, and, ds_add, and its loads are buffer loads bounded by the bucket.
constexpr uint32_t CK_LOW = 21; // bits below the occurrence number
enum { MODE_SPARSE = 0, MODE_GENERIC = 1 };

struct Layout {
    uint32_t nbk, nbk_log2, nc, nc_log2, fpc_log2;
    size_t off_flagsX, off_flagsY, off_irrX, off_regX, off_irrY, off_ovfX;
    size_t off_start, off_gcount, off_cstart, off_gcur, off_pos, off_g4count, off_dup, off_duplist, off_c4start, off_g4cur, off_citems, off_items;
    size_t total;
};
// The SLICED build on 4-byte intermediate items (round 5; B4 for short).  A sketch is ascending, so its hashes of one part of
// the value range are CONTIGUOUS in it.  The coarse bucket of a hash is `value >> 16`; an intermediate item is
//     [ value & 0xFFFF : 16 | sketch id & 0xFFFF : 16 ]
// -- the coarse bucket is where the item lies, and the sketch id's bit 16 (up to 131,072 sketches: what the one-stripe dense
// join takes) is WHICH of a coarse bucket's two segments it lies in -- half the bytes of the (value, id) pair the two-level
// build moves, written once and read once:
//   plan    geometry from the largest value (device side): NC = 2^(bits - 16) coarse buckets, R parts of CPP coarse buckets
//   check   a wave per sketch (as before) + the histogram over value >> 16 + where the sketch crosses from part to part
//   level 1 work item (B sketches, part r): the B slices, up to SLOTS hashes each per round, ordered by coarse bucket in an
//           LDS stage of 16,384 four-byte items and written out as runs (a quarter wave per run)
//   level 2 a workgroup per coarse bucket: the bucket in REGISTERS (up to 32 items per thread), the fine histogram in LDS, the
//           atomic's return value as the item's rank: one read of the intermediate items, the final compact items written by
//           the bucket's only owner; a coarse bucket beyond the registers takes two passes over its intermediate items.
// The occurrence number (copies of a value inside ONE sketch) has no room in 32 bits.  Such copies are rare (~1 sketch in
// 1000 at config 3): the check pass, which walks them anyway, LOGS every copy after the first as (value, id | number << 17) in
// a list and marks the value's coarse bucket (dupmap); level 2 places all items as "first copy" and then, in a marked bucket,
// one thread per logged record of that bucket finds the record's slot among the equal items of its fine bucket (the number-th
// one in slot order) and writes the number in.  More than B4_DUP_MAX records: the two-level build.
// Conditions (else the two-level build): compact items (so one-stripe dense join: <= 131,072 sketches here), SketchSize <=
// 1024, the largest value below 2^30 and at least 2^16, 4 <= bucket shift <= 10.  POLYHIP_K2_B4=0 switches it off.
constexpr uint32_t B4_RMAX = 64;       // parts of the value range at most (pos: R + 1 positions per sketch)
constexpr uint32_t B4_CPP_MAX = 1024;  // coarse buckets per part at most (level 1's counters in LDS)
constexpr uint32_t B4_NC_MAX = 16384;  // coarse buckets at most (largest value < 2^30)
constexpr uint32_t B4_STAGE = 16384;   // items of a level-1 stage (64 KB: two workgroups per CU)
constexpr uint32_t b4_cpp_max(int batch) { return batch > 256 ? 512u : B4_CPP_MAX; } // (level 1's LDS: two workgroups per CU)
constexpr uint32_t B4_DUP_MAX = 65536; // records of repeated hashes the build numbers itself; more: the two-level build

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static Layout layout(uint64_t nx, uint32_t sx, uint64_t ny, uint32_t sy)
{
    Layout L;
    (void)sx;
    // ~PH_K2_BUCKET_ITEMS Y items per bucket; a bucket is a value range of one width (a power of two).  A row's bucket holds
    // the copies of ITS hash in the row's family (~80 at config 3) plus whatever else falls into the value range: 32 per
    // bucket made a quarter of the walk's items foreign ones.  Measured at config 3 (join per row block / index / full
    // matrix): 64 -> 1.76 / 2.12 / 15.4 ms, 32 -> 1.35 / 2.11 / 13.0, 16 -> 1.26 / 2.15 / 11.9, 8 -> 1.21 / 2.74 / 12.1
    // (2^24 buckets need 2048 coarse ones, which the LDS-staged level-1 scatter does not hold).
#ifndef PH_K2_BUCKET_ITEMS
#define PH_K2_BUCKET_ITEMS 16
#endif
    const uint64_t itemsY = ny * (uint64_t)sy;
    uint32_t nbk = 2048;
    while (nbk < (1u << 24) && (uint64_t)nbk * PH_K2_BUCKET_ITEMS < itemsY)
        nbk <<= 1;
    L.nbk = nbk;
    L.nbk_log2 = 0;
    while ((1u << L.nbk_log2) < nbk)
        ++L.nbk_log2;
    // coarse buckets: few enough that a level-1 workgroup's slice of one is several cache lines long (its 8-byte items
    // are scattered straight to HBM), many enough that level 2 splits a coarse bucket with an LDS histogram (FPC_MAX)
    uint32_t ncl = L.nbk_log2 < (uint32_t)PH_K2_NCLOG ? L.nbk_log2 : (uint32_t)PH_K2_NCLOG;
    if (L.nbk_log2 - ncl > 13u)
        ncl = L.nbk_log2 - 13u; // FPC_MAX = 2^13
    L.nc_log2 = ncl;
    L.nc = 1u << L.nc_log2;
    L.fpc_log2 = L.nbk_log2 - L.nc_log2;
    // the Y side (flags, irregular list, inverted index) comes first and does not depend on nx: a later call with
    // another X (polyhip_mash_shared_counts_reuse_dev) finds it where the call that built it left it
    size_t o = al(H_WORDS * 4);
    L.off_flagsY = o; o += al(ny);
    L.off_irrY = o; o += al(ny * 4);
    L.off_start = o; o += al(((size_t)nbk + 1) * 4);
    L.off_gcount = o; o += al((size_t)L.nc * 4);
    L.off_cstart = o; o += al(((size_t)L.nc + 1) * 4);
    L.off_gcur = o; o += al((size_t)L.nc * 4);
    L.off_pos = o; o += al(ny * (size_t)(B4_RMAX + 1) * 2);
    L.off_g4count = o; o += al((size_t)2 * B4_NC_MAX * 4); // (zeroed together with the dupmap behind it)
    L.off_dup = o; o += al(B4_NC_MAX / 8);
    L.off_duplist = o; o += al((size_t)B4_DUP_MAX * 8);
    L.off_c4start = o; o += al(((size_t)2 * B4_NC_MAX + 1) * 4);
    L.off_g4cur = o; o += al((size_t)2 * B4_NC_MAX * 4);
    L.off_citems = o; o += al(ny * (size_t)sy * 8);
    L.off_items = o; o += al(ny * (size_t)sy * 8);
    L.off_flagsX = o; o += al(nx);
    L.off_irrX = o; o += al(nx * 4);
    L.off_regX = o; o += al(nx * 4);
    L.off_ovfX = o; o += al(nx * 4);
    L.total = o;
    return L;
}

// ---- the largest value of the Y side: an ascending sketch ends with its largest value, and a sketch that is not
// ascending never enters the index -- one load per sketch gives the bucket shift before the sketches are read at all
__global__ __launch_bounds__(1024) void maxlast_kernel(const uint32_t *__restrict__ sk, uint64_t n, uint32_t s,
                                                      uint32_t *__restrict__ hdr)
{
    // grid-stride, one atomic per WORKGROUP: a thousand and a half wave-level atomicMax on one word took most of the
    // kernel's 21 us (profiles/r05c_k2_stats.md)
    __shared__ uint32_t wmax[16];
    uint32_t v = 0;
    for (uint64_t q = (uint64_t)blockIdx.x * 1024 + threadIdx.x; q < n; q += (uint64_t)gridDim.x * 1024)
        v = max(v, sk[q * s + (s - 1)]);
    v = dpp_wave_max(v);
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        v = dpp_wave_max(threadIdx.x < 16 ? wmax[threadIdx.x] : 0u);
        if (threadIdx.x == 0 && v > __hip_atomic_load(&hdr[H_MAXVAL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&hdr[H_MAXVAL], v);
    }
}

__device__ __forceinline__ uint32_t bucket_shift(uint32_t maxval, uint32_t nbk_log2)
{
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(maxval | 1u);
    return bits > nbk_log2 ? bits - nbk_log2 : 0u;
}

// ---- check: ascending? occurrence number representable?  One WAVE per sketch, grid-stride, no barrier: up to 1024
// elements of the sketch sit in the lanes' registers from one round of loads, the right-hand neighbour comes from the next
// lane (the last lane loads its own), and an ascending sketch without a repeated value -- nearly all -- is done after 16
// compares.  (A workgroup per sketch, one element per trip: 0.42 ms for 100k sketches of 1000.)  On the Y side the same
// pass counts the regular sketches' items per coarse bucket from those registers: the separate counting pass over Y
// (0.09 ms) and its read of the flags are gone.
template <bool YSIDE>
__global__ __launch_bounds__(1024) void check_kernel(const uint32_t *__restrict__ sk, uint64_t n, uint32_t s,
                                                    uint8_t *__restrict__ flags, uint32_t *__restrict__ hdr,
                                                    int force_irregular, uint32_t max_occ, uint32_t nbk_log2,
                                                    uint32_t cshift_extra, uint32_t nc, uint32_t *__restrict__ gcount,
                                                    uint32_t run_if_b4)
{
    extern __shared__ uint32_t lh[]; // YSIDE: nc coarse counters
    uint32_t cshift = 0;
    if (YSIDE) {
        // the sliced build's own check pass (check4_kernel) has run instead (H_B4 == 1); a second launch (run_if_b4 == 2)
        // takes over when that build was planned and then called off (lists_kernel)
        if (hdr[H_B4] != run_if_b4)
            return;
        for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x)
            lh[c] = 0;
        cshift = bucket_shift(hdr[H_MAXVAL], nbk_log2) + cshift_extra;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * (blockDim.x / 64);
    uint32_t multmax = 0; // over this wave's regular sketches
    for (uint64_t q = wave; q < n; q += nwaves) {
        const uint32_t *p = sk + q * s;
        bool bad = force_irregular != 0, anyeq = false;
        uint32_t x[16];
        for (uint32_t e0 = 0; e0 < s; e0 += 1024) {
            uint32_t edge[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t e = e0 + u * 64 + lane;
                x[u] = e < s ? p[e] : 0u;
                edge[u] = (lane == 63u && e + 1 < s) ? p[e + 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t e = e0 + u * 64 + lane;
                uint32_t nx = (uint32_t)__shfl_down((int)x[u], 1, 64);
                if (lane == 63u)
                    nx = edge[u];
                const bool has = e + 1 < s;
                bad = bad || (has && x[u] > nx);
                anyeq = anyeq || (has && x[u] == nx);
            }
        }
        bool anybad = __ballot(bad) != 0ull; // an irregular sketch never enters the index
        uint32_t mult = 1;
        if (YSIDE && !anybad && __ballot(anyeq) != 0ull) {
            // repeated values inside an ascending sketch (rare): the longest run, a first copy counting its own
            mult = 0;
            for (uint32_t e = lane; e < s; e += 64) {
                const uint32_t x0 = p[e];
                if (e == 0 || p[e - 1] != x0) {
                    uint32_t a = 1;
                    while (e + a < s && p[e + a] == x0)
                        ++a;
                    mult = max(mult, a);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1)
                mult = max(mult, (uint32_t)__shfl_xor((int)mult, d, 64));
            if (mult - 1u > max_occ) // more equal values in one sketch than an item can number
                anybad = true;
        }
        if (anybad && lane == 0)
            flags[q] = 1;
        if (YSIDE && !anybad) {
            multmax = max(multmax, mult);
            if (s <= 1024) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u * 64 + lane < s)
                        atomicAdd(&lh[x[u] >> cshift], 1u);
            } else {
                for (uint32_t e = lane; e < s; e += 64)
                    atomicAdd(&lh[p[e] >> cshift], 1u);
            }
        }
    }
    if (YSIDE) {
        // nearly every wave sees a maximum that is already recorded: read before the atomic
        if (lane == 0 && multmax > __hip_atomic_load(&hdr[H_MAXMULT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&hdr[H_MAXMULT], multmax);
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x)
            if (lh[c])
                atomicAdd(&gcount[c], lh[c]);
    }
}

static inline unsigned check_grid(uint64_t n) // a wave per sketch, at most 8 workgroups per CU
{
    const uint64_t wg = (n + THREADS / 64 - 1) / (THREADS / 64);
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(wg, 256ull * 8ull));
}

// ---- lists of irregular / regular sketches, bucket shift ---------------------------
__global__ __launch_bounds__(THREADS) void lists_kernel(const uint8_t *__restrict__ flagsX, uint64_t nx,
                                                       const uint8_t *__restrict__ flagsY, uint64_t ny,
                                                       uint32_t *__restrict__ irrX, uint32_t *__restrict__ regX,
                                                       uint32_t *__restrict__ irrY, uint32_t *__restrict__ hdr,
                                                       uint32_t nbk_log2, int allow_compact)
{
    const uint64_t t = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    // one atomic per wave and list, not one per sketch: 100k increments of one word took as long as the join of 600 rows
    auto append = [&](bool mine, uint32_t *counter, uint32_t *list) {
        const uint64_t m = __ballot(mine);
        if (m == 0ull)
            return;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m))
            base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), 64);
        if (mine)
            list[base + (uint32_t)__builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (uint32_t)t;
    };
    const bool inX = t < nx, badX = inX && flagsX[t] != 0;
    append(badX, &hdr[H_NIRRX], irrX);
    append(inX && !badX, &hdr[H_NREGX], regX);
    append(t < ny && flagsY[t] != 0, &hdr[H_NIRRY], irrY);
    if (t == 0) {
        const uint32_t shift = bucket_shift(hdr[H_MAXVAL], nbk_log2);
        hdr[H_SHIFT] = shift;
        // compact items: the value's low bits and the occurrence number + 1 share 11 bits (the all-zero word is the join's
        // "no item": what a buffer load returns beyond the end of a bucket)
        if (allow_compact >= 0) {
            hdr[H_FMT] = (allow_compact && shift <= 10u && hdr[H_MAXMULT] <= (1u << (11u - shift)) - 1u) ? 1u : 0u;
            // the sliced build makes compact items only, and numbers at most B4_DUP_MAX repeated hashes
            if (hdr[H_B4] == 1u && (hdr[H_FMT] == 0u || hdr[H_B4_NDUP] > B4_DUP_MAX))
                hdr[H_B4] = 2u;
        }
    }
}

// ---- two-level partition of the Y items into value buckets ---------------------------------
// 1e8 global atomics (one per item, twice) cost 10 ms; partitioning in two levels keeps the
// per-item atomics in LDS.  Level 1: NC coarse buckets (the top bits of the fine bucket): every
// workgroup histograms its batch of sketches in LDS and touches global memory once per (workgroup,
// coarse bucket).  Level 2: one workgroup per coarse bucket splits it into its fine buckets with an
// LDS histogram + scan, writes start[] and the final item order.

constexpr uint32_t FPC_MAX = 8192;   // fine buckets per coarse bucket
constexpr uint32_t BATCH_ITEMS = 65536; // items a level-1 workgroup takes at a time

// exclusive scan of gcount[nc] (nc <= 4096) -> cstart[nc + 1], gcur = copy; one workgroup
__global__ __launch_bounds__(1024) void coarse_scan_kernel(const uint32_t *__restrict__ gcount, uint32_t nc,
                                                          uint32_t *__restrict__ cstart, uint32_t *__restrict__ gcur,
                                                          uint32_t *__restrict__ start, uint32_t nbk)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nc; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < nc ? gcount[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d)
                incl += t;
        }
        if ((tid & 63) == 63)
            wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t pre = carry;
        for (int w = 0; w < (tid >> 6); ++w)
            pre += wsum[w];
        if (i < nc) {
            cstart[i] = pre + incl - v;
            gcur[i] = pre + incl - v;
        }
        __syncthreads();
        if (tid == 1023)
            carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) {
        cstart[nc] = carry;
        start[nbk] = carry;
    }
}

// level 1 scatter: items of my batch -> citems, grouped by coarse bucket
__global__ __launch_bounds__(THREADS) void coarse_scatter_kernel(const uint32_t *__restrict__ sk, uint64_t n, uint32_t s,
                                                                const uint8_t *__restrict__ flags,
                                                                const uint32_t *__restrict__ hdr, uint32_t cshift_extra,
                                                                uint32_t nc, uint32_t per_batch, uint32_t id_bits,
                                                                uint32_t c0, uint32_t c1,
                                                                uint32_t *__restrict__ gcur, uint2 *__restrict__ citems,
                                                                uint32_t id_base)
{
    extern __shared__ uint32_t lh[]; // count[nc] then base[nc]
    uint32_t *lbase = lh + nc;
    if (hdr[H_B4] == 1u)
        return; // the sliced build's level 1 (scatter4_kernel) runs instead
    const uint32_t cshift = hdr[H_SHIFT] + cshift_extra;
    for (uint32_t c = threadIdx.x; c < nc; c += THREADS)
        lh[c] = 0;
    __syncthreads();
    const uint64_t q0 = (uint64_t)blockIdx.x * per_batch, q1 = min(n, q0 + per_batch);
    for (uint64_t q = q0; q < q1; ++q) {
        if (flags[q])
            continue;
        const uint32_t *p = sk + q * s;
        for (uint32_t e = threadIdx.x; e < s; e += THREADS) {
            const uint32_t c = p[e] >> cshift;
            if (c - c0 < c1 - c0) // a part of the index takes the coarse buckets [c0, c1) only
                atomicAdd(&lh[c], 1u);
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < nc; c += THREADS) {
        const uint32_t cnt = lh[c];
        lbase[c] = cnt ? atomicAdd(&gcur[c], cnt) : 0u; // my slice of coarse bucket c
        lh[c] = 0;                                     // becomes the cursor inside the slice
    }
    __syncthreads();
    for (uint64_t q = q0; q < q1; ++q) {
        if (flags[q])
            continue;
        const uint32_t *p = sk + q * s;
        for (uint32_t e = threadIdx.x; e < s; e += THREADS) {
            const uint32_t v = p[e];
            uint32_t occ = 0; // equal values before this one in the same (ascending) sketch
            while (occ < e && p[e - occ - 1] == v)
                ++occ;
            const uint32_t c = v >> cshift;
            if (c - c0 < c1 - c0)
                citems[lbase[c] + atomicAdd(&lh[c], 1u)] = make_uint2(v, (id_base + (uint32_t)q) | (occ << id_bits));
        }
    }
}

// level 1 scatter through LDS: the batch's items are first ordered by coarse bucket in LDS (STAGE_ITEMS of them,
// 64 KB: two workgroups per CU overlap each other's phases; 16384 measured 2.33 ms for the index, 8192 2.19), then leave as runs -- consecutive threads write consecutive items of one bucket's slice, so a wave's store
// touches a handful of cache lines instead of 64 (the direct scatter writes 8 bytes per lane into 64 different lines).
#ifndef PH_K2_STAGE_ITEMS
#define PH_K2_STAGE_ITEMS 8192
#endif
#ifndef PH_K2_STAGE_THREADS
#define PH_K2_STAGE_THREADS 1024
#endif
#ifndef PH_K2_STAGE_GRID
#define PH_K2_STAGE_GRID 512ull // two 1024-thread workgroups per CU
#endif
constexpr uint32_t STAGE_ITEMS = PH_K2_STAGE_ITEMS;
// A wave places items of consecutive buckets: slots ~8 items (64 bytes) apart, four bank groups for 64 lanes.  Flipping an
// item's low three slot bits by bits 5..7 of its slot spreads such a stride over all banks; a run read back in order
// stays inside its own 8 slots.
__device__ __forceinline__ uint32_t stage_swz(uint32_t pos)
{
#ifdef PH_K2_NO_SWZ
    return pos;
#else
    return pos ^ ((pos >> 5) & 7u);
#endif
}
constexpr int STAGE_THREADS = PH_K2_STAGE_THREADS;
__global__ __launch_bounds__(STAGE_THREADS, 8) void coarse_scatter_staged_kernel( // 64 registers: two workgroups per CU
   
    const uint32_t *__restrict__ sk, uint64_t n, uint32_t s, const uint8_t *__restrict__ flags,
    const uint32_t *__restrict__ hdr, uint32_t cshift_extra, uint32_t nc, uint32_t per_batch, uint32_t id_bits,
    uint32_t c0, uint32_t c1, uint32_t *__restrict__ gcur, uint2 *__restrict__ citems, uint32_t id_base)
{
    // id_base: the number of the first sketch of `sk` in the whole set (a device that indexes its own shard of it)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stage[];
    uint2 *stage = reinterpret_cast<uint2 *>(lds_stage);      // STAGE_ITEMS
    uint32_t *cnt = lds_stage + 2 * (size_t)STAGE_ITEMS;        // nc: count, then cursor
    uint32_t *lstart = cnt + nc, *gbase = lstart + nc;          // nc each
    __shared__ uint32_t wsum[STAGE_THREADS / 64];
    const int tid = threadIdx.x;
    if (hdr[H_B4] == 1u)
        return; // the sliced build's level 1 (scatter4_kernel) runs instead
    const uint32_t cshift = hdr[H_SHIFT] + cshift_extra;
    // persistent workgroups, a batch per turn: 12,500 launches of 1024 threads each cost more than the turns' one barrier
    // (the barrier behind the zeroing also keeps a fast wave out of the stage while a slow one still writes it out)
    for (uint64_t batch = blockIdx.x; batch * per_batch < n; batch += gridDim.x) {
        for (uint32_t c = tid; c < nc; c += STAGE_THREADS)
            cnt[c] = 0;
        __syncthreads();
        const uint64_t q0 = batch * per_batch, q1 = min(n, q0 + per_batch);
        // The batch's sketches are one contiguous array of at most STAGE_ITEMS values: a thread keeps its SPT items in
        // registers from ONE round of loads (value, the value in front of it for the occurrence number, the sketch's flag) --
        // a sketch at a time the kernel was a chain of ~20 dependent round trips per workgroup, 83 % of the wave cycles
        // waiting (profiles/r03_k2_pmc_sq.md) -- and the counting atomic's return value IS the item's rank inside its
        // bucket, so the items are placed without a second atomic pass.
        constexpr int SPT = (int)(STAGE_ITEMS / STAGE_THREADS);
        const uint32_t *base = sk + q0 * s;
        const uint32_t nb = (uint32_t)(q1 - q0) * s; // <= STAGE_ITEMS
        const uint32_t sinv = (uint32_t)(((1ull << 32) + s - 1) / s); // i / s by multiply-high: exact for i < 2^16 <= 2^32 / s
        uint32_t v[SPT], rk[SPT], in = 0, dup = 0; // masks: item u is mine / has its own value in front of it in its sketch
        {
            uint32_t pv[SPT];
            uint8_t fl[SPT];
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                const uint32_t i = tid + u * STAGE_THREADS;
                const bool ok = i < nb;
                const uint32_t qi = s > 1 ? __umulhi(i, sinv) : i;
                v[u] = ok ? base[i] : 0u;
                pv[u] = (ok && i > qi * s) ? base[i - 1] : ~v[u]; // the value in front of it in the same sketch
                fl[u] = ok ? flags[q0 + qi] : (uint8_t)1;
            }
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                const uint32_t c = v[u] >> cshift;
                const bool mine = !fl[u] && c - c0 < c1 - c0; // a part of the index takes the coarse buckets [c0, c1) only
                in |= (mine ? 1u : 0u) << u;
                dup |= (pv[u] == v[u] ? 1u : 0u) << u;
                rk[u] = mine ? atomicAdd(&cnt[c], 1u) : 0u;
            }
        }
        __syncthreads();
        // exclusive scan of cnt[0..nc) -> lstart; my slice of every coarse bucket (minus lstart) -> gbase
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < nc; c0 += STAGE_THREADS) {
            const uint32_t c = c0 + tid;
            const uint32_t v = c < nc ? cnt[c] : 0u;
            // my slice of the bucket: the atomic's round trip runs under the scan, its result is not needed before the write-out
            const uint32_t slice = v ? atomicAdd(&gcur[c], v) : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d)
                    incl += t;
            }
            if ((tid & 63) == 63)
                wsum[tid >> 6] = incl;
            __syncthreads();
            uint32_t pre = carry, tot = 0;
            for (int w = 0; w < STAGE_THREADS / 64; ++w) {
                if (w < (tid >> 6))
                    pre += wsum[w];
                tot += wsum[w];
            }
            if (c < nc) {
                lstart[c] = pre + incl - v;
                gbase[c] = slice - (pre + incl - v); // slice start minus the run's place in the stage: the write-out adds t
            }
            carry += tot;
            __syncthreads();
        }
        const uint32_t nitems = carry;
#pragma unroll
        for (int u = 0; u < SPT; ++u)
            if (in >> u & 1u) {
                const uint32_t i = tid + u * STAGE_THREADS, qi = s > 1 ? __umulhi(i, sinv) : i;
                uint32_t occ = 0; // equal values before this one in the same (ascending) sketch
                if (dup >> u & 1u) {
                    const uint32_t e = i - qi * s;
                    occ = 1;
                    while (occ < e && base[i - occ - 1] == v[u])
                        ++occ;
                }
                stage[stage_swz(lstart[v[u] >> cshift] + rk[u])] = make_uint2(v[u], (id_base + (uint32_t)(q0 + qi)) | (occ << id_bits));
            }
        __syncthreads();
        for (uint32_t t = tid; t < nitems; t += STAGE_THREADS) {
            const uint2 it = stage[stage_swz(t)];
            const uint32_t c = it.x >> cshift;
            citems[gbase[c] + t] = it; // (values and ids in two arrays halve level 2's counting pass but cost
                                                      // level 1 more than that: 0.64 -> 0.87 ms, two half-length runs per bucket)
        }
    }
}

// level 2: one workgroup per coarse bucket -> fine start[] + final item order (+ self-join size)
#ifndef PH_K2_FINE_THREADS
#define PH_K2_FINE_THREADS 512
#endif
constexpr int FINE_THREADS = PH_K2_FINE_THREADS; // a coarse bucket is a chain of memory round trips: 8 waves per SIMD hide more of them
// SEG (the item exchange of a device list, round 4): a coarse bucket arrives as `nseg` pieces, one per device that ran
// level 1 on its own sketches -- piece g of bucket c = segbase[g][seglo[g * segld + (c - cfirst)] ..
// seglo[g * segld + (c - cfirst) + 1]) -- and leaves at cstart[c] of the WHOLE set's coarse offsets, as ever.
template <bool SEG>
__global__ __launch_bounds__(FINE_THREADS, 8) void fine_kernel(const uint2 *__restrict__ citems,
                                                           const uint32_t *__restrict__ cstart, uint32_t cfirst, uint32_t nc,
                                                           uint32_t fpc_log2, uint32_t *__restrict__ hdr,
                                                           uint32_t *__restrict__ start, uint2 *__restrict__ items,
                                                           uint32_t id_bits, uint32_t ndw, uint32_t field_bits,
                                                           const uint2 *const *__restrict__ segbase,
                                                           const uint32_t *__restrict__ seglo, uint32_t nseg, uint32_t segld)
{
    if (hdr[H_B4] == 1u)
        return; // the sliced build's level 2 (fine4_kernel) makes this index
    constexpr int T = FINE_THREADS;
    __shared__ uint32_t cnt[FPC_MAX];
    __shared__ uint32_t ws[T / 64];
    const uint32_t fpc = 1u << fpc_log2;
    const uint32_t shift = hdr[H_SHIFT];
    const bool compact = hdr[H_FMT] != 0u;
    uint32_t *items32 = reinterpret_cast<uint32_t *>(items);
    const uint32_t id_mask = (1u << id_bits) - 1u, low_mask = (1u << shift) - 1u, kmul = (uint32_t)(((1ull << 32) + ndw - 1) / ndw);
    const int tid = threadIdx.x;
    unsigned long long sq = 0;
    for (uint32_t c = cfirst + blockIdx.x; c < nc; c += gridDim.x) { // coarse buckets [cfirst, nc)
        const uint32_t lo = cstart[c];
        const uint32_t npieces = SEG ? nseg : 1u;
        __syncthreads();
        for (uint32_t f = tid; f < fpc; f += T)
            cnt[f] = 0;
        __syncthreads();
        // eight loads in flight per thread: the loop is a chain of memory round trips otherwise (96 % of the wave
        // cycles were s_waitcnt, profiles/r02_k2_pmc_sq.md)
        for (uint32_t g = 0; g < npieces; ++g) {
            const uint2 *__restrict__ src = SEG ? segbase[g] : citems;
            const uint32_t plo = SEG ? seglo[(size_t)g * segld + (c - cfirst)] : lo;
            const uint32_t phi = SEG ? seglo[(size_t)g * segld + (c - cfirst) + 1u] : cstart[c + 1];
            for (uint32_t t0 = plo + tid; t0 < phi; t0 += 8 * T) {
                uint32_t x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    x[u] = t0 + u * T < phi ? src[t0 + u * T].x : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (t0 + u * T < phi)
                        atomicAdd(&cnt[(x[u] >> shift) & (fpc - 1u)], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of cnt[0..fpc): PER consecutive entries per thread
        constexpr int PERMAX = (int)(FPC_MAX / T);
        const uint32_t per = (fpc + T - 1) / T; // <= PERMAX
        uint32_t v[PERMAX], sum = 0;
#pragma unroll
        for (int i = 0; i < PERMAX; ++i) {
            const uint32_t f = tid * per + i;
            v[i] = ((uint32_t)i < per && f < fpc) ? cnt[f] : 0u;
            sum += v[i];
            sq += (unsigned long long)v[i] * v[i];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d)
                incl += t;
        }
        if ((tid & 63) == 63)
            ws[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = lo + incl - sum;
        for (int w = 0; w < (tid >> 6); ++w)
            run += ws[w];
#pragma unroll
        for (int i = 0; i < PERMAX; ++i) {
            const uint32_t f = tid * per + i;
            if ((uint32_t)i < per && f < fpc) {
                start[((size_t)c << fpc_log2) + f] = run;
                cnt[f] = run; // becomes the write cursor of fine bucket f
            }
            run += v[i];
        }
        __syncthreads();
        for (uint32_t g = 0; g < npieces; ++g) {
        const uint2 *__restrict__ src = SEG ? segbase[g] : citems;
        const uint32_t plo = SEG ? seglo[(size_t)g * segld + (c - cfirst)] : lo;
        const uint32_t hi = SEG ? seglo[(size_t)g * segld + (c - cfirst) + 1u] : cstart[c + 1];
        for (uint32_t t0 = plo + tid; t0 < hi; t0 += 8 * T) {
            uint2 it[8];
            uint32_t at[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                it[u] = t0 + u * T < hi ? src[t0 + u * T] : make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                at[u] = t0 + u * T < hi ? atomicAdd(&cnt[(it[u].x >> shift) & (fpc - 1u)], 1u) : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u * T < hi) {
                    if (compact) {
                        const uint32_t col = it[u].y & id_mask, occ = it[u].y >> id_bits;
                        const uint32_t k = __umulhi(col, kmul); // col / ndw (exact: see rowjoin_dense_kernel)
                        // occurrence number + 1: an all-zero word is then no item at all -- what a buffer load returns
                        // beyond the end of a bucket, so the join needs no bounds test
                        const uint32_t occ1 = (occ + 1u) << CK_LOW;
                        const uint32_t hi_part = shift ? (((it[u].x & low_mask) << (32u - shift)) | occ1) : occ1;
                        items32[at[u]] = hi_part | ((col - k * ndw) << 5) | (k * field_bits);
                    } else {
                        items[at[u]] = it[u];
                    }
                }
        }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        sq += __shfl_xor(sq, d, 64);
    if ((tid & 63) == 0 && sq)
        atomicAdd(reinterpret_cast<unsigned long long *>(&hdr[H_EST_LO]), sq);
}

// ---- the sliced build on 4-byte intermediate items (B4; the plan is at struct Layout) ---------------------------------------
// geometry from the largest value: one thread, between maxlast_kernel and the check pass
__global__ void plan4_kernel(uint32_t *__restrict__ hdr, uint32_t nbk_log2, uint32_t s, uint32_t slice_len, uint32_t cpp_max)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    const uint32_t maxval = hdr[H_MAXVAL];
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(maxval | 1u);
    const uint32_t shift = bucket_shift(maxval, nbk_log2);
    // a coarse bucket is value >> 16: at most B4_NC_MAX of them, at most 2^12 fine buckets in one (level 2's LDS histogram),
    // and the shift compact items allow
    if (bits < 17u || bits > 30u || shift < 4u || shift > 10u)
        return; // H_B4 stays 0: the two-level build
    const uint32_t nce = (maxval >> 16) + 1u; // coarse buckets that can hold an item
    // parts of the value range: a sketch's slice of one part is ~slice_len hashes when the sketch spreads over the whole range
    uint32_t R = min(max((s + slice_len - 1u) / slice_len, 1u), B4_RMAX);
    uint32_t cpp = min(max((nce + R - 1u) / R, 2u), cpp_max); // (cpp_max: what level 1's stage shape keeps in LDS, 512 or 1024)
    R = (nce + cpp - 1u) / cpp; // <= 32 when cpp was capped (nce <= 16,384), else <= the first R
    hdr[H_B4_NC] = 1u << (bits - 16u);
    hdr[H_B4_R] = R;
    hdr[H_B4_CPP] = cpp;
    hdr[H_B4_MAGIC] = (uint32_t)(((1ull << 32) + cpp - 1u) / cpp); // coarse / cpp = umulhi(coarse, magic): exact for coarse < 2^16
    hdr[H_B4] = 1u;
}

__device__ __forceinline__ uint32_t b4_part(uint32_t v, uint32_t magic, uint32_t R) { return min(__umulhi(v >> 16, magic), R - 1u); }

// check pass of the sliced build: check_kernel<true>'s wave-per-sketch test (ascending?  a hash repeated more often than an
// item can number?) + the histogram over value >> 16 (per group of 65,536 sketches: the id bit an intermediate item does not
// carry) + where the sketch crosses from one part of the value range into the next (pos[q][r] = its first element of part r,
// pos[q][R] = s; a part without an element of the sketch begins where the next one does) + the list of repeated hashes.
// A workgroup takes a contiguous run of sketches; its counters (up to 64 KB of LDS) are flushed once per group.
__global__ __launch_bounds__(1024) void check4_kernel(const uint32_t *__restrict__ sk, uint64_t n, uint32_t s, uint8_t *__restrict__ flags,
                                                     uint32_t *__restrict__ hdr, int force_irregular, uint32_t max_occ, uint32_t G,
                                                     uint32_t *__restrict__ g4count, uint32_t *__restrict__ dupmap,
                                                     uint2 *__restrict__ duplist, uint16_t *__restrict__ pos)
{
    extern __shared__ uint32_t lh[]; // H_B4_NC counters
    if (hdr[H_B4] != 1u)
        return;
    const uint32_t nc = hdr[H_B4_NC], R = hdr[H_B4_R], magic = hdr[H_B4_MAGIC];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x, qa = (uint64_t)blockIdx.x * per, qb = min(n, qa + per);
    uint32_t multmax = 0;
    for (uint32_t g = 0; g < G; ++g) {
        const uint64_t lo = max(qa, (uint64_t)g << 16), hi = g + 1u < G ? min(qb, (uint64_t)(g + 1u) << 16) : qb;
        if (lo >= hi)
            continue;
        for (uint32_t c = threadIdx.x; c < nc; c += 1024)
            lh[c] = 0;
        __syncthreads();
        for (uint64_t q = lo + wv; q < hi; q += 16) {
            const uint32_t *p = sk + q * s;
            uint32_t x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t e = (uint32_t)u * 64u + lane;
                x[u] = e < s ? p[e] : 0u;
            }
            bool bad = force_irregular != 0, anyeq = false;
            auto next_of = [&](int u) -> uint32_t { // the element behind x[u]'s (valid while it is not the sketch's last)
                const uint32_t nv = (uint32_t)__shfl_down((int)x[u], 1, 64);
                const uint32_t first_of_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)x[u < 15 ? u + 1 : 15]);
                return lane == 63u ? first_of_next : nv;
            };
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t e = (uint32_t)u * 64u + lane;
                const uint32_t nv = next_of(u);
                const bool has = e + 1u < s;
                bad = bad || (has && x[u] > nv);
                anyeq = anyeq || (has && x[u] == nv);
            }
            bool anybad = __ballot(bad) != 0ull; // an irregular sketch never enters the index
            uint32_t mult = 1;
            if (!anybad && __ballot(anyeq) != 0ull) {
                // repeated values inside an ascending sketch (rare): the longest run; every copy after the first is logged
                mult = 0;
                for (uint32_t e = lane; e < s; e += 64) {
                    const uint32_t x0 = p[e];
                    if (e == 0 || p[e - 1] != x0) {
                        uint32_t a = 1;
                        while (e + a < s && p[e + a] == x0)
                            ++a;
                        mult = max(mult, a);
                        if (a > 1u) {
                            atomicOr(&dupmap[x0 >> 21], 1u << ((x0 >> 16) & 31u));
                            const uint32_t at = atomicAdd(&hdr[H_B4_NDUP], a - 1u);
                            if (at + (a - 1u) <= B4_DUP_MAX)
                                for (uint32_t k = 1; k < a; ++k)
                                    duplist[at + k - 1u] = make_uint2(x0, (uint32_t)q | (k << 17));
                        }
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
                    mult = max(mult, (uint32_t)__shfl_xor((int)mult, d, 64));
                if (mult - 1u > max_occ)
                    anybad = true;
            }
            if (anybad) {
                if (lane == 0)
                    flags[q] = 1;
                continue;
            }
            multmax = max(multmax, mult);
            uint16_t *pq = pos + q * (R + 1u);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t e = (uint32_t)u * 64u + lane;
                const uint32_t nv = next_of(u); // (by every lane: a cross-lane read of a lane that sits out a branch returns 0)
                // (working a part out once per element and passing it to the lane below by DPP was measured: 0.241 against
                // 0.211 ms -- the readfirstlane for lane 63 waits where the two multiplies did not)
                if (e < s) {
#ifndef PH_K2_C4_NOHIST // ablation probes (a wrong index): the check pass without its histogram / without its part bounds
                    atomicAdd(&lh[x[u] >> 16], 1u);
#endif
#ifdef PH_K2_C4_NOPOS
                    if (x[u] == 0x12345u)
                        pq[0] = 1;
                    continue;
#endif
                    const uint32_t ra = b4_part(x[u], magic, R);
                    if (e == 0u)
                        for (uint32_t r = 0; r <= ra; ++r)
                            pq[r] = 0;
                    if (e + 1u < s) {
                        const uint32_t rb = b4_part(nv, magic, R);
                        for (uint32_t r = ra + 1u; r <= rb; ++r)
                            pq[r] = (uint16_t)(e + 1u);
                    } else {
                        for (uint32_t r = ra + 1u; r <= R; ++r)
                            pq[r] = (uint16_t)s;
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < nc; c += 1024)
            if (lh[c])
                atomicAdd(&g4count[c * G + g], lh[c]);
        __syncthreads();
    }
    if (lane == 0 && multmax > __hip_atomic_load(&hdr[H_MAXMULT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&hdr[H_MAXMULT], multmax);
}

// exclusive scan of g4count[NC * G] -> c4start[NC * G + 1], g4cur = copy; one workgroup, a thread's (up to 32) consecutive
// counters in registers from ONE round of 16-byte loads (counter by counter, twice over, the kernel took 33 us)
__global__ __launch_bounds__(1024) void scan4_kernel(const uint32_t *__restrict__ g4count, uint32_t G, const uint32_t *__restrict__ hdr,
                                                    uint32_t *__restrict__ c4start, uint32_t *__restrict__ g4cur,
                                                    uint32_t *__restrict__ start, uint32_t nbk)
{
    __shared__ uint32_t wsum[16];
    if (hdr[H_B4] != 1u)
        return;
    const uint32_t m = hdr[H_B4_NC] * G; // <= 32,768 = the array's size: whole 16-byte pieces can be read, what lies behind m is zero
    const uint32_t per = (((m + 1023u) / 1024u) + 3u) & ~3u, tid = threadIdx.x, i0 = tid * per;
    uint32_t v[32], sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < 32; i += 4) {
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (i < per && i0 + i < m)
            x = *reinterpret_cast<const uint4 *>(g4count + i0 + i);
        v[i] = x.x, v[i + 1] = x.y, v[i + 2] = x.z, v[i + 3] = x.w;
        sum += x.x + x.y + x.z + x.w;
    }
    const uint32_t incl = dpp_incl_scan(sum);
    if ((tid & 63u) == 63u)
        wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum, total = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < (tid >> 6))
            run += wsum[w];
        total += wsum[w];
    }
#pragma unroll
    for (uint32_t i = 0; i < 32; ++i) {
        if (i < per && i0 + i < m) {
            c4start[i0 + i] = run;
            g4cur[i0 + i] = run;
        }
        run += v[i];
    }
    if (tid == 0) {
        c4start[m] = total;
        start[nbk] = total;
    }
}

// level 1 of the sliced build.  Work item = (batch of B sketches, part r of the value range): the batch's B slices
// pos[q][r] .. pos[q][r + 1], SLOTS hashes of each per round (a longer slice takes further rounds), ordered by coarse bucket
// in the LDS stage -- count with the atomic's return value as the rank, scan, one slice of every bucket from the global
// cursor, placement -- and written out as runs, a quarter wave per coarse bucket (a stage item does not say which bucket it
// belongs to, so the write-out goes bucket by bucket, not item by item).  The bounds of the NEXT work item are fetched while
// this one is worked on: they are a round trip that nothing else of a work item can start without.
template <int B, int SLOTS>
__global__ __launch_bounds__(1024, 8) void scatter4_kernel(const uint32_t *__restrict__ sk, uint32_t n, uint32_t s,
                                                          const uint8_t *__restrict__ flags, const uint16_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ hdr, uint32_t G, uint32_t *__restrict__ g4cur,
                                                          uint32_t *__restrict__ citems4)
{
    static_assert(B * SLOTS == (int)B4_STAGE && (B & (B - 1)) == 0 && B >= 64 && 65536 % B == 0, "a batch fills the stage and lies in one id group");
    constexpr uint32_t T = 1024, SPT = B4_STAGE / T, QSTEP = T / SLOTS; // slot i = tid + u * T: sketch (tid / SLOTS) + u * QSTEP of the batch
    extern __shared__ __attribute__((aligned(16))) uint32_t lds4[];
    constexpr uint32_t CPPM = b4_cpp_max(B);        // coarse buckets per part at most (plan4_kernel was told)
    uint32_t *stage = lds4;                         // B4_STAGE (+ 64 words the empty slots' items go to)
    uint32_t *cnt = stage + B4_STAGE + 64;          // CPPM (+ 64 words the empty slots count into)
    uint32_t *lstart = cnt + CPPM + 64;             // CPPM + 1 (+ padding)
    uint32_t *gbase = lstart + CPPM + 16;           // CPPM
    uint32_t *spl = gbase + CPPM;                   // 2 x B: start | length << 16 of this and of the next work item's slices
    __shared__ uint32_t wsum[T / 64];
    if (hdr[H_B4] != 1u)
        return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t R = hdr[H_B4_R], cpp = hdr[H_B4_CPP];
    // Work items in an XCD-aware order.  Workgroup b runs on XCD b % 8 (round-robin dispatch), and an XCD has its own L2: the
    // batches are dealt to the XCDs (batch % 8), and an XCD's workgroups take its batches' work items in order -- so the
    // parts of ONE batch, whose slices share the cache lines at their ends, are read through one L2 at about the same time
    // (dealt out w = blockIdx.x, + gridDim.x, ... the kernel fetched 1.87 x the sketches' bytes: profiles/r05a_k2_fetch.md).
    // The XCD's workgroups take the list ROUND ROBIN: neighbours run at the same time, and the second request for a line meets
    // the first in the L2.  (A contiguous run of the list per workgroup -- neighbouring parts one after the other on one CU --
    // was measured too and fetched MORE, 0.89 against 0.70 GB at 64 slots: at full rate an XCD's 4 MB of L2 turn over in a
    // few microseconds, a work item takes ~40; profiles/r05c_k2_contiguous_runs.log.)
    const uint32_t xcd = blockIdx.x & 7u, wg_x = gridDim.x >> 3, nbat = (n + B - 1) / B;
    const uint32_t nwork = ((nbat + 7u - xcd) >> 3) * R; // this XCD's work items (<= 1024 batches x 64 parts)
    auto fetch = [&](uint32_t w) -> uint32_t { // slice r of sketch batch * B + tid: start | length << 16 (0: none)
        const uint32_t bx = w / R, r = w - bx * R, batch = xcd + 8u * bx, q = batch * B + tid;
        if (q >= n || flags[q])
            return 0u;
        const uint16_t *pq = pos + q * (R + 1u) + r;
        const uint32_t a = pq[0], b = pq[1];
        if (b < a || b > s)
            return 0u; // (never for a row check4_kernel wrote: a guard against thousands of rounds on a corrupted table)
        return a | ((b - a) << 16);
    };
    uint32_t w = blockIdx.x >> 3;
    uint32_t nxt = (tid < (uint32_t)B && w < nwork) ? fetch(w) : 0u;
    uint32_t cur = 0;
    const uint32_t qt = tid / SLOTS, et = tid % SLOTS;
    for (; w < nwork; w += wg_x, cur ^= (uint32_t)B) {
        if (tid < (uint32_t)B)
            spl[cur + tid] = nxt;
        __syncthreads(); // the bounds are there; everybody is through with the previous work item's stage
        uint32_t maxlen = 0;
#pragma unroll
        for (int j = 0; j < B / 64; ++j)
            maxlen = max(maxlen, spl[cur + j * 64 + lane] >> 16);
        maxlen = dpp_wave_max(maxlen);
        {
            const uint32_t w2 = w + wg_x;
            nxt = (tid < (uint32_t)B && w2 < nwork) ? fetch(w2) : 0u;
        }
        const uint32_t bx = w / R, r = w - bx * R, q0 = (xcd + 8u * bx) * B, cb = r * cpp, g = q0 >> 16;
        // the batch's sketches as ONE buffer: a slot without a hash asks beyond its end and reads 0 -- no branch around a load
        const uint32_t nb = min((uint32_t)B, n - q0);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(sk + (size_t)q0 * s), 0, (int)(nb * s * 4u), 0x00020000);
        for (uint32_t e0 = 0; e0 < maxlen; e0 += SLOTS) { // one round unless a slice is longer than its slots
            if (tid < cpp)
                cnt[tid] = 0;
            __syncthreads();
            uint32_t v[SPT], in = 0;
            const uint32_t e = et + e0, at0 = qt * s + e;
#pragma unroll
            for (uint32_t u = 0; u < SPT; ++u) {
                // (the sketch's offset inside the batch rides in the load's scalar offset: sixteen per-slot vector offsets,
                // loop invariants all, are what the compiler would otherwise hoist out of the loops and spill)
                const uint32_t plv = spl[cur + qt + u * QSTEP];
                const bool ok = e < (plv >> 16);
                v[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (int)((at0 + (plv & 0xFFFFu)) * 4u) : -1,
                                                                      (int)(u * QSTEP * s * 4u), 0);
                in |= (ok ? 1u : 0u) << u;
            }
            // count (a slot without a hash counts into a word of its own lane behind the counters: no branch per slot).
            // v becomes [bucket inside the part : 16 | value & 0xFFFF : 16] -- all the placement needs
#pragma unroll
            for (uint32_t u = 0; u < SPT; ++u) {
                v[u] -= cb << 16;
                __hip_atomic_fetch_add(&cnt[(in >> u & 1u) ? v[u] >> 16 : CPPM + lane], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (uint32_t u = 0; u < SPT; ++u)
                asm volatile("" : "+v"(v[u])); // (the placement works its addresses out again: sixteen registers, not thirty-two)
            __syncthreads();
            // exclusive scan of cnt[0..cpp) -> lstart, and back into cnt as the buckets' cursors; my slice of every coarse
            // bucket -> gbase (the atomic's round trip runs under the scan; its result is not needed before the write-out)
            const uint32_t vc = tid < cpp ? cnt[tid] : 0u;
            const uint32_t slice = vc ? atomicAdd(&g4cur[(cb + tid) * G + g], vc) : 0u;
            const uint32_t incl = dpp_incl_scan(vc);
            if (lane == 63u)
                wsum[wv] = incl;
            __syncthreads();
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (uint32_t ww = 0; ww < T / 64; ++ww) {
                const uint32_t x = wsum[ww];
                pre += ww < wv ? x : 0u;
                tot += x;
            }
            if (tid < cpp) {
                lstart[tid] = pre + incl - vc;
                cnt[tid] = pre + incl - vc;
                gbase[tid] = slice;
            }
            if (tid == 0)
                lstart[cpp] = tot;
            __syncthreads();
            // placement: a second atomic on the bucket's cursor gives the item's slot -- as many LDS operations as reading the
            // bucket's start and adding a rank kept since the count, and sixteen registers less (the kernel lives on 64)
            uint32_t idb = (q0 + qt) & 0xFFFFu;
            asm volatile("" : "+v"(idb)); // (not sixteen hoisted id registers either)
#pragma unroll
            for (uint32_t u0 = 0; u0 < SPT; u0 += 8) { // eight atomics in flight, then their eight stores; no branch per slot:
                uint32_t at[8];                         // an empty slot's item goes to a word of its lane behind the stage
#pragma unroll
                for (uint32_t u = u0; u < u0 + 8; ++u)
                    at[u - u0] = atomicAdd(&cnt[(in >> u & 1u) ? v[u] >> 16 : CPPM + lane], 1u);
#pragma unroll
                for (uint32_t u = u0; u < u0 + 8; ++u)
                    stage[(in >> u & 1u) ? at[u - u0] : B4_STAGE + lane] = ((v[u] << 16) | idb) + u * QSTEP;
            }
            __syncthreads();
            // write-out: a quarter wave per coarse bucket (runs are a few dozen items)
            for (uint32_t c = tid >> 4; c < cpp; c += T / 16) {
                const uint32_t ls = lstart[c], nn = lstart[c + 1u] - ls, gb = gbase[c];
                for (uint32_t j = tid & 15u; j < nn; j += 16)
                    citems4[gb + j] = stage[ls + j];
            }
        }
    }
}

// level 2 of the sliced build: a workgroup per coarse bucket (value >> 16).  Up to CAP x T intermediate items sit in the
// threads' registers -- ONE read of them; the fine histogram (and, as the atomic's return value, every item's rank inside its
// fine bucket) in LDS; the final compact items written by the bucket's only owner.  A larger coarse bucket (a skewed input)
// takes two passes over its intermediate items.  Then, in a bucket the check pass marked, the repeated hashes get their
// occurrence numbers (see struct Layout).
// One more of counter hist[key], returning its value before: the rank of this lane's item inside its fine bucket (or, on a
// cursor, its slot).  The copies of one hash that a family of related sketches holds arrive TOGETHER -- level 1 wrote them as
// one run -- so the 64 lanes of a wave mostly hold one or two keys, and 64 atomics on one LDS address serialise (the first
// version of fine4_kernel spent 21 us per coarse bucket that way: profiles/r05a_k2_stats.md).  So the key of the wave's first
// lane is counted ONCE, by that lane, for all the lanes that hold it (ballot, popcount, the lane's rank among them from
// mbcnt), then the first remaining lane's key likewise; lanes left after two rounds -- distinct keys -- take the plain atomic.
__device__ __forceinline__ uint32_t ranked_add(uint32_t *hist, uint32_t key, bool valid)
{
    const uint32_t lane = __lane_id();
    unsigned long long rem = __ballot(valid);
    uint32_t rk = 0;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (rem == 0ull)
            break;
        const int src = __builtin_ctzll(rem);
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, src);
        const unsigned long long m = __ballot(valid && key == k) & rem;
        const uint32_t c = (uint32_t)__builtin_popcountll(m);
        if (c < 4u)
            break; // (a handful of equal keys costs less as plain atomics than as a round of this)
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t base = 0;
        if (lane == (uint32_t)src)
            base = atomicAdd(&hist[k], c);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, src);
        if (m >> lane & 1ull)
            rk = base + below;
        rem &= ~m;
    }
    if (rem >> lane & 1ull)
        rk = atomicAdd(&hist[key], 1u);
    return rk;
}

template <int T, int CAP>
__global__ __launch_bounds__(T, 4) void fine4_kernel(const uint32_t *__restrict__ citems4, const uint32_t *__restrict__ c4start, uint32_t G,
                                                 uint32_t *__restrict__ hdr, uint32_t *__restrict__ start,
                                                 uint32_t *__restrict__ items32, const uint32_t *__restrict__ dupmap,
                                                 const uint2 *__restrict__ duplist, uint32_t ndw, uint32_t field_bits)
{
    __shared__ uint32_t hist[4096];
    __shared__ uint32_t ws[T / 64];
    if (hdr[H_B4] != 1u)
        return;
    const uint32_t nc = hdr[H_B4_NC], shift = hdr[H_SHIFT], fpc_log2 = 16u - shift, fpc = 1u << fpc_log2;
    const uint32_t low_mask = (1u << shift) - 1u, kmul = (uint32_t)(((1ull << 32) + ndw - 1) / ndw);
    const uint32_t occ_mask = ((1u << (11u - shift)) - 1u) << CK_LOW;
    const uint32_t tid = threadIdx.x;
    unsigned long long sq = 0;
    uint32_t nover = 0;
    for (uint32_t c = blockIdx.x; c < nc; c += gridDim.x) {
        const uint32_t lo = c4start[c * G], mid = c4start[c * G + G - 1u], cntc = c4start[c * G + G] - lo;
        // the compact item of intermediate item `item` at position p of the bucket: as fine_kernel's, occurrence number 0
        auto compact_of = [&](uint32_t item, uint32_t col) -> uint32_t {
            const uint32_t k = __umulhi(col, kmul); // col / ndw (exact: see rowjoin_dense_kernel)
            return (((item >> 16) & low_mask) << (32u - shift)) | (1u << CK_LOW) | ((col - k * ndw) << 5) | (k * field_bits);
        };
        __syncthreads();
        for (uint32_t f = tid; f < fpc; f += T)
            hist[f] = 0;
        __syncthreads();
        const bool inreg = cntc <= (uint32_t)CAP * T; // (uniform)
        static_assert(CAP % 8 == 0, "items are loaded, and placed, eight per thread at a time");
        // The coarse bucket as a buffer, for loads and for stores: an item beyond its end reads as 0 and a store beyond it is
        // dropped, so a thread's loads go out back to back with no branch between them.  (With `p < cntc ? load : 0` the
        // compiler put every load behind a branch of its own and waited for each before ranking the item in front of it:
        // fifteen HBM round trips per bucket, 19 us of them -- profiles/r05b_k2_stats_slots64.md, fine4 0.50 ms.)
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(citems4 + lo), 0, (int)(cntc * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(items32 + lo, 0, (int)(cntc * 4u), 0x00020000);
        const uint32_t hasdup = dupmap[c >> 5] >> (c & 31u) & 1u; // (asked for now, looked at behind the placement)
        uint32_t it[CAP], rk2[CAP / 2]; // (an item's rank is below 2^16 here: two per register, the kernel lives on 128)
        // Which item a lane takes.  A family's copies of one hash arrive TOGETHER (level 1 wrote them as one run), and 64
        // atomics on ONE LDS address serialise: with lane l on item l the rank phase was half the kernel (ablation,
        // profiles/r05d_f4_ablation.log: 0.24 of 0.465 ms), and counting a wave's common key once (ballot, popcount, mbcnt:
        // ranked_add) cost as many instructions as it saved cycles.  So a wave's lanes take eight 32-byte pieces that lie
        // T / 8 items apart: eight-way conflicts at worst, loads still whole sectors, no instruction more per item.
#ifndef PH_K2_F4_PIECE_LOG2
#define PH_K2_F4_PIECE_LOG2 3
#endif
        constexpr uint32_t PL = PH_K2_F4_PIECE_LOG2, PPW = 64u >> PL; // lanes per piece (log2), pieces per wave
        const uint32_t ptid = ((((tid >> PL) & (PPW - 1u)) * (T / 64) + (tid >> 6)) << PL) | (tid & ((1u << PL) - 1u));
        if (inreg) {
#pragma unroll
            for (int u = 0; u < CAP; ++u) // (all of them, whatever the bucket's size: a load beyond the end costs an issue slot)
                it[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (int)(ptid * 4u), (int)((uint32_t)u * T * 4u), 0);
#pragma unroll
            for (int u = 0; u < CAP; ++u) {
                if ((uint32_t)u * T >= cntc)
                    break; // (uniform)
#if defined(PH_K2_F4_NORANK)
                const uint32_t r = tid & 7u;
#elif defined(PH_K2_F4_RANKED)
                const uint32_t r = ranked_add(hist, it[u] >> (16u + shift), ptid + (uint32_t)u * T < cntc);
#else
                const uint32_t r = ptid + (uint32_t)u * T < cntc ? atomicAdd(&hist[it[u] >> (16u + shift)], 1u) : 0u;
#endif
                rk2[u / 2] = (u & 1) ? rk2[u / 2] | (r << 16) : r;
            }
        } else {
            ++nover;
            for (uint32_t p0 = 0; p0 < cntc; p0 += T) {
                const bool ok = p0 + tid < cntc;
                (void)ranked_add(hist, ok ? citems4[lo + p0 + tid] >> (16u + shift) : 0u, ok);
            }
        }
        __syncthreads();
        // exclusive scan of hist[0..fpc): PER consecutive entries per thread
        constexpr int PERMAX = (4096 + T - 1) / T;
        const uint32_t per = (fpc + T - 1) / T;
        uint32_t hv[PERMAX], sum = 0;
#pragma unroll
        for (int i = 0; i < PERMAX; ++i) {
            const uint32_t f = tid * per + i;
            hv[i] = ((uint32_t)i < per && f < fpc) ? hist[f] : 0u;
            sum += hv[i];
            sq += (unsigned long long)hv[i] * hv[i];
        }
        const uint32_t incl = dpp_incl_scan(sum);
        if ((tid & 63u) == 63u)
            ws[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = lo + incl - sum;
        for (uint32_t w = 0; w < (tid >> 6); ++w)
            run += ws[w];
#pragma unroll
        for (int i = 0; i < PERMAX; ++i) {
            const uint32_t f = tid * per + i;
            if ((uint32_t)i < per && f < fpc) {
                start[((size_t)c << fpc_log2) + f] = run;
                hist[f] = run - lo; // the fine bucket's first position inside the coarse bucket (two-pass path: its write cursor)
            }
            run += hv[i];
        }
        __syncthreads();
        if (inreg) {
#pragma unroll
            for (int u0 = 0; u0 < CAP; u0 += 8) {
                if ((uint32_t)u0 * T >= cntc)
                    break; // (uniform)
                uint32_t at[8];
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) // eight LDS reads in flight (an item beyond the end reads hist[0]: harmless)
                    at[u - u0] = hist[it[u] >> (16u + shift)] + ((u & 1) ? rk2[u / 2] >> 16 : rk2[u / 2] & 0xFFFFu);
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) {
                    const uint32_t p = ptid + (uint32_t)u * T;
#if defined(PH_K2_F4_NOSTORE)
                    if (at[u - u0] == 0xFFFFFFFFu)
#elif defined(PH_K2_F4_LINSTORE)
                    at[u - u0] = p;
#endif
                    __builtin_amdgcn_raw_buffer_store_b32(
                        compact_of(it[u], (it[u] & 0xFFFFu) | ((G == 2u && lo + p >= mid) ? 65536u : 0u)), rout,
                        p < cntc ? (int)(at[u - u0] * 4u) : -1, 0, 0);
                }
            }
        } else {
            for (uint32_t p0 = 0; p0 < cntc; p0 += T) {
                const uint32_t p = p0 + tid;
                const bool ok = p < cntc;
                const uint32_t item = ok ? citems4[lo + p] : 0u;
                const uint32_t at = ranked_add(hist, item >> (16u + shift), ok);
                if (ok)
                    items32[lo + at] = compact_of(item, (item & 0xFFFFu) | ((G == 2u && lo + p >= mid) ? 65536u : 0u));
            }
        }
        if (hasdup) {
            // repeated hashes of this coarse bucket: record (value, id | number << 17) -> the number-th (0-based) of the equal
            // items of its fine bucket, in slot order, gets the number (the first copy keeps 0: it is not in the list)
            __threadfence();
            __syncthreads();
            const uint32_t ndup = min(hdr[H_B4_NDUP], B4_DUP_MAX);
            for (uint32_t j = tid; j < ndup; j += T) {
                const uint2 rec = duplist[j];
                if ((rec.x >> 16) != c)
                    continue;
                const uint32_t f = (rec.x & 0xFFFFu) >> shift, number = rec.y >> 17;
                const size_t fb = ((size_t)c << fpc_log2) + f;
                const uint32_t b0 = __hip_atomic_load(&start[fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t b1 = f + 1u < fpc ? __hip_atomic_load(&start[fb + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lo + cntc;
                const uint32_t want = compact_of(rec.x << 16, rec.y & 0x1FFFFu);
                uint32_t seen = 0;
                for (uint32_t at = b0; at < b1; ++at) {
                    const uint32_t got = __hip_atomic_load(&items32[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((got ^ want) & ~occ_mask) == 0u && seen++ == number) {
                        __hip_atomic_store(&items32[at], (want & ~occ_mask) | ((number + 1u) << CK_LOW), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        sq += __shfl_xor(sq, d, 64);
    if ((tid & 63u) == 0 && sq)
        atomicAdd(reinterpret_cast<unsigned long long *>(&hdr[H_EST_LO]), sq);
    if (nover && tid == 0)
        atomicAdd(&hdr[H_B4_OVER], nover);
}

// The join's cost model, in ONE place (round-4 advice: k2_exchange_index had its own copy of the formula).  Merging every
// pair costs nx * ny * (sx + sy) dependent steps; the join compares every X value with its whole bucket: about
// (nx * sx / (ny * sy)) * sum_b |Y_b|^2 compares when X is distributed like Y (exact for the all-vs-all).
struct JoinCost {
    double est_scale, generic_cost;
    __host__ __device__ bool merge_wins(unsigned long long self_join) const { return (double)self_join * est_scale > generic_cost; }
};
static inline JoinCost join_cost(uint64_t nx, uint32_t sx, uint64_t ny, uint32_t sy)
{
    return JoinCost{((double)nx * sx) / ((double)ny * sy), (double)nx * (double)ny * (double)(sx + sy) * 4.0};
}

// sparse / generic decision from the index's self-join size
__global__ void decide_kernel(uint32_t *__restrict__ hdr, JoinCost cost)
{
    const unsigned long long self = *reinterpret_cast<unsigned long long *>(&hdr[H_EST_LO]);
    if (cost.merge_wins(self))
        hdr[H_MODE] = MODE_GENERIC;
}

// ---- rowjoin: one workgroup per X row -------------------------------------------------
// counts one shared hash for column j in the row's LDS table; false = table full
__device__ __forceinline__ bool table_add(uint32_t *__restrict__ keys, uint32_t *__restrict__ cnts,
                                          uint32_t *__restrict__ nkeys, uint32_t j)
{
    static_assert(TAB == 2048, "hash shift below assumes 2^11 slots");
    uint32_t h = (j * 2654435761u) >> (32 - 11);
    const uint32_t key = j + 1u; // 0 = empty slot
    for (uint32_t probe = 0; probe < TAB; ++probe) {
        uint32_t k = keys[h];
        if (k == 0u) {
            k = atomicCAS(&keys[h], 0u, key);
            if (k == 0u) {
                atomicAdd(nkeys, 1u);
                k = key;
            }
        }
        if (k == key) {
            atomicAdd(&cnts[h], 1u);
            return true;
        }
        h = (h + 1u) & (TAB - 1u);
    }
    return false;
}

__global__ __launch_bounds__(THREADS) void rowjoin_kernel(const uint32_t *__restrict__ X, uint64_t nx, uint32_t sx,
                                                         const uint8_t *__restrict__ flagsX,
                                                         const uint32_t *__restrict__ start,
                                                         const uint2 *__restrict__ items, uint32_t nbk, uint32_t id_bits,
                                                         uint32_t *__restrict__ hdr, uint32_t *__restrict__ ovfX,
                                                         uint16_t *__restrict__ counts, uint64_t ld)
{
    if (hdr[H_MODE] != MODE_SPARSE)
        return;
    const uint32_t id_mask = (1u << id_bits) - 1u;
    // per distinct value d of the row: dval, dmul (multiplicity), dbeg/dend (its bucket in `items`)
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *xv = dyn, *dval = dyn + sx, *dmul = dyn + 2 * (size_t)sx, *dbeg = dyn + 3 * (size_t)sx,
             *dend = dyn + 4 * (size_t)sx;
    __shared__ uint32_t keys[TAB], cnts[TAB];
    __shared__ uint32_t nkeys, ndist;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t shift = hdr[H_SHIFT];

    for (uint64_t i = blockIdx.x; i < nx; i += gridDim.x) {
        if (flagsX[i])
            continue;
        __syncthreads(); // previous row flushed
        for (uint32_t h = tid; h < TAB; h += THREADS) {
            keys[h] = 0;
            cnts[h] = 0;
        }
        if (tid == 0) {
            nkeys = 0;
            ndist = 0;
        }
        const uint32_t *xp = X + i * sx;
        for (uint32_t p = tid; p < sx; p += THREADS)
            xv[p] = xp[p];
        __syncthreads();
        // distinct values of the row and their buckets (all the `start` loads in flight at once)
        for (uint32_t p = tid; p < sx; p += THREADS) {
            const uint32_t v = xv[p];
            if (p != 0 && xv[p - 1] == v)
                continue; // not the first copy
            const uint32_t b = v >> shift;
            if (b >= nbk)
                continue; // beyond every Y value: shares nothing
            const uint32_t bs = start[b], be = start[b + 1];
            if (be == bs)
                continue;
            uint32_t a = 1;
            while (p + a < sx && xv[p + a] == v)
                ++a;
            const uint32_t slot = atomicAdd(&ndist, 1u);
            dval[slot] = v;
            dmul[slot] = a;
            dbeg[slot] = bs;
            dend[slot] = be;
        }
        __syncthreads();
        const uint32_t nd = ndist;
        bool ok = true;
        // waves take JOIN_U distinct values at a time: their buckets' first 64 items are
        // loaded back to back, then consumed
        for (uint32_t d0 = wave * JOIN_U; d0 < nd && ok; d0 += (THREADS / 64) * JOIN_U) {
            uint2 it[JOIN_U];
#pragma unroll
            for (int u = 0; u < JOIN_U; ++u) {
                const uint32_t d = d0 + u;
                it[u] = make_uint2(0u, 0xFFFFFFFFu);
                if (d < nd && dbeg[d] + lane < dend[d])
                    it[u] = items[dbeg[d] + lane];
            }
#pragma unroll
            for (int u = 0; u < JOIN_U; ++u) {
                const uint32_t d = d0 + u;
                if (d >= nd)
                    break;
                const uint32_t v = dval[d], a = dmul[d];
                if (it[u].y != 0xFFFFFFFFu && it[u].x == v && (it[u].y >> id_bits) < a)
                    ok &= table_add(keys, cnts, &nkeys, it[u].y & id_mask);
                for (uint32_t t = dbeg[d] + 64 + lane; t < dend[d]; t += 64) { // rest of a long bucket
                    const uint2 r = items[t];
                    if (r.x == v && (r.y >> id_bits) < a)
                        ok &= table_add(keys, cnts, &nkeys, r.y & id_mask);
                }
            }
            if (nkeys > TAB_LIMIT)
                ok = false;
        }
        // a full table / too many distinct columns sends the whole row to the merge
        const int bad = __syncthreads_or((!ok || nkeys > TAB_LIMIT) ? 1 : 0);
        if (bad) {
            if (tid == 0)
                ovfX[atomicAdd(&hdr[H_NOVF], 1u)] = (uint32_t)i;
            continue;
        }
        for (uint32_t h = tid; h < TAB; h += THREADS)
            if (keys[h])
                counts[i * ld + (keys[h] - 1u)] = (uint16_t)cnts[h];
    }
}

constexpr int DENSE_THREADS = 1024; // one workgroup per CU (its LDS is the whole CU's): 16 waves keep the bucket loads coming
#ifndef PH_K2_DENSE_U
#define PH_K2_DENSE_U 8
#endif
#ifndef PH_K2_PIPE2
#define PH_K2_PIPE2 0 // 1: two groups of DENSE_U buckets in flight per wave (round 6; measured in profiles/r06_k2_join_pipe.log)
#endif
constexpr int DENSE_U = PH_K2_DENSE_U; // buckets a wave loads back to back, 128 items of each
static_assert(64 % DENSE_U == 0, "a chunk of 64 bucket descriptors (one per lane) is walked DENSE_U at a time by v_readlane: "
                                 "DENSE_U must divide 64 (12 read lanes 64..71 = lanes 0..7 again and counted buckets twice)");

// ---- dense join: a counter per COLUMN in LDS ------------------------------------------------------------
// Same bucket walk as rowjoin_kernel, but the accumulator is a dense array of BITS-bit counters in LDS, one per
// column of a stripe (PER = 32 / BITS per dword, bumped with one 32-bit LDS atomic: a count never exceeds the smaller
// SketchSize < 2^BITS, so the fields cannot carry into each other).  No hash probing, no zero-fill of the
// output (a stripe is flushed whole, zeros included: the 2 B per pair the matrix costs anyway), and the work
// follows the shared hashes.  With 10-bit counters (SketchSize <= 1023) the ~138 KB of LDS next to a 1000-hash
// row hold 105k columns: config 3's 100k sketches are ONE stripe, every bucket is read once per row.
// Rows: all regular ones (`rows` == NULL), or the rows the sparse join handed over (hdr[H_NOVF] of them).
// Barrier for LDS traffic only: the wave's LDS operations are complete, nothing is said about its global loads and
// stores.  rowjoin_dense_kernel shares nothing through global memory inside a workgroup, and __syncthreads() would
// also wait for the flush's stores (and any load issued ahead) to complete -- a memory round trip per row.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// REG (round 4; rows of at most DENSE_THREADS hashes): no LDS staging of the row at all.  Thread t keeps element t of the
// row -- its value, the values on either side, its bucket's bounds, all loaded a row ahead -- and IS the descriptor of that
// element: "first copy of its value in the row" is a compare with the element in front, the multiplicity is 1 unless the
// next element is equal (then a short scan of the row in global memory), and a wave walks the buckets of ITS OWN 64
// consecutive elements by v_readlane (an element that is not a first copy, or whose bucket is empty, is an empty bucket).
// What goes: the row's copy in LDS, the LDS atomic per distinct value, four descriptor arrays and the three barriers
// around them -- 5.5 of a row's 24 us at config 3 (profiles/r04_write_bw.md).  POLYHIP_K2_REGROW=0: the staged form.
template <int BITS, bool COMPACT, bool REG = false>
__global__ __launch_bounds__(DENSE_THREADS) void rowjoin_dense_kernel(const uint32_t *__restrict__ X, uint64_t nx, uint32_t sx,
                                                               const uint8_t *__restrict__ flagsX,
                                                               const uint32_t *__restrict__ start,
                                                               const void *__restrict__ items_v, uint32_t nbk,
                                                               const uint32_t *__restrict__ hdr,
                                                               const uint32_t *__restrict__ rows, uint64_t ny,
                                                               uint32_t stripe_dwords, uint32_t id_bits,
                                                               uint16_t *__restrict__ counts, uint64_t ld, int zero_ahead)
{
    constexpr uint32_t NWAVES = DENSE_THREADS / 64;
    if (hdr[H_MODE] != MODE_SPARSE)
        return;
    if ((hdr[H_FMT] != 0u) != COMPACT) // the index says which item format it holds; the other instantiation has nothing to do
        return;
    constexpr uint32_t PER = 32 / BITS, FMASK = (1u << BITS) - 1u;
    // 8-byte items (value, id | occurrence number) or compact 4-byte ones (H_FMT)
    typedef typename std::conditional<COMPACT, uint32_t, uint2>::type Item;
    const Item *__restrict__ items = static_cast<const Item *>(items_v);
    // the counters come FIRST: their LDS address is then the item's own byte offset plus a constant the instruction carries
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *dense = dyn, *xv = dyn + stripe_dwords, *dval = xv + sx, *dmul = xv + 2 * (size_t)sx, *dbeg = xv + 3 * (size_t)sx,
             *dend = xv + 4 * (size_t)sx;
    __shared__ uint32_t ndist;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t shift = hdr[H_SHIFT], id_mask = (1u << id_bits) - 1u;
    const uint32_t low_mask = (1u << shift) - 1u, occ_cap = (1u << (11u - (COMPACT ? shift : 0u))) - 1u; // compact: shift <= 10
    const uint64_t nrows = rows ? hdr[H_NOVF] : nx;
    const uint32_t stripe_cols = stripe_dwords * PER;
    const bool one_stripe = ny <= stripe_cols;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD with its own L2.  The rows in flight on one XCD
    // are CONSECUTIVE ones (gridDim / 8 of them): neighbouring sketches share most of their hashes -- a family's
    // copies -- so their buckets are fetched into that L2 once, not into all eight.
    const uint32_t G = gridDim.x, per_xcd = G / 8u;
    const bool by_xcd = per_xcd != 0 && G % 8u == 0;
    const uint64_t off = by_xcd ? (uint64_t)(blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : blockIdx.x;
    // Two rows ahead of the walk: row k + 2's hashes and row k + 1's bucket bounds are loaded while row k's buckets
    // are walked (one hash per thread; sketches above 1024 hashes take the plain loads).  A row costs two dependent
    // memory round trips before its first bucket can be touched -- off the critical path this way.
    const bool ahead = sx <= (uint32_t)DENSE_THREADS;
    struct Pre {
        int64_t i;       // row of X, -1: past the end
        uint32_t skip;   // irregular row (flagsX): the merge's
        uint32_t v;      // my hash of it
        uint32_t bs, be; // its bucket
        uint32_t pv, nv; // REG: the hashes in front of and behind mine
    };
    auto load_row = [&](uint64_t k, Pre &q) {
        const uint64_t r = k * G + off;
        q.i = -1;
        q.skip = 0;
        q.v = 0;
        q.pv = q.nv = 0;
        if (r < nrows) {
            q.i = (int64_t)(rows ? rows[r] : r);
            q.skip = rows ? 0u : flagsX[q.i];
            if (ahead && (uint32_t)tid < sx) {
                const uint32_t *xp = X + (uint64_t)q.i * sx;
                q.v = xp[tid];
                if constexpr (REG) { // (the same cache lines)
                    q.pv = tid ? xp[tid - 1] : 0u;
                    q.nv = (uint32_t)tid + 1u < sx ? xp[tid + 1] : 0u;
                }
            }
        }
    };
    auto load_bounds = [&](Pre &q) {
        q.bs = q.be = 0;
        if (ahead && q.i >= 0 && (uint32_t)tid < sx) {
            const uint32_t b = q.v >> shift;
            if (b < nbk) {
                q.bs = start[b];
                q.be = start[b + 1];
            }
        }
    };
    // the counters start at zero and every flush leaves them so (a thread clears the dwords it has just written out)
    for (uint32_t t = tid * 4; t < stripe_dwords; t += DENSE_THREADS * 4) // stripe_dwords is a multiple of 8
        *reinterpret_cast<uint4 *>(dense + t) = make_uint4(0, 0, 0, 0);
    if constexpr (REG)
        lds_barrier();
    // ZERO-AHEAD (round 5; MEASURED, NOT FASTER, opt-in: POLYHIP_K2_ZAHEAD=1).  A row's flush writes all its columns, zeros
    // included, AFTER its walk, with the memory pipes idle during the walk and the LDS idle during the flush.  In this variant
    // the zeros of the NEXT row go out during THIS row's walk -- a wave slips two 1 KB stores behind each group of bucket loads
    // -- and what is left behind the walk is a scan of the LDS counters that writes the row's few hundred non-zero counts over
    // its zeros and clears them.  Per 12,500 x 100,000 row block (profiles/r05e_join_zero_ahead_ablation.log): whole-row flush
    // 1.154 ms, zero-ahead 1.179, zero-ahead with NO zeros written at all 0.983 -- so the flush costs 0.17 ms, not the 5.9 of
    // 24 us per row the round-4 ablation suggested, and vmcnt counts a wave's loads and stores in ONE order: the next group's
    // wait covers the stores slipped in front of it, and the walk slows by what the flush had cost.  Giving all the stores to
    // the wave with spare time (a 1000-hash row: wave 15 holds 40 of 64 elements) made that wave the critical path (1.57 ms:
    // one wave streams ~8 GB/s of such stores).  One stripe, 16-byte aligned rows; anything else keeps the whole-row flush.
    const bool zahead = zero_ahead != 0 && one_stripe && ld % 8u == 0 && (reinterpret_cast<uintptr_t>(counts) & 15u) == 0;
    const uint32_t row_bytes = (uint32_t)ny * 2u;                      // (one stripe: ny <= stripe_cols < 2^17)
    const uint32_t wave_share = (((row_bytes + NWAVES - 1) / NWAVES) + 1023u) & ~1023u; // whole 1 KB wave stores
    constexpr int per_group = 2;
    int64_t zeroed = -1;                                               // the row whose zeros have been issued
    uint8_t *zbase = nullptr;                                          // row being zero-filled by this wave
    uint32_t zpos = 0, zend = 0;                                       // ... its share [zpos, zend) of the row's bytes
    typedef uint32_t zvec_t __attribute__((ext_vector_type(4)));
    auto zero_step = [&](int nstores) __attribute__((always_inline)) {
        for (int q = 0; q < nstores && zpos < zend; ++q, zpos += 1024u) { // (wave-uniform)
            const uint32_t at = zpos + (uint32_t)lane * 16u;
#ifdef PH_K2_ZA_NOSTORE // ablation probe (no zeros written: wrong matrix)
            if (at == 0xFFFFFFFFu)
                zbase[0] = 0;
            continue;
#endif
            if (at + 16u <= row_bytes) {
                __builtin_nontemporal_store(zvec_t{0u, 0u, 0u, 0u}, reinterpret_cast<zvec_t *>(zbase + at));
            } else if (at < row_bytes) { // the row's last, short piece (ny no multiple of 8)
                for (uint32_t b = at; b < row_bytes; b += 2)
                    *reinterpret_cast<uint16_t *>(zbase + b) = 0;
            }
        }
    };
    auto zero_begin = [&](int64_t row, bool) __attribute__((always_inline)) { // this wave's share of `row` becomes its pending zero-fill
        zbase = reinterpret_cast<uint8_t *>(counts + (uint64_t)row * ld);
        zpos = min((uint32_t)wave * wave_share, row_bytes);
        zend = min(zpos + wave_share, row_bytes);
        zend = zpos + ((zend - zpos + 1023u) & ~1023u); // (the last piece's lanes beyond the row store nothing)
    };
    Pre cur, nxt;
    load_row(0, cur);
    load_bounds(cur);
    load_row(1, nxt);
    for (uint64_t k = 0; cur.i >= 0; ++k) {
        const uint64_t i = (uint64_t)cur.i;
        const bool work = !cur.skip; // wave-uniform (irregular rows belong to the merge)
        uint32_t rval = 0, rlim = 0, rbeg = 0, rend = 0; // REG: my element as a bucket descriptor (empty unless it is a first copy)
        if constexpr (REG) {
            const bool first = work && (uint32_t)tid < sx && (tid == 0 || cur.pv != cur.v);
            if (first && cur.be > cur.bs) { // (bounds of a bucket beyond nbk were never loaded: 0, 0)
                uint32_t a = 1;
                if ((uint32_t)tid + 1u < sx && cur.nv == cur.v) { // a value the row repeats (rare)
                    const uint32_t *xp = X + i * sx;
                    while ((uint32_t)tid + a < sx && xp[tid + a] == cur.v)
                        ++a;
                }
                if (COMPACT) {
                    rval = (shift ? (cur.v & low_mask) << (32u - shift) : 0u) + (1u << CK_LOW);
                    rlim = min(a, occ_cap) << CK_LOW;
                } else {
                    rval = cur.v;
                    rlim = a > (0xFFFFFFFFu >> id_bits) ? 0xFFFFFFFFu : a << id_bits;
                }
                rbeg = cur.bs;
                rend = cur.be;
            }
        }
        if (work && !REG) {
            lds_barrier();
            if (tid == 0)
                ndist = 0;
            if (ahead) {
                if ((uint32_t)tid < sx)
                    xv[tid] = cur.v;
            } else {
                const uint32_t *xp = X + i * sx;
                for (uint32_t p = tid; p < sx; p += DENSE_THREADS)
                    xv[p] = xp[p];
            }
            lds_barrier();
            for (uint32_t p = tid; p < sx; p += DENSE_THREADS) { // distinct values of the row and their buckets
                const uint32_t v = xv[p];
                if (p != 0 && xv[p - 1] == v)
                    continue;
                const uint32_t b = v >> shift;
                if (b >= nbk)
                    continue;
                const uint32_t bs = ahead ? cur.bs : start[b], be = ahead ? cur.be : start[b + 1];
                if (be == bs)
                    continue;
                uint32_t a = 1;
                while (p + a < sx && xv[p + a] == v)
                    ++a;
                const uint32_t slot = atomicAdd(&ndist, 1u);
                dval[slot] = v;
                dmul[slot] = a;
                dbeg[slot] = bs;
                dend[slot] = be;
            }
            lds_barrier();
        }
        // issue the loads of the rows ahead now: they land while this row's buckets are walked
        Pre nn;
        load_bounds(nxt);
        load_row(k + 2, nn);
        cur = nxt;
        nxt = nn;
        if (!work)
            continue;
        if (zahead) {
            if (zeroed != (int64_t)i) { // the workgroup's first row, or the row behind one the merge took: its zeros now, in the open
                zero_begin((int64_t)i, true);
                zero_step(1 << 20);
            }
            zeroed = -1;
            zpos = zend = 0;
            if (cur.i >= 0 && !cur.skip) { // (cur is the NEXT row by now)
                zero_begin(cur.i, false);
                zeroed = cur.i;
            }
        }
        const uint32_t nd = REG ? 0u : ndist;
        (void)nd;
        for (uint64_t c0 = 0; c0 < ny; c0 += stripe_cols) {
            const uint32_t ncols = (uint32_t)min((uint64_t)stripe_cols, ny - c0);
            // column c of the stripe = field c / ndw of dword c % ndw: neighbouring columns (a bucket is a family's
            // copies of one hash -- a run of them) sit in neighbouring dwords, so one LDS atomic's lanes hit distinct
            // banks, and the flush reads whole dwords of one field
            const uint32_t ndw = (((ncols + PER - 1) / PER) + 7u) & ~7u; // <= stripe_dwords (a multiple of 8)
            // k = col / ndw by one multiply-high: kmul = ceil(2^32 / ndw) is exact for col < PER * ndw while PER * ndw^2 < 2^32
            // (the host caps the stripe accordingly)
            const uint32_t kmul = (uint32_t)(((1ull << 32) + ndw - 1) / ndw);
            auto consume_wide = [&](const uint2 it, uint32_t v, uint32_t alim) {
                // alim = multiplicity << id_bits: "occurrence number < multiplicity" is one compare of the whole word
                // ((0, 0xFFFFFFFF) = no item fails it)
                if (it.x != v || it.y >= alim)
                    return;
                uint32_t col = it.y & id_mask;
                if (!one_stripe) {
                    if (col < c0 || col - c0 >= ncols)
                        return;
                    col -= (uint32_t)c0;
                }
                const uint32_t k = __umulhi(col, kmul);
                atomicAdd(&dense[col - k * ndw], 1u << (BITS * k));
            };
            // compact: key = the row value's low bits in the item's top field, alim = multiplicity << CK_LOW; "same value
            // and occurrence number < multiplicity" is ONE subtract and ONE compare (an item of another value wraps or
            // overshoots; the all-zero word = no item fails too), the counter's byte offset and field shift are in the item
            auto consume_compact = [&](const uint32_t it, uint32_t key, uint32_t alim) {
#if defined(PH_K2_J_NOCONSUME) // ablation probes (wrong counts): the walk without its consume / without its LDS atomics
                if (it == 0x12345u && key == 77u)
                    dense[0] = alim;
#elif defined(PH_K2_J_NOATOM)
                if (it - key < alim && it == 0x12345u)
                    dense[0] = alim;
#else
                if (it - key < alim)
                    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dense) + ((it >> 3) & 0x3FFFCu)), 1u << (it & 31u));
#endif
            };
            // Wave w owns the distinct values w, w + 16, w + 32, ...: lane l keeps the descriptor of the wave's l-th one
            // in registers (one LDS pass per 64 buckets) and the walk takes them from there by v_readlane -- bucket
            // bounds, value and multiplicity are scalars, no LDS round trip stands between two buckets.  DENSE_U
            // buckets at a time: the first 128 items of each are loaded back to back (a family's copies of one hash are
            // one bucket), then consumed.
            constexpr uint32_t NW = DENSE_THREADS / 64;
            // up to 64 buckets whose descriptors sit in the wave's lanes
            auto walk_chunk = [&](const uint32_t mval, const uint32_t mlim, const uint32_t mbeg, const uint32_t mend,
                                  const uint32_t cnt) __attribute__((always_inline)) {
                if constexpr (COMPACT) {
                    // Buffer loads: a bucket is its own little buffer (base and size are scalars built on the scalar
                    // unit), every lane reads at the constant offset 4 * lane, and a lane beyond the bucket's end gets 0 =
                    // no item -- no address arithmetic, no bounds compare, no select on the vector unit.
                    // (round 5, measured and not kept: ONE 8-byte load per lane -- items 2 l and 2 l + 1 -- instead of two 4-byte
                    // ones, 1.26 against 1.17-1.21 ms per row block.  What the ablations say a row's walk IS
                    // (profiles/r05f_join_ablation.log, r05f_join_flush_ablation.log): the consume and its LDS atomics are free
                    // (1.165 without them, 1.166 with), every bucket read from ONE place in L1 still 0.94, no walk at all 0.57 of
                    // which 0.43 are the flush's STORES at the HBM write rate (5.9 TB/s) and 0.15 everything else; with the walk
                    // the stores cost 0.23: a wave's next loads wait behind its own flush stores -- vmcnt is one queue.)
                    const uint32_t lane4 = (uint32_t)lane * 4u;
#if PH_K2_PIPE2
                    // (round 6) two groups of DENSE_U buckets in flight: group g + 1's loads are issued BEFORE group g's items
                    // are consumed (vmcnt counts in order: the wait in front of a group's consume leaves the younger group's
                    // loads outstanding), so a wave's eight dependent load-wait-consume rounds per row become one wait plus
                    // seven that overlap the consume in front of them
                    {
                        uint32_t itA[DENSE_U][2], lenA[DENSE_U], itB[DENSE_U][2], lenB[DENSE_U];
                        __amdgpu_buffer_rsrc_t rsA[DENSE_U], rsB[DENSE_U];
                        auto issue = [&](uint32_t j0, uint32_t (&it)[DENSE_U][2], uint32_t (&len)[DENSE_U],
                                         __amdgpu_buffer_rsrc_t (&rs)[DENSE_U]) __attribute__((always_inline)) {
#pragma unroll
                            for (int u = 0; u < DENSE_U; ++u) {
                                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)mbeg, (int)(j0 + u));
                                len[u] = (uint32_t)__builtin_amdgcn_readlane((int)mend, (int)(j0 + u)) - b;
                                rs[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(items + b), 0, (int)(len[u] * 4u), 0x00020000);
                                it[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, 0, 0);
                                it[u][1] = __builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, 256, 0);
                            }
                        };
                        auto eat = [&](uint32_t j0, uint32_t (&it)[DENSE_U][2], uint32_t (&len)[DENSE_U],
                                       __amdgpu_buffer_rsrc_t (&rs)[DENSE_U]) __attribute__((always_inline)) {
#pragma unroll
                            for (int u = 0; u < DENSE_U; ++u) {
                                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)mval, (int)(j0 + u));
                                const uint32_t alim = (uint32_t)__builtin_amdgcn_readlane((int)mlim, (int)(j0 + u));
                                consume_compact(it[u][0], key, alim);
                                if (len[u] > 64u) { // wave-uniform
                                    consume_compact(it[u][1], key, alim);
                                    for (uint32_t t = 128; t < len[u]; t += 64) // rest of a long bucket
                                        consume_compact(__builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, (int)(t * 4u), 0), key, alim);
                                }
                            }
                        };
                        if (cnt)
                            issue(0, itA, lenA, rsA);
                        for (uint32_t j0 = 0; j0 < cnt; j0 += 2 * DENSE_U) {
                            const bool second = j0 + DENSE_U < cnt;
                            if (second)
                                issue(j0 + DENSE_U, itB, lenB, rsB);
                            zero_step(per_group);
                            eat(j0, itA, lenA, rsA);
                            if (second) {
                                if (j0 + 2 * DENSE_U < cnt)
                                    issue(j0 + 2 * DENSE_U, itA, lenA, rsA);
                                zero_step(per_group);
                                eat(j0 + DENSE_U, itB, lenB, rsB);
                            }
                        }
                    }
#elif defined(PH_K2_J_X4A) || defined(PH_K2_J_X2A)
                    // PROBES (round 6; counts may be wrong: a foreign item in front of the bucket can pass the key test): ONE wide
                    // load per bucket from the bucket's start rounded DOWN to 16 (8) bytes -- is the walk bound by the number of
                    // vector-memory instructions (2000 dword loads per row through one CU's address unit)?
                    {
#ifdef PH_K2_J_X4A
                        constexpr uint32_t W = 4;
#else
                        constexpr uint32_t W = 2;
#endif
                        const uint32_t laneW = (uint32_t)lane * 4u * W;
                        for (uint32_t j0 = 0; j0 < cnt; j0 += DENSE_U) {
                            uint32_t it[DENSE_U][W], len[DENSE_U];
                            __amdgpu_buffer_rsrc_t rs[DENSE_U];
#pragma unroll
                            for (int u = 0; u < DENSE_U; ++u) {
                                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)mbeg, (int)(j0 + u));
                                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)mend, (int)(j0 + u));
                                const uint32_t ba = b & ~(W - 1u);
                                len[u] = e - ba;
                                rs[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(items + ba), 0, (int)(len[u] * 4u), 0x00020000);
                                if constexpr (W == 4) {
                                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs[u], (int)laneW, 0, 0);
                                    it[u][0] = v[0], it[u][1] = v[1], it[u][2] = v[2], it[u][3] = v[3];
                                } else {
                                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs[u], (int)laneW, 0, 0);
                                    it[u][0] = v[0], it[u][1] = v[1];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < DENSE_U; ++u) {
                                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)mval, (int)(j0 + u));
                                const uint32_t alim = (uint32_t)__builtin_amdgcn_readlane((int)mlim, (int)(j0 + u));
#pragma unroll
                                for (uint32_t w = 0; w < W; ++w)
                                    consume_compact(it[u][w], key, alim);
                                for (uint32_t t = 64u * W; t < len[u]; t += 64) // rest of a long bucket
                                    consume_compact(__builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)(lane * 4u), (int)(t * 4u), 0), key, alim);
                            }
                        }
                    }
#else
                    for (uint32_t j0 = 0; j0 < cnt; j0 += DENSE_U) {
                        uint32_t it[DENSE_U][2], len[DENSE_U];
                        __amdgpu_buffer_rsrc_t rs[DENSE_U];
#pragma unroll
                        for (int u = 0; u < DENSE_U; ++u) {
                            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)mbeg, (int)(j0 + u));
                            len[u] = (uint32_t)__builtin_amdgcn_readlane((int)mend, (int)(j0 + u)) - b;
#ifdef PH_K2_J_L1 // ablation probe (wrong counts): every bucket's items from ONE place (L1 hits)
                            rs[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(items), 0, (int)(len[u] * 4u), 0x00020000);
#else
                            rs[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(items + b), 0, (int)(len[u] * 4u), 0x00020000);
#endif
                            it[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, 0, 0);
                            it[u][1] = __builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, 256, 0);
                        }
                        zero_step(per_group); // (zero-ahead variant: the next row's zeros ride behind this group's loads)
#pragma unroll
                        for (int u = 0; u < DENSE_U; ++u) {
                            const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)mval, (int)(j0 + u));
                            const uint32_t alim = (uint32_t)__builtin_amdgcn_readlane((int)mlim, (int)(j0 + u));
                            consume_compact(it[u][0], key, alim);
                            if (len[u] > 64u) { // wave-uniform
                                consume_compact(it[u][1], key, alim);
                                for (uint32_t t = 128; t < len[u]; t += 64) // rest of a long bucket
                                    consume_compact(__builtin_amdgcn_raw_buffer_load_b32(rs[u], (int)lane4, (int)(t * 4u), 0), key, alim);
                            }
                        }
                    }
#endif
                } else {
                    for (uint32_t j0 = 0; j0 < cnt; j0 += DENSE_U) {
                        uint2 it[DENSE_U][2];
                        uint32_t beg[DENSE_U], end[DENSE_U];
#pragma unroll
                        for (int u = 0; u < DENSE_U; ++u) {
                            beg[u] = (uint32_t)__builtin_amdgcn_readlane((int)mbeg, (int)(j0 + u));
                            end[u] = (uint32_t)__builtin_amdgcn_readlane((int)mend, (int)(j0 + u));
                            it[u][0] = it[u][1] = make_uint2(0u, 0xFFFFFFFFu);
                            const uint32_t t = beg[u] + lane;
                            if (t < end[u])
                                it[u][0] = items[t];
                            if (t + 64 < end[u])
                                it[u][1] = items[t + 64];
                        }
                        zero_step(per_group);
#pragma unroll
                        for (int u = 0; u < DENSE_U; ++u) {
                            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)mval, (int)(j0 + u));
                            const uint32_t alim = (uint32_t)__builtin_amdgcn_readlane((int)mlim, (int)(j0 + u));
                            consume_wide(it[u][0], v, alim);
                            if (end[u] - beg[u] > 64u) { // wave-uniform
                                consume_wide(it[u][1], v, alim);
                                for (uint32_t t = beg[u] + 128 + lane; t < end[u]; t += 64) // rest of a long bucket
                                    consume_wide(items[t], v, alim);
                            }
                        }
                    }
                }
            };
            if constexpr (REG) {
                // my wave's 64 consecutive elements of the row (the last wave of a 1000-hash row: 40)
                const uint32_t w0 = (uint32_t)wave * 64u;
#ifndef PH_K2_NOWALK
                if (w0 < sx)
                    walk_chunk(rval, rlim, rbeg, rend, min(64u, sx - w0));
#endif
            } else {
#ifdef PH_K2_NOWALK // ablation probe (wrong counts): how long is a row without its bucket walk?
                for (uint32_t jb = 0; false && wave + NW * jb < nd; jb += 64) {
#else
                for (uint32_t jb = 0; wave + NW * jb < nd; jb += 64) {
#endif
                    const uint32_t mine = wave + NW * (jb + lane);
                    uint32_t mval = 0, mlim = 0, mbeg = 0, mend = 0; // beyond nd: an empty bucket
                    if (mine < nd) {
                        const uint32_t a = dmul[mine];
                        mval = dval[mine];
                        if (COMPACT) {
                            // key = the value's low bits in the item's top field, + 1 in the occurrence field (items store
                            // occurrence + 1); multiplicity capped at what the field numbers
                            mval = (shift ? (mval & low_mask) << (32u - shift) : 0u) + (1u << CK_LOW);
                            mlim = min(a, occ_cap) << CK_LOW;
                        } else {
                            mlim = a > (0xFFFFFFFFu >> id_bits) ? 0xFFFFFFFFu : a << id_bits;
                        }
                        mbeg = dbeg[mine];
                        mend = dend[mine];
                    }
                    walk_chunk(mval, mlim, mbeg, mend, min(64u, (nd - wave - NW * jb + NW - 1) / NW)); // my buckets in this chunk
                }
            }
            if (zahead) {
                zero_step(1 << 20); // what the walk's groups did not carry (a wave with few buckets, a short row)
                // every zero of the next row (and, a row ago, of this one) is in L2 before anybody patches over it
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                // the row's non-zero counts over its zeros; the counters go back to zero where they were touched
                uint16_t *prow = counts + i * ld;
#ifdef PH_K2_NOFLUSH
                for (uint32_t t = tid * 8; false && t < ndw; t += DENSE_THREADS * 8) {
#else
                for (uint32_t t = tid * 8; t < ndw; t += DENSE_THREADS * 8) {
#endif
                    const uint4 d0 = *reinterpret_cast<const uint4 *>(dense + t);
                    const uint4 d1 = *reinterpret_cast<const uint4 *>(dense + t + 4);
                    if ((d0.x | d0.y | d0.z | d0.w | d1.x | d1.y | d1.z | d1.w) == 0u)
                        continue;
                    const uint32_t d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (d[q]) {
                            dense[t + q] = 0;
#pragma unroll
                            for (uint32_t kf = 0; kf < PER; ++kf) {
                                const uint32_t f = (d[q] >> (BITS * kf)) & FMASK, col = kf * ndw + t + (uint32_t)q;
                                if (f && col < ncols)
                                    prow[col] = (uint16_t)f;
                            }
                        }
                }
                lds_barrier();
                continue;
            }
            lds_barrier();
            // flush the stripe whole, zeros included: field k of dwords [0, ndw) = columns [k * ndw, (k + 1) * ndw)
            uint16_t *crow = counts + i * ld + c0;
            const uint32_t al = (uint32_t)(((uintptr_t)crow) & 15u);
            if (al == 0) {
                // aligned rows (every row of a matrix whose stride is a multiple of 8): ONE pass -- a thread reads eight
                // counter dwords once, clears them, and sends each of their PER fields to its own run of eight columns
                // (field k = columns [k * ndw, (k + 1) * ndw), k * ndw a multiple of 8: all stores 16-byte aligned)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#ifdef PH_K2_NOFLUSH // ablation probe (nothing written): how long is a row without its flush?
                for (uint32_t t = tid * 8; false && t < ndw; t += DENSE_THREADS * 8) {
#else
                for (uint32_t t = tid * 8; t < ndw; t += DENSE_THREADS * 8) {
#endif
#ifdef PH_K2_F_NOLDS // ablation probe: the flush without its LDS traffic (stores what the thread index says)
                    const uint4 d0 = make_uint4(t, t, t, t), d1 = d0;
#else
                    const uint4 d0 = *reinterpret_cast<const uint4 *>(dense + t);
                    const uint4 d1 = *reinterpret_cast<const uint4 *>(dense + t + 4);
                    *reinterpret_cast<uint4 *>(dense + t) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4 *>(dense + t + 4) = make_uint4(0, 0, 0, 0);
#endif
                    const uint32_t d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (uint32_t k = 0; k < PER; ++k) {
                        const uint32_t cb = k * ndw;
                        if (cb + t >= ncols)
                            break;
                        uint32_t f[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            f[q] = (d[q] >> (BITS * k)) & FMASK;
                        if (cb + t + 8 <= ncols) {
                            // written once, never read here: nontemporal, so the 2 B per pair do not push the index out of L2
                            const u32x4_t o = {f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16)};
#ifdef PH_K2_F_NOSTORE // ablation probe: the flush without its global stores
                            if (o.x == 0x12345u)
#endif
#ifdef PH_K2_F_PLAIN // measured: ordinary (temporal) stores instead of nontemporal ones
                            *reinterpret_cast<u32x4_t *>(crow + cb + t) = o;
#else
                            __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(crow + cb + t));
#endif
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (cb + t + q < ncols)
                                    crow[cb + t + q] = (uint16_t)f[q];
                        }
                    }
                }
                lds_barrier();
                continue;
            }
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t cb = k * ndw; // a multiple of 8: crow + cb keeps crow's alignment
                if (cb >= ncols)
                    break;
                const uint32_t n = min(ndw, ncols - cb);
                // the last field's pass runs over all ndw dwords and clears them (the same thread has read a dword in
                // every pass: no barrier needed in between)
                const bool last = cb + ndw >= ncols;
                const uint32_t lim = last ? ndw : n;
                if (al == 0) { // eight columns per 16-byte store (kept for reference: aligned rows take the single pass above)
                    for (uint32_t t = tid * 8; t < lim; t += DENSE_THREADS * 8) {
                        const uint4 d0 = *reinterpret_cast<const uint4 *>(dense + t);
                        const uint4 d1 = *reinterpret_cast<const uint4 *>(dense + t + 4);
                        if (last) {
                            *reinterpret_cast<uint4 *>(dense + t) = make_uint4(0, 0, 0, 0);
                            *reinterpret_cast<uint4 *>(dense + t + 4) = make_uint4(0, 0, 0, 0);
                        }
                        uint32_t f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            f[q] = (f[q] >> (BITS * k)) & FMASK;
                        if (t + 8 <= n) {
                            // written once, never read here: nontemporal, so the 2 B per pair do not push the index out of L2
                            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                            const u32x4_t o = {f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16)};
                            __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t *>(crow + cb + t));
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (t + q < n)
                                    crow[cb + t + q] = (uint16_t)f[q];
                        }
                    }
                } else if ((al & 7u) == 0) { // four columns per 8-byte store
                    for (uint32_t t = tid * 4; t < lim; t += DENSE_THREADS * 4) {
                        const uint4 d = *reinterpret_cast<const uint4 *>(dense + t);
                        if (last)
                            *reinterpret_cast<uint4 *>(dense + t) = make_uint4(0, 0, 0, 0);
                        const uint32_t f0 = (d.x >> (BITS * k)) & FMASK, f1 = (d.y >> (BITS * k)) & FMASK,
                                       f2 = (d.z >> (BITS * k)) & FMASK, f3 = (d.w >> (BITS * k)) & FMASK;
                        if (t + 4 <= n) {
                            *reinterpret_cast<uint2 *>(crow + cb + t) = make_uint2(f0 | (f1 << 16), f2 | (f3 << 16));
                        } else {
                            if (t < n)
                                crow[cb + t] = (uint16_t)f0;
                            if (t + 1 < n)
                                crow[cb + t + 1] = (uint16_t)f1;
                            if (t + 2 < n)
                                crow[cb + t + 2] = (uint16_t)f2;
                        }
                    }
                } else if ((al & 3u) == 0) { // two per 4-byte store
                    for (uint32_t t = tid * 2; t < lim; t += DENSE_THREADS * 2) {
                        const uint2 d = *reinterpret_cast<const uint2 *>(dense + t);
                        if (last)
                            *reinterpret_cast<uint2 *>(dense + t) = make_uint2(0, 0);
                        const uint32_t f0 = (d.x >> (BITS * k)) & FMASK, f1 = (d.y >> (BITS * k)) & FMASK;
                        if (t + 2 <= n)
                            *reinterpret_cast<uint32_t *>(crow + cb + t) = f0 | (f1 << 16);
                        else if (t < n)
                            crow[cb + t] = (uint16_t)f0;
                    }
                } else {
                    for (uint32_t t = tid; t < lim; t += DENSE_THREADS) {
                        const uint32_t d = dense[t];
                        if (last)
                            dense[t] = 0;
                        if (t < n)
                            crow[cb + t] = (uint16_t)((d >> (BITS * k)) & FMASK);
                    }
                }
            }
            lds_barrier();
        }
    }
}

// mash.go:107-135 for one pair, receiver = X_i
__device__ uint32_t similarity_count(const uint32_t *__restrict__ x, uint32_t sx, const uint32_t *__restrict__ y,
                                     uint32_t sy)
{
    const uint32_t *lg = x, *sm = y;
    uint32_t sl = sx, ss = sy;
    if (sx < sy) { // :112-115
        lg = y;
        sm = x;
        sl = sy;
        ss = sx;
    }
    if (lg[sl - 1] < sm[0] || sm[ss - 1] < lg[0]) // :117
        return 0;
    uint32_t same = 0, a = 0, b = 0;
    uint32_t va = sm[0], vb = lg[0];
    while (true) { // :121-132
        if (va == vb) {
            ++same;
            ++a;
            ++b;
            if (a >= ss || b >= sl)
                break;
            va = sm[a];
            vb = lg[b];
        } else if (va < vb) {
            if (++a >= ss)
                break;
            va = sm[a];
        } else {
            if (++b >= sl)
                break;
            vb = lg[b];
        }
    }
    return same;
}

__global__ __launch_bounds__(THREADS) void generic_kernel(const uint32_t *__restrict__ X, uint64_t nx, uint32_t sx,
                                                         const uint32_t *__restrict__ Y, uint64_t ny, uint32_t sy,
                                                         const uint32_t *__restrict__ hdr,
                                                         const uint32_t *__restrict__ irrX,
                                                         const uint32_t *__restrict__ regX,
                                                         const uint32_t *__restrict__ irrY,
                                                         const uint32_t *__restrict__ ovfX, int ovf_done,
                                                         uint16_t *__restrict__ counts, uint64_t ld)
{
    if (Y == nullptr)
        return; // no raw Y sketches on this device (round-5 advice): the caller reads mode / irregular / overflow counters and reports
    const bool all = hdr[H_MODE] == MODE_GENERIC;
    const uint64_t nIrrX = hdr[H_NIRRX], nIrrY = hdr[H_NIRRY], nRegX = hdr[H_NREGX], nOvf = hdr[H_NOVF];
    const uint64_t partA = all ? nx * ny : nIrrX * ny; // (irregular row) x (every column)
    const uint64_t partB = all ? 0 : nRegX * nIrrY;    // (regular row) x (irregular column)
    const uint64_t partC = (all || ovf_done) ? 0 : nOvf * ny; // (overflowed row) x (every column), unless the dense join took them
    const uint64_t total = partA + partB + partC;
    for (uint64_t p = (uint64_t)blockIdx.x * THREADS + threadIdx.x; p < total; p += (uint64_t)gridDim.x * THREADS) {
        uint64_t i, j;
        if (p < partA) {
            const uint64_t r = p / ny;
            j = p - r * ny;
            i = all ? r : irrX[r];
        } else if (p < partA + partB) {
            const uint64_t q = p - partA;
            const uint64_t r = q / nIrrY;
            i = regX[r];
            j = irrY[q - r * nIrrY];
        } else {
            const uint64_t q = p - partA - partB;
            const uint64_t r = q / ny;
            i = ovfX[r];
            j = q - r * ny;
        }
        counts[i * ld + j] = (uint16_t)similarity_count(X + i * sx, sx, Y + j * sy, sy);
    }
}

// mash.go:134,139: 1 - float64(same)/float64(smaller.SketchSize)
// Distance = 1 - float64(same) / float64(size) (mash.go:134,139) for a whole count matrix.  A count is one of size + 1
// integers, so the IEEE division happens once per VALUE -- a table of the size + 1 results in LDS, filled by every
// workgroup for itself -- and a pair costs one table read: the kernel is the stream of its 2 + 8 bytes per pair (round 3
// divided twice per pair: a 64-bit index division and the fp64 one).  A workgroup takes tiles of 2048 consecutive pairs
// of one row; a wave's load is 256 contiguous bytes (two counts per lane) and its store 1 KB (two distances per lane),
// whole 128-byte lines where the rows allow.
constexpr uint32_t DIST_TAB_MAX = 8192; // table entries (64 KB of LDS); larger sketch sizes divide per pair
constexpr uint32_t DIST_TILE = THREADS * 8;
__global__ __launch_bounds__(THREADS) void distance_kernel(const uint16_t *__restrict__ counts, uint64_t nx,
                                                          uint64_t ny, uint64_t ldc, double smaller, uint32_t ntab,
                                                          double *__restrict__ dist, uint64_t ldd)
{
    extern __shared__ __attribute__((aligned(16))) double dtab[];
    for (uint32_t c = threadIdx.x; c < ntab; c += THREADS)
        dtab[c] = 1 - (double)c / smaller;
    __syncthreads();
    auto value = [&](uint32_t c) { return c < ntab ? dtab[c] : 1 - (double)c / smaller; };
    const uint64_t tpr = (ny + DIST_TILE - 1) / DIST_TILE, ntiles = nx * tpr;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t i = t / tpr, j0 = (t - i * tpr) * DIST_TILE; // wave-uniform division
        const uint16_t *cp = counts + i * ldc + j0;
        double *dp = dist + i * ldd + j0;
        const uint64_t left = ny - j0;
        const bool fast = left >= DIST_TILE && ((reinterpret_cast<uintptr_t>(cp) & 3u) | (reinterpret_cast<uintptr_t>(dp) & 15u)) == 0;
        if (fast) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = reinterpret_cast<const uint32_t *>(cp)[q * THREADS + threadIdx.x];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                reinterpret_cast<double2 *>(dp)[q * THREADS + threadIdx.x] = make_double2(value(w[q] & 0xFFFFu), value(w[q] >> 16));
        } else {
            for (uint64_t e = threadIdx.x; e < left && e < DIST_TILE; e += THREADS)
                dp[e] = value(cp[e]);
        }
    }
}

// the index's self-join size sum_b |Y_b|^2 from the finished start[] (an index assembled from parts built elsewhere:
// fine_kernel only counted the buckets it built itself)
__global__ __launch_bounds__(THREADS) void self_join_kernel(const uint32_t *__restrict__ start, uint32_t nbk,
                                                           uint32_t *__restrict__ hdr)
{
    unsigned long long sq = 0;
    for (uint32_t b = blockIdx.x * THREADS + threadIdx.x; b < nbk; b += gridDim.x * THREADS) {
        const unsigned long long c = start[b + 1] - start[b];
        sq += c * c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        sq += __shfl_xor(sq, d, 64);
    if ((threadIdx.x & 63) == 0 && sq)
        atomicAdd(reinterpret_cast<unsigned long long *>(&hdr[H_EST_LO]), sq);
}

// Parts of one index: coarse bucket boundaries b[0] = 0 <= b[1] <= ... <= b[nparts] = nc such that every part holds about
// the same number of ITEMS (bottom-s sketches crowd the low values: equal value ranges would be far from equal work).
// A pure function of the coarse histogram's scan, which every rank computes from the same gathered sketches.
static void part_bounds(const uint32_t *cstart, uint32_t nc, uint32_t nparts, uint32_t *b)
{
    const uint64_t total = cstart[nc];
    uint32_t c = 0;
    for (uint32_t p = 0; p <= nparts; ++p) {
        const uint64_t want = (total * p + nparts - 1) / nparts;
        while (c < nc && cstart[c] < want)
            ++c;
        b[p] = p == nparts ? nc : c;
    }
    b[0] = 0;
}

} // namespace k2
} // namespace polyhip

using namespace polyhip;

extern "C" {

// Y sets of 2^32 hashes and more are joined in column stripes of fewer than that (an index item's position is 32 bits);
// POLYHIP_K2_MAX_ITEMS lowers the bound (testing aid: the striping then runs on small inputs)
static uint64_t stripe_sketches(uint64_t ny, uint32_t sy)
{
    uint64_t max_items = (1ull << 32) - 1;
    if (const char *e = getenv("POLYHIP_K2_MAX_ITEMS")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 1 && v < max_items)
            max_items = v;
    }
    const uint64_t per = std::max<uint64_t>(1, max_items / std::max<uint32_t>(sy, 1u));
    return std::min<uint64_t>(std::min<uint64_t>(ny, per), (1ull << 31) - 1);
}

size_t polyhip_mash_shared_counts_workspace_bytes(uint64_t nx, uint32_t sx, uint64_t ny, uint32_t sy)
{
    return k2::layout(nx, sx, stripe_sketches(ny, sy), sy).total;
}

// resets what an earlier call left of its X side in a workspace whose index is being reused
static __global__ void reset_x_kernel(uint32_t *__restrict__ hdr)
{
    hdr[k2::H_NIRRX] = hdr[k2::H_NREGX] = hdr[k2::H_NOVF] = 0;
    hdr[k2::H_MODE] = k2::MODE_SPARSE;
}

// dense join geometry: counter width from the largest possible count, stripe from what LDS holds next to the row
struct DenseGeom {
    int bits;
    uint32_t per, stripe_dwords, ndw; // ndw: counter dwords of a one-stripe join over all ny columns
    uint64_t stripe_cols, stripes;
    size_t row_bytes;
    bool force;      // every pair through the merge (ids beyond 24 bits, X rows beyond the LDS stage)
    bool dense_all;  // the dense join takes every regular row
    bool compact_ok; // ... in one stripe whose counter addresses fit a compact item
};
static DenseGeom dense_geom(uint32_t sx, uint32_t sy, uint64_t ny)
{
    DenseGeom g;
    g.force = ny > (1ull << k2::ID_BITS_MAX) || sx > k2::S_MAX;
    g.bits = std::min(sx, sy) <= 1023u ? 10 : 16;
    g.per = 32u / (uint32_t)g.bits;
    g.row_bytes = (((size_t)5 * sx + 3) & ~(size_t)3) * 4;
    const size_t lds_max = 160 * 1024 - 256;
    // <= 37832 dwords with three fields (the kernel's multiply-high `column / ndw` is exact for 3 * ndw^2 < 2^32)
    const size_t sdw_cap = g.per == 3 ? 37832u : 46328u;
    g.stripe_dwords =
        g.row_bytes + 4096 <= lds_max ? (uint32_t)std::min<size_t>(((lds_max - g.row_bytes) / 4) & ~(size_t)7, sdw_cap) : 0u;
    g.stripe_cols = (uint64_t)g.stripe_dwords * g.per;
    g.stripes = g.stripe_cols ? (ny + g.stripe_cols - 1) / g.stripe_cols : ~0ull;
    g.ndw = (uint32_t)std::min<uint64_t>((((ny + g.per - 1) / g.per) + 7) & ~7ull, 0xFFFFFFFFull);
    // Up to two stripes the dense join takes EVERY regular row: it reads each bucket at most twice, bumps one LDS counter
    // per shared hash and writes the row's counts once, zeros included -- no zero-fill of the matrix, no hash probing.
    // Beyond that (hundreds of thousands of columns) rows go through the sparse join first and only the ones whose
    // table overflows come here.  POLYHIP_K2_DENSE=0 keeps the sparse join in front, POLYHIP_K2_COMPACT=0 the 8-byte
    // items (testing aids).
    g.dense_all = !g.force && g.stripes <= 2 && !env_is("POLYHIP_K2_DENSE", '0');
    g.compact_ok = g.dense_all && g.stripes == 1 && g.ndw < 65536u && !env_is("POLYHIP_K2_COMPACT", '0');
    return g;
}

// what: 1 = build the index of Y, 2 = join X against the index in the workspace, 3 = both
static int shared_counts_impl(int what, const uint32_t *d_X, uint64_t nx, uint32_t sx, const uint32_t *d_Y, uint64_t ny,
                              uint32_t sy, uint16_t *d_counts, uint64_t ld, void *d_work, size_t work_bytes,
                              polyhip_stream_t stream, uint32_t part = 0, uint32_t nparts = 1, bool parts_api = false)
{
    bool build = what & 1;
    const bool join = what & 2;
    if ((join && sx == 0) || sy == 0)
        return set_error(POLYHIP_ERR_PANIC,
                         "mash.Similarity with SketchSize 0 indexes Sketches[-1] (mash.go:117): the reference panics");
    PH_REQUIRE(sx <= 65535 && sy <= 65535, "polyhip_mash_shared_counts: SketchSize > 65535 does not fit the u16 counts");
    if ((join && nx == 0) || ny == 0)
        return POLYHIP_OK;
    // (d_Y may be null when a prebuilt index is reused and the caller holds no raw Y sketches -- the item exchange of a device
    // list: the merge, the only reader of raw Y behind the build, then refuses instead of reading rows that are not there)
    PH_REQUIRE((d_Y || !build) && d_work && (!join || (d_X && d_counts)), "polyhip_mash_shared_counts: null pointer");
    PH_REQUIRE(!join || ld >= ny, "polyhip_mash_shared_counts: row stride %llu < ny %llu", (unsigned long long)ld,
               (unsigned long long)ny);
    PH_REQUIRE(nx < (1ull << 31) && ny < (1ull << 31) && ny * (uint64_t)sy < (1ull << 32),
               "polyhip_mash_shared_counts: 2^31 sketches / 2^32 Y hashes or more behind ONE index (polyhip_mash_shared_counts_dev "
               "stripes such sets itself; the index_build / reuse pair does not)");
    const k2::Layout L = k2::layout(nx, sx, ny, sy);
    PH_REQUIRE(work_bytes >= (join ? L.total : L.off_flagsX), "polyhip_mash_shared_counts: workspace too small (%zu < %zu)",
               work_bytes, join ? L.total : L.off_flagsX);
    hipStream_t st = as_stream(stream);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint32_t *hdr = reinterpret_cast<uint32_t *>(w);
    uint8_t *flagsX = w + L.off_flagsX, *flagsY = w + L.off_flagsY;
    uint32_t *irrX = reinterpret_cast<uint32_t *>(w + L.off_irrX), *regX = reinterpret_cast<uint32_t *>(w + L.off_regX),
             *ovfX = reinterpret_cast<uint32_t *>(w + L.off_ovfX), *irrY = reinterpret_cast<uint32_t *>(w + L.off_irrY);
    uint32_t *start = reinterpret_cast<uint32_t *>(w + L.off_start),
             *gcount = reinterpret_cast<uint32_t *>(w + L.off_gcount),
             *cstart = reinterpret_cast<uint32_t *>(w + L.off_cstart), *gcur = reinterpret_cast<uint32_t *>(w + L.off_gcur);
    uint2 *citems = reinterpret_cast<uint2 *>(w + L.off_citems);
    uint2 *items = reinterpret_cast<uint2 *>(w + L.off_items);
    uint16_t *pos = reinterpret_cast<uint16_t *>(w + L.off_pos);
    uint32_t *g4count = reinterpret_cast<uint32_t *>(w + L.off_g4count), *dupmap = reinterpret_cast<uint32_t *>(w + L.off_dup),
             *c4start = reinterpret_cast<uint32_t *>(w + L.off_c4start), *g4cur = reinterpret_cast<uint32_t *>(w + L.off_g4cur);
    uint2 *duplist = reinterpret_cast<uint2 *>(w + L.off_duplist);

    // the join packs the Y sketch id into 24 bits and stages an X row in LDS
    const int force = (ny > (1ull << k2::ID_BITS_MAX) || sx > k2::S_MAX) ? 1 : 0;
    (void)stripe_sketches;
    uint32_t id_bits = 1; // bits of a Y sketch id; the rest of an item's second dword numbers the copies of a value
    while ((1ull << id_bits) < ny && id_bits < k2::ID_BITS_MAX)
        ++id_bits;
    const uint32_t max_occ = (1u << (32 - id_bits)) - 2u; // all-ones stays free (the join's "no item" marker)

    // Item format.  An index built on its own assumes that the X sets to come have Y's SketchSize (an all-vs-all, a
    // database of one sketch size); built together with a join it takes that join's geometry.  If a later join does not
    // fit what the build assumed (another counter width, more than one stripe), the index is rebuilt with 8-byte items
    // first -- correct, at the price of a build.
    // Compact items are ALWAYS made for gY -- a function of (ny, sy) alone, which every later reuse call has to repeat --,
    // never for the geometry of the join that happened to come with the build: a workspace then holds either 8-byte items
    // or gY's compact ones (the device says which, H_FMT), whoever built it, and a reuse call can tell from its own
    // arguments whether it can read them (round-3 advice: a fused call with min(sx, sy) <= 1023 < sy used to leave items
    // for 10-bit counters that a later reuse with sx >= 1024 decoded as 16-bit ones).
    const DenseGeom gJ = dense_geom(join ? sx : sy, sy, ny), gY = dense_geom(sy, sy, ny);
    const bool fits_Y = gJ.compact_ok && gJ.bits == gY.bits; // the join at hand reads items made for gY
    bool allow_compact = gY.compact_ok && (!join || fits_Y);
    if (join && !build && gY.compact_ok && !fits_Y) {
        build = true; // the index in the workspace may be compact and this join cannot read that
        allow_compact = false;
    }
    const DenseGeom &gB = gY; // the geometry compact items are made for

    if (build) {
        // header, Y flags and the histogram start at zero
        PH_HIP(hipMemsetAsync(w, 0, L.off_irrY, st));
        PH_HIP(hipMemsetAsync(gcount, 0, (size_t)L.nc * 4, st));
        // The sliced build on 4-byte intermediate items (struct Layout) where the HOST's conditions hold -- compact items may be
        // made, SketchSize <= 1024, at most 131,072 sketches, the whole index in one go (the parts API and the item exchange keep
        // the two-level build's coarse buckets) --; the device's conditions (the largest value, the bucket shift, how often a
        // sketch repeats a hash) are decided by plan4_kernel / lists_kernel in hdr[H_B4], and BOTH builds are launched: the
        // kernels of the one that is not to run return at once.  POLYHIP_K2_B4=0 (and POLYHIP_K2_STAGE=0): the two-level build.
        const bool b4 = allow_compact && sy <= 1024 && ny <= 131072 && nparts == 1 && !parts_api && !env_is("POLYHIP_K2_B4", '0') &&
                        !env_is("POLYHIP_K2_STAGE", '0');
        const uint32_t G4 = ny > 65536 ? 2u : 1u;
        // level 1's stage shape: sketches per batch x slots per sketch and round (POLYHIP_K2_B4_SLOTS=128|64|32: measuring aid)
        const int slots = env_is("POLYHIP_K2_B4_SLOTS", '1') ? 128 : env_is("POLYHIP_K2_B4_SLOTS", '3') ? 32 : 64;
        const int bsk = (int)k2::B4_STAGE / slots;
        if (b4)
            PH_HIP(hipMemsetAsync(g4count, 0, L.off_duplist - L.off_g4count, st)); // the histogram and the dupmap
        hipLaunchKernelGGL(k2::maxlast_kernel, dim3((unsigned)std::min<uint64_t>((ny + 1023) / 1024, 256)), dim3(1024), 0, st, d_Y, ny, sy, hdr);
        if (b4) {
            uint32_t slice_len = (uint32_t)slots * 41u / 64u; // ~0.65 of a sketch's slots per round: a slice beyond them costs a round
            if (const char *e = getenv("POLYHIP_K2_B4_TL")) {
                const long v = strtol(e, nullptr, 10);
                if (v >= 8 && v <= 1024)
                    slice_len = (uint32_t)v;
            }
            hipLaunchKernelGGL(k2::plan4_kernel, dim3(1), dim3(64), 0, st, hdr, L.nbk_log2, sy, slice_len, k2::b4_cpp_max(bsk));
            const size_t smem = (size_t)k2::B4_NC_MAX * 4;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2::check4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem));
            hipLaunchKernelGGL(k2::check4_kernel, dim3((unsigned)std::min<uint64_t>((ny + 15) / 16, 256)), dim3(1024), smem, st, d_Y, ny, sy,
                               flagsY, hdr, force, max_occ, G4, g4count, dupmap, duplist, pos);
        }
        hipLaunchKernelGGL(k2::check_kernel<true>, dim3(k2::check_grid(ny)), dim3(k2::THREADS), (size_t)L.nc * 4, st, d_Y, ny, sy,
                           flagsY, hdr, force, max_occ, L.nbk_log2, L.fpc_log2, L.nc, gcount, 0u);
        hipLaunchKernelGGL(k2::lists_kernel, dim3((unsigned)((ny + k2::THREADS - 1) / k2::THREADS)), dim3(k2::THREADS), 0, st,
                           flagsX, (uint64_t)0, flagsY, ny, irrX, regX, irrY, hdr, L.nbk_log2, allow_compact ? 1 : 0);
        if (b4) // the sliced build was planned and then called off (hdr[H_B4] == 2): the two-level build's histogram after all
            hipLaunchKernelGGL(k2::check_kernel<true>, dim3(k2::check_grid(ny)), dim3(k2::THREADS), (size_t)L.nc * 4, st, d_Y, ny, sy,
                               flagsY, hdr, force, max_occ, L.nbk_log2, L.fpc_log2, L.nc, gcount, 2u);
        // ---- inverted index of the Y side: two-level partition by value
        const uint32_t per_batch = std::max<uint32_t>(1u, k2::BATCH_ITEMS / sy);
        const unsigned batches = (unsigned)((ny + per_batch - 1) / per_batch);
        hipLaunchKernelGGL(k2::coarse_scan_kernel, dim3(1), dim3(1024), 0, st, gcount, L.nc, cstart, gcur, start, L.nbk);
        if (b4) // (after coarse_scan_kernel: both write start[nbk])
            hipLaunchKernelGGL(k2::scan4_kernel, dim3(1), dim3(1024), 0, st, g4count, G4, hdr, c4start, g4cur, start, L.nbk);
        // one part of the index (multi-rank build): the coarse buckets [c0, c1) only, every item at its final place.  The
        // bounds come from the coarse histogram -- the one point where the host has to look at device data.
        uint32_t c0 = 0, c1 = L.nc;
        if (nparts > 1) {
            std::vector<uint32_t> hc(L.nc + 1), b(nparts + 1);
            PH_HIP(hipMemcpyAsync(hc.data(), cstart, (size_t)(L.nc + 1) * 4, hipMemcpyDeviceToHost, st));
            PH_HIP(hipStreamSynchronize(st));
            k2::part_bounds(hc.data(), L.nc, nparts, b.data());
            c0 = b[part];
            c1 = b[part + 1];
        }
        // level-1 scatter: through LDS when a sketch fits the stage (POLYHIP_K2_STAGE=0: the direct scatter, testing aid)
        if (sy <= k2::STAGE_ITEMS && L.nc <= 1024 && !env_is("POLYHIP_K2_STAGE", '0')) {
            const uint32_t pb = std::max<uint32_t>(1u, k2::STAGE_ITEMS / sy);
            const size_t smem = (size_t)k2::STAGE_ITEMS * 8 + (size_t)L.nc * 12;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2::coarse_scatter_staged_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(k2::coarse_scatter_staged_kernel, dim3((unsigned)std::min<uint64_t>((ny + pb - 1) / pb, PH_K2_STAGE_GRID)), dim3(k2::STAGE_THREADS),
                               smem, st, d_Y, ny, sy, flagsY, hdr, L.fpc_log2, L.nc, pb, id_bits, c0, c1, gcur, citems, 0u);
        } else {
            hipLaunchKernelGGL(k2::coarse_scatter_kernel, dim3(batches), dim3(k2::THREADS), (size_t)L.nc * 8, st, d_Y, ny, sy,
                               flagsY, hdr, L.fpc_log2, L.nc, per_batch, id_bits, c0, c1, gcur, citems, 0u);
        }
        if (b4) {
            const size_t smem = ((size_t)k2::B4_STAGE + 64 + 3 * (size_t)k2::b4_cpp_max(bsk) + 64 + 16 + 2 * (size_t)bsk) * 4;
            auto kern = slots == 128 ? k2::scatter4_kernel<128, 128> : slots == 32 ? k2::scatter4_kernel<512, 32> : k2::scatter4_kernel<256, 64>;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(kern, dim3(512), dim3(1024), smem, st, d_Y, (uint32_t)ny, sy, flagsY, pos, hdr, G4, g4cur,
                               reinterpret_cast<uint32_t *>(citems));
        }
        if (c1 > c0)
            hipLaunchKernelGGL(k2::fine_kernel<false>, dim3(std::min<uint32_t>(c1 - c0, 256u * 8u)), dim3(k2::FINE_THREADS), 0, st, citems,
                               cstart, c0, c1, L.fpc_log2, hdr, start, items, id_bits, gB.ndw ? gB.ndw : 8u, (uint32_t)gB.bits,
                               (const uint2 *const *)nullptr, (const uint32_t *)nullptr, 0u, 0u);
        if (b4)
        {
            if (env_is("POLYHIP_K2_B4_FINE", '5')) // (measuring aid: two 512-thread workgroups per CU, 32 items per thread)
                hipLaunchKernelGGL((k2::fine4_kernel<512, 32>), dim3(1024), dim3(512), 0, st, reinterpret_cast<const uint32_t *>(citems),
                                   c4start, G4, hdr, start, reinterpret_cast<uint32_t *>(items), dupmap, duplist,
                                   gB.ndw ? gB.ndw : 8u, (uint32_t)gB.bits);
            else
                hipLaunchKernelGGL((k2::fine4_kernel<1024, 32>), dim3(512), dim3(1024), 0, st, reinterpret_cast<const uint32_t *>(citems),
                                   c4start, G4, hdr, start, reinterpret_cast<uint32_t *>(items), dupmap, duplist,
                                   gB.ndw ? gB.ndw : 8u, (uint32_t)gB.bits);
        }
        PH_HIP(hipGetLastError());
    }
    if (!join)
        return POLYHIP_OK;

    // ---- X side
    if (!build)
        hipLaunchKernelGGL(reset_x_kernel, dim3(1), dim3(1), 0, st, hdr);
    PH_HIP(hipMemsetAsync(flagsX, 0, nx, st));
    hipLaunchKernelGGL(k2::check_kernel<false>, dim3(k2::check_grid(nx)), dim3(k2::THREADS), 0, st, d_X, nx, sx, flagsX, hdr, force,
                       0xFFFFFFFEu, 0u, 0u, 0u, (uint32_t *)nullptr, 0u);
    hipLaunchKernelGGL(k2::lists_kernel, dim3((unsigned)((nx + k2::THREADS - 1) / k2::THREADS)), dim3(k2::THREADS), 0, st,
                       flagsX, nx, flagsY, (uint64_t)0, irrX, regX, irrY, hdr, L.nbk_log2, -1);
    hipLaunchKernelGGL(k2::decide_kernel, dim3(1), dim3(1), 0, st, hdr, k2::join_cost(nx, sx, ny, sy)); // the merge only wins on huge buckets

    const int bits = gJ.bits;
    const uint32_t per = gJ.per, stripe_dwords = gJ.stripe_dwords;
    const size_t row_bytes = gJ.row_bytes;
    const uint64_t stripe_cols = gJ.stripe_cols;
    auto launch_dense = [&](const uint32_t *rows, unsigned blocks) -> int {
        const uint32_t sdw = (uint32_t)std::min<uint64_t>(stripe_dwords, (((ny + per - 1) / per) + 7) & ~7ull);
        const size_t smem = row_bytes + (size_t)sdw * 4;
        // rows of at most 1024 hashes: a thread per element, no staging of the row (REG; POLYHIP_K2_REGROW=0: the staged form)
        const bool regrow = sx <= (uint32_t)k2::DENSE_THREADS && !env_is("POLYHIP_K2_REGROW", '0');
#define PH_K2_DENSE_LAUNCH(BITS_, COMPACT_)                                                                                   \
    do {                                                                                                                      \
        auto kern = regrow ? k2::rowjoin_dense_kernel<BITS_, COMPACT_, true> : k2::rowjoin_dense_kernel<BITS_, COMPACT_, false>; \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k2::DENSE_THREADS), smem, st, d_X, nx, sx, flagsX, start,                 \
                           static_cast<const void *>(items), L.nbk, hdr, rows, ny, sdw, id_bits, d_counts, ld,                \
                           env_is("POLYHIP_K2_ZAHEAD", '1') ? 1 : 0);                                                         \
    } while (0)
        // both item formats are launched when the index MAY be compact: the device decided (H_FMT), the instantiation
        // that does not match returns at once
        if (bits == 10) {
            PH_K2_DENSE_LAUNCH(10, false);
            if (allow_compact && rows == nullptr)
                PH_K2_DENSE_LAUNCH(10, true);
        } else {
            PH_K2_DENSE_LAUNCH(16, false);
            if (allow_compact && rows == nullptr)
                PH_K2_DENSE_LAUNCH(16, true);
        }
#undef PH_K2_DENSE_LAUNCH
        return POLYHIP_OK;
    };
    const bool dense_all = gJ.dense_all;
    int ovf_done = 0;
    if (dense_all) {
        if (int rc = launch_dense(nullptr, (unsigned)std::min<uint64_t>(nx, 256ull)))
            return rc;
        ovf_done = 1;
    } else {
        // rowjoin stores only non-zero cells
        if (ld == ny) {
            PH_HIP(hipMemsetAsync(d_counts, 0, nx * ny * 2, st));
        } else {
            PH_HIP(hipMemset2DAsync(d_counts, ld * 2, 0, ny * 2, nx, st));
        }
        if (!force) {
            const size_t smem = (size_t)sx * 20;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2::rowjoin_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            const unsigned blocks = (unsigned)std::min<uint64_t>(nx, 256ull * 16ull);
            hipLaunchKernelGGL(k2::rowjoin_kernel, dim3(blocks), dim3(k2::THREADS), smem, st, d_X, nx, sx, flagsX, start, items,
                               L.nbk, id_bits, hdr, ovfX, d_counts, ld);
            // rows that overflowed their hash table: the dense join, as many columns per stripe as LDS holds
            if (stripe_cols >= 8192) {
                if (int rc = launch_dense(ovfX, (unsigned)std::min<uint64_t>(nx, 256ull)))
                    return rc;
                ovf_done = 1;
            }
        }
    }
    {
        const uint64_t pairs = nx * ny;
        const unsigned blocks = (unsigned)std::min<uint64_t>((pairs + k2::THREADS - 1) / k2::THREADS, 256ull * 16ull);
        hipLaunchKernelGGL(k2::generic_kernel, dim3(blocks), dim3(k2::THREADS), 0, st, d_X, nx, sx, d_Y, ny, sy, hdr, irrX,
                           regX, irrY, ovfX, ovf_done, d_counts, ld);
    }
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

int polyhip_mash_shared_counts_dev(const uint32_t *d_X, uint64_t nx, uint32_t sx, const uint32_t *d_Y, uint64_t ny,
                                   uint32_t sy, uint16_t *d_counts, uint64_t ld, void *d_work, size_t work_bytes,
                                   polyhip_stream_t stream)
{
    const uint64_t per = sy ? stripe_sketches(ny, sy) : ny;
    if (ny <= per || sx == 0 || sy == 0)
        return shared_counts_impl(3, d_X, nx, sx, d_Y, ny, sy, d_counts, ld, d_work, work_bytes, stream);
    // column stripes: each with its own index, one after the other in the same workspace; the pairs of a stripe depend
    // on nothing outside it, so the matrix is the stripes side by side
    PH_REQUIRE(ld >= ny, "polyhip_mash_shared_counts: row stride %llu < ny %llu", (unsigned long long)ld, (unsigned long long)ny);
    for (uint64_t c0 = 0; c0 < ny; c0 += per) {
        const uint64_t m = std::min<uint64_t>(per, ny - c0);
        if (int rc = shared_counts_impl(3, d_X, nx, sx, d_Y + c0 * (uint64_t)sy, m, sy, d_counts ? d_counts + c0 : nullptr, ld, d_work,
                                        work_bytes, stream))
            return rc;
    }
    return POLYHIP_OK;
}

int polyhip_mash_index_build_dev(const uint32_t *d_Y, uint64_t ny, uint32_t sy, void *d_work, size_t work_bytes,
                                 polyhip_stream_t stream)
{
    return shared_counts_impl(1, nullptr, 0, 1, d_Y, ny, sy, nullptr, 0, d_work, work_bytes, stream);
}

int polyhip_mash_index_build_part_dev(const uint32_t *d_Y, uint64_t ny, uint32_t sy, uint32_t part, uint32_t nparts,
                                      void *d_work, size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(nparts >= 1 && part < nparts, "polyhip_mash_index_build_part: part %u of %u", part, nparts);
    return shared_counts_impl(1, nullptr, 0, 1, d_Y, ny, sy, nullptr, 0, d_work, work_bytes, stream, part, nparts, true);
}

int polyhip_mash_index_part_spans(uint64_t ny, uint32_t sy, uint32_t nparts, const void *d_work, size_t work_bytes,
                                  uint64_t *item_spans, uint64_t *start_spans, polyhip_stream_t stream)
{
    PH_REQUIRE(nparts >= 1 && d_work && item_spans && start_spans, "polyhip_mash_index_part_spans: bad argument");
    PH_REQUIRE(ny >= 1 && sy >= 1, "polyhip_mash_index_part_spans: empty index");
    const k2::Layout L = k2::layout(0, 1, ny, sy);
    PH_REQUIRE(work_bytes >= L.off_flagsX, "polyhip_mash_index_part_spans: workspace too small");
    const uint8_t *w = static_cast<const uint8_t *>(d_work);
    std::vector<uint32_t> hc(L.nc + 1), b(nparts + 1);
    uint32_t h[k2::H_WORDS];
    PH_HIP(hipMemcpyAsync(hc.data(), w + L.off_cstart, (size_t)(L.nc + 1) * 4, hipMemcpyDeviceToHost, as_stream(stream)));
    PH_HIP(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, as_stream(stream)));
    PH_HIP(hipStreamSynchronize(as_stream(stream)));
    k2::part_bounds(hc.data(), L.nc, nparts, b.data());
    const uint64_t item_bytes = h[k2::H_FMT] ? 4 : 8; // compact items are half the size (and half the all-gather)
    for (uint32_t p = 0; p <= nparts; ++p) {
        item_spans[p] = L.off_items + (uint64_t)hc[b[p]] * item_bytes;
        start_spans[p] = L.off_start + ((uint64_t)b[p] << L.fpc_log2) * 4;
    }
    return POLYHIP_OK;
}

int polyhip_mash_index_format_dev(const void *d_work, uint32_t *item_bytes)
{
    PH_REQUIRE(d_work && item_bytes, "polyhip_mash_index_format: null pointer");
    uint32_t h[k2::H_WORDS];
    PH_HIP(hipMemcpy(h, d_work, sizeof h, hipMemcpyDeviceToHost));
    *item_bytes = h[k2::H_FMT] ? 4u : 8u;
    return POLYHIP_OK;
}

int polyhip_mash_index_build_info_dev(const void *d_work, uint32_t info[6])
{
    PH_REQUIRE(d_work && info, "polyhip_mash_index_build_info: null pointer");
    uint32_t h[k2::H_WORDS];
    PH_HIP(hipMemcpy(h, d_work, sizeof h, hipMemcpyDeviceToHost));
    info[0] = h[k2::H_B4];
    info[1] = h[k2::H_B4_NC];
    info[2] = h[k2::H_B4_R];
    info[3] = h[k2::H_B4_CPP];
    info[4] = h[k2::H_B4_OVER];
    info[5] = h[k2::H_B4_NDUP];
    return POLYHIP_OK;
}

int polyhip_mash_index_finalize_dev(uint64_t ny, uint32_t sy, void *d_work, size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(d_work && ny >= 1 && sy >= 1, "polyhip_mash_index_finalize: bad argument");
    const k2::Layout L = k2::layout(0, 1, ny, sy);
    PH_REQUIRE(work_bytes >= L.off_flagsX, "polyhip_mash_index_finalize: workspace too small");
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint32_t *hdr = reinterpret_cast<uint32_t *>(w);
    PH_HIP(hipMemsetAsync(hdr + k2::H_EST_LO, 0, 8, as_stream(stream)));
    hipLaunchKernelGGL(k2::self_join_kernel, dim3(std::min<uint32_t>((L.nbk + k2::THREADS - 1) / k2::THREADS, 2048u)),
                       dim3(k2::THREADS), 0, as_stream(stream), reinterpret_cast<const uint32_t *>(w + L.off_start), L.nbk, hdr);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

int polyhip_mash_index_allgather_dev(polyhip_comm *c, uint64_t ny, uint32_t sy, void *d_work, size_t work_bytes,
                                     polyhip_stream_t stream)
{
    const int nranks = polyhip_comm_size(c);
    PH_REQUIRE(nranks >= 1, "polyhip_mash_index_allgather: null communicator");
    std::vector<uint64_t> items(nranks + 1), starts(nranks + 1);
    if (int rc = polyhip_mash_index_part_spans(ny, sy, (uint32_t)nranks, d_work, work_bytes, items.data(), starts.data(), stream))
        return rc;
    // two ragged all-gathers in place: every rank's items sit at their final offsets already
    if (int rc = polyhip_allgatherv_dev(c, d_work, items.data(), stream))
        return rc;
    if (int rc = polyhip_allgatherv_dev(c, d_work, starts.data(), stream))
        return rc;
    return polyhip_mash_index_finalize_dev(ny, sy, d_work, work_bytes, stream);
}

int polyhip_mash_shared_counts_reuse_dev(const uint32_t *d_X, uint64_t nx, uint32_t sx, const uint32_t *d_Y, uint64_t ny,
                                         uint32_t sy, uint16_t *d_counts, uint64_t ld, void *d_work, size_t work_bytes,
                                         polyhip_stream_t stream)
{
    return shared_counts_impl(2, d_X, nx, sx, d_Y, ny, sy, d_counts, ld, d_work, work_bytes, stream);
}

int polyhip_mash_shared_counts_mode_dev(const void *d_work, uint32_t *mode, uint32_t *n_irregular_x,
                                        uint32_t *n_irregular_y, uint32_t *n_overflow_rows, uint64_t *join_estimate)
{
    PH_REQUIRE(d_work, "polyhip_mash_shared_counts_mode: null workspace");
    uint32_t h[k2::H_WORDS];
    PH_HIP(hipMemcpy(h, d_work, sizeof h, hipMemcpyDeviceToHost));
    if (mode)
        *mode = h[k2::H_MODE];
    if (n_irregular_x)
        *n_irregular_x = h[k2::H_NIRRX];
    if (n_irregular_y)
        *n_irregular_y = h[k2::H_NIRRY];
    if (n_overflow_rows)
        *n_overflow_rows = h[k2::H_NOVF];
    if (join_estimate)
        *join_estimate = (uint64_t)h[k2::H_EST_LO] | ((uint64_t)h[k2::H_EST_HI] << 32);
    return POLYHIP_OK;
}

int polyhip_mash_distance_from_counts_dev(const uint16_t *d_counts, uint64_t nx, uint64_t ny, uint64_t ld_counts,
                                          uint32_t sx, uint32_t sy, double *d_dist, uint64_t ld_dist,
                                          polyhip_stream_t stream)
{
    if (sx == 0 || sy == 0)
        return set_error(POLYHIP_ERR_PANIC, "mash.Distance with SketchSize 0: the reference panics (mash.go:117)");
    if (nx == 0 || ny == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_counts && d_dist, "polyhip_mash_distance_from_counts: null pointer");
    PH_REQUIRE(ld_counts >= ny && ld_dist >= ny, "polyhip_mash_distance_from_counts: row stride < ny");
    const uint32_t smaller = std::min(sx, sy);
    const uint32_t ntab = smaller + 1 <= k2::DIST_TAB_MAX ? smaller + 1 : 0u;
    const uint64_t tiles = nx * ((ny + k2::DIST_TILE - 1) / k2::DIST_TILE);
    const unsigned blocks = (unsigned)std::min<uint64_t>(tiles, 256ull * 32ull);
    hipLaunchKernelGGL(k2::distance_kernel, dim3(blocks), dim3(k2::THREADS), (size_t)ntab * sizeof(double), as_stream(stream),
                       d_counts, nx, ny, ld_counts, (double)smaller, ntab, d_dist, ld_dist);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

} // extern "C"

// Row blocks of the matrix from DEVICE-resident sketches to HOST buffers (the calling thread's current device): what
// crosses PCIe is the matrix (2 and/or 8 bytes per pair), so row blocks of ~64 MB alternate between two device slots;
// block b is joined on the calling thread's first stream while block b-1 travels back on the second (the joins
// themselves share the workspace's X side and stay in order on one stream).  One index for all blocks.  Everything the
// caller enqueued on the thread's first stream before (uploads, a sketching pass) is ordered in front.
uint64_t polyhip::k2_rows_per_block(uint64_t nx, uint64_t ny, bool counts, bool dist)
{
    const uint64_t per_pair = (counts ? 2 : 0) + (dist ? 8 : 0);
    return std::max<uint64_t>(1, std::min<uint64_t>(nx, HOST_CHUNK_BYTES / std::max<uint64_t>(1, ny * per_pair)));
}

int polyhip::k2_rows_to_host(const uint32_t *dX, uint64_t nx, uint32_t sx, const uint32_t *dY, uint64_t ny, uint32_t sy,
                             uint16_t *counts, double *dist, void *index_work, size_t index_work_bytes)
{
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    hipStream_t sc = hs.s[0], sd = hs.s[1];
    const uint64_t rows = k2_rows_per_block(nx, ny, counts != nullptr, dist != nullptr);
    const uint64_t nblocks = (nx + rows - 1) / rows;
    const bool one_index = stripe_sketches(ny, sy) >= ny; // else every block builds its stripes' indexes itself
    DevBuf dW;
    struct Slot {
        DevBuf dC, dD;
        hipEvent_t computed = nullptr, downloaded = nullptr;
        ~Slot()
        {
            if (computed)
                (void)hipEventDestroy(computed);
            if (downloaded)
                (void)hipEventDestroy(downloaded);
        }
    } slot[2];
    size_t wb = polyhip_mash_shared_counts_workspace_bytes(rows, sx, ny, sy);
    void *wp = index_work;
    if (index_work) {
        PH_REQUIRE(one_index && index_work_bytes >= wb, "k2_rows_to_host: the prebuilt index's workspace is too small (%zu < %zu)",
                   index_work_bytes, wb);
        wb = index_work_bytes;
    } else {
        PH_HIP(dW.alloc(wb));
        wp = dW.p;
    }
    for (uint64_t q = 0; q < std::min<uint64_t>(2, nblocks); ++q) {
        PH_HIP(slot[q].dC.alloc(rows * ny * 2 + 16));
        if (dist)
            PH_HIP(slot[q].dD.alloc(rows * ny * 8));
        PH_HIP(hipEventCreateWithFlags(&slot[q].computed, hipEventDisableTiming));
        PH_HIP(hipEventCreateWithFlags(&slot[q].downloaded, hipEventDisableTiming));
    }
    int rc = POLYHIP_OK;
    PH_REQUIRE(dY || index_work, "k2_rows_to_host: no raw Y sketches and no prebuilt index");
    if (one_index && !index_work)
        rc = polyhip_mash_index_build_dev(dY, ny, sy, wp, wb, sc);
    for (uint64_t b = 0; b < nblocks && rc == POLYHIP_OK; ++b) {
        Slot &S = slot[b & 1];
        const uint64_t r0 = b * rows, m = std::min<uint64_t>(rows, nx - r0);
        if (b >= 2)
            PH_HIP(hipStreamWaitEvent(sc, S.downloaded, 0)); // block b-2 has left this slot
        const uint32_t *dx = dX + r0 * (uint64_t)sx;
        rc = one_index ? polyhip_mash_shared_counts_reuse_dev(dx, m, sx, dY, ny, sy, S.dC.as<uint16_t>(), ny, wp, wb, sc)
                       : polyhip_mash_shared_counts_dev(dx, m, sx, dY, ny, sy, S.dC.as<uint16_t>(), ny, wp, wb, sc);
        if (rc == POLYHIP_OK && dist)
            rc = polyhip_mash_distance_from_counts_dev(S.dC.as<uint16_t>(), m, ny, ny, sx, sy, S.dD.as<double>(), ny, sc);
        if (rc != POLYHIP_OK)
            break;
        PH_HIP(hipEventRecord(S.computed, sc));
        PH_HIP(hipStreamWaitEvent(sd, S.computed, 0));
        if (counts)
            PH_HIP(hipMemcpyAsync(counts + r0 * ny, S.dC.p, m * ny * 2, hipMemcpyDeviceToHost, sd));
        if (dist)
            PH_HIP(hipMemcpyAsync(dist + r0 * ny, S.dD.p, m * ny * 8, hipMemcpyDeviceToHost, sd));
        PH_HIP(hipEventRecord(S.downloaded, sd));
    }
    const hipError_t e = hs.sync_both();
    if (rc != POLYHIP_OK)
        return rc;
    PH_HIP(e);
    return POLYHIP_OK;
}

extern "C" {

static int distance_matrix_one(const uint32_t *X, uint64_t nx, uint32_t sx, const uint32_t *Y, uint64_t ny, uint32_t sy,
                               uint16_t *counts, double *dist)
{
    if (sx == 0 || sy == 0)
        return polyhip_mash_shared_counts_dev(nullptr, nx, sx, nullptr, ny, sy, nullptr, 0, nullptr, 0, nullptr);
    if (nx == 0 || ny == 0)
        return POLYHIP_OK;
    PH_REQUIRE(X && Y && (counts || dist), "polyhip_mash_distance_matrix: null pointer");
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    DevBuf dX, dY;
    PH_HIP(dX.alloc(nx * (size_t)sx * 4));
    PH_HIP(dY.alloc(ny * (size_t)sy * 4));
    PH_HIP(hipMemcpyAsync(dY.p, Y, ny * (size_t)sy * 4, hipMemcpyHostToDevice, hs.s[0]));
    PH_HIP(hipMemcpyAsync(dX.p, X, nx * (size_t)sx * 4, hipMemcpyHostToDevice, hs.s[0]));
    const int rc = k2_rows_to_host(dX.as<uint32_t>(), nx, sx, dY.as<uint32_t>(), ny, sy, counts, dist);
    if (rc != POLYHIP_OK)
        (void)hs.sync_both(); // the uploads read the caller's memory
    return rc;
}

int polyhip_mash_distance_matrix(const uint32_t *X, uint64_t nx, uint32_t sx, const uint32_t *Y, uint64_t ny, uint32_t sy,
                                 uint16_t *counts, double *dist)
{
    std::shared_ptr<md::Pool> P = nx && ny && sx && sy ? md::pool() : nullptr;
    if (!P)
        return distance_matrix_one(X, nx, sx, Y, ny, sy, counts, dist);
    // every device holds all of Y (uploaded over its own link: the host is the source, so N parallel uploads
    // cost what one does) and its own index, and joins a contiguous block of X's rows; the row blocks go back to the
    // host side by side.  No collective: the matrix stays sharded by rows all the way.
    PH_REQUIRE(X && Y && (counts || dist), "polyhip_mash_distance_matrix: null pointer");
    const size_t nsh = md::size(*P);
    return md::run(*P, [&](size_t q) {
        const uint64_t r0 = (uint64_t)(((unsigned __int128)nx * q) / nsh), r1 = (uint64_t)(((unsigned __int128)nx * (q + 1)) / nsh);
        if (r0 == r1)
            return (int)POLYHIP_OK;
        return distance_matrix_one(X + r0 * (uint64_t)sx, r1 - r0, sx, Y, ny, sy, counts ? counts + r0 * ny : nullptr,
                                   dist ? dist + r0 * ny : nullptr);
    });
}

} // extern "C"

// ---- the index of a sketch set that is spread over a device list, without gathering the sketches (round 4) --------------
// DESIGN.md section 4.  Device q holds rows [i0, i1) of the n sketches (it has just made them).  Gathering the set (400 MB
// into every device at config 3) only to turn it into index items N times over is what bounds the step at ~2.8x on 8 GPUs;
// here every device
//   A  takes the largest last element of ITS rows                                  -> host: the set's maximum (bucket shift)
//   B  checks its rows and counts their items per coarse bucket                    -> host: the set's histogram, the part
//                                                                                     bounds (equal numbers of items)
//   C  runs level 1 on its rows (1/N of the scatter; sketch ids are the set's)     -> its items, grouped by coarse bucket
//   D  pulls, from every device, the items of ITS part of the value range (one peer copy per device: a device's items of
//      a range of coarse buckets are contiguous) and runs level 2 on them -- a coarse bucket is N pieces (fine_kernel<true>)
//   E  pulls the other parts' finished items and bucket starts (the same ragged exchange as the N-rank build's,
//      polyhip_mash_index_allgather_dev, by peer copies) and takes the self-join size of the whole index
// -- 1/N of level 1 and level 2 each, and what travels is 7/8 x (8-byte items of 1/N of the set + the 4-byte index) instead
// of 7/8 x (the sketches + nothing).  Rounds are md::run calls (each a barrier between the devices' threads); the host
// reduces a few kilobytes between them.  Falls back (*built = false, nothing lost but the rounds so far) when a sketch is
// irregular (the merge reads raw sketches of both sides), the join would not be the dense one, or the self-join size says
// "merge everything": the caller then gathers the sketches as before.
int polyhip::k2_enable_peer(int me, int other, int *transport)
{
    *transport = 0;
    if (me == other)
        return POLYHIP_OK;
    int can = 0;
    PH_HIP(hipDeviceCanAccessPeer(&can, me, other));
    *transport = can ? 1 : 2; // 2: the runtime stages the copy through the host -- slower, still correct, and REPORTED
    if (can) {               // (polyhip_mash_sketch_distance_matrix_last_info: round-4 verdict, the fall-back used to be silent)
        const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            PH_HIP(e);
        (void)hipGetLastError(); // "already enabled" is sticky otherwise
    }
    return POLYHIP_OK;
}

int polyhip::k2_exchange_index(md::Pool &P, std::vector<K2XShard> &sh, uint64_t n, uint32_t s, uint64_t rows_blk, bool *built)
{
    *built = false;
    const size_t N = sh.size();
    if (N < 2 || n == 0 || s == 0 || stripe_sketches(n, s) < n || n >= (1ull << 31))
        return POLYHIP_OK;
    const k2::Layout L = k2::layout(rows_blk, s, n, s);
    const DenseGeom gY = dense_geom(s, s, n);
    if (gY.force || !gY.dense_all)
        return POLYHIP_OK;
    const bool allow_compact = gY.compact_ok;
    uint32_t id_bits = 1;
    while ((1ull << id_bits) < n && id_bits < k2::ID_BITS_MAX)
        ++id_bits;
    const uint32_t max_occ = (1u << (32 - id_bits)) - 2u;
    const bool staged = s <= k2::STAGE_ITEMS && L.nc <= 1024 && !env_is("POLYHIP_K2_STAGE", '0');
    struct View { // one device's workspace
        uint8_t *w;
        uint32_t *hdr, *start, *gcount, *cstart, *gcur, *irrY;
        uint8_t *flagsY;
        uint2 *citems, *items;
        uint16_t *pos;
    };
    auto view = [&](K2XShard &x) {
        View v;
        v.w = x.work.as<uint8_t>();
        v.hdr = reinterpret_cast<uint32_t *>(v.w);
        v.flagsY = v.w + L.off_flagsY;
        v.irrY = reinterpret_cast<uint32_t *>(v.w + L.off_irrY);
        v.start = reinterpret_cast<uint32_t *>(v.w + L.off_start);
        v.gcount = reinterpret_cast<uint32_t *>(v.w + L.off_gcount);
        v.cstart = reinterpret_cast<uint32_t *>(v.w + L.off_cstart);
        v.gcur = reinterpret_cast<uint32_t *>(v.w + L.off_gcur);
        v.citems = reinterpret_cast<uint2 *>(v.w + L.off_citems);
        v.items = reinterpret_cast<uint2 *>(v.w + L.off_items);
        v.pos = reinterpret_cast<uint16_t *>(v.w + L.off_pos);
        return v;
    };
    // ---- A: workspace, the largest last element of my rows
    int rc = md::run(P, [&](size_t q) {
        K2XShard &x = sh[q];
        HostStreams &hs = host_streams();
        PH_HIP(hs.init());
        hipStream_t st = hs.s[0];
        x.work_bytes = L.total;
        PH_HIP(x.work.alloc(x.work_bytes));
        const View v = view(x);
        PH_HIP(hipMemsetAsync(v.w, 0, L.off_irrY, st));
        PH_HIP(hipMemsetAsync(v.gcount, 0, (size_t)L.nc * 4, st));
        const uint64_t m = x.i1 - x.i0;
        if (m)
            hipLaunchKernelGGL(k2::maxlast_kernel, dim3((unsigned)std::min<uint64_t>((m + 1023) / 1024, 256)), dim3(1024), 0, st,
                               x.sk + x.i0 * (uint64_t)s, m, s, v.hdr);
        PH_HIP(hipGetLastError());
        PH_HIP(hipMemcpyAsync(&x.h_maxval, v.hdr + k2::H_MAXVAL, 4, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
        return (int)POLYHIP_OK;
    });
    if (rc != POLYHIP_OK)
        return rc;
    uint32_t maxval = 0;
    for (const K2XShard &x : sh)
        maxval = std::max(maxval, x.h_maxval);
    // ---- B: check my rows, count their items per coarse bucket
    rc = md::run(P, [&](size_t q) {
        K2XShard &x = sh[q];
        hipStream_t st = host_streams().s[0];
        SyncOnExit wait_for_copies(st); // (copies into this scope's locals and the shard's host vectors)
        const View v = view(x);
        const uint64_t m = x.i1 - x.i0;
        PH_HIP(hipMemcpyAsync(v.hdr + k2::H_MAXVAL, &maxval, 4, hipMemcpyHostToDevice, st));
        if (m)
            hipLaunchKernelGGL(k2::check_kernel<true>, dim3(k2::check_grid(m)), dim3(k2::THREADS), (size_t)L.nc * 4, st,
                               x.sk + x.i0 * (uint64_t)s, m, s, v.flagsY + x.i0, v.hdr, 0, max_occ, L.nbk_log2, L.fpc_log2, L.nc,
                               v.gcount, 0u);
        PH_HIP(hipGetLastError());
        x.h_gcount.assign(L.nc, 0);
        std::vector<uint8_t> fl(m);
        PH_HIP(hipMemcpyAsync(x.h_gcount.data(), v.gcount, (size_t)L.nc * 4, hipMemcpyDeviceToHost, st));
        PH_HIP(hipMemcpyAsync(&x.h_maxmult, v.hdr + k2::H_MAXMULT, 4, hipMemcpyDeviceToHost, st));
        if (m)
            PH_HIP(hipMemcpyAsync(fl.data(), v.flagsY + x.i0, m, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
        x.h_nirr = 0;
        for (uint8_t f : fl)
            x.h_nirr += f ? 1 : 0;
        return (int)POLYHIP_OK;
    });
    if (rc != POLYHIP_OK)
        return rc;
    uint64_t nirr = 0;
    uint32_t maxmult = 0;
    std::vector<uint32_t> gstart(L.nc + 1, 0); // the SET's coarse offsets
    for (const K2XShard &x : sh) {
        nirr += x.h_nirr;
        maxmult = std::max(maxmult, x.h_maxmult);
    }
    if (nirr)
        return POLYHIP_OK; // an irregular sketch: its pairs go through the merge, which reads raw sketches
    {
        uint64_t run = 0;
        for (uint32_t c = 0; c < L.nc; ++c) {
            gstart[c] = (uint32_t)run;
            for (const K2XShard &x : sh)
                run += x.h_gcount[c];
        }
        gstart[L.nc] = (uint32_t)run;
    }
    std::vector<uint32_t> bnd(N + 1);
    k2::part_bounds(gstart.data(), L.nc, (uint32_t)N, bnd.data());
    // ---- C: level 1 on my rows
    rc = md::run(P, [&](size_t q) {
        K2XShard &x = sh[q];
        hipStream_t st = host_streams().s[0];
        SyncOnExit wait_for_copies(st); // (copies into this scope's locals and the shard's host vectors)
        const View v = view(x);
        const uint64_t m = x.i1 - x.i0;
        PH_HIP(hipMemcpyAsync(v.hdr + k2::H_MAXMULT, &maxmult, 4, hipMemcpyHostToDevice, st));
        // shift and item format: the same inputs on every device, the same answer (no irregular sketch: the lists stay empty)
        hipLaunchKernelGGL(k2::lists_kernel, dim3(1), dim3(k2::THREADS), 0, st, v.flagsY, (uint64_t)0, v.flagsY, (uint64_t)0,
                           v.irrY, v.irrY, v.irrY, v.hdr, L.nbk_log2, allow_compact ? 1 : 0);
        hipLaunchKernelGGL(k2::coarse_scan_kernel, dim3(1), dim3(1024), 0, st, v.gcount, L.nc, v.cstart, v.gcur, v.start, L.nbk);
        if (m) {
            const uint32_t *sk = x.sk + x.i0 * (uint64_t)s;
            if (staged) {
                const uint32_t pb = std::max<uint32_t>(1u, k2::STAGE_ITEMS / s);
                const size_t smem = (size_t)k2::STAGE_ITEMS * 8 + (size_t)L.nc * 12;
                PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2::coarse_scatter_staged_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                hipLaunchKernelGGL(k2::coarse_scatter_staged_kernel, dim3((unsigned)std::min<uint64_t>((m + pb - 1) / pb, PH_K2_STAGE_GRID)),
                                   dim3(k2::STAGE_THREADS), smem, st, sk, m, s, v.flagsY + x.i0, v.hdr, L.fpc_log2, L.nc, pb, id_bits,
                                   0u, L.nc, v.gcur, v.citems, (uint32_t)x.i0);
            } else {
                const uint32_t per_batch = std::max<uint32_t>(1u, k2::BATCH_ITEMS / s);
                hipLaunchKernelGGL(k2::coarse_scatter_kernel, dim3((unsigned)((m + per_batch - 1) / per_batch)), dim3(k2::THREADS),
                                   (size_t)L.nc * 8, st, sk, m, s, v.flagsY + x.i0, v.hdr, L.fpc_log2, L.nc, per_batch, id_bits, 0u,
                                   L.nc, v.gcur, v.citems, (uint32_t)x.i0);
            }
        }
        PH_HIP(hipGetLastError());
        x.h_cstart.assign(L.nc + 1, 0);
        PH_HIP(hipMemcpyAsync(x.h_cstart.data(), v.cstart, (size_t)(L.nc + 1) * 4, hipMemcpyDeviceToHost, st));
        PH_HIP(hipMemcpyAsync(&x.h_fmt, v.hdr + k2::H_FMT, 4, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
        return (int)POLYHIP_OK;
    });
    if (rc != POLYHIP_OK)
        return rc;
    const uint64_t item_bytes = sh[0].h_fmt ? 4 : 8;
    // ---- D: the items of my part of the value range from everybody, level 2 on them
    rc = md::run(P, [&](size_t q) {
        K2XShard &x = sh[q];
        hipStream_t st = host_streams().s[0];
        SyncOnExit wait_for_copies(st); // (copies into this scope's locals and the shard's host vectors)
        const View v = view(x);
        const uint32_t c0 = bnd[q], c1 = bnd[q + 1], ncp = c1 - c0 + 1;
        // the set's coarse offsets replace mine (every device's own offsets are on the host by now)
        PH_HIP(hipMemcpyAsync(v.cstart, gstart.data(), (size_t)(L.nc + 1) * 4, hipMemcpyHostToDevice, st));
        PH_HIP(hipMemcpyAsync(v.start + L.nbk, &gstart[L.nc], 4, hipMemcpyHostToDevice, st));
        PH_HIP(hipMemsetAsync(v.hdr + k2::H_EST_LO, 0, 8, st));
        if (c1 > c0) {
            std::vector<const uint2 *> base(N);
            std::vector<uint32_t> lo((size_t)N * ncp);
            uint64_t goff = x.h_cstart[L.nc]; // pulled pieces go behind my own items
            for (size_t p = 0; p < N; ++p) {
                const K2XShard &o = sh[p];
                const uint32_t a = o.h_cstart[c0], len = o.h_cstart[c1] - a;
                for (uint32_t c = c0; c <= c1; ++c)
                    lo[p * ncp + (c - c0)] = o.h_cstart[c] - a;
                if (p == q) {
                    base[p] = v.citems + a;
                    continue;
                }
                base[p] = v.citems + goff;
                if (len) {
                    int tr = 0;
                    if (int e = k2_enable_peer(x.dev, o.dev, &tr))
                        return e;
                    PH_HIP(hipMemcpyPeerAsync(v.citems + goff, x.dev, view(const_cast<K2XShard &>(o)).citems + a, o.dev, (size_t)len * 8, st));
                    if (x.stats)
                        x.stats->count(tr, (uint64_t)len * 8);
                }
                goff += len;
            }
            PH_REQUIRE(goff <= n * (uint64_t)s, "k2_exchange_index: pieces beyond the item region");
            PH_HIP(x.segptr.alloc(N * sizeof(const uint2 *)));
            PH_HIP(x.segtab.alloc(lo.size() * 4));
            PH_HIP(hipMemcpyAsync(x.segptr.p, base.data(), N * sizeof(const uint2 *), hipMemcpyHostToDevice, st));
            PH_HIP(hipMemcpyAsync(x.segtab.p, lo.data(), lo.size() * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k2::fine_kernel<true>, dim3(std::min<uint32_t>(c1 - c0, 256u * 8u)), dim3(k2::FINE_THREADS), 0, st,
                               v.citems, v.cstart, c0, c1, L.fpc_log2, v.hdr, v.start, v.items, id_bits, gY.ndw ? gY.ndw : 8u,
                               (uint32_t)gY.bits, x.segptr.as<const uint2 *>(), x.segtab.as<uint32_t>(), (uint32_t)N, ncp);
            PH_HIP(hipGetLastError());
            PH_HIP(hipStreamSynchronize(st)); // (the tables are host vectors of this scope)
        } else {
            PH_HIP(hipStreamSynchronize(st));
        }
        return (int)POLYHIP_OK;
    });
    if (rc != POLYHIP_OK)
        return rc;
    // ---- E: the other parts, finished; the self-join size of the whole index
    rc = md::run(P, [&](size_t q) {
        K2XShard &x = sh[q];
        hipStream_t st = host_streams().s[0];
        SyncOnExit wait_for_copies(st); // (copies into this scope's locals and the shard's host vectors)
        const View v = view(x);
        for (size_t p = 0; p < N; ++p) {
            if (p == q || bnd[p + 1] == bnd[p])
                continue;
            const K2XShard &o = sh[p];
            const View ov = view(const_cast<K2XShard &>(o));
            int tr = 0;
            if (int e = k2_enable_peer(x.dev, o.dev, &tr))
                return e;
            const uint64_t ia = (uint64_t)gstart[bnd[p]] * item_bytes, ib = (uint64_t)gstart[bnd[p + 1]] * item_bytes;
            if (ib > ia) {
                PH_HIP(hipMemcpyPeerAsync(reinterpret_cast<uint8_t *>(v.items) + ia, x.dev, reinterpret_cast<const uint8_t *>(ov.items) + ia,
                                          o.dev, ib - ia, st));
                if (x.stats)
                    x.stats->count(tr, ib - ia);
            }
            const uint64_t sa = (uint64_t)bnd[p] << L.fpc_log2, sb = (uint64_t)bnd[p + 1] << L.fpc_log2;
            PH_HIP(hipMemcpyPeerAsync(v.start + sa, x.dev, ov.start + sa, o.dev, (sb - sa) * 4, st));
            if (x.stats)
                x.stats->count(tr, (sb - sa) * 4);
        }
        if (int e = polyhip_mash_index_finalize_dev(n, s, v.w, x.work_bytes, st))
            return e;
        uint32_t est[2] = {0, 0};
        PH_HIP(hipMemcpyAsync(est, v.hdr + k2::H_EST_LO, 8, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
        x.h_est = (uint64_t)est[0] | ((uint64_t)est[1] << 32);
        return (int)POLYHIP_OK;
    });
    if (rc != POLYHIP_OK)
        return rc;
    // the join's own decision (decide_kernel), taken here for the largest X a device will bring: if the merge would take
    // the input, it needs every raw sketch -- gather them after all
    {
        uint64_t nx_max = 1; // the largest X a device will bring (the cost model scales both sides with nx: any nx decides alike)
        for (const K2XShard &x : sh)
            nx_max = std::max<uint64_t>(nx_max, x.i1 - x.i0);
        if (k2::join_cost(nx_max, s, n, s).merge_wins(sh[0].h_est))
            return POLYHIP_OK;
    }
    *built = true;
    return POLYHIP_OK;
}
