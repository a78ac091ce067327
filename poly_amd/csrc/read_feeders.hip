// read_feeders.hip -- the read feeders: a FASTQ or FASTA file image resident in HBM becomes the packed
// batch (bytes + offsets) the hot-path kernels take, without a host parse and without per-read objects.
//
// ---- FASTQ ----
// Follows io/fastq/fastq.go:117-216 ((*Parser).ParseNext) and :84-96 (ParseN): records are exactly
// four '\n'-terminated lines (identifier, sequence, '+', quality); Sequence is line 2 minus the '\n'
// (a '\r' is NOT stripped, as in the reference); parsing stops at the first bad record and the records
// before it are returned (ParseN).  Error conditions, in the order the reference meets them:
//   5  identifier line is empty                (fastq.go:156 indexes string(line)[0]: the reference panics)
//   6  an identifier field after a ' ' has no '=' (fastq.go:163 indexes optionalSplits[1]: panics)
//   2  empty sequence line                     (:176-178)
//   3  empty quality line                      (:197-199)
//   1  identifier line does not start with '@' (:203-205; reported after the four lines were read)
//   4  end of file inside a record, including a last line without '\n' (:142-148: "unexpected EOF")
//
// Device algorithm: newline positions by a block-count / scan / ranked-write compaction; one thread
// per 4-line record validates it and measures its sequence; an exclusive scan of the lengths gives the
// offsets; one workgroup per record copies the bytes.  Pure byte work, HBM bound: reads the file twice
// (newlines, gather) and writes the sequences once.
//
// ---- FASTA ----
// Follows io/fasta/fasta.go:102-238 ((*Parser).ParseNext / ParseN), quirks included:
//   * a line is skipped when it is empty or starts with ';' (:169); lines before the first '>' line are
//     skipped too (:208-216);
//   * a record ends when the NEXT line starts with '>' (:197-204) -- the check happens after a line is
//     read, so a '>' line directly after a header is part of that header's SEQUENCE: in a run of
//     consecutive '>' lines the 1st, 3rd, ... are headers and the 2nd, 4th, ... sequence lines;
//   * Sequence is the concatenation of the record's lines without their '\n' (a '\r' stays);
//   * a header without sequence is an error (code 2, :226-229), the records before it are kept;
//     a non-empty file without any header is an error (code 1, :223) -- unless its last line is an
//     unterminated non-skippable one, in which case the error wraps io.EOF and ParseN drops it (:107-110);
//   * a last record that ends in an unterminated, non-skippable line is returned WITH io.EOF by
//     ParseNext and therefore dropped by ParseN/ParseAll (fasta_test.go:139-142); an unterminated line of
//     at most one byte, or one starting with ';', is skippable and changes nothing.
// Device algorithm: the same newline compaction, which also leaves per 16-byte chunk a mask of the bytes that could
// make a line special ('>' / ';'), so that the ranked write knows every line's kind; one thread per line classifies
// it (the parity inside a run of '>' lines by walking back over the run; the file is read only for special lines);
// ONE pass of exclusive scans over (header flag, sequence bytes) gives every line its record and its destination
// (lines in front of the first header count as empty); the gather is output-driven: a wave writes the contiguous
// output range of 64 consecutive lines in aligned 16-byte pieces.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "host_pipeline.h"

namespace polyhip {
namespace fq {

constexpr int THREADS = 256;
constexpr uint32_t PER_BLOCK = THREADS * 16; // bytes a workgroup scans for newlines

enum { R_NREC = 0, R_CODE = 1, R_LINE = 2, R_SEQBYTES = 3, R_NLINES = 4, R_FIRSTBAD = 5, R_WORDS = 8 };

// ---- single-workgroup exclusive scan of u32 counts into u64 offsets (out[n] = total) --------
// n = n_host, or *n_dev / div when n_dev != NULL (a count that only exists on the device)
// entries beyond out[cap] are not written (the caller's buffer may hold only cap + 1 of them)
__global__ __launch_bounds__(1024) void scan_u32_kernel(const uint32_t *__restrict__ in, uint64_t n_host,
                                                       const uint64_t *__restrict__ n_dev, uint64_t div,
                                                       uint64_t *__restrict__ out, uint64_t cap)
{
    const uint64_t n = n_dev ? *n_dev / div : n_host;
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + tid;
        const uint64_t v = i < n ? in[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d)
                incl += t;
        }
        if ((tid & 63) == 63)
            wsum[tid >> 6] = incl;
        __syncthreads();
        uint64_t pre = carry;
        for (int w = 0; w < (tid >> 6); ++w)
            pre += wsum[w];
        if (i < n && i <= cap)
            out[i] = pre + incl - v;
        __syncthreads();
        if (tid == 1023)
            carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0 && n <= cap)
        out[n] = carry;
}

// ---- the same scan over SCAN_SEGS workgroups for long inputs (millions of lines): per-segment sums, a scan of
// the sums, then every workgroup scans its own segment from its offset.  n may exist only on the device.
constexpr int SCAN_SEGS = 512;

__device__ __forceinline__ void scan_segment(uint64_t n, uint64_t &lo, uint64_t &hi)
{
    const uint64_t per = ((n + SCAN_SEGS - 1) / SCAN_SEGS + 1023) / 1024 * 1024; // whole 1024-element chunks
    lo = min(n, (uint64_t)blockIdx.x * per);
    hi = min(n, lo + per);
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(const uint32_t *__restrict__ in, uint64_t n_host,
                                                        const uint64_t *__restrict__ n_dev, uint64_t div,
                                                        uint64_t *__restrict__ partial)
{
    const uint64_t n = n_dev ? *n_dev / div : n_host;
    uint64_t lo, hi;
    scan_segment(n, lo, hi);
    __shared__ uint64_t ws[16];
    uint64_t sum = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024)
        sum += in[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0)
        ws[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < 16; ++w)
            t += ws[w];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(SCAN_SEGS) void scan_offsets_kernel(uint64_t *__restrict__ partial, uint64_t n_host,
                                                                const uint64_t *__restrict__ n_dev, uint64_t div,
                                                                uint64_t *__restrict__ out, uint64_t cap)
{
    __shared__ uint64_t ws[SCAN_SEGS / 64];
    const int tid = threadIdx.x;
    const uint64_t v = partial[tid];
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d)
            incl += t;
    }
    if ((tid & 63) == 63)
        ws[tid >> 6] = incl;
    __syncthreads();
    uint64_t pre = 0;
    for (int w = 0; w < (tid >> 6); ++w)
        pre += ws[w];
    partial[tid] = pre + incl - v; // exclusive: where segment tid starts
    const uint64_t n = n_dev ? *n_dev / div : n_host;
    if (tid == SCAN_SEGS - 1 && n <= cap)
        out[n] = pre + incl;
}

__global__ __launch_bounds__(1024) void scan_apply_kernel(const uint32_t *__restrict__ in, uint64_t n_host,
                                                         const uint64_t *__restrict__ n_dev, uint64_t div,
                                                         const uint64_t *__restrict__ partial, uint64_t *__restrict__ out,
                                                         uint64_t cap)
{
    const uint64_t n = n_dev ? *n_dev / div : n_host;
    uint64_t lo, hi;
    scan_segment(n, lo, hi);
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0)
        carry = partial[blockIdx.x];
    __syncthreads();
    for (uint64_t base = lo; base < hi; base += 1024) {
        const uint64_t i = base + tid;
        const uint64_t v = i < hi ? in[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d)
                incl += t;
        }
        if ((tid & 63) == 63)
            wsum[tid >> 6] = incl;
        __syncthreads();
        uint64_t pre = carry;
        for (int w = 0; w < (tid >> 6); ++w)
            pre += wsum[w];
        if (i < hi && i <= cap)
            out[i] = pre + incl - v;
        __syncthreads();
        if (tid == 1023)
            carry = pre + incl;
        __syncthreads();
    }
}

// ---- two exclusive scans in one pass (FASTA: headers before a line, sequence bytes before a line) -----------------
// a[] holds 0 / 1 flags, b[] lengths; b[i] counts as 0 in front of the first set flag (lines before the first header carry no
// sequence: the rule that used to sit between the two scans as a kernel of its own) and is written back as 0 there.  Four
// consecutive elements per thread (one 16-byte load per array), so a chunk of 4096 elements costs one round of wave
// scans and two barriers.  outA[n] / outB[n] = the totals.
__global__ __launch_bounds__(1024) void scan2_sums_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                         const uint64_t *__restrict__ n_dev, uint64_t *__restrict__ partial)
{
    // per segment: [0] flags set, [1] sum of b behind the segment's first flag, [2] sum of b in front of it (all of b if the
    // segment has no flag), [3] index of that first flag (~0: none) -- where the file's first header is, and with it which
    // lines count, is only known once every segment has looked
    const uint64_t n = *n_dev;
    uint64_t lo, hi;
    scan_segment(n, lo, hi);
    __shared__ uint64_t ws[3][16];
    __shared__ unsigned long long firstl;
    if (threadIdx.x == 0)
        firstl = ~0ull;
    __syncthreads();
    uint64_t mine = ~0ull;
    for (uint64_t i0 = lo + 4ull * threadIdx.x; i0 < hi && mine == ~0ull; i0 += 4096) {
        const uint4 va = *reinterpret_cast<const uint4 *>(a + i0);
        const uint32_t xa[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
        for (int u = 3; u >= 0; --u)
            if (i0 + u < hi && xa[u])
                mine = i0 + u;
    }
    if (mine != ~0ull)
        atomicMin(&firstl, (unsigned long long)mine);
    __syncthreads();
    const uint64_t first = firstl;
    uint64_t sa = 0, sb = 0, sf = 0;
    for (uint64_t i0 = lo + 4ull * threadIdx.x; i0 < hi; i0 += 4096) {
        const uint4 va = *reinterpret_cast<const uint4 *>(a + i0), vb = *reinterpret_cast<const uint4 *>(b + i0);
        const uint32_t xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u < hi) {
                sa += xa[u];
                if (i0 + u < first)
                    sf += xb[u];
                else
                    sb += xb[u];
            }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sa += __shfl_xor(sa, d, 64);
        sb += __shfl_xor(sb, d, 64);
        sf += __shfl_xor(sf, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        ws[0][threadIdx.x >> 6] = sa;
        ws[1][threadIdx.x >> 6] = sb;
        ws[2][threadIdx.x >> 6] = sf;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint64_t t = 0;
        for (int w = 0; w < 16; ++w)
            t += ws[threadIdx.x][w];
        partial[threadIdx.x * SCAN_SEGS + blockIdx.x] = t;
    }
    if (threadIdx.x == 3)
        partial[3 * SCAN_SEGS + blockIdx.x] = first;
}

__global__ __launch_bounds__(2 * SCAN_SEGS) void scan2_offsets_kernel(uint64_t *__restrict__ partial,
                                                                     const uint64_t *__restrict__ n_dev,
                                                                     uint64_t *__restrict__ outA, uint64_t *__restrict__ outB,
                                                                     unsigned long long *__restrict__ first_out)
{
    __shared__ uint64_t ws[2 * SCAN_SEGS / 64];
    __shared__ unsigned long long gfirst;
    const int tid = threadIdx.x, half = tid / SCAN_SEGS, t = tid % SCAN_SEGS; // the two arrays side by side
    if (tid == 0)
        gfirst = ~0ull;
    __syncthreads();
    if (half == 0 && partial[3 * SCAN_SEGS + t] != ~0ull)
        atomicMin(&gfirst, (unsigned long long)partial[3 * SCAN_SEGS + t]);
    __syncthreads();
    const uint64_t first = gfirst;
    if (tid == 0)
        *first_out = first;
    // lines in front of the file's first header carry no sequence: a segment in front of it adds nothing, the segment that
    // holds it what lies behind its own first flag (which is the file's), the segments behind it all they have
    uint64_t v;
    if (half == 0) {
        v = partial[t];
    } else {
        const uint64_t n = *n_dev;
        const uint64_t per = ((n + SCAN_SEGS - 1) / SCAN_SEGS + 1023) / 1024 * 1024; // scan_segment's
        const uint64_t lo = min(n, (uint64_t)t * per);
        const uint64_t after = partial[SCAN_SEGS + t], before = partial[2 * SCAN_SEGS + t];
        if (first == ~0ull)
            v = 0;
        else if (lo > first)
            v = after + before;
        else if (partial[3 * SCAN_SEGS + t] == first)
            v = after;
        else
            v = 0;
    }
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t x = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d)
            incl += x;
    }
    if ((tid & 63) == 63)
        ws[tid >> 6] = incl;
    __syncthreads();
    uint64_t pre = 0;
    for (int w = half * (SCAN_SEGS / 64); w < (tid >> 6); ++w)
        pre += ws[w];
    __syncthreads(); // every thread has read what it needs of partial[]
    partial[tid] = pre + incl - v; // exclusive: where segment t starts
    if (t == SCAN_SEGS - 1)
        (half ? outB : outA)[*n_dev] = pre + incl;
}

__global__ __launch_bounds__(1024) void scan2_apply_kernel(const uint32_t *__restrict__ a, uint32_t *__restrict__ b,
                                                          const uint64_t *__restrict__ n_dev,
                                                          const unsigned long long *__restrict__ first_p,
                                                          const uint64_t *__restrict__ partial, uint64_t *__restrict__ outA,
                                                          uint64_t *__restrict__ outB)
{
    const uint64_t n = *n_dev, first = *first_p;
    uint64_t lo, hi;
    scan_segment(n, lo, hi);
    __shared__ uint64_t wsum[2][16];
    const int tid = threadIdx.x;
    uint64_t carryA = partial[blockIdx.x], carryB = partial[SCAN_SEGS + blockIdx.x];
    for (uint64_t base = lo; base < hi; base += 4096) {
        const uint64_t i0 = base + 4ull * tid;
        uint32_t xa[4] = {0u, 0u, 0u, 0u}, xb[4] = {0u, 0u, 0u, 0u};
        if (i0 < hi) {
            const uint4 va = *reinterpret_cast<const uint4 *>(a + i0), vb = *reinterpret_cast<const uint4 *>(b + i0);
            xa[0] = va.x, xa[1] = va.y, xa[2] = va.z, xa[3] = va.w;
            xb[0] = vb.x, xb[1] = vb.y, xb[2] = vb.z, xb[3] = vb.w;
            bool cut = false;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u >= hi) {
                    xa[u] = 0;
                    xb[u] = 0;
                } else if (i0 + u < first && xb[u]) {
                    xb[u] = 0;
                    cut = true;
                }
            if (cut) { // (rare: sequence-like lines in front of the first header)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < hi && i0 + u < first)
                        b[i0 + u] = 0;
            }
        }
        const uint64_t ta = (uint64_t)xa[0] + xa[1] + xa[2] + xa[3], tb = (uint64_t)xb[0] + xb[1] + xb[2] + xb[3];
        uint64_t ia = ta, ib = tb; // inclusive over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
            if ((tid & 63) >= d) {
                ia += ua;
                ib += ub;
            }
        }
        __syncthreads(); // wsum of the chunk before is used up
        if ((tid & 63) == 63) {
            wsum[0][tid >> 6] = ia;
            wsum[1][tid >> 6] = ib;
        }
        __syncthreads();
        uint64_t pa = carryA + ia - ta, pb = carryB + ib - tb, totA = 0, totB = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < (tid >> 6)) {
                pa += wsum[0][w];
                pb += wsum[1][w];
            }
            totA += wsum[0][w];
            totB += wsum[1][w];
        }
        carryA += totA;
        carryB += totB;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u < hi) {
                outA[i0 + u] = pa;
                outB[i0 + u] = pb;
                pa += xa[u];
                pb += xb[u];
            }
    }
}

// ---- newline compaction, 16 bytes per lane ---------------------------------------------------------
// A workgroup takes one 4096-byte window of the file's 16-byte-ALIGNED address space (the file itself may start
// anywhere: `mis` = its offset inside the first window); every lane loads one uint4 and turns it into a 16-bit
// mask of the bytes that are '\n' and lie inside the file.

// 0x01 in every byte of x that is NOT zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x)
{
    return ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7) & 0x01010101u;
}

// low 16 bits: the newlines of one aligned 16-byte chunk; SPEC: high 16 bits: its ':' ';' '>' '?' bytes -- one compare
// finds the four (0x3A | bits 0 and 2), and only '>' and ';' make a FASTA line anything but sequence when they stand
// first: whoever acts on the bit looks at the byte itself.
// The four flag bits of a dword (bits 0, 8, 16, 24) are gathered into a nibble by one multiply: every partial product
// of 0x00204081 lands on a bit of its own, the nibble stands at bits 21..24; the second set of flags rides along four
// bits higher (nibble at 25..28) -- the per-bit shifts of the plain gather made the FASTA count pass compute-bound
// (0.154 ms against 0.10 ms for the newlines alone).
template <bool SPEC>
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *__restrict__ abase, uint64_t chunk, uint32_t mis,
                                                   uint64_t nbytes)
{
    // chunk = index of the aligned 16-byte chunk; file-relative position of its first byte = chunk * 16 - mis
    const int64_t p0 = (int64_t)(chunk * 16) - (int64_t)mis;
    if (p0 >= (int64_t)nbytes || p0 + 16 <= 0)
        return 0u;
    const uint4 v = *reinterpret_cast<const uint4 *>(abase + chunk * 16);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t other = 0; // bytes that are NOT a newline (low half) / NOT one of the four (high half)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t z = nonzero_bytes(w[q] ^ 0x0A0A0A0Au);
        if (SPEC)
            z |= nonzero_bytes((w[q] & 0xFAFAFAFAu) ^ 0x3A3A3A3Au) << 4;
        const uint32_t g = z * 0x00204081u;
        other |= ((g >> 21) & 0xFu) << (4 * q);
        if (SPEC)
            other |= ((g >> 25) & 0xFu) << (16 + 4 * q);
    }
    uint32_t keep = 0xFFFFu;
    if (p0 < 0)
        keep &= 0xFFFFu << (uint32_t)(-p0);
    if (p0 + 16 > (int64_t)nbytes)
        keep &= 0xFFFFu >> (uint32_t)(p0 + 16 - (int64_t)nbytes);
    return ~other & (SPEC ? (keep | (keep << 16)) : keep);
}

// newlines per 4096-byte window
// (also leaves every 16-byte chunk's masks -- 2 (FASTQ) or 4 (FASTA) bytes per 16 of file -- so that the ranked write below
// does not read the file a second time: 51 MB instead of 408 MB for the bench's image)
constexpr int COUNT_SUB = 4; // 4 KB windows per workgroup of the count pass
template <bool SPEC, class MaskT>
__global__ __launch_bounds__(THREADS) void count_newlines_kernel(const uint8_t *__restrict__ abase, uint32_t mis,
                                                                uint64_t nbytes, uint64_t nblocks, uint32_t *__restrict__ counts,
                                                                MaskT *__restrict__ masks)
{
    // COUNT_SUB windows of 4 KB per workgroup, their four 16-byte loads per lane in flight together (one window per
    // workgroup was 100k workgroups of one load each for a 400 MB image: 3.5 TB/s); a count per 4 KB window as before
    __shared__ uint32_t ws[COUNT_SUB][THREADS / 64];
    uint32_t mk[COUNT_SUB];
#pragma unroll
    for (int q = 0; q < COUNT_SUB; ++q) {
        const uint64_t blk = (uint64_t)blockIdx.x * COUNT_SUB + q;
        mk[q] = blk < nblocks ? newline_mask16<SPEC>(abase, blk * THREADS + threadIdx.x, mis, nbytes) : 0u;
    }
#pragma unroll
    for (int q = 0; q < COUNT_SUB; ++q) {
        const uint64_t blk = (uint64_t)blockIdx.x * COUNT_SUB + q;
        if (blk < nblocks)
            masks[blk * THREADS + threadIdx.x] = (MaskT)mk[q];
        uint32_t c = (uint32_t)__popc(mk[q] & 0xFFFFu);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            c += (uint32_t)__shfl_xor((int)c, d, 64);
        if ((threadIdx.x & 63) == 0)
            ws[q][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x < COUNT_SUB) {
        const uint64_t blk = (uint64_t)blockIdx.x * COUNT_SUB + threadIdx.x;
        if (blk < nblocks)
            counts[blk] = ws[threadIdx.x][0] + ws[threadIdx.x][1] + ws[threadIdx.x][2] + ws[threadIdx.x][3];
    }
}

// line_end[k] = file-relative position of the k-th '\n'.  A wave per 4096-byte window (the unit the counts were scanned in), four
// 16-byte chunks per lane: one scan step per 64 bytes of file instead of one per 16 (a thread per chunk: 0.07 ms for the
// bench's FASTA image, as long as reading the image itself took).
// SPEC (FASTA): kind[k] = 1 if line k starts with '>' or ';' -- the byte behind the newline that ends line k - 1, read off
// the chunk masks, so that the per-line classification does not have to gather every line's first byte from the file.
template <bool SPEC, class MaskT>
__global__ __launch_bounds__(THREADS) void write_newlines_kernel(const MaskT *__restrict__ masks, uint64_t nchunks, uint32_t mis,
                                                                const uint64_t *__restrict__ block_off, uint64_t nwindows,
                                                                uint64_t *__restrict__ line_end, uint8_t *__restrict__ kind)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t window = (uint64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (window >= nwindows)
        return;
    const uint64_t chunk0 = window * THREADS + 4u * lane; // THREADS chunks per window
    uint32_t m[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
        m[j] = (chunk0 + j < nchunks && (j < 4 || SPEC)) ? (uint32_t)masks[chunk0 + j] : 0u;
    const uint32_t c = (uint32_t)(__popc(m[0] & 0xFFFFu) + __popc(m[1] & 0xFFFFu) + __popc(m[2] & 0xFFFFu) + __popc(m[3] & 0xFFFFu));
    uint32_t incl = c; // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= (uint32_t)d)
            incl += t;
    }
    uint64_t at = block_off[window] + incl - c;
    if (SPEC && chunk0 == 0) // line 0 starts with the file
        kind[0] = (uint8_t)((m[0] >> (16u + mis)) & 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t p0 = (int64_t)((chunk0 + j) * 16) - (int64_t)mis;
        uint32_t nl = m[j] & 0xFFFFu;
        const uint32_t sp = SPEC ? ((m[j] >> 16) | ((m[j + 1] >> 16) << 16)) : 0u; // special bytes of this chunk and the next
        while (nl) {
            const int k = __builtin_ctz(nl);
            nl &= nl - 1u;
            if (SPEC)
                kind[at + 1] = (uint8_t)((sp >> (k + 1)) & 1u);
            line_end[at++] = (uint64_t)(p0 + k);
        }
    }
}

// `G` consecutive lanes (this lane is number `gl` of its group) copy len bytes from src to dst, both in global memory
// and of any alignment: whole destination dwords funnelled out of two source dwords, bytes only at the two ends
template <int G>
__device__ __forceinline__ void group_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t len, int gl)
{
    const uint32_t head = (uint32_t)min(len, (uint64_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
    if ((uint32_t)gl < head)
        dst[gl] = src[gl];
    dst += head;
    src += head;
    len -= head;
    const uint64_t ndw = len >> 2;
    const uint32_t sh = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t *__restrict__ s4 = reinterpret_cast<const uint32_t *>(src - sh);
    uint32_t *__restrict__ d4 = reinterpret_cast<uint32_t *>(dst);
    if (sh == 0) {
        for (uint64_t w = gl; w < ndw; w += G)
            d4[w] = s4[w];
    } else {
        for (uint64_t w = gl; w < ndw; w += G)
            d4[w] = __builtin_amdgcn_alignbyte(s4[w + 1], s4[w], sh); // s4[w + 1] holds a byte of the source: in bounds
    }
    const uint32_t tail = (uint32_t)(len & 3u);
    if ((uint32_t)gl < tail)
        dst[4 * ndw + gl] = src[4 * ndw + gl];
}

// identifier-line checks of fastq.go:155-167: 0, or 5 (empty line), 6 (field without '=')
__device__ uint32_t header_panics(const uint8_t *__restrict__ file, uint64_t s, uint64_t e)
{
    if (e == s)
        return 5;
    // tokens after the first ' ' must each contain '=' (optionalSplits[1])
    uint64_t p = s;
    while (p < e && file[p] != ' ')
        ++p;
    while (p < e) { // file[p] == ' ': a token starts at p + 1
        ++p;
        bool eq = false;
        while (p < e && file[p] != ' ') {
            eq |= file[p] == '=';
            ++p;
        }
        if (!eq)
            return 6;
    }
    return 0;
}

// one thread per complete 4-line record: validate, measure the sequence
__global__ __launch_bounds__(THREADS) void records_kernel(const uint8_t *__restrict__ file,
                                                         const uint64_t *__restrict__ line_end,
                                                         const uint64_t *__restrict__ nlines_dev,
                                                         uint32_t *__restrict__ seq_len, uint64_t *__restrict__ seq_start,
                                                         uint64_t *__restrict__ rec_start, uint64_t max_records,
                                                         unsigned long long *__restrict__ res)
{
    const uint64_t nrec = *nlines_dev / 4; // complete 4-line records
    for (uint64_t r = (uint64_t)blockIdx.x * THREADS + threadIdx.x; r < nrec; r += (uint64_t)gridDim.x * THREADS) {
    const uint64_t l0 = r == 0 ? 0 : line_end[4 * r - 1] + 1;
    const uint64_t e0 = line_end[4 * r], e1 = line_end[4 * r + 1], e2 = line_end[4 * r + 2], e3 = line_end[4 * r + 3];
    uint32_t code = header_panics(file, l0, e0);
    if (!code && e1 == e0 + 1)
        code = 2;
    if (!code && e3 == e2 + 1)
        code = 3;
    if (!code && file[l0] != '@')
        code = 1;
    seq_len[r] = (uint32_t)(e1 - e0 - 1);
    seq_start[r] = e0 + 1;
    if (rec_start && r < max_records)
        rec_start[r] = l0;
    if (code) // the first bad record wins; (record << 8 | code) orders by record
        atomicMin(&res[R_FIRSTBAD], (unsigned long long)((r << 8) | code));
    }
}

// n_records, error code / line, and the trailing partial record
__global__ void finish_kernel(const uint8_t *__restrict__ file, uint64_t nbytes, const uint64_t *__restrict__ line_end,
                              const uint64_t *__restrict__ nlines_dev, const uint64_t *__restrict__ offsets,
                              uint64_t max_records, unsigned long long *__restrict__ res)
{
    const uint64_t nlines = *nlines_dev;
    const uint64_t nrec = nlines / 4;
    uint64_t n = nrec, code = 0, line = 0;
    const unsigned long long fb = res[R_FIRSTBAD];
    if (fb != ~0ull) {
        n = fb >> 8;
        code = fb & 0xFF;
        const uint64_t r = n;
        line = code == 2 ? 4 * r + 2 : (code == 1 || code == 3) ? 4 * r + 4 : 4 * r + 1;
    } else {
        // bytes after the last complete record: j complete lines + maybe an unterminated one
        const uint64_t tail0 = nrec == 0 ? 0 : line_end[4 * nrec - 1] + 1;
        if (tail0 < nbytes) {
            const uint64_t j = nlines - 4 * nrec; // 0..3 complete lines
            const uint64_t e0 = j >= 1 ? line_end[4 * nrec] : nbytes;
            if (j >= 1)
                code = header_panics(file, tail0, e0);
            line = 4 * nrec + 1;
            if (!code && j >= 2 && line_end[4 * nrec + 1] == e0 + 1) {
                code = 2;
                line = 4 * nrec + 2;
            }
            if (!code) { // the (j+1)-th ReadSlice hits EOF; the message names parser.line + 1
                code = 4;
                line = 4 * nrec + j + 2;
            }
        }
    }
    if (n > max_records) {
        n = max_records;
        code = 7; // caller's buffers hold fewer records than the file
        line = 0;
    }
    res[R_NREC] = n;
    res[R_CODE] = code;
    res[R_LINE] = line;
    res[R_SEQBYTES] = offsets[n];
    res[R_NLINES] = nlines;
}

// one wave per record (grid-stride): sequence bytes -> packed buffer, whole dwords
__global__ __launch_bounds__(THREADS) void gather_kernel(const uint8_t *__restrict__ file,
                                                        const uint64_t *__restrict__ seq_start,
                                                        const uint64_t *__restrict__ offsets,
                                                        const unsigned long long *__restrict__ res,
                                                        uint8_t *__restrict__ seqs)
{
    const uint64_t n = res[R_NREC];
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * THREADS + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * (THREADS / 64);
    for (uint64_t r = wave; r < n; r += nwaves) {
        const uint64_t dst = offsets[r];
        group_copy<64>(seqs + dst, file + seq_start[r], offsets[r + 1] - dst, lane);
    }
}

// ---- FASTA ----------------------------------------------------------------------------------------
enum { F_NREC = 0, F_CODE = 1, F_SEQBYTES = 2, F_NHEADERS = 3, F_FIRSTEMPTY = 4, F_FIRSTHDR = 5, F_WORDS = 8 };

// per complete line: is it a header?  how many sequence bytes does it contribute?
__global__ __launch_bounds__(THREADS) void fasta_classify_kernel(const uint8_t *__restrict__ file,
                                                                const uint64_t *__restrict__ line_end,
                                                                const uint64_t *__restrict__ nlines_dev,
                                                                const uint8_t *__restrict__ kind,
                                                                uint32_t *__restrict__ is_header,
                                                                uint32_t *__restrict__ seq_len)
{
    const uint64_t nl = *nlines_dev;
    for (uint64_t k = (uint64_t)blockIdx.x * THREADS + threadIdx.x; k < nl; k += (uint64_t)gridDim.x * THREADS) {
    const uint64_t start = k == 0 ? 0 : line_end[k - 1] + 1;
    const uint64_t len = line_end[k] - start;
    // only a line that starts with '>' or ';' (its kind, from the chunk masks) has its first byte looked at: one line in
    // fifty of a usual file, instead of a gather of every line's first byte (0.3 GB of sectors for the bench's image)
    const uint8_t b = (len && kind[k]) ? file[start] : 0;
    bool header = false;
    if (len && b == '>') {
        // position inside the run of consecutive '>' lines that ends here: odd = header
        uint64_t run = 1, j = k;
        while (j > 0) {
            const uint64_t ps = j == 1 ? 0 : line_end[j - 2] + 1;
            const uint64_t pl = line_end[j - 1] - ps;
            if (pl == 0 || file[ps] != '>')
                break;
            ++run;
            --j;
        }
        header = (run & 1) != 0;
    }
    const bool skippable = len == 0 || b == ';';
    is_header[k] = header ? 1u : 0u;
    seq_len[k] = (!header && !skippable) ? (uint32_t)len : 0u;
    }
}

// offsets[r] = sequence bytes before record r's header; first record without sequence
__global__ __launch_bounds__(THREADS) void fasta_offsets_kernel(const uint64_t *__restrict__ nlines_dev,
                                                               const uint32_t *__restrict__ is_header,
                                                               const uint64_t *__restrict__ hrank,
                                                               const uint64_t *__restrict__ dst,
                                                               const uint64_t *__restrict__ line_end,
                                                               uint64_t *__restrict__ offsets,
                                                               uint64_t *__restrict__ rec_start)
{
    const uint64_t nl = *nlines_dev;
    for (uint64_t k = (uint64_t)blockIdx.x * THREADS + threadIdx.x; k < nl; k += (uint64_t)gridDim.x * THREADS) {
        if (is_header[k]) {
            offsets[hrank[k]] = dst[k];
            if (rec_start)
                rec_start[hrank[k]] = k == 0 ? 0 : line_end[k - 1] + 1;
        }
        if (k == nl - 1)
            offsets[hrank[nl]] = dst[nl]; // hrank[nl] = number of headers, dst[nl] = all sequence bytes
    }
}

__global__ __launch_bounds__(THREADS) void fasta_empty_kernel(const uint64_t *__restrict__ hrank,
                                                             const uint64_t *__restrict__ nlines_dev,
                                                             const uint64_t *__restrict__ offsets,
                                                             unsigned long long *__restrict__ res)
{
    const uint64_t H = hrank[*nlines_dev];
    for (uint64_t r = (uint64_t)blockIdx.x * THREADS + threadIdx.x; r < H; r += (uint64_t)gridDim.x * THREADS)
        if (offsets[r + 1] == offsets[r])
            atomicMin(&res[F_FIRSTEMPTY], (unsigned long long)r);
}

__global__ void fasta_finish_kernel(const uint8_t *__restrict__ file, uint64_t nbytes,
                                    const uint64_t *__restrict__ line_end, const uint64_t *__restrict__ nlines_dev,
                                    const uint64_t *__restrict__ hrank, const uint64_t *__restrict__ offsets,
                                    uint64_t max_records, unsigned long long *__restrict__ res)
{
    const uint64_t nl = *nlines_dev;
    const uint64_t H = nl ? hrank[nl] : 0;
    // the unterminated tail line, if any: non-skippable = longer than one byte and not a comment (:169)
    const uint64_t tail0 = nl ? line_end[nl - 1] + 1 : 0;
    bool tail_counts = nbytes - tail0 > 1 && file[tail0] != ';';
    const bool tail_nonskippable = tail_counts; // any such tail makes a "no header" error wrap io.EOF
    bool lone_gt = false; // the tail is the single byte '>' and starts a record of its own
    if (nbytes > tail0 && file[tail0] == '>') {
        // a '>' tail line is a (nameless, dropped) NEW record unless it sits at an even position of a
        // run of '>' lines, where it is sequence of the last record like any other line
        uint64_t run = 1, j = nl;
        while (j > 0) {
            const uint64_t ps = j == 1 ? 0 : line_end[j - 2] + 1;
            const uint64_t pl = line_end[j - 1] - ps;
            if (pl == 0 || file[ps] != '>')
                break;
            ++run;
            --j;
        }
        if (run & 1) {
            // The records before it are complete.  A longer tail is read as a sequence line with io.EOF
            // while still looking for a name: the error wraps EOF and is dropped.  The one-byte tail ">"
            // is skippable (:169), so the same parse ends with a plain "did not find fasta start" (:223).
            lone_gt = nbytes - tail0 == 1;
            tail_counts = false;
        }
    }
    uint64_t n = H, code = 0;
    unsigned long long fe = res[F_FIRSTEMPTY];
    if (tail_counts && H > 0 && fe == H - 1)
        fe = ~0ull; // the tail line is this record's sequence; it is dropped below, not an error
    if (fe != ~0ull) {
        n = fe;
        code = 2;
    } else if (H == 0) {
        code = (nbytes > 0 && !tail_nonskippable) ? 1 : 0;
    } else if (tail_counts) {
        n = H - 1; // ParseNext returns the last record with io.EOF; ParseN drops it
    } else if (lone_gt) {
        code = 1;
    }
    if (n > max_records) {
        n = max_records;
        code = 7;
    }
    res[F_NREC] = n;
    res[F_CODE] = code;
    res[F_SEQBYTES] = H ? offsets[n] : 0;
    res[F_NHEADERS] = H;
}

// ---- FASTA gather: sequence lines of the kept records -> packed buffer ------------------------------------------------
// OUTPUT-driven (round 3).  Consecutive sequence lines land back to back in the output, so a wave takes 64 consecutive
// lines (lane l loads line l's start, length and destination: three coalesced loads), and then produces the whole
// contiguous output range of those lines with ALIGNED, fully coalesced 16-byte stores -- 1 KB per store instruction
// whatever the line width.  A lane finds the line its output dword belongs to from the block's mean line length, corrected by a
// short walk over the 64 destinations (in LDS), and reads the four source bytes with one unaligned dword load when they lie in one line (19 of 20
// dwords of an 80-column file), byte by byte across a line end.  The per-line form it replaces (8 lanes per line, whole
// dwords funnelled to the destination's alignment) wrote 32-byte pieces at line-sized strides and stayed at 2.3 TB/s of
// traffic however many lines a group kept in flight (0.36 ms for the 408 MB image; four lines per group: the same);
// a workgroup-per-file-window compaction through LDS drowned in per-byte arithmetic (28 vector instructions per byte).
// POLYHIP_FASTA_STREAM=0 keeps the per-line kernel (testing aid; tests/test_fasta_gpu.py runs both).
__global__ __launch_bounds__(THREADS) void fasta_gather_stream_kernel(const uint8_t *__restrict__ file,
                                                                     const uint64_t *__restrict__ line_end,
                                                                     const uint64_t *__restrict__ nlines_dev,
                                                                     const uint32_t *__restrict__ seq_len,
                                                                     const uint64_t *__restrict__ dst,
                                                                     const unsigned long long *__restrict__ res,
                                                                     uint8_t *__restrict__ seqs)
{
    __shared__ uint64_t dls[THREADS / 64][64], els[THREADS / 64][64], sls[THREADS / 64][64];
    const uint64_t nl = *nlines_dev, total = res[F_SEQBYTES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t *dl = dls[wv], *el = els[wv], *sl = sls[wv];
    const uint64_t wave = ((uint64_t)blockIdx.x * THREADS + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * (THREADS / 64);
    for (uint64_t kA = wave * 64; kA < nl; kA += nwaves * 64) {
        const uint64_t k = kA + lane;
        uint64_t len = 0, d = ~0ull, st = 0;
        if (k < nl) {
            len = seq_len[k];
            d = dst[k];
            st = k == 0 ? 0 : line_end[k - 1] + 1;
            if (d >= total) { // behind the kept records
                len = 0;
                d = ~0ull;
            }
        }
        const uint64_t kept = __ballot(len != 0);
        if (kept == 0ull)
            continue; // wave-uniform
        dl[lane] = d;
        el[lane] = d + len; // (d = ~0: never looked at)
        sl[lane] = st;
        const int first = __builtin_ctzll(kept), last = 63 - __builtin_clzll(kept);
        const uint64_t Dlo = __shfl(d, first, 64), Dhi = __shfl(d + len, last, 64);
        __builtin_amdgcn_wave_barrier(); // the LDS rows are this wave's own; DS operations of a wave execute in order
        // line of output position o: the last one whose destination is <= o (zero-length lines in front of it share its
        // destination and lose; lines behind the kept records carry ~0)
        // (a guess from the block's mean line length -- exact to within a line or two for the usual fixed-width files --
        // corrected by walking the table; the walk is what makes it exact for any widths)
        const float per_byte = (float)(last - first + 1) / (float)(Dhi - Dlo);
        auto line_of = [&](uint64_t o) {
            int kk = first + min(last - first, (int)((float)(o - Dlo) * per_byte));
            while (dl[kk] > o)
                --kk;
            while (kk < last && dl[kk + 1] <= o)
                ++kk;
            return kk;
        };
        auto byte_at = [&](uint64_t o, int kk) -> uint32_t { // o in [Dlo, Dhi); kk = a line at or before o's
            while (o >= el[kk])
                ++kk;
            return file[sl[kk] + (o - dl[kk])];
        };
        uint8_t *out = seqs;
        // first output position on a 16-byte boundary of the destination, and the whole 16-byte pieces from there
        const uint64_t oa = Dlo + ((16u - (uint32_t)((uintptr_t)(out + Dlo) & 15u)) & 15u);
        if (oa >= Dhi) { // not one aligned piece: bytes
            if (Dlo + lane < Dhi)
                out[Dlo + lane] = (uint8_t)byte_at(Dlo + lane, first);
            continue;
        }
        if (Dlo + lane < oa) // head bytes (at most 15)
            out[Dlo + lane] = (uint8_t)byte_at(Dlo + lane, first);
        const uint64_t n16 = (Dhi - oa) >> 4;
        for (uint64_t w = lane; w < n16; w += 64) {
            const uint64_t o = oa + 16 * w;
            int kk = line_of(o);
            uint4 piece;
            if (o + 16 <= el[kk]) {
                __builtin_memcpy(&piece, file + sl[kk] + (o - dl[kk]), 16); // one line: (unaligned) dword loads
            } else { // a line ends inside the piece: dword by dword, byte by byte across the end
                uint32_t wd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t oq = o + 4 * q;
                    while (oq >= el[kk])
                        ++kk;
                    if (oq + 4 <= el[kk])
                        __builtin_memcpy(&wd[q], file + sl[kk] + (oq - dl[kk]), 4);
                    else
                        wd[q] = byte_at(oq, kk) | (byte_at(oq + 1, kk) << 8) | (byte_at(oq + 2, kk) << 16) | (byte_at(oq + 3, kk) << 24);
                }
                piece = make_uint4(wd[0], wd[1], wd[2], wd[3]);
            }
            *reinterpret_cast<uint4 *>(out + o) = piece;
        }
        const uint64_t ot = oa + 16 * n16; // tail bytes (at most 15)
        if (ot + lane < Dhi)
            out[ot + lane] = (uint8_t)byte_at(ot + lane, line_of(ot + lane));
        __builtin_amdgcn_wave_barrier(); // the rows are rewritten by the next block of lines
    }
}

// 8 lanes per line (grid-stride; 4 / 8 / 16 / 32 lanes measured 0.80 / 0.75 / 0.89 / 1.07 ms on 80-column lines): sequence
// lines of the kept records -> packed buffer, whole dwords.
// Round 3: a group takes GU lines per iteration and keeps them in flight TOGETHER -- all their (line_end, seq_len, dst)
// loads are issued back to back, then all their source dwords, then the stores.  One line per iteration was two dependent
// memory round trips for 80 bytes per group (the kernel moved 0.8 GB at 2.3 TB/s with every wave slot taken: latency
// bound); a workgroup-per-file-window version that compacted the kept bytes through LDS was tried first and lost to its
// own per-byte arithmetic (28 vector instructions per byte, 0.36-0.84 ms).
#ifndef PH_FASTA_G
#define PH_FASTA_G 8
#endif
#ifndef PH_FASTA_GU
#define PH_FASTA_GU 4
#endif
__global__ __launch_bounds__(THREADS) void fasta_gather_kernel(const uint8_t *__restrict__ file,
                                                              const uint64_t *__restrict__ line_end,
                                                              const uint64_t *__restrict__ nlines_dev,
                                                              const uint32_t *__restrict__ seq_len,
                                                              const uint64_t *__restrict__ dst,
                                                              const unsigned long long *__restrict__ res,
                                                              uint8_t *__restrict__ seqs)
{
    constexpr int G = PH_FASTA_G, GU = PH_FASTA_GU;
    const uint64_t nl = *nlines_dev;
    const uint64_t total = res[F_SEQBYTES];
    const int gl = threadIdx.x & (G - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * THREADS + threadIdx.x) / G, ngroups = (uint64_t)gridDim.x * (THREADS / G);
    for (uint64_t k0 = group * GU; k0 < nl; k0 += ngroups * GU) { // GU CONSECUTIVE lines: their metadata shares cache lines
        // ---- all the lines' metadata
        uint64_t len[GU], d[GU], st[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const uint64_t k = k0 + u;
            len[u] = 0;
            d[u] = 0;
            st[u] = 0;
            if (k < nl) {
                len[u] = seq_len[k];
                d[u] = dst[k];
                st[u] = k == 0 ? 0 : line_end[k - 1] + 1;
            }
        }
        // ---- geometry of each copy: head bytes up to the destination's dword boundary, whole dwords, tail bytes
        uint32_t head[GU], sh[GU];
        uint64_t ndw[GU];
        const uint32_t *s4[GU];
        uint32_t *d4[GU];
        uint64_t maxdw = 0;
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            if (len[u] == 0 || d[u] >= total)
                len[u] = 0;
            uint8_t *dp = seqs + d[u];
            const uint8_t *sp = file + st[u];
            head[u] = (uint32_t)min(len[u], (uint64_t)((4u - (uint32_t)((uintptr_t)dp & 3u)) & 3u));
            if ((uint32_t)gl < head[u])
                dp[gl] = sp[gl];
            dp += head[u];
            sp += head[u];
            const uint64_t rest = len[u] - head[u];
            ndw[u] = rest >> 2;
            sh[u] = (uint32_t)((uintptr_t)sp & 3u);
            s4[u] = reinterpret_cast<const uint32_t *>(sp - sh[u]);
            d4[u] = reinterpret_cast<uint32_t *>(dp);
            const uint32_t tail = (uint32_t)(rest & 3u);
            if ((uint32_t)gl < tail)
                dp[4 * ndw[u] + gl] = sp[4 * ndw[u] + gl];
            maxdw = max(maxdw, ndw[u]);
        }
        // ---- the dwords: every line's loads of a round before any store
        for (uint64_t w = gl; w < maxdw; w += G) {
            uint32_t lo[GU], hi[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                lo[u] = hi[u] = 0;
                if (w < ndw[u]) {
                    lo[u] = s4[u][w];
                    if (sh[u])
                        hi[u] = s4[u][w + 1]; // holds a byte of the source: in bounds
                }
            }
#pragma unroll
            for (int u = 0; u < GU; ++u)
                if (w < ndw[u])
                    d4[u][w] = sh[u] ? __builtin_amdgcn_alignbyte(hi[u], lo[u], sh[u]) : lo[u];
        }
    }
}

struct Layout {
    uint64_t nblocks, max_lines;
    size_t off_counts, off_blockoff, off_lineend, off_seqlen, off_seqstart, off_res, off_scanpart, off_masks, total;
    size_t off_ishdr, off_hrank, off_dst, off_kind; // FASTA (per line)
};

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static Layout layout(uint64_t nbytes)
{
    Layout L;
    L.nblocks = (nbytes + PER_BLOCK - 1) / PER_BLOCK + 1; // + 1: the file may start anywhere inside its first window
    L.max_lines = nbytes; // every byte a newline
    size_t o = 0;
    L.off_res = o; o += al(R_WORDS * 8);
    L.off_scanpart = o; o += al(4 * SCAN_SEGS * 8);
    L.off_counts = o; o += al(L.nblocks * 4);
    L.off_masks = o; o += al(L.nblocks * (size_t)THREADS * 4 + 16); // 2 bytes per chunk (FASTQ), 4 (FASTA)
    L.off_blockoff = o; o += al((L.nblocks + 1) * 8);
    L.off_lineend = o; o += al((L.max_lines + 1) * 8);
    L.off_seqlen = o; o += al((L.max_lines / 4 + 1) * 4);
    L.off_seqstart = o; o += al((L.max_lines / 4 + 1) * 8);
    L.total = o;
    L.off_ishdr = L.off_hrank = L.off_dst = L.off_kind = 0;
    return L;
}

// FASTA needs per-LINE arrays (a line can be as short as one byte)
static Layout layout_fasta(uint64_t nbytes)
{
    Layout L = layout(nbytes);
    size_t o = L.off_seqlen;
    L.off_seqlen = o; o += al((L.max_lines + 1) * 4);
    L.off_ishdr = o; o += al((L.max_lines + 1) * 4);
    L.off_hrank = o; o += al((L.max_lines + 2) * 8);
    L.off_dst = o; o += al((L.max_lines + 2) * 8);
    L.off_kind = o; o += al(L.max_lines + 2);
    L.off_seqstart = 0;
    L.total = o;
    return L;
}

} // namespace fq
} // namespace polyhip

using namespace polyhip;

// exclusive scan of `in` (n = n_host, or *n_dev / div) into out[0..n]; `upper` bounds n on the host: long
// inputs go through SCAN_SEGS workgroups, whose partial sums live in `partial` (SCAN_SEGS words of the workspace)
static int scan_u32(const uint32_t *in, uint64_t n_host, const uint64_t *n_dev, uint64_t div, uint64_t upper,
                    uint64_t *out, uint64_t *partial, hipStream_t st, uint64_t cap = ~0ull)
{
    if (upper <= 32768) {
        hipLaunchKernelGGL(fq::scan_u32_kernel, dim3(1), dim3(1024), 0, st, in, n_host, n_dev, div, out, cap);
        return POLYHIP_OK;
    }
    hipLaunchKernelGGL(fq::scan_sums_kernel, dim3(fq::SCAN_SEGS), dim3(1024), 0, st, in, n_host, n_dev, div, partial);
    hipLaunchKernelGGL(fq::scan_offsets_kernel, dim3(1), dim3(fq::SCAN_SEGS), 0, st, partial, n_host, n_dev, div, out, cap);
    hipLaunchKernelGGL(fq::scan_apply_kernel, dim3(fq::SCAN_SEGS), dim3(1024), 0, st, in, n_host, n_dev, div, partial, out, cap);
    return POLYHIP_OK;
}

extern "C" {

size_t polyhip_fastq_workspace_bytes(uint64_t nbytes) { return fq::layout(nbytes).total; }

int polyhip_fastq_pack_dev(const uint8_t *d_file, uint64_t nbytes, uint8_t *d_seqs, uint64_t *d_offsets,
                           uint64_t *d_rec_start, uint64_t max_records, uint64_t *d_result, void *d_work,
                           size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(d_result && d_offsets && d_work && (d_file || nbytes == 0) && (d_seqs || nbytes == 0),
               "polyhip_fastq_pack: null pointer");
    const fq::Layout L = fq::layout(nbytes);
    PH_REQUIRE(work_bytes >= L.total, "polyhip_fastq_pack: workspace too small (%zu < %zu)", work_bytes, L.total);
    PH_REQUIRE(L.nblocks < (1ull << 31), "polyhip_fastq_pack: file too large for one call");
    hipStream_t st = as_stream(stream);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    unsigned long long *res = reinterpret_cast<unsigned long long *>(w + L.off_res);
    uint64_t *scanpart = reinterpret_cast<uint64_t *>(w + L.off_scanpart);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + L.off_counts);
    uint64_t *blockoff = reinterpret_cast<uint64_t *>(w + L.off_blockoff);
    uint64_t *line_end = reinterpret_cast<uint64_t *>(w + L.off_lineend);
    uint32_t *seq_len = reinterpret_cast<uint32_t *>(w + L.off_seqlen);
    uint64_t *seq_start = reinterpret_cast<uint64_t *>(w + L.off_seqstart);

    PH_HIP(hipMemsetAsync(res, 0, fq::R_WORDS * 8, st));
    PH_HIP(hipMemsetAsync(res + fq::R_FIRSTBAD, 0xFF, 8, st));
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(d_file) & 15u);
    const uint8_t *abase = d_file - mis; // 16-byte aligned; bytes in front of the file are masked out
    hipLaunchKernelGGL((fq::count_newlines_kernel<false, uint16_t>), dim3((unsigned)((L.nblocks + fq::COUNT_SUB - 1) / fq::COUNT_SUB)), dim3(fq::THREADS), 0, st, abase, mis,
                       nbytes, (uint64_t)L.nblocks, counts, reinterpret_cast<uint16_t *>(w + L.off_masks));
    if (int rc = scan_u32(counts, L.nblocks, nullptr, 1, L.nblocks, blockoff, scanpart, st))
        return rc;
    hipLaunchKernelGGL((fq::write_newlines_kernel<false, uint16_t>), dim3((unsigned)((L.nblocks + 3) / 4)), dim3(fq::THREADS), 0, st,
                       reinterpret_cast<const uint16_t *>(w + L.off_masks), (uint64_t)L.nblocks * fq::THREADS, mis, blockoff,
                       (uint64_t)L.nblocks, line_end, (uint8_t *)nullptr);
    // The line count exists only on the device (blockoff[nblocks]); the record kernels are launched for
    // the most records the file could hold and read the real count there.  The shortest record is 7 bytes: the
    // reference reads the third line without looking at it (fastq.go:182), so "@\nA\n\nI\n" parses.
    const uint64_t *nlines_dev = blockoff + L.nblocks;
    const uint64_t most = nbytes / 7 + 1;
    hipLaunchKernelGGL(fq::records_kernel, dim3((unsigned)std::min<uint64_t>((most + fq::THREADS - 1) / fq::THREADS, 256ull * 16ull)), dim3(fq::THREADS), 0, st,
                       d_file, line_end, nlines_dev, seq_len, seq_start, d_rec_start, max_records, res);
    if (int rc = scan_u32(seq_len, 0, nlines_dev, 4, most, d_offsets, scanpart, st, max_records))
        return rc;
    hipLaunchKernelGGL(fq::finish_kernel, dim3(1), dim3(1), 0, st, d_file, nbytes, line_end, nlines_dev, d_offsets,
                       max_records, res);
    hipLaunchKernelGGL(fq::gather_kernel, dim3((unsigned)std::min<uint64_t>(most, 256ull * 32ull)), dim3(fq::THREADS), 0, st,
                       d_file, seq_start, d_offsets, res, d_seqs);
    PH_HIP(hipGetLastError());
    PH_HIP(hipMemcpyAsync(d_result, res, 4 * 8, hipMemcpyDeviceToDevice, st));
    return POLYHIP_OK;
}

int polyhip_fastq_pack(const uint8_t *file, uint64_t nbytes, uint8_t *seqs, uint64_t *offsets, uint64_t *rec_start,
                       uint64_t max_records, uint64_t *result)
{
    PH_REQUIRE(result && offsets && (file || nbytes == 0) && (seqs || nbytes == 0), "polyhip_fastq_pack: null pointer");
    const uint64_t most = nbytes / 7 + 1;
    DevBuf dfile, dseqs, doffs, drec, dres, dwork;
    PH_HIP(dfile.alloc(nbytes));
    PH_HIP(dseqs.alloc(nbytes));
    const uint64_t held = std::min(most, max_records); // the caller's offsets / rec_start hold this many records
    PH_HIP(doffs.alloc((held + 1) * 8));
    PH_HIP(drec.alloc((held + 1) * 8));
    PH_HIP(dres.alloc(4 * 8));
    const size_t wb = polyhip_fastq_workspace_bytes(nbytes);
    PH_HIP(dwork.alloc(wb));
    // the calling thread's own streams (host_pipeline.h), never the null stream: the parse is one pass over the whole
    // image, so there is nothing to overlap the upload with, but the two result copies travel side by side
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    if (nbytes)
        PH_HIP(hipMemcpyAsync(dfile.p, file, nbytes, hipMemcpyHostToDevice, hs.s[0]));
    int rc = polyhip_fastq_pack_dev(dfile.as<uint8_t>(), nbytes, dseqs.as<uint8_t>(), doffs.as<uint64_t>(),
                                    drec.as<uint64_t>(), max_records, dres.as<uint64_t>(), dwork.p, wb, hs.s[0]);
    if (rc != POLYHIP_OK) {
        (void)hs.sync_both();
        return rc;
    }
    PH_HIP(hipMemcpyAsync(result, dres.p, 4 * 8, hipMemcpyDeviceToHost, hs.s[0]));
    PH_HIP(hipStreamSynchronize(hs.s[0]));
    const uint64_t n = result[0];
    PH_HIP(hipMemcpyAsync(offsets, doffs.p, (n + 1) * 8, hipMemcpyDeviceToHost, hs.s[1]));
    if (rec_start && n)
        PH_HIP(hipMemcpyAsync(rec_start, drec.p, n * 8, hipMemcpyDeviceToHost, hs.s[1]));
    if (result[3])
        PH_HIP(hipMemcpyAsync(seqs, dseqs.p, result[3], hipMemcpyDeviceToHost, hs.s[0]));
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

size_t polyhip_fasta_workspace_bytes(uint64_t nbytes) { return fq::layout_fasta(nbytes).total; }

int polyhip_fasta_pack_dev(const uint8_t *d_file, uint64_t nbytes, uint8_t *d_seqs, uint64_t *d_offsets,
                           uint64_t *d_rec_start, uint64_t max_records, uint64_t *d_result, void *d_work,
                           size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(d_result && d_offsets && d_work && (d_file || nbytes == 0) && (d_seqs || nbytes == 0),
               "polyhip_fasta_pack: null pointer");
    const fq::Layout L = fq::layout_fasta(nbytes);
    PH_REQUIRE(work_bytes >= L.total, "polyhip_fasta_pack: workspace too small (%zu < %zu)", work_bytes, L.total);
    PH_REQUIRE(L.nblocks < (1ull << 31) && nbytes < (1ull << 39), "polyhip_fasta_pack: file too large for one call");
    hipStream_t st = as_stream(stream);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    unsigned long long *res = reinterpret_cast<unsigned long long *>(w + L.off_res);
    uint64_t *scanpart = reinterpret_cast<uint64_t *>(w + L.off_scanpart);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + L.off_counts);
    uint64_t *blockoff = reinterpret_cast<uint64_t *>(w + L.off_blockoff);
    uint64_t *line_end = reinterpret_cast<uint64_t *>(w + L.off_lineend);
    uint32_t *seq_len = reinterpret_cast<uint32_t *>(w + L.off_seqlen);
    uint32_t *is_header = reinterpret_cast<uint32_t *>(w + L.off_ishdr);
    uint64_t *hrank = reinterpret_cast<uint64_t *>(w + L.off_hrank);
    uint64_t *dst = reinterpret_cast<uint64_t *>(w + L.off_dst);
    uint8_t *kind = w + L.off_kind;

    PH_HIP(hipMemsetAsync(res, 0, fq::F_WORDS * 8, st));
    PH_HIP(hipMemsetAsync(res + fq::F_FIRSTEMPTY, 0xFF, 16, st)); // F_FIRSTEMPTY, F_FIRSTHDR: none yet
    PH_HIP(hipMemsetAsync(hrank, 0, 16, st));
    PH_HIP(hipMemsetAsync(dst, 0, 16, st));
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(d_file) & 15u);
    const uint8_t *abase = d_file - mis; // 16-byte aligned; bytes in front of the file are masked out
    hipLaunchKernelGGL((fq::count_newlines_kernel<true, uint32_t>), dim3((unsigned)((L.nblocks + fq::COUNT_SUB - 1) / fq::COUNT_SUB)), dim3(fq::THREADS), 0, st, abase, mis,
                       nbytes, (uint64_t)L.nblocks, counts, reinterpret_cast<uint32_t *>(w + L.off_masks));
    if (int rc = scan_u32(counts, L.nblocks, nullptr, 1, L.nblocks, blockoff, scanpart, st))
        return rc;
    hipLaunchKernelGGL((fq::write_newlines_kernel<true, uint32_t>), dim3((unsigned)((L.nblocks + 3) / 4)), dim3(fq::THREADS), 0, st,
                       reinterpret_cast<const uint32_t *>(w + L.off_masks), (uint64_t)L.nblocks * fq::THREADS, mis, blockoff,
                       (uint64_t)L.nblocks, line_end, kind);
    const uint64_t *nlines_dev = blockoff + L.nblocks; // the line count exists only on the device
    // per-line kernels: as many workgroups as keep the chip busy, each striding over the lines (their count exists only on the device)
    const unsigned gl = (unsigned)std::min<uint64_t>((nbytes + fq::THREADS - 1) / fq::THREADS + 1, 256ull * 16ull);
    hipLaunchKernelGGL(fq::fasta_classify_kernel, dim3(gl), dim3(fq::THREADS), 0, st, d_file, line_end, nlines_dev, kind,
                       is_header, seq_len);
    // hrank[k] = headers before line k, dst[k] = sequence bytes before line k: one pass over both arrays
    hipLaunchKernelGGL(fq::scan2_sums_kernel, dim3(fq::SCAN_SEGS), dim3(1024), 0, st, is_header, seq_len, nlines_dev, scanpart);
    hipLaunchKernelGGL(fq::scan2_offsets_kernel, dim3(1), dim3(2 * fq::SCAN_SEGS), 0, st, scanpart, nlines_dev, hrank, dst,
                       res + fq::F_FIRSTHDR);
    hipLaunchKernelGGL(fq::scan2_apply_kernel, dim3(fq::SCAN_SEGS), dim3(1024), 0, st, is_header, seq_len, nlines_dev,
                       res + fq::F_FIRSTHDR, scanpart, hrank, dst);
    hipLaunchKernelGGL(fq::fasta_offsets_kernel, dim3(gl), dim3(fq::THREADS), 0, st, nlines_dev, is_header, hrank, dst, line_end,
                       d_offsets, d_rec_start);
    hipLaunchKernelGGL(fq::fasta_empty_kernel, dim3(gl), dim3(fq::THREADS), 0, st, hrank, nlines_dev, d_offsets, res);
    hipLaunchKernelGGL(fq::fasta_finish_kernel, dim3(1), dim3(1), 0, st, d_file, nbytes, line_end, nlines_dev, hrank, d_offsets,
                       max_records, res);
    if (env_is("POLYHIP_FASTA_STREAM", '0'))
        hipLaunchKernelGGL(fq::fasta_gather_kernel, dim3((unsigned)std::min<uint64_t>(gl, 256ull * 32ull)), dim3(fq::THREADS), 0, st,
                           d_file, line_end, nlines_dev, seq_len, dst, res, d_seqs);
    else
        hipLaunchKernelGGL(fq::fasta_gather_stream_kernel, dim3((unsigned)std::min<uint64_t>(gl, 256ull * 32ull)), dim3(fq::THREADS), 0,
                           st, d_file, line_end, nlines_dev, seq_len, dst, res, d_seqs);
    PH_HIP(hipGetLastError());
    PH_HIP(hipMemcpyAsync(d_result, res, 4 * 8, hipMemcpyDeviceToDevice, st));
    return POLYHIP_OK;
}

int polyhip_fasta_pack(const uint8_t *file, uint64_t nbytes, uint8_t *seqs, uint64_t *offsets, uint64_t *rec_start,
                       uint64_t max_records, uint64_t *result)
{
    PH_REQUIRE(result && offsets && (file || nbytes == 0) && (seqs || nbytes == 0), "polyhip_fasta_pack: null pointer");
    const uint64_t most = nbytes / 2 + 2; // ">\n" is the shortest header line
    DevBuf dfile, dseqs, doffs, drec, dres, dwork;
    PH_HIP(dfile.alloc(nbytes));
    PH_HIP(dseqs.alloc(nbytes));
    PH_HIP(doffs.alloc((most + 1) * 8));
    PH_HIP(drec.alloc((most + 1) * 8));
    PH_HIP(dres.alloc(4 * 8));
    const size_t wb = polyhip_fasta_workspace_bytes(nbytes);
    PH_HIP(dwork.alloc(wb));
    HostStreams &hs = host_streams(); // the calling thread's own streams, never the null stream (see polyhip_fastq_pack)
    PH_HIP(hs.init());
    if (nbytes)
        PH_HIP(hipMemcpyAsync(dfile.p, file, nbytes, hipMemcpyHostToDevice, hs.s[0]));
    PH_HIP(hipMemsetAsync(doffs.p, 0, 16, hs.s[0]));
    int rc = polyhip_fasta_pack_dev(dfile.as<uint8_t>(), nbytes, dseqs.as<uint8_t>(), doffs.as<uint64_t>(),
                                    drec.as<uint64_t>(), max_records, dres.as<uint64_t>(), dwork.p, wb, hs.s[0]);
    if (rc != POLYHIP_OK) {
        (void)hs.sync_both();
        return rc;
    }
    PH_HIP(hipMemcpyAsync(result, dres.p, 4 * 8, hipMemcpyDeviceToHost, hs.s[0]));
    PH_HIP(hipStreamSynchronize(hs.s[0]));
    const uint64_t n = result[0];
    PH_HIP(hipMemcpyAsync(offsets, doffs.p, (n + 1) * 8, hipMemcpyDeviceToHost, hs.s[1]));
    if (rec_start && n)
        PH_HIP(hipMemcpyAsync(rec_start, drec.p, n * 8, hipMemcpyDeviceToHost, hs.s[1]));
    if (result[2])
        PH_HIP(hipMemcpyAsync(seqs, dseqs.p, result[2], hipMemcpyDeviceToHost, hs.s[0]));
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

} // extern "C"
