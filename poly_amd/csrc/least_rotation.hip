// least_rotation.hip -- K5: batched seqhash.RotateSequence for gfx950.
//
// Replaces boothLeastRotation + RotateSequence (seqhash/seqhash.go:78-138) for a
// packed batch of circular sequences: for each one, the index of its
// lexicographically (byte order) least rotation -- the SMALLEST such index, which
// is what Booth's algorithm returns on s+s (checked against the oracle on
// periodic strings) -- and, optionally, the rotated sequence itself.
//
// Booth's algorithm is a serial scan with a 2n-entry failure table.  On a GPU
// the same answer comes from candidate elimination with the sequence staged in
// LDS -- one WAVE per sequence up to 7 kB (least_rotation_wave_kernel: no
// barrier, four sequences per workgroup), one workgroup per sequence beyond
// that (least_rotation_kernel<true>), reading global memory when LDS cannot
// hold it (<false>):
//   1. every position's first 4 bytes as one big-endian word; block-wide min;
//      the candidates are the positions that attain it (for DNA: ~n/256 of them)
//   2. rounds of 4 more bytes: min of the next word over the surviving
//      candidates, keep those that attain it -- until one is left, or the
//      compared depth reaches n (the survivors are then equal rotations and the
//      smallest index wins)
//   3. low-complexity / periodic input that does not thin out (more candidates
//      than the LDS list holds, or too many rounds) is finished by wave 0 with
//      the exact two-pointer minimal-rotation algorithm, its match-extension
//      loop vectorised 64 bytes per step with a ballot.
//   An exact repetition of a block of d bytes (c evenly spread candidates, one
//   pass comparing s[p] with s[p + n/c]) restarts the search on that block.
//
// Byte compare / integer work, no MFMA.  Algorithmic HBM bytes per sequence:
// n (read) [+ n if the rotated sequence is written] + 8.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "host_pipeline.h"

namespace polyhip {
namespace k5 {

#ifndef PH_K5_THREADS
#define PH_K5_THREADS 256
#endif
constexpr int THREADS = PH_K5_THREADS;
constexpr uint32_t LIST_CAP = 1024;      // candidates kept in LDS (random DNA: ~n/256 after the first word)
constexpr uint32_t MAX_ROUNDS = 64;      // 4 bytes per round before the serial fallback
constexpr uint32_t LDS_SEQ_MAX = 120 * 1024;
constexpr uint64_t MARK = ~0ull;       // rot[q] of a sequence the wave kernel leaves to the workgroup kernels
constexpr uint64_t WAVE_SEQ_MAX = 7168; // longest sequence a wave takes alone (8 kB: the workgroup kernel is ahead, 0.32 vs 0.36 ms per 0.5 GB; 6 kB: behind, 0.39 vs 0.29)
#ifndef PH_K5_WLIST
#define PH_K5_WLIST 1024
#endif
constexpr uint32_t WLIST = PH_K5_WLIST; // candidates one wave keeps, as 16-bit positions (random DNA: n / 256 after the first word)
constexpr uint32_t WRAP = 24; // bytes of s[0..] staged again behind s[n-1]: a 16-byte output piece + its funnel dword never wrap

// least value of a wave, in every lane: the DPP ladder (pairs, quads, rows of 16 by rotation, then the two row broadcasts
// land the result in lane 63) -- six v_min with no LDS round trip
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#define PH_DPP_MIN(ctrl, rows) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false))
    PH_DPP_MIN(0xb1, 0xf);  // quad_perm [1,0,3,2]
    PH_DPP_MIN(0x4e, 0xf);  // quad_perm [2,3,0,1]
    PH_DPP_MIN(0x124, 0xf); // row_ror 4
    PH_DPP_MIN(0x128, 0xf); // row_ror 8
    PH_DPP_MIN(0x142, 0xa); // row_bcast15 into rows 1, 3
    PH_DPP_MIN(0x143, 0xc); // row_bcast31 into rows 2, 3
#undef PH_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// least value of the work group; ONE barrier: the per-wave slots alternate between two sets (`par`, block-uniform, flips
// per call), so the next call's writes cannot overtake this call's reads
__device__ __forceinline__ uint32_t block_min(uint32_t v, uint32_t (*red)[THREADS / 64], uint32_t &par)
{
    v = wave_min(v);
    par ^= 1u;
    if ((threadIdx.x & 63) == 0)
        red[par][threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = red[par][0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w)
        r = min(r, red[par][w]);
    return r;
}

// big-endian word of the 4 bytes at cyclic position p (p < n); the staged copy
// carries s[0..3] again behind s[n-1] so no wrap is needed inside the word
template <class Ptr> __device__ __forceinline__ uint32_t word_at(Ptr s, uint64_t p)
{
    return ((uint32_t)s[p] << 24) | ((uint32_t)s[p + 1] << 16) | ((uint32_t)s[p + 2] << 8) | (uint32_t)s[p + 3];
}

// the reverse complement of g[0..n) as an indexable string (the global-memory search on a sequence LDS cannot hold)
struct RcView {
    const uint8_t *g;
    uint64_t n;
    const uint8_t *cmpT;
    __device__ __forceinline__ uint32_t operator[](uint64_t p) const { return cmpT[g[n - 1 - p]]; }
};

// exact minimal rotation (smallest index) by the two-pointer algorithm, run by
// ONE wave; cyc(p) reads the byte at cyclic position p < 2n
template <class Ptr> __device__ uint64_t two_pointer_wave(Ptr s, uint64_t n)
{
    const int lane = threadIdx.x & 63;
    uint64_t i = 0, j = 1, k = 0;
    while (i < n && j < n && k < n) {
        const uint64_t t = k + lane;
        const bool valid = t < n;
        uint64_t pi = i + t, pj = j + t;
        pi -= pi >= n ? n : 0;
        pj -= pj >= n ? n : 0;
        const uint32_t a = valid ? s[pi] : 0u, b = valid ? s[pj] : 0u;
        const uint64_t ne = __ballot(valid && a != b);
        if (ne == 0) {
            k += 64;
            continue;
        }
        const int f = __builtin_ctzll(ne);
        k += f;
        const uint32_t af = (uint32_t)__builtin_amdgcn_readlane((int)a, f);
        const uint32_t bf = (uint32_t)__builtin_amdgcn_readlane((int)b, f);
        if (af > bf)
            i += k + 1;
        else
            j += k + 1;
        if (i == j)
            ++j;
        k = 0;
    }
    return i < j ? i : j;
}

// big-endian word at byte K of the 8-byte window {d1:d0}: one v_perm_b32
template <uint32_t K> __device__ __forceinline__ uint32_t word_k(uint32_t d0, uint32_t d1)
{
    return __builtin_amdgcn_perm(d1, d0, (K << 24) | ((K + 1) << 16) | ((K + 2) << 8) | (K + 3));
}

// 4 bytes of an LDS-staged sequence at byte position p as a big-endian word: two aligned dwords, a
// funnel shift and a byte swap (the staged copy carries WRAP wrapped bytes behind s[n-1])
__device__ __forceinline__ uint32_t word_lds(const uint32_t *__restrict__ L, uint32_t p)
{
    const uint32_t d0 = L[p >> 2], d1 = L[(p >> 2) + 1];
    const uint32_t le = __builtin_amdgcn_alignbyte(d1, d0, p & 3u);
    return __builtin_bswap32(le);
}

// Reverse-complement view (round 5, seqhash.Hash of a double-stranded sequence, seqhash.go:180-193): the least rotation of
// transform.ReverseComplement(s) is found on a copy staged in LDS in THAT order -- byte j of the staged string is the
// complement of s[n - 1 - j] -- so the second strand is never written to memory (round 4 wrote it, 0.5 GB per 100k x 5 kb,
// only to read it back here).  `cmpT`: the complement of every byte value (LDS).
__device__ __forceinline__ uint32_t revcmp4(uint32_t w, const uint8_t *__restrict__ cmpT)
{
    return (uint32_t)cmpT[w >> 24] | ((uint32_t)cmpT[(w >> 16) & 0xFFu] << 8) | ((uint32_t)cmpT[(w >> 8) & 0xFFu] << 16) |
           ((uint32_t)cmpT[w & 0xFFu] << 24);
}
// stage s[0..n) (rc: its reverse complement) + WRAP wrapped bytes into lds: 16 bytes per thread and step, read from the
// sequence's own (unaligned) address, written to aligned LDS; the last n % 16 singly.  first / step: the thread's place.
__device__ __forceinline__ void stage_sequence(uint8_t *__restrict__ lds, const uint8_t *__restrict__ g, uint32_t n32, bool rc,
                                               const uint8_t *__restrict__ cmpT, uint32_t first, uint32_t step)
{
    const uint32_t n16 = n32 >> 4;
    uint4 *L4 = reinterpret_cast<uint4 *>(lds);
    if (!rc) {
        for (uint32_t t = first; t < n16; t += step) {
            uint4 v;
            __builtin_memcpy(&v, g + 16u * t, 16);
            L4[t] = v;
        }
        if (first < (n32 & 15u))
            lds[16u * n16 + first] = g[16u * n16 + first];
        if (first < WRAP) // s[0..] again behind s[n-1] (n may be tiny: cyclic)
            lds[n32 + first] = g[n32 >= WRAP ? first : first % n32];
    } else {
        for (uint32_t t = first; t < n16; t += step) { // staged bytes [16 t, 16 t + 16) = complement of s[n-16t-16 .. n-16t) reversed
            uint4 v, o;
            __builtin_memcpy(&v, g + (n32 - 16u * t - 16u), 16);
            o.x = revcmp4(v.w, cmpT);
            o.y = revcmp4(v.z, cmpT);
            o.z = revcmp4(v.y, cmpT);
            o.w = revcmp4(v.x, cmpT);
            L4[t] = o;
        }
        if (first < (n32 & 15u))
            lds[16u * n16 + first] = cmpT[g[n32 - 1u - (16u * n16 + first)]];
        if (first < WRAP) {
            const uint32_t j = n32 >= WRAP ? first : first % n32;
            lds[n32 + first] = cmpT[g[n32 - 1u - j]];
        }
    }
}

// RAW staging (round 6, seqhash.Hash on sequences a wave takes alone): g holds the caller's bytes, not yet normalised.  The
// staging maps every byte through a 16-bit table entry -- low byte = the normalised letter (strings.ToUpper, RNA U -> T:
// seqhash.go:143-148), bit 8 = "not in the alphabet" -- on its way into LDS, and writes the normalised bytes out to `norm`
// as well (what the comparison and BLAKE3 read afterwards): the streaming normalise pass (one read and one launch) is gone.
// Returns the OR of the entries seen (bit 8: a letter outside the alphabet somewhere in this thread's bytes).
__device__ __forceinline__ uint32_t stage_sequence_raw(uint8_t *__restrict__ lds, const uint8_t *__restrict__ g, uint32_t n32,
                                                       const uint16_t *__restrict__ upT, uint8_t *__restrict__ norm, uint32_t first,
                                                       uint32_t step)
{
    const uint32_t n16 = n32 >> 4;
    uint4 *L4 = reinterpret_cast<uint4 *>(lds);
    uint32_t bad = 0;
    auto map4 = [&](uint32_t w) {
        const uint32_t e0 = upT[w & 0xFFu], e1 = upT[(w >> 8) & 0xFFu], e2 = upT[(w >> 16) & 0xFFu], e3 = upT[w >> 24];
        bad |= e0 | e1 | e2 | e3;
        return (e0 & 0xFFu) | ((e1 & 0xFFu) << 8) | ((e2 & 0xFFu) << 16) | (e3 << 24);
    };
    for (uint32_t t = first; t < n16; t += step) {
        uint4 v, o;
        __builtin_memcpy(&v, g + 16u * t, 16);
        o.x = map4(v.x);
        o.y = map4(v.y);
        o.z = map4(v.z);
        o.w = map4(v.w);
        L4[t] = o;
        __builtin_memcpy(norm + 16u * t, &o, 16); // (the sequence's own, unaligned, address -- as the load)
    }
    if (first < (n32 & 15u)) {
        const uint32_t e = upT[g[16u * n16 + first]];
        bad |= e;
        lds[16u * n16 + first] = (uint8_t)e;
        norm[16u * n16 + first] = (uint8_t)e;
    }
    if (first < WRAP) // s[0..] again behind s[n-1] (n may be tiny: cyclic)
        lds[n32 + first] = (uint8_t)upT[g[n32 >= WRAP ? first : first % n32]];
    return bad;
}

template <bool IN_LDS>
__global__ __launch_bounds__(THREADS) void least_rotation_kernel(const uint8_t *__restrict__ seqs,
                                                                const uint64_t *__restrict__ offs, uint64_t nseq,
                                                                uint64_t lds_seq_bytes, uint64_t *__restrict__ rot,
                                                                uint8_t *__restrict__ rotated, uint32_t chunk, int rc)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t cmpT[256];
    cmpT[threadIdx.x] = (uint8_t)dna_complement_upper(threadIdx.x); // THREADS == 256; visible after the first barrier
    __shared__ uint32_t listA[LIST_CAP], listB[LIST_CAP];
    __shared__ uint32_t red[2][THREADS / 64];
    __shared__ uint32_t cnt, cntq;
    __shared__ uint64_t answer;
    const int tid = threadIdx.x;
    uint32_t *L = reinterpret_cast<uint32_t *>(lds);
    uint32_t par = 0;

    // The wave kernel ran first and MARKed what it left (sequences too long for a wave's LDS share): a
    // workgroup looks at `chunk` consecutive rot[] entries at a time (one coalesced load) and works through the marked ones.
    __shared__ uint32_t marked[THREADS];
    __shared__ uint32_t nmark;
    for (uint64_t base = (uint64_t)blockIdx.x * chunk; base < nseq; base += (uint64_t)gridDim.x * chunk)
    {
        __syncthreads(); // the list of the chunk before is used up
        if (tid == 0)
            nmark = 0;
        __syncthreads();
        if ((uint32_t)tid < chunk && base + tid < nseq && rot[base + tid] == MARK)
            marked[atomicAdd(&nmark, 1u)] = (uint32_t)tid;
        __syncthreads();
        const uint32_t nm = nmark;
    for (uint32_t km = 0; km < nm; ++km) {
        const uint64_t q = base + marked[km];
        const uint64_t o0 = offs[q];
        const uint64_t n = offs[q + 1] - o0;
        const uint8_t *g = seqs + o0;
        if (n <= 1) {
            if (tid == 0)
                rot[q] = 0;
            if (rotated && n == 1 && tid == 0)
                rotated[o0] = g[0];
            continue;
        }
        if (IN_LDS && n + WRAP > lds_seq_bytes)
            continue; // the global-memory launch handles this one
        if (!IN_LDS && n + WRAP <= lds_seq_bytes)
            continue;

        // ---- stage the sequence (+ WRAP wrapped bytes so a word or an output piece never wraps): 16 bytes per thread
        // and step, read from the sequence's own (unaligned) address, written to aligned LDS; the last n % 16 singly
        __syncthreads();
        if (IN_LDS)
            stage_sequence(lds, g, (uint32_t)n, rc != 0, cmpT, (uint32_t)tid, THREADS);
        auto gbyte = [&](uint64_t p) -> uint32_t { return rc ? cmpT[g[n - 1 - p]] : g[p]; }; // byte p < n of the searched string
        if (tid == 0) {
            cnt = 0;
            cntq = 0;
            answer = ~0ull;
        }
        __syncthreads();

        auto byte_at = [&](uint64_t p) -> uint32_t { // cyclic position p < 2n
            return IN_LDS ? lds[p >= n ? p - n : p] : gbyte(p >= n ? p - n : p);
        };
        // `ne`: the length the search runs on.  A sequence that is an exact repetition of a block of d bytes (a tandem
        // repeat that closes on itself) has its least rotation -- smallest index -- inside the first block, and it is
        // the least rotation of that block taken as a circular string of its own: the search restarts with ne = d
        // (bytes behind position d - 1 continue with the block again, so nothing has to be re-staged).
        uint64_t ne = n;
        auto word = [&](uint64_t p) -> uint32_t { // big-endian 4 bytes at cyclic position p < 2 * ne
            p -= p >= ne ? ne : 0;
            if (IN_LDS) {
                if (n >= 4)
                    return word_lds(L, (uint32_t)p);
                return word_at(lds, p); // n = 2, 3: the wrapped bytes cover it
            }
            return (gbyte(p) << 24) | (gbyte((p + 1) % n) << 16) | (gbyte((p + 2) % n) << 8) | gbyte((p + 3) % n);
        };

    search:
        // ---- 1. least first word (LDS: a thread takes 4 neighbouring positions from two dwords)
        // A least word made of one byte (aaaa) comes in runs: a position whose predecessor also starts with
        // aaaa loses to it (the byte behind the run is larger than a, else a smaller word would exist), so only
        // the first position of each run is a candidate -- a poly-A tail is ONE candidate, not thousands.
        // A sequence that is nothing but that byte leaves no candidate at all: every rotation is equal, index 0.
        uint32_t m = 0xFFFFFFFFu;
        bool serial = false, homo = false;
        if (IN_LDS && ne >= 8) {
            // whole quads of positions from two dwords each, the ne % 4 positions behind them singly
            const uint32_t ne32 = (uint32_t)ne, nfull = ne32 >> 2, ntail = ne32 & 3u;
            for (uint32_t t = tid; t < nfull; t += THREADS) {
                const uint32_t d0 = L[t], d1 = L[t + 1];
                m = min(min(m, word_k<0>(d0, d1)), word_k<1>(d0, d1));
                m = min(min(m, word_k<2>(d0, d1)), word_k<3>(d0, d1));
            }
            if ((uint32_t)tid < ntail)
                m = min(m, word(4u * nfull + tid));
            m = block_min(m, red, par);
            homo = (m & 0xFFFFu) == (m >> 16) && (m & 0xFFu) == ((m >> 8) & 0xFFu);
            auto push = [&](uint32_t p) {
                if (homo && word(p ? p - 1 : ne32 - 1) == m)
                    return; // inside a run of the least byte: the run's first position beats it
                const uint32_t slot = atomicAdd(&cnt, 1u);
                if (slot < LIST_CAP)
                    listA[slot] = p;
            };
            // the quads that hold the least word at all (random DNA: one in 64) go on a list first, so that the pass over
            // every position stays a handful of instructions; the positions come from the list
            for (uint32_t t = tid; t < nfull; t += THREADS) {
                const uint32_t d0 = L[t], d1 = L[t + 1];
                const uint32_t lo = min(min(word_k<0>(d0, d1), word_k<1>(d0, d1)), min(word_k<2>(d0, d1), word_k<3>(d0, d1)));
                if (lo == m) {
                    const uint32_t slot = atomicAdd(&cntq, 1u);
                    if (slot < LIST_CAP)
                        listB[slot] = t;
                }
            }
            if ((uint32_t)tid < ntail && word(4u * nfull + tid) == m)
                push(4u * nfull + tid);
            __syncthreads();
            const uint32_t cq = cntq;
            if (cq <= LIST_CAP) {
                for (uint32_t e = tid; e < cq; e += THREADS) {
                    const uint32_t t = listB[e], d0 = L[t], d1 = L[t + 1];
                    if (word_k<0>(d0, d1) == m)
                        push(4u * t);
                    if (word_k<1>(d0, d1) == m)
                        push(4u * t + 1);
                    if (word_k<2>(d0, d1) == m)
                        push(4u * t + 2);
                    if (word_k<3>(d0, d1) == m)
                        push(4u * t + 3);
                }
            } else { // low complexity: more quads than the list holds, every position is looked at directly
                for (uint32_t t = tid; t < nfull; t += THREADS) {
                    const uint32_t d0 = L[t], d1 = L[t + 1];
                    if (word_k<0>(d0, d1) == m)
                        push(4u * t);
                    if (word_k<1>(d0, d1) == m)
                        push(4u * t + 1);
                    if (word_k<2>(d0, d1) == m)
                        push(4u * t + 2);
                    if (word_k<3>(d0, d1) == m)
                        push(4u * t + 3);
                }
            }
        } else {
            for (uint64_t p = tid; p < ne; p += THREADS)
                m = min(m, word(p));
            m = block_min(m, red, par);
            homo = ne >= 4 && (m & 0xFFFFu) == (m >> 16) && (m & 0xFFu) == ((m >> 8) & 0xFFu);
            for (uint64_t p0 = 0; p0 < ne; p0 += THREADS) {
                const uint64_t p = p0 + tid;
                const bool is = p < ne && word(p) == m && !(homo && word(p ? p - 1 : ne - 1) == m);
                if (is) {
                    const uint32_t slot = atomicAdd(&cnt, 1u);
                    if (slot < LIST_CAP)
                        listA[slot] = (uint32_t)p;
                }
            }
        }
        __syncthreads();
        uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
        // c occurrences of the least word, evenly spread if the sequence is periodic: is it a repetition of its first
        // ne / c bytes?  (short-period tandem repeats used to sit in the rounds below until the serial fallback took
        // over: 6.6 ms per 100k sequences against 0.5 ms for random DNA.)  One mismatch anywhere rules it out, so a
        // first look takes one byte per thread -- random sequence leaves here -- and only then the whole length
        if (c >= 2 && ne <= 0xFFFFFFFFull && (uint32_t)ne % c == 0) {
            const uint32_t ne32 = (uint32_t)ne, d = ne32 / c;
            auto differs_at = [&](uint32_t p) { return byte_at(p) != byte_at(p + d >= ne32 ? p + d - ne32 : p + d); };
            bool differs = (uint32_t)tid < ne32 && differs_at(tid);
            if (!__syncthreads_or(differs ? 1 : 0)) {
                for (uint32_t p = tid + THREADS; p < ne32 && !differs; p += THREADS)
                    differs = differs_at(p);
                if (!__syncthreads_or(differs ? 1 : 0)) {
                    ne = d; // block-uniform
                    if (tid == 0) {
                        cnt = 0;
                        cntq = 0;
                    }
                    __syncthreads();
                    goto search; // again, on one block
                }
            }
        }
        if (c > LIST_CAP || ne > 0xFFFFFFFFull)
            serial = true;
        if (c == 0) { // homopolymer: all rotations equal, the smallest index is 0
            if (tid == 0)
                listA[0] = 0;
            c = 1;
            __syncthreads();
        }

        // ---- 2. rounds of 4 more bytes
        uint32_t *cur = listA, *nxt = listB;
        uint64_t depth = 4;
        uint32_t rounds = 0;
        while (!serial && c > 64 && depth < ne) { // work-group rounds while a wave cannot hold the candidates
            if (++rounds > MAX_ROUNDS) {
                serial = true;
                break;
            }
            uint32_t mm = 0xFFFFFFFFu;
            for (uint32_t e = tid; e < c; e += THREADS)
                mm = min(mm, word((uint64_t)cur[e] + depth));
            mm = block_min(mm, red, par);
            if (tid == 0)
                cnt = 0;
            __syncthreads();
            for (uint32_t e = tid; e < c; e += THREADS) {
                const uint32_t p = cur[e];
                if (word((uint64_t)p + depth) == mm)
                    nxt[atomicAdd(&cnt, 1u)] = p;
            }
            __syncthreads();
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            uint32_t *t = cur;
            cur = nxt;
            nxt = t;
            depth += 4;
        }

        // ---- 3. answer
        if (serial) {
            if (tid < 64) {
                uint64_t r;
                if (IN_LDS)
                    r = two_pointer_wave(lds, ne);
                else
                    r = rc ? two_pointer_wave(RcView{g, n, cmpT}, ne) : two_pointer_wave(g, ne);
                if (tid == 0)
                    answer = r;
            }
        } else if (c <= 64) {
            // one candidate per lane of wave 0, the rounds go on without a barrier; what is left when the compared depth
            // reaches ne are equal rotations (or a single one): the smallest index wins
            if (tid < 64) {
                bool alive = (uint32_t)tid < c;
                const uint32_t p = alive ? cur[tid] : 0u;
                bool wserial = false;
                while (c > 1 && depth < ne) {
                    if (++rounds > MAX_ROUNDS) {
                        wserial = true;
                        break;
                    }
                    const uint32_t w = alive ? word((uint64_t)p + depth) : 0xFFFFFFFFu;
                    const uint32_t mm = wave_min(w);
                    alive = alive && w == mm;
                    c = (uint32_t)__builtin_popcountll(__ballot(alive));
                    depth += 4;
                }
                uint64_t r;
                if (wserial)
                    r = IN_LDS ? two_pointer_wave(lds, ne) : rc ? two_pointer_wave(RcView{g, n, cmpT}, ne) : two_pointer_wave(g, ne);
                else
                    r = wave_min(alive ? p : 0xFFFFFFFFu);
                if (tid == 0)
                    answer = r;
            }
        } else { // more than a wave of equal rotations
            uint32_t best = 0xFFFFFFFFu;
            for (uint32_t e = tid; e < c; e += THREADS)
                best = min(best, cur[e]);
            best = block_min(best, red, par);
            if (tid == 0)
                answer = best;
        }
        __syncthreads();
        const uint64_t r = answer;
        if (tid == 0)
            rot[q] = r;
        if (rotated) { // RotateSequence: (s + s)[r : r + n], seqhash.go:131-137
            uint8_t *out = rotated + o0;
            if (IN_LDS && n >= 32) {
                // bytes up to the first 16-byte-aligned output address and behind the last whole piece singly; between
                // them 16 bytes per thread and step: five aligned LDS dwords funnelled to the rotation's byte phase
                const uint32_t n32 = (uint32_t)n, r32 = (uint32_t)r;
                const uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u)) & 15u;
                const uint32_t nd = (n32 - head) >> 4, tail0 = head + 16u * nd;
                if ((uint32_t)tid < head)
                    out[tid] = (uint8_t)byte_at((uint64_t)tid + r);
                if ((uint32_t)tid < n32 - tail0)
                    out[tail0 + tid] = (uint8_t)byte_at((uint64_t)tail0 + tid + r);
                uint4 *od = reinterpret_cast<uint4 *>(out + head);
                for (uint32_t t = tid; t < nd; t += THREADS) {
                    uint32_t p = head + 16u * t + r32;
                    p -= p >= n32 ? n32 : 0;
                    // p + 15 may run past s[n-1]: the wrapped bytes behind it continue with s[0..]
                    const uint32_t *src = L + (p >> 2);
                    const uint32_t sh = p & 3u;
                    const uint32_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3], a4 = src[4];
                    uint4 v;
                    v.x = __builtin_amdgcn_alignbyte(a1, a0, sh);
                    v.y = __builtin_amdgcn_alignbyte(a2, a1, sh);
                    v.z = __builtin_amdgcn_alignbyte(a3, a2, sh);
                    v.w = __builtin_amdgcn_alignbyte(a4, a3, sh);
                    od[t] = v;
                }
            } else {
                for (uint64_t t = tid; t < n; t += THREADS) {
                    uint64_t p = t + r;
                    p -= p >= n ? n : 0;
                    out[t] = (uint8_t)byte_at(p);
                }
            }
        }
    }
    }
}


// ---- one WAVE per sequence (sequences up to a few kB): the same search with no barrier at all.  The workgroup kernel
// spends most of a 5 kb sequence's time on its fixed part -- a dozen barriers, every wave running the one-candidate
// bookkeeping (4.3 us per sequence and workgroup slot whatever the length: 2.5M sequences of 200 bp took 5.3 ms) --;
// here four waves of a workgroup work on four sequences, lists are appended to with ballots on a scalar count instead of LDS
// atomics, lists hold 16-bit positions; a sequence that does not fit the wave's LDS share is MARKed for the workgroup
// kernels that run behind this one.
__device__ __forceinline__ void wave_sync() // LDS written by one lane, read by another of the same wave
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) // set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(256) void least_rotation_wave_kernel(const uint8_t *__restrict__ seqs,
                                                                 const uint64_t *__restrict__ offs, uint64_t nseq,
                                                                 uint32_t lds_seq_bytes, uint64_t *__restrict__ rot,
                                                                 uint8_t *__restrict__ rotated, uint64_t *__restrict__ rot_rc,
                                                                 int raw_type, uint8_t *__restrict__ norm_out,
                                                                 uint32_t *__restrict__ any_bad)
{
    // raw_type >= 0 (seqhash.Hash's batch, every sequence within a wave's share): seqs are the caller's bytes of sequence
    // type raw_type (0 DNA, 1 RNA, 2 protein); they are normalised while they are staged (stage_sequence_raw), the
    // normalised copy goes to norm_out, a letter outside the alphabet raises *any_bad (the per-sequence pass behind this
    // kernel then names it).  raw_type < 0: seqs are searched as they are.
    // rot_rc != nullptr: ALSO the least rotation of the reverse complement (a second staging of the same bytes, in that
    // order, while they are still in this CU's caches; the strand itself is never written anywhere)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    __shared__ uint8_t cmpT[256];
    __shared__ uint16_t upT[256];
    {
        const uint32_t b = threadIdx.x; // 256 threads
        uint32_t c = b;
        if (raw_type >= 0) { // the table of seqhash.hip's normalise_stream_kernel
            c = (b - 'a' < 26u) ? b - 32u : b;
            if (raw_type == 1 && c == 'U')
                c = 'T';
            const char *set = raw_type == 2 ? "ACDEFGHIKLMNPQRSTVWYUO*BXZ" : "ATUGCYRSWKMBDHVNZ";
            bool ok = false;
            for (const char *p = set; *p; ++p)
                ok |= (uint32_t)(uint8_t)*p == c;
            upT[b] = (uint16_t)(c | (ok ? 0u : 0x100u));
        }
        cmpT[b] = (uint8_t)dna_complement_upper(c); // raw: the complement of the NORMALISED letter, straight from the raw byte
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t bad = 0;
    uint8_t *lds = lds_all + (size_t)wv * (lds_seq_bytes + 2u * WLIST * 2u);
    uint32_t *L = reinterpret_cast<uint32_t *>(lds);
    uint16_t *listA = reinterpret_cast<uint16_t *>(lds + lds_seq_bytes), *listB = listA + WLIST; // positions < 2^16

    for (uint64_t q = (uint64_t)blockIdx.x * 4 + wv; q < nseq; q += (uint64_t)gridDim.x * 4) {
        const uint64_t o0 = offs[q];
        const uint64_t n = offs[q + 1] - o0;
        const uint8_t *g = seqs + o0;
        if (n <= 1) {
            if (lane == 0) {
                rot[q] = 0;
                if (rot_rc)
                    rot_rc[q] = 0;
            }
            if (rotated && n == 1 && lane == 0)
                rotated[o0] = g[0];
            if (raw_type >= 0 && n == 1 && lane == 0) {
                const uint32_t e = upT[g[0]];
                bad |= e;
                norm_out[o0] = (uint8_t)e;
            }
            continue;
        }
        if (n + WRAP > lds_seq_bytes) {
            if (lane == 0) {
                rot[q] = MARK;
                if (rot_rc)
                    rot_rc[q] = MARK;
            }
            continue;
        }
        const uint32_t n32 = (uint32_t)n;
        for (int strand = 0; strand < (rot_rc ? 2 : 1); ++strand) {
        // ---- stage (+ WRAP wrapped bytes), 16 bytes per lane and step
        wave_sync();
        if (raw_type >= 0 && strand == 0)
            bad |= stage_sequence_raw(lds, g, n32, upT, norm_out + o0, lane, 64u);
        else
            stage_sequence(lds, g, n32, strand != 0, cmpT, lane, 64u);
        wave_sync();

        uint32_t ne = n32; // the length the search runs on (an exact repetition restarts on its first block)
        auto byte_at = [&](uint32_t p) -> uint32_t { return lds[p >= n32 ? p - n32 : p]; }; // cyclic position p < 2n
        auto word = [&](uint32_t p) -> uint32_t { // big-endian 4 bytes at cyclic position p < 2 * ne
            p -= p >= ne ? ne : 0;
            return n32 >= 4 ? word_lds(L, p) : word_at(lds, (uint64_t)p);
        };
        uint32_t c, m;
        bool homo;
    search:
        c = 0;
        m = 0xFFFFFFFFu;
        {
            // a list grows by a ballot: the lanes that add write behind the scalar count, in lane order
            auto push_if = [&](bool is, uint32_t p) {
                const uint64_t mask = __ballot(is);
                if (mask != 0ull) {
                    const uint32_t slot = c + lanes_below(mask);
                    if (is && slot < WLIST)
                        listA[slot] = (uint16_t)p;
                    c += (uint32_t)__builtin_popcountll(mask);
                }
            };
            if (ne >= 8) {
                const uint32_t nfull = ne >> 2, ntail = ne & 3u;
                for (uint32_t t = lane; t < nfull; t += 64) {
                    const uint32_t d0 = L[t], d1 = L[t + 1];
                    m = min(min(m, word_k<0>(d0, d1)), word_k<1>(d0, d1));
                    m = min(min(m, word_k<2>(d0, d1)), word_k<3>(d0, d1));
                }
                if (lane < ntail)
                    m = min(m, word(4u * nfull + lane));
                m = wave_min(m);
                homo = (m & 0xFFFFu) == (m >> 16) && (m & 0xFFu) == ((m >> 8) & 0xFFu);
                // inside a run of the least byte the run's first position beats the others (see the workgroup kernel)
                auto is_cand = [&](uint32_t w, uint32_t p) { return w == m && !(homo && word(p ? p - 1 : ne - 1) == m); };
                // the quads that hold the least word at all, then the positions inside them
                uint32_t cq = 0;
                for (uint32_t t0 = 0; t0 < nfull; t0 += 64) {
                    const uint32_t t = t0 + lane;
                    bool hit = false;
                    if (t < nfull) {
                        const uint32_t d0 = L[t], d1 = L[t + 1];
                        hit = min(min(word_k<0>(d0, d1), word_k<1>(d0, d1)), min(word_k<2>(d0, d1), word_k<3>(d0, d1))) == m;
                    }
                    const uint64_t mask = __ballot(hit);
                    if (mask != 0ull) {
                        const uint32_t slot = cq + lanes_below(mask);
                        if (hit && slot < WLIST)
                            listB[slot] = (uint16_t)t;
                        cq += (uint32_t)__builtin_popcountll(mask);
                    }
                }
                wave_sync();
                {
                    const uint32_t p = 4u * nfull + lane;
                    push_if(lane < ntail && is_cand(word(p), p), p);
                }
                if (cq <= WLIST) {
                    for (uint32_t e0 = 0; e0 < cq; e0 += 64) {
                        const bool valid = e0 + lane < cq;
                        const uint32_t t = valid ? listB[e0 + lane] : 0u, d0 = L[t], d1 = L[t + 1];
                        push_if(valid && is_cand(word_k<0>(d0, d1), 4u * t), 4u * t);
                        push_if(valid && is_cand(word_k<1>(d0, d1), 4u * t + 1), 4u * t + 1);
                        push_if(valid && is_cand(word_k<2>(d0, d1), 4u * t + 2), 4u * t + 2);
                        push_if(valid && is_cand(word_k<3>(d0, d1), 4u * t + 3), 4u * t + 3);
                    }
                } else { // low complexity: every position is looked at directly
                    for (uint32_t t0 = 0; t0 < nfull; t0 += 64) { // (to the end: c stays exact for the period test)
                        const bool valid = t0 + lane < nfull;
                        const uint32_t t = valid ? t0 + lane : 0u, d0 = L[t], d1 = L[t + 1];
                        push_if(valid && is_cand(word_k<0>(d0, d1), 4u * t), 4u * t);
                        push_if(valid && is_cand(word_k<1>(d0, d1), 4u * t + 1), 4u * t + 1);
                        push_if(valid && is_cand(word_k<2>(d0, d1), 4u * t + 2), 4u * t + 2);
                        push_if(valid && is_cand(word_k<3>(d0, d1), 4u * t + 3), 4u * t + 3);
                    }
                }
            } else { // a handful of positions
                if (lane < ne)
                    m = word(lane);
                m = wave_min(m);
                homo = ne >= 4 && (m & 0xFFFFu) == (m >> 16) && (m & 0xFFu) == ((m >> 8) & 0xFFu);
                push_if(lane < ne && word(lane) == m && !(homo && word(lane ? lane - 1 : ne - 1) == m), lane);
            }
        }
        wave_sync();
        // c occurrences of the least word, evenly spread if the sequence is periodic: a repetition of its first ne / c bytes?
        // (c is exact even when the list is full)
        if (c >= 2 && ne % c == 0) {
            const uint32_t d = ne / c;
            auto differs_at = [&](uint32_t p) { return byte_at(p) != byte_at(p + d >= ne ? p + d - ne : p + d); };
            if (__ballot(lane < ne && differs_at(lane)) == 0ull) {
                bool differs = false;
                for (uint32_t p0 = 64; p0 < ne && !differs; p0 += 64)
                    differs = __ballot(p0 + lane < ne && differs_at(p0 + lane)) != 0ull;
                if (!differs) {
                    ne = d;
                    goto search; // again, on one block
                }
            }
        }
        uint64_t r = 0; // c == 0: a homopolymer, all rotations equal, the smallest index is 0
        if (c != 0) {
            uint16_t *cur = listA, *nxt = listB;
            uint32_t depth = 4, rounds = 0, stalled = 0;
            bool serial = c > WLIST; // more candidates than the list holds: the exact two-pointer search below
            while (!serial && c > 64 && depth < ne) { // rounds over the list while the lanes cannot hold the candidates
                // a tandem repeat that does not close on itself keeps all its candidates until the seam comes into view:
                // three rounds without a loss and the two-pointer search takes over (it runs through a period in one step)
                if (++rounds > MAX_ROUNDS || stalled >= 3) {
                    serial = true;
                    break;
                }
                uint32_t mm = 0xFFFFFFFFu;
                for (uint32_t e = lane; e < c; e += 64)
                    mm = min(mm, word(cur[e] + depth));
                mm = wave_min(mm);
                uint32_t c2 = 0;
                for (uint32_t e0 = 0; e0 < c; e0 += 64) {
                    const bool valid = e0 + lane < c;
                    const uint32_t p = valid ? cur[e0 + lane] : 0u;
                    const bool keep = valid && word(p + depth) == mm;
                    const uint64_t mask = __ballot(keep);
                    if (keep)
                        nxt[c2 + lanes_below(mask)] = (uint16_t)p;
                    c2 += (uint32_t)__builtin_popcountll(mask);
                }
                wave_sync();
                stalled = c2 == c ? stalled + 1 : 0;
                c = c2;
                uint16_t *t = cur;
                cur = nxt;
                nxt = t;
                depth += 4;
            }
            if (!serial && c <= 64) { // one candidate per lane
                bool alive = lane < c;
                const uint32_t p = alive ? cur[lane] : 0u;
                while (c > 1 && depth < ne) {
                    if (++rounds > MAX_ROUNDS) {
                        serial = true;
                        break;
                    }
                    const uint32_t w = alive ? word(p + depth) : 0xFFFFFFFFu;
                    const uint32_t mm = wave_min(w);
                    alive = alive && w == mm;
                    c = (uint32_t)__builtin_popcountll(__ballot(alive));
                    depth += 4;
                }
                if (!serial)
                    r = wave_min(alive ? p : 0xFFFFFFFFu); // equal rotations (or a single one): the smallest index
            } else if (!serial) { // more than a wave of equal rotations
                uint32_t best = 0xFFFFFFFFu;
                for (uint32_t e = lane; e < c; e += 64)
                    best = min(best, cur[e]);
                r = wave_min(best);
            }
            if (serial)
                r = two_pointer_wave(lds, (uint64_t)ne);
        }
        if (lane == 0)
            (strand ? rot_rc : rot)[q] = r;
        if (rotated && strand == 0) { // RotateSequence: (s + s)[r : r + n], seqhash.go:131-137
            uint8_t *out = rotated + o0;
            const uint32_t r32 = (uint32_t)r;
            if (n32 >= 32) {
                const uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u)) & 15u;
                const uint32_t nd = (n32 - head) >> 4, tail0 = head + 16u * nd;
                if (lane < head)
                    out[lane] = (uint8_t)byte_at(lane + r32);
                if (lane < n32 - tail0)
                    out[tail0 + lane] = (uint8_t)byte_at(tail0 + lane + r32);
                uint4 *od = reinterpret_cast<uint4 *>(out + head);
                for (uint32_t t = lane; t < nd; t += 64) {
                    uint32_t p = head + 16u * t + r32;
                    p -= p >= n32 ? n32 : 0;
                    const uint32_t *src = L + (p >> 2);
                    const uint32_t sh = p & 3u;
                    const uint32_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3], a4 = src[4];
                    uint4 v;
                    v.x = __builtin_amdgcn_alignbyte(a1, a0, sh);
                    v.y = __builtin_amdgcn_alignbyte(a2, a1, sh);
                    v.z = __builtin_amdgcn_alignbyte(a3, a2, sh);
                    v.w = __builtin_amdgcn_alignbyte(a4, a3, sh);
                    od[t] = v;
                }
            } else if (lane < n32) {
                out[lane] = (uint8_t)byte_at(lane + r32);
            }
        }
        } // strand
    }
    if (raw_type >= 0 && __ballot((bad & 0x100u) != 0u) != 0ull && lane == 0)
        atomicOr(any_bad, 1u);
}

} // namespace k5
} // namespace polyhip

using namespace polyhip;

extern "C" {

} // extern "C"

// d_rot_rc != nullptr: also the least rotation of every sequence's reverse complement (upper-case letters, the complement
// table of transform.go:78-109) -- what seqhash.Hash needs for a circular double-stranded sequence (seqhash.go:180-193) --
// from ONE staging of the bytes per kernel, the second strand never written to memory
bool polyhip::k5_wave_takes_all(uint64_t max_len)
{
    uint64_t wave_max = k5::WAVE_SEQ_MAX;
    if (const char *e = getenv("POLYHIP_K5_WAVE_MAX"))
        wave_max = std::min<uint64_t>(strtoull(e, nullptr, 10), 32768);
    uint32_t lds_w = (uint32_t)std::min<uint64_t>(max_len, wave_max) + k5::WRAP;
    lds_w = (lds_w + 15u) & ~15u;
    return max_len + k5::WRAP <= lds_w;
}

int polyhip::k5_least_rotation_strands_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint64_t max_len,
                                           uint64_t *d_rot_index, uint8_t *d_rotated, uint64_t *d_rot_rc, polyhip_stream_t stream,
                                           int raw_type, uint8_t *d_norm_out, uint32_t *d_any_bad)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && d_rot_index, "polyhip_least_rotation_batch: null pointer");
    hipStream_t st = as_stream(stream);
    // 1. a wave per sequence for everything up to WAVE_SEQ_MAX bytes (POLYHIP_K5_WAVE_MAX overrides; testing aid): what it
    //    leaves -- longer sequences -- carries MARK in d_rot_index
    uint64_t wave_max = k5::WAVE_SEQ_MAX;
    if (const char *e = getenv("POLYHIP_K5_WAVE_MAX"))
        wave_max = std::min<uint64_t>(strtoull(e, nullptr, 10), 32768);
    uint32_t lds_w = (uint32_t)std::min<uint64_t>(max_len, wave_max) + k5::WRAP;
    lds_w = (lds_w + 15u) & ~15u;
    {
        const size_t smem = 4 * ((size_t)lds_w + 2 * k5::WLIST * 2);
        const unsigned blocks = (unsigned)std::min<uint64_t>((n + 3) / 4, 256ull * 8ull);
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k5::least_rotation_wave_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        PH_REQUIRE(raw_type < 0 || (d_norm_out && d_any_bad && max_len + k5::WRAP <= lds_w),
                   "k5_least_rotation_strands_dev: the normalising form needs every sequence within a wave's share");
        hipLaunchKernelGGL(k5::least_rotation_wave_kernel, dim3(blocks), dim3(256), smem, st, d_seqs, d_offsets, n, lds_w,
                           d_rot_index, d_rotated, d_rot_rc, raw_type, d_norm_out, d_any_bad);
        PH_HIP(hipGetLastError());
    }
    if (max_len + k5::WRAP <= lds_w)
        return POLYHIP_OK; // nothing was left
    // 2. a workgroup per marked sequence, staged in LDS up to LDS_SEQ_MAX (the allocation is sized to the batch's longest)
    uint64_t lds_seq = max_len + k5::WRAP;
    if (lds_seq > k5::LDS_SEQ_MAX)
        lds_seq = k5::LDS_SEQ_MAX;
    lds_seq = (lds_seq + 15) & ~15ull;
    // a workgroup reads `chunk` marks at a time: few sequences -> small chunks, so that the marked ones spread over the chip
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(k5::THREADS, n / (256ull * 32ull)));
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + chunk - 1) / chunk, 256ull * 32ull);
    auto kl = k5::least_rotation_kernel<true>;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_seq));
    hipLaunchKernelGGL(kl, dim3(blocks), dim3(k5::THREADS), lds_seq, st, d_seqs, d_offsets, n, lds_seq, d_rot_index,
                       d_rotated, chunk, 0);
    if (d_rot_rc) // the marked sequences' other strand: the same kernel on the reverse-complement view
        hipLaunchKernelGGL(kl, dim3(blocks), dim3(k5::THREADS), lds_seq, st, d_seqs, d_offsets, n, lds_seq, d_rot_rc,
                           (uint8_t *)nullptr, chunk, 1);
    PH_HIP(hipGetLastError());
    // 3. the same search reading global memory for what LDS cannot hold
    if (max_len + k5::WRAP > lds_seq) {
        hipLaunchKernelGGL((k5::least_rotation_kernel<false>), dim3(blocks), dim3(k5::THREADS), 0, st, d_seqs, d_offsets,
                           n, lds_seq, d_rot_index, d_rotated, chunk, 0);
        if (d_rot_rc)
            hipLaunchKernelGGL((k5::least_rotation_kernel<false>), dim3(blocks), dim3(k5::THREADS), 0, st, d_seqs, d_offsets,
                               n, lds_seq, d_rot_rc, (uint8_t *)nullptr, chunk, 1);
        PH_HIP(hipGetLastError());
    }
    return POLYHIP_OK;
}

extern "C" {

int polyhip_least_rotation_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint64_t max_len,
                                     uint64_t *d_rot_index, uint8_t *d_rotated, polyhip_stream_t stream)
{
    return k5_least_rotation_strands_dev(d_seqs, d_offsets, n, max_len, d_rot_index, d_rotated, nullptr, stream);
}

// the single-device body; `rotated` is indexed by the caller's offsets (a shard passes the whole batch's buffers)
static int least_rotation_batch_one(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint64_t *rot_index,
                                    uint8_t *rotated)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets && rot_index, "polyhip_least_rotation_batch: null pointer");
    uint64_t max_len = 0;
    for (uint64_t i = 0; i < n; ++i) {
        PH_REQUIRE(offsets[i] <= offsets[i + 1], "polyhip_least_rotation_batch: offsets not ascending at %llu",
                   (unsigned long long)(i + md::base().item));
        if (offsets[i + 1] - offsets[i] > max_len)
            max_len = offsets[i + 1] - offsets[i];
    }
    (void)max_len;
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    const Chunks ch = cut_packed(offsets, n, 8, HOST_CHUNK_BYTES);
    struct Slot {
        PackedSlot in;
        DevBuf drot, dout;
    } slot[2];
    for (size_t q = 0; q < std::min<size_t>(2, ch.count()); ++q) {
        PH_HIP(slot[q].in.alloc(ch, hs.s[q]));
        PH_HIP(slot[q].drot.alloc(ch.max_items * 8));
        if (rotated)
            PH_HIP(slot[q].dout.alloc(ch.max_bytes + 16));
    }
    // The rotated sequences are as many bytes as went up.  Their download on the pipeline's helper thread (Duplex) beside
    // the next chunk's upload does NOT pay here: with two slots chunk c + 2's upload waits for chunk c's download, and this
    // runtime's pageable copies slow each other down when both directions run (64 MB down takes 2.7 ms beside an upload
    // against 1.2 ms alone) -- 19.7 -> 18.5 ms on one box, 19.4 -> 22.6 on another.  K1, whose uploads are 2.5x its
    // downloads, gains 1.3x from the same helper; this call keeps the copies on the slot's own stream (init(0)).
    Duplex dx;
    PH_HIP(dx.init(0));
    for (size_t c = 0; c < ch.count(); ++c) {
        Slot &S = slot[c & 1];
        const uint64_t i0 = ch.cut[c], m = ch.cut[c + 1] - i0, cb = offsets[i0 + m] - offsets[i0];
        PH_HIP(dx.slot_free(c, S.in.st)); // chunk c-2 has left this slot
        PH_HIP(S.in.upload(seqs, offsets, i0, m));
        uint64_t ml = 0;
        for (uint64_t i = 0; i < m; ++i)
            ml = std::max(ml, offsets[i0 + i + 1] - offsets[i0 + i]);
        const int rc = polyhip_least_rotation_batch_dev(S.in.dseq.as<uint8_t>(), S.in.doff.as<uint64_t>(), m, ml, S.drot.as<uint64_t>(),
                                                        rotated ? S.dout.as<uint8_t>() : nullptr, S.in.st);
        if (rc != POLYHIP_OK) {
            (void)dx.finish();
            (void)hs.sync_both();
            return rc;
        }
        uint64_t *dst_rot = rot_index + i0;
        uint8_t *dst_seq = rotated && cb ? rotated + offsets[i0] : nullptr;
        const void *src_rot = S.drot.p, *src_seq = S.dout.p;
        PH_HIP(dx.download(c, S.in.st, [=](hipStream_t st) -> hipError_t {
            hipError_t e = hipMemcpyAsync(dst_rot, src_rot, m * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && dst_seq)
                e = hipMemcpyAsync(dst_seq, src_seq, cb, hipMemcpyDeviceToHost, st);
            return e;
        }));
    }
    PH_HIP(dx.finish());
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

int polyhip_least_rotation_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint64_t *rot_index,
                                 uint8_t *rotated)
{
    std::shared_ptr<md::Pool> P = n ? md::pool() : nullptr;
    if (!P)
        return least_rotation_batch_one(seqs, offsets, n, rot_index, rotated);
    // sequences are independent -- block split by bytes
    PH_REQUIRE(seqs && offsets && rot_index, "polyhip_least_rotation_batch: null pointer");
    const std::vector<uint64_t> cut =
        md::split(n, md::size(*P), [&](uint64_t i) { return (offsets[i] - offsets[0]) * (rotated ? 2 : 1) + i * 8; });
    return md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, offsets[i0] - offsets[0]);
        return least_rotation_batch_one(seqs, offsets + i0, m, rot_index + i0, rotated);
    });
}

} // extern "C"
