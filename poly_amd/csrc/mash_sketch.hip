// mash_sketch.hip -- K1: batched (*Mash).Sketch for gfx950.
//
// Replaces search/mash/mash.go:68-104 (+ murmur3.Sum32, mash.go:76) for a
// packed batch of sequences.  One 256-thread workgroup per sequence; the
// sequence is streamed through LDS in tiles of TW windows:
//
//   stage   tile bytes   global -> LDS   (coalesced dwords, funnel-shifted so
//                                         window 0 sits at LDS byte 0)
//   premix  P[p] = rotl(LE32(bytes p..p+3) * c1, 15) * c2   for every byte
//           position p of the tile: each 4-byte murmur3 block is mixed ONCE per
//           position and shared by the k/4 windows that use it
//           (2 multiplies per window instead of 2*(k/4))
//   hash    each lane owns 4 consecutive windows: k/4 ds_read_b128 of P, the
//           h = rotl(h ^ P, 13) * 5 + c chain, the tail byte(s) -- for
//           k % 4 == 1 a 256-entry LDS table of premix(byte) ^ k, the 4 windows'
//           tail bytes being one aligned dword -- and fmix32
//   select  FAST kernel: hashes <= tau are appended to a small shared LDS buffer (hipcc
//           aggregates the append per wave: one LDS atomic + lane ranks); tau is the value
//           a uniform hash would need for s + 6 sqrt(s) + 16 survivors, and the pass is
//           VERIFIED: fewer than s survivors, or more than the buffer holds, and the
//           sequence is marked in its output row for the GENERAL kernel (tau = 2^32-1, buffer of
//           s + one round of tiles, shrunk whenever it could overflow), so the output is
//           exact for any input.  Keeping the fast kernel's LDS at ~24 KB (6 workgroups per
//           CU) matters more than its instruction count: the kernel is latency bound
//           (halving the occupancy costs 1.6x, profiles/r01_k1_ablation.md).
//   shrink  exact bottom-s of the survivors by LDS counting sort on the top
//           bits (2048 bins) + in-bin ranking; duplicates keep distinct ranks,
//           as the reference keeps duplicate hashes.
//
// Integer VALU / LDS-latency bound (no MFMA: this is hashing, not a contraction).
// Algorithmic HBM bytes per sequence: len + 4*s (read once, sketch written once).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "host_pipeline.h"
#include "multi_device.h"
#include <memory>
#include <string>
#include <vector>

// PH_ABL != 0 only in scripts/ubench/k1_ablate.hip: knocks out one phase to measure its cost
// (results are then wrong by construction).  1 premix, 2 chain, 3 fmix+tail, 4 select, 5 bottom_s, 6 stage: the TILE pass.
// The SLAB pass (round 5): 11 = stage + premix + hash only (no select, no bottom-s), 12 = no bottom-s, 14 = the per-read
// prologue and barriers alone (no slabs, no bottom-s), 15 = no premix (the hash reads stale quads), 16 = ONE chain block of the
// k / 4 (-11 instructions per k-mer), 17 = fmix32 cut to one multiply (-4): is the kernel bound by its instruction count at all?
// Round 6: 21 = the fast bottom-s without its barriers, 22 = without its rank pass.  None of them marks a
// row for the general kernel, so the timed launch is the slab kernel alone.
#ifndef PH_WPE
#define PH_WPE 6 // waves per SIMD the fast kernel is register-allocated for (6 workgroups per CU fit its LDS)
#endif
#ifndef PH_CAPK
#define PH_CAPK 12u // candidate buffer of the fast pass: s + PH_CAPK * sqrt(s) + 64 entries
#endif
#ifndef PH_ABL
#define PH_ABL 0
#endif
#ifndef PH_SELBINS
#define PH_SELBINS 0 // EXPERIMENT (round 6, lever (a)): n > 0 = the slab pass counts the bottom-s's 2^n bins while it selects
#endif
#ifndef PH_SEL_ATOMIC
#define PH_SEL_ATOMIC 0 // EXPERIMENT (round 6): 1 = a survivor's slot from a returning LDS increment on the wave's own counter
#endif                  // (compare + shift per hash) instead of its ballot rank (compare + two mbcnt + shift-add)
#ifndef PH_BS_WIN
#define PH_BS_WIN 4 // the fast bottom-s ranks an element against its PH_BS_WIN neighbours on either side (0: against its whole bin)
#endif
#ifndef PH_BS_U
#define PH_BS_U 4 // candidates a thread takes per trip of the bottom-s's loops (bottom_s_fast): 4 / 5 / 6 measure the same
#endif
// slab pass (scripts/ubench/k1_ablate.hip sweeps these): sigmas of head-room of the survivor target over s, of a
// wave's segment and of the sorted buffer over their expectations, and the waves per SIMD it is allocated for
#ifndef PH_SLAB_SIG
#define PH_SLAB_SIG 6u
#endif
#ifndef PH_SLAB_CW
#define PH_SLAB_CW 6u
#endif
#ifndef PH_SLAB_CF
#define PH_SLAB_CF 0u // 0: the tile pass's capf
#endif
#ifndef PH_SLAB_WPE
#define PH_SLAB_WPE PH_WPE
#endif

namespace polyhip {
namespace k1 {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
#ifndef PH_WTW
#define PH_WTW 512
#endif
constexpr int WTW = PH_WTW;        // windows per WAVE tile (4 * GROUPS per lane)
constexpr int TW = WTW * WAVES;    // windows the workgroup covers per round
constexpr int GROUPS = WTW / (4 * 64);
constexpr int NB = 2048;  // counting-sort bins
// A read the verified fast pass cannot finish (too few survivors, a full buffer) is MARKED in its own output row --
// out[0] = 0xFFFFFFFF > out[1] = 0, which no sketch of a read with >= s windows can be (those are ascending) -- and the
// general kernel, which scans the batch for marks, redoes it.  No side list: the _dev entry point needs no scratch.
constexpr uint32_t MARK0 = 0xFFFFFFFFu, MARK1 = 0u;
constexpr int GBATCH = 32; // reads a general-kernel workgroup scans at a time
constexpr uint32_t WL_POSITIONAL = 0x80000000u; // work-list flag of the general kernel: fewer windows than SketchSize
constexpr uint32_t C1 = 0xcc9e2d51u, C2 = 0x1b873593u;

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t premix(uint32_t k)
{
    k *= C1;
    k = rotl32(k, 15);
    k *= C2;
    return k;
}

// one instruction (v_mad_u64_u32) for "* 5 + c" beats shift-add + add: issue slots, not ALU width, bound this kernel
__device__ __forceinline__ uint32_t chain(uint32_t h) { return rotl32(h, 13) * 5u + 0xe6546b64u; }

__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// bytes [sh, sh+4) of the little-endian pair hi:lo
__device__ __forceinline__ uint32_t funnel_bytes(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// Inclusive scan over the 64 lanes of a wave with DPP adds: four row shifts inside each row of 16,
// then the last lane of rows 0/2 into rows 1/3 and lane 31 into the upper half.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

struct Smem {
    uint32_t *seqb;   // tile bytes, window 0 at byte 0
    uint32_t *P;      // premixed blocks per byte position; aliased by `bins`
    uint32_t *cand;   // candidate hashes (<= tau)
    uint32_t *binned; // counting-sort scratch
    uint32_t *misc;   // [0] count  [1] tau  [3] survivors  [4..8) wave totals
    uint32_t *lut;    // [256] premix(byte) ^ k for a single tail byte (k % 4 == 1)
};

// ---- exact bottom-s ------------------------------------------------------------------
// The survivors are given by `for_each(f)`: every thread calls f(h) for its share.
// Returns (block-uniform) the number T of values <= tau.  If T >= s and T <= binned_cap:
// cand[0..s) = the s smallest ascending, misc[0] = s, misc[1] = cand[s-1].
// If T < s (and keep_partial): cand[0..T) = those values (any order), misc[0] = T.
// ---- bins that hold many values ------------------------------------------------------------------
// Repeated k-mers (tandem repeats, homopolymers, poly-A tails) put hundreds of EQUAL hashes into one bin of
// the counting sort, and ranking every element against its whole bin is quadratic in the bin size.  Bins
// with more than BIG_BIN values are therefore left out of the per-element ranking and handled by whole
// waves: the smallest not-yet-placed value of the bin (wave min), then every copy of it in index order
// (ballot prefix), and so on -- one pass per DISTINCT value, which is what such a bin has few of.
constexpr uint32_t BIG_BIN = 32;
constexpr uint32_t BIG_LIST_CAP = 512; // the list lives in seqb (>= WAVES * WTW / 4 dwords); more big bins than that: per-element ranking

template <class Emit>
__device__ __attribute__((noinline)) void rank_big_bins(const uint32_t *__restrict__ binned, const uint32_t *__restrict__ bins,
                                              const uint32_t *__restrict__ biglist, uint32_t nbig, uint32_t s, Emit emit)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t q = wave; q < nbig; q += WAVES) {
        const uint32_t b = biglist[q];
        const uint32_t st = b ? bins[b - 1] : 0u, en = bins[b]; // after the scatter bins[b] is the END of bin b
        uint32_t base = st, lastv = 0;
        bool first = true;
        while (base < s && base < en) {
            uint32_t m = 0xFFFFFFFFu;
            bool has = false;
            for (uint32_t x = st + lane; x < en; x += 64) {
                const uint32_t o = binned[x];
                if (first || o > lastv) {
                    m = min(m, o);
                    has = true;
                }
            }
            if (__ballot(has) == 0ull)
                break;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1)
                m = min(m, (uint32_t)__shfl_xor((int)m, d, 64));
            for (uint32_t x0 = st; x0 < en; x0 += 64) {
                const uint32_t x = x0 + lane;
                const bool eq = x < en && binned[x] == m;
                const uint64_t mask = __ballot(eq);
                const uint32_t pos = base + (uint32_t)__builtin_popcountll(mask & below);
                if (eq && pos < s)
                    emit(pos, m);
                base += (uint32_t)__builtin_popcountll(mask);
            }
            lastv = m;
            first = false;
        }
    }
}

template <class ForEach>
__device__ uint32_t bottom_s(const Smem &sm, uint32_t s, uint32_t tau, uint32_t binned_cap, bool keep_partial,
                             ForEach for_each)
{
    const int tid = threadIdx.x;
    uint32_t *bins = sm.P;
    // every survivor is <= tau: pick the shift that spreads [0, tau] over <= NB bins
    const int sig = 32 - __builtin_clz(tau | 1u);
    const int shift = sig > 11 ? sig - 11 : 0;
    __syncthreads(); // everyone is done with P before it becomes bins

    for (int b = tid; b < NB; b += THREADS)
        bins[b] = 0;
    if (tid == 0)
        sm.misc[8] = 0; // bins with more than BIG_BIN values
    __syncthreads();
    for_each([&](uint32_t h) {
        if (h <= tau)
            atomicAdd(&bins[h >> shift], 1u);
    });
    __syncthreads();

    // exclusive scan of NB bins: 8 per thread, wave scan, cross-wave fix-up
    {
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = bins[8 * tid + i];
            sum += v[i];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d)
                incl += t;
        }
        if ((tid & 63) == 63)
            sm.misc[4 + (tid >> 6)] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < (tid >> 6); ++w)
            run += sm.misc[4 + w];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bins[8 * tid + i] = run; // start of bin
            run += v[i];
            if (v[i] > BIG_BIN) { // seqb is idle while the candidates are sorted: it holds the list
                const uint32_t slot = atomicAdd(&sm.misc[8], 1u);
                if (slot < BIG_LIST_CAP)
                    sm.seqb[slot] = (uint32_t)(8 * tid + i);
            }
        }
        if (tid == THREADS - 1)
            sm.misc[3] = run; // T
    }
    __syncthreads();
    const uint32_t T = sm.misc[3];
    if (T > binned_cap || (T < s && !keep_partial))
        return T;
    // scatter: afterwards bins[b] = end of bin b, start of bin b = bins[b-1]
    for_each([&](uint32_t h) {
        if (h <= tau)
            sm.binned[atomicAdd(&bins[h >> shift], 1u)] = h;
    });
    __syncthreads();
    if (T < s) {
        for (uint32_t j = tid; j < T; j += THREADS)
            sm.cand[j] = sm.binned[j];
        __syncthreads();
        if (tid == 0)
            sm.misc[0] = T;
        __syncthreads();
        return T;
    }
    // rank inside the bin; ties broken by slot so duplicates get distinct ranks
    const uint32_t nbig = sm.misc[8];
    const bool by_waves = nbig <= BIG_LIST_CAP; // else the list is incomplete: rank every element the slow way
    for (uint32_t j = tid; j < T; j += THREADS) {
        const uint32_t h = sm.binned[j];
        const uint32_t b = h >> shift;
        const uint32_t start = b ? bins[b - 1] : 0u;
        if (start >= s)
            continue;
        const uint32_t end = bins[b];
        if (by_waves && end - start > BIG_BIN)
            continue; // a whole wave places this bin's values (rank_big_bins)
        uint32_t rank = 0;
        for (uint32_t x = start; x < end; ++x) {
            const uint32_t o = sm.binned[x];
            rank += (o < h) || (o == h && x < j);
        }
        const uint32_t pos = start + rank;
        if (pos < s)
            sm.cand[pos] = h;
    }
    if (by_waves && nbig)
        rank_big_bins(sm.binned, bins, sm.seqb, nbig, s, [cand = sm.cand](uint32_t pos, uint32_t h) { cand[pos] = h; });
    __syncthreads();
    if (tid == 0) {
        sm.misc[0] = s;
        sm.misc[1] = sm.cand[s - 1];
    }
    __syncthreads();
    return T;
}

// Tiles are WAVE-private: each wave stages, premixes and hashes its own WTW windows in its own
// slice of LDS, so the tile loop has no workgroup barrier at all (LDS operations of one wave
// execute in order; the fences below only stop the compiler from reordering them).
__device__ __forceinline__ void wave_sync()
{
    // A wavefront-scope fence would do, but hipcc lowers it to s_waitcnt vmcnt(0) lgkmcnt(0), which
    // also waits for the NEXT tile's global loads that were just put in flight.  The DS queue of a
    // wave is in order, so a compiler barrier is all that is needed.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---- staging: tile bytes [t0, t0 + WTW + k + 4) -> the wave's seqb, through registers so that
// the NEXT tile's global loads are in flight while the current tile is premixed and hashed
template <int KS> struct Stager {
    // dwords per lane for a compile-time k; runtime k stages without the register hop
    static constexpr int NI = KS > 0 ? ((WTW / 4 + KS / 4 + 2 + 3) + 63) / 64 : 1;
    uint32_t lo[NI], hi[NI]; // raw aligned dwords; the byte funnel is applied when they are stored

    __device__ __forceinline__ static void fetch(const uint32_t *__restrict__ gdw, uint32_t gsh, int64_t gbytes,
                                                 int64_t gi, bool full, uint32_t &l, uint32_t &h)
    {
        if (PH_ABL == 6) {
            l = (uint32_t)gi * 0x9E3779B9u;
            h = l;
            return;
        }
        h = 0u;
        if (full) { // the whole staging range lies inside the sequence's aligned view
            l = gdw[gi];
            if (gsh)
                h = gdw[gi + 1];
        } else {
            l = (gi * 4 < gbytes) ? gdw[gi] : 0u;
            if (gsh)
                h = ((gi + 1) * 4 < gbytes) ? gdw[gi + 1] : 0u;
        }
    }

    __device__ __forceinline__ void load(const uint32_t *__restrict__ gdw, uint32_t gsh, int64_t gbytes, int64_t t0,
                                         uint32_t n_seq_dw, bool full)
    {
        if (KS > 0) {
            const int64_t dbase = t0 >> 2; // t0 is a multiple of WTW, so of 4
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t d = (threadIdx.x & 63) + i * 64;
                if (d < n_seq_dw)
                    fetch(gdw, gsh, gbytes, dbase + d, full, lo[i], hi[i]);
            }
        }
    }

    __device__ __forceinline__ void store(uint32_t *__restrict__ seqb, const uint32_t *__restrict__ gdw, uint32_t gsh,
                                          int64_t gbytes, int64_t t0, uint32_t n_seq_dw, bool full)
    {
        if (KS > 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t d = (threadIdx.x & 63) + i * 64;
                if (d < n_seq_dw)
                    seqb[d] = gsh ? funnel_bytes(hi[i], lo[i], gsh) : lo[i];
            }
        } else { // runtime k: fetch + store in one go
            const int64_t dbase = t0 >> 2;
            for (uint32_t d = threadIdx.x & 63; d < n_seq_dw; d += 64) {
                uint32_t l, h;
                fetch(gdw, gsh, gbytes, dbase + d, full, l, h);
                seqb[d] = gsh ? funnel_bytes(h, l, gsh) : l;
            }
        }
    }
};

// ---- one staged wave tile: premix, then hash; `emit(w0, h[4])` consumes a lane's 4 hashes ----
template <int KS, class Emit>
__device__ __forceinline__ void tile_compute(const uint32_t *__restrict__ seqb, uint32_t *__restrict__ P,
                                             const uint32_t *__restrict__ lut, uint32_t k, Emit emit)
{
    const int lane = threadIdx.x & 63;
    const int nblk = (int)(k >> 2);
    const int tail = (int)(k & 3);
    const uint32_t tailmask = tail == 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * tail));

    // ---- premix: P[p..p+3] for p = 4*q
    {
        const int nq = (WTW >> 2) + nblk; // quads of byte positions needed
        for (int q = lane; q < (PH_ABL == 1 ? 0 : nq); q += 64) {
            const uint32_t d0 = seqb[q], d1 = seqb[q + 1];
            uint4 p;
            p.x = premix(d0);
            p.y = premix(funnel_bytes(d1, d0, 1));
            p.z = premix(funnel_bytes(d1, d0, 2));
            p.w = premix(funnel_bytes(d1, d0, 3));
            reinterpret_cast<uint4 *>(P)[q] = p;
        }
    }
    wave_sync();

    // ---- hash
    const uint4 *P4 = reinterpret_cast<const uint4 *>(P);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const int wq = lane + 64 * g; // quad of windows inside the tile
        uint32_t h[4] = {0u, 0u, 0u, 0u};
        if (KS > 0) {
#pragma unroll
            for (int j = 0; j < (PH_ABL == 2 ? 1 : (KS >> 2)); ++j) {
                const uint4 p = P4[wq + j];
                h[0] = chain(h[0] ^ p.x);
                h[1] = chain(h[1] ^ p.y);
                h[2] = chain(h[2] ^ p.z);
                h[3] = chain(h[3] ^ p.w);
            }
        } else {
            for (int j = 0; j < nblk; ++j) {
                const uint4 p = P4[wq + j];
                h[0] = chain(h[0] ^ p.x);
                h[1] = chain(h[1] ^ p.y);
                h[2] = chain(h[2] ^ p.z);
                h[3] = chain(h[3] ^ p.w);
            }
        }
        if (PH_ABL == 3) {
        } else if (tail == 1) {
            // the 4 windows' tail bytes are one aligned dword; lut[b] = premix(b) ^ k
            const uint32_t tb = seqb[wq + nblk];
            h[0] ^= lut[tb & 0xFFu];
            h[1] ^= lut[(tb >> 8) & 0xFFu];
            h[2] ^= lut[(tb >> 16) & 0xFFu];
            h[3] ^= lut[tb >> 24];
        } else if (tail) {
            const uint32_t d0 = seqb[wq + nblk], d1 = seqb[wq + nblk + 1];
            h[0] ^= premix(d0 & tailmask) ^ k;
            h[1] ^= premix(funnel_bytes(d1, d0, 1) & tailmask) ^ k;
            h[2] ^= premix(funnel_bytes(d1, d0, 2) & tailmask) ^ k;
            h[3] ^= premix(funnel_bytes(d1, d0, 3) & tailmask) ^ k;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                h[c] ^= k;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            h[c] = PH_ABL == 3 ? h[c] : fmix32(h[c]);
        emit(4u * (uint32_t)wq, h); // windows w0 .. w0+3 of the tile
    }
    wave_sync(); // seqb / P are rewritten by this wave's next tile
}

// All tiles of one sequence; wave w takes tiles w, w + WAVES, ...  LOCKSTEP: the workgroup
// meets at `before(T0)` (block-uniform hook; the general pass shrinks there) before every round
// of WAVES tiles; otherwise the waves run free.  emit_full / emit_part(t0, w0, h[4], nvalid): a lane's
// 4 consecutive hashes; nvalid == 4 always for emit_full (tiles whose WTW windows all exist).
template <int KS, bool LOCKSTEP, class Before, class EmitFull, class EmitPart>
__device__ __forceinline__ void run_tiles(const Smem &sm, const uint32_t *__restrict__ gdw, uint32_t gsh,
                                          int64_t gbytes, int64_t nwin, uint32_t k, uint32_t n_seq_dw, uint32_t n_P_w,
                                          Before before, EmitFull emit_full, EmitPart emit_part)
{
    const int wave = threadIdx.x >> 6;
    uint32_t *seqb = sm.seqb + wave * n_seq_dw;
    uint32_t *P = sm.P + wave * n_P_w;
    const int64_t stage_bytes = (int64_t)n_seq_dw * 4 + 4;
    auto is_full = [&](int64_t t0) { return (nwin - t0) >= (int64_t)WTW && t0 + stage_bytes <= gbytes; };
    Stager<KS> st;
    const int64_t first = (int64_t)wave * WTW;
    if (first < nwin)
        st.load(gdw, gsh, gbytes, first, n_seq_dw, is_full(first));
    for (int64_t T0 = 0; T0 < nwin; T0 += TW) {
        if (LOCKSTEP)
            before(T0);
        const int64_t t0 = T0 + first;
        if (t0 >= nwin)
            continue;
        const bool full = is_full(t0);
        st.store(seqb, gdw, gsh, gbytes, t0, n_seq_dw, full);
        wave_sync();
        if (t0 + TW < nwin)
            st.load(gdw, gsh, gbytes, t0 + TW, n_seq_dw, is_full(t0 + TW));
        // emit(t0, w0, h[4], nvalid): the lane's 4 hashes of windows t0 + w0 ..; the first nvalid exist
        if (full) {
            tile_compute<KS>(seqb, P, sm.lut, k, [&](uint32_t w0, const uint32_t(&h)[4]) { emit_full(t0, w0, h, 4u); });
        } else {
            const uint32_t wl = (uint32_t)((nwin - t0) < (int64_t)WTW ? (nwin - t0) : (int64_t)WTW);
            tile_compute<KS>(seqb, P, sm.lut, k, [&](uint32_t w0, const uint32_t(&h)[4]) {
                emit_part(t0, w0, h, w0 >= wl ? 0u : (wl - w0 < 4u ? wl - w0 : 4u));
            });
        }
    }
}

// Append the lane's hashes h[c] (c < nvalid, h[c] <= tau) to cand[]: ONE LDS atomic per wave for
// the whole quad (4 ballots, scalar popcounts), then every survivor stores at base + its rank.
// Slots beyond `cap` are dropped (the caller sees the total in *counter and redoes the sequence).
__device__ __forceinline__ void append4(uint32_t *__restrict__ counter, uint32_t *__restrict__ cand, uint32_t cap,
                                        const uint32_t (&h)[4], uint32_t nvalid, uint32_t tau)
{
    bool a[4];
    uint64_t m[4];
    uint32_t cnt[4], total = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = (PH_ABL == 4 ? h[c] == 12345u : h[c] <= tau) && (uint32_t)c < nvalid;
        m[c] = __ballot(a[c]);
        cnt[c] = (uint32_t)__popcll(m[c]);
        total += cnt[c];
    }
    if (total == 0) // wave-uniform
        return;
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0)
        base = atomicAdd(counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a[c]) {
            const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[c] >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)m[c], 0u));
            if (idx < cap)
                cand[idx] = h[c];
        }
        base += cnt[c];
    }
}

// ---- fast pass's bottom-s: cand[0..C) all <= tau0, s <= C <= capf.  Counting sort on the top
// bits + in-bin ranking, written straight to the output row.  All threads call it.
// `src` / `first` / `step` / `cnt`: where this thread's candidates are -- the shared buffer (sm.cand, tid, THREADS, C)
// or, for the slab pass, the thread's own wave's segment (segment, lane, 64, that wave's count).  FIN: the
// candidates still lack fmix32's last `h ^= h >> 16` (the slab pass thresholds on the bits that step leaves alone).
constexpr int BSU = PH_BS_U;
constexpr uint32_t BS_MARGIN = (PH_BS_WIN + 3u) & ~3u; // dwords kept free before binned[0] and behind binned[capf)
// (PH_ABL 21: the fast bottom-s without its workgroup barriers -- wrong results, timing only: the upper bound of what
// overlapping one read's bottom-s with the next read's hashing could give)
#define BS_SYNC()              \
    do {                       \
        if (PH_ABL != 21)      \
            __syncthreads();   \
    } while (0)
template <bool FIN, bool SELB = false>
__device__ void bottom_s_fast(const Smem &sm, uint32_t s, uint32_t tau, uint32_t C, uint32_t nbf_log2,
                              uint32_t *__restrict__ outp, const uint32_t *__restrict__ src, uint32_t first,
                              uint32_t step, uint32_t cnt, uint32_t *__restrict__ selbins = nullptr)
{
    const int tid = threadIdx.x;
    auto fin = [](uint32_t h) { return FIN ? h ^ (h >> 16) : h; };
    uint32_t *bins = sm.P;
    const uint32_t nbf = 1u << nbf_log2; // 1024 or 2048 bins, whatever fits in the P region
    const int sig = 32 - __builtin_clz(tau | 1u);
    const int shift = sig > (int)nbf_log2 ? sig - (int)nbf_log2 : 0;
    // a thread owns `per` = 4 or 8 consecutive bins = one or two 16-byte words (the caller's barrier freed P)
    uint4 *bins4 = reinterpret_cast<uint4 *>(bins);
    const int q4 = (int)(nbf / (4 * THREADS)); // 1 or 2
    if (!SELB) {
        for (int q = 0; q < q4; ++q)
            bins4[q4 * tid + q] = make_uint4(0, 0, 0, 0);
        if (tid == 0)
            sm.misc[8] = 0; // bins with more than BIG_BIN values
    }
    if (PH_BS_WIN) { // the window pass below reads PH_BS_WIN entries on either side of binned[0, C): nothing there may count
        if (tid < PH_BS_WIN)
            sm.binned[-1 - tid] = 0u;
        else if (tid < 2 * PH_BS_WIN)
            sm.binned[C + (uint32_t)tid - PH_BS_WIN] = 0xFFFFFFFFu;
    }
    if (!SELB) // (SELB: the select counted the bins, the caller zeroed misc[8] before its barrier; the margins are read two barriers on)
        BS_SYNC();
    // the loops over candidates are unrolled (PH_BS_U) so that the LDS round trips of a thread's elements overlap instead of
    // queueing behind each other.  A wave's segment holds ~300 survivors and the sorted buffer ~1200, i.e. 4.7 per lane / per
    // thread, so by four every loop runs a second, nearly empty trip -- but by five or six the kernel measures the same
    // (1.46-1.50 ms per 100k reads all three, profiles/r05_k1_bottom_s_unroll.log): the bottom-s is its five barriers and
    // the atomics' round trips, not its instruction count.
    if (!SELB) {
        for (uint32_t i0 = first; i0 < cnt; i0 += BSU * step) {
            uint32_t h[BSU];
#pragma unroll
            for (int u = 0; u < BSU; ++u)
                h[u] = i0 + u * step < cnt ? fin(src[i0 + u * step]) : 0u;
#pragma unroll
            for (int u = 0; u < BSU; ++u)
                if (i0 + u * step < cnt)
                    atomicAdd(&bins[h[u] >> shift], 1u);
        }
        BS_SYNC();
    }
    {
        uint4 v[2];
        uint32_t sum = 0;
        uint4 *cnt4 = SELB ? reinterpret_cast<uint4 *>(selbins) : bins4;
        for (int q = 0; q < 2; ++q) {
            v[q] = q < q4 ? cnt4[q4 * tid + q] : make_uint4(0, 0, 0, 0);
            if (SELB && q < q4)
                cnt4[q4 * tid + q] = make_uint4(0, 0, 0, 0); // ready for the next read's select
            sum += v[q].x + v[q].y + v[q].z + v[q].w;
        }
        const uint32_t incl = wave_incl_scan(sum);
        if ((tid & 63) == 63)
            sm.misc[4 + (tid >> 6)] = incl;
        BS_SYNC();
        uint32_t run = incl - sum;
        for (int w = 0; w < (tid >> 6); ++w)
            run += sm.misc[4 + w];
        for (int q = 0; q < 2; ++q) {
            if (q < q4) {
                uint4 o; // start of each bin
                o.x = run;
                o.y = o.x + v[q].x;
                o.z = o.y + v[q].y;
                o.w = o.z + v[q].z;
                run = o.w + v[q].w;
                bins4[q4 * tid + q] = o;
                if (max(max(v[q].x, v[q].y), max(v[q].z, v[q].w)) > BIG_BIN) { // rare: repeated k-mers
                    const uint32_t cnt4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (cnt4[c] > BIG_BIN) {
                            const uint32_t slot = atomicAdd(&sm.misc[8], 1u);
                            if (slot < BIG_LIST_CAP)
                                sm.seqb[slot] = (uint32_t)(4 * (q4 * tid + q) + c);
                        }
                }
            }
        }
    }
    BS_SYNC();
    for (uint32_t i0 = first; i0 < cnt; i0 += BSU * step) { // afterwards bins[b] = end of bin b
        uint32_t h[BSU], at[BSU];
#pragma unroll
        for (int u = 0; u < BSU; ++u)
            h[u] = i0 + u * step < cnt ? fin(src[i0 + u * step]) : 0u;
#pragma unroll
        for (int u = 0; u < BSU; ++u)
            if (i0 + u * step < cnt)
                at[u] = atomicAdd(&bins[h[u] >> shift], 1u);
#pragma unroll
        for (int u = 0; u < BSU; ++u)
            if (i0 + u * step < cnt)
                sm.binned[at[u]] = h[u];
    }
    BS_SYNC();
    const uint32_t nbig = sm.misc[8];
    const bool by_waves = nbig <= BIG_LIST_CAP; // else the list is incomplete: rank every element the slow way
    // an element against its whole bin: start of the bin + the values of the bin that sort before it (ties by slot:
    // duplicates keep distinct ranks)
    auto rank_in_bin = [&](uint32_t j, uint32_t hv) {
        const uint32_t b = hv >> shift;
        const uint32_t start = b ? bins[b - 1] : 0u, end = bins[b];
        if (start >= s || (by_waves && end - start > BIG_BIN)) // (a whole wave places a big bin's values: rank_big_bins)
            return;
        uint32_t pos = start;
        for (uint32_t x = start; x < end; ++x) {
            const uint32_t o = sm.binned[x];
            pos += (o < hv) || (o == hv && x < j);
        }
        if (pos < s)
            outp[pos] = hv;
    };
#if PH_BS_WIN
    // binned[] is sorted up to the order INSIDE the bins, and with ~0.6 values per bin a bin rarely holds more than
    // three.  So an element's place is its slot, minus the left neighbours that are larger, plus the right neighbours that
    // are smaller: a left neighbour of another bin is smaller and a right one larger by construction, so no bin test and
    // no bins[] lookup -- one compare and one add-with-carry per neighbour.  Exact whenever the bin cannot reach beyond
    // the window, i.e. when the farthest neighbour on either side is of another bin; the few elements for which it is not
    // (a bin of five or more: 0.2 % of them) are ranked against their whole bin as before.
    if (PH_ABL != 22) { // (PH_ABL 22: no rank pass -- timing only)
        const uint32_t lim = 1u << shift; // (a ^ b) < lim: same bin
        for (uint32_t j = tid; j < C; j += THREADS) {
            const uint32_t *__restrict__ w = sm.binned + j;
            const uint32_t hv = w[0];
            uint32_t pos = j;
            bool far;
            {
                uint32_t l[PH_BS_WIN], r[PH_BS_WIN];
#pragma unroll
                for (int d = 0; d < PH_BS_WIN; ++d) {
                    l[d] = w[-1 - d];
                    r[d] = w[1 + d];
                }
#pragma unroll
                for (int d = 0; d < PH_BS_WIN; ++d) {
                    pos -= l[d] > hv ? 1u : 0u;
                    pos += r[d] < hv ? 1u : 0u;
                }
                far = ((l[PH_BS_WIN - 1] ^ hv) < lim) | ((r[PH_BS_WIN - 1] ^ hv) < lim);
            }
            if (__builtin_expect(far, 0))
                rank_in_bin(j, hv);
            else if (pos < s)
                outp[pos] = hv;
        }
    }
#else
    for (uint32_t j = tid; j < C; j += THREADS)
        rank_in_bin(j, sm.binned[j]);
#endif
    if (by_waves && nbig) // rare: keep it out of line (and out of the common path's register budget)
        rank_big_bins(sm.binned, bins, sm.seqb, nbig, s, [outp](uint32_t pos, uint32_t hv) { outp[pos] = hv; }); // by VALUE: a reference would
                                                                                                            // park `outp` in scratch, per read
}

struct ReadView {
    const uint32_t *gdw; // dword-aligned view of the sequence
    uint32_t gsh;        // byte offset of the sequence inside it
    int64_t gbytes, nwin;
};

__device__ __forceinline__ ReadView view(const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ offs,
                                         uint64_t r, uint32_t k)
{
    ReadView v;
    const uint64_t o0 = offs[r], o1 = offs[r + 1];
    const int64_t n = (int64_t)(o1 - o0);
    v.nwin = n - (int64_t)k; // mash.go:73: len-k windows, last k-mer skipped
    // derived from the kernel argument so the loads stay global_load
    v.gsh = (uint32_t)((uintptr_t)(seqs + o0) & 3);
    v.gdw = reinterpret_cast<const uint32_t *>(seqs + o0 - v.gsh);
    v.gbytes = n + v.gsh;
    return v;
}

// ---- FAST kernel: every sequence, under a verified threshold -------------------------------
// KS > 0: k known at compile time (full unroll of the block chain); KS == 0: runtime k.
// Persistent workgroups: each one walks the batch with stride gridDim.x.  Sequences the fast
// pass cannot finish exactly are marked (MARK0 / MARK1) for the general kernel.
template <int KS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(PH_WPE, 8))) void sketch_fast_kernel(const uint8_t *__restrict__ seqs,
                                                             const uint64_t *__restrict__ offs, uint64_t nseq,
                                                             uint32_t k_rt, uint32_t s, uint32_t *__restrict__ out,
                                                             uint32_t n_seq_dw, uint32_t n_P_w, uint32_t n_P,
                                                             uint32_t capf, uint32_t nbf_log2)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_raw[];
    Smem sm;
    sm.seqb = smem_raw;                    // WAVES slices of n_seq_dw
    sm.P = sm.seqb + WAVES * n_seq_dw;     // WAVES slices of n_P_w (whole region doubles as `bins`)
    sm.cand = sm.P + n_P;
    sm.binned = sm.cand + capf + BS_MARGIN; // (bottom_s_fast's window reads BS_MARGIN entries on either side)
    sm.misc = sm.binned + capf + BS_MARGIN;
    sm.lut = sm.misc + 16;

    const uint32_t k = KS > 0 ? (uint32_t)KS : k_rt;
    const int tid = threadIdx.x;
    sm.lut[tid] = premix((uint32_t)tid) ^ k; // THREADS == 256; visible after the first barrier
    auto nothing = [](int64_t) {};

    for (uint64_t r = blockIdx.x; r < nseq; r += gridDim.x) {
        const ReadView rv = view(seqs, offs, r, k);
        if (rv.nwin <= 0)
            continue;
        uint32_t *__restrict__ outp = out + r * (uint64_t)s;

        // ---- mash.go:81-84: fewer windows than SketchSize -> positional, unsorted, tail untouched
        if (rv.nwin < (int64_t)s) {
            __syncthreads();
            auto put = [&](int64_t t0, uint32_t w0, const uint32_t(&h)[4], uint32_t nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if ((uint32_t)c < nvalid)
                        outp[t0 + w0 + c] = h[c];
            };
            run_tiles<KS, false>(sm, rv.gdw, rv.gsh, rv.gbytes, rv.nwin, k, n_seq_dw, n_P_w, nothing, put, put);
            continue;
        }

        // threshold a uniform hash would need for s + 6 sqrt(s) + 16 survivors
        uint32_t tau0 = 0xFFFFFFFFu;
        {
            const uint64_t target = (uint64_t)s + 6ull * (uint64_t)__builtin_sqrtf((float)s) + 16ull;
            if ((int64_t)target < rv.nwin)
                tau0 = (uint32_t)((target << 32) / (uint64_t)rv.nwin);
        }
        bool ok = tau0 != 0xFFFFFFFFu || rv.nwin <= (int64_t)capf; // short sequence: keep every hash
        __syncthreads(); // the previous sequence is done with LDS
        if (tid == 0)
            sm.misc[0] = 0;
        __syncthreads();
        uint32_t C = 0;
        if (ok) {
            auto keep = [&](int64_t, uint32_t, const uint32_t(&h)[4], uint32_t nvalid) {
                append4(&sm.misc[0], sm.cand, capf, h, nvalid, tau0);
            };
            run_tiles<KS, false>(sm, rv.gdw, rv.gsh, rv.gbytes, rv.nwin, k, n_seq_dw, n_P_w, nothing, keep, keep);
            __syncthreads();
            C = sm.misc[0];
            ok = C >= s && C <= capf; // enough survivors, none lost
        }
        if (ok) {
            if (PH_ABL == 5) {
                for (uint32_t i = tid; i < s; i += THREADS)
                    outp[i] = sm.cand[i];
            } else {
                bottom_s_fast<false>(sm, s, tau0, C, nbf_log2, outp, sm.cand, (uint32_t)tid, THREADS, C);
            }
        } else if (tid == 0) {
            outp[0] = MARK0;
            outp[1] = MARK1;
        }
    }
}

// ---- SLAB pass (the fast kernel for compile-time k): the same select-under-a-verified-threshold scheme with
// the per-window overheads of the tile loop taken out.
//   * A wave walks a CONTIGUOUS quarter of the read in slabs of 256 windows (4 per lane), and both its byte buffer
//     and its premixed-block buffer are RINGS of two slabs with the first few entries duplicated behind the end,
//     so a window that runs into the next slab reads on without a wrap.  Staging a slab is one dword per lane and
//     premixing it one quad of byte positions per lane: no partial third pass for the k-1 bytes that tiles of
//     independent windows have to overlap (that pass cost a quarter of the tile loop's staging + premix).
//     Premix unit u covers quads [64u-1, 64u+63) -- shifted by one so that its last quad needs no byte of slab u+1;
//     order per step: stage slab i+1, premix unit i+1, hash slab i.
//   * Every wave appends to its OWN candidate segment; the running count is a scalar register, so the tile loop
//     has no LDS atomic and no round trip, and a survivor's slot is mbcnt(ballot) seeded with that count.
//   * The threshold is applied to the hash before fmix32's last `h ^= h >> 16` against tau | 0xFFFF (that step
//     cannot change the top 16 bits), and only the ~12 % survivors take the last step, in the bottom-s.
// Verified like the tile pass: a wave whose segment overflows, or fewer than s survivors in total, sends the read
// to the general kernel.
template <int KS> struct Slabs {
    static constexpr int NBLK = KS >> 2, TAIL = KS & 3;
    static constexpr int DUPD = NBLK + 2; // dwords [0, DUPD) of the ring live again at [128, 128 + DUPD)
    const uint32_t *__restrict__ gdw;
    uint32_t gsh;
    int64_t gbytes;
    uint32_t *__restrict__ seqb; // physical index = ring index + 1; [0] repeats ring dword 127
    uint32_t *__restrict__ P;    // quads [0, 128) + quads [0, NBLK) again at [128, 128 + NBLK)
    const uint32_t *__restrict__ lut;
    int lane;

    // dword 64u + lane of the read's aligned view (and its successor when the read does not start on a dword);
    // u is wave-uniform, so the slab's base is a scalar pointer and the lane adds a constant offset.  Slabs below
    // `u_inside` lie, with one dword beyond them, inside the read: no per-lane guard.
    uint32_t u_inside;
    __device__ __forceinline__ void gload(uint32_t u, uint32_t &lo, uint32_t &hi) const
    {
        const uint32_t *__restrict__ slab = gdw + (uint64_t)u * 64;
        hi = 0u;
        if (__builtin_expect(u < u_inside, 1)) {
            lo = slab[lane];
            if (gsh)
                hi = slab[lane + 1];
        } else {
            const int64_t left = gbytes - (int64_t)u * 256; // bytes of the view from this slab on (may be <= 0)
            lo = (int64_t)lane * 4 < left ? slab[lane] : 0u;
            if (gsh)
                hi = (int64_t)(lane + 1) * 4 < left ? slab[lane + 1] : 0u;
        }
    }

    template <int PAR> __device__ __forceinline__ void stage(uint32_t lo, uint32_t hi) const
    {
        const uint32_t v = gsh ? funnel_bytes(hi, lo, gsh) : lo;
        seqb[1 + 64 * PAR + lane] = v;
        if (PAR == 0) {
            if (lane < DUPD)
                seqb[129 + lane] = v;
        } else if (lane == 63) {
            seqb[0] = v;
        }
    }

    template <int PAR> __device__ __forceinline__ void premix_unit() const
    {
        const uint32_t d0 = seqb[64 * PAR + lane], d1 = seqb[64 * PAR + lane + 1];
        uint4 p;
        p.x = premix(d0);
        p.y = premix(funnel_bytes(d1, d0, 1));
        p.z = premix(funnel_bytes(d1, d0, 2));
        p.w = premix(funnel_bytes(d1, d0, 3));
        uint4 *P4 = reinterpret_cast<uint4 *>(P);
        if (PAR == 0) {
            P4[(lane + 127) & 127] = p; // quads 127, 0, 1, ..., 62
            if (lane >= 1 && lane <= NBLK)
                P4[127 + lane] = p;
        } else {
            P4[63 + lane] = p;
        }
    }

    // the lane's 4 windows of a slab: hashes WITHOUT fmix32's last xor-shift
    template <int PAR> __device__ __forceinline__ void hash(uint32_t (&h)[4]) const
    {
        const uint4 *b = reinterpret_cast<const uint4 *>(P) + 64 * PAR + lane;
        h[0] = h[1] = h[2] = h[3] = 0u;
#pragma unroll
        for (int j = 0; j < (PH_ABL == 16 ? 1 : NBLK); ++j) { // (PH_ABL 16: one chain block of the k / 4 -- timing only)
            const uint4 p = b[j];
            h[0] = chain(h[0] ^ p.x);
            h[1] = chain(h[1] ^ p.y);
            h[2] = chain(h[2] ^ p.z);
            h[3] = chain(h[3] ^ p.w);
        }
        const uint32_t *t = seqb + 1 + 64 * PAR + lane + NBLK;
        if (TAIL == 1) {
            // the 4 windows' tail bytes are one aligned dword; lut[b] = premix(b) ^ k.  Byte select and the * 4 of the
            // LDS address in ONE SDWA shift per byte (hipcc spends a v_bfe_u32 + a v_lshl_add_u32 on each)
            const uint32_t tb = t[0];
            uint32_t off[4];
            const uint32_t two = 2u;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off[0]) : "v"(two), "v"(tb));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off[1]) : "v"(two), "v"(tb));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off[2]) : "v"(two), "v"(tb));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off[3]) : "v"(two), "v"(tb));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                h[c] ^= *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lut) + off[c]);
        } else if (TAIL) {
            constexpr uint32_t tailmask = 0xFFFFFFFFu >> (32 - 8 * (TAIL ? TAIL : 1));
            const uint32_t d0 = t[0], d1 = t[1];
            h[0] ^= premix(d0 & tailmask) ^ (uint32_t)KS;
            h[1] ^= premix(funnel_bytes(d1, d0, 1) & tailmask) ^ (uint32_t)KS;
            h[2] ^= premix(funnel_bytes(d1, d0, 2) & tailmask) ^ (uint32_t)KS;
            h[3] ^= premix(funnel_bytes(d1, d0, 3) & tailmask) ^ (uint32_t)KS;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                h[c] ^= (uint32_t)KS;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { // fmix32 up to its last multiply
            uint32_t x = h[c];
            if (PH_ABL != 17) { // (PH_ABL 17: no fmix32 -- timing only)
                x ^= x >> 16;
                x *= 0x85ebca6bu;
                x ^= x >> 13;
            }
            x *= 0xc2b2ae35u;
            h[c] = x;
        }
    }
};

// survivors of the lane's 4 hashes -> the wave's own segment; `cnt` is wave-uniform (kept in a scalar register)
template <bool PARTIAL>
__device__ __forceinline__ void append_own(uint32_t *__restrict__ seg, uint32_t capw, uint32_t &cnt, const uint32_t (&h)[4],
                                           uint32_t nvalid, uint32_t tauq, uint32_t *__restrict__ selbins = nullptr,
                                           uint32_t selshift = 0)
{
    bool a[4];
    uint64_t m[4];
    uint32_t n[4], total = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = h[c] <= tauq && (!PARTIAL || (uint32_t)c < nvalid);
        m[c] = __ballot(a[c]);
        n[c] = (uint32_t)__popcll(m[c]);
        total += n[c];
    }
    uint32_t base = __builtin_amdgcn_readfirstlane(cnt);
    cnt = base + total;
#if PH_SEL_ATOMIC
    {
        // ds_inc wraps at capw - 1, so a slot is always inside the segment; the scalar count above is what tells an overflow
        uint32_t *ctr = seg + capw; // (the experiment keeps the wave's counter behind its segment: capw is one entry smaller)
        uint32_t slot[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (a[c])
                slot[c] = atomicInc(ctr, capw - 1u);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (a[c])
                seg[slot[c]] = h[c];
        return;
    }
#endif
    if (cnt <= capw) { // wave-uniform; an overflowing wave stops storing and the read is redone
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t *__restrict__ at = seg + base; // scalar: the lane only adds its rank among the survivors
            if (a[c]) {
                at[__builtin_amdgcn_mbcnt_hi((uint32_t)(m[c] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[c], 0u))] = h[c];
                if (PH_SELBINS) // the bin is in the top 16 bits, which fmix32's last xor-shift leaves alone
                    atomicAdd(&selbins[h[c] >> selshift], 1u);
            }
            base += n[c];
        }
    }
}

// all slabs of one read; returns this wave's survivor count (wave-uniform; > capw: the segment overflowed)
template <int KS>
__device__ __forceinline__ uint32_t run_slabs(const Smem &sm, const ReadView &rv, uint32_t n_seq_dw, uint32_t n_P_w,
                                              uint32_t *__restrict__ seg, uint32_t capw, uint32_t tauq,
                                              uint32_t *__restrict__ selbins = nullptr, uint32_t selshift = 0)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // scalar: slab indices and ring bases stay in SGPRs
    Slabs<KS> S;
    S.gdw = rv.gdw;
    S.gsh = rv.gsh;
    S.gbytes = rv.gbytes;
    S.seqb = sm.seqb + wave * n_seq_dw;
    S.P = sm.P + wave * n_P_w;
    S.lut = sm.lut;
    S.lane = threadIdx.x & 63;
    // 32-bit slab indices: a read of 2^40 bytes does not fit the 288 GB of HBM
    const uint32_t nslab = (uint32_t)((rv.nwin + 255) >> 8);
    const uint32_t spw = (nslab + WAVES - 1) / WAVES;
    const uint32_t a = (uint32_t)wave * spw, b = a + spw < nslab ? a + spw : nslab; // this wave hashes slabs [a, b)
    S.u_inside = rv.gbytes >= 65 * 4 ? (uint32_t)(((rv.gbytes >> 2) - 65) >> 6) + 1u : 0u;
    uint32_t cnt = 0;
#if PH_SEL_ATOMIC
    capw -= 1u; // the last entry of the segment is the wave's slot counter
    if (S.lane == 0)
        seg[capw] = 0u;
#endif
    if (a >= b)
        return cnt;
    uint32_t l1, h1;
    S.gload(a, l1, h1);
    if (a & 1) {
        S.template stage<1>(l1, h1);
        wave_sync();
        S.gload(a + 1, l1, h1);
        S.template premix_unit<1>();
    } else {
        S.template stage<0>(l1, h1);
        wave_sync();
        S.gload(a + 1, l1, h1);
        S.template premix_unit<0>();
    }
    for (uint32_t i = a; i < b; ++i) {
        uint32_t h[4];
        // stage slab i+1 from the registers, then put slab i+2's loads in flight in the same registers: they land
        // while this slab is premixed and hashed
        if (i & 1) {
            S.template stage<0>(l1, h1);
            wave_sync();
            S.gload(i + 2, l1, h1);
            if (PH_ABL != 15)
                S.template premix_unit<0>();
            wave_sync();
            S.template hash<1>(h);
        } else {
            S.template stage<1>(l1, h1);
            wave_sync();
            S.gload(i + 2, l1, h1);
            if (PH_ABL != 15)
                S.template premix_unit<1>();
            wave_sync();
            S.template hash<0>(h);
        }
        const int64_t left = rv.nwin - ((int64_t)i << 8); // windows of the read from this slab on
        if (PH_ABL == 11) { // keeps the hashes alive, stores (almost) never
            if ((h[0] ^ h[1] ^ h[2] ^ h[3]) == 0x12345u)
                seg[0] = h[0];
        } else if (__builtin_expect(left >= 256, 1)) {
            append_own<false>(seg, capw, cnt, h, 4u, tauq, selbins, selshift);
        } else {
            const int64_t mine = left - 4 * S.lane;
            append_own<true>(seg, capw, cnt, h, mine <= 0 ? 0u : (mine < 4 ? (uint32_t)mine : 4u), tauq, selbins, selshift);
        }
        wave_sync(); // the rings are rewritten by the next step
    }
    return cnt;
}

template <int KS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(PH_SLAB_WPE, 8))) void sketch_slab_kernel(
    const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ offs, uint64_t nseq, uint32_t s, uint32_t *__restrict__ out,
    uint32_t n_seq_dw, uint32_t n_P_w, uint32_t n_P, uint32_t capw, uint32_t capf, uint32_t nbf_log2)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_raw[];
    Smem sm;
    sm.lut = smem_raw;                 // first, so that a tail byte's entry is at LDS address 4 * byte (no base to add)
    sm.seqb = smem_raw + 256;          // WAVES rings of n_seq_dw
    sm.P = sm.seqb + WAVES * n_seq_dw; // WAVES rings of n_P_w (whole region doubles as `bins`)
    sm.cand = sm.P + n_P;              // WAVES segments of capw
    sm.binned = sm.cand + WAVES * capw + BS_MARGIN;
    sm.misc = sm.binned + capf + BS_MARGIN;
    constexpr uint32_t k = (uint32_t)KS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    sm.lut[tid] = premix((uint32_t)tid) ^ k; // THREADS == 256; visible after the first barrier
    uint32_t *seg = sm.cand + wave * capw;
    uint32_t *selbins = sm.misc + 16; // (PH_SELBINS only: 2^nbf_log2 counters behind everything else)
    if (PH_SELBINS)
        for (uint32_t b = tid; b < (1u << nbf_log2); b += THREADS)
            selbins[b] = 0u;

    for (uint64_t r = blockIdx.x; r < nseq; r += gridDim.x) {
        const ReadView rv = view(seqs, offs, r, k);
        if (rv.nwin <= 0)
            continue;
        uint32_t *__restrict__ outp = out + r * (uint64_t)s;
        if (rv.nwin < (int64_t)s)
            continue; // mash.go:81-84: positional, unsorted, tail untouched -- the general kernel picks these up
        // threshold a uniform hash would need for s + 6 sqrt(s) + 16 survivors, rounded up to 16 bits
        uint32_t tauq = 0xFFFFFFFFu;
        {
            const uint64_t target = (uint64_t)s + (uint64_t)PH_SLAB_SIG * (uint64_t)__builtin_sqrtf((float)s) + 16ull;
            if ((int64_t)target < rv.nwin)
                tauq = (uint32_t)((target << 32) / (uint64_t)rv.nwin) | 0xFFFFu;
        }
        __syncthreads(); // the previous read is done with LDS
        uint32_t selshift = 0;
        if (PH_SELBINS) {
            const int sig = 32 - __builtin_clz(tauq | 1u);
            selshift = sig > (int)nbf_log2 ? (uint32_t)(sig - (int)nbf_log2) : 0u; // bottom_s_fast's shift (experiment: needs >= 16)
            if (tid == 0)
                sm.misc[8] = 0;
        }
        const uint32_t cw = PH_ABL == 14 ? tauq >> 31 : run_slabs<KS>(sm, rv, n_seq_dw, n_P_w, seg, capw, tauq, selbins, selshift);
        if ((tid & 63) == 0)
            sm.misc[10 + wave] = cw;
        __syncthreads();
        if (PH_ABL == 11 || PH_ABL == 12 || PH_ABL == 14) { // no bottom-s, and no mark either (the general kernel stays out)
            if (sm.misc[10] + sm.misc[11] + sm.misc[12] + sm.misc[13] == 0xFFFFFFF0u)
                outp[0] = cw;
            continue;
        }
        const uint32_t c0 = sm.misc[10], c1 = sm.misc[11], c2 = sm.misc[12], c3 = sm.misc[13];
        const uint32_t C = c0 + c1 + c2 + c3;
        const bool ok = max(max(c0, c1), max(c2, c3)) <= capw - (PH_SEL_ATOMIC ? 1u : 0u) && C >= s && C <= capf; // enough survivors, none lost
        if (ok)
            bottom_s_fast<true, PH_SELBINS != 0>(sm, s, tauq, C, nbf_log2, outp, seg, (uint32_t)(tid & 63), 64u, cw, selbins);
        else {
            if (PH_SELBINS) // (a read that is handed on leaves its counts behind)
                for (uint32_t b = tid; b < (1u << nbf_log2); b += THREADS)
                    selbins[b] = 0u;
            if (tid == 0) {
                outp[0] = MARK0;
                outp[1] = MARK1;
            }
        }
    }
}

// ---- GENERAL kernel: the marked reads (and, behind the slab pass, the positional ones), any input ------------
// Accepts every hash, shared candidate buffer of cap >= s + TW + 64, shrunk to the exact bottom-s
// whenever a round of tiles might overflow it.
template <int KS>
__global__ __launch_bounds__(THREADS) void sketch_general_kernel(const uint8_t *__restrict__ seqs,
                                                                const uint64_t *__restrict__ offs, uint32_t k_rt,
                                                                uint32_t s, uint32_t *__restrict__ out,
                                                                uint32_t n_seq_dw, uint32_t n_P_w, uint32_t n_P,
                                                                uint32_t cap, uint64_t nseq, int take_positional)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_raw[];
    Smem sm;
    sm.seqb = smem_raw;
    sm.P = sm.seqb + WAVES * n_seq_dw;
    sm.cand = sm.P + n_P;
    sm.binned = sm.cand + cap;
    sm.misc = sm.binned + cap;
    sm.lut = sm.misc + 16;
    // A workgroup scans GBATCH consecutive reads at a time and then works through the ones that need this kernel.
    // Small batches: when every read is marked (low-complexity input) the work must still spread over the whole chip.
    __shared__ uint32_t wl[GBATCH];
    __shared__ uint32_t nwl;

    const uint32_t k = KS > 0 ? (uint32_t)KS : k_rt;
    const int tid = threadIdx.x;
    sm.lut[tid] = premix((uint32_t)tid) ^ k;

    for (uint64_t base = (uint64_t)blockIdx.x * GBATCH; base < nseq; base += (uint64_t)gridDim.x * GBATCH) {
        __syncthreads();
        if (tid == 0)
            nwl = 0;
        __syncthreads();
        {   // one read per thread: marked by the fast pass, or (slab pass in front) positional?
            const uint64_t r = base + tid;
            if (tid < GBATCH && r < nseq) {
                const int64_t nwin = (int64_t)(offs[r + 1] - offs[r]) - (int64_t)k;
                uint32_t need = 0;
                if (nwin >= (int64_t)s) {
                    const uint32_t *row = out + r * (uint64_t)s;
                    if (row[0] == MARK0 && row[1] == MARK1)
                        need = 1;
                } else if (nwin > 0 && take_positional) {
                    need = 2;
                }
                if (need)
                    wl[atomicAdd(&nwl, 1u)] = (uint32_t)tid | (need == 2 ? WL_POSITIONAL : 0u);
            }
        }
        __syncthreads();
        const uint32_t nredo = nwl;
    for (uint32_t q = 0; q < nredo; ++q) {
        const uint32_t entry = wl[q];
        const uint64_t r = base + (entry & ~WL_POSITIONAL);
        const ReadView rv = view(seqs, offs, r, k);
        uint32_t *__restrict__ outp = out + r * (uint64_t)s;
        __syncthreads(); // the previous sequence is done with LDS
        if (entry & WL_POSITIONAL) { // fewer windows than SketchSize (mash.go:81-84), left to this kernel by the slab pass
            auto put = [&](int64_t t0, uint32_t w0, const uint32_t(&h)[4], uint32_t nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if ((uint32_t)c < nvalid)
                        outp[t0 + w0 + c] = h[c];
            };
            run_tiles<KS, false>(sm, rv.gdw, rv.gsh, rv.gbytes, rv.nwin, k, n_seq_dw, n_P_w, [](int64_t) {}, put, put);
            continue;
        }
        if (tid == 0) {
            sm.misc[0] = 0;
            sm.misc[1] = 0xFFFFFFFFu;
        }
        __syncthreads();
        auto shrink_shared = [&]() {
            const uint32_t C = sm.misc[0];
            bottom_s(sm, s, sm.misc[1], cap, true, [&](auto f) {
                for (uint32_t i = tid; i < C; i += THREADS)
                    f(sm.cand[i]);
            });
        };
        auto append = [&](int64_t, uint32_t, const uint32_t(&h)[4], uint32_t nvalid) {
            append4(&sm.misc[0], sm.cand, cap, h, nvalid, sm.misc[1]);
        };
        run_tiles<KS, true>(
            sm, rv.gdw, rv.gsh, rv.gbytes, rv.nwin, k, n_seq_dw, n_P_w,
            [&](int64_t) {
                __syncthreads(); // every wave's appends of the previous round are counted
                const bool room = sm.misc[0] + (uint32_t)TW <= cap; // this round may append up to TW
                __syncthreads();
                if (!room)
                    shrink_shared();
            },
            append, append);
        __syncthreads();
        shrink_shared();
        for (uint32_t i = tid; i < s; i += THREADS)
            outp[i] = sm.cand[i];
    }
    }
}


// ---- WIDE kernel: SketchSize or KmerSize beyond what the LDS layouts above hold --------------------------------
// (mash.New(21, 10000) is ordinary usage; a KmerSize of thousands is not, but the reference takes any.)  One workgroup per
// read, every window hashed straight from global memory -- no premixed blocks shared between windows, k/4 block mixes
// per window --, candidates and the sort buffer in a GLOBAL scratch slice per workgroup (2 * cap words, allocated
// stream-ordered by the entry point), the same exact bottom-s as the general kernel (counting sort with its bins in LDS).
// Rounds of WIDE_ROUND windows; the buffer is shrunk to the s smallest whenever a round might overflow it, and after a
// shrink only hashes at most tau = the s-th smallest so far are appended.
constexpr uint32_t WIDE_ROUND = 4 * THREADS;

__device__ __forceinline__ uint32_t murmur3_window(const uint8_t *__restrict__ p, uint32_t k)
{
    uint32_t h = 0;
    const uint32_t nblk = k >> 2;
    for (uint32_t j = 0; j < nblk; ++j) {
        uint32_t w;
        __builtin_memcpy(&w, p + 4 * (size_t)j, 4); // any alignment: the compiler picks the loads
        h = chain(h ^ premix(w));
    }
    const uint8_t *t = p + 4 * (size_t)nblk;
    uint32_t w = 0;
    switch (k & 3u) {
    case 3: w ^= (uint32_t)t[2] << 16; [[fallthrough]];
    case 2: w ^= (uint32_t)t[1] << 8; [[fallthrough]];
    case 1: w ^= (uint32_t)t[0]; h ^= premix(w); break;
    default: break;
    }
    return fmix32(h ^ k);
}

__global__ __launch_bounds__(THREADS) void sketch_wide_kernel(const uint8_t *__restrict__ seqs,
                                                             const uint64_t *__restrict__ offs, uint64_t nseq, uint32_t k,
                                                             uint32_t s, uint32_t *__restrict__ out,
                                                             uint32_t *__restrict__ scratch, uint32_t cap)
{
    __shared__ uint32_t bins[NB];
    __shared__ uint32_t biglist[BIG_LIST_CAP];
    __shared__ uint32_t misc[16];
    Smem sm;
    sm.seqb = biglist;
    sm.P = bins;
    sm.cand = scratch + (size_t)blockIdx.x * 2 * cap;
    sm.binned = sm.cand + cap;
    sm.misc = misc;
    sm.lut = nullptr;
    const int tid = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < nseq; r += gridDim.x) {
        const uint64_t o0 = offs[r];
        const int64_t nwin = (int64_t)(offs[r + 1] - o0) - (int64_t)k; // mash.go:73
        if (nwin <= 0)
            continue;
        const uint8_t *__restrict__ sp = seqs + o0;
        uint32_t *__restrict__ outp = out + r * (uint64_t)s;
        if (nwin < (int64_t)s) { // mash.go:81-84: positional, unsorted, tail untouched
            for (int64_t w = tid; w < nwin; w += THREADS)
                outp[w] = murmur3_window(sp + w, k);
            continue;
        }
        __syncthreads(); // the previous read is done with LDS and the scratch
        if (tid == 0) {
            misc[0] = 0;
            misc[1] = 0xFFFFFFFFu;
        }
        __syncthreads();
        auto shrink = [&]() {
            const uint32_t C = misc[0];
            bottom_s(sm, s, misc[1], cap, true, [&](auto f) {
                for (uint32_t i = tid; i < C; i += THREADS)
                    f(sm.cand[i]);
            });
        };
        for (int64_t base = 0; base < nwin; base += WIDE_ROUND) {
            __syncthreads(); // the previous round's appends are counted
            const bool room = misc[0] + WIDE_ROUND <= cap;
            __syncthreads();
            if (!room)
                shrink();
            const uint32_t tau = misc[1];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t w = base + c * THREADS + tid;
                if (w < nwin) {
                    const uint32_t h = murmur3_window(sp + w, k);
                    if (h <= tau)
                        sm.cand[atomicAdd(&misc[0], 1u)] = h;
                }
            }
        }
        __syncthreads();
        shrink();
        for (uint32_t i = tid; i < s; i += THREADS)
            outp[i] = sm.cand[i];
    }
}

// ---- SketchSize 0 and 1: what the reference does read by read (mash.go:68-104 with maxShiftedSketchSize = -1 / 0) ----
// s == 0: the first window already evaluates Sketches[-1] (:96) -> the read PANICS iff it has a window.
// s == 1: window 0 fills Sketches[0] (:88-92); any later window whose hash is below it is stored and then Sketches[-1] is
//         read (:98) -> PANICS iff some later hash undercuts the first one; otherwise Sketches[0] = hash of window 0.
// first_panic: the smallest read index on which the reference panics (atomicMin; ~0 = none).
__global__ __launch_bounds__(THREADS) void sketch_tiny_kernel(const uint8_t *__restrict__ seqs,
                                                             const uint64_t *__restrict__ offs, uint64_t nseq, uint32_t k,
                                                             uint32_t s, uint32_t *__restrict__ out,
                                                             unsigned long long *__restrict__ first_panic)
{
    const int tid = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < nseq; r += gridDim.x) {
        const uint64_t o0 = offs[r];
        const int64_t nwin = (int64_t)(offs[r + 1] - o0) - (int64_t)k;
        if (nwin <= 0)
            continue;
        if (s == 0) {
            if (tid == 0)
                atomicMin(first_panic, (unsigned long long)r);
            continue;
        }
        const uint8_t *__restrict__ sp = seqs + o0;
        const uint32_t h0 = murmur3_window(sp, k);
        bool under = false;
        for (int64_t w = 1 + tid; w < nwin && !under; w += THREADS)
            under = murmur3_window(sp + w, k) < h0;
        if (__syncthreads_or(under ? 1 : 0)) {
            if (tid == 0)
                atomicMin(first_panic, (unsigned long long)r);
        } else if (tid == 0) {
            out[r] = h0;
        }
    }
}

struct Launch {
    uint32_t n_seq_dw, n_P_w, n_P, n_P_fast, nbf_log2, nbf_log2_slab, capf, cap, capw, capf_slab;
    size_t smem_fast, smem_general, smem_slab;
};

static Launch plan(uint32_t k, uint32_t s)
{
    Launch L;
    const uint32_t nblk = k / 4;
    L.n_seq_dw = ((WTW / 4 + nblk + 2) + 3u) & ~3u;   // per wave
    L.n_P_w = WTW + 4 * nblk;                         // per wave
    L.n_P = WAVES * L.n_P_w;
    L.n_P_fast = L.n_P < 1024u ? 1024u : L.n_P; // the fast pass sorts with 1024 bins if that is all P offers
    L.nbf_log2 = L.n_P_fast >= 2048u ? 11u : 10u;
    if (L.n_P < (uint32_t)NB)
        L.n_P = NB; // `bins` aliases P
    const uint32_t s4 = (s + 3u) & ~3u;
    // fast pass: expected survivors s + 6 sqrt(s) + 16, standard deviation ~ sqrt(s): 12 sigma of room
    uint32_t rt = 1;
    while ((uint64_t)rt * rt < s)
        ++rt;
    L.capf = (s4 + PH_CAPK * rt + 64u + 63u) & ~63u;
    L.cap = s4 + TW + 64u; // general pass: shrink to s, then one more round always fits
    const size_t common = (size_t)WAVES * L.n_seq_dw + L.n_P + 16 + 256;
    L.smem_fast = ((size_t)WAVES * L.n_seq_dw + L.n_P_fast + 16 + 256 + 2 * (size_t)L.capf + 2 * BS_MARGIN) * 4;
    if (PH_ABL == 7)
        L.smem_fast = 70 * 1024; // occupancy probe
    L.smem_general = (common + 2 * (size_t)L.cap) * 4;
    // slab pass: one segment per wave (a quarter of the expected survivors + PH_SLAB_CW sigma of that quarter), and
    // a sorted buffer for all of them
    {
        const uint32_t target = s + PH_SLAB_SIG * rt + 16u;
        const uint32_t exp_w = (target + WAVES - 1) / WAVES;
        uint32_t rw = 1;
        while ((uint64_t)rw * rw < exp_w)
            ++rw;
        L.capw = (exp_w + PH_SLAB_CW * rw + 8u + 63u) & ~63u;
        L.capf_slab = PH_SLAB_CF ? ((target + PH_SLAB_CF * rt + 8u + 63u) & ~63u) : L.capf;
    }
    L.smem_slab = ((size_t)WAVES * L.n_seq_dw + L.n_P_fast + 16 + 256 + (size_t)WAVES * L.capw + (size_t)L.capf_slab + 2 * BS_MARGIN) * 4;
    L.nbf_log2_slab = L.nbf_log2;
    if (PH_SELBINS) {
        L.nbf_log2_slab = PH_SELBINS;
        L.smem_slab += (size_t)4 << PH_SELBINS;
    }
#ifdef PH_LDS_PAD
    L.smem_slab += PH_LDS_PAD; // occupancy probe
#endif
    return L;
}

static unsigned persistent_grid(size_t smem, uint64_t n)
{
    // as many workgroups as fit the chip at once (LDS-limited, <= 8 per CU), a few rounds deep
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(8, (160 * 1024) / std::max<size_t>(smem, 1)));
    return (unsigned)std::min<uint64_t>(n, 256 * per_cu * 4);
}

template <int KS>
static int launch(const uint8_t *d_seqs, const uint64_t *d_offs, uint64_t n, uint32_t k, uint32_t s,
                  uint32_t *d_out, const Launch &L, hipStream_t st)
{
    auto general = sketch_general_kernel<KS>;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(general), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)L.smem_general));
    bool slabs = false;
    if constexpr (KS > 0) {
        // POLYHIP_K1_SLABS=0 keeps the tile pass (testing aid: the two passes are cross-checked in tests/)
        slabs = !env_is("POLYHIP_K1_SLABS", '0') && L.smem_slab <= 64 * 1024;
        if (slabs) {
            auto slab = sketch_slab_kernel<KS>;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(slab), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L.smem_slab));
            hipLaunchKernelGGL(slab, dim3(persistent_grid(L.smem_slab, n)), dim3(THREADS), L.smem_slab, st, d_seqs, d_offs, n, s,
                               d_out, L.n_seq_dw, L.n_P_w, L.n_P_fast, L.capw, L.capf_slab, L.nbf_log2_slab);
        }
    }
    if (!slabs) {
        auto fast = sketch_fast_kernel<KS>;
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fast), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)L.smem_fast));
        hipLaunchKernelGGL(fast, dim3(persistent_grid(L.smem_fast, n)), dim3(THREADS), L.smem_fast, st, d_seqs, d_offs, n,
                           k, s, d_out, L.n_seq_dw, L.n_P_w, L.n_P_fast, L.capf, L.nbf_log2);
    }
    PH_HIP(hipGetLastError());
    // scans the batch (one read per thread) for marked rows -- and, behind the slab pass, for reads with fewer windows
    // than SketchSize; normally there are none and the workgroups are done after the scan
    const unsigned ggrid = (unsigned)std::min<uint64_t>((n + GBATCH - 1) / GBATCH, persistent_grid(L.smem_general, n));
    hipLaunchKernelGGL(general, dim3(ggrid), dim3(THREADS), L.smem_general, st, d_seqs, d_offs, k, s, d_out, L.n_seq_dw,
                       L.n_P_w, L.n_P, L.cap, n, slabs ? 1 : 0);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

} // namespace k1
} // namespace polyhip

using namespace polyhip;

extern "C" {

int polyhip_mash_sketch_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint32_t k,
                                  uint32_t s, uint32_t *d_out, polyhip_stream_t stream)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && (d_out || s == 0), "polyhip_mash_sketch_batch: null pointer");
    PH_REQUIRE(s <= (1u << 24), "polyhip_mash_sketch_batch: SketchSize %u > 2^24 is not implemented", s);
    hipStream_t st = as_stream(stream);
    if (s < 2) {
        // the reference's behaviour read by read (see sketch_tiny_kernel); the verdict needs the device's answer, so this
        // rare path synchronises the stream
        unsigned long long *d_first = nullptr, h_first = ~0ull;
        PH_HIP(hipMallocAsync(reinterpret_cast<void **>(&d_first), sizeof h_first, st));
        StreamFree guard{d_first, st}; // an early return below must not leak the stream-ordered allocation
        PH_HIP(hipMemsetAsync(d_first, 0xFF, sizeof h_first, st));
        hipLaunchKernelGGL(k1::sketch_tiny_kernel, dim3((unsigned)std::min<uint64_t>(n, 4096)), dim3(k1::THREADS), 0, st, d_seqs,
                           d_offsets, n, k, s, d_out, d_first);
        PH_HIP(hipGetLastError());
        PH_HIP(hipMemcpyAsync(&h_first, d_first, sizeof h_first, hipMemcpyDeviceToHost, st));
        PH_HIP(guard.release());
        PH_HIP(hipStreamSynchronize(st));
        if (h_first != ~0ull)
            return set_error(POLYHIP_ERR_PANIC,
                             "mash.Sketch with SketchSize %u indexes Sketches[-1] on sequence %llu (mash.go:%s): the reference panics",
                             s, h_first + (unsigned long long)md::base().item, s == 0 ? "96" : "98");
        return POLYHIP_OK;
    }
    const k1::Launch L = k1::plan(k <= 4096 ? k : 4096, s <= 8192 ? s : 8192);
    if (s > 8192 || k > 4096 || L.smem_general > 160 * 1024) {
        // beyond the LDS layouts: the wide kernel, candidates in a stream-ordered global scratch
        const uint32_t s4 = (s + 3u) & ~3u;
        const uint32_t cap = 2 * s4 + 2 * k1::WIDE_ROUND;
        // two candidate buffers of `cap` hashes per workgroup: the grid shrinks so that the scratch stays within 4 GiB
        // (s = 2^24 is 268 MB per workgroup: 16 of them; s = 10,000 keeps its 1024)
        const uint64_t per_wg = 2ull * cap * sizeof(uint32_t), budget = 4ull << 30;
        const unsigned grid = (unsigned)std::max<uint64_t>(
            1, std::min<uint64_t>(std::min<uint64_t>(n, s > 16384 ? 256 : 1024), budget / per_wg));
        uint32_t *scratch = nullptr;
        PH_HIP(hipMallocAsync(reinterpret_cast<void **>(&scratch), (size_t)grid * per_wg, st));
        StreamFree guard{scratch, st};
        hipLaunchKernelGGL(k1::sketch_wide_kernel, dim3(grid), dim3(k1::THREADS), 0, st, d_seqs, d_offsets, n, k, s, d_out, scratch,
                           cap);
        PH_HIP(hipGetLastError());
        PH_HIP(guard.release());
        return POLYHIP_OK;
    }
    const uint64_t CHUNK = 1ull << 30;
    for (uint64_t i0 = 0; i0 < n; i0 += CHUNK) {
        const uint64_t m = n - i0 < CHUNK ? n - i0 : CHUNK;
        const uint64_t *offs = d_offsets + i0;
        uint32_t *outp = d_out + i0 * (uint64_t)s;
        int rc;
        switch (k) {
        case 17: rc = k1::launch<17>(d_seqs, offs, m, k, s, outp, L, st); break;
        case 21: rc = k1::launch<21>(d_seqs, offs, m, k, s, outp, L, st); break;
        case 31: rc = k1::launch<31>(d_seqs, offs, m, k, s, outp, L, st); break;
        default: rc = k1::launch<0>(d_seqs, offs, m, k, s, outp, L, st); break;
        }
        if (rc != POLYHIP_OK)
            return rc;
    }
    return POLYHIP_OK;
}

// the single-device body: the calling thread's current device (a fan-out worker's, or the caller's own)
static int sketch_batch_one(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s, uint32_t *out)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets && (out || s == 0), "polyhip_mash_sketch_batch: null pointer");
    PH_REQUIRE(s <= (1u << 24), "polyhip_mash_sketch_batch: SketchSize %u > 2^24 is not implemented", s);
    // `out` is in/out: rows of sequences with fewer than s windows keep (part of) the caller's prior
    // Sketches (mash.go:81-84), so those rows have to travel to the device first; a batch without such
    // sequences -- the normal case -- skips that upload.
    bool need_prior = false;
    for (uint64_t i = 0; i < n; ++i) {
        PH_REQUIRE(offsets[i] <= offsets[i + 1], "polyhip_mash_sketch_batch: offsets not ascending at %llu",
                   (unsigned long long)(i + md::base().item));
        need_prior |= offsets[i + 1] - offsets[i] < (uint64_t)k + s;
    }
    // Chunks of about 256 MB (sequence bytes + sketch bytes) through two slots, each with its own stream:
    // the upload and the kernel of chunk c overlap the download of chunk c-1.
    const uint64_t CHUNK = 256ull << 20, row = (uint64_t)s * sizeof(uint32_t);
    std::vector<uint64_t> cut{0};
    uint64_t max_bytes = 0, max_reads = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i, sz = 0;
        do {
            sz += offsets[j + 1] - offsets[j] + row;
            ++j;
        } while (j < n && sz < CHUNK);
        cut.push_back(j);
        max_bytes = std::max(max_bytes, offsets[j] - offsets[i]);
        max_reads = std::max(max_reads, j - i);
        i = j;
    }
    // Two slots, each with its own stream for uploads and kernels; the downloads run on a helper thread and a third stream
    // (host_pipeline.h Downloader): a copy to or from pageable memory holds the thread that issued it, so one thread alone
    // alternates between the two directions of the link -- 200k reads x 10 kb took 51.9 ms (54 GB/s, up + down) that way.
    struct Slot {
        DevBuf dseq, doff, dout;
        hipStream_t st = nullptr;
        hipEvent_t computed = nullptr;
        std::vector<uint64_t> hoff;
        ~Slot()
        {
            if (st) {
                (void)hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
            }
            if (computed)
                (void)hipEventDestroy(computed);
        }
    } slot[2];
    struct DlStream {
        hipStream_t s = nullptr;
        ~DlStream()
        {
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
        }
    } dls;
    const size_t nchunks = cut.size() - 1;
    int first_panic = POLYHIP_OK;
    std::string panic_text;
    for (size_t q = 0; q < std::min<size_t>(2, nchunks); ++q) {
        PH_HIP(slot[q].dseq.alloc(max_bytes + 16));
        PH_HIP(slot[q].doff.alloc((max_reads + 1) * sizeof(uint64_t)));
        PH_HIP(slot[q].dout.alloc(max_reads * row));
        PH_HIP(hipStreamCreateWithFlags(&slot[q].st, hipStreamNonBlocking));
        PH_HIP(hipEventCreateWithFlags(&slot[q].computed, hipEventDisableTiming));
        slot[q].hoff.resize(max_reads + 1);
    }
    PH_HIP(hipStreamCreateWithFlags(&dls.s, hipStreamNonBlocking));
    int dev = 0;
    PH_HIP(hipGetDevice(&dev));
    int rc_loop = POLYHIP_OK;
    {
        // (a single chunk -- a single sequence from mash.Sketch -- has nothing to overlap: no helper thread, the copy runs here)
        std::unique_ptr<Downloader> dlp(nchunks >= 2 ? new Downloader(dev) : nullptr); // joined at the end of this block
        size_t pushed = 0;
        for (size_t c = 0; c < nchunks; ++c) {
            Slot &S = slot[c & 1];
            const uint64_t i0 = cut[c], m = cut[c + 1] - i0, b0 = offsets[i0];
            if (c >= 2) { // chunk c-2 has left this slot: its download is through (and with it its kernels and uploads)
                const hipError_t e = dlp->wait(c - 1);
                if (e != hipSuccess) {
                    rc_loop = set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_batch: download: %s", hipGetErrorString(e));
                    break;
                }
            }
            for (uint64_t i = 0; i <= m; ++i)
                S.hoff[i] = offsets[i0 + i] - b0; // rebased: the device copy starts at byte 0
            hipError_t e = hipMemcpyAsync(S.doff.p, S.hoff.data(), (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, S.st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(S.dseq.p, seqs + b0, offsets[i0 + m] - b0, hipMemcpyHostToDevice, S.st);
            if (e == hipSuccess && need_prior)
                e = hipMemcpyAsync(S.dout.p, out + i0 * (uint64_t)s, m * row, hipMemcpyHostToDevice, S.st);
            if (e != hipSuccess) {
                rc_loop = set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_batch: upload: %s", hipGetErrorString(e));
                break;
            }
            int rc;
            {
                md::BaseScope pos(i0, b0 - offsets[0]); // a message of this chunk names positions of the whole batch
                rc = polyhip_mash_sketch_batch_dev(S.dseq.as<uint8_t>(), S.doff.as<uint64_t>(), m, k, s, S.dout.as<uint32_t>(), S.st);
            }
            if (rc == POLYHIP_ERR_PANIC && first_panic == POLYHIP_OK) {
                // SketchSize < 2 is decided read by read (mash.go:96,98): the first panicking sequence is named, and the rows
                // of the sequences that do not panic are written -- in the later chunks too
                first_panic = rc;
                panic_text = polyhip_last_error();
            } else if (rc != POLYHIP_OK && rc != POLYHIP_ERR_PANIC) {
                rc_loop = rc;
                break;
            }
            if ((e = hipEventRecord(S.computed, S.st)) != hipSuccess) {
                rc_loop = set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_batch: %s", hipGetErrorString(e));
                break;
            }
            uint32_t *dst = out + i0 * (uint64_t)s;
            const void *src = S.dout.p;
            hipEvent_t ev = S.computed;
            hipStream_t ds = dls.s;
            const size_t bytes = m * row;
            auto job = [=]() -> hipError_t {
                hipError_t x = hipStreamWaitEvent(ds, ev, 0);
                if (x == hipSuccess)
                    x = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ds);
                if (x == hipSuccess)
                    x = hipStreamSynchronize(ds);
                return x;
            };
            if (dlp) {
                dlp->push(job);
                ++pushed;
            } else if ((e = job()) != hipSuccess) {
                rc_loop = set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_batch: download: %s", hipGetErrorString(e));
                break;
            }
        }
        const hipError_t e = dlp ? dlp->wait(pushed) : hipSuccess;
        if (e != hipSuccess && rc_loop == POLYHIP_OK)
            rc_loop = set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_batch: download: %s", hipGetErrorString(e));
    }
    if (rc_loop != POLYHIP_OK)
        return rc_loop;
    if (first_panic != POLYHIP_OK)
        return set_error(first_panic, "%s", panic_text.c_str());
    return POLYHIP_OK;
}

int polyhip_mash_sketch_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s,
                              uint32_t *out)
{
    std::shared_ptr<md::Pool> P = n ? md::pool() : nullptr;
    if (!P)
        return sketch_batch_one(seqs, offsets, n, k, s, out);
    // reads are independent -- contiguous blocks of reads balanced by bytes (sequence + sketch row), one per
    // device of the list, each through its own two-slot pipeline over its own PCIe link
    PH_REQUIRE(seqs && offsets && (out || s == 0), "polyhip_mash_sketch_batch: null pointer");
    const uint64_t row = (uint64_t)s * sizeof(uint32_t);
    const std::vector<uint64_t> cut =
        md::split(n, md::size(*P), [&](uint64_t i) { return offsets[i] - offsets[0] + i * row; });
    return md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, offsets[i0] - offsets[0]);
        return sketch_batch_one(seqs, offsets + i0, m, k, s, out ? out + i0 * (uint64_t)s : nullptr);
    });
}

} // extern "C"
