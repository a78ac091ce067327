// sw_batch.hip -- K3: batched Smith-Waterman score pass for gfx950.
//
// Replaces the fill loop + argmax of align.SmithWaterman
// (search/align/align.go:171-203) for a batch of pairs:
//   H[i][j] = max(0, H[i-1][j-1] + S(a_i,b_j), H[i-1][j] + gap, H[i][j-1] + gap)
// linear gap, arbitrary (possibly asymmetric) substitution matrix, argmax =
// first maximum in row-major order (i over A outer, j over B inner).
//
// This file holds the entry points, the plan that picks a path (polyhip_sw_last_path), and three kernels; the
// packed two-pairs-per-lane pass (the default at BASELINE config 4) lives in sw_packed.hip, the one-wave-per-pair
// kernel (small batches, ties of the packed pass, reads of 257..4096 symbols) in sw_wave.hip.
//
//  sw_shared_kernel<RA, CP>  -- 32-bit lane-per-pair kernel (BASELINE config 4: 1M x 150 bp
//    reads against ONE shared 5 kb reference).  Inter-sequence parallel: one
//    pair per lane, all 64 lanes of a wave walk the SAME column b_j.  The whole
//    H column of the pair (RA rows, int32) lives in VGPRs; the reference is
//    turned once into a byte "profile" prof[j][code] = S(sym(code), b_j)
//    (profile_kernel) that is streamed through LDS in chunks of <= 1024
//    columns, so a cell costs one ds_read_i8 + ~6 VALU ops and no global
//    traffic at all.  The row-major-first argmax is folded into one integer max
//    per cell on the key (h << 18 | (255-i) << 10 | (1023-jrel)); chunks are
//    visited in increasing j, so folding chunk winners with a strict compare
//    on (h, 255-i) keeps the reference's tie-break.
//    Conditions: shared B, lenA <= 256, scores in int8, gap <= -1,
//    maxS * min(lenA, lenB) < 2^14.
//
//  sw_pair_kernel<RA> -- per-pair B (reads against reads), lenA <= 256: the same register tiling with the score
//    looked up in the compact int32 table in LDS by the lane's own B symbol.
//
//  sw_generic_kernel -- the last resort (A beyond 4096 symbols, tables too large for LDS):
//    one pair per lane, the reference's own loop nest with the previous row in
//    a lane-interleaved global scratch.  Correct for any input; not tuned.
//
// HBM traffic of the hot kernel is the reads (150 B/pair) + 24 B/pair of
// output: it is VALU bound by construction, the roofline that matters is the
// integer-ALU one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "host_pipeline.h"

#include "sw_scoring.h"

namespace polyhip {
namespace k3 {

constexpr int THREADS = 256;
constexpr int U = 4;          // columns per unrolled block
constexpr int JC_MAX = 1024;  // columns per LDS chunk (10 key bits)
constexpr int SCORE_LIMIT = 1 << 14;
constexpr uint64_t LONG_PACKED_ROWS = 8ull << 20; // pairs x rows from which the packed banded pass pays for reads > 256 (a quarter of the chip's lanes busy)
constexpr uint64_t WAVE_BATCH = 49152; // below this many pairs the one-wave-per-pair kernel (2.5e12 cell updates/s flat) beats the ~15 ms floor of one lane-per-pair wave

static thread_local int g_last_path = 0;
static thread_local int g_last_half = 0;
static thread_local int g_last_lanes = 0; // lanes that share a lane's two read pairs in the last packed score pass (0: not packed)

// prof[j][c] = S(symA[c], b_j) as int8; pad columns / pad code = -128.
// Also finds the first byte of B that is not in SecondAlphabet.
__global__ __launch_bounds__(256) void profile_kernel(const uint8_t *__restrict__ B, uint32_t lenB, uint32_t lenB_pad,
                                                     const int8_t *__restrict__ lutc, int ncodes, int cp,
                                                     const uint8_t *__restrict__ validB, int8_t *__restrict__ prof,
                                                     uint32_t *__restrict__ binfo)
{
    // layout: prof[((j / 4) * cp + code) * 4 + (j % 4)] -- one dword holds the
    // scores of one code against 4 consecutive columns
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lenB_pad)
        return;
    int8_t *col = prof + (size_t)(j >> 2) * cp * 4 + (j & 3);
    if (j < lenB) {
        const uint8_t b = B[j];
        for (int c = 0; c < ncodes; ++c)
            col[c * 4] = lutc[c * 256 + b];
        for (int c = ncodes; c < cp; ++c)
            col[c * 4] = -128;
        if (!validB[b])
            atomicMin(&binfo[0], j);
    } else {
        for (int c = 0; c < cp; ++c)
            col[c * 4] = -128;
    }
}

// one row of the 4-column block (i is a compile-time constant after unrolling)
#define PH_SW_ROW(I, W)                                                              \
    do {                                                                             \
        const int i_ = (I);                                                          \
        const uint32_t w_ = (W);                                                     \
        const int s0 = (int)(int8_t)(w_);                                            \
        const int s1 = (int)(int8_t)(w_ >> 8);                                       \
        const int s2 = (int)(int8_t)(w_ >> 16);                                      \
        const int s3 = (int)w_ >> 24;                                                \
        const int left = H[i_]; /* H[i][jb-1] */                                     \
        const int h0 = max(max(pdiag + s0, max(pr0, left) + gap), 0);                \
        const int h1 = max(max(pr0 + s1, max(pr1, h0) + gap), 0);                    \
        const int h2 = max(max(pr1 + s2, max(pr2, h1) + gap), 0);                    \
        const int h3 = max(max(pr2 + s3, max(pr3, h2) + gap), 0);                    \
        pdiag = left;                                                                \
        pr0 = h0;                                                                    \
        pr1 = h1;                                                                    \
        pr2 = h2;                                                                    \
        pr3 = h3;                                                                    \
        H[i_] = h3;                                                                  \
        const uint32_t ci = (uint32_t)((255 - i_) << 10);                            \
        const uint32_t k0 = ((uint32_t)h0 << 18) | (ci | sj0);                       \
        const uint32_t k1 = ((uint32_t)h1 << 18) | (ci | (sj0 - 1u));                \
        const uint32_t k2 = ((uint32_t)h2 << 18) | (ci | (sj0 - 2u));                \
        const uint32_t k3 = ((uint32_t)h3 << 18) | (ci | (sj0 - 3u));                \
        best = max(max(best, k0), k1);                                               \
        best = max(max(best, k2), k3);                                               \
    } while (0)

template <int RA, int CP>
__global__ __launch_bounds__(THREADS) void sw_shared_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    uint32_t lenB, uint32_t lenB_pad, const int8_t *__restrict__ prof, uint32_t jc_max,
    const uint8_t *__restrict__ codeA, const uint32_t *__restrict__ binfo, int ncodes, int gap,
    int64_t *__restrict__ score, uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err,
    const uint32_t *__restrict__ list, const uint32_t *__restrict__ count)
{
    static_assert(RA % 4 == 0 && RA <= 256, "RA");
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *P = lds;
    uint8_t *codeL = reinterpret_cast<uint8_t *>(lds + (size_t)jc_max * CP);

    const int tid = threadIdx.x;
    // either pairs [0, npairs) in order, or the pairs on `list` (the packed pass's ties, sw_packed.hip)
    uint64_t pair = (uint64_t)blockIdx.x * THREADS + tid;
    bool active = pair < npairs;
    if (list) {
        const uint32_t cnt = *count;
        if ((uint64_t)blockIdx.x * THREADS >= cnt)
            return; // whole workgroup: before any barrier
        active = pair < cnt;
        pair = active ? list[pair] : 0;
    }
    codeL[tid] = codeA[tid];
    __syncthreads();

    // ---- my read -> packed codes (4 per VGPR); rows >= lenA use the pad code
    const uint32_t PAD = (uint32_t)ncodes;
    uint64_t o0 = 0;
    uint32_t lenA = 0;
    if (active) {
        o0 = offA[pair];
        const uint64_t l = offA[pair + 1] - o0;
        lenA = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
    }
    const bool too_long = lenA > RA;
    if (too_long)
        lenA = 0;
    uint32_t apk[RA / 4];
    int firstbad = -1;
    uint32_t badsym = 0, a0sym = 0;
    {
        const uint8_t *ap = A + o0;
#pragma unroll
        for (int w = 0; w < RA / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * w + b;
                uint32_t code = PAD;
                if ((uint32_t)i < lenA) {
                    const uint32_t sym = ap[i];
                    if (i == 0)
                        a0sym = sym;
                    code = codeL[sym];
                    if (code == 0xFF) {
                        if (firstbad < 0) {
                            firstbad = i;
                            badsym = sym;
                        }
                        code = PAD;
                    }
                }
                pk |= (code * 4u) << (8 * b); // byte offset of the code's dword inside a profile block
            }
            apk[w] = pk;
        }
    }

    int H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;
    uint32_t gbest = 0, gbestj = 0; // gbest = h << 8 | (255 - i)

    for (uint32_t c0 = 0; c0 < lenB_pad; c0 += jc_max) {
        const uint32_t jc = min(jc_max, lenB_pad - c0);
        __syncthreads(); // previous chunk fully consumed
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(prof + (size_t)c0 * CP);
            uint4 *dst = reinterpret_cast<uint4 *>(P);
            const uint32_t nvec = jc * CP / 16;
            for (uint32_t v = tid; v < nvec; v += THREADS)
                dst[v] = src[v];
        }
        __syncthreads();

        uint32_t best = 0;
        const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P)); // LDS byte address of P
        for (uint32_t jb = 0; jb < jc; jb += U) {
            // One block of U = 4 columns, swept row by row: the 4 cells of a row
            // chain through `left`, the previous row's 4 values stay in pr[].
            // The profile dwords are fetched by hand-placed ds_read_b32 one
            // 4-row group ahead (hipcc sinks a plain load next to its use and
            // exposes the LDS latency); addresses come from one SDWA byte-add
            // per row on the packed code registers.
            const uint32_t blk = lds_base + (jb >> 2) * (CP * 4);
            const uint32_t sj0 = 1023u - jb; // key column part of column jb (low two bits are 3)
            int pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0; // H[i-1][jb..jb+3]; row 0 of H is 0
            int pdiag = 0;                          // H[i-1][jb-1]
            uint32_t wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3;
            PH_PROF_ISSUE(apk[0], wa0, wa1, wa2, wa3);
#pragma unroll
            for (int g = 0; g < RA / 4; ++g) {
                if (g + 1 < RA / 4) {
                    PH_PROF_ISSUE(apk[g + 1], wb0, wb1, wb2, wb3);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wa0), "+v"(wa1), "+v"(wa2), "+v"(wa3));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa0), "+v"(wa1), "+v"(wa2), "+v"(wa3));
                }
                PH_SW_ROW(4 * g + 0, wa0);
                PH_SW_ROW(4 * g + 1, wa1);
                PH_SW_ROW(4 * g + 2, wa2);
                PH_SW_ROW(4 * g + 3, wa3);
                wa0 = wb0;
                wa1 = wb1;
                wa2 = wb2;
                wa3 = wb3;
            }
        }
        const uint32_t si = best >> 10;
        if (si > gbest) { // strict: an earlier chunk (smaller j) wins ties on (h, i)
            gbest = si;
            gbestj = c0 + (1023u - (best & 1023u));
        }
    }

    if (!active)
        return;
    uint32_t e = 0;
    if (too_long) {
        e = 0xFFFFFFFFu;
    } else if (lenA > 0 && lenB > 0) {
        // align.go:189-191 + matrix.go:29-36: row-major first failing cell
        const uint32_t bbad = binfo[0];
        if (firstbad == 0)
            e = (1u << 8) | a0sym;
        else if (bbad != 0xFFFFFFFFu)
            e = (2u << 8) | B[bbad];
        else if (firstbad > 0)
            e = (1u << 8) | badsym;
    }
    const uint32_t sc = gbest >> 8;
    const bool hit = e == 0 && sc > 0;
    score[pair] = hit ? (int64_t)sc : 0;
    endA[pair] = hit ? (255u - (gbest & 255u)) + 1u : 0u;
    endB[pair] = hit ? gbestj + 1u : 0u;
    err[pair] = e;
}

// ---- per-pair B (reads against reads), register-tiled --------------------------------------------------
// One pair per lane, H column (RA rows) in VGPRs like sw_shared_kernel, but every lane walks its OWN
// B: the score of a cell is a lookup in the compact int32 table T[codeA][codeB] in LDS (row offset
// from the packed row registers, column offset from the lane's current B symbol).  Same key trick for
// the row-major-first argmax, columns folded in chunks of 1024.
template <int RA>
__global__ __launch_bounds__(THREADS) void sw_pair_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    const uint64_t *__restrict__ offB, uint32_t max_lenB, const uint8_t *__restrict__ codeA,
    const uint8_t *__restrict__ codeB, const int32_t *__restrict__ lutcc, int na, int nb, int gap,
    int64_t *__restrict__ score, uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) int32_t Tp[]; // [na][nb] then codeA[256], codeB[256]
    uint8_t *cA = reinterpret_cast<uint8_t *>(Tp + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x;
    for (int t = tid; t < na * nb; t += THREADS)
        Tp[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();

    const uint64_t pair = (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < npairs;
    uint32_t m = 0, n = 0;
    const uint8_t *a = A, *b = B;
    bool too_long = false;
    if (active) {
        const uint64_t o0 = offA[pair], l = offA[pair + 1] - o0;
        too_long = l > (uint64_t)RA;
        m = too_long ? 0u : (uint32_t)l;
        a = A + o0;
        b = B + offB[pair];
        n = (uint32_t)(offB[pair + 1] - offB[pair]);
    }
    // align.go:189-191 + matrix.go:29-36: the first failing Score() in row-major order
    uint32_t e = too_long ? 0xFFFFFFFFu : 0u;
    if (m > 0 && n > 0) {
        if (cA[a[0]] == 0xFFu) {
            e = (1u << 8) | a[0];
        } else {
            for (uint32_t j = 0; j < n && !e; ++j)
                if (cB[b[j]] == 0xFFu)
                    e = (2u << 8) | b[j];
            for (uint32_t i = 1; i < m && !e; ++i)
                if (cA[a[i]] == 0xFFu)
                    e = (1u << 8) | a[i];
        }
    }
    const bool work = active && e == 0u && m > 0 && n > 0;

    uint32_t aoff[RA / 2]; // row offsets into T (code * nb), two per register; rows >= m use the pad row (zeros)
#pragma unroll
    for (int r = 0; r < RA / 2; ++r) {
        uint32_t pk = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * r + h;
            uint32_t code = (uint32_t)(na - 1);
            if (work && (uint32_t)i < m)
                code = cA[a[i]];
            pk |= (code * (uint32_t)nb) << (16 * h);
        }
        aoff[r] = pk;
    }
    int H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;
    uint32_t gbest = 0, gbestj = 0; // gbest = h << 8 | (255 - i)
    const uint32_t ncol = work ? n : 0u;
    for (uint32_t c0 = 0; c0 < max_lenB; c0 += JC_MAX) {
        if (!__any(c0 < ncol))
            break;
        uint32_t best = 0;
        const uint32_t jc = min((uint32_t)JC_MAX, max_lenB - c0);
        for (uint32_t jr = 0; jr < jc; ++jr) {
            if (!__any(c0 + jr < ncol))
                break;
            if (c0 + jr < ncol) {
                const uint32_t cb = cB[b[c0 + jr]];
                const uint32_t sj = 1023u - jr;
                int diag = 0, up = 0;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const uint32_t ro = (aoff[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                    const int sc = Tp[ro + cb];
                    const int left = H[i];
                    const int h = max(max(diag + sc, max(up, left) + gap), 0);
                    // rows >= m sit on the pad row of T (zeros): whatever they hold is below the row above
                    // them by at least |gap| ... unless gap >= 0, so they are kept out of the argmax by value 0
                    const uint32_t hv = (uint32_t)i < m ? (uint32_t)h : 0u;
                    best = max(best, (hv << 18) | ((uint32_t)((255 - i) << 10) | sj));
                    diag = left;
                    up = h;
                    H[i] = h;
                }
            }
        }
        const uint32_t si = best >> 10;
        if (si > gbest) { // strict: an earlier chunk (smaller j) wins ties on (h, i)
            gbest = si;
            gbestj = c0 + (1023u - (best & 1023u));
        }
    }
    if (!active)
        return;
    const uint32_t sc = gbest >> 8;
    const bool hit = e == 0u && sc > 0u;
    score[pair] = hit ? (int64_t)sc : 0;
    endA[pair] = hit ? (255u - (gbest & 255u)) + 1u : 0u;
    endB[pair] = hit ? gbestj + 1u : 0u;
    err[pair] = e;
}

// The reference's loop nest, one pair per lane; previous row in global scratch
// laid out [j][pair] so a wave's accesses coalesce.
__global__ __launch_bounds__(256) void sw_generic_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    const uint64_t *__restrict__ offB, uint64_t lenB_shared, const int32_t *__restrict__ lut,
    const uint8_t *__restrict__ validA, const uint8_t *__restrict__ validB, int gap, int32_t *__restrict__ work,
    int64_t *__restrict__ score, uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err)
{
    const uint64_t pair = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= npairs)
        return;
    const uint8_t *a = A + offA[pair];
    const uint64_t m = offA[pair + 1] - offA[pair];
    const uint8_t *b = offB ? B + offB[pair] : B;
    const uint64_t n = offB ? offB[pair + 1] - offB[pair] : lenB_shared;
    uint32_t e = 0;
    if (m > 0 && n > 0) {
        if (!validA[a[0]]) {
            e = (1u << 8) | a[0];
        } else {
            for (uint64_t j = 0; j < n && !e; ++j)
                if (!validB[b[j]])
                    e = (2u << 8) | b[j];
            for (uint64_t i = 1; i < m && !e; ++i)
                if (!validA[a[i]])
                    e = (1u << 8) | a[i];
        }
    }
    int32_t best = 0;
    uint64_t bi = 0, bj = 0;
    if (!e && m > 0 && n > 0) {
        int32_t *Hrow = work + pair;
        const uint64_t stride = npairs;
        for (uint64_t j = 0; j <= n; ++j)
            Hrow[j * stride] = 0;
        for (uint64_t i = 1; i <= m; ++i) {
            const int32_t *lrow = lut + (uint32_t)a[i - 1] * 256;
            int32_t diag = 0, left = 0;
            for (uint64_t j = 1; j <= n; ++j) {
                const int32_t up = Hrow[j * stride];
                const int32_t s = lrow[b[j - 1]];
                int32_t h = max(diag + s, max(up + gap, left + gap));
                h = max(h, 0);
                Hrow[j * stride] = h;
                diag = up;
                left = h;
                if (h > best) {
                    best = h;
                    bi = i;
                    bj = j;
                }
            }
        }
    }
    score[pair] = best;
    endA[pair] = (uint32_t)bi;
    endB[pair] = (uint32_t)bj;
    err[pair] = e;
}

struct Plan {
    int path;      // 1 fast, 2 generic, 3 packed (sw_packed.hip) + wave kernel for its ties, 4 wave kernel (small batch),
                   // 5 per-pair B register-tiled, 6 wave kernel for what the others cannot take (long reads, ...),
                   // 7 long reads, shared B: packed banded pass (maximum + its block) + wave kernel in locate mode
    int ra, cp;    // fast: template parameters
    uint32_t lenB_pad, jc_max;
    size_t work_bytes, smem_bytes;
    size_t fast_bytes; // path 3: where the packed pass's workspace starts
    k3p::PackedPlan pk;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static bool wave_kernel_off()
{
    return env_is("POLYHIP_SW_WAVE", '0'); // testing aid: no one-wave-per-pair kernel
}

static bool pair_kernel_off()
{
    return env_is("POLYHIP_SW_PAIR", '0'); // testing aid: generic kernel for per-pair B
}

static Plan plan(const polyhip_scoring *sc, uint64_t npairs, uint32_t max_lenA, uint64_t lenB, bool shared)
{
    Plan p{};
    const uint64_t minlen = std::min<uint64_t>(max_lenA, lenB);
    const bool wave_possible = max_lenA > 0 && max_lenA <= k3w::WAVE_MAX_LENA && lenB > 0 && lenB < (1ull << 31) - 64 &&
                               (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) * 4 + 512 <= 60 * 1024 && !wave_kernel_off();
    const bool fast = shared && sc->int8_ok && sc->gap <= -1 && max_lenA <= 256 && lenB < (1ull << 31) &&
                      sc->cp <= 32 && (uint64_t)std::max(sc->smax, 0) * minlen < (uint64_t)SCORE_LIMIT;
    if (fast) {
        p.path = 1;
        p.ra = max_lenA <= 64 ? 64 : max_lenA <= 152 ? 152 : 256;
        p.cp = sc->cp <= 8 ? 8 : 32;
        p.lenB_pad = (uint32_t)align_up(lenB, U);
        p.jc_max = JC_MAX;
        if (p.jc_max > p.lenB_pad)
            p.jc_max = std::max<uint32_t>(p.lenB_pad, U);
        p.smem_bytes = (size_t)p.jc_max * p.cp + 256;
        p.work_bytes = 256 + align_up((size_t)p.lenB_pad * p.cp, 256);
        p.fast_bytes = p.work_bytes;
        const bool wave_ok = (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) * 4 + 512 <= 60 * 1024;
        if (wave_ok && npairs < WAVE_BATCH && !wave_kernel_off()) {
            p.path = 4; // too few pairs to fill the chip one per lane: one wave per pair (sw_wave.hip)
        } else if (wave_ok && k3p::packed_plan(sc, npairs, max_lenA, lenB, &p.pk) && p.pk.ra == p.ra && p.cp == 8) {
            p.path = 3;
            p.work_bytes += p.pk.work_bytes;
        }
    } else if (!shared && (max_lenA <= 64 || (max_lenA <= 256 && !wave_possible)) && max_lenA > 0 && lenB > 0 &&
               lenB < (1ull << 31) && // beyond 64 rows the wave kernel is faster (200k pairs of 150 x 150: 2.7 vs 3.9 ms)
               (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) * 4 + 512 <= 60 * 1024 &&
               (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) < 65536 &&
               (uint64_t)std::max(sc->smax, 0) * minlen < (uint64_t)SCORE_LIMIT && !pair_kernel_off()) {
        p.path = 5; // per-pair B, register-tiled (sw_pair_kernel)
        p.ra = max_lenA <= 64 ? 64 : max_lenA <= 152 ? 152 : 256;
        p.work_bytes = 256;
    } else if (shared && max_lenA > 256 && lenB < (1ull << 31) - 64 &&
               (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) * 4 + 512 <= 60 * 1024 && !wave_kernel_off() &&
               k3p::packed_plan(sc, npairs, max_lenA, lenB, &p.pk) && npairs * (uint64_t)p.pk.ra >= LONG_PACKED_ROWS) {
        // long reads against one reference, enough of them to fill the chip K lanes per pair: the packed banded
        // pass finds each pair's maximum and the block it sits in, the wave kernel then sweeps only the columns
        // that can reach it (a third of the reference for a read that aligns well)
        p.path = 7;
        p.fast_bytes = 256;
        p.work_bytes = 256 + p.pk.work_bytes;
    } else if (max_lenA > 0 && max_lenA <= k3w::WAVE_MAX_LENA && lenB > 0 && lenB < (1ull << 31) - 64 &&
               (size_t)(sc->ncodes + 1) * (sc->ncodesB + 1) * 4 + 512 <= 60 * 1024 && !wave_kernel_off()) {
        // whatever the lane-per-pair kernels cannot take (reads longer than 256, gap >= 0, wide scores), shared or
        // per-pair B: one wave per pair, plain int32, up to 4096 rows (sw_wave.hip)
        p.path = 6;
        p.work_bytes = 256;
    } else {
        p.path = 2;
        p.work_bytes = align_up((size_t)npairs * (lenB + 1) * sizeof(int32_t), 256);
    }
    return p;
}

template <int RA, int CP>
static int launch_fast(const polyhip_scoring *sc, const Plan &p, const uint8_t *d_A, const uint64_t *d_offA,
                       uint64_t npairs, const uint8_t *d_B, uint32_t lenB, int8_t *prof, uint32_t *binfo,
                       int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, hipStream_t st,
                       const uint32_t *list = nullptr, const uint32_t *count = nullptr)
{
    auto kern = sw_shared_kernel<RA, CP>;
    if (p.smem_bytes > 48 * 1024)
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)p.smem_bytes));
    const uint64_t blocks = (npairs + THREADS - 1) / THREADS;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), p.smem_bytes, st, d_A, d_offA, npairs, d_B, lenB,
                       p.lenB_pad, prof, p.jc_max, sc->d_codeA, binfo, sc->ncodes, (int)sc->gap, d_score, d_endA,
                       d_endB, d_err, list, count);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

} // namespace k3
} // namespace polyhip

using namespace polyhip;

extern "C" {

int polyhip_scoring_create(const int32_t *lut, const uint8_t *validA, const uint8_t *validB, int64_t gap,
                           polyhip_scoring **out)
{
    PH_REQUIRE(lut && validA && validB && out, "polyhip_scoring_create: null pointer");
    polyhip_scoring *sc = new polyhip_scoring();
    sc->gap = gap;
    memcpy(sc->lut, lut, sizeof sc->lut);
    for (int i = 0; i < 256; ++i) {
        sc->validA[i] = validA[i] && i < 0x80;
        sc->validB[i] = validB[i] && i < 0x80;
    }
    sc->ncodes = 0;
    memset(sc->codeA, 0xFF, 256);
    uint8_t symA[128];
    for (int a = 0; a < 128; ++a)
        if (sc->validA[a]) {
            symA[sc->ncodes] = (uint8_t)a;
            sc->codeA[a] = (uint8_t)sc->ncodes++;
        }
    sc->ncodesB = 0;
    memset(sc->codeB, 0xFF, 256);
    uint8_t symB[128];
    for (int b = 0; b < 128; ++b)
        if (sc->validB[b]) {
            symB[sc->ncodesB] = (uint8_t)b;
            sc->codeB[b] = (uint8_t)sc->ncodesB++;
        }
    sc->cp = (sc->ncodes + 1 + 3) & ~3;
    sc->smin = 0;
    sc->smax = 0;
    bool any = false;
    for (int a = 0; a < 128; ++a)
        for (int b = 0; b < 128; ++b)
            if (sc->validA[a] && sc->validB[b]) {
                const int32_t v = lut[a * 256 + b];
                sc->smin = any ? std::min(sc->smin, v) : v;
                sc->smax = any ? std::max(sc->smax, v) : v;
                any = true;
            }
    const int64_t ag = gap < 0 ? -gap : gap;
    const int64_t am = std::max<int64_t>(std::max<int64_t>(std::llabs((long long)sc->smin), std::llabs((long long)sc->smax)), ag);
    if (am > (1 << 24)) {
        delete sc;
        return set_error(POLYHIP_ERR_UNSUPPORTED, "polyhip_scoring_create: |score| or |gap| %lld exceeds 2^24",
                         (long long)am);
    }
    sc->absmax = (int32_t)am;
    sc->int8_ok = sc->smin >= -127 && sc->smax <= 127; // -128 is the pad marker
    sc->d_lutc = nullptr;
    sc->d_codeB = nullptr;
    sc->d_lutcc = nullptr;
    sc->d_codeA = nullptr;
    sc->d_lut = nullptr;
    sc->d_validA = sc->d_validB = nullptr;
    sc->rep_m = new std::mutex();
    sc->rep = new std::vector<polyhip_scoring *>();

    auto fail = [&](hipError_t e, const char *what) {
        polyhip_scoring_destroy(sc);
        return set_error(POLYHIP_ERR_HIP, "polyhip_scoring_create: %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipGetDevice(&sc->device)) != hipSuccess)
        return fail(e, "hipGetDevice");
    if ((e = hipMalloc(&sc->d_lut, sizeof sc->lut)) != hipSuccess)
        return fail(e, "hipMalloc");
    if ((e = hipMalloc(&sc->d_codeA, 256)) != hipSuccess)
        return fail(e, "hipMalloc");
    if ((e = hipMalloc(&sc->d_validA, 256)) != hipSuccess)
        return fail(e, "hipMalloc");
    if ((e = hipMalloc(&sc->d_validB, 256)) != hipSuccess)
        return fail(e, "hipMalloc");
    if ((e = hipMemcpy(sc->d_lut, sc->lut, sizeof sc->lut, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(sc->d_codeA, sc->codeA, 256, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(sc->d_validA, sc->validA, 256, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(sc->d_validB, sc->validB, 256, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy");
    {
        const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
        std::vector<int32_t> cc((size_t)na * nb, 0);
        for (int a = 0; a < sc->ncodes; ++a)
            for (int b = 0; b < sc->ncodesB; ++b)
                cc[(size_t)a * nb + b] = lut[symA[a] * 256 + symB[b]];
        if ((e = hipMalloc(&sc->d_codeB, 256)) != hipSuccess)
            return fail(e, "hipMalloc");
        if ((e = hipMemcpy(sc->d_codeB, sc->codeB, 256, hipMemcpyHostToDevice)) != hipSuccess)
            return fail(e, "hipMemcpy");
        if ((e = hipMalloc(&sc->d_lutcc, cc.size() * 4)) != hipSuccess)
            return fail(e, "hipMalloc");
        if ((e = hipMemcpy(sc->d_lutcc, cc.data(), cc.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
            return fail(e, "hipMemcpy");
    }
    if (sc->int8_ok && sc->ncodes > 0) {
        std::vector<int8_t> lutc((size_t)sc->ncodes * 256);
        for (int c = 0; c < sc->ncodes; ++c)
            for (int b = 0; b < 256; ++b)
                lutc[(size_t)c * 256 + b] = sc->validB[b] ? (int8_t)lut[symA[c] * 256 + b] : (int8_t)0;
        if ((e = hipMalloc(&sc->d_lutc, lutc.size())) != hipSuccess)
            return fail(e, "hipMalloc");
        if ((e = hipMemcpy(sc->d_lutc, lutc.data(), lutc.size(), hipMemcpyHostToDevice)) != hipSuccess)
            return fail(e, "hipMemcpy");
    }
    *out = sc;
    return POLYHIP_OK;
}

int polyhip_scoring_destroy(polyhip_scoring *sc)
{
    if (!sc)
        return POLYHIP_OK;
    (void)hipFree(sc->d_lutc);
    (void)hipFree(sc->d_codeB);
    (void)hipFree(sc->d_lutcc);
    (void)hipFree(sc->d_codeA);
    (void)hipFree(sc->d_lut);
    (void)hipFree(sc->d_validA);
    (void)hipFree(sc->d_validB);
    if (sc->rep)
        for (polyhip_scoring *r : *sc->rep)
            polyhip_scoring_destroy(r);
    delete sc->rep;
    delete sc->rep_m;
    delete sc;
    return POLYHIP_OK;
}

size_t polyhip_sw_workspace_bytes(const polyhip_scoring *sc, uint64_t npairs, uint32_t max_lenA, uint64_t lenB,
                                  int shared_B)
{
    if (!sc)
        return 0;
    return k3::plan(sc, npairs, max_lenA, lenB, shared_B != 0).work_bytes;
}

} // extern "C"
void polyhip::k3::score_choice(int *path, int *half, bool set)
{
    if (set) {
        g_last_path = *path;
        g_last_half = *half;
    } else {
        *path = g_last_path;
        *half = g_last_half;
    }
}
extern "C" {
int polyhip_sw_last_path(void) { return k3::g_last_path; }
int polyhip_sw_last_packed_half(void) { return k3::g_last_half; }
int polyhip_sw_last_packed_lanes(void) { return k3::g_last_lanes; }

} // extern "C"

// the score pass; `defer` != 0 asks the packed paths (3: reads of at most 256 rows; 7: 257..1024 rows with the byte-profile
// locate kernel) to leave the end cell of a pair to the traceback kernel (k3p::SW_END_DEFERRED): *deferred says whether it did
int polyhip::k3::score_pass(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                            uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB, int64_t *d_score,
                            uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, void *d_work, size_t work_bytes,
                            polyhip_stream_t stream, int defer, int *deferred)
{
    if (deferred)
        *deferred = 0;
    PH_REQUIRE(sc, "polyhip_sw_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_offA && d_score && d_endA && d_endB && d_err, "polyhip_sw_batch: null pointer");
    PH_REQUIRE(npairs < (1ull << 31) * 256, "polyhip_sw_batch: too many pairs");
    if ((int64_t)sc->absmax * (int64_t)((uint64_t)max_lenA + lenB) >= (1ll << 31))
        return set_error(POLYHIP_ERR_UNSUPPORTED, "polyhip_sw_batch: scores could overflow int32 (|s|max %d, lengths %u+%llu)",
                         sc->absmax, max_lenA, (unsigned long long)lenB);
    const bool shared = d_offB == nullptr;
    const k3::Plan p = k3::plan(sc, npairs, max_lenA, lenB, shared);
    PH_REQUIRE(d_work && work_bytes >= p.work_bytes, "polyhip_sw_batch: workspace too small (%zu < %zu)", work_bytes,
               p.work_bytes);
    hipStream_t st = as_stream(stream);
    k3::g_last_path = p.path;
    k3::g_last_half = (p.path == 3 || p.path == 7) && p.pk.f16 ? 1 : 0;
    k3::g_last_lanes = (p.path == 3 || p.path == 7) ? (p.pk.k > 1 ? p.pk.k : (p.pk.x2_rb ? 2 : 1)) : 0;
    if (p.path == 1 || p.path == 3 || p.path == 4) {
        uint32_t *binfo = static_cast<uint32_t *>(d_work);
        int8_t *prof = static_cast<int8_t *>(d_work) + 256;
        PH_HIP(hipMemsetAsync(binfo, 0xFF, 256, st));
        if (p.lenB_pad > 0) {
            hipLaunchKernelGGL(k3::profile_kernel, dim3((p.lenB_pad + 255) / 256), dim3(256), 0, st, d_B,
                               (uint32_t)lenB, p.lenB_pad, sc->d_lutc, sc->ncodes, p.cp, sc->d_validB, prof, binfo);
            PH_HIP(hipGetLastError());
        }
        if (p.path == 4)
            return k3w::wave_run(sc, d_A, d_offA, npairs, max_lenA, d_B, nullptr, (uint32_t)lenB, binfo, nullptr, nullptr,
                                 npairs, d_score, d_endA, d_endB, d_err, st);
        // packed pass first (two pairs per lane); the few pairs it leaves on its tie list go through the
        // exact one-wave-per-pair kernel (a lane-per-pair kernel would take a full DP's time for them)
        if (p.path == 3) {
            const int do_defer = defer && p.pk.ra <= 256 ? 1 : 0;
            if (deferred)
                *deferred = do_defer;
            // Sub-batches of 262,144 pairs (one full round of the packed kernel), alternating between the caller's stream
            // and the library's second one, each with its own slice of the packed pass's workspace: the packed kernel's
            // last, partly filled round and the short locate / tie kernels of one sub-batch run beside the packed kernel of
            // the next.  POLYHIP_SW_OVERLAP=0: the whole batch in one piece (testing aid).
            const uint64_t SUB = 262144;
            k3p::PackedPlan ps{};
            const bool split = npairs >= 2 * SUB && !env_is("POLYHIP_SW_OVERLAP", '0') &&
                               k3p::packed_plan(sc, SUB, max_lenA, lenB, &ps) && ps.ra == p.pk.ra && ps.k == p.pk.k &&
                               2 * (ps.work_bytes + 256) <= p.pk.work_bytes;
            if (!split) {
                uint32_t *list = nullptr, *count = nullptr;
                const uint32_t *infoM = nullptr, *infoQ = nullptr;
                const int rc = k3p::packed_run(sc, p.pk, d_A, d_offA, npairs, d_B, (uint32_t)lenB, prof, binfo,
                                               static_cast<uint8_t *>(d_work) + p.fast_bytes, d_score, d_endA, d_endB, d_err,
                                               &list, &count, st, &infoM, &infoQ, do_defer);
                if (rc != POLYHIP_OK)
                    return rc;
                // (round 6: the tie list with the packed pass's M / first block / span -- a near tie is swept on its window,
                // not over the whole reference: sw_wave_kernel's locate mode)
                return k3w::wave_run(sc, d_A, d_offA, npairs, max_lenA, d_B, nullptr, (uint32_t)lenB, binfo, list, count,
                                     npairs, d_score, d_endA, d_endB, d_err, st, infoM, infoQ);
            }
            ps.skip_rows = p.pk.skip_rows; // (same kernels as the whole batch would take)
            ps.x2_rb = p.pk.x2_rb;
            AuxStream &aux = aux_stream(st);
            const size_t slice = (ps.work_bytes + 255) & ~(size_t)255;
            // both slices' tables once, in front of the fork (a tiny kernel queued beside a full-chip one waits for it)
            for (int q = 0; q < 2; ++q)
                if (int rc = k3p::packed_profiles(sc, ps, d_B, (uint32_t)lenB, static_cast<uint8_t *>(d_work) + p.fast_bytes + q * slice, st))
                    return rc;
            ps.reuse_profiles = true;
            PH_HIP(aux.fork(st)); // the byte profile and the tables are ready
            uint64_t k = 0;
            for (uint64_t i0 = 0; i0 < npairs; i0 += SUB, ++k) {
                const uint64_t m = std::min(SUB, npairs - i0);
                hipStream_t sk = (k & 1) ? aux.s : st;
                uint8_t *wk = static_cast<uint8_t *>(d_work) + p.fast_bytes + (k & 1) * slice;
                uint32_t *list = nullptr, *count = nullptr;
                const uint32_t *infoM = nullptr, *infoQ = nullptr;
                int rc = k3p::packed_run(sc, ps, d_A, d_offA + i0, m, d_B, (uint32_t)lenB, prof, binfo, wk, d_score + i0,
                                         d_endA + i0, d_endB + i0, d_err + i0, &list, &count, sk, &infoM, &infoQ, do_defer);
                if (rc == POLYHIP_OK)
                    rc = k3w::wave_run(sc, d_A, d_offA + i0, m, max_lenA, d_B, nullptr, (uint32_t)lenB, binfo, list, count, m,
                                       d_score + i0, d_endA + i0, d_endB + i0, d_err + i0, sk, infoM, infoQ);
                if (rc != POLYHIP_OK) {
                    (void)aux.join(st);
                    return rc;
                }
            }
            PH_HIP(aux.join(st));
            return POLYHIP_OK;
        }
#define PH_SW_CASE(RA_, CP_)                                                                                       \
    if (p.ra == RA_ && p.cp == CP_)                                                                                \
        return k3::launch_fast<RA_, CP_>(sc, p, d_A, d_offA, npairs, d_B, (uint32_t)lenB, prof, binfo, d_score,     \
                                         d_endA, d_endB, d_err, st);
#ifndef PH_SW_FAST_LIST
#define PH_SW_FAST_LIST(X) X(64, 8) X(152, 8) X(256, 8) X(64, 32) X(152, 32) X(256, 32)
#endif
        PH_SW_FAST_LIST(PH_SW_CASE)
#undef PH_SW_CASE
        return set_error(POLYHIP_ERR_UNSUPPORTED, "polyhip_sw_batch: no kernel for RA=%d CP=%d", p.ra, p.cp);
    }
    if (p.path == 7) {
        uint32_t *list = nullptr, *count = nullptr;
        const uint32_t *infoM = nullptr, *infoQ = nullptr;
        const int rc = k3p::packed_run(sc, p.pk, d_A, d_offA, npairs, d_B, (uint32_t)lenB, nullptr, nullptr,
                                       static_cast<uint8_t *>(d_work) + p.fast_bytes, d_score, d_endA, d_endB, d_err, &list,
                                       &count, st, &infoM, &infoQ);
        if (rc != POLYHIP_OK)
            return rc;
        // (one call for score + strings, 257..1024 rows: the traceback kernel finds the end cell of a maximum that sits in
        // one block during its own sweep; the locate step then only takes the ties)
        const int do_defer = defer && k3w::wave8_ok(sc, max_lenA) ? 1 : 0;
        if (deferred)
            *deferred = do_defer;
        return k3w::wave_run(sc, d_A, d_offA, npairs, max_lenA, d_B, nullptr, (uint32_t)lenB, nullptr, nullptr, nullptr,
                             npairs, d_score, d_endA, d_endB, d_err, st, infoM, infoQ, do_defer);
    }
    if (p.path == 6)
        return k3w::wave_run(sc, d_A, d_offA, npairs, max_lenA, d_B, d_offB, (uint32_t)lenB, nullptr, nullptr, nullptr,
                             npairs, d_score, d_endA, d_endB, d_err, st);
    if (p.path == 5) {
        const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
        const size_t smem = (size_t)na * nb * 4 + 512;
        const uint64_t nblk = (npairs + k3::THREADS - 1) / k3::THREADS;
#define PH_SW_PAIR(RA_)                                                                                               \
    do {                                                                                                              \
        auto kern = k3::sw_pair_kernel<RA_>;                                                                          \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)smem));                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(k3::THREADS), smem, st, d_A, d_offA, npairs, d_B, d_offB,  \
                           (uint32_t)lenB, sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, d_score,      \
                           d_endA, d_endB, d_err);                                                                    \
    } while (0)
        if (p.ra == 64)
            PH_SW_PAIR(64);
        else if (p.ra == 152)
            PH_SW_PAIR(152);
        else
            PH_SW_PAIR(256);
#undef PH_SW_PAIR
        PH_HIP(hipGetLastError());
        return POLYHIP_OK;
    }
    const uint64_t blocks = (npairs + 255) / 256;
    hipLaunchKernelGGL(k3::sw_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_A, d_offA, npairs, d_B, d_offB,
                       lenB, sc->d_lut, sc->d_validA, sc->d_validB, (int)sc->gap, static_cast<int32_t *>(d_work),
                       d_score, d_endA, d_endB, d_err);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

const polyhip_scoring *polyhip::scoring_here(const polyhip_scoring *sc)
{
    int dev = -1;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error(POLYHIP_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (dev == sc->device)
        return sc;
    std::lock_guard<std::mutex> lk(*sc->rep_m);
    for (const polyhip_scoring *r : *sc->rep)
        if (r->device == dev)
            return r;
    polyhip_scoring *r = nullptr; // the tables are rebuilt from the handle's host copies of the creation arguments
    if (polyhip_scoring_create(sc->lut, sc->validA, sc->validB, sc->gap, &r) != POLYHIP_OK)
        return nullptr;
    sc->rep->push_back(r);
    return r;
}

extern "C" {

int polyhip_sw_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                         uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB,
                         int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, void *d_work,
                         size_t work_bytes, polyhip_stream_t stream)
{
    return polyhip::k3::score_pass(sc, d_A, d_offA, npairs, max_lenA, d_B, d_offB, lenB, d_score, d_endA, d_endB, d_err, d_work,
                                   work_bytes, stream, 0, nullptr);
}

// the single-device body: the calling thread's current device (a fan-out worker's, or the caller's own)
static int sw_batch_one(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs, const uint8_t *B,
                        const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA, uint32_t *endB, uint32_t *err)
{
    PH_REQUIRE(sc, "polyhip_sw_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(offA && score && endA && endB && err, "polyhip_sw_batch: null pointer");
    if (!(sc = scoring_here(sc)))
        return POLYHIP_ERR_HIP;
    HostStreams &hs = host_streams(); // the calling thread's own stream, not the null stream
    PH_HIP(hs.init());
    hipStream_t st = hs.s[0];
    PairStage in;
    if (int rc0 = in.load("polyhip_sw_batch", A, offA, npairs, B, offB, lenB, st)) {
        (void)hipStreamSynchronize(st);
        return rc0;
    }
    const uint64_t maxA = in.maxA, maxB = in.maxB;
    DevBuf dscore, dea, deb, derr, dwork;
    PH_HIP(dscore.alloc(npairs * 8));
    PH_HIP(dea.alloc(npairs * 4));
    PH_HIP(deb.alloc(npairs * 4));
    PH_HIP(derr.alloc(npairs * 4));
    const size_t wb = polyhip_sw_workspace_bytes(sc, npairs, (uint32_t)maxA, maxB, offB == nullptr);
    PH_HIP(dwork.alloc(wb));
    int rc = polyhip_sw_batch_dev(sc, in.A(), in.offA(), npairs, (uint32_t)maxA, in.B(), in.offB(), maxB,
                                  dscore.as<int64_t>(), dea.as<uint32_t>(), deb.as<uint32_t>(), derr.as<uint32_t>(), dwork.p,
                                  wb, st);
    if (rc != POLYHIP_OK) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    PH_HIP(hipMemcpyAsync(score, dscore.p, npairs * 8, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(endA, dea.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(endB, deb.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(err, derr.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipStreamSynchronize(st));
    return POLYHIP_OK;
}

int polyhip_sw_batch(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                     const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA,
                     uint32_t *endB, uint32_t *err)
{
    std::shared_ptr<md::Pool> P = npairs && sc ? md::pool() : nullptr;
    if (!P)
        return sw_batch_one(sc, A, offA, npairs, B, offB, lenB, score, endA, endB, err);
    // pairs are independent -- the reads split by bytes, the shared reference goes to every device
    PH_REQUIRE(offA && score && endA && endB && err, "polyhip_sw_batch: null pointer");
    const std::vector<uint64_t> cut = split_pairs(*P, offA, offB, npairs, 24);
    size_t first = 0;
    while (first + 1 < md::size(*P) && cut[first + 1] == cut[first])
        ++first;
    KernelChoice kc;
    const int rc = md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, 0);
        const int r = sw_batch_one(sc, A, offA + i0, m, B, offB ? offB + i0 : nullptr, lenB, score + i0, endA + i0, endB + i0, err + i0);
        if (q == first)
            kc = kernel_choice_get();
        return r;
    });
    kernel_choice_set(kc);
    return rc;
}

} // extern "C"
