// sw_traceback.hip -- K3b: the traceback half of align.SmithWaterman for gfx950.
//
// Replaces search/align/align.go:205-229: starting from the score pass's argmax
// (endA, endB) walk back while H > 0, preferring the diagonal, then "up" (gap in
// B, alignB gets '-'), then "left" (gap in A), and return the two aligned strings.
//
// The reference keeps the whole (m+1) x (n+1) int matrix (6 MB for 150 x 5000).
// Here the score pass (sw_batch.hip) keeps nothing; the traceback re-runs the
// recurrence on a WINDOW of columns that ends at endB and stores 2 bits per
// cell (0 = H is 0, 1 = diag, 2 = up, 3 = left -- the first equality the
// reference's if-chain would hit), then walks the bits.
//
// Why a window is exact (gap < 0): a cell with H > 0 is the end of a path whose
// score is positive; it has at most lenA diagonal/up moves and, because every
// left move costs |gap| out of at most smax per diagonal move, fewer than
// smax * lenA / |gap| left moves.  So no positive path spans more than
//     W = lenA + floor(smax * lenA / |gap|)
// columns: H[.][j] computed from a zero boundary at column j - W - 1 equals the
// true H[.][j].  The walk itself is such a path (<= W columns back from endB) and
// reads neighbours one column further, so a DP that starts at endB - 2W - 1 has
// exact values everywhere the walk looks.  With gap >= 0 (or no positive score)
// the window is the whole of B.  BASELINE config 4: W = 150 + 5*150/2 = 525, the
// window was 1052 of the 5000 columns.  (Round 4: the end row need not be counted
// twice -- see pair_window -- so the batch-wide window is lenA + 2 * smax*lenA/|gap|
// + 2 = 902 columns, and a read that aligns well takes ~205.)
//
// One pair per lane, H column in registers (RA rows) like the score pass; the
// direction words are written lane-interleaved (a wave stores 256 contiguous
// bytes per word) into the caller's workspace, which bounds how many pairs one
// launch covers (~42 KB per pair at config 4; the entry point loops over chunks).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "host_pipeline.h"
#include "sw_scoring.h"

namespace polyhip {
namespace k3t {

constexpr int THREADS = 256;

static thread_local int g_tb_last_path = 0;
static thread_local int g_tb_last_half = 0;
static thread_local int g_nw_last_path = 0;

struct Window {
    uint32_t wcols;   // columns of the re-run DP (<= lenB)
    uint32_t stride;  // bytes per aligned string slot
};

static Window window(const polyhip_scoring *sc, uint32_t max_lenA, uint64_t lenB)
{
    Window w;
    const uint64_t full = lenB;
    uint64_t W = (uint64_t)max_lenA + lenB; // unbounded
    if (sc->gap < 0 && sc->smax > 0)
        W = (uint64_t)max_lenA + ((uint64_t)sc->smax * max_lenA) / (uint64_t)(-sc->gap);
    if (sc->smax <= 0 && sc->gap <= 0)
        W = 0; // every H is 0: nothing to trace (with a POSITIVE gap score gap moves alone make positive cells: round-4 sweep)
    // the batch-wide window (no score known: M >= 1): pair_window's bound with lw, over <= smax * lenA / |gap| -- the end row counted
    // once, as there (round 4: it was 2 W + 2)
    uint64_t wide = 2 * W + 2;
    if (sc->gap < 0 && sc->smax > 0)
        wide = (uint64_t)max_lenA + 2 * (((uint64_t)sc->smax * max_lenA) / (uint64_t)(-sc->gap)) + 2;
    w.wcols = (uint32_t)std::min<uint64_t>(full, wide);
    const uint64_t len = std::min<uint64_t>((uint64_t)max_lenA + lenB, W);
    w.stride = (uint32_t)std::max<uint64_t>(len, 1);
    return w;
}

// Columns THIS pair needs (<= wcols, the batch-wide bound), from its own end row eA and score M.
//  * The walk spans at most  span = eA + (smax*eA - M)/|gap|  columns: it climbs at most eA rows,
//    gains at most smax per row and ends up with M, so that is all it can pay for left moves.
//  * A windowed H never exceeds the true H (fewer paths), and equals it when an optimal path into the
//    cell lies inside the window.  At a walk cell (row i, value h >= M - smax*(eA - i), since walking
//    back one row loses at most smax) the decision -- which candidate equals h first -- only needs
//    the candidates that reach h to be exact; a smaller one may be underestimated without changing
//    it.  Such a predecessor holds H' >= h - smax at a row i' <= i, so every path achieving it has at
//    most (smax*i' - H')/|gap| <= (smax*eA - M + smax)/|gap| left moves and i' other moves:
//    cand = eA + (smax*eA + smax - M)/|gap| columns to its left are enough.
//  `wide` (POLYHIP_TB_WIDE=1, a testing aid) replaces cand by the bound that makes EVERY cell of the
//  rows <= eA exact, eA + smax*eA/|gap|; tests check both give the same alignments.
__device__ __forceinline__ uint32_t pair_window(uint32_t wcols, uint32_t eA, int64_t M, int smax, int gap, int wide)
{
    if (gap >= 0 || smax <= 0 || M <= 0)
        return wcols;
    const uint64_t g = (uint64_t)(-gap), top = (uint64_t)smax * eA;
    const uint64_t span = eA + (top > (uint64_t)M ? (top - (uint64_t)M) / g : 0);
    const uint64_t over = top + (uint64_t)smax > (uint64_t)M ? (top + (uint64_t)smax - (uint64_t)M) / g + 1 : 0;
    // Round 4: the two bounds are not independent.  A walk cell in row i sits at column j >= eB - (eA - i) - lw (it is
    // eA - i rows and at most lw = (smax*eA - M)/|gap| left moves away from the end cell), and a path into one of its
    // candidates climbs at most i rows: it begins at column >= j - 1 - i - over >= eB - eA - lw - over - 1, wherever on
    // the walk the cell is.  So the DP needs span + over columns, not span + eA + over: the eA columns "to the left of the
    // walk" were counted as if the walk could be at its leftmost column while still in its bottom row.  (`wide` keeps the
    // old sum with every cell of the rows <= eA exact; the tests compare the two on every pair.)
    const uint64_t need = (wide ? span + eA + top / g : span + std::min<uint64_t>(over, top / g)) + 2;
    return need < wcols ? (uint32_t)need : wcols;
}

// shared by both kernels: walk the direction bits of one pair (one lane)
__device__ __forceinline__ uint32_t walk(const uint32_t *__restrict__ dirw, uint32_t nw, uint32_t lane_stride,
                                         const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t eA,
                                         uint32_t eB, uint32_t c_s, uint8_t *__restrict__ outA,
                                         uint8_t *__restrict__ outB, uint32_t stride)
{
    uint32_t i = eA, j = eB, len = 0;
    while (i > 0 && j >= c_s && j > 0) {
        const uint32_t jr = j - c_s;
        // L2-served load: the words were stored by this same lane a moment ago
        const uint32_t word = __hip_atomic_load(&dirw[((size_t)jr * nw + ((i - 1) >> 4)) * lane_stride], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t code = (word >> (2 * ((i - 1) & 15))) & 3u;
        if (code == 0u || len >= stride)
            break;
        uint8_t ca, cb;
        if (code == 1u) { // align.go:215-219
            ca = a[i - 1];
            cb = b[j - 1];
            --i;
            --j;
        } else if (code == 2u) { // :220-223
            ca = a[i - 1];
            cb = '-';
            --i;
        } else { // :224-227
            ca = '-';
            cb = b[j - 1];
            --j;
        }
        outA[stride - 1 - len] = ca; // strings are built by prepending: fill from the back
        outB[stride - 1 - len] = cb;
        ++len;
    }
    return len;
}

template <int RA>
__global__ __launch_bounds__(THREADS) void tb_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                    uint64_t pair0, uint64_t pair1, const uint8_t *__restrict__ B,
                                                    const uint64_t *__restrict__ offB, uint64_t lenB_shared,
                                                    const uint8_t *__restrict__ codeA, const uint8_t *__restrict__ codeB,
                                                    const int32_t *__restrict__ lutcc, int na, int nb, int gap,
                                                    const uint32_t *__restrict__ endA, const uint32_t *__restrict__ endB,
                                                    const uint32_t *__restrict__ err,
                                                    const int64_t *__restrict__ score, int smax, uint32_t wcols, int wide,
                                                    uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA,
                                                    uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen,
                                                    uint32_t stride)
{
    static_assert(RA % 16 == 0 || RA == 152, "RA");
    constexpr int NW = (RA + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) int32_t T[]; // [na][nb] then codeA[256], codeB[256]
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();

    const uint64_t pair = pair0 + (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < pair1;
    uint32_t lenA = 0, eA = 0, eB = 0;
    const uint8_t *ap = A, *bp = B;
    if (active) {
        const uint64_t o0 = offA[pair];
        lenA = (uint32_t)(offA[pair + 1] - o0);
        ap = A + o0;
        if (offB)
            bp = B + offB[pair];
        if (err[pair] == 0u) {
            eA = endA[pair];
            eB = endB[pair];
        }
    }
    const bool work = active && eA > 0 && eB > 0 && lenA <= RA;
    const uint32_t mycols = work ? pair_window(wcols, eA, score ? score[pair] : 0, smax, gap, wide) : 0u;
    const uint32_t c_s = (work && eB > mycols) ? eB - mycols + 1u : 1u; // first column (1-based) of my window
    const uint32_t ncol = work ? eB - c_s + 1u : 0u;

    // row offsets into T (code * nb), two per register; rows >= lenA use the pad row (all zero)
    uint32_t aoff[RA / 2];
#pragma unroll
    for (int r = 0; r < RA / 2; ++r) {
        uint32_t pk = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * r + h;
            uint32_t code = (uint32_t)(na - 1);
            if (work && (uint32_t)i < lenA) {
                const uint32_t c = cA[ap[i]];
                code = c == 0xFFu ? (uint32_t)(na - 1) : c;
            }
            pk |= (code * (uint32_t)nb) << (16 * h);
        }
        aoff[r] = pk;
    }

    int H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;

    // my direction words: word (jr, w) at dirbuf[((wave slab) + jr * NW + w) * 64 + lane]
    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    uint32_t *dirw = dirbuf + wave_global * ((size_t)wcols * NW * 64) + lane;

    for (uint32_t jr = 0; jr < wcols; ++jr) {
        if (!__any(jr < ncol))
            break;
        if (jr < ncol) {
            const uint32_t j = c_s + jr; // 1-based column
            uint32_t cb = cB[bp[j - 1]];
            cb = cb == 0xFFu ? (uint32_t)(nb - 1) : cb;
            int diag = 0, up = 0;
            uint32_t word = 0;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const uint32_t ro = (aoff[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const int s = T[ro + cb];
                const int left = H[i];
                const int d = diag + s, u = up + gap, l = left + gap;
                const int h = max(0, max(d, max(u, l)));
                // the reference's if-chain (align.go:215-228): diag, then up, then left
                const uint32_t code = h == 0 ? 0u : (h == d ? 1u : (h == u ? 2u : 3u));
                word |= code << (2 * (i & 15));
                diag = left;
                up = h;
                H[i] = h;
                if ((i & 15) == 15 || i == RA - 1) {
                    dirw[((size_t)jr * NW + (i >> 4)) * 64] = word;
                    word = 0;
                }
            }
        }
    }

    if (!active)
        return;
    uint32_t len = 0;
    if (work) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my own stores, read back by me
        len = walk(dirw, NW, 64, ap, bp, eA, eB, c_s, alnA + pair * stride, alnB + pair * stride, stride);
    }
    alnLen[pair] = (active && eA > 0 && lenA > RA) ? 0xFFFFFFFFu : len;
}

// ---- shared-reference traceback on the byte profile (the hot one: BASELINE config 4) ----------------
// Same window, one pair per lane and H column in registers as tb_kernel, but built like the score
// pass (sw_batch.hip): the reference's profile prof[j/4][code][j%4] = S(sym(code), b_j) (int8) sits
// WHOLE in LDS, a lane sweeps its own window in blocks of 4 columns, a row of a block costs one
// ds_read_b32 (the lane's own block address + its row's code byte, one SDWA add), and a cell is
//     d0 = max(diag + s, 0)   t = max(up, left) + gap   h = max(d0, t)
// plus two recorded bits, each the carry of one compare shifted into a word by v_addc_co_u32:
//     G = t > d0     the cell was reached by a gap move (never set where the diagonal ties: the
//                    reference tests the diagonal first, align.go:215)
//     L = left > up  ... and that gap move is "left" (only on a strict win: "up" is tested first, :220)
// The "H is 0" case needs no bit: the walk carries the running score, starting from the score
// pass's maximum and subtracting what each move contributed (H of the predecessor, exactly), and
// stops when it reaches 0 -- the reference's `for H[i][j] > 0`.
// Words: per column and group of 32 rows one G word and one L word (row r of the group at bit
// rows_in_group-1-r), stored lane-interleaved like tb_kernel's.
constexpr int TBU = 4; // columns per block

__global__ __launch_bounds__(256) void tb_profile_kernel(const uint8_t *__restrict__ B, uint32_t lenB, uint32_t lenB_pad,
                                                        const int8_t *__restrict__ lutc, int ncodes, int cp,
                                                        int8_t *__restrict__ prof)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lenB_pad)
        return;
    int8_t *col = prof + (size_t)(j >> 2) * cp * 4 + (j & 3);
    const int live = j < lenB ? ncodes : 0; // pad columns / pad codes: -128 keeps every H at 0
    const uint8_t b = j < lenB ? B[j] : 0;
    for (int c = 0; c < cp; ++c)
        col[c * 4] = c < live ? lutc[c * 256 + b] : (int8_t)-128;
}

// w = 2 * w + (x > y).  FIND (the wave's last block only): the first cell worth M in row-major order -- rows are
// visited in ascending order and a row's four columns left to right, so the first hit is it
#define PH_TB_CELL(S, DIAG, UP, LEFT, HOUT, C)         \
    do {                                               \
        const int up_ = (UP), left_ = (LEFT);          \
        const int d0_ = max((DIAG) + (S), 0);          \
        const int t_ = max(up_, left_) + gap;          \
        HOUT = max(d0_, t_);                           \
        PH_CARRY_BIT(wG[C], t_, d0_);                     \
        PH_CARRY_BIT(wL[C], left_, up_);                  \
        if (FIND)                                      \
            key = (key == 0xFFFFFFFFu && HOUT == Mi) ? (uint32_t)((i_ << 2) | (C)) : key; \
    } while (0)
#define PH_TB_ROW(I, W)                                         \
    do {                                                        \
        const int i_ = (I);                                     \
        const uint32_t w_ = (W);                                \
        const int s0 = (int)(int8_t)(w_);                       \
        const int s1 = (int)(int8_t)(w_ >> 8);                  \
        const int s2 = (int)(int8_t)(w_ >> 16);                 \
        const int s3 = (int)w_ >> 24;                           \
        const int left = H[i_];                                 \
        int h0, h1, h2, h3;                                     \
        PH_TB_CELL(s0, pdiag, pr0, left, h0, 0);                \
        PH_TB_CELL(s1, pr0, pr1, h0, h1, 1);                    \
        PH_TB_CELL(s2, pr1, pr2, h1, h2, 2);                    \
        PH_TB_CELL(s3, pr2, pr3, h2, h3, 3);                    \
        pdiag = left;                                           \
        pr0 = h0;                                               \
        pr1 = h1;                                               \
        pr2 = h2;                                               \
        pr3 = h3;                                               \
        H[i_] = h3;                                             \
        if ((i_ & 31) == 31 || i_ == RA - 1) {                  \
            uint32_t *o_ = dirw + ((size_t)tt * TBU * NG + (i_ >> 5)) * 2 * 64; \
            _Pragma("unroll") for (int c_ = 0; c_ < TBU; ++c_)  \
            {                                                   \
                o_[((size_t)c_ * NG * 2 + 0) * 64] = wG[c_];    \
                o_[((size_t)c_ * NG * 2 + 1) * 64] = wL[c_];    \
            }                                                   \
        }                                                       \
    } while (0)

template <int RA, int CP>
__global__ __launch_bounds__(THREADS) void tb_prof_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ B, uint32_t lenB_pad, const int8_t *__restrict__ prof,
    const uint8_t *__restrict__ codeA, int ncodes, int gap, uint32_t *__restrict__ endA,
    uint32_t *__restrict__ endB, uint32_t *__restrict__ err, const int64_t *__restrict__ score, int smax,
    uint32_t wcols, int wide, uint32_t nblk_alloc, uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA,
    uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen, uint32_t stride)
{
    static_assert(RA % 4 == 0 && RA <= 256, "RA");
    constexpr int NG = (RA + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) int8_t lds_tb[];
    int8_t *P = lds_tb;
    uint8_t *codeL = reinterpret_cast<uint8_t *>(lds_tb + (size_t)lenB_pad * CP);
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(prof);
        uint4 *dst = reinterpret_cast<uint4 *>(P);
        const uint32_t nvec = lenB_pad * CP / 16; // lenB_pad % 4 == 0 and CP % 8 == 0
        for (uint32_t v = tid; v < nvec; v += THREADS)
            dst[v] = src[v];
    }
    codeL[tid] = codeA[tid];
    __syncthreads();

    const uint64_t pair = pair0 + (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < pair1;
    uint32_t lenA = 0, eA = 0, eB = 0;
    int64_t M = 0;
    const uint8_t *ap = A;
    if (active) {
        const uint64_t o0 = offA[pair];
        lenA = (uint32_t)(offA[pair + 1] - o0);
        ap = A + o0;
        if (err[pair] == 0u) {
            eA = endA[pair];
            eB = endB[pair];
            M = score[pair];
        }
    }
    // The score pass may have left the end cell to this kernel (k3p::SW_END_DEFERRED): eB is then the last column of
    // the ONE block of four columns that holds the maximum, and the first cell worth M in row-major order is found
    // while that block -- the last of the window -- is swept.  The window is sized with lenA for the unknown end row.
    // `wide` bit 1: the caller is the fused entry point, whose score pass may have deferred end cells; without it the
    // end arrays are the caller's read-only input, a sentinel (or any row beyond the read) there means "no alignment"
    const bool locate = active && (wide & 2) && eA == k3p::SW_END_DEFERRED;
    const uint32_t rowsA = locate ? lenA : eA;
    const bool work = active && rowsA > 0 && rowsA <= lenA && eB > 0 && M > 0 && lenA <= RA;
    const uint32_t mycols = work ? min(wcols + 4u, pair_window(wcols, rowsA, M, smax, gap, wide & 1) + (locate ? 4u : 0u)) : 0u;
    const uint32_t c_s = (work && eB > mycols) ? eB - mycols + 1u : 1u; // first column (1-based) of my window
    const uint32_t jb0 = (c_s - 1u) & ~3u;                               // 0-based, on a block boundary
    const uint32_t nblk = work ? (eB - jb0 + TBU - 1) / TBU : 0u;        // <= nblk_alloc

    // packed byte offsets (code * 4) of my rows inside a profile block; rows >= lenA use the pad code
    uint32_t apk[RA / 4];
#pragma unroll
    for (int w = 0; w < RA / 4; ++w) {
        uint32_t pk = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = 4 * w + b;
            uint32_t code = (uint32_t)ncodes;
            if (work && (uint32_t)i < lenA) {
                const uint32_t c = codeL[ap[i]];
                code = c == 0xFFu ? (uint32_t)ncodes : c;
            }
            pk |= (code * 4u) << (8 * b);
        }
        apk[w] = pk;
    }

    int H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;

    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    uint32_t *dirw = dirbuf + wave_global * ((size_t)nblk_alloc * TBU * NG * 2 * 64) + lane;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P));

    // Lanes sweep different numbers of blocks.  They are aligned at the END: a lane with fewer blocks idles first, so
    // that the wave's last iteration is every lane's last block and the locate code is paid once per wave.
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    const int Mi = (int)M;
    uint32_t key = 0xFFFFFFFFu;
    // tt = the wave's iteration (uniform: the direction words of one iteration are stored side by side, 256 B per
    // store), bt = tt - lag = the lane's own block
    auto sweep = [&](uint32_t tt, auto find_tag) {
        constexpr bool FIND = decltype(find_tag)::value;
        const uint32_t bt = tt - lag;
        const uint32_t blk = lds_base + ((jb0 >> 2) + bt) * (CP * 4);
        int pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pdiag = 0;
        uint32_t wG[TBU] = {0u, 0u, 0u, 0u}, wL[TBU] = {0u, 0u, 0u, 0u};
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
            PH_TB_ROW(4 * g + 0, wa0);
            PH_TB_ROW(4 * g + 1, wa1);
            PH_TB_ROW(4 * g + 2, wa2);
            PH_TB_ROW(4 * g + 3, wa3);
            wa0 = wb0;
            wa1 = wb1;
            wa2 = wb2;
            wa3 = wb3;
        }
    };
    const bool any_locate = __any(locate) != 0; // wave-uniform
    const uint32_t nplain = any_locate && nmax > 0 ? nmax - 1u : nmax;
    for (uint32_t t = 0; t < nplain; ++t)
        if (t >= lag) // (nblk == 0: lag == nmax, never)
            sweep(t, std::false_type{});
    if (nplain < nmax && nblk > 0) // the wave's last iteration, every sweeping lane's last block: with the search
        sweep(nmax - 1u, std::true_type{});

    if (!active)
        return;
    uint32_t len = 0;
    bool lost = false;
    if (work && locate) {
        lost = key == 0xFFFFFFFFu; // cannot happen: the packed pass saw M in this block
        eA = lost ? 0u : (key >> 2) + 1u;
        eB = lost ? 0u : eB - 3u + (key & 3u);
        endA[pair] = eA;
        endB[pair] = eB;
        if (lost)
            err[pair] = 0xFFFFFFFEu;
    } else if (locate) { // deferred, but nothing to walk (cannot happen either: M > 0 and lenA > 0)
        endA[pair] = 0u;
        endB[pair] = 0u;
    }
    if (work && !lost) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my own stores, read back by me
        uint8_t *outA = alnA + pair * stride, *outB = alnB + pair * stride;
        uint32_t i = eA, j = eB;
        int h = (int)M;
        while (h > 0 && i > 0 && j > jb0 && len < stride) {
            const uint32_t jj = j - 1u, rel = jj - jb0 + TBU * lag, r = i - 1u, g = r >> 5; // (stored by wave iteration)
            const uint32_t rows = min(32u, (uint32_t)RA - 32u * g);
            const uint32_t bit = rows - 1u - (r & 31u);
            const uint32_t *wp = dirw + ((size_t)rel * NG + g) * 2 * 64;
            // L2-served loads: the words were stored by this same lane a moment ago
            const uint32_t wg = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t wl = __hip_atomic_load(wp + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint8_t sa = ap[r], sb = B[jj];
            uint8_t ca, cb;
            if (((wg >> bit) & 1u) == 0u) { // align.go:215-219
                h -= (int)P[((size_t)(jj >> 2) * CP + codeL[sa]) * 4 + (jj & 3u)];
                ca = sa;
                cb = sb;
                --i;
                --j;
            } else if (((wl >> bit) & 1u) == 0u) { // :220-223
                h -= gap;
                ca = sa;
                cb = '-';
                --i;
            } else { // :224-227
                h -= gap;
                ca = '-';
                cb = sb;
                --j;
            }
            outA[stride - 1 - len] = ca; // strings are built by prepending: fill from the back
            outB[stride - 1 - len] = cb;
            ++len;
        }
    }
    alnLen[pair] = (active && rowsA > 0 && lenA > RA) ? 0xFFFFFFFFu : len;
}
#undef PH_TB_ROW
#undef PH_TB_CELL

// ---- the same kernel on gfx950's packed half-floats: TWO BANDS of rows per lane ---------------------------
// When every H stays below 2048 (the packed score pass's condition, sw_packed.hip) the recurrence runs on halves scaled
// by 2^-11, three instructions per cell PAIR (add, v_pk_maximum3_f16, clamped add). One pair still owns a lane -- lanes
// sit in different windows, so two pairs cannot share a register as they do in the score pass -- but its rows are split
// into two bands of RB: rows [0, RB) in the low halves, rows [RB, 2 RB) in the high halves, the lower band one 4-column
// block behind (its row RB needs row RB - 1 of the same columns: band 0's last row, four values + their gap-decayed
// copies + one diagonal value, moves from the low to the high halves between two blocks). Direction bits: G = the gap
// move won = h > diag + s (wherever h > 0, the only cells the walk reads), L = left > up, told from the gap-decayed
// values (equal to comparing left and up wherever G holds); each is a clamped difference, an unsigned min with 1 and
// one v_pk_mad_u16 (w = 2 w + bit) -- nine instructions per cell pair against eighteen. A G / L word carries 16 rows of
// band 0 (low half) and 16 rows of band 1 (high half, one block earlier). The profile is a table of halves,
// P16[block][code][4], read with two ds_read_b64 per row pair (one per band) and interleaved by four v_perm_b32.
__device__ __forceinline__ uint32_t tbf_half_bits(int v)
{
    const _Float16 h = (_Float16)((float)v * (1.0f / 2048.0f));
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ int tbf_half_score(uint32_t bits)
{
    return (int)((float)__builtin_bit_cast(_Float16, (unsigned short)bits) * 2048.0f);
}
__device__ __forceinline__ uint32_t tbf_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t tbf_addc(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t tbf_subc(uint32_t a, uint32_t b) // max(0, a - b), per half
{
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t tbf_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// w = 2 * w + (x != 0), per 16-bit half (x: non-negative halves)
__device__ __forceinline__ uint32_t tbf_push(uint32_t w, uint32_t x, uint32_t one2, uint32_t two2)
{
    uint32_t f, r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(x), "v"(one2));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(w), "v"(two2), "v"(f));
    return r;
}

// P16 entry (q, c) = the four halves S(sym c, b_{4q..4q+3}) * 2^-11; pad columns, the pad code (c = ncodes) and one
// all-pad block behind the last one (q = lenB_pad / 4: what a band reads where it has no block) hold -128
__global__ __launch_bounds__(256) void tb_profile16_kernel(const uint8_t *__restrict__ B, uint32_t lenB, uint32_t lenB_pad,
                                                          const int8_t *__restrict__ lutc, int ncodes, int ncp,
                                                          uint2 *__restrict__ prof16)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nq = lenB_pad / 4 + 1;
    if (e >= nq * (uint32_t)ncp)
        return;
    const uint32_t q = e / (uint32_t)ncp;
    const int c = (int)(e % (uint32_t)ncp);
    uint32_t h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t j = 4 * q + u;
        int sv = -128;
        if (j < lenB && c < ncodes)
            sv = lutc[c * 256 + B[j]];
        h[u] = tbf_half_bits(sv);
    }
    prof16[e] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
}

#define PH_TBF_CELL(W, DIAG, UPG, LEFTG, H, GOUT, C)                        \
    do {                                                                    \
        const uint32_t t_ = tbf_add((DIAG), (W));                           \
        H = tbf_max3(t_, (UPG), (LEFTG));                                   \
        wG[C] = tbf_push(wG[C], tbf_subc(H, t_), one2, two2);               \
        wL[C] = tbf_push(wL[C], tbf_subc((LEFTG), (UPG)), one2, two2);      \
        GOUT = tbf_addc(H, gap2);                                           \
        if (FIND)                                                           \
            key = (key == 0xFFFFFFFFu && ((H >> (FIND == 2 ? 16 : 0)) & 0xFFFFu) == Mh) \
                      ? (uint32_t)(((r_ + (FIND == 2 ? RB : 0)) << 2) | (C)) \
                      : key;                                                \
    } while (0)
// one row pair: row r_ of band 0 (low halves, block bt) and row RB + r_ of band 1 (high halves, block bt - 1);
// (X0, X1) / (Y0, Y1) = the two dwords of band 0's / band 1's profile entry
#define PH_TBF_ROW(R, X0, X1, Y0, Y1)                                       \
    do {                                                                    \
        const int r_ = (R);                                                 \
        const uint32_t w0 = __builtin_amdgcn_perm((Y0), (X0), 0x05040100u); \
        const uint32_t w1 = __builtin_amdgcn_perm((Y0), (X0), 0x07060302u); \
        const uint32_t w2 = __builtin_amdgcn_perm((Y1), (X1), 0x05040100u); \
        const uint32_t w3 = __builtin_amdgcn_perm((Y1), (X1), 0x07060302u); \
        const uint32_t left = H[r_];                                        \
        const uint32_t gl = tbf_addc(left, gap2);                           \
        uint32_t h0, h1, h2, h3, g0, g1, g2, g3;                            \
        PH_TBF_CELL(w0, pdiag, pg0, gl, h0, g0, 0);                         \
        PH_TBF_CELL(w1, pr0, pg1, g0, h1, g1, 1);                           \
        PH_TBF_CELL(w2, pr1, pg2, g1, h2, g2, 2);                           \
        PH_TBF_CELL(w3, pr2, pg3, g2, h3, g3, 3);                           \
        pdiag = left;                                                       \
        pr0 = h0;                                                           \
        pr1 = h1;                                                           \
        pr2 = h2;                                                           \
        pr3 = h3;                                                           \
        pg0 = g0;                                                           \
        pg1 = g1;                                                           \
        pg2 = g2;                                                           \
        pg3 = g3;                                                           \
        H[r_] = h3;                                                         \
        if ((r_ & 15) == 15 || r_ == RB - 1) {                              \
            /* a lane's eight words of (iteration, row group) are ONE 32-byte piece, (G, L) of a column side by side: the \
               walk, which visits a block's four columns one after the other, then fetches one piece per block and row  \
               group instead of two words in two different lines per step (19 GB of 64-byte fetches for 1M reads)        \
             */                                                             \
            uint4 *o_ = reinterpret_cast<uint4 *>(dirw + ((size_t)tt * NG + (r_ >> 4)) * (64 * 8)); \
            o_[0] = make_uint4(wG[0], wL[0], wG[1], wL[1]);                 \
            o_[1] = make_uint4(wG[2], wL[2], wG[3], wL[3]);                 \
        }                                                                   \
    } while (0)
// Two row pairs at a time with their cells interleaved along the anti-diagonal (row r's cell c + 1 and row r + 1's cell c
// depend on nothing of each other): twice the independent work between two dependent packed instructions.  Only where no
// end cell is searched (FIND == 0: the search needs row-major order).  -DPH_TBF_WF: a measured variant (scripts/build_variant.sh).
#define PH_TBF_ROW2(R, XA0, XA1, YA0, YA1, XB0, XB1, YB0, YB1)                 \
    do {                                                                       \
        const uint32_t wa0 = __builtin_amdgcn_perm((YA0), (XA0), 0x05040100u); \
        const uint32_t wa1 = __builtin_amdgcn_perm((YA0), (XA0), 0x07060302u); \
        const uint32_t wa2 = __builtin_amdgcn_perm((YA1), (XA1), 0x05040100u); \
        const uint32_t wa3 = __builtin_amdgcn_perm((YA1), (XA1), 0x07060302u); \
        const uint32_t wb0 = __builtin_amdgcn_perm((YB0), (XB0), 0x05040100u); \
        const uint32_t wb1 = __builtin_amdgcn_perm((YB0), (XB0), 0x07060302u); \
        const uint32_t wb2 = __builtin_amdgcn_perm((YB1), (XB1), 0x05040100u); \
        const uint32_t wb3 = __builtin_amdgcn_perm((YB1), (XB1), 0x07060302u); \
        const uint32_t lefta = H[(R)], leftb = H[(R) + 1];                     \
        const uint32_t gla = tbf_addc(lefta, gap2), glb = tbf_addc(leftb, gap2); \
        uint32_t ha0, ha1, ha2, ha3, ga0, ga1, ga2, ga3, hb0, hb1, hb2, hb3, gb0, gb1, gb2, gb3; \
        { const int r_ = (R);     PH_TBF_CELL(wa0, pdiag, pg0, gla, ha0, ga0, 0); } \
        { const int r_ = (R);     PH_TBF_CELL(wa1, pr0, pg1, ga0, ha1, ga1, 1); }   \
        { const int r_ = (R) + 1; PH_TBF_CELL(wb0, lefta, ga0, glb, hb0, gb0, 0); } \
        { const int r_ = (R);     PH_TBF_CELL(wa2, pr1, pg2, ga1, ha2, ga2, 2); }   \
        { const int r_ = (R) + 1; PH_TBF_CELL(wb1, ha0, ga1, gb0, hb1, gb1, 1); }   \
        { const int r_ = (R);     PH_TBF_CELL(wa3, pr2, pg3, ga2, ha3, ga3, 3); }   \
        { const int r_ = (R) + 1; PH_TBF_CELL(wb2, ha1, ga2, gb1, hb2, gb2, 2); }   \
        { const int r_ = (R) + 1; PH_TBF_CELL(wb3, ha2, ga3, gb2, hb3, gb3, 3); }   \
        pdiag = leftb;                                                         \
        pr0 = hb0;                                                             \
        pr1 = hb1;                                                             \
        pr2 = hb2;                                                             \
        pr3 = hb3;                                                             \
        pg0 = gb0;                                                             \
        pg1 = gb1;                                                             \
        pg2 = gb2;                                                             \
        pg3 = gb3;                                                             \
        H[(R)] = ha3;                                                          \
        H[(R) + 1] = hb3;                                                      \
        if ((((R) + 1) & 15) == 15 || (R) + 1 == RB - 1) {                     \
            uint4 *o_ = reinterpret_cast<uint4 *>(dirw + ((size_t)tt * NG + (((R) + 1) >> 4)) * (64 * 8)); \
            o_[0] = make_uint4(wG[0], wL[0], wG[1], wL[1]);                    \
            o_[1] = make_uint4(wG[2], wL[2], wG[3], wL[3]);                    \
        }                                                                      \
    } while (0)
// the profile entries of four row pairs (one packed register of code offsets per band): eight ds_read_b64
#define PH_TBF_ISSUE(pk0, pk1, X, Y)                                                                       \
    do {                                                                                                   \
        uint32_t a_[8];                                                                                    \
        PH_TBF_ADDR(a_[0], base0, pk0, "BYTE_0");                                                          \
        PH_TBF_ADDR(a_[1], base1, pk1, "BYTE_0");                                                          \
        PH_TBF_ADDR(a_[2], base0, pk0, "BYTE_1");                                                          \
        PH_TBF_ADDR(a_[3], base1, pk1, "BYTE_1");                                                          \
        PH_TBF_ADDR(a_[4], base0, pk0, "BYTE_2");                                                          \
        PH_TBF_ADDR(a_[5], base1, pk1, "BYTE_2");                                                          \
        PH_TBF_ADDR(a_[6], base0, pk0, "BYTE_3");                                                          \
        PH_TBF_ADDR(a_[7], base1, pk1, "BYTE_3");                                                          \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[0]) : "v"(a_[0]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[0]) : "v"(a_[1]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[1]) : "v"(a_[2]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[1]) : "v"(a_[3]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[2]) : "v"(a_[4]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[2]) : "v"(a_[5]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[3]) : "v"(a_[6]));                                      \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[3]) : "v"(a_[7]));                                      \
    } while (0)
#define PH_TBF_ADDR(dst, base, pk, SEL)                                                                    \
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL \
                 : "=v"(dst)                                                                               \
                 : "v"(base), "v"(pk))

typedef uint32_t tbf_u32x2 __attribute__((ext_vector_type(2)));

// The walk over the half-float kernels' direction words (tb_prof16_kernel: LP = 1 lane per pair, two bands; tb_prof16x2_kernel:
// LP = 2, four bands): `dirw` = the pair's (first) lane's piece inside its wave's buffer, jb0 / lag = where the pair's
// window began and how many iterations its wave started ahead of it.  COHERENT: the words were stored by this very wave a
// moment ago (agent-scope loads); else by a kernel that has finished (plain loads: a block's four columns are one 32-byte
// piece, the second to fourth step of a diagonal hit L1).  P16 / codeL may live in LDS or in global memory.
template <int RB, int LP, bool COHERENT, bool BYTAB = false> // BYTAB: P16 = halves [code of a][byte of b] (every pair its own B)
__device__ __forceinline__ uint32_t tbf_walk(const uint32_t *__restrict__ dirw, uint32_t jb0, uint32_t lag, uint32_t eA, uint32_t eB,
                                             int M, int gap, int ncp, const uint16_t *P16, const uint8_t *codeL,
                                             const uint8_t *__restrict__ ap, const uint8_t *__restrict__ B,
                                             uint8_t *__restrict__ outA, uint8_t *__restrict__ outB, uint32_t stride)
{
    // A step used to be a chain of its own: two byte loads, the direction words' load, two byte stores per cell -- 150
    // dependent round trips per read.  Now the lane keeps (i) the 32-byte piece of direction words it is in (a block's four
    // columns x (G, L) of one row group: a diagonal stays in it for up to four steps), (ii) the aligned dword of the read
    // that holds the current row's symbol, (iii) eight finished characters of either string (one unaligned 8-byte store
    // per eight steps: strings are filled from the back, so the newest character is the lowest byte).
    constexpr int NG = (RB + 15) / 16;
    uint32_t i = eA, j = eB, len = 0;
    int h = M;
    uint64_t pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0; // the piece: (G, L) of columns 0..3
    uint32_t pkey = 0xFFFFFFFFu;
    const uint32_t a0 = (uint32_t)(reinterpret_cast<uintptr_t>(ap) & 3u);
    const uint32_t *apw = reinterpret_cast<const uint32_t *>(ap - a0); // aligned dwords over the read (the same words its bytes live in)
    uint32_t aw = 0, akey = 0xFFFFFFFFu;
    uint64_t accA = 0, accB = 0;
    while (h > 0 && i > 0 && j > jb0 && len < stride) {
        const uint32_t jj = j - 1u, r = i - 1u, band = r / (uint32_t)RB, rr = r - band * RB, g = rr >> 4;
        const uint32_t rows = min(16u, (uint32_t)RB - 16u * g);
        const uint32_t bit = rows - 1u - (rr & 15u) + 16u * (band & 1u);
        const uint32_t tt = ((jj - jb0) >> 2) + lag + band; // the wave iteration that swept this cell
        const uint32_t key = (tt * NG + g) * 2u + (LP == 2 ? (band >> 1) : 0u);
        if (key != pkey) {
            const uint32_t *pp = dirw + (LP == 2 ? (band >> 1) * 8u : 0u) + ((size_t)tt * NG + g) * (64 * 8);
            if (COHERENT) { // stored by this very wave a moment ago
                const uint64_t *p8 = reinterpret_cast<const uint64_t *>(pp);
                pc0 = __hip_atomic_load(p8 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pc1 = __hip_atomic_load(p8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pc2 = __hip_atomic_load(p8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pc3 = __hip_atomic_load(p8 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const uint4 q0 = reinterpret_cast<const uint4 *>(pp)[0], q1 = reinterpret_cast<const uint4 *>(pp)[1];
                pc0 = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
                pc1 = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
                pc2 = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
                pc3 = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
            }
            pkey = key;
        }
        const uint32_t ar = a0 + r;
        if ((ar >> 2) != akey) {
            akey = ar >> 2;
            aw = apw[akey];
        }
        const uint32_t c = jj & 3u;
        const uint64_t w2 = c == 0u ? pc0 : c == 1u ? pc1 : c == 2u ? pc2 : pc3;
        const uint32_t wg = (uint32_t)w2, wl = (uint32_t)(w2 >> 32);
        const uint8_t sa = (uint8_t)(aw >> (8u * (ar & 3u))), sb = B[jj];
        uint8_t ca, cb;
        if (((wg >> bit) & 1u) == 0u) { // align.go:215-219
            h -= tbf_half_score(BYTAB ? P16[(size_t)codeL[sa] * 256 + sb] : P16[((size_t)(jj >> 2) * ncp + codeL[sa]) * 4 + (jj & 3u)]);
            ca = sa;
            cb = sb;
            --i;
            --j;
        } else if (((wl >> bit) & 1u) == 0u) { // :220-223
            h -= gap;
            ca = sa;
            cb = '-';
            --i;
        } else { // :224-227
            h -= gap;
            ca = '-';
            cb = sb;
            --j;
        }
        accA = (accA << 8) | ca;
        accB = (accB << 8) | cb;
        ++len;
        if ((len & 7u) == 0u) { // characters stride - len .. stride - len + 7
            __builtin_memcpy(outA + (stride - len), &accA, 8);
            __builtin_memcpy(outB + (stride - len), &accB, 8);
        }
    }
    for (uint32_t k = 0, n = len & 7u; k < n; ++k) { // the last, partly filled group: its newest character is the lowest byte
        outA[stride - len + k] = (uint8_t)(accA >> (8u * k));
        outB[stride - len + k] = (uint8_t)(accB >> (8u * k));
    }
    return len;
}

// The walk as a kernel of its own (round 4; POLYHIP_TB_SPLITWALK=1 -- it lost, see traceback_impl): a thread per pair, no
// LDS, under 30 registers, eight waves per SIMD.  The sweep kernel leaves (jb0, lag) per pair in `walkinfo` ((0 | 1, ~0):
// nothing to walk, alnLen = 0 | ~0).
template <int RB, int LP>
__global__ __launch_bounds__(256) void tb_walk16_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0,
                                                       uint64_t pair1, const uint8_t *__restrict__ B,
                                                       const uint16_t *__restrict__ prof16, const uint8_t *__restrict__ codeA, int ncodes,
                                                       int gap, const uint32_t *__restrict__ endA, const uint32_t *__restrict__ endB,
                                                       const int64_t *__restrict__ score, uint32_t nblk_alloc,
                                                       const uint32_t *__restrict__ dirbuf, const uint2 *__restrict__ walkinfo,
                                                       uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB,
                                                       uint32_t *__restrict__ alnLen, uint32_t stride)
{
    constexpr int NG = (RB + 15) / 16;
    constexpr uint32_t PPW = 64 / LP; // pairs of one sweep wave
    const uint64_t pl = (uint64_t)blockIdx.x * 256 + threadIdx.x, pair = pair0 + pl;
    if (pair >= pair1)
        return;
    const uint2 info = walkinfo[pl];
    if (info.y == 0xFFFFFFFFu) {
        alnLen[pair] = info.x ? 0xFFFFFFFFu : 0u;
        return;
    }
    const uint32_t *dirw = dirbuf + (pl / PPW) * ((size_t)nblk_alloc * TBU * NG * 2 * 64) + (uint32_t)(pl % PPW) * (LP * 8u);
    alnLen[pair] = tbf_walk<RB, LP, false>(dirw, info.x, info.y, endA[pair], endB[pair], (int)score[pair], gap, ncodes + 1, prof16, codeA,
                                           A + offA[pair], B, alnA + pair * stride, alnB + pair * stride, stride);
}

template <int RB>
__global__ __launch_bounds__(THREADS, 2) void tb_prof16_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ B, uint32_t lenB_pad, const uint2 *__restrict__ prof16,
    const uint8_t *__restrict__ codeA, int ncodes, int gap, uint32_t *__restrict__ endA,
    uint32_t *__restrict__ endB, uint32_t *__restrict__ err, const int64_t *__restrict__ score, int smax,
    uint32_t wcols, int wide, uint32_t nblk_alloc, uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA,
    uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen, uint32_t stride, uint2 *__restrict__ walkinfo)
{
    static_assert(RB % 4 == 0 && RB <= 76, "RB");
    constexpr int RA = 2 * RB;
    constexpr int NG = (RB + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tbf[];
    const int ncp = ncodes + 1;
    const uint32_t nqB = lenB_pad / 4; // real blocks; block nqB is the all-pad one
    const uint32_t pstride = (uint32_t)ncp * 8u;
    uint2 *P = reinterpret_cast<uint2 *>(lds_tbf);
    uint8_t *codeL = lds_tbf + (size_t)(nqB + 1) * pstride;
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t v = tid; v < (nqB + 1) * (uint32_t)ncp; v += THREADS)
        P[v] = prof16[v];
    codeL[tid] = codeA[tid];
    __syncthreads();

    const uint64_t pair = pair0 + (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < pair1;
    uint32_t lenA = 0, eA = 0, eB = 0;
    int64_t M = 0;
    const uint8_t *ap = A;
    uint64_t oA = 0;
    if (active) {
        const uint64_t o0 = offA[pair];
        oA = o0;
        lenA = (uint32_t)(offA[pair + 1] - o0);
        ap = A + o0;
        if (err[pair] == 0u) {
            eA = endA[pair];
            eB = endB[pair];
            M = score[pair];
        }
    }
    const bool locate = active && (wide & 2) && eA == k3p::SW_END_DEFERRED; // as tb_prof_kernel
    const uint32_t rowsA = locate ? lenA : eA;
    const bool work = active && rowsA > 0 && rowsA <= lenA && eB > 0 && M > 0 && lenA <= RA;
    const uint32_t mycols = work ? min(wcols + 4u, pair_window(wcols, rowsA, M, smax, gap, wide & 1) + (locate ? 4u : 0u)) : 0u;
    const uint32_t c_s = (work && eB > mycols) ? eB - mycols + 1u : 1u;
    const uint32_t jb0 = (c_s - 1u) & ~3u;
    const uint32_t nblk = work ? (eB - jb0 + TBU - 1) / TBU : 0u; // <= nblk_alloc - 1

    // byte offsets (code * 8) of my rows inside a profile block, four rows per register and band.  The read's bytes come as
    // RA / 4 + 1 aligned dwords, all in flight at once, through a buffer resource over this launch's reads (as
    // sw_pk1_kernel does; a byte at a time the prologue was RA dependent round trips); 4 GB and more keep the byte loads.
    uint32_t apk0[RB / 4], apk1[RB / 4];
    const uint64_t totalA = offA[pair1];
    if (totalA < 0xFFFFFFF0ull) {
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 3u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A) - misA, 0,
                                                                            (int)(((uint32_t)totalA + misA + 3u) & ~3u), 0x00020000);
        const uint32_t b0 = (uint32_t)oA + misA;
        uint32_t dd[RA / 4];
        {
            uint32_t aw[RA / 4 + 1];
#pragma unroll
            for (int w = 0; w <= RA / 4; ++w)
                aw[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b0 & ~3u) + 4u * w), 0, 0);
#pragma unroll
            for (int w = 0; w < RA / 4; ++w)
                dd[w] = __builtin_amdgcn_alignbyte(aw[w + 1], aw[w], b0 & 3u);
        }
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i0 = 4 * w + b, i1 = RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && (uint32_t)i0 < lenA) {
                    const uint32_t c = codeL[(dd[w] >> (8 * b)) & 0xFFu];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && (uint32_t)i1 < lenA) {
                    const uint32_t c = codeL[(dd[RB / 4 + w] >> (8 * b)) & 0xFFu];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    } else {
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i0 = 4 * w + b, i1 = RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && (uint32_t)i0 < lenA) {
                    const uint32_t c = codeL[ap[i0]];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && (uint32_t)i1 < lenA) {
                    const uint32_t c = codeL[ap[i1]];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    }

    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;

    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    static_assert(TBU == 4, "a lane's piece of direction words is (G, L) x four columns");
    uint32_t *dirw = dirbuf + wave_global * ((size_t)nblk_alloc * TBU * NG * 2 * 64) + lane * 8;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P));
    const uint32_t pad_base = lds_base + nqB * pstride;

    // end-aligned lanes as in tb_prof_kernel; a lane runs nblk + 1 iterations (band 1 finishes one block later)
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    const uint32_t gh = tbf_half_bits(gap); // gap < 0
    const uint32_t gap2 = gh | (gh << 16), one2 = 0x00010001u, two2 = 0x00020002u;
    const uint32_t Mh = tbf_half_bits((int)M);
    uint32_t key = 0xFFFFFFFFu;
    // band 0's last row of the block before, already in the high halves: what band 1 finds above its first row
    uint32_t hh0 = 0, hh1 = 0, hh2 = 0, hh3 = 0, hg0 = 0, hg1 = 0, hg2 = 0, hg3 = 0, hd = 0;
    auto sweep = [&](uint32_t tt, auto find_tag) {
        constexpr int FIND = decltype(find_tag)::value; // 0, 1 = search band 0's cells, 2 = band 1's
        const uint32_t bt = tt - lag;                   // band 0's block; band 1 works on bt - 1
        const uint32_t base0 = bt < nblk ? lds_base + ((jb0 >> 2) + bt) * pstride : pad_base;
        const uint32_t base1 = bt >= 1u ? lds_base + ((jb0 >> 2) + bt - 1u) * pstride : pad_base;
        uint32_t pr0 = hh0, pr1 = hh1, pr2 = hh2, pr3 = hh3, pg0 = hg0, pg1 = hg1, pg2 = hg2, pg3 = hg3, pdiag = hd;
        uint32_t wG[TBU] = {0u, 0u, 0u, 0u}, wL[TBU] = {0u, 0u, 0u, 0u};
        tbf_u32x2 xa[4], ya[4], xb[4], yb[4];
        PH_TBF_ISSUE(apk0[0], apk1[0], xa, ya);
#pragma unroll
        for (int g = 0; g < RB / 4; ++g) {
            if (g + 1 < RB / 4) {
                PH_TBF_ISSUE(apk0[g + 1], apk1[g + 1], xb, yb);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            }
#ifdef PH_TBF_WF
            if constexpr (FIND == 0) {
                PH_TBF_ROW2(4 * g + 0, xa[0].x, xa[0].y, ya[0].x, ya[0].y, xa[1].x, xa[1].y, ya[1].x, ya[1].y);
                PH_TBF_ROW2(4 * g + 2, xa[2].x, xa[2].y, ya[2].x, ya[2].y, xa[3].x, xa[3].y, ya[3].x, ya[3].y);
            } else
#endif
            {
            PH_TBF_ROW(4 * g + 0, xa[0].x, xa[0].y, ya[0].x, ya[0].y);
            PH_TBF_ROW(4 * g + 1, xa[1].x, xa[1].y, ya[1].x, ya[1].y);
            PH_TBF_ROW(4 * g + 2, xa[2].x, xa[2].y, ya[2].x, ya[2].y);
            PH_TBF_ROW(4 * g + 3, xa[3].x, xa[3].y, ya[3].x, ya[3].y);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xa[q] = xb[q];
                ya[q] = yb[q];
            }
        }
        // band 0's last row (low halves) becomes band 1's row above (high halves) of the next iteration
        hd = hh3;
        hh0 = pr0 << 16;
        hh1 = pr1 << 16;
        hh2 = pr2 << 16;
        hh3 = pr3 << 16;
        hg0 = pg0 << 16;
        hg1 = pg1 << 16;
        hg2 = pg2 << 16;
        hg3 = pg3 << 16;
    };
    const bool any_locate = __any(locate) != 0; // wave-uniform
    // iterations 0 .. nmax; with a deferred end cell in the wave the last two carry the search (band 0's last block,
    // then band 1's)
    for (uint32_t t = 0; t <= nmax; ++t) {
        if (nblk == 0u || t < lag)
            continue;
        if (any_locate && t + 1u == nmax)
            sweep(t, std::integral_constant<int, 1>{});
        else if (any_locate && t == nmax)
            sweep(t, std::integral_constant<int, 2>{});
        else
            sweep(t, std::integral_constant<int, 0>{});
    }

    if (!active)
        return;
    uint32_t len = 0;
    bool lost = false;
    if (work && locate) {
        lost = key == 0xFFFFFFFFu; // cannot happen: the packed pass saw M in this block
        eA = lost ? 0u : (key >> 2) + 1u;
        eB = lost ? 0u : eB - 3u + (key & 3u);
        endA[pair] = eA;
        endB[pair] = eB;
        if (lost)
            err[pair] = 0xFFFFFFFEu;
    } else if (locate) {
        endA[pair] = 0u;
        endB[pair] = 0u;
    }
    if (walkinfo) { // the walk is a kernel of its own (tb_walk16_kernel)
        walkinfo[pair - pair0] = (work && !lost) ? make_uint2(jb0, lag) : make_uint2((rowsA > 0 && lenA > RA) ? 1u : 0u, 0xFFFFFFFFu);
        return;
    }
    if (work && !lost && !(wide & 4)) { // (wide & 4: POLYHIP_TB_NOWALK=1, ablation probe -- what does the sweep cost on its own?)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my own stores, read back by me
        len = tbf_walk<RB, 1, true>(dirw, jb0, lag, eA, eB, (int)M, gap, ncp, reinterpret_cast<const uint16_t *>(P), codeL, ap, B,
                                    alnA + pair * stride, alnB + pair * stride, stride);
    }
    alnLen[pair] = (active && rowsA > 0 && lenA > RA) ? 0xFFFFFFFFu : len;
}

// ---- every pair with its own B (reads against reads) on packed halves ---------------------------------------------------
// tb_prof16_kernel's sweep, rows and walk; what changes is where a row finds its scores.  There is no table of the one
// reference's blocks: a lane builds the profile entries of ITS block -- for every symbol code c the four halves
// S(c, b_j .. b_j+3) -- from its own four B symbols when the block starts (24 lookups in a table of halves [code][byte] in
// LDS, six 8-byte stores into the lane's slot: ~60 of the block's ~3300 instructions) and keeps two slots: band 0 works on
// the new block, band 1 on the one before.  Slots are 56 bytes per lane (up to seven codes), so a wave's reads spread over
// the banks two lanes at a time.  The next block's four B symbols are loaded an iteration ahead.
template <int RB>
__global__ __launch_bounds__(THREADS, 2) void tb_pair16_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ Ball, const uint64_t *__restrict__ offB, const int8_t *__restrict__ lutc,
    const uint8_t *__restrict__ codeA, int ncodes, int gap, uint32_t *__restrict__ endA,
    uint32_t *__restrict__ endB, uint32_t *__restrict__ err, const int64_t *__restrict__ score, int smax,
    uint32_t wcols, int wide, uint32_t nblk_alloc, uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA,
    uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen, uint32_t stride)
{
    static_assert(RB % 4 == 0 && RB <= 76, "RB");
    constexpr uint32_t SLOT = 56; // bytes per lane and slot: seven codes (ncodes + the pad code <= 7)
    uint2 *const walkinfo = nullptr;
    constexpr int RA = 2 * RB;
    constexpr int NG = (RB + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tbf[];
    const int ncp = ncodes + 1;
    // LDS: two slots per lane, the table of halves H16[code][byte] (the pad code's row: -128 everywhere), the A codes
    uint8_t *slots = lds_tbf;
    uint16_t *H16 = reinterpret_cast<uint16_t *>(lds_tbf + 2 * (size_t)THREADS * SLOT);
    uint8_t *codeL = reinterpret_cast<uint8_t *>(H16 + (size_t)ncp * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t v = tid; v < (uint32_t)ncp * 256u; v += THREADS)
        H16[v] = (uint16_t)tbf_half_bits(v < (uint32_t)ncodes * 256u ? (int)lutc[v] : -128);
    codeL[tid] = codeA[tid];
    {
        const uint32_t padh = tbf_half_bits(-128), pad2 = padh | (padh << 16);
        for (int q = 0; q < 2; ++q)
            for (int c = 0; c < 7; ++c)
                *reinterpret_cast<uint2 *>(slots + ((size_t)q * THREADS + tid) * SLOT + c * 8) = make_uint2(pad2, pad2);
    }
    __syncthreads();

    const uint64_t pair = pair0 + (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < pair1;
    uint32_t lenA = 0, eA = 0, eB = 0;
    int64_t M = 0;
    const uint8_t *ap = A, *B = Ball;
    uint64_t oA = 0;
    uint32_t lenB = 0;
    if (active) {
        const uint64_t o0 = offA[pair];
        oA = o0;
        lenA = (uint32_t)(offA[pair + 1] - o0);
        ap = A + o0;
        const uint64_t ob = offB[pair];
        B = Ball + ob;
        lenB = (uint32_t)(offB[pair + 1] - ob);
        if (err[pair] == 0u) {
            eA = endA[pair];
            eB = endB[pair];
            M = score[pair];
        }
    }
    const bool locate = active && (wide & 2) && eA == k3p::SW_END_DEFERRED; // as tb_prof_kernel
    const uint32_t rowsA = locate ? lenA : eA;
    const bool work = active && rowsA > 0 && rowsA <= lenA && eB > 0 && eB <= lenB && M > 0 && lenA <= RA;
    const uint32_t mycols = work ? min(wcols + 4u, pair_window(wcols, rowsA, M, smax, gap, wide & 1) + (locate ? 4u : 0u)) : 0u;
    const uint32_t c_s = (work && eB > mycols) ? eB - mycols + 1u : 1u;
    const uint32_t jb0 = (c_s - 1u) & ~3u;
    const uint32_t nblk = work ? (eB - jb0 + TBU - 1) / TBU : 0u; // <= nblk_alloc - 1

    // byte offsets (code * 8) of my rows inside a profile block, four rows per register and band.  The read's bytes come as
    // RA / 4 + 1 aligned dwords, all in flight at once, through a buffer resource over this launch's reads (as
    // sw_pk1_kernel does; a byte at a time the prologue was RA dependent round trips); 4 GB and more keep the byte loads.
    uint32_t apk0[RB / 4], apk1[RB / 4];
    const uint64_t totalA = offA[pair1];
    if (totalA < 0xFFFFFFF0ull) {
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 3u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A) - misA, 0,
                                                                            (int)(((uint32_t)totalA + misA + 3u) & ~3u), 0x00020000);
        const uint32_t b0 = (uint32_t)oA + misA;
        uint32_t dd[RA / 4];
        {
            uint32_t aw[RA / 4 + 1];
#pragma unroll
            for (int w = 0; w <= RA / 4; ++w)
                aw[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b0 & ~3u) + 4u * w), 0, 0);
#pragma unroll
            for (int w = 0; w < RA / 4; ++w)
                dd[w] = __builtin_amdgcn_alignbyte(aw[w + 1], aw[w], b0 & 3u);
        }
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i0 = 4 * w + b, i1 = RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && (uint32_t)i0 < lenA) {
                    const uint32_t c = codeL[(dd[w] >> (8 * b)) & 0xFFu];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && (uint32_t)i1 < lenA) {
                    const uint32_t c = codeL[(dd[RB / 4 + w] >> (8 * b)) & 0xFFu];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    } else {
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i0 = 4 * w + b, i1 = RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && (uint32_t)i0 < lenA) {
                    const uint32_t c = codeL[ap[i0]];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && (uint32_t)i1 < lenA) {
                    const uint32_t c = codeL[ap[i1]];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    }

    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;

    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    static_assert(TBU == 4, "a lane's piece of direction words is (G, L) x four columns");
    uint32_t *dirw = dirbuf + wave_global * ((size_t)nblk_alloc * TBU * NG * 2 * 64) + lane * 8;
    const uint32_t slot_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(slots)) + (uint32_t)tid * SLOT;
    // the four B symbols of a block, as the bytes of one register (0xFF = past the end of B: the pad score); block 0 now,
    // the others an iteration ahead
    auto block_syms = [&](uint32_t bt) -> uint32_t {
        uint32_t w = 0xFFFFFFFFu;
        if (bt < nblk) {
            const uint32_t j = jb0 + 4u * bt;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + u < lenB)
                    w = (w & ~(0xFFu << (8 * u))) | ((uint32_t)B[j + u] << (8 * u));
        }
        return w;
    };
    uint32_t bnext = block_syms(0);

    // end-aligned lanes as in tb_prof_kernel; a lane runs nblk + 1 iterations (band 1 finishes one block later)
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    const uint32_t gh = tbf_half_bits(gap); // gap < 0
    const uint32_t gap2 = gh | (gh << 16), one2 = 0x00010001u, two2 = 0x00020002u;
    const uint32_t Mh = tbf_half_bits((int)M);
    uint32_t key = 0xFFFFFFFFu;
    // band 0's last row of the block before, already in the high halves: what band 1 finds above its first row
    uint32_t hh0 = 0, hh1 = 0, hh2 = 0, hh3 = 0, hg0 = 0, hg1 = 0, hg2 = 0, hg3 = 0, hd = 0;
    auto sweep = [&](uint32_t tt, auto find_tag) {
        constexpr int FIND = decltype(find_tag)::value; // 0, 1 = search band 0's cells, 2 = band 1's
        const uint32_t bt = tt - lag;                   // band 0's block; band 1 works on bt - 1
        // my slot of this iteration takes block bt's entries; the other one still holds block bt - 1's (pad entries before
        // the first block: the slots start that way)
        const uint32_t base0 = slot_base + (tt & 1u) * (THREADS * SLOT), base1 = slot_base + ((tt & 1u) ^ 1u) * (THREADS * SLOT);
        {
            const uint32_t bw = bnext;
            bnext = block_syms(bt + 1u);
            const uint32_t s0 = bw & 0xFFu, s1 = (bw >> 8) & 0xFFu, s2 = (bw >> 16) & 0xFFu, s3 = bw >> 24;
            const uint32_t padh = tbf_half_bits(-128);
            for (int c = 0; c < ncp; ++c) {
                const uint16_t *row = H16 + c * 256;
                const uint32_t h0 = s0 != 0xFFu ? row[s0] : padh, h1 = s1 != 0xFFu ? row[s1] : padh;
                const uint32_t h2 = s2 != 0xFFu ? row[s2] : padh, h3 = s3 != 0xFFu ? row[s3] : padh;
                const uint64_t e64 = (uint64_t)(h0 | (h1 << 16)) | ((uint64_t)(h2 | (h3 << 16)) << 32);
                asm volatile("ds_write_b64 %0, %1" ::"v"(base0 + (uint32_t)c * 8u), "v"(e64) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        uint32_t pr0 = hh0, pr1 = hh1, pr2 = hh2, pr3 = hh3, pg0 = hg0, pg1 = hg1, pg2 = hg2, pg3 = hg3, pdiag = hd;
        uint32_t wG[TBU] = {0u, 0u, 0u, 0u}, wL[TBU] = {0u, 0u, 0u, 0u};
        tbf_u32x2 xa[4], ya[4], xb[4], yb[4];
        PH_TBF_ISSUE(apk0[0], apk1[0], xa, ya);
#pragma unroll
        for (int g = 0; g < RB / 4; ++g) {
            if (g + 1 < RB / 4) {
                PH_TBF_ISSUE(apk0[g + 1], apk1[g + 1], xb, yb);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            }
            PH_TBF_ROW(4 * g + 0, xa[0].x, xa[0].y, ya[0].x, ya[0].y);
            PH_TBF_ROW(4 * g + 1, xa[1].x, xa[1].y, ya[1].x, ya[1].y);
            PH_TBF_ROW(4 * g + 2, xa[2].x, xa[2].y, ya[2].x, ya[2].y);
            PH_TBF_ROW(4 * g + 3, xa[3].x, xa[3].y, ya[3].x, ya[3].y);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xa[q] = xb[q];
                ya[q] = yb[q];
            }
        }
        // band 0's last row (low halves) becomes band 1's row above (high halves) of the next iteration
        hd = hh3;
        hh0 = pr0 << 16;
        hh1 = pr1 << 16;
        hh2 = pr2 << 16;
        hh3 = pr3 << 16;
        hg0 = pg0 << 16;
        hg1 = pg1 << 16;
        hg2 = pg2 << 16;
        hg3 = pg3 << 16;
    };
    const bool any_locate = __any(locate) != 0; // wave-uniform
    // iterations 0 .. nmax; with a deferred end cell in the wave the last two carry the search (band 0's last block,
    // then band 1's)
    for (uint32_t t = 0; t <= nmax; ++t) {
        if (nblk == 0u || t < lag)
            continue;
        if (any_locate && t + 1u == nmax)
            sweep(t, std::integral_constant<int, 1>{});
        else if (any_locate && t == nmax)
            sweep(t, std::integral_constant<int, 2>{});
        else
            sweep(t, std::integral_constant<int, 0>{});
    }

    if (!active)
        return;
    uint32_t len = 0;
    bool lost = false;
    if (work && locate) {
        lost = key == 0xFFFFFFFFu; // cannot happen: the packed pass saw M in this block
        eA = lost ? 0u : (key >> 2) + 1u;
        eB = lost ? 0u : eB - 3u + (key & 3u);
        endA[pair] = eA;
        endB[pair] = eB;
        if (lost)
            err[pair] = 0xFFFFFFFEu;
    } else if (locate) {
        endA[pair] = 0u;
        endB[pair] = 0u;
    }
    if (walkinfo) { // the walk is a kernel of its own (tb_walk16_kernel)
        walkinfo[pair - pair0] = (work && !lost) ? make_uint2(jb0, lag) : make_uint2((rowsA > 0 && lenA > RA) ? 1u : 0u, 0xFFFFFFFFu);
        return;
    }
    if (work && !lost && !(wide & 4)) { // (wide & 4: POLYHIP_TB_NOWALK=1, ablation probe -- what does the sweep cost on its own?)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my own stores, read back by me
        len = tbf_walk<RB, 1, true, true>(dirw, jb0, lag, eA, eB, (int)M, gap, ncp, H16, codeL, ap, B, alnA + pair * stride,
                                          alnB + pair * stride, stride);
    }
    alnLen[pair] = (active && rowsA > 0 && lenA > RA) ? 0xFFFFFFFFu : len;
}

// ---- 153..256 rows on packed halves: TWO LANES per pair, four bands of RB rows ---------------------------------------
// One lane cannot hold more than two bands of 76 rows at two waves per SIMD (tb_prof16_kernel<128> spills 357 registers
// into its sweep), and one wave per pair (tb_wave_kernel<4>) pays a lane shift per cell.  So a pair takes the lanes 2p
// and 2p + 1 of its wave: rows [0, RB) / [RB, 2 RB) in the low / high halves of lane 2p, rows [2 RB, 3 RB) / [3 RB, 4 RB)
// in those of lane 2p + 1, band b one 4-column block behind band b - 1 -- the same skew as between the two bands of one
// lane, continued across the lane boundary: what band 1 leaves behind (last row: four values, their gap-decayed copies,
// one diagonal value) moves from the high halves of lane 2p to the low halves of lane 2p + 1 by one DPP row_shr:1 per
// register and block (27 instructions against the ~2750 of a block's 64 row pairs).  A lane runs nblk + 3 iterations.
// Row macros, direction-word layout (per lane) and the walk's arithmetic are tb_prof16_kernel's; the walk is the even
// lane's, which reads its partner's words too.  A deferred end cell is searched in the wave's last four iterations (band b
// sweeps the pair's last block in iteration nmax - 1 + b; the even lane's rows come first in row-major order).
template <int RB>
__global__ __launch_bounds__(THREADS, 2) void tb_prof16x2_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ B, uint32_t lenB_pad, const uint2 *__restrict__ prof16,
    const uint8_t *__restrict__ codeA, int ncodes, int gap, uint32_t *__restrict__ endA,
    uint32_t *__restrict__ endB, uint32_t *__restrict__ err, const int64_t *__restrict__ score, int smax,
    uint32_t wcols, int wide, uint32_t nblk_alloc, uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA,
    uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen, uint32_t stride, uint2 *__restrict__ walkinfo)
{
    static_assert(RB % 16 == 0 && RB <= 64, "RB");
    constexpr int RL = 2 * RB; // rows per lane
    constexpr int RA = 4 * RB; // rows per pair
    constexpr int NG = RB / 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tbf[];
    const int ncp = ncodes + 1;
    const uint32_t nqB = lenB_pad / 4; // real blocks; block nqB is the all-pad one
    const uint32_t pstride = (uint32_t)ncp * 8u;
    uint2 *P = reinterpret_cast<uint2 *>(lds_tbf);
    uint8_t *codeL = lds_tbf + (size_t)(nqB + 1) * pstride;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t hl = (uint32_t)lane & 1u; // which half of the pair's rows this lane holds
    for (uint32_t v = tid; v < (nqB + 1) * (uint32_t)ncp; v += THREADS)
        P[v] = prof16[v];
    codeL[tid] = codeA[tid];
    __syncthreads();

    const uint64_t pair = pair0 + (((uint64_t)blockIdx.x * THREADS + tid) >> 1);
    const bool active = pair < pair1;
    uint32_t lenA = 0, eA = 0, eB = 0;
    int64_t M = 0;
    const uint8_t *ap = A;
    uint64_t oA = 0;
    if (active) {
        const uint64_t o0 = offA[pair];
        oA = o0;
        lenA = (uint32_t)(offA[pair + 1] - o0);
        ap = A + o0;
        if (err[pair] == 0u) {
            eA = endA[pair];
            eB = endB[pair];
            M = score[pair];
        }
    }
    // everything below is the same in both lanes of a pair (its rows apart)
    const bool locate = active && (wide & 2) && eA == k3p::SW_END_DEFERRED; // as tb_prof_kernel
    const uint32_t rowsA = locate ? lenA : eA;
    const bool work = active && rowsA > 0 && rowsA <= lenA && eB > 0 && M > 0 && lenA <= RA;
    const uint32_t mycols = work ? min(wcols + 4u, pair_window(wcols, rowsA, M, smax, gap, wide & 1) + (locate ? 4u : 0u)) : 0u;
    const uint32_t c_s = (work && eB > mycols) ? eB - mycols + 1u : 1u;
    const uint32_t jb0 = (c_s - 1u) & ~3u;
    const uint32_t nblk = work ? (eB - jb0 + TBU - 1) / TBU : 0u; // <= nblk_alloc - 3

    // byte offsets (code * 8) of my RL rows inside a profile block, four rows per register and band (as tb_prof16_kernel)
    uint32_t apk0[RB / 4], apk1[RB / 4];
    const uint32_t row0 = hl * RL;
    const uint64_t totalA = offA[pair1];
    if (totalA < 0xFFFFFFF0ull) {
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 3u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A) - misA, 0,
                                                                            (int)(((uint32_t)totalA + misA + 3u) & ~3u), 0x00020000);
        const uint32_t b0 = (uint32_t)oA + misA + row0; // beyond the batch's last byte a buffer load returns 0: masked below
        uint32_t dd[RL / 4];
        {
            uint32_t aw[RL / 4 + 1];
#pragma unroll
            for (int w = 0; w <= RL / 4; ++w)
                aw[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b0 & ~3u) + 4u * w), 0, 0);
#pragma unroll
            for (int w = 0; w < RL / 4; ++w)
                dd[w] = __builtin_amdgcn_alignbyte(aw[w + 1], aw[w], b0 & 3u);
        }
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t i0 = row0 + 4 * w + b, i1 = row0 + RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && i0 < lenA) {
                    const uint32_t c = codeL[(dd[w] >> (8 * b)) & 0xFFu];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && i1 < lenA) {
                    const uint32_t c = codeL[(dd[RB / 4 + w] >> (8 * b)) & 0xFFu];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    } else {
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t k0 = 0, k1 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t i0 = row0 + 4 * w + b, i1 = row0 + RB + 4 * w + b;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes;
                if (work && i0 < lenA) {
                    const uint32_t c = codeL[ap[i0]];
                    c0 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                if (work && i1 < lenA) {
                    const uint32_t c = codeL[ap[i1]];
                    c1 = c == 0xFFu ? (uint32_t)ncodes : c;
                }
                k0 |= (c0 * 8u) << (8 * b);
                k1 |= (c1 * 8u) << (8 * b);
            }
            apk0[w] = k0;
            apk1[w] = k1;
        }
    }

    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;

    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    uint32_t *dirw = dirbuf + wave_global * ((size_t)nblk_alloc * TBU * NG * 2 * 64) + lane * 8;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P));
    const uint32_t pad_base = lds_base + nqB * pstride;

    // end-aligned lanes as in tb_prof_kernel; a lane runs nblk + 3 iterations (band b finishes b blocks later)
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    const uint32_t gh = tbf_half_bits(gap); // gap < 0
    const uint32_t gap2 = gh | (gh << 16), one2 = 0x00010001u, two2 = 0x00020002u;
    const uint32_t Mh_pair = tbf_half_bits((int)M);
    uint32_t key = 0xFFFFFFFFu;
    const uint32_t oddmask = hl ? 0xFFFFFFFFu : 0u;
    // what the band above my first band left behind for the block it finished one iteration ago, and my first band's own
    // last row already in the high halves: what my two bands find above their first rows
    uint32_t hh0 = 0, hh1 = 0, hh2 = 0, hh3 = 0, hg0 = 0, hg1 = 0, hg2 = 0, hg3 = 0, hd = 0;
    // find_tag: 0, 1 = search my first band's cells for the deferred end cell, 2 = my second band's; Mh = the maximum as a
    // half, or a pattern no H has (0xFFFF) in the lanes whose turn it is not
    auto sweep = [&](uint32_t tt, auto find_tag, const uint32_t Mh) __attribute__((always_inline)) {
        constexpr int FIND = decltype(find_tag)::value;
        const uint32_t bt = tt - lag - 2u * hl;         // my first band's block (as an unsigned number: "negative" = none yet)
        const uint32_t base0 = bt < nblk ? lds_base + ((jb0 >> 2) + bt) * pstride : pad_base;
        const uint32_t base1 = bt - 1u < nblk ? lds_base + ((jb0 >> 2) + bt - 1u) * pstride : pad_base;
        uint32_t pr0 = hh0, pr1 = hh1, pr2 = hh2, pr3 = hh3, pg0 = hg0, pg1 = hg1, pg2 = hg2, pg3 = hg3, pdiag = hd;
        uint32_t wG[TBU] = {0u, 0u, 0u, 0u}, wL[TBU] = {0u, 0u, 0u, 0u};
        tbf_u32x2 xa[4], ya[4], xb[4], yb[4];
        PH_TBF_ISSUE(apk0[0], apk1[0], xa, ya);
#pragma unroll
        for (int g = 0; g < RB / 4; ++g) {
            if (g + 1 < RB / 4) {
                PH_TBF_ISSUE(apk0[g + 1], apk1[g + 1], xb, yb);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            }
            PH_TBF_ROW(4 * g + 0, xa[0].x, xa[0].y, ya[0].x, ya[0].y);
            PH_TBF_ROW(4 * g + 1, xa[1].x, xa[1].y, ya[1].x, ya[1].y);
            PH_TBF_ROW(4 * g + 2, xa[2].x, xa[2].y, ya[2].x, ya[2].y);
            PH_TBF_ROW(4 * g + 3, xa[3].x, xa[3].y, ya[3].x, ya[3].y);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xa[q] = xb[q];
                ya[q] = yb[q];
            }
        }
        // My first band's last row (low halves) becomes my second band's row above (high halves) of the next iteration; the
        // low halves of an odd lane take what the even lane's second band (its high halves) has just finished -- the row
        // above this lane's first band in the next iteration.  Even lanes: zeros (row 0 has H = 0 above it).
        auto pass = [&](uint32_t mine) {
            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111 /* row_shr:1 */, 0xF, 0xF, true) & oddmask;
            return __builtin_amdgcn_alignbit(mine, up, 16); // low half = the partner's high half, high half = my low half
        };
        hd = hh3;
        hh0 = pass(pr0);
        hh1 = pass(pr1);
        hh2 = pass(pr2);
        hh3 = pass(pr3);
        hg0 = pass(pg0);
        hg1 = pass(pg1);
        hg2 = pass(pg2);
        hg3 = pass(pg3);
    };
    const bool any_locate = __any(locate) != 0; // wave-uniform
    const uint32_t Mh_even = hl ? 0xFFFFu : Mh_pair, Mh_odd = hl ? Mh_pair : 0xFFFFu;
    // iterations 0 .. nmax + 2 (nblk is the same in both lanes of a pair, so a lane and the partner it reads are in step);
    // with a deferred end cell in the wave the last four carry the search: band b sweeps the pair's last block in
    // iteration nmax - 1 + b
    for (uint32_t t = 0; t <= nmax + 2u; ++t) {
        if (nblk == 0u || t < lag)
            continue;
        // (one call site per instantiation: with two, hipcc stops inlining the lambda and every array it captures -- H, the
        // row codes -- moves to scratch: 99 ms instead of 17 for 400k reads of 250 bp)
        if (any_locate && (t + 1u == nmax || t == nmax + 1u))
            sweep(t, std::integral_constant<int, 1>{}, t + 1u == nmax ? Mh_even : Mh_odd);
        else if (any_locate && (t == nmax || t == nmax + 2u))
            sweep(t, std::integral_constant<int, 2>{}, t == nmax ? Mh_even : Mh_odd);
        else
            sweep(t, std::integral_constant<int, 0>{}, 0xFFFFu);
    }
    // row-major order: the even lane's rows come first
    const uint32_t key_odd = (uint32_t)__shfl_down((int)key, 1, 64);
    if (key == 0xFFFFFFFFu && key_odd != 0xFFFFFFFFu)
        key = key_odd + ((uint32_t)RL << 2);

    if (!active || hl != 0u)
        return;
    uint32_t len = 0;
    bool lost = false;
    if (work && locate) {
        lost = key == 0xFFFFFFFFu; // cannot happen: the packed pass saw M in this block
        eA = lost ? 0u : (key >> 2) + 1u;
        eB = lost ? 0u : eB - 3u + (key & 3u);
        endA[pair] = eA;
        endB[pair] = eB;
        if (lost)
            err[pair] = 0xFFFFFFFEu;
    } else if (locate) {
        endA[pair] = 0u;
        endB[pair] = 0u;
    }
    if (walkinfo) { // the walk is a kernel of its own (tb_walk16_kernel)
        walkinfo[pair - pair0] = (work && !lost) ? make_uint2(jb0, lag) : make_uint2((rowsA > 0 && lenA > RA) ? 1u : 0u, 0xFFFFFFFFu);
        return;
    }
    if (work && !lost && !(wide & 4)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // this wave's own stores (my partner's included), read back by me
        len = tbf_walk<RB, 2, true>(dirw, jb0, lag, eA, eB, (int)M, gap, ncp, reinterpret_cast<const uint16_t *>(P), codeL, ap, B,
                                    alnA + pair * stride, alnB + pair * stride, stride);
    }
    alnLen[pair] = (rowsA > 0 && lenA > RA) ? 0xFFFFFFFFu : len;
}
#undef PH_TBF_ISSUE
#undef PH_TBF_ADDR
#undef PH_TBF_ROW
#undef PH_TBF_ROW2
#undef PH_TBF_CELL

// ---- reads longer than the 256 rows a lane can hold: ONE WAVE PER PAIR (like sw_wave.hip) ---------------
// Lane l owns rows [l*R, l*R + R) and works on column s - l of the pair's window in step s; the row above
// its first row and the column's B code arrive by one lane shift per step.  Same G / L bits as
// tb_prof_kernel, one word set per (step, lane) so a wave stores contiguously; the walk (all lanes in step,
// lane 0 writes) carries the running score down from the score pass's maximum.

// Direction bits of one (step, lane) with up to 16 rows per lane: G | L << BITS, a field of 2 * BITS bits; SP
// consecutive steps of a lane share one dword (4 rows per lane: four steps), so a lane stores a word every SP-th
// step instead of 8 useful bits in 32 every step.  Word (s / SP, lane), field s % SP.
template <int R>
struct WavePack {
    static constexpr int BITS = R <= 4 ? 4 : R <= 8 ? 8 : 16;
    static constexpr int FB = 2 * BITS;
    static constexpr int SP = 32 / FB;
};

// P8 (R = 8 or 16; round 5): the sweep on a BYTE PROFILE of the pair in LDS.  The wave writes, per B code b, the R bytes
// score(row, b) - gap of every lane's rows side by side (plane b: 64 lanes x R bytes), so a step costs a lane ONE ds_read
// (b64 / b128) for its whole column instead of a table lookup per cell, the byte goes into the add by operand selection, and
// with H + gap kept instead of H (left and up both arrive with the gap added; the profile carries - gap) the cell is
//     x = diag' + byte;  t = max(up', left');  G = t > x;  L = left' > up';  h = max3(x, t, 0);  h' = h + gap
// -- eight instructions against sixteen.  G compares with x, not with max(x, 0): the two differ only where h = 0, and the
// walk never reads the bits of such a cell (it stops at h = 0).  Lanes outside the window need no select: a lane that has
// not started sees pad codes (score 0) over zeros and stays at zero; what a lane computes past its last column is read by
// nobody (the lane below is one column behind).  Condition (host): smax - gap <= 127, smin - gap >= -128, the planes fit.
// LOC: the instantiation that can find a deferred end cell (the one-call form); without it the kernel keeps 71 registers
// instead of 95 (seven waves per SIMD instead of five: 39.2 against 40.6 ms per 80k x 1 kb tracebacks).
template <int R, bool P8 = false, bool LOC = false>
__global__ __launch_bounds__(THREADS) void tb_wave_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ Bbase, const uint64_t *__restrict__ offB, const uint8_t *__restrict__ codeA,
    const uint8_t *__restrict__ codeB, const int32_t *__restrict__ lutcc, int na, int nb, int gap,
    uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err,
    const int64_t *__restrict__ score, int smax, uint32_t wcols, int wide, uint32_t *__restrict__ dirbuf,
    uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen, uint32_t stride)
{
    constexpr int NG = (R + 31) / 32;            // 32-row groups per lane
    constexpr int NWL = R <= 16 ? 1 : 2 * NG;    // words per (step, lane): G | L << 16, or NG words of each
    extern __shared__ __attribute__((aligned(16))) int32_t T[]; // [na][nb] then codeA[256], codeB[256]
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();
    // P8: this wave's planes behind the tables, 16-byte aligned
    uint8_t *prof8 = reinterpret_cast<uint8_t *>(T) + ((((size_t)na * nb * 4 + 512) + 15) & ~(size_t)15) + (size_t)(tid >> 6) * ((size_t)nb * R * 64);
    const uint64_t wslot = (uint64_t)blockIdx.x * (THREADS / 64) + (tid >> 6);
    const uint64_t pair = pair0 + wslot;
    if (pair >= pair1)
        return; // wave-uniform, no barrier below
    const uint64_t o0 = offA[pair];
    const uint32_t lenA = (uint32_t)(offA[pair + 1] - o0);
    const uint8_t *ap = A + o0;
    const uint8_t *B = offB ? Bbase + offB[pair] : Bbase;
    uint32_t eA = 0, eB = 0;
    int64_t M = 0;
    if (err[pair] == 0u) {
        eA = endA[pair];
        eB = endB[pair];
        M = score[pair];
    }
    uint32_t len = 0;
    if (eA > 0 && eB > 0 && M > 0 && lenA <= 64u * R) {
        // The score pass may have left the end cell to this kernel (k3p::SW_END_DEFERRED, byte-profile form only): eB is then
        // the last column of the only 4-column block that holds the maximum, the window is sized for the whole read (the end
        // row is not known yet) and four columns longer, and the sweep notes the first cell worth M in row-major order
        const bool locate = P8 && LOC && (wide & 2) && eA == k3p::SW_END_DEFERRED;
        const uint32_t mycols = min(wcols + 4u, pair_window(wcols, locate ? lenA : eA, M, smax, gap, wide & 1) + (locate ? 4u : 0u));
        const uint32_t c_s = eB > mycols ? eB - mycols + 1u : 1u; // first column (1-based) of the window
        const uint32_t ncol = eB - c_s + 1u;
        uint32_t besti = 0xFFFFFFFFu, bestj = 0u; // locate: the smallest row with a cell worth M, its first column (window-relative)
        const int Mg = (int)M + gap;
        uint32_t ro[R];
        constexpr int NAB = R <= 16 ? (R + 3) / 4 : 1;
        uint32_t abytes[NAB]; // R <= 16: the lane's rows' symbols, for the walk
#pragma unroll
        for (int q = 0; q < NAB; ++q)
            abytes[q] = 0u;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t r = (uint32_t)lane * R + k;
            uint32_t code = (uint32_t)(na - 1);
            if (r < lenA) {
                const uint32_t sym = ap[r];
                const uint32_t c = cA[sym];
                code = c == 0xFFu ? (uint32_t)(na - 1) : c;
                if constexpr (R <= 16)
                    abytes[k >> 2] |= sym << (8 * (k & 3));
            }
            ro[k] = code * (uint32_t)nb;
        }
        uint32_t *dirw = dirbuf + wslot * ((size_t)(wcols + 67u) * 64 * NWL) + (size_t)lane * NWL;
        int Hrow[R];
#pragma unroll
        for (int k = 0; k < R; ++k)
            Hrow[k] = 0;
        int topprev = 0, last_h = 0;
        uint32_t last_b = (uint32_t)(nb - 1);
        auto load_chunk = [&](uint32_t s0) -> uint32_t { // B codes of window columns s0 + lane
            uint32_t c = (uint32_t)(nb - 1);
            if (s0 + (uint32_t)lane < ncol) {
                const uint32_t cc = cB[B[c_s - 1u + s0 + (uint32_t)lane]];
                c = cc == 0xFFu ? (uint32_t)(nb - 1) : cc;
            }
            return c;
        };
        uint32_t chunk = load_chunk(0), next_chunk = 0u;
        const uint32_t steps = ncol + 63u;
        uint32_t acc = 0u; // the word being filled (R <= 16)
        bool accany = false;
        if constexpr (P8) {
            static_assert(R == 8 || R == 16, "byte-profile sweep: 8 or 16 rows per lane");
            constexpr int BITS = WavePack<R>::BITS, FB = WavePack<R>::FB, SP = WavePack<R>::SP;
            constexpr int NQ = R / 4;
            // the planes: byte (b, lane, k) = score(row lane * R + k, b) - gap
            for (int b = 0; b < nb; ++b) {
                uint32_t w[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    w[q] = 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        w[q] |= ((uint32_t)(T[ro[q * 4 + k] + b] - gap) & 0xFFu) << (8 * k);
                }
                uint32_t *dst = reinterpret_cast<uint32_t *>(prof8 + ((size_t)b * 64 + lane) * R);
                if constexpr (R == 16)
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                else
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(w[0], w[1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int lg[R]; // H + gap of the lane's rows, previous column
#pragma unroll
            for (int k = 0; k < R; ++k)
                lg[k] = gap;
            int tprev = gap, lastg = gap;
            // (two steps per trip: the new H + gap of a row cannot overwrite the old one while the row below still needs it
            // as its diagonal, so a single-step loop ends in R register copies; steps + 1 when odd -- the extra step is
            // outside every lane's window and stores nothing)
            auto one = [&](uint32_t s) __attribute__((always_inline)) {
                int top_in = from_lane_below(lastg);
                uint32_t b_in = (uint32_t)from_lane_below((int)last_b);
                const uint32_t b_new = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)(s & 63u));
                if (lane == 0) {
                    top_in = gap;
                    b_in = b_new;
                }
                uint32_t pw[NQ];
                {
                    const uint8_t *src = prof8 + ((size_t)b_in * 64 + lane) * R;
                    if constexpr (R == 16) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(src);
                        pw[0] = v.x, pw[1] = v.y, pw[2] = v.z, pw[3] = v.w;
                    } else {
                        const uint2 v = *reinterpret_cast<const uint2 *>(src);
                        pw[0] = v.x, pw[1] = v.y;
                    }
                }
                const uint32_t jr = s - (uint32_t)lane; // wraps for lanes that have not started
                const bool valid = jr < ncol;
                int diag = tprev, up = top_in;
                uint32_t gw = 0u, lw = 0u;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int left = lg[k];
                    const int x = diag + (int)(int8_t)(pw[k >> 2] >> (8 * (k & 3)));
                    const int t = max(up, left);
#ifdef PH_TB_BITS_ALIGN // (ablation: the bit through v_sub + v_alignbit instead of v_cmp + v_addc -- no VCC between the two)
                    gw = __builtin_amdgcn_alignbit(gw, (uint32_t)(x - t), 31);
                    lw = __builtin_amdgcn_alignbit(lw, (uint32_t)(up - left), 31);
#else
                    PH_CARRY_BIT(gw, t, x);
                    PH_CARRY_BIT(lw, left, up);
#endif
                    const int hg = max(max(x, t), 0) + gap;
                    diag = left;
                    up = hg;
                    lg[k] = hg;
                }
                tprev = top_in;
                lastg = lg[R - 1];
                last_b = b_in;
                if (locate) { // (wave-uniform) did a cell of this column, in some lane, reach M?  Rarely: then look at the rows
                    int m = lg[0];
#pragma unroll
                    for (int k = 1; k < R; ++k)
                        m = max(m, lg[k]);
                    if (__any(valid && m == Mg)) {
                        if (valid) {
#pragma unroll
                            for (int k = 0; k < R; ++k) {
                                const uint32_t r = (uint32_t)lane * R + k;
                                if (lg[k] == Mg && r < lenA && r < besti) { // smaller row wins; columns come in order
                                    besti = r;
                                    bestj = jr;
                                }
                            }
                        }
                    }
                }
                const uint32_t sub = s % SP;
                if (valid) {
                    acc |= (gw | (lw << BITS)) << (sub * FB);
                    accany = true;
                }
                if (sub == SP - 1) {
                    if (accany)
                        dirw[(size_t)(s / SP) * 64 * NWL] = acc;
                    acc = 0u;
                    accany = false;
                }
            };
            const uint32_t steps2 = (steps + 1u) & ~1u;
            for (uint32_t s = 0; s < steps2; s += 2) {
                if ((s & 63u) == 0u) {
                    if (s)
                        chunk = next_chunk;
                    next_chunk = load_chunk(s + 64u);
                }
                one(s);
                one(s + 1u);
            }
        } else
        for (uint32_t s = 0; s < steps; ++s) {
            if ((s & 63u) == 0u) {
                if (s)
                    chunk = next_chunk;
                next_chunk = load_chunk(s + 64u);
            }
            int top_in = from_lane_below(last_h);
            uint32_t b_in = (uint32_t)from_lane_below((int)last_b);
            const uint32_t b_new = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)(s & 63u));
            if (lane == 0) {
                top_in = 0;
                b_in = b_new;
            }
            const uint32_t jr = s - (uint32_t)lane; // wraps for lanes that have not started
            const bool valid = jr < ncol;
            int diag = topprev, up = top_in;
            uint32_t gw[NG], lw[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                gw[g] = lw[g] = 0u;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int left = Hrow[k];
                const int sc = T[ro[k] + b_in];
                const int d0 = max(diag + sc, 0);
                const int t = max(up, left) + gap;
                int h = max(d0, t);
                PH_CARRY_BIT(gw[k >> 5], t, d0);
                PH_CARRY_BIT(lw[k >> 5], left, up);
                h = valid ? h : 0;
                diag = left;
                up = h;
                Hrow[k] = h;
            }
            topprev = valid ? top_in : 0;
            last_h = Hrow[R - 1];
            last_b = b_in;
            if constexpr (R <= 16) {
                constexpr int BITS = WavePack<R>::BITS, FB = WavePack<R>::FB, SP = WavePack<R>::SP;
                const uint32_t sub = s % SP;
                if (valid) {
                    acc |= (gw[0] | (lw[0] << BITS)) << (sub * FB);
                    accany = true;
                }
                if (sub == SP - 1 || s == steps - 1) {
                    if (accany)
                        dirw[(size_t)(s / SP) * 64 * NWL] = acc;
                    acc = 0u;
                    accany = false;
                }
            } else if (valid) {
                uint32_t *o = dirw + (size_t)s * 64 * NWL;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    o[g] = gw[g];
                    o[NG + g] = lw[g];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (locate) { // the smallest row over the lanes (rows are distinct across lanes), its first column
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t oi = (uint32_t)__shfl_xor((int)besti, d, 64), oj = (uint32_t)__shfl_xor((int)bestj, d, 64);
                if (oi < besti) {
                    besti = oi;
                    bestj = oj;
                }
            }
            const bool found = besti != 0xFFFFFFFFu; // (cannot fail: the packed pass saw M in this block)
            eA = found ? besti + 1u : 0u;
            eB = found ? c_s + bestj : 0u;
            if (lane == 0) {
                endA[pair] = eA;
                endB[pair] = eB;
                if (!found)
                    err[pair] = 0xFFFFFFFEu;
            }
        }
        // ---- walk (uniform over the wave; lane 0 writes)
        uint8_t *outA = alnA + pair * stride, *outB = alnB + pair * stride;
        uint32_t i = eA, j = eB;
        int h = (int)M;
        const uint32_t *dbase = dirbuf + wslot * ((size_t)(wcols + 67u) * 64 * NWL);
        if (wide & 4) { // POLYHIP_TB_NOWALK=1: ablation probe -- what does the sweep cost on its own?
        } else if (R <= 16 && !(wide & 8)) {
            // Round 5: the walk out of the wave's REGISTERS, steered by the scalar unit.  A step of the plain walk below is
            // one dependent HBM round trip (272 KB of direction words per 1 kb pair: nothing of it stays in a cache) and
            // ~60 vector instructions that every lane executes for lane 0's two bytes.  Here the wave fetches, in ONE
            // round trip, the words of the next 32 steps of the lane row the walk is in and of the one above (lane q: word
            // wtop[q >> 5] - (q & 31) of lane row lc - (q >> 5)); a step reads its word with v_readlane and only a step that
            // leaves the window fetches again (a diagonal leaves it after ~30 steps).  The B symbols (with their codes) come
            // the same way, 64 at a time; the A symbols sit in the registers of the lane that owns the row.  Position,
            // score and windows live in SGPRs (every value that comes back from the vector side goes through
            // v_readfirstlane), so a step is ~40 scalar instructions beside the sweeps of the other waves, not 4 x 60
            // cycles of their SIMD.  Lane len % 64 keeps a step's two bytes; 64 steps go out as one store per string (a store
            // per step would put a wait for its completion in front of every step: the fetches share the counter).
            constexpr int SP = WavePack<R>::SP, FB = WavePack<R>::FB, BITS = WavePack<R>::BITS;
#define PH_SGPR(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
            uint32_t wi_ = PH_SGPR(i), wj_ = PH_SGPR(j), wlen = 0u;
            int wh_ = (int)PH_SGPR(h);
            const uint32_t cs_ = PH_SGPR(c_s), stride_ = PH_SGPR(stride), gap_ = PH_SGPR(gap), nb_ = PH_SGPR(nb);
            const uint32_t padA = PH_SGPR(na - 1), padB = PH_SGPR(nb - 1);
            const uint32_t m_ = (uint32_t)lane >> 5, x_ = (uint32_t)lane & 31u;
            uint32_t lc = 0xFFFFFFF0u;        // lane row of window 0 (none yet)
            uint32_t wtop0 = 0u, wtop1 = 0u;  // newest word of the two windows
            uint32_t cw = 0u;                 // my word of them
            uint32_t btop = 0xFFFFFFFFu;      // lane q: B[btop - q] | code << 8 (none yet: jb > btop never holds, btop - jb > 63 does)
            uint32_t bwin = 0u;
            uint32_t mya = 0u, myb = 0u;      // lane q: the bytes of step (wlen & ~63) + q
            while (wh_ > 0 && wi_ > 0u && wj_ >= cs_ && wlen < stride_) {
                const uint32_t r = wi_ - 1u, l = r / R, k = r % R;
                const uint32_t s = (wj_ - cs_) + l, wi = s / SP;
                const uint32_t m = lc - l;
                const uint32_t top = m == 0u ? wtop0 : wtop1;
                if (!(m <= 1u && wi <= top && top - wi < 32u)) { // fetch: window 0 from here, window 1 one column less
                    lc = l;
                    wtop0 = wi;
                    wtop1 = s ? (s - 1u) / SP : 0u;
                    const uint32_t wt = m_ ? wtop1 : wtop0;
                    // (no branch on a lane's own condition here: with one, the compiler merges it with the `continue` and takes
                    // the whole loop for divergent -- every scalar below becomes a vector instruction under an exec mask)
                    const bool ok = (l >= m_) & (x_ <= wt) & ((m_ == 0u) | (s > 0u));
                    cw = __hip_atomic_load(dbase + (ok ? ((size_t)(wt - x_) * 64 + (l - m_)) * NWL : (size_t)0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cw = ok ? cw : 0u;
                    continue;
                }
                const uint32_t jb = wj_ - 1u;
                if (jb > btop || btop - jb > 63u) {
                    btop = jb;
                    const bool okb = (uint32_t)lane <= jb;
                    const uint32_t sym = B[okb ? jb - (uint32_t)lane : 0u];
                    const uint32_t kb = cB[sym];
                    bwin = okb ? sym | ((kb == 0xFFu ? padB : kb) << 8) : 0u;
                    continue;
                }
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(m * 32u + (top - wi))) >> ((s % SP) * FB);
                const uint32_t gbit = (w >> (R - 1 - k)) & 1u;
                const uint32_t lbit = (w >> (BITS + R - 1 - k)) & 1u;
                uint32_t aw = abytes[0];
#pragma unroll
                for (int q = 1; q < NAB; ++q)
                    aw = (k >> 2) == (uint32_t)q ? abytes[q] : aw;
                const uint32_t sa = ((uint32_t)__builtin_amdgcn_readlane((int)aw, (int)l) >> (8u * (k & 3u))) & 0xFFu;
                const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)bwin, (int)(btop - jb));
                const uint32_t sb = bw & 0xFFu;
                uint32_t ca, cb;
                if (gbit == 0u) { // align.go:215-219
                    if constexpr (P8) { // the byte profile has the score of (row, code) in one read: score - gap
                        wh_ -= (int)(int8_t)PH_SGPR(prof8[(((bw >> 8) * 64u + l) * R) + k]) + (int)gap_;
                    } else {
                        const uint32_t ka = PH_SGPR(cA[sa]);
                        wh_ -= (int)PH_SGPR(T[(ka == 0xFFu ? padA : ka) * nb_ + (bw >> 8)]);
                    }
                    ca = sa;
                    cb = sb;
                    --wi_;
                    --wj_;
                } else if (lbit == 0u) { // :220-223
                    wh_ -= (int)gap_;
                    ca = sa;
                    cb = '-';
                    --wi_;
                } else { // :224-227
                    wh_ -= (int)gap_;
                    ca = '-';
                    cb = sb;
                    --wj_;
                }
                if ((uint32_t)lane == (wlen & 63u)) {
                    mya = ca;
                    myb = cb;
                }
                ++wlen;
                if ((wlen & 63u) == 0u) {
                    const uint32_t pos = wlen - 64u + (uint32_t)lane;
                    outA[stride - 1 - pos] = (uint8_t)mya;
                    outB[stride - 1 - pos] = (uint8_t)myb;
                }
            }
#undef PH_SGPR
            if ((uint32_t)lane < (wlen & 63u)) {
                const uint32_t pos = (wlen & ~63u) + (uint32_t)lane;
                outA[stride - 1 - pos] = (uint8_t)mya;
                outB[stride - 1 - pos] = (uint8_t)myb;
            }
            len = wlen;
        } else
        while (h > 0 && i > 0 && j >= c_s && len < stride) {
            const uint32_t r = i - 1u, l = r / R, k = r % R;
            const uint32_t s = (j - c_s) + l;
            const uint32_t *wp = dbase + ((size_t)(R <= 16 ? s / WavePack<R>::SP : s) * 64 + l) * NWL;
            uint32_t gbit, lbit;
            if (R <= 16) {
                const uint32_t w = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                                   ((s % WavePack<R>::SP) * WavePack<R>::FB);
                gbit = (w >> (R - 1 - k)) & 1u;
                lbit = (w >> (WavePack<R>::BITS + R - 1 - k)) & 1u;
            } else {
                const uint32_t g = k >> 5, bit = 31u - (k & 31u);
                gbit = (__hip_atomic_load(wp + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> bit) & 1u;
                lbit = (__hip_atomic_load(wp + NG + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> bit) & 1u;
            }
            const uint8_t sa = ap[r], sb = B[j - 1u];
            uint8_t ca, cb;
            if (gbit == 0u) { // align.go:215-219
                const uint32_t ka = cA[sa], kb = cB[sb];
                h -= T[(ka == 0xFFu ? (uint32_t)(na - 1) : ka) * (uint32_t)nb + (kb == 0xFFu ? (uint32_t)(nb - 1) : kb)];
                ca = sa;
                cb = sb;
                --i;
                --j;
            } else if (lbit == 0u) { // :220-223
                h -= gap;
                ca = sa;
                cb = '-';
                --i;
            } else { // :224-227
                h -= gap;
                ca = '-';
                cb = sb;
                --j;
            }
            if (lane == 0) {
                outA[stride - 1 - len] = ca;
                outB[stride - 1 - len] = cb;
            }
            ++len;
        }
    }
    if (lane == 0)
        alnLen[pair] = (eA > 0 && lenA > 64u * R) ? 0xFFFFFFFFu : len;
}

// any lenA: H column and direction words in global scratch, lane-interleaved
__global__ __launch_bounds__(THREADS) void tb_generic_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ B, const uint64_t *__restrict__ offB, uint64_t lenB_shared,
    const int32_t *__restrict__ lut, int gap, const uint32_t *__restrict__ endA, const uint32_t *__restrict__ endB,
    const uint32_t *__restrict__ err, const int64_t *__restrict__ score, int smax, uint32_t wcols, int wide,
    uint32_t max_lenA, int32_t *__restrict__ hbuf,
    uint32_t *__restrict__ dirbuf, uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB,
    uint32_t *__restrict__ alnLen, uint32_t stride)
{
    const uint64_t local = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const uint64_t pair = pair0 + local;
    if (pair >= pair1)
        return;
    const uint64_t nl = (uint64_t)gridDim.x * THREADS; // lanes in this launch
    const uint32_t nw = (max_lenA + 15) / 16;
    const uint64_t o0 = offA[pair];
    const uint32_t lenA = (uint32_t)(offA[pair + 1] - o0);
    const uint8_t *ap = A + o0;
    const uint8_t *bp = offB ? B + offB[pair] : B;
    uint32_t eA = 0, eB = 0;
    if (err[pair] == 0u) {
        eA = endA[pair];
        eB = endB[pair];
    }
    uint32_t len = 0;
    if (eA > 0 && eB > 0) {
        const uint32_t mycols = pair_window(wcols, eA, score ? score[pair] : 0, smax, gap, wide);
        const uint32_t c_s = eB > mycols ? eB - mycols + 1u : 1u;
        const uint32_t ncol = eB - c_s + 1u;
        int32_t *Hc = hbuf + local;
        uint32_t *dirw = dirbuf + local;
        for (uint32_t i = 0; i < eA; ++i)
            Hc[(size_t)i * nl] = 0;
        for (uint32_t jr = 0; jr < ncol; ++jr) {
            const uint32_t bsym = bp[c_s + jr - 1];
            int diag = 0, up = 0;
            uint32_t word = 0;
            for (uint32_t i = 0; i < eA; ++i) { // rows below endA never matter
                const int s = lut[(uint32_t)ap[i] * 256u + bsym];
                const int left = Hc[(size_t)i * nl];
                const int d = diag + s, u = up + gap, l = left + gap;
                const int h = max(0, max(d, max(u, l)));
                const uint32_t code = h == 0 ? 0u : (h == d ? 1u : (h == u ? 2u : 3u));
                word |= code << (2 * (i & 15));
                diag = left;
                up = h;
                Hc[(size_t)i * nl] = h;
                if ((i & 15) == 15 || i == eA - 1) {
                    dirw[((size_t)jr * nw + (i >> 4)) * nl] = word;
                    word = 0;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        len = walk(dirw, nw, (uint32_t)nl, ap, bp, eA, eB, c_s, alnA + pair * stride, alnB + pair * stride, stride);
    }
    (void)lenB_shared;
    (void)lenA;
    alnLen[pair] = len;
}

// ---- NeedlemanWunsch (align.go:100-166): global alignment, any lengths --------------------------
// One pair per lane; the H column and the direction words of the WHOLE matrix live in global
// scratch (lane-interleaved).  Boundary H[i][0] = i*gap, H[0][j] = j*gap (:112-120); the cell is
// max(diag + s, up + gap, left + gap) (:131-134); the traceback stops as soon as EITHER index
// reaches 0 (:141), so leading residues are dropped exactly like the reference drops them; its
// last branch is an unconditional else (:154-158).
__global__ __launch_bounds__(THREADS) void nw_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                    uint64_t pair0, uint64_t pair1, const uint8_t *__restrict__ B,
                                                    const uint64_t *__restrict__ offB, uint64_t lenB_shared,
                                                    const int32_t *__restrict__ lut,
                                                    const uint8_t *__restrict__ validA,
                                                    const uint8_t *__restrict__ validB, int gap, uint32_t max_lenA,
                                                    int32_t *__restrict__ hbuf, uint32_t *__restrict__ dirbuf,
                                                    int64_t *__restrict__ score, uint32_t *__restrict__ err,
                                                    uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB,
                                                    uint32_t *__restrict__ alnLen, uint32_t stride)
{
    const uint64_t local = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const uint64_t pair = pair0 + local;
    if (pair >= pair1)
        return;
    const uint64_t nl = (uint64_t)gridDim.x * THREADS;
    const uint32_t nw = (max_lenA + 15) / 16;
    const uint64_t o0 = offA[pair];
    const uint32_t m = (uint32_t)(offA[pair + 1] - o0);
    const uint8_t *a = A + o0;
    const uint8_t *b = offB ? B + offB[pair] : B;
    const uint32_t n = (uint32_t)(offB ? offB[pair + 1] - offB[pair] : lenB_shared);
    // align.go:126-129 + matrix.go:29-36: the first failing Score() in row-major order
    uint32_t e = 0;
    if (m > 0 && n > 0) {
        if (!validA[a[0]]) {
            e = (1u << 8) | a[0];
        } else {
            for (uint32_t j = 0; j < n && !e; ++j)
                if (!validB[b[j]])
                    e = (2u << 8) | b[j];
            for (uint32_t i = 1; i < m && !e; ++i)
                if (!validA[a[i]])
                    e = (1u << 8) | a[i];
        }
    }
    err[pair] = e;
    if (e) {
        score[pair] = 0;
        alnLen[pair] = 0;
        return;
    }
    int32_t *Hc = hbuf + local;
    uint32_t *dirw = dirbuf + local;
    for (uint32_t i = 0; i < m; ++i)
        Hc[(size_t)i * nl] = (int32_t)(i + 1) * gap; // H[i+1][0]
    int top = 0;                                      // H[0][j-1]
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t bsym = b[j - 1];
        int diag = top;          // H[i-1][j-1], starting at H[0][j-1]
        int up = top + gap;      // H[0][j]
        top = up;
        uint32_t word = 0;
        for (uint32_t i = 0; i < m; ++i) {
            const int s = lut[(uint32_t)a[i] * 256u + bsym];
            const int left = Hc[(size_t)i * nl]; // H[i+1][j-1]
            const int d = diag + s, u = up + gap, l = left + gap;
            const int h = max(d, max(u, l));
            const uint32_t code = h == d ? 1u : (h == u ? 2u : 3u);
            word |= code << (2 * (i & 15));
            diag = left;
            up = h;
            Hc[(size_t)i * nl] = h;
            if ((i & 15) == 15 || i == m - 1) {
                dirw[((size_t)(j - 1) * nw + (i >> 4)) * nl] = word;
                word = 0;
            }
        }
    }
    score[pair] = m == 0 ? (int64_t)n * gap : (n == 0 ? (int64_t)m * gap : (int64_t)Hc[(size_t)(m - 1) * nl]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // traceback (:141-160); strings are built by append + reverse = filled from the back here
    uint8_t *outA = alnA + pair * stride, *outB = alnB + pair * stride;
    uint32_t i = m, j = n, len = 0;
    while (i > 0 && j > 0 && len < stride) {
        const uint32_t word = __hip_atomic_load(&dirw[((size_t)(j - 1) * nw + ((i - 1) >> 4)) * nl], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t code = (word >> (2 * ((i - 1) & 15))) & 3u;
        uint8_t ca, cb;
        if (code == 1u) {
            ca = a[i - 1];
            cb = b[j - 1];
            --i;
            --j;
        } else if (code == 2u) {
            ca = a[i - 1];
            cb = '-';
            --i;
        } else {
            ca = '-';
            cb = b[j - 1];
            --j;
        }
        outA[stride - 1 - len] = ca;
        outB[stride - 1 - len] = cb;
        ++len;
    }
    alnLen[pair] = len;
}

// ---- NeedlemanWunsch, register-tiled (lenA <= 256, compact table in LDS) -------------------------
// Same cell and bit recording as tb_prof_kernel minus the zero: d = diag + s, t = max(up, left) + gap,
// h = max(d, t), G = t > d (the diagonal wins ties, align.go:146), L = left > up ("up" is tested before the
// final else, :150-158); boundary column H[i][0] = i*gap in the registers, boundary row carried in `top`.
// Per-pair B: every lane looks its own column's symbol up in the table (one ds_read per cell).

template <int RA>
__global__ __launch_bounds__(THREADS) void nw_reg_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                        uint64_t pair0, uint64_t pair1, const uint8_t *__restrict__ B,
                                                        const uint64_t *__restrict__ offB, uint64_t lenB_shared,
                                                        const uint8_t *__restrict__ codeA, const uint8_t *__restrict__ codeB,
                                                        const int32_t *__restrict__ lutcc, int na, int nb, int gap,
                                                        uint32_t max_lenB, uint32_t *__restrict__ dirbuf,
                                                        int64_t *__restrict__ score, uint32_t *__restrict__ err,
                                                        uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB,
                                                        uint32_t *__restrict__ alnLen, uint32_t stride)
{
    constexpr int NG = (RA + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) int32_t T[]; // [na][nb] then codeA[256], codeB[256]
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();

    const uint64_t pair = pair0 + (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < pair1;
    uint32_t m = 0, n = 0;
    const uint8_t *a = A, *b = B;
    if (active) {
        const uint64_t o0 = offA[pair];
        m = (uint32_t)(offA[pair + 1] - o0);
        a = A + o0;
        b = offB ? B + offB[pair] : B;
        n = (uint32_t)(offB ? offB[pair + 1] - offB[pair] : lenB_shared);
    }
    // align.go:126-129 + matrix.go:29-36: the first failing Score() in row-major order
    uint32_t e = 0;
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
    const bool work = active && e == 0u && m > 0 && n > 0 && m <= RA;

    uint32_t aoff[RA / 2]; // row offsets into T (code * nb), two per register; rows >= m use the pad row
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
        H[i] = (i + 1) * gap; // H[i+1][0], align.go:112-115

    const uint64_t wave_global = ((uint64_t)blockIdx.x * THREADS + tid) >> 6;
    uint32_t *dirw = dirbuf + wave_global * ((size_t)max_lenB * NG * 2 * 64) + lane;
    const uint32_t ncol = work ? n : 0u;
    int top = 0; // H[0][j-1]
    for (uint32_t jr = 0; jr < max_lenB; ++jr) {
        if (!__any(jr < ncol))
            break;
        if (jr < ncol) {
            const uint32_t cb = cB[b[jr]];
            int diag = top, up = top + gap; // H[0][j-1], H[0][j] (:117-120)
            top = up;
            uint32_t wG = 0, wL = 0;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const uint32_t ro = (aoff[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const int s = T[ro + cb];
                const int left = H[i];
                const int d = diag + s;
                const int t = max(up, left) + gap;
                const int h = max(d, t);
                PH_CARRY_BIT(wG, t, d);
                PH_CARRY_BIT(wL, left, up);
                diag = left;
                up = h;
                H[i] = h;
                if ((i & 31) == 31 || i == RA - 1) {
                    dirw[(((size_t)jr * NG + (i >> 5)) * 2 + 0) * 64] = wG;
                    dirw[(((size_t)jr * NG + (i >> 5)) * 2 + 1) * 64] = wL;
                }
            }
        }
    }
    if (!active)
        return;
    err[pair] = e;
    if (e || m > RA) {
        score[pair] = 0;
        alnLen[pair] = m > RA && !e ? 0xFFFFFFFFu : 0u;
        return;
    }
    int last = 0; // H[m][n]
#pragma unroll
    for (int i = 0; i < RA; ++i)
        last = (uint32_t)i + 1u == m ? H[i] : last;
    score[pair] = m == 0 ? (int64_t)n * gap : (n == 0 ? (int64_t)m * gap : (int64_t)last);
    uint32_t len = 0;
    if (work) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my own stores, read back by me
        uint8_t *outA = alnA + pair * stride, *outB = alnB + pair * stride;
        uint32_t i = m, j = n;
        while (i > 0 && j > 0 && len < stride) { // :141: stops as soon as EITHER index reaches 0
            const uint32_t r = i - 1u, g = r >> 5;
            const uint32_t rows = min(32u, (uint32_t)RA - 32u * g);
            const uint32_t bit = rows - 1u - (r & 31u);
            const uint32_t *wp = dirw + ((size_t)(j - 1u) * NG + g) * 2 * 64;
            const uint32_t wg = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t wl = __hip_atomic_load(wp + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint8_t ca, cb;
            if (((wg >> bit) & 1u) == 0u) {
                ca = a[i - 1];
                cb = b[j - 1];
                --i;
                --j;
            } else if (((wl >> bit) & 1u) == 0u) {
                ca = a[i - 1];
                cb = '-';
                --i;
            } else {
                ca = '-';
                cb = b[j - 1];
                --j;
            }
            outA[stride - 1 - len] = ca;
            outB[stride - 1 - len] = cb;
            ++len;
        }
    }
    alnLen[pair] = len;
}

// ---- NeedlemanWunsch for 256 < lenA <= 4096: one wave per pair (same sweep as tb_wave_kernel) ----------
// Boundaries H[i][0] = i*gap (the lanes' initial registers), H[0][j] = j*gap (fed to lane 0); the whole matrix's
// G / L bits are kept (no window in a global alignment); the walk stops when either index reaches 0 (align.go:141).

template <int R>
__global__ __launch_bounds__(THREADS) void nw_wave_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t pair0, uint64_t pair1,
    const uint8_t *__restrict__ Bbase, const uint64_t *__restrict__ offB, uint64_t lenB_shared,
    const uint8_t *__restrict__ codeA, const uint8_t *__restrict__ codeB, const int32_t *__restrict__ lutcc, int na,
    int nb, int gap, uint32_t max_lenB, uint32_t *__restrict__ dirbuf, int64_t *__restrict__ score,
    uint32_t *__restrict__ err, uint8_t *__restrict__ alnA, uint8_t *__restrict__ alnB, uint32_t *__restrict__ alnLen,
    uint32_t stride)
{
    constexpr int NG = (R + 31) / 32;
    constexpr int NWL = R <= 16 ? 1 : 2 * NG;
    extern __shared__ __attribute__((aligned(16))) int32_t T[];
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();
    const uint64_t wslot = (uint64_t)blockIdx.x * (THREADS / 64) + (tid >> 6);
    const uint64_t pair = pair0 + wslot;
    if (pair >= pair1)
        return; // wave-uniform
    const uint64_t o0 = offA[pair];
    const uint32_t m = (uint32_t)(offA[pair + 1] - o0);
    const uint8_t *a = A + o0;
    const uint8_t *b = offB ? Bbase + offB[pair] : Bbase;
    const uint32_t n = (uint32_t)(offB ? offB[pair + 1] - offB[pair] : lenB_shared);
    // align.go:126-129 + matrix.go:29-36: the first failing Score() in row-major order, found by the wave
    uint32_t e = 0;
    if (m > 0 && n > 0) {
        uint32_t abad = 0xFFFFFFFFu, bbad = 0xFFFFFFFFu;
        for (uint32_t i0 = 0; i0 < m && abad == 0xFFFFFFFFu; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const uint64_t bad = __ballot(i < m && cA[a[i]] == 0xFFu);
            if (bad)
                abad = i0 + (uint32_t)__builtin_ctzll(bad);
        }
        for (uint32_t j0 = 0; j0 < n && bbad == 0xFFFFFFFFu; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint64_t bad = __ballot(j < n && cB[b[j]] == 0xFFu);
            if (bad)
                bbad = j0 + (uint32_t)__builtin_ctzll(bad);
        }
        if (abad == 0u)
            e = (1u << 8) | a[0];
        else if (bbad != 0xFFFFFFFFu)
            e = (2u << 8) | b[bbad];
        else if (abad != 0xFFFFFFFFu)
            e = (1u << 8) | a[abad];
    }
    if (e || m == 0 || n == 0 || m > 64u * R) {
        if (lane == 0) {
            err[pair] = e;
            score[pair] = e ? 0 : (m == 0 ? (int64_t)n * gap : (int64_t)m * gap);
            alnLen[pair] = (!e && m > 64u * R) ? 0xFFFFFFFFu : 0u;
        }
        return;
    }
    uint32_t ro[R];
    int Hrow[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t r = (uint32_t)lane * R + k;
        ro[k] = (r < m ? (uint32_t)cA[a[r]] : (uint32_t)(na - 1)) * (uint32_t)nb;
        Hrow[k] = (int)(r + 1u) * gap; // H[r+1][0], align.go:112-115
    }
    uint32_t *dirw = dirbuf + wslot * ((size_t)(max_lenB + 63u) * 64 * NWL) + (size_t)lane * NWL;
    int topprev = (int)((uint32_t)lane * R) * gap; // H[l*R][0]: the diagonal of my first row in column 1
    int last_h = 0;
    uint32_t last_b = (uint32_t)(nb - 1);
    auto load_chunk = [&](uint32_t s0) -> uint32_t {
        return s0 + (uint32_t)lane < n ? (uint32_t)cB[b[s0 + (uint32_t)lane]] : (uint32_t)(nb - 1);
    };
    uint32_t chunk = load_chunk(0), next_chunk = 0u;
    const uint32_t steps = n + 63u;
    uint32_t acc = 0u; // the word being filled (R <= 16)
    bool accany = false;
    for (uint32_t s = 0; s < steps; ++s) {
        if ((s & 63u) == 0u) {
            if (s)
                chunk = next_chunk;
            next_chunk = load_chunk(s + 64u);
        }
        int top_in = from_lane_below(last_h);
        uint32_t b_in = (uint32_t)from_lane_below((int)last_b);
        const uint32_t b_new = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)(s & 63u));
        const uint32_t jr = s - (uint32_t)lane; // 0-based column; wraps for lanes that have not started
        const bool valid = jr < n;
        if (lane == 0) {
            top_in = (int)(jr + 1u) * gap; // H[0][j], :117-120
            b_in = b_new;
        }
        int diag = topprev, up = top_in;
        uint32_t gw[NG], lw[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
            gw[g] = lw[g] = 0u;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int left = Hrow[k];
            const int sc = T[ro[k] + b_in];
            const int d = diag + sc;
            const int t = max(up, left) + gap;
            const int h = max(d, t);
            PH_CARRY_BIT(gw[k >> 5], t, d);
            PH_CARRY_BIT(lw[k >> 5], left, up);
            diag = left;
            up = valid ? h : left;          // lanes outside the matrix keep their boundary column
            Hrow[k] = valid ? h : left;
        }
        if (valid) {
            topprev = top_in;
            last_h = Hrow[R - 1];
        }
        if constexpr (R <= 16) {
            constexpr int BITS = WavePack<R>::BITS, FB = WavePack<R>::FB, SP = WavePack<R>::SP;
            const uint32_t sub = s % SP;
            if (valid) {
                acc |= (gw[0] | (lw[0] << BITS)) << (sub * FB);
                accany = true;
            }
            if (sub == SP - 1 || s == steps - 1) {
                if (accany)
                    dirw[(size_t)(s / SP) * 64 * NWL] = acc;
                acc = 0u;
                accany = false;
            }
        } else if (valid) {
            uint32_t *o = dirw + (size_t)s * 64 * NWL;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                o[g] = gw[g];
                o[NG + g] = lw[g];
            }
        }
        last_b = b_in;
    }
    // H[m][n] sits in the lane that owns row m-1
    int mine = 0;
#pragma unroll
    for (int k = 0; k < R; ++k)
        mine = (uint32_t)lane * R + k + 1u == m ? Hrow[k] : mine;
    const int total = __shfl(mine, (int)((m - 1u) / R), 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint8_t *outA = alnA + pair * stride, *outB = alnB + pair * stride;
    const uint32_t *dbase = dirbuf + wslot * ((size_t)(max_lenB + 63u) * 64 * NWL);
    uint32_t i = m, j = n, len = 0;
    while (i > 0 && j > 0 && len < stride) { // :141: stops as soon as EITHER index reaches 0
        const uint32_t r = i - 1u, l = r / R, k = r % R;
        const uint32_t s = (j - 1u) + l;
        const uint32_t *wp = dbase + ((size_t)(R <= 16 ? s / WavePack<R>::SP : s) * 64 + l) * NWL;
        uint32_t gbit, lbit;
        if (R <= 16) {
            const uint32_t w = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                               ((s % WavePack<R>::SP) * WavePack<R>::FB);
            gbit = (w >> (R - 1 - k)) & 1u;
            lbit = (w >> (WavePack<R>::BITS + R - 1 - k)) & 1u;
        } else {
            const uint32_t g = k >> 5, bit = 31u - (k & 31u);
            gbit = (__hip_atomic_load(wp + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> bit) & 1u;
            lbit = (__hip_atomic_load(wp + NG + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> bit) & 1u;
        }
        uint8_t ca, cb;
        if (gbit == 0u) {
            ca = a[i - 1];
            cb = b[j - 1];
            --i;
            --j;
        } else if (lbit == 0u) {
            ca = a[i - 1];
            cb = '-';
            --i;
        } else {
            ca = '-';
            cb = b[j - 1];
            --j;
        }
        if (lane == 0) {
            outA[stride - 1 - len] = ca;
            outB[stride - 1 - len] = cb;
        }
        ++len;
    }
    if (lane == 0) {
        err[pair] = 0;
        score[pair] = (int64_t)total;
        alnLen[pair] = len;
    }
}

struct Plan {
    int ra;            // 0 = generic
    Window win;
    size_t per_pair;   // workspace bytes per pair (the larger of the two register-tiled layouts)
    size_t smem;
    // profile variant (shared reference, score known): tb_prof_kernel
    bool prof_ok;
    int cp;
    uint32_t lenB_pad, nblk_alloc;
    size_t prof_bytes, prof_smem;
    // its half-float two-band form (tb_prof16_kernel): every H < 2048, the table of halves fits twice into a CU's LDS
    bool half_ok;
    size_t half_smem;
    // 153..256 rows, two lanes per pair (tb_prof16x2_kernel): the same table, four bands of 64 rows
    bool half2_ok;
    uint32_t nblk_alloc2;
    size_t half2_per_pair;
    // every pair its own B on packed halves (tb_pair16_kernel): rows <= 152, at most six symbol codes
    bool pair16_ok;
    uint32_t pair16_nblk_alloc;
    size_t pair16_smem;
    // one wave per pair for 256 < lenA <= 4096 (score known): tb_wave_kernel
    int wave_r;            // 0 = not applicable
    size_t wave_per_pair;
    // ... with the sweep on a byte profile of the pair in LDS (tb_wave_kernel<R, true>): 257..1024 rows, scores - gap in int8
    bool wave8_ok;
    size_t wave8_smem;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static Plan plan(const polyhip_scoring *sc, uint32_t max_lenA, uint64_t lenB)
{
    Plan p{};
    p.win = window(sc, max_lenA, lenB);
    const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
    p.smem = (size_t)na * nb * 4 + 512;
    const bool reg = max_lenA <= 256 && p.smem <= 60 * 1024 && (size_t)na * nb < 65536;
    if (reg) {
        p.ra = max_lenA <= 64 ? 64 : max_lenA <= 152 ? 152 : 256;
        p.per_pair = (size_t)p.win.wcols * ((p.ra + 15) / 16) * 4;
    } else {
        p.ra = 0;
        p.per_pair = (size_t)p.win.wcols * ((max_lenA + 15) / 16) * 4 + (size_t)max_lenA * 4;
    }
    if (p.per_pair == 0)
        p.per_pair = 4;
    p.wave_r = 0;
    // (153..256 rows: a lane-per-pair kernel holds them only at one wave per SIMD with part of H in AGPRs / scratch;
    // per-pair B at 250 x 250 the table kernel took 27.6 ms per 200k pairs, the wave kernel takes a third of that)
    // (not below 153 rows, although per-pair B at 150 x 150 it would be 5.2 against the table kernel's 6.6 ms: its
    // 256 B per step and pair would shrink the chunks of every batch sized with polyhip_sw_traceback_workspace_bytes,
    // config 4's byte-profile traceback included -- first the layout has to shrink to the packed word count)
    if (max_lenA > 152 && max_lenA <= 4096 && p.smem <= 60 * 1024 && (size_t)na * nb < 65536) {
        p.wave_r = max_lenA <= 256 ? 4 : max_lenA <= 512 ? 8 : max_lenA <= 1024 ? 16 : max_lenA <= 2048 ? 32 : 64;
        const size_t nwl = p.wave_r <= 16 ? 1 : 2 * ((p.wave_r + 31) / 32);
        p.wave_per_pair = ((size_t)p.win.wcols + 4 + 63) * 64 * nwl * 4; // (+ 4 columns: a deferred end cell's window starts at a block's end)
        p.per_pair = std::max(p.per_pair, p.wave_per_pair);
        p.wave8_smem = align_up(p.smem, 16) + (size_t)(THREADS / 64) * nb * p.wave_r * 64;
        p.wave8_ok = (p.wave_r == 8 || p.wave_r == 16) && sc->gap <= -1 && -sc->gap <= 127 && (int64_t)sc->smax - sc->gap <= 127 &&
                     (int64_t)sc->smin - sc->gap >= -128 && p.wave8_smem <= 64 * 1024;
    }
    p.pair16_ok = reg && (p.ra == 64 || p.ra == 152) && sc->ncodes <= 6 && sc->int8_ok && sc->gap <= -1 && sc->smax > 0 && lenB > 0 &&
                  (uint64_t)sc->smax * std::min<uint64_t>(max_lenA, lenB) <= 2047ull && (int64_t)sc->smax - sc->gap <= 2048;
    if (p.pair16_ok) {
        p.pair16_nblk_alloc = (p.win.wcols + 2 * (TBU - 1)) / TBU + 3;
        p.pair16_smem = 2 * (size_t)THREADS * 56 + (size_t)(sc->ncodes + 1) * 512 + 256;
        p.per_pair = std::max(p.per_pair, (size_t)p.pair16_nblk_alloc * TBU * ((p.ra + 31) / 32) * 2 * 4);
    }
    p.cp = sc->cp <= 8 ? 8 : 32;
    p.lenB_pad = (uint32_t)align_up((size_t)std::min<uint64_t>(lenB, 1u << 30), TBU);
    p.prof_smem = (size_t)p.lenB_pad * p.cp + 256;
    p.prof_ok = reg && sc->int8_ok && sc->gap <= -1 && sc->smax > 0 && sc->cp <= 32 && lenB > 0 && lenB < (1u << 30) &&
                p.prof_smem <= 160 * 1024 && (uint64_t)sc->smax * max_lenA < (1ull << 30);
    if (p.prof_ok) {
        // a lane's window starts on a block boundary (up to 3 columns early) and ends inside a block
        p.nblk_alloc = (p.win.wcols + 2 * (TBU - 1)) / TBU + 3; // (+ one block for the deferred end cell's slack, + one
                                                                // iteration for the half-float kernel's second band)
        p.prof_bytes = align_up((size_t)p.lenB_pad * p.cp, 256);
        const size_t half_tab = ((size_t)p.lenB_pad / 4 + 1) * (size_t)(sc->ncodes + 1) * 8;
        p.half_smem = half_tab + 256;
        p.half_ok = (p.ra == 64 || p.ra == 152) && p.half_smem <= 79 * 1024 &&
                    (uint64_t)sc->smax * std::min<uint64_t>(max_lenA, lenB) <= 2047ull && (int64_t)sc->smax - sc->gap <= 2048;
        p.half2_ok = p.ra == 256 && p.half_smem <= 79 * 1024 &&
                     (uint64_t)sc->smax * std::min<uint64_t>(max_lenA, lenB) <= 2047ull && (int64_t)sc->smax - sc->gap <= 2048;
        p.nblk_alloc2 = p.nblk_alloc + 2; // bands 2 and 3 finish two iterations after band 1
        if (p.half_ok || p.half2_ok)
            p.prof_bytes = std::max(p.prof_bytes, align_up(half_tab, 256));
        const size_t per = (size_t)p.nblk_alloc * TBU * ((p.ra + 31) / 32) * 2 * 4;
        p.per_pair = std::max(p.per_pair, per);
        // two lanes per pair, each with the words of its two bands (4 groups of 16 rows, G and L)
        p.half2_per_pair = (size_t)p.nblk_alloc2 * TBU * 4 * 2 * 4 * 2;
        if (p.half2_ok)
            p.per_pair = std::max(p.per_pair, p.half2_per_pair);
    }
    return p;
}

// NW workspace per pair: the larger of the generic layout (2-bit codes + the H column) and the
// register-tiled one (G and L words per 32 rows of RA)
static inline int nw_ra(uint32_t max_lenA) { return max_lenA <= 64 ? 64 : 0; } // beyond: one wave per pair
static inline int nw_wave_r(uint32_t max_lenA)
{
    return max_lenA <= 64 ? 0 : max_lenA <= 256 ? (int)((max_lenA + 63) / 64) : max_lenA <= 512 ? 8 : max_lenA <= 1024 ? 16 : max_lenA <= 2048 ? 32 : max_lenA <= 4096 ? 64 : 0;
}
static uint64_t nw_per_pair(uint32_t max_lenA, uint64_t max_lenB)
{
    const uint64_t generic = max_lenB * ((max_lenA + 15) / 16) * 4 + (uint64_t)max_lenA * 4 + 8;
    const int ra = nw_ra(max_lenA);
    const uint64_t reg = ra ? max_lenB * (uint64_t)((ra + 31) / 32) * 8 : 0;
    const int wr = nw_wave_r(max_lenA);
    const uint64_t wave = wr ? (max_lenB + 63) * 64 * (uint64_t)(wr <= 16 ? 1 : 2 * ((wr + 31) / 32)) * 4 : 0;
    return std::max<uint64_t>(std::max(std::max(generic, reg), wave), 8);
}


// ---- packed strings: the last alnLen[p] bytes of pair p's two slots -> [off[p], off[p] + alnLen[p]) of two packed buffers ----
constexpr int PACK_BLOCK = 1024; // pairs per scan block

__global__ __launch_bounds__(256) void pack_sums_kernel(const uint32_t *__restrict__ alnLen, uint64_t n, uint64_t *__restrict__ bsum)
{
    __shared__ uint64_t ws[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * PACK_BLOCK;
    uint64_t v = 0;
    for (uint32_t t = threadIdx.x; t < PACK_BLOCK; t += 256)
        if (i0 + t < n)
            v += alnLen[i0 + t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0)
        ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the block sums in place (one workgroup; nb <= a few thousand), total behind them
__global__ __launch_bounds__(1024) void pack_scan_kernel(uint64_t *__restrict__ bsum, uint32_t nb)
{
    __shared__ uint64_t ws[16];
    __shared__ uint64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t i = b0 + tid;
        const uint64_t v = i < nb ? bsum[i] : 0;
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
        uint64_t pre = carry;
        for (int w = 0; w < (tid >> 6); ++w)
            pre += ws[w];
        if (i < nb)
            bsum[i] = pre + incl - v;
        __syncthreads();
        if (tid == 1023)
            carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0)
        bsum[nb] = carry;
}

// offsets (global: + base) and the copies; 16 lanes per pair, unaligned dwords
// base_ptr != nullptr (the multi-device shards, whose strings stay on the device until every shard's total is known): the
// bytes in front of this chunk are read from the device -- the previous chunk's last offset -- and outA / outB are the
// shard's whole buffers, so chunk after chunk is appended without the host waiting for a total.
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint32_t *__restrict__ alnLen, uint64_t n, const uint64_t *__restrict__ bsum,
                                                       uint64_t base, const uint8_t *__restrict__ slotA, const uint8_t *__restrict__ slotB,
                                                       uint32_t stride, uint64_t *__restrict__ off, uint8_t *__restrict__ outA,
                                                       uint8_t *__restrict__ outB, const uint64_t *base_ptr = nullptr)
{
    if (base_ptr) {
        base = *base_ptr; // == off[0] of this chunk: written by the previous chunk's kernel, rewritten below with the same value
        outA += base;
        outB += base;
    }
    __shared__ uint32_t lens[PACK_BLOCK];
    __shared__ uint64_t offs[PACK_BLOCK];
    __shared__ uint64_t ws[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * PACK_BLOCK;
    const int tid = threadIdx.x;
    // the block's own exclusive scan: 4 pairs per thread
    uint32_t l[4];
    uint64_t sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t p = i0 + 4 * tid + q;
        l[q] = p < n ? alnLen[p] : 0u;
        sum += l[q];
    }
    uint64_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d)
            incl += t;
    }
    if ((tid & 63) == 63)
        ws[tid >> 6] = incl;
    __syncthreads();
    uint64_t run = bsum[blockIdx.x] + incl - sum; // chunk-local
    for (int w = 0; w < (tid >> 6); ++w)
        run += ws[w];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t p = i0 + 4 * tid + q;
        lens[4 * tid + q] = l[q];
        offs[4 * tid + q] = run;
        if (p < n)
            off[p] = base + run;
        run += l[q];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255)
        off[n] = base + bsum[gridDim.x]; // = base + the chunk's total
    __syncthreads();
    const int g = tid >> 4, gl = tid & 15; // 16 groups of 16 lanes
    for (uint32_t j = g; j < PACK_BLOCK && i0 + j < n; j += 16) {
        const uint32_t len = lens[j];
        if (len == 0)
            continue;
        const uint64_t p = i0 + j;
        const uint8_t *sa = slotA + (p + 1) * (uint64_t)stride - len, *sb = slotB + (p + 1) * (uint64_t)stride - len;
        uint8_t *da = outA + offs[j], *db = outB + offs[j];
        const uint32_t nd = len >> 2;
        for (uint32_t w = gl; w < nd; w += 16) {
            uint32_t x, y;
            __builtin_memcpy(&x, sa + 4 * w, 4);
            __builtin_memcpy(&y, sb + 4 * w, 4);
            __builtin_memcpy(da + 4 * w, &x, 4);
            __builtin_memcpy(db + 4 * w, &y, 4);
        }
        if ((uint32_t)gl < (len & 3u)) {
            da[4 * nd + gl] = sa[4 * nd + gl];
            db[4 * nd + gl] = sb[4 * nd + gl];
        }
    }
}
} // namespace k3t
} // namespace polyhip

using namespace polyhip;

extern "C" {

} // extern "C"
polyhip::KernelChoice polyhip::kernel_choice_get()
{
    KernelChoice c;
    k3::score_choice(&c.sw_path, &c.sw_half, false);
    c.tb_path = k3t::g_tb_last_path;
    c.tb_half = k3t::g_tb_last_half;
    c.nw_path = k3t::g_nw_last_path;
    return c;
}
void polyhip::kernel_choice_set(const KernelChoice &c)
{
    int p = c.sw_path, h = c.sw_half;
    k3::score_choice(&p, &h, true);
    k3t::g_tb_last_path = c.tb_path;
    k3t::g_tb_last_half = c.tb_half;
    k3t::g_nw_last_path = c.nw_path;
}
extern "C" {
int polyhip_sw_traceback_last_path(void) { return k3t::g_tb_last_path; }
int polyhip_sw_traceback_last_half(void) { return k3t::g_tb_last_half; }

uint32_t polyhip_sw_traceback_stride(const polyhip_scoring *sc, uint32_t max_lenA, uint64_t lenB)
{
    if (!sc)
        return 0;
    return k3t::window(sc, max_lenA, lenB).stride;
}

size_t polyhip_sw_traceback_workspace_bytes(const polyhip_scoring *sc, uint64_t npairs, uint32_t max_lenA, uint64_t lenB)
{
    if (!sc)
        return 0;
    const k3t::Plan p = k3t::plan(sc, max_lenA, lenB);
    // enough for every pair in one launch, capped at 8 GiB (the entry point loops over chunks);
    // never less than one workgroup's worth
    const uint64_t padded = (npairs + k3t::THREADS - 1) / k3t::THREADS * k3t::THREADS;
    uint64_t want = padded * (p.per_pair + 8); // (+ 8: what the half-float kernels hand to their walk kernel per pair)
    const uint64_t cap = 8ull << 30, floor_ = (uint64_t)k3t::THREADS * (p.per_pair + 8);
    if (want > cap)
        want = cap / floor_ * floor_;
    if (want < floor_)
        want = floor_;
    return (size_t)want + 256 + p.prof_bytes;
}

// would polyhip_sw_traceback_dev take the byte-profile kernel for this batch (score given, one shared reference)?
// Only that kernel can find a deferred end cell.
static bool traceback_uses_prof(const polyhip_scoring *sc, uint32_t max_lenA, uint64_t lenB)
{
    const k3t::Plan p = k3t::plan(sc, max_lenA, lenB);
    const bool wave_ok = p.wave_r != 0 && !env_is("POLYHIP_TB_WAVE", '0');
    if (p.prof_ok && p.half2_ok && !env_is("POLYHIP_TB_PROF", '0') && !env_is("POLYHIP_TB_F16", '0') && !env_is("POLYHIP_TB_HALF2", '0'))
        return true; // its two-lanes-per-pair form (153..256 rows)
    if (p.ra == 0 && wave_ok && p.wave8_ok && !env_is("POLYHIP_TB_WAVE8", '0'))
        return true; // 257..1024 rows: the one-wave-per-pair kernel on a byte profile of the pair (path 7)
    return p.prof_ok && !(p.ra == 256 && wave_ok) && !env_is("POLYHIP_TB_PROF", '0');
}

// endA / endB / err are rewritten only for pairs whose end cell the score pass deferred (k3p::SW_END_DEFERRED)
static int traceback_impl(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                          uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB,
                          uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err,
                          const int64_t *d_score, uint8_t *d_alnA,
                          uint8_t *d_alnB, uint32_t *d_alnLen, uint32_t aln_stride, void *d_work, size_t work_bytes,
                          polyhip_stream_t stream, int deferred)
{
    PH_REQUIRE(sc, "polyhip_sw_traceback: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_offA && d_endA && d_endB && d_err && d_alnA && d_alnB && d_alnLen && d_work,
               "polyhip_sw_traceback: null pointer");
    const k3t::Plan p = k3t::plan(sc, max_lenA, lenB);
    PH_REQUIRE(aln_stride >= p.win.stride, "polyhip_sw_traceback: aln_stride %u < %u (polyhip_sw_traceback_stride)",
               aln_stride, p.win.stride);
    // 153..256 rows: one wave per pair beats both lane-per-pair kernels (400k x 250 bp vs 5 kb: 34.9 ms against the
    // byte-profile kernel's 44.0 at one wave per SIMD); they remain for tables too large for the wave kernel's LDS
    const bool wave_ok = p.wave_r != 0 && d_score != nullptr && d_B != nullptr &&
                         !env_is("POLYHIP_TB_WAVE", '0'); // testing aid: no one-wave-per-pair traceback
    // 153..256 rows against one reference on packed halves, two lanes per pair (path 5; POLYHIP_TB_HALF2=0 or
    // POLYHIP_TB_F16=0: the one-wave-per-pair kernel as before, testing aids): 400k x 250 bp vs 5 kb 34.5 ms -> see DESIGN.md
    const bool use_half2 = p.prof_ok && p.half2_ok && d_offB == nullptr && d_score != nullptr && d_B != nullptr &&
                           !env_is("POLYHIP_TB_PROF", '0') && !env_is("POLYHIP_TB_F16", '0') && !env_is("POLYHIP_TB_HALF2", '0');
    const bool use_prof = !use_half2 && p.prof_ok && d_offB == nullptr && d_score != nullptr && d_B != nullptr &&
                          !(p.ra == 256 && wave_ok) &&
                          !env_is("POLYHIP_TB_PROF", '0'); // testing aid: the table kernel for a shared reference
    const bool use_wave = !use_half2 && !use_prof && (p.ra == 0 || p.ra == 256) && wave_ok;
    // every pair its own B, rows <= 152, on packed halves (path 6; POLYHIP_TB_PAIR16=0 or POLYHIP_TB_F16=0: the table kernel)
    const bool use_pair16 = p.pair16_ok && d_offB != nullptr && d_score != nullptr && d_B != nullptr && !use_wave &&
                            !env_is("POLYHIP_TB_F16", '0') && !env_is("POLYHIP_TB_PAIR16", '0');
    // 257..1024 rows: the one-wave-per-pair kernel's sweep on a byte profile of the pair (path 7; POLYHIP_TB_WAVE8=0: path 4)
    const bool use_wave8 = use_wave && p.wave8_ok && !env_is("POLYHIP_TB_WAVE8", '0');
    k3t::g_tb_last_path = use_pair16 ? 6 : use_half2 ? 5 : use_prof ? 1 : use_wave8 ? 7 : use_wave ? 4 : (p.ra ? 2 : 3);
    // only the byte-profile kernels know a deferred end cell: the fused entry point decided with traceback_uses_prof();
    // should the two conditions ever drift apart, fail here instead of walking from row 4e9
    PH_REQUIRE(!deferred || use_prof || use_half2 || use_wave8, "polyhip_sw_align_batch: end cells were deferred but the byte-profile traceback is not taken");
    // bit 0: the conservative per-pair window (POLYHIP_TB_WIDE=1, testing aid); bit 1: deferred end cells allowed
    const int wide = (env_is("POLYHIP_TB_WIDE", '1') ? 1 : 0) | (deferred ? 2 : 0) | (env_is("POLYHIP_TB_NOWALK", '1') ? 4 : 0) |
                     (env_is("POLYHIP_TB_WALKREG", '0') ? 8 : 0); // bit 3: the one-wave-per-pair kernel's plain walk (testing aid)
    PH_REQUIRE(work_bytes >= p.prof_bytes + 256, "polyhip_sw_traceback: workspace too small (%zu B)", work_bytes);
    const size_t usable = (work_bytes - p.prof_bytes) & ~(size_t)255;
    // (the two-lane kernel's direction words take a quarter of the one-wave-per-pair kernel's, which sizes the workspace of
    // its row class: its chunks are cut by its own figure, or they would fill a third of the chip)
    // POLYHIP_TB_SPLITWALK=1 (a measured alternative, kept as a cross-check): the half-float kernels' walk as a kernel of its
    // own behind each sweep (tb_walk16_kernel; 8 bytes per pair behind a chunk's direction words carry (window start, wave
    // lag) over).  With a thread per pair and eight waves per SIMD every pair of a chunk walks at once -- and a million
    // half-written output lines are in flight against 32 MB of L2: 9.4 ms per 1M config-4 reads against 8.5 ms with the
    // walk at the end of the sweep kernel (profiles/r04_tb_walk.log).
    const bool half_any = use_half2 || (use_prof && p.half_ok && !env_is("POLYHIP_TB_F16", '0'));
    const bool split_walk = half_any && env_is("POLYHIP_TB_SPLITWALK", '1');
    const size_t per_pair = (use_half2 ? p.half2_per_pair : p.per_pair) + (split_walk ? 8 : 0);
    uint64_t chunk = usable / per_pair / k3t::THREADS * k3t::THREADS;
    // A batch that needs several chunks of the direction workspace: the byte-profile and the one-wave-per-pair kernels take them through the two
    // HALVES of the workspace on two streams (the caller's and one of the library's), so that the end of one chunk --
    // waves finish at different times, and the walk that closes a wave's work waits on memory, not on issue -- overlaps
    // the sweeps of the next (one chunk after the other: the waves of a chunk were resident 67 % of its time,
    // profiles/r02_tbh_pmc_a.md).  POLYHIP_TB_OVERLAP=0: one chunk after the other (testing aid).
    const size_t half_bytes = (usable / 2) & ~(size_t)255; // where the upper half starts
    const uint64_t half_chunk = half_bytes / per_pair / k3t::THREADS * k3t::THREADS;
    const bool overlap = (use_prof || use_wave || use_half2) && npairs > chunk && half_chunk >= 16384 && !env_is("POLYHIP_TB_OVERLAP", '0');
    if (overlap)
        chunk = half_chunk;
    else if (use_prof && chunk >= 131072)
        chunk = chunk / 131072 * 131072; // whole rounds of 256 CUs x 2 workgroups x 256 pairs
    else if (use_half2 && chunk >= 65536)
        chunk = chunk / 65536 * 65536; // the same with 128 pairs per workgroup
    PH_REQUIRE(chunk >= (uint64_t)k3t::THREADS, "polyhip_sw_traceback: workspace too small (%zu B; %zu B per pair, >= %d pairs)",
               work_bytes, per_pair, k3t::THREADS);
    hipStream_t st = as_stream(stream);
    const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
    int8_t *prof = static_cast<int8_t *>(d_work);
    void *d_dir = static_cast<uint8_t *>(d_work) + p.prof_bytes;
    // half-float two-band form of the byte-profile kernel (POLYHIP_TB_F16=0: the 32-bit one, testing aid)
    const bool use_half = use_prof && p.half_ok && !env_is("POLYHIP_TB_F16", '0');
    k3t::g_tb_last_half = (use_half || use_half2 || use_pair16) ? 1 : 0;
    if (use_half || use_half2) {
        const uint32_t n16 = (p.lenB_pad / 4 + 1) * (uint32_t)(sc->ncodes + 1);
        hipLaunchKernelGGL(k3t::tb_profile16_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, d_B, (uint32_t)lenB, p.lenB_pad,
                           sc->d_lutc, sc->ncodes, sc->ncodes + 1, reinterpret_cast<uint2 *>(prof));
        PH_HIP(hipGetLastError());
    } else if (use_prof) {
        hipLaunchKernelGGL(k3t::tb_profile_kernel, dim3((p.lenB_pad + 255) / 256), dim3(256), 0, st, d_B, (uint32_t)lenB,
                           p.lenB_pad, sc->d_lutc, sc->ncodes, p.cp, prof);
        PH_HIP(hipGetLastError());
    }
    const hipStream_t caller_st = st;
    AuxStream &aux = aux_stream(st);
    // whatever way this function is left after the fork, the caller's stream waits for the library's
    struct Joiner {
        AuxStream &a;
        hipStream_t caller;
        bool armed = false;
        ~Joiner()
        {
            if (armed)
                (void)a.join(caller);
        }
    } joiner{aux, caller_st};
    if (overlap) {
        PH_HIP(aux.fork(caller_st)); // the profile table (and everything before this call) is ready
        joiner.armed = true;
    }
    uint64_t chunk_no = 0;
    for (uint64_t p0 = 0; p0 < npairs; p0 += chunk, ++chunk_no) {
        const uint64_t p1 = std::min(npairs, p0 + chunk);
        const unsigned blocks = (unsigned)((p1 - p0 + k3t::THREADS - 1) / k3t::THREADS);
        uint32_t *dirbuf = static_cast<uint32_t *>(d_dir);
        if (overlap) { // odd chunks: the library's stream and the upper half of the workspace
            st = (chunk_no & 1) ? aux.s : caller_st;
            if (chunk_no & 1)
                dirbuf = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_dir) + half_bytes);
        }
        if (use_pair16) {
#define PH_TBQ_LAUNCH(RB_)                                                                                             \
    do {                                                                                                               \
        auto kern = k3t::tb_pair16_kernel<RB_>;                                                                        \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)p.pair16_smem));                                                               \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k3t::THREADS), p.pair16_smem, st, d_A, d_offA, p0, p1, d_B, d_offB, \
                           sc->d_lutc, sc->d_codeA, sc->ncodes, (int)sc->gap, d_endA, d_endB, d_err, d_score,          \
                           (int)sc->smax, p.win.wcols, wide, p.pair16_nblk_alloc, dirbuf, d_alnA, d_alnB, d_alnLen,    \
                           aln_stride);                                                                                \
    } while (0)
            if (p.ra == 64)
                PH_TBQ_LAUNCH(32);
            else
                PH_TBQ_LAUNCH(76);
#undef PH_TBQ_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
        if (use_wave) {
            const unsigned wblocks = (unsigned)((p1 - p0 + k3t::THREADS / 64 - 1) / (k3t::THREADS / 64));
#define PH_TBW_LAUNCH(R_)                                                                                             \
    do {                                                                                                              \
        auto kern = k3t::tb_wave_kernel<R_>;                                                                          \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)p.smem));                                                                     \
        hipLaunchKernelGGL(kern, dim3(wblocks), dim3(k3t::THREADS), p.smem, st, d_A, d_offA, p0, p1, d_B, d_offB,     \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, d_endA, d_endB, d_err, d_score, \
                           (int)sc->smax, p.win.wcols, wide, dirbuf, d_alnA, d_alnB, d_alnLen, aln_stride);           \
    } while (0)
#define PH_TBW8_LAUNCH(R_)                                                                                            \
    do {                                                                                                              \
        auto kern = deferred ? k3t::tb_wave_kernel<R_, true, true> : k3t::tb_wave_kernel<R_, true, false>;            \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)p.wave8_smem));                                                               \
        hipLaunchKernelGGL(kern, dim3(wblocks), dim3(k3t::THREADS), p.wave8_smem, st, d_A, d_offA, p0, p1, d_B, d_offB, \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, d_endA, d_endB, d_err, d_score, \
                           (int)sc->smax, p.win.wcols, wide, dirbuf, d_alnA, d_alnB, d_alnLen, aln_stride);           \
    } while (0)
            if (use_wave8 && p.wave_r == 8)
                PH_TBW8_LAUNCH(8);
            else if (use_wave8)
                PH_TBW8_LAUNCH(16);
            else if (p.wave_r == 2)
                PH_TBW_LAUNCH(2);
            else if (p.wave_r == 3)
                PH_TBW_LAUNCH(3);
            else if (p.wave_r == 4)
                PH_TBW_LAUNCH(4);
            else if (p.wave_r == 8)
                PH_TBW_LAUNCH(8);
            else if (p.wave_r == 16)
                PH_TBW_LAUNCH(16);
            else if (p.wave_r == 32)
                PH_TBW_LAUNCH(32);
            else
                PH_TBW_LAUNCH(64);
#undef PH_TBW_LAUNCH
#undef PH_TBW8_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
        // (chunk = the pairs a (half) workspace holds: its direction words first, then the walk's 8 bytes per pair)
        uint2 *walkinfo = split_walk ? reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(dirbuf) + chunk * (per_pair - 8)) : nullptr;
        const unsigned wblocks16 = (unsigned)((p1 - p0 + 255) / 256);
        const bool walk_now = split_walk && !(wide & 4);
        if (use_half2) {
            const unsigned blocks2 = (unsigned)((p1 - p0 + k3t::THREADS / 2 - 1) / (k3t::THREADS / 2));
            auto kern = k3t::tb_prof16x2_kernel<64>;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)p.half_smem));
            hipLaunchKernelGGL(kern, dim3(blocks2), dim3(k3t::THREADS), p.half_smem, st, d_A, d_offA, p0, p1, d_B, p.lenB_pad,
                               reinterpret_cast<const uint2 *>(prof), sc->d_codeA, sc->ncodes, (int)sc->gap, d_endA, d_endB, d_err,
                               d_score, (int)sc->smax, p.win.wcols, wide, p.nblk_alloc2, dirbuf, d_alnA, d_alnB, d_alnLen,
                               aln_stride, walkinfo);
            if (walk_now)
                hipLaunchKernelGGL((k3t::tb_walk16_kernel<64, 2>), dim3(wblocks16), dim3(256), 0, st, d_A, d_offA, p0, p1, d_B,
                                   reinterpret_cast<const uint16_t *>(prof), sc->d_codeA, sc->ncodes, (int)sc->gap, d_endA, d_endB,
                                   d_score, p.nblk_alloc2, dirbuf, walkinfo, d_alnA, d_alnB, d_alnLen, aln_stride);
            PH_HIP(hipGetLastError());
            continue;
        }
        if (use_half) {
#define PH_TBH_LAUNCH(RB_)                                                                                             \
    do {                                                                                                               \
        auto kern = k3t::tb_prof16_kernel<RB_>;                                                                        \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)p.half_smem));                                                                 \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k3t::THREADS), p.half_smem, st, d_A, d_offA, p0, p1, d_B,          \
                           p.lenB_pad, reinterpret_cast<const uint2 *>(prof), sc->d_codeA, sc->ncodes, (int)sc->gap,   \
                           d_endA, d_endB, d_err, d_score, (int)sc->smax, p.win.wcols, wide, p.nblk_alloc, dirbuf,     \
                           d_alnA, d_alnB, d_alnLen, aln_stride, walkinfo);                                            \
        if (walk_now)                                                                                                  \
            hipLaunchKernelGGL((k3t::tb_walk16_kernel<RB_, 1>), dim3(wblocks16), dim3(256), 0, st, d_A, d_offA, p0, p1, d_B, \
                               reinterpret_cast<const uint16_t *>(prof), sc->d_codeA, sc->ncodes, (int)sc->gap, d_endA,  \
                               d_endB, d_score, p.nblk_alloc, dirbuf, walkinfo, d_alnA, d_alnB, d_alnLen, aln_stride);   \
    } while (0)
            if (p.ra == 64)
                PH_TBH_LAUNCH(32);
            else
                PH_TBH_LAUNCH(76);
#undef PH_TBH_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
        if (use_prof) {
#define PH_TBP_LAUNCH(RA_, CP_)                                                                                        \
    do {                                                                                                               \
        auto kern = k3t::tb_prof_kernel<RA_, CP_>;                                                                     \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)p.prof_smem));                                                                 \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k3t::THREADS), p.prof_smem, st, d_A, d_offA, p0, p1, d_B,           \
                           p.lenB_pad, prof, sc->d_codeA, sc->ncodes, (int)sc->gap, d_endA, d_endB, d_err, d_score,    \
                           (int)sc->smax, p.win.wcols, wide, p.nblk_alloc, dirbuf, d_alnA, d_alnB, d_alnLen, aln_stride);     \
    } while (0)
            if (p.ra == 64 && p.cp == 8)
                PH_TBP_LAUNCH(64, 8);
            else if (p.ra == 64)
                PH_TBP_LAUNCH(64, 32);
            else if (p.ra == 152 && p.cp == 8)
                PH_TBP_LAUNCH(152, 8);
            else if (p.ra == 152)
                PH_TBP_LAUNCH(152, 32);
            else if (p.cp == 8)
                PH_TBP_LAUNCH(256, 8);
            else
                PH_TBP_LAUNCH(256, 32);
#undef PH_TBP_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
#define PH_TB_LAUNCH(RA_)                                                                                              \
    do {                                                                                                               \
        auto kern = k3t::tb_kernel<RA_>;                                                                               \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)p.smem));                                                                      \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k3t::THREADS), p.smem, st, d_A, d_offA, p0, p1, d_B, d_offB, lenB,  \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, d_endA, d_endB, d_err,         \
                           d_score, (int)sc->smax, p.win.wcols, wide, dirbuf, d_alnA, d_alnB, d_alnLen, aln_stride);   \
    } while (0)
        if (p.ra == 64)
            PH_TB_LAUNCH(64);
        else if (p.ra == 152)
            PH_TB_LAUNCH(152);
        else if (p.ra == 256)
            PH_TB_LAUNCH(256);
        else {
            const size_t nl = (size_t)blocks * k3t::THREADS;
            int32_t *hbuf = static_cast<int32_t *>(d_dir);
            uint32_t *dirg = reinterpret_cast<uint32_t *>(hbuf + nl * max_lenA);
            hipLaunchKernelGGL(k3t::tb_generic_kernel, dim3(blocks), dim3(k3t::THREADS), 0, st, d_A, d_offA, p0, p1, d_B,
                               d_offB, lenB, sc->d_lut, (int)sc->gap, d_endA, d_endB, d_err, d_score, (int)sc->smax,
                               p.win.wcols, wide, max_lenA, hbuf,
                               dirg, d_alnA, d_alnB, d_alnLen, aln_stride);
        }
#undef PH_TB_LAUNCH
        PH_HIP(hipGetLastError());
    }
    if (overlap) { // the caller's stream continues after both
        joiner.armed = false;
        PH_HIP(aux.join(caller_st));
    }
    return POLYHIP_OK;
}

int polyhip_sw_traceback_dev(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                             uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB,
                             const uint32_t *d_endA, const uint32_t *d_endB, const uint32_t *d_err,
                             const int64_t *d_score, uint8_t *d_alnA,
                             uint8_t *d_alnB, uint32_t *d_alnLen, uint32_t aln_stride, void *d_work, size_t work_bytes,
                             polyhip_stream_t stream)
{
    // deferred = 0: the kernels never write the end arrays (a sentinel or an end row beyond the read means "no alignment")
    return traceback_impl(sc, d_A, d_offA, npairs, max_lenA, d_B, d_offB, lenB, const_cast<uint32_t *>(d_endA),
                          const_cast<uint32_t *>(d_endB), const_cast<uint32_t *>(d_err), d_score, d_alnA, d_alnB, d_alnLen,
                          aln_stride, d_work, work_bytes, stream, 0);
}

// The whole SmithWaterman on device pointers: score pass + traceback in one call.  For batches that take the packed
// score pass and the byte-profile traceback (config 4's shape) the score pass skips its locate step -- a second DP over
// the columns around each pair's maximum -- and the traceback kernel, which sweeps those columns anyway, finds the
// end cell in its last block.  Outputs as the two separate calls'.
int polyhip_sw_align_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                               uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB,
                               int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, uint8_t *d_alnA,
                               uint8_t *d_alnB, uint32_t *d_alnLen, uint32_t aln_stride, void *d_work, size_t work_bytes,
                               void *d_tb_work, size_t tb_work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(sc, "polyhip_sw_align_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    const int want_defer = d_offB == nullptr && d_B != nullptr && traceback_uses_prof(sc, max_lenA, lenB) &&
                           !env_is("POLYHIP_SW_FUSE", '0'); // testing aid: the two separate passes
    int deferred = 0;
    int rc = polyhip::k3::score_pass(sc, d_A, d_offA, npairs, max_lenA, d_B, d_offB, lenB, d_score, d_endA, d_endB, d_err, d_work,
                                     work_bytes, stream, want_defer, &deferred);
    if (rc != POLYHIP_OK)
        return rc;
    return traceback_impl(sc, d_A, d_offA, npairs, max_lenA, d_B, d_offB, lenB, d_endA, d_endB, d_err, d_score, d_alnA, d_alnB,
                          d_alnLen, aln_stride, d_tb_work, tb_work_bytes, stream, deferred);
}

// the single-device body: the calling thread's current device (a fan-out worker's, or the caller's own)
static int sw_align_batch_one(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                              const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA,
                              uint32_t *endB, uint32_t *err, uint8_t *alnA, uint8_t *alnB, uint32_t *alnLen,
                              uint32_t aln_stride)
{
    PH_REQUIRE(sc, "polyhip_sw_align_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(offA && score && endA && endB && err && alnA && alnB && alnLen, "polyhip_sw_align_batch: null pointer");
    if (!(sc = scoring_here(sc)))
        return POLYHIP_ERR_HIP;
    HostStreams &hs = host_streams(); // the calling thread's two streams carry the two slots (never the null stream)
    PH_HIP(hs.init());
    PairStage in;
    if (int rc0 = in.load("polyhip_sw_align_batch", A, offA, npairs, B, offB, lenB, hs.s[0])) {
        (void)hipStreamSynchronize(hs.s[0]);
        return rc0;
    }
    const uint64_t maxA = in.maxA, maxB = in.maxB;
    // Chunks of pairs through two slots, each with its own stream, outputs and workspaces: the strings of chunk c cross
    // PCIe (1 GB for config 4, as long as the kernels take) while chunk c + 1 is being aligned.  The reads and the
    // reference are on the device already (PairStage); a chunk is a window of offA.  Shared reference only -- per-pair
    // references keep the single shot.
    // Chunk = a multiple of 262,144 pairs (one full round of the packed pass: 512 workgroups of 512 pairs), at most
    // eight chunks, none when the strings are below ~200 MB.
    const uint64_t out_bytes = npairs * (2ull * aln_stride + 24);
    uint64_t per = npairs;
    if (!offB) {
        const uint64_t unit = 262144;
        if (out_bytes >= (192ull << 20) && npairs > unit)
            per = ((npairs + 7) / 8 + unit - 1) / unit * unit;
        if (const char *e = getenv("POLYHIP_SW_HOST_CHUNKS")) // testing aid: 0 / 1 = single shot, 2..8 = that many chunks
            if (e[0] >= '0' && e[0] <= '8') {
                const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(e[0] - '0'), npairs));
                per = (npairs + want - 1) / want;
            }
    }
    const uint64_t nchunks = (npairs + per - 1) / per;
    struct Slot {
        DevBuf dscore, dea, deb, derr, dwork, dalA, dalB, dlen, dtb;
        hipStream_t st = nullptr;
        ~Slot()
        {
            if (st)
                (void)hipStreamSynchronize(st); // the buffers are freed next
        }
    } slot[2];
    const size_t wb = polyhip_sw_workspace_bytes(sc, per, (uint32_t)maxA, maxB, offB == nullptr);
    const size_t tb = polyhip_sw_traceback_workspace_bytes(sc, per, (uint32_t)maxA, maxB);
    for (uint64_t q = 0; q < std::min<uint64_t>(2, nchunks); ++q) {
        Slot &S = slot[q];
        PH_HIP(S.dscore.alloc(per * 8));
        PH_HIP(S.dea.alloc(per * 4));
        PH_HIP(S.deb.alloc(per * 4));
        PH_HIP(S.derr.alloc(per * 4));
        PH_HIP(S.dlen.alloc(per * 4));
        PH_HIP(S.dalA.alloc(per * (size_t)aln_stride));
        PH_HIP(S.dalB.alloc(per * (size_t)aln_stride));
        PH_HIP(S.dwork.alloc(wb));
        PH_HIP(S.dtb.alloc(tb));
        S.st = hs.s[q];
    }
    PH_HIP(hipStreamSynchronize(hs.s[0])); // PairStage's uploads: both slots read them
    auto download = [&](uint64_t c) -> hipError_t {
        Slot &S = slot[c & 1];
        const uint64_t i0 = c * per, m = std::min(per, npairs - i0);
        hipError_t e;
        if ((e = hipMemcpyAsync(score + i0, S.dscore.p, m * 8, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(endA + i0, S.dea.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(endB + i0, S.deb.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(err + i0, S.derr.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(alnLen + i0, S.dlen.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            // whole slots in one contiguous copy each: a strided (2D) copy of only the columns the strings reach moves
            // a third of the bytes but runs row by row on this runtime (measured 14 s for config 4 against 40 ms)
            (e = hipMemcpyAsync(alnA + i0 * (size_t)aln_stride, S.dalA.p, m * (size_t)aln_stride, hipMemcpyDeviceToHost,
                                S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(alnB + i0 * (size_t)aln_stride, S.dalB.p, m * (size_t)aln_stride, hipMemcpyDeviceToHost,
                                S.st)) != hipSuccess)
            return e;
        return hipSuccess;
    };
    for (uint64_t c = 0; c < nchunks; ++c) {
        Slot &S = slot[c & 1];
        const uint64_t i0 = c * per, m = std::min(per, npairs - i0);
        PH_HIP(hipStreamSynchronize(S.st)); // chunk c - 2 has left this slot
        const int rc = polyhip_sw_align_batch_dev(sc, in.A(), in.offA() + i0, m, (uint32_t)maxA, in.B(), in.offB() ? in.offB() + i0 : nullptr,
                                                  maxB, S.dscore.as<int64_t>(), S.dea.as<uint32_t>(), S.deb.as<uint32_t>(),
                                                  S.derr.as<uint32_t>(), S.dalA.as<uint8_t>(), S.dalB.as<uint8_t>(),
                                                  S.dlen.as<uint32_t>(), aln_stride, S.dwork.p, wb, S.dtb.p, tb, S.st);
        if (rc != POLYHIP_OK)
            return rc;
        if (c > 0)
            PH_HIP(download(c - 1)); // the host waits here for chunk c - 1's copy while chunk c's kernels run
    }
    PH_HIP(download(nchunks - 1));
    for (uint64_t q = 0; q < std::min<uint64_t>(2, nchunks); ++q)
        PH_HIP(hipStreamSynchronize(slot[q].st));
    return POLYHIP_OK;
}

int polyhip_sw_align_batch(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                           const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA,
                           uint32_t *endB, uint32_t *err, uint8_t *alnA, uint8_t *alnB, uint32_t *alnLen,
                           uint32_t aln_stride)
{
    std::shared_ptr<md::Pool> P = npairs && sc ? md::pool() : nullptr;
    if (!P)
        return sw_align_batch_one(sc, A, offA, npairs, B, offB, lenB, score, endA, endB, err, alnA, alnB, alnLen, aln_stride);
    // pairs are independent; a shard's strings land in its own run of the caller's fixed-stride slots
    PH_REQUIRE(offA && score && endA && endB && err && alnA && alnB && alnLen, "polyhip_sw_align_batch: null pointer");
    const std::vector<uint64_t> cut = split_pairs(*P, offA, offB, npairs, 24 + 2ull * aln_stride);
    size_t first = 0;
    while (first + 1 < md::size(*P) && cut[first + 1] == cut[first])
        ++first;
    KernelChoice kc;
    const int rc = md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, 0);
        const int r = sw_align_batch_one(sc, A, offA + i0, m, B, offB ? offB + i0 : nullptr, lenB, score + i0, endA + i0, endB + i0,
                                         err + i0, alnA + i0 * (size_t)aln_stride, alnB + i0 * (size_t)aln_stride, alnLen + i0,
                                         aln_stride);
        if (q == first)
            kc = kernel_choice_get();
        return r;
    });
    kernel_choice_set(kc); // polyhip_sw_last_path & co. on the caller's thread: what the first shard's kernels were
    return rc;
}

// The same with PACKED strings: a pair's strings are a few hundred bytes of its aln_stride-byte slots (151 of 525 at
// config 4), and the slots are what crossed PCIe above (1.05 GB per 1M reads).  Here each chunk's strings are compacted
// on the device (scan of the lengths, 16 lanes per pair) and only the packed bytes travel (0.3 GB): alignA_p =
// alnA[alnOff[p] .. alnOff[p + 1]), alignB_p the same range of alnB (the two strings of a pair have one length).
static int sw_align_packed_one(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                               const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA,
                               uint32_t *endB, uint32_t *err, uint8_t *alnA, uint8_t *alnB, uint64_t *alnOff,
                               uint64_t aln_capacity)
{
    PH_REQUIRE(sc, "polyhip_sw_align_batch_packed: null scoring");
    PH_REQUIRE(alnOff, "polyhip_sw_align_batch_packed: null pointer");
    alnOff[0] = 0;
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(offA && score && endA && endB && err && (aln_capacity == 0 || (alnA && alnB)), "polyhip_sw_align_batch_packed: null pointer");
    if (!(sc = scoring_here(sc)))
        return POLYHIP_ERR_HIP;
    HostStreams &hs = host_streams(); // the calling thread's two streams carry the two slots (never the null stream)
    PH_HIP(hs.init());
    PairStage in;
    if (int rc0 = in.load("polyhip_sw_align_batch_packed", A, offA, npairs, B, offB, lenB, hs.s[0])) {
        (void)hipStreamSynchronize(hs.s[0]);
        return rc0;
    }
    const uint64_t maxA = in.maxA, maxB = in.maxB;
    const uint32_t aln_stride = polyhip_sw_traceback_stride(sc, (uint32_t)maxA, maxB);
    // Chunks of pairs through two slots, each with its own stream, outputs and workspaces: the strings of chunk c cross
    // PCIe (1 GB for config 4, as long as the kernels take) while chunk c + 1 is being aligned.  The reads and the
    // reference are on the device already (PairStage); a chunk is a window of offA.  Shared reference only -- per-pair
    // references keep the single shot.
    // Chunk = a multiple of 262,144 pairs (one full round of the packed pass: 512 workgroups of 512 pairs), at most
    // eight chunks, none when the strings are below ~200 MB.
    const uint64_t out_bytes = npairs * (2ull * aln_stride + 24);
    uint64_t per = npairs;
    if (!offB) {
        const uint64_t unit = 262144;
        if (out_bytes >= (192ull << 20) && npairs > unit)
            per = ((npairs + 7) / 8 + unit - 1) / unit * unit;
        if (const char *e = getenv("POLYHIP_SW_HOST_CHUNKS")) // testing aid: 0 / 1 = single shot, 2..8 = that many chunks
            if (e[0] >= '0' && e[0] <= '8') {
                const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(e[0] - '0'), npairs));
                per = (npairs + want - 1) / want;
            }
    }
    const uint64_t nchunks = (npairs + per - 1) / per;
    struct Slot {
        DevBuf dscore, dea, deb, derr, dwork, dalA, dalB, dlen, dtb, dpA, dpB, doff, dbsum;
        hipStream_t st = nullptr;
        ~Slot()
        {
            if (st)
                (void)hipStreamSynchronize(st); // the buffers are freed next
        }
    } slot[2];
    const size_t wb = polyhip_sw_workspace_bytes(sc, per, (uint32_t)maxA, maxB, offB == nullptr);
    const size_t tb = polyhip_sw_traceback_workspace_bytes(sc, per, (uint32_t)maxA, maxB);
    for (uint64_t q = 0; q < std::min<uint64_t>(2, nchunks); ++q) {
        Slot &S = slot[q];
        PH_HIP(S.dscore.alloc(per * 8));
        PH_HIP(S.dea.alloc(per * 4));
        PH_HIP(S.deb.alloc(per * 4));
        PH_HIP(S.derr.alloc(per * 4));
        PH_HIP(S.dlen.alloc(per * 4));
        PH_HIP(S.dalA.alloc(per * (size_t)aln_stride));
        PH_HIP(S.dalB.alloc(per * (size_t)aln_stride));
        PH_HIP(S.dwork.alloc(wb));
        PH_HIP(S.dtb.alloc(tb));
        PH_HIP(S.dpA.alloc(per * (size_t)aln_stride));
        PH_HIP(S.dpB.alloc(per * (size_t)aln_stride));
        PH_HIP(S.doff.alloc((per + 1) * 8));
        PH_HIP(S.dbsum.alloc(((per + k3t::PACK_BLOCK - 1) / k3t::PACK_BLOCK + 2) * 8));
        S.st = hs.s[q];
    }
    PH_HIP(hipStreamSynchronize(hs.s[0])); // PairStage's uploads: both slots read them
    uint64_t base = 0;     // packed bytes of the chunks finished so far
    bool overflow = false; // the caller's string buffers are too small: offsets and scores still complete
    // chunk c's results -> host: its total first (the one thing the host has to wait for), then exactly that many bytes
    auto finish = [&](uint64_t c, uint64_t cbase, uint64_t &total) -> hipError_t {
        Slot &S = slot[c & 1];
        const uint64_t i0 = c * per, m = std::min(per, npairs - i0);
        hipError_t e;
        uint64_t last = 0;
        if ((e = hipMemcpyAsync(&last, S.doff.as<uint64_t>() + m, 8, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipStreamSynchronize(S.st)) != hipSuccess)
            return e;
        total = last - cbase;
        if ((e = hipMemcpyAsync(score + i0, S.dscore.p, m * 8, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(endA + i0, S.dea.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(endB + i0, S.deb.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(err + i0, S.derr.p, m * 4, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
            (e = hipMemcpyAsync(alnOff + i0, S.doff.p, (m + 1) * 8, hipMemcpyDeviceToHost, S.st)) != hipSuccess)
            return e;
        if (cbase + total > aln_capacity) {
            overflow = true;
            return hipSuccess;
        }
        if (total && ((e = hipMemcpyAsync(alnA + cbase, S.dpA.p, total, hipMemcpyDeviceToHost, S.st)) != hipSuccess ||
                      (e = hipMemcpyAsync(alnB + cbase, S.dpB.p, total, hipMemcpyDeviceToHost, S.st)) != hipSuccess))
            return e;
        return hipSuccess;
    };
    for (uint64_t c = 0; c < nchunks; ++c) {
        Slot &S = slot[c & 1];
        const uint64_t i0 = c * per, m = std::min(per, npairs - i0);
        PH_HIP(hipStreamSynchronize(S.st)); // chunk c - 2 has left this slot
        const int rc = polyhip_sw_align_batch_dev(sc, in.A(), in.offA() + i0, m, (uint32_t)maxA, in.B(), in.offB() ? in.offB() + i0 : nullptr,
                                                  maxB, S.dscore.as<int64_t>(), S.dea.as<uint32_t>(), S.deb.as<uint32_t>(),
                                                  S.derr.as<uint32_t>(), S.dalA.as<uint8_t>(), S.dalB.as<uint8_t>(),
                                                  S.dlen.as<uint32_t>(), aln_stride, S.dwork.p, wb, S.dtb.p, tb, S.st);
        if (rc != POLYHIP_OK) {
            (void)hs.sync_both();
            return rc;
        }
        if (c > 0) { // the host waits here for chunk c - 1's total while chunk c's kernels run
            uint64_t t = 0;
            PH_HIP(finish(c - 1, base, t));
            base += t;
        }
        // compact chunk c's strings behind everything before it
        const unsigned nb = (unsigned)((m + k3t::PACK_BLOCK - 1) / k3t::PACK_BLOCK);
        hipLaunchKernelGGL(k3t::pack_sums_kernel, dim3(nb), dim3(256), 0, S.st, S.dlen.as<uint32_t>(), m, S.dbsum.as<uint64_t>());
        hipLaunchKernelGGL(k3t::pack_scan_kernel, dim3(1), dim3(1024), 0, S.st, S.dbsum.as<uint64_t>(), nb);
        hipLaunchKernelGGL(k3t::pack_copy_kernel, dim3(nb), dim3(256), 0, S.st, S.dlen.as<uint32_t>(), m, S.dbsum.as<uint64_t>(), base,
                           S.dalA.as<uint8_t>(), S.dalB.as<uint8_t>(), aln_stride, S.doff.as<uint64_t>(), S.dpA.as<uint8_t>(),
                           S.dpB.as<uint8_t>());
        PH_HIP(hipGetLastError());
    }
    {
        uint64_t t = 0;
        PH_HIP(finish(nchunks - 1, base, t));
        base += t;
    }
    PH_HIP(hs.sync_both());
    if (overflow)
        return set_error(POLYHIP_ERR_INVALID,
                         "polyhip_sw_align_batch_packed: the strings need %llu bytes per buffer, aln_capacity is %llu (scores, ends and "
                         "alnOff are complete: call again with buffers of alnOff[npairs] bytes)",
                         (unsigned long long)base, (unsigned long long)aln_capacity);
    return POLYHIP_OK;
}

// One shard of the packed call on a device list.  Where a shard's strings start in the caller's buffers depends on the
// totals of all shards before it, so the call has two rounds.  Round A (all shards side by side): upload, align chunk
// after chunk, compact every chunk's strings behind the previous one's in a buffer that STAYS on the device (the running
// total is read from the device: no host wait between chunks), download scores / ends / errors to their final places
// and the shard-local offsets to a private vector.  The caller turns the totals into bases.  Round B: the strings go to
// alnA / alnB + base, the offsets to alnOff + base.
struct PackedShard {
    uint64_t i0 = 0, m = 0, total = 0;
    DevBuf dpA, dpB;
    std::vector<uint64_t> hoff;
};

static int packed_shard_align(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, const uint8_t *B, const uint64_t *offB,
                              uint64_t lenB, int64_t *score, uint32_t *endA, uint32_t *endB, uint32_t *err, PackedShard &sh)
{
    const uint64_t npairs = sh.m;
    sh.hoff.assign(npairs + 1, 0);
    sh.total = 0;
    if (npairs == 0)
        return POLYHIP_OK;
    if (!(sc = scoring_here(sc)))
        return POLYHIP_ERR_HIP;
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    hipStream_t st = hs.s[0];
    PairStage in;
    if (int rc0 = in.load("polyhip_sw_align_batch_packed", A, offA, npairs, B, offB, lenB, st))
        return rc0;
    const uint64_t maxA = in.maxA, maxB = in.maxB;
    const uint32_t aln_stride = polyhip_sw_traceback_stride(sc, (uint32_t)maxA, maxB);
    // chunks bound the slots and the workspaces, not the transfers: 262,144 pairs is one full round of the packed pass
    const uint64_t per = std::min<uint64_t>(npairs, 262144);
    DevBuf dscore, dea, deb, derr, doff, dwork, dalA, dalB, dlen, dtb, dbsum;
    PH_HIP(dscore.alloc(npairs * 8));
    PH_HIP(dea.alloc(npairs * 4));
    PH_HIP(deb.alloc(npairs * 4));
    PH_HIP(derr.alloc(npairs * 4));
    PH_HIP(doff.alloc((npairs + 1) * 8));
    PH_HIP(sh.dpA.alloc(npairs * (size_t)aln_stride)); // upper bound: a pair's strings never exceed its slot
    PH_HIP(sh.dpB.alloc(npairs * (size_t)aln_stride));
    const size_t wb = polyhip_sw_workspace_bytes(sc, per, (uint32_t)maxA, maxB, offB == nullptr);
    const size_t tb = polyhip_sw_traceback_workspace_bytes(sc, per, (uint32_t)maxA, maxB);
    PH_HIP(dwork.alloc(wb));
    PH_HIP(dtb.alloc(tb));
    PH_HIP(dalA.alloc(per * (size_t)aln_stride));
    PH_HIP(dalB.alloc(per * (size_t)aln_stride));
    PH_HIP(dlen.alloc(per * 4));
    PH_HIP(dbsum.alloc(((per + k3t::PACK_BLOCK - 1) / k3t::PACK_BLOCK + 2) * 8));
    PH_HIP(hipMemsetAsync(doff.p, 0, 8, st)); // the first chunk's base
    for (uint64_t i0 = 0; i0 < npairs; i0 += per) {
        const uint64_t m = std::min(per, npairs - i0);
        const int rc = polyhip_sw_align_batch_dev(sc, in.A(), in.offA() + i0, m, (uint32_t)maxA, in.B(), in.offB() ? in.offB() + i0 : nullptr,
                                                  maxB, dscore.as<int64_t>() + i0, dea.as<uint32_t>() + i0, deb.as<uint32_t>() + i0,
                                                  derr.as<uint32_t>() + i0, dalA.as<uint8_t>(), dalB.as<uint8_t>(), dlen.as<uint32_t>(),
                                                  aln_stride, dwork.p, wb, dtb.p, tb, st);
        if (rc != POLYHIP_OK) {
            (void)hipStreamSynchronize(st);
            return rc;
        }
        const unsigned nb = (unsigned)((m + k3t::PACK_BLOCK - 1) / k3t::PACK_BLOCK);
        hipLaunchKernelGGL(k3t::pack_sums_kernel, dim3(nb), dim3(256), 0, st, dlen.as<uint32_t>(), m, dbsum.as<uint64_t>());
        hipLaunchKernelGGL(k3t::pack_scan_kernel, dim3(1), dim3(1024), 0, st, dbsum.as<uint64_t>(), nb);
        hipLaunchKernelGGL(k3t::pack_copy_kernel, dim3(nb), dim3(256), 0, st, dlen.as<uint32_t>(), m, dbsum.as<uint64_t>(), (uint64_t)0,
                           dalA.as<uint8_t>(), dalB.as<uint8_t>(), aln_stride, doff.as<uint64_t>() + i0, sh.dpA.as<uint8_t>(),
                           sh.dpB.as<uint8_t>(), doff.as<uint64_t>() + i0);
        PH_HIP(hipGetLastError());
    }
    PH_HIP(hipMemcpyAsync(score, dscore.p, npairs * 8, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(endA, dea.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(endB, deb.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(err, derr.p, npairs * 4, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(sh.hoff.data(), doff.p, (npairs + 1) * 8, hipMemcpyDeviceToHost, st));
    PH_HIP(hipStreamSynchronize(st));
    sh.total = sh.hoff[npairs];
    return POLYHIP_OK;
}

static int packed_shard_deliver(PackedShard &sh, uint64_t base, bool fits, uint8_t *alnA, uint8_t *alnB, uint64_t *alnOff)
{
    for (uint64_t j = 1; j <= sh.m; ++j) // alnOff[i0] is the previous shard's last entry (or the caller's 0)
        alnOff[sh.i0 + j] = base + sh.hoff[j];
    if (fits && sh.total) {
        HostStreams &hs = host_streams();
        PH_HIP(hs.init());
        PH_HIP(hipMemcpyAsync(alnA + base, sh.dpA.p, sh.total, hipMemcpyDeviceToHost, hs.s[0]));
        PH_HIP(hipMemcpyAsync(alnB + base, sh.dpB.p, sh.total, hipMemcpyDeviceToHost, hs.s[1]));
        PH_HIP(hs.sync_both());
    }
    return POLYHIP_OK;
}

int polyhip_sw_align_batch_packed(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                                  const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *endA,
                                  uint32_t *endB, uint32_t *err, uint8_t *alnA, uint8_t *alnB, uint64_t *alnOff,
                                  uint64_t aln_capacity)
{
    std::shared_ptr<md::Pool> P = npairs && sc && alnOff ? md::pool() : nullptr;
    if (!P)
        return sw_align_packed_one(sc, A, offA, npairs, B, offB, lenB, score, endA, endB, err, alnA, alnB, alnOff, aln_capacity);
    PH_REQUIRE(offA && score && endA && endB && err && (aln_capacity == 0 || (alnA && alnB)), "polyhip_sw_align_batch_packed: null pointer");
    alnOff[0] = 0;
    const size_t nsh = md::size(*P);
    const std::vector<uint64_t> cut = split_pairs(*P, offA, offB, npairs, 24 + 8 + 2 * 160);
    std::vector<PackedShard> sh(nsh);
    size_t first = 0;
    while (first + 1 < nsh && cut[first + 1] == cut[first])
        ++first;
    KernelChoice kc;
    int rc = md::run(*P, [&](size_t q) {
        sh[q].i0 = cut[q];
        sh[q].m = cut[q + 1] - cut[q];
        const uint64_t i0 = sh[q].i0;
        md::BaseScope pos(i0, 0);
        const int r = packed_shard_align(sc, A, offA + i0, B, offB ? offB + i0 : nullptr, lenB, score + i0, endA + i0, endB + i0,
                                         err + i0, sh[q]);
        if (q == first)
            kc = kernel_choice_get();
        return r;
    });
    kernel_choice_set(kc);
    if (rc != POLYHIP_OK)
        return rc; // the shards' device buffers go with `sh`
    std::vector<uint64_t> base(nsh + 1, 0);
    for (size_t q = 0; q < nsh; ++q)
        base[q + 1] = base[q] + sh[q].total;
    const bool fits = base[nsh] <= aln_capacity;
    rc = md::run(*P, [&](size_t q) {
        const int r = packed_shard_deliver(sh[q], base[q], fits, alnA, alnB, alnOff);
        sh[q].dpA.reset(); // freed on the device's own worker
        sh[q].dpB.reset();
        return r;
    });
    if (rc != POLYHIP_OK)
        return rc;
    if (!fits)
        return set_error(POLYHIP_ERR_INVALID,
                         "polyhip_sw_align_batch_packed: the strings need %llu bytes per buffer, aln_capacity is %llu (scores, ends and "
                         "alnOff are complete: call again with buffers of alnOff[npairs] bytes)",
                         (unsigned long long)base[nsh], (unsigned long long)aln_capacity);
    return POLYHIP_OK;
}


} // extern "C"


extern "C" {

int polyhip_nw_last_path(void) { return k3t::g_nw_last_path; }

size_t polyhip_nw_workspace_bytes(uint64_t npairs, uint32_t max_lenA, uint64_t max_lenB)
{
    const uint64_t per_pair = k3t::nw_per_pair(max_lenA, max_lenB);
    const uint64_t padded = (npairs + k3t::THREADS - 1) / k3t::THREADS * k3t::THREADS;
    uint64_t want = padded * per_pair;
    const uint64_t cap = 8ull << 30, floor_ = (uint64_t)k3t::THREADS * per_pair;
    if (want > cap)
        want = std::max(cap / floor_, (uint64_t)1) * floor_;
    return (size_t)want + 256;
}

int polyhip_nw_align_batch_dev(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs,
                               uint32_t max_lenA, const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB,
                               int64_t *d_score, uint32_t *d_err, uint8_t *d_alnA, uint8_t *d_alnB, uint32_t *d_alnLen,
                               uint32_t aln_stride, void *d_work, size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(sc, "polyhip_nw_align_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_offA && d_score && d_err && d_alnA && d_alnB && d_alnLen && d_work, "polyhip_nw_align_batch: null pointer");
    PH_REQUIRE((uint64_t)aln_stride >= (uint64_t)max_lenA + lenB || aln_stride == 0xFFFFFFFFu,
               "polyhip_nw_align_batch: aln_stride %u < max_lenA + lenB", aln_stride);
    if ((int64_t)sc->absmax * (int64_t)((uint64_t)max_lenA + lenB) >= (1ll << 31))
        return set_error(POLYHIP_ERR_UNSUPPORTED, "polyhip_nw_align_batch: scores could overflow int32");
    const uint64_t per_pair = k3t::nw_per_pair(max_lenA, lenB);
    const uint64_t chunk = (work_bytes & ~(size_t)255) / per_pair / k3t::THREADS * k3t::THREADS;
    PH_REQUIRE(chunk >= (uint64_t)k3t::THREADS, "polyhip_nw_align_batch: workspace too small (%zu B; %llu B per pair, >= %d pairs)",
               work_bytes, (unsigned long long)per_pair, k3t::THREADS);
    hipStream_t st = as_stream(stream);
    // register-tiled kernel: lenA <= 64, the compact table fits LDS, columns below 2^31; else the generic one
    const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
    const size_t reg_smem = (size_t)na * nb * 4 + 512;
    const bool nw_generic = env_is("POLYHIP_NW_GENERIC", '1'); // testing aid: force the generic kernel
    const int reg_ra = (reg_smem <= 60 * 1024 && (size_t)na * nb < 65536 && lenB < (1ull << 31) && max_lenA > 0 &&
                        lenB > 0 && !nw_generic)
                           ? k3t::nw_ra(max_lenA)
                           : 0;
    const int wave_r = (reg_ra == 0 && reg_smem <= 60 * 1024 && (size_t)na * nb < 65536 && lenB > 0 &&
                        lenB < (1ull << 31) - 64 && !nw_generic)
                           ? k3t::nw_wave_r(max_lenA)
                           : 0;
    k3t::g_nw_last_path = reg_ra ? 1 : wave_r ? 3 : 2;
    for (uint64_t p0 = 0; p0 < npairs; p0 += chunk) {
        const uint64_t p1 = std::min(npairs, p0 + chunk);
        const unsigned blocks = (unsigned)((p1 - p0 + k3t::THREADS - 1) / k3t::THREADS);
        if (wave_r) {
            const unsigned wblocks = (unsigned)((p1 - p0 + k3t::THREADS / 64 - 1) / (k3t::THREADS / 64));
#define PH_NWW_LAUNCH(R_)                                                                                             \
    do {                                                                                                              \
        auto kern = k3t::nw_wave_kernel<R_>;                                                                          \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)reg_smem));                                                                   \
        hipLaunchKernelGGL(kern, dim3(wblocks), dim3(k3t::THREADS), reg_smem, st, d_A, d_offA, p0, p1, d_B, d_offB, lenB, \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, (uint32_t)lenB,               \
                           static_cast<uint32_t *>(d_work), d_score, d_err, d_alnA, d_alnB, d_alnLen, aln_stride);    \
    } while (0)
            if (wave_r == 2)
                PH_NWW_LAUNCH(2);
            else if (wave_r == 3)
                PH_NWW_LAUNCH(3);
            else if (wave_r == 4)
                PH_NWW_LAUNCH(4);
            else if (wave_r == 8)
                PH_NWW_LAUNCH(8);
            else if (wave_r == 16)
                PH_NWW_LAUNCH(16);
            else if (wave_r == 32)
                PH_NWW_LAUNCH(32);
            else
                PH_NWW_LAUNCH(64);
#undef PH_NWW_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
        if (reg_ra) {
#define PH_NW_LAUNCH(RA_)                                                                                             \
    do {                                                                                                              \
        auto kern = k3t::nw_reg_kernel<RA_>;                                                                          \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)reg_smem));                                                                   \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(k3t::THREADS), reg_smem, st, d_A, d_offA, p0, p1, d_B, d_offB, lenB, \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, (uint32_t)lenB,               \
                           static_cast<uint32_t *>(d_work), d_score, d_err, d_alnA, d_alnB, d_alnLen, aln_stride);    \
    } while (0)
            PH_NW_LAUNCH(64);
#undef PH_NW_LAUNCH
            PH_HIP(hipGetLastError());
            continue;
        }
        const size_t nl = (size_t)blocks * k3t::THREADS;
        int32_t *hbuf = static_cast<int32_t *>(d_work);
        uint32_t *dirg = reinterpret_cast<uint32_t *>(hbuf + nl * (max_lenA ? max_lenA : 1));
        hipLaunchKernelGGL(k3t::nw_kernel, dim3(blocks), dim3(k3t::THREADS), 0, st, d_A, d_offA, p0, p1, d_B, d_offB, lenB,
                           sc->d_lut, sc->d_validA, sc->d_validB, (int)sc->gap, max_lenA, hbuf, dirg, d_score, d_err, d_alnA,
                           d_alnB, d_alnLen, aln_stride);
        PH_HIP(hipGetLastError());
    }
    return POLYHIP_OK;
}

static int nw_align_batch_one(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                              const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *err,
                              uint8_t *alnA, uint8_t *alnB, uint32_t *alnLen, uint32_t aln_stride)
{
    PH_REQUIRE(sc, "polyhip_nw_align_batch: null scoring");
    if (npairs == 0)
        return POLYHIP_OK;
    PH_REQUIRE(offA && score && err && alnA && alnB && alnLen, "polyhip_nw_align_batch: null pointer");
    if (!(sc = scoring_here(sc)))
        return POLYHIP_ERR_HIP;
    HostStreams &hs = host_streams(); // the calling thread's own stream, not the null stream
    PH_HIP(hs.init());
    hipStream_t hst = hs.s[0];
    PairStage in;
    if (int rc0 = in.load("polyhip_nw_align_batch", A, offA, npairs, B, offB, lenB, hst)) {
        (void)hipStreamSynchronize(hst);
        return rc0;
    }
    const uint64_t maxA = in.maxA, maxB = in.maxB;
    PH_REQUIRE(maxB < 0xFFFFFFFFull, "polyhip_nw_align_batch: sequence longer than 2^32");
    DevBuf dscore, derr, dalA, dalB, dlen, dwork;
    PH_HIP(dscore.alloc(npairs * 8));
    PH_HIP(derr.alloc(npairs * 4));
    PH_HIP(dalA.alloc(npairs * (size_t)aln_stride + 1));
    PH_HIP(dalB.alloc(npairs * (size_t)aln_stride + 1));
    PH_HIP(dlen.alloc(npairs * 4));
    const size_t wb = polyhip_nw_workspace_bytes(npairs, (uint32_t)maxA, maxB);
    PH_HIP(dwork.alloc(wb));
    int rc = polyhip_nw_align_batch_dev(sc, in.A(), in.offA(), npairs, (uint32_t)maxA, in.B(), in.offB(), maxB,
                                        dscore.as<int64_t>(), derr.as<uint32_t>(),
                                        dalA.as<uint8_t>(), dalB.as<uint8_t>(), dlen.as<uint32_t>(), aln_stride, dwork.p, wb,
                                        hst);
    if (rc != POLYHIP_OK) {
        (void)hipStreamSynchronize(hst);
        return rc;
    }
    PH_HIP(hipMemcpyAsync(score, dscore.p, npairs * 8, hipMemcpyDeviceToHost, hst));
    PH_HIP(hipMemcpyAsync(err, derr.p, npairs * 4, hipMemcpyDeviceToHost, hst));
    if (aln_stride) { // the two string planes side by side on the thread's two streams
        PH_HIP(hipEventRecord(hs.ev, hst));
        PH_HIP(hipStreamWaitEvent(hs.s[1], hs.ev, 0));
        PH_HIP(hipMemcpyAsync(alnA, dalA.p, npairs * (size_t)aln_stride, hipMemcpyDeviceToHost, hst));
        PH_HIP(hipMemcpyAsync(alnB, dalB.p, npairs * (size_t)aln_stride, hipMemcpyDeviceToHost, hs.s[1]));
    }
    PH_HIP(hipMemcpyAsync(alnLen, dlen.p, npairs * 4, hipMemcpyDeviceToHost, hst));
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

int polyhip_nw_align_batch(const polyhip_scoring *sc, const uint8_t *A, const uint64_t *offA, uint64_t npairs,
                           const uint8_t *B, const uint64_t *offB, uint64_t lenB, int64_t *score, uint32_t *err,
                           uint8_t *alnA, uint8_t *alnB, uint32_t *alnLen, uint32_t aln_stride)
{
    std::shared_ptr<md::Pool> P = npairs && sc ? md::pool() : nullptr;
    if (!P)
        return nw_align_batch_one(sc, A, offA, npairs, B, offB, lenB, score, err, alnA, alnB, alnLen, aln_stride);
    PH_REQUIRE(offA && score && err && alnA && alnB && alnLen, "polyhip_nw_align_batch: null pointer");
    const std::vector<uint64_t> cut = split_pairs(*P, offA, offB, npairs, 16 + 2ull * aln_stride);
    size_t first = 0;
    while (first + 1 < md::size(*P) && cut[first + 1] == cut[first])
        ++first;
    KernelChoice kc;
    const int rc = md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, 0);
        const int r = nw_align_batch_one(sc, A, offA + i0, m, B, offB ? offB + i0 : nullptr, lenB, score + i0, err + i0,
                                         alnA + i0 * (size_t)aln_stride, alnB + i0 * (size_t)aln_stride, alnLen + i0, aln_stride);
        if (q == first)
            kc = kernel_choice_get();
        return r;
    });
    kernel_choice_set(kc);
    return rc;
}

} // extern "C"
