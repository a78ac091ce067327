// sw_wave.hip -- K3, low-latency exact Smith-Waterman score pass: ONE WAVE PER PAIR.
//
// The lane-per-pair kernels (sw_batch.hip, sw_packed.hip) need thousands of pairs to fill the chip and
// take the time of one full DP (10-20 ms at 150 x 5000) no matter how few pairs there are.  This kernel
// serves the other end: the handful of pairs the packed pass leaves on its tie list, small batches (a single
// align.SmithWaterman call is a batch of one), reads longer than the 256 rows a lane can hold in registers
// (up to 4096, shared or per-pair B).  Same recurrence and argmax as
// search/align/align.go:171-203, same outputs as sw_shared_kernel.
//
// Systolic sweep: lane l owns rows [l*R, l*R + R) of the pair (R = 1..64 for lenA <= 64..4096) and
// works on column j = s - l in step s, so the row above its first row (lane l-1's last row, same
// column) was finished one step earlier and arrives, together with that column's B code, by one
// lane shift per step.  lenB + 63 steps of R cells per lane; S(a, b) from the compact int32 table in LDS.
// Argmax: per lane the best (h, then smaller row, then smaller column), folded across lanes at the
// end -- the first maximum in row-major order (align.go:197-201, strict `>`).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "sw_scoring.h"

namespace polyhip {
namespace k3w {

constexpr int THREADS = 256; // 4 waves = 4 pairs per workgroup

template <int R>
__global__ __launch_bounds__(THREADS) void sw_wave_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ Bbase,
    const uint64_t *__restrict__ offB, uint32_t lenB_shared, const uint8_t *__restrict__ codeA,
    const uint8_t *__restrict__ codeB,
    const int32_t *__restrict__ lutcc, int na, int nb, int gap, const uint32_t *__restrict__ binfo,
    const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, const uint32_t *__restrict__ infoM,
    const uint32_t *__restrict__ infoQ, int smax, int64_t *__restrict__ score,
    uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err, int m0_only)
{
    extern __shared__ __attribute__((aligned(16))) int32_t T[]; // [na][nb] (last row / column: pad, zeros), codeA, codeB
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();

    const uint64_t total = list ? (uint64_t)*count : npairs;
    // grid-stride over work items, one per wave (wave-uniform control flow, no barrier below)
    for (uint64_t w = (uint64_t)blockIdx.x * (THREADS / 64) + (tid >> 6); w < total;
         w += (uint64_t)gridDim.x * (THREADS / 64)) {
    const uint64_t pair = list ? (uint64_t)list[w] : w;
    if (m0_only && infoM[pair] != 0u)
        continue; // sw_wave8_kernel located this pair's maximum

    // shared reference, or this pair's own B
    const uint8_t *B = offB ? Bbase + offB[pair] : Bbase;
    const uint32_t lenB = offB ? (uint32_t)(offB[pair + 1] - offB[pair]) : lenB_shared;
    const uint64_t o0 = offA[pair];
    const uint64_t l64 = offA[pair + 1] - o0;
    const bool too_long = l64 > (uint64_t)(64 * R);
    const uint32_t lenA = too_long ? 0u : (uint32_t)l64;
    const uint8_t *ap = A + o0;

    // my rows' table offsets (code * nb, in bytes); the first byte outside FirstAlphabet, wave-wide
    uint32_t ro[R];
    uint32_t mybad = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t r = (uint32_t)lane * R + k;
        uint32_t code = (uint32_t)(na - 1);
        if (r < lenA) {
            const uint32_t c = cA[ap[r]];
            if (c == 0xFFu)
                mybad = min(mybad, r);
            else
                code = c;
        }
        ro[k] = code * (uint32_t)nb * 4u; // byte offset of the row in T: the cell's address is one full-rate add
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        mybad = min(mybad, (uint32_t)__shfl_xor((int)mybad, d, 64));
    uint32_t e = 0;
    if (too_long) {
        e = 0xFFFFFFFFu;
    } else if (lenA > 0 && lenB > 0) { // align.go:189-191 + matrix.go:29-36: row-major first failing cell
        uint32_t bbad = 0xFFFFFFFFu; // first byte of B outside SecondAlphabet
        if (offB || !binfo) {
            for (uint32_t j0 = 0; j0 < lenB && bbad == 0xFFFFFFFFu; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane;
                const uint64_t bad = __ballot(j < lenB && cB[B[j]] == 0xFFu);
                if (bad)
                    bbad = j0 + (uint32_t)__builtin_ctzll(bad);
            }
        } else {
            bbad = binfo[0];
        }
        if (mybad == 0u)
            e = (1u << 8) | ap[0];
        else if (bbad != 0xFFFFFFFFu)
            e = (2u << 8) | B[bbad];
        else if (mybad != 0xFFFFFFFFu)
            e = (1u << 8) | ap[mybad];
    }

    // Locate mode (infoM / infoQ from the packed banded pass, sw_packed.hip): the maximum M and the only 4-column
    // block that reaches it are known, so only the columns that can feed a cell worth M there are swept --
    // lenA + (smax*lenA - M)/|gap| + 4 of them, as sw_locate_kernel.  A windowed H never exceeds the true H and
    // equals it wherever the true value is M, so the first maximum of the window is the first maximum of the matrix.
    // A pair whose maximum turned up in several blocks (tie bit) or that has no maximum (M = 0) is swept in full.
    uint32_t j0 = 0, ncols = lenB;
    if (infoM) {
        const uint32_t M = infoM[pair], iq = infoQ[pair];
        // (round 6) a NEAR tie -- every block worth M within `span` < 255 blocks of the first (infoQ bits 16..23; exact while the
        // reference has at most 32,768 blocks) -- stretches the window over first..last instead of the whole reference: all
        // cells worth M are inside it and computed exactly, and the first maximum of the window is the matrix's
        const uint32_t span = (iq >> 31) ? (iq >> 16) & 0xFFu : 0u;
        if (M != 0u && ((iq >> 31) == 0u || (span < 255u && lenB <= 131072u))) { // M == 0: nothing positive, or a read the packed pass did not take -- full sweep
            const uint32_t g = (uint32_t)(-gap), top = (uint32_t)smax * lenA;
            const uint32_t need = lenA + (top > M ? (top - M) / g : 0u) + 4u + 4u * span;
            const uint32_t jend = min(4u * ((iq & 0xFFFFu) + span) + 4u, lenB);
            j0 = jend > need ? jend - need : 0u;
            ncols = jend - j0;
        }
    }

    int Hrow[R];
#pragma unroll
    for (int k = 0; k < R; ++k)
        Hrow[k] = 0;
    int topprev = 0, last_h = 0;
    uint32_t last_b = (uint32_t)(nb - 1);
    int besth = 0;
    uint32_t besti = 0, bestj = 0;
    // up to 16 rows per lane: the best cell of every row (columns come in order, so strict > keeps the first), folded
    // over rows and lanes at the end -- a compare and two selects per cell instead of the full (h, row) comparison
    constexpr bool ROWBEST = R <= 16;
    int bh[ROWBEST ? R : 1];
    uint32_t bj[ROWBEST ? R : 1];
#pragma unroll
    for (int k = 0; k < (ROWBEST ? R : 1); ++k) {
        bh[k] = 0;
        bj[k] = 0;
    }
    const uint32_t steps = (e == 0u && lenA > 0 && ncols > 0) ? ncols + 63u : 0u;
    // B codes enter at lane 0, 64 columns per coalesced load, the next chunk in flight while this one is used
    auto load_chunk = [&](uint32_t s0) -> uint32_t {
        uint32_t c = (uint32_t)(nb - 1);
        if (s0 + (uint32_t)lane < ncols) {
            const uint32_t cc = cB[B[j0 + s0 + (uint32_t)lane]];
            c = cc == 0xFFu ? (uint32_t)(nb - 1) : cc;
        }
        return c;
    };
    uint32_t chunk = steps ? load_chunk(0) : 0u, next_chunk = 0u;
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
        const uint32_t j = s - (uint32_t)lane; // column inside the window; wraps for lanes that have not started
        const bool valid = j < ncols;
        int diag = topprev, up = top_in;
        const uint32_t b4 = b_in << 2;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int left = Hrow[k];
            const int sc = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(T) + (ro[k] + b4));
            int h = max(max(diag + sc, 0), max(up, left) + gap);
            h = valid ? h : 0;
            if constexpr (ROWBEST) {
                const bool better = h > bh[k];
                bh[k] = better ? h : bh[k];
                bj[k] = better ? j : bj[k];
            } else {
                const uint32_t r = (uint32_t)lane * R + k;
                // first maximum in row-major order: higher h, else smaller row (columns come in order)
                const bool better = (r < lenA) & ((h > besth) | ((h == besth) & (h > 0) & (r < besti)));
                besth = better ? h : besth;
                besti = better ? r : besti;
                bestj = better ? j : bestj;
            }
            diag = left;
            up = h;
            Hrow[k] = h;
        }
        topprev = valid ? top_in : 0;
        last_h = Hrow[R - 1];
        last_b = b_in;
    }
    if constexpr (ROWBEST) {
#pragma unroll
        for (int k = 0; k < R; ++k) { // my rows in order: strict > keeps the smaller row
            const uint32_t r = (uint32_t)lane * R + k;
            if (r < lenA && bh[k] > besth) {
                besth = bh[k];
                besti = r;
                bestj = bj[k];
            }
        }
    }
    bestj += j0; // window-relative so far
    // fold the lanes: max h, then min row, then min column
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int oh = __shfl_xor(besth, d, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)besti, d, 64), oj = (uint32_t)__shfl_xor((int)bestj, d, 64);
        if (oh > besth || (oh == besth && (oi < besti || (oi == besti && oj < bestj)))) {
            besth = oh;
            besti = oi;
            bestj = oj;
        }
    }
    if (lane == 0) {
        const bool hit = e == 0u && besth > 0;
        score[pair] = hit ? (int64_t)besth : 0;
        endA[pair] = hit ? besti + 1u : 0u;
        endB[pair] = hit ? bestj + 1u : 0u;
        err[pair] = e;
    }
    } // work items
}

// ---- locate mode on a BYTE PROFILE of the pair (round 5; reads of 257..1024 rows against one reference) ----------
// The packed pass knows the maximum M of a pair; what is left is the first cell, in row-major order, that holds it.  So
// the sweep needs no argmax -- it needs the recurrence and one question per step, "did a cell of mine reach M?":
//   * the wave writes, per B code b, the R bytes score(row, b) - gap of every lane's rows side by side into LDS (plane b:
//     64 lanes x R bytes; rows behind the read's end: -128, they fade out and can never hold M), so a step costs a lane
//     ONE ds_read (b64 / b128) instead of a table lookup per cell, and the byte goes into the add by operand selection;
//   * H + gap is kept instead of H (left and up both arrive with the gap added, the profile carries - gap):
//         x = diag' + byte;  t = max(up', left');  h' = max3(x, t, 0) + gap          -- four instructions per cell
//   * the step's largest h' (one v_max3 per two cells) is compared with M + gap; only a wave in which some lane says
//     yes looks at its rows (smaller row wins, then the earlier column: columns come in order).
// 4.5 instructions per cell against ten (table lookup, validity select, running best per row).  A lane that has not
// started sees pad codes over zeros and stays at zero; what a lane computes past its last column is masked out of the
// question and read by nobody.  Pairs without a maximum from the packed pass (M = 0) are left to sw_wave_kernel
// (m0_only).  Condition (host): gap <= -1, smax - gap <= 127, smin - gap >= -128, the planes of four pairs fit LDS.
template <int R>
__global__ __launch_bounds__(THREADS) void sw_wave8_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    uint32_t lenB, const uint8_t *__restrict__ codeA, const uint8_t *__restrict__ codeB,
    const int32_t *__restrict__ lutcc, int na, int nb, int gap, const uint32_t *__restrict__ infoM,
    const uint32_t *__restrict__ infoQ, int smax, int64_t *__restrict__ score, uint32_t *__restrict__ endA,
    uint32_t *__restrict__ endB, uint32_t *__restrict__ err, int defer)
{
    static_assert(R == 8 || R == 16, "byte-profile sweep: 8 or 16 rows per lane");
    constexpr int NQ = R / 4;
    extern __shared__ __attribute__((aligned(16))) int32_t T[]; // [na][nb], codeA, codeB, then the waves' planes
    uint8_t *cA = reinterpret_cast<uint8_t *>(T + (size_t)na * nb);
    uint8_t *cB = cA + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < na * nb; t += THREADS)
        T[t] = lutcc[t];
    cA[tid] = codeA[tid];
    cB[tid] = codeB[tid];
    __syncthreads();
    uint8_t *prof8 = reinterpret_cast<uint8_t *>(T) + ((((size_t)na * nb * 4 + 512) + 15) & ~(size_t)15) + (size_t)(tid >> 6) * ((size_t)nb * R * 64);

    for (uint64_t pair = (uint64_t)blockIdx.x * (THREADS / 64) + (tid >> 6); pair < npairs; pair += (uint64_t)gridDim.x * (THREADS / 64)) {
        const uint32_t M = infoM[pair], iq = infoQ[pair];
        if (M == 0u)
            continue; // nothing positive, or a read the packed pass did not take: sw_wave_kernel's
        const uint64_t o0 = offA[pair];
        const uint64_t l64 = offA[pair + 1] - o0;
        const bool too_long = l64 > (uint64_t)(64 * R);
        const uint32_t lenA = too_long ? 0u : (uint32_t)l64;
        const uint8_t *ap = A + o0;
        uint32_t ro[R]; // code * nb; 0xFFFFFFFF behind the read's end
        uint32_t mybad = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t r = (uint32_t)lane * R + k;
            uint32_t v = 0xFFFFFFFFu;
            if (r < lenA) {
                const uint32_t c = cA[ap[r]];
                if (c == 0xFFu)
                    mybad = min(mybad, r);
                v = (c == 0xFFu ? (uint32_t)(na - 1) : c) * (uint32_t)nb;
            }
            ro[k] = v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            mybad = min(mybad, (uint32_t)__shfl_xor((int)mybad, d, 64));
        uint32_t e = 0;
        if (too_long) {
            e = 0xFFFFFFFFu;
        } else if (lenA > 0 && lenB > 0) { // align.go:189-191 + matrix.go:29-36: row-major first failing cell
            uint32_t bbad = 0xFFFFFFFFu;   // first byte of B outside SecondAlphabet
            for (uint32_t j0 = 0; j0 < lenB && bbad == 0xFFFFFFFFu; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane;
                const uint64_t bad = __ballot(j < lenB && cB[B[j]] == 0xFFu);
                if (bad)
                    bbad = j0 + (uint32_t)__builtin_ctzll(bad);
            }
            if (mybad == 0u)
                e = (1u << 8) | ap[0];
            else if (bbad != 0xFFFFFFFFu)
                e = (2u << 8) | B[bbad];
            else if (mybad != 0xFFFFFFFFu)
                e = (1u << 8) | ap[mybad];
        }
        // the window (as sw_wave_kernel's locate mode): only the columns that can feed a cell worth M in the block that holds
        // it; a maximum seen in several blocks (tie bit): every column
        uint32_t j0 = 0, ncols = lenB;
        const uint32_t span = (iq >> 31) ? (iq >> 16) & 0xFFu : 0u; // (a near tie: the window runs over first..last block, as above)
        if ((iq >> 31) == 0u || (span < 255u && lenB <= 131072u)) {
            const uint32_t g = (uint32_t)(-gap), top = (uint32_t)smax * lenA;
            const uint32_t need = lenA + (top > M ? (top - M) / g : 0u) + 4u + 4u * span;
            const uint32_t jend = min(4u * ((iq & 0xFFFFu) + span) + 4u, lenB);
            j0 = jend > need ? jend - need : 0u;
            ncols = jend - j0;
        }
        // defer != 0 (polyhip_sw_align_batch_dev: the strings are wanted too): a maximum that sits in ONE block is left to
        // the traceback kernel, which sweeps these columns anyway -- score = M, endA = SW_END_DEFERRED, endB = the 1-based
        // last column of that block (tb_wave_kernel<R, true> finds the first cell worth M during its sweep)
        if (defer && e == 0u && lenA > 0 && ncols > 0 && (iq >> 31) == 0u) {
            if (lane == 0) {
                score[pair] = (int64_t)M;
                endA[pair] = k3p::SW_END_DEFERRED;
                endB[pair] = j0 + ncols;
                err[pair] = 0u;
            }
            continue;
        }
        const uint32_t steps = (e == 0u && lenA > 0 && ncols > 0) ? ncols + 63u : 0u;
        uint32_t besti = 0xFFFFFFFFu, bestj = 0u;
        if (steps) {
            for (int b = 0; b < nb; ++b) {
                uint32_t w[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    w[q] = 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t rk = ro[q * 4 + k];
                        const int v = rk == 0xFFFFFFFFu ? -128 : T[rk + b] - gap;
                        w[q] |= ((uint32_t)v & 0xFFu) << (8 * k);
                    }
                }
                uint32_t *dst = reinterpret_cast<uint32_t *>(prof8 + ((size_t)b * 64 + lane) * R);
                if constexpr (R == 16)
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                else
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(w[0], w[1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            auto load_chunk = [&](uint32_t s0) -> uint32_t { // B codes of window columns s0 + lane
                uint32_t c = (uint32_t)(nb - 1);
                if (s0 + (uint32_t)lane < ncols) {
                    const uint32_t cc = cB[B[j0 + s0 + (uint32_t)lane]];
                    c = cc == 0xFFu ? (uint32_t)(nb - 1) : cc;
                }
                return c;
            };
            uint32_t chunk = load_chunk(0), next_chunk = 0u;
            int lg[R]; // H + gap of the lane's rows, previous column
#pragma unroll
            for (int k = 0; k < R; ++k)
                lg[k] = gap;
            int tprev = gap, lastg = gap;
            uint32_t last_b = (uint32_t)(nb - 1);
            const int Mg = (int)M + gap;
            // (two steps per trip, as tb_wave_kernel's byte-profile sweep: no register copies at the loop's end)
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
                const bool valid = jr < ncols;
                int diag = tprev, up = top_in;
                int m = gap;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int left = lg[k];
                    const int x = diag + (int)(int8_t)(pw[k >> 2] >> (8 * (k & 3)));
                    const int t = max(up, left);
                    const int hg = max(max(x, t), 0) + gap;
                    m = max(m, hg);
                    diag = left;
                    up = hg;
                    lg[k] = hg;
                }
                tprev = top_in;
                lastg = lg[R - 1];
                last_b = b_in;
                if (__any(valid && m == Mg)) { // rare: a cell of this column, in some lane, holds M
                    if (valid) {
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const uint32_t r = (uint32_t)lane * R + k;
                            if (lg[k] == Mg && r < lenA && r < besti) {
                                besti = r;
                                bestj = jr;
                            }
                        }
                    }
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
        }
        bestj += j0; // window-relative so far
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { // the smallest row (rows are distinct across lanes)
            const uint32_t oi = (uint32_t)__shfl_xor((int)besti, d, 64), oj = (uint32_t)__shfl_xor((int)bestj, d, 64);
            if (oi < besti) {
                besti = oi;
                bestj = oj;
            }
        }
        if (lane == 0) {
            const bool hit = e == 0u && besti != 0xFFFFFFFFu;
            score[pair] = hit ? (int64_t)M : 0;
            endA[pair] = hit ? besti + 1u : 0u;
            endB[pair] = hit ? bestj + 1u : 0u;
            // (a maximum the packed pass saw and this sweep does not find cannot happen; if it ever does, say so)
            err[pair] = (e == 0u && !hit && lenA > 0 && lenB > 0) ? 0xFFFFFFFEu : e;
        }
    }
}

bool wave8_ok(const polyhip_scoring *sc, uint32_t max_lenA)
{
    const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
    const size_t smem = (size_t)na * nb * 4 + 512;
    const int r8 = max_lenA <= 512 ? 8 : 16;
    const size_t smem8 = ((smem + 15) & ~(size_t)15) + (size_t)(THREADS / 64) * nb * r8 * 64;
    return max_lenA > 256 && max_lenA <= 1024 && sc->gap <= -1 && -sc->gap <= 127 && (int64_t)sc->smax - sc->gap <= 127 &&
           (int64_t)sc->smin - sc->gap >= -128 && smem8 <= 64 * 1024 && !env_is("POLYHIP_SW_WAVE8", '0');
}

int wave_run(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs, uint32_t max_lenA,
             const uint8_t *d_B, const uint64_t *d_offB, uint32_t lenB, const uint32_t *binfo, const uint32_t *list,
             const uint32_t *count,
             uint64_t max_items, int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, hipStream_t st,
             const uint32_t *infoM, const uint32_t *infoQ, int defer)
{
    const int na = sc->ncodes + 1, nb = sc->ncodesB + 1;
    const size_t smem = (size_t)na * nb * 4 + 512;
    const uint64_t blocks = std::min<uint64_t>((max_items + THREADS / 64 - 1) / (THREADS / 64), 4096);
    if (blocks == 0)
        return POLYHIP_OK;
    // locate mode, 257..1024 rows, one reference: the byte-profile kernel takes every pair whose maximum the packed pass
    // knows; the general kernel below then only the others (POLYHIP_SW_WAVE8=0: the general kernel for all; testing aid)
    int m0_only = 0;
    PH_REQUIRE(!defer || (infoM && infoQ && !list && !d_offB && wave8_ok(sc, max_lenA)), "polyhip_sw_batch: end cells deferred without the byte-profile locate kernel");
    if (infoM && infoQ && !list && !d_offB && wave8_ok(sc, max_lenA)) {
        const int r8 = max_lenA <= 512 ? 8 : 16;
        const size_t smem8 = ((smem + 15) & ~(size_t)15) + (size_t)(THREADS / 64) * nb * r8 * 64;
        {
            if (r8 == 8) {
                auto kern = sw_wave8_kernel<8>;
                PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem8));
                hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), smem8, st, d_A, d_offA, npairs, d_B, lenB, sc->d_codeA,
                                   sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, infoM, infoQ, (int)sc->smax, d_score, d_endA, d_endB, d_err, defer);
            } else {
                auto kern = sw_wave8_kernel<16>;
                PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem8));
                hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), smem8, st, d_A, d_offA, npairs, d_B, lenB, sc->d_codeA,
                                   sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, infoM, infoQ, (int)sc->smax, d_score, d_endA, d_endB, d_err, defer);
            }
            PH_HIP(hipGetLastError());
            m0_only = 1;
        }
    }
#define PH_WAVE_LAUNCH(R_)                                                                                            \
    do {                                                                                                              \
        auto kern = sw_wave_kernel<R_>;                                                                               \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)smem));                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), smem, st, d_A, d_offA, npairs, d_B, d_offB, lenB, \
                           sc->d_codeA, sc->d_codeB, sc->d_lutcc, na, nb, (int)sc->gap, binfo, list, count, infoM,   \
                           infoQ, (int)sc->smax, d_score, d_endA, d_endB, d_err, m0_only);                            \
    } while (0)
    if (max_lenA <= 64)
        PH_WAVE_LAUNCH(1);
    else if (max_lenA <= 128)
        PH_WAVE_LAUNCH(2);
    else if (max_lenA <= 192)
        PH_WAVE_LAUNCH(3);
    else if (max_lenA <= 256)
        PH_WAVE_LAUNCH(4);
    else if (max_lenA <= 512)
        PH_WAVE_LAUNCH(8);
    else if (max_lenA <= 1024)
        PH_WAVE_LAUNCH(16);
    else if (max_lenA <= 2048)
        PH_WAVE_LAUNCH(32);
    else
        PH_WAVE_LAUNCH(64); // up to WAVE_MAX_LENA rows
#undef PH_WAVE_LAUNCH
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

} // namespace k3w
} // namespace polyhip
