// sw_packed.hip -- K3, packed variant of the Smith-Waterman score pass for gfx950.
//
// Same job as sw_shared_kernel (sw_batch.hip): fill + argmax of align.SmithWaterman
// (search/align/align.go:171-203) for many reads against ONE shared reference -- but every lane carries
// TWO pairs, one in each 16-bit half of its registers, and a cell of both costs four packed
// instructions instead of five 32-bit ones:
//     d = v_pk_add_i16(diag, s)            s = (S(a0_i, b_j), S(a1_i, b_j)) straight from LDS
//     m = v_pk_max_i16(up, left)
//     t = v_pk_sub_u16(m, |gap|) clamp     = max(0, m + gap): the recurrence's zero comes for free
//     h = v_pk_max_i16(d, t)
// plus one v_pk_max_i16 for the running maximum of the 4-column block.  The per-cell (row, column)
// key of sw_shared_kernel does not fit 16 bits, so the position is found differently:
//   1. sw_pk_kernel keeps, per pair, the maximum M, the FIRST 4-column block that reaches it and
//      whether a later block reaches it again (a tie);
//   2. sw_locate_kernel re-runs the recurrence (32-bit, one pair per lane, the lane's own columns,
//      byte profile whole in LDS like the traceback) on the few columns that can feed a cell worth M
//      in that block -- lenA + (smax*lenA - M)/|gap| of them -- and takes the first cell equal to M in
//      row-major order (align.go:197-201, strict `>`).  A windowed H never exceeds the true H and equals
//      it wherever the true value is M (all paths of that score fit the window), so the cell is exact;
//   3. pairs with a tie (the row-major-first maximum may sit in a later block with a smaller row)
//      go on a list that sw_shared_kernel, the exact 32-bit kernel, works off afterwards.
//
// The combined profile prof2[j/4][code0 * ncp + code1][j%4] holds both pairs' scores as packed
// int16, 16 bytes per (block, code pair): one ds_read_b128 per row and block, address = block base +
// a 16-bit offset picked from the packed row registers by one SDWA add.  It is streamed through LDS in
// chunks of 64 blocks (36 KB for the 6 codes of {-,A,C,G,T} + pad).
//
// Conditions (plan below): what sw_shared_kernel needs, plus <= 7 symbols in FirstAlphabet, scores
// below 2^15 with headroom, and a reference whose byte profile fits LDS for step 2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "common.h"
#include "sw_scoring.h"

#ifndef PH_SW_NEAR_SPAN
#define PH_SW_NEAR_SPAN 15 // ties whose blocks lie within this many blocks of the first are resolved by sw_locate16_kernel itself
#endif
#ifndef PH_SW_TILE64_DEFAULT
#define PH_SW_TILE64_DEFAULT 1 // 1: reads above 152 rows take 64 rows per lane (four waves per SIMD) unless POLYHIP_SW_TILE64=0
#endif

namespace polyhip {
namespace k3p {

constexpr int THREADS = 256;
constexpr int PADS = -1024; // score of a pad row / pad column: keeps d far below every real value

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// an integer score |v| <= 2048 as the half v * 2^-11 (exact), and back
__device__ __forceinline__ uint32_t half_bits(int v)
{
    const _Float16 h = (_Float16)((float)v * (1.0f / 2048.0f));
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ uint32_t half_score(uint32_t bits)
{
    return (uint32_t)((float)__builtin_bit_cast(_Float16, (unsigned short)bits) * 2048.0f);
}

// prof2 entry (q, cidx) = 4 dwords, one per column of block q: lo half = S(sym(code0), b_j), hi = S(sym(code1), b_j)
// `lead` all-pad blocks in front and behind (nq counts them): the banded kernel's lanes run up to `lead` blocks apart
__global__ __launch_bounds__(256) void profile2_kernel(const uint8_t *__restrict__ B, uint32_t lenB, uint32_t nq,
                                                      uint32_t lead, const int8_t *__restrict__ lutc, int ncodes,
                                                      int ncp, uint32_t *__restrict__ prof2, int f16, int gapabs)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; // one (block, code pair) each
    const uint32_t ncc = (uint32_t)(ncp * ncp);
    if (e >= nq * ncc)
        return;
    const uint32_t q = e / ncc, cidx = e % ncc;
    const int c0 = (int)(cidx / (uint32_t)ncp), c1 = (int)(cidx % (uint32_t)ncp);
    uint32_t w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t j = 4 * (q - lead) + c; // wraps for the leading pad blocks: not < lenB
        int s0 = PADS, s1 = PADS;
        if (q >= lead && j < lenB) {
            const uint8_t b = B[j];
            if (c0 < ncodes)
                s0 = lutc[c0 * 256 + b];
            if (c1 < ncodes)
                s1 = lutc[c1 * 256 + b];
        }
        if (f16) { // F16 cell: scores as halves, scaled by 2^-11 (exact: |s| <= 2048); column 0 of a block carries + |gap|:
            // its diagonal value arrives with the gap already taken (PH_PKF_ROW)
            const int bias = c == 0 ? gapabs : 0;
            w[c] = half_bits(s0 + bias) | (half_bits(s1 + bias) << 16);
        }
        else
            w[c] = ((uint32_t)s0 & 0xFFFFu) | ((uint32_t)s1 << 16);
    }
    reinterpret_cast<uint4 *>(prof2)[e] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// F16 flavour (scores < 2048): the same integers held as halves scaled by 2^-11, every sum exact.  gfx950's
// three-operand packed maximum and the [0, 1] clamp of a packed float add make a cell three instructions:
//     d = v_pk_add_f16(diag, s)
//     h = v_pk_maximum3_f16(d, up - |gap|, left - |gap|)
//     g = v_pk_add_f16(h, -|gap|) clamp          = max(0, h + gap): what the cell below and the cell right take
// and the block maximum two v_pk_maximum3_f16 per row.  Non-negative halves order like their bit patterns, so the
// block / tie bookkeeping is the integer one.
__device__ __forceinline__ uint32_t pkf_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pkf_addc(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pkf_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The value left of column 0 (H[i], from the block before) serves twice: as "left - |gap|" of this row and as the
// diagonal of the row below.  Both take it with the gap ALREADY subtracted (one unclamped add into a fresh register;
// column 0 of the profile is biased by + |gap| to make up for it on the diagonal, and the maximum's other operand
// pg0 >= 0 makes the missing clamp invisible), so H[i]'s register is free for this row's h3 and no copy is needed.
#define PH_PKF_ROW(I, W)                                   \
    do {                                                   \
        const int i_ = (I);                                \
        const uint32_t gu = pkf_add(H[i_], gap2);          \
        const uint32_t h0 = pkf_max3(pkf_add(pdiag, (W).x), pg0, gu); \
        const uint32_t g0 = pkf_addc(h0, gap2);            \
        const uint32_t h1 = pkf_max3(pkf_add(pr0, (W).y), pg1, g0);   \
        const uint32_t g1 = pkf_addc(h1, gap2);            \
        const uint32_t h2 = pkf_max3(pkf_add(pr1, (W).z), pg2, g1);   \
        const uint32_t g2 = pkf_addc(h2, gap2);            \
        const uint32_t h3 = pkf_max3(pkf_add(pr2, (W).w), pg3, g2);   \
        bm = pkf_max3(pkf_max3(bm, h0, h1), h2, h3);       \
        pdiag = gu;                                        \
        pr0 = h0;                                          \
        pr1 = h1;                                          \
        pr2 = h2;                                          \
        pr3 = h3;                                          \
        pg0 = g0;                                          \
        pg1 = g1;                                          \
        pg2 = g2;                                          \
        pg3 = pkf_addc(h3, gap2);                          \
        H[i_] = h3;                                        \
    } while (0)

// address of a row's table entry = (block base / 16 + code-pair index) * 16: the index is a byte of the
// packed row registers (SDWA add), the shift restores bytes
#define PH_PK_ISSUE(dst, rp, SEL)                                                                                  \
    do {                                                                                                           \
        uint32_t ad_;                                                                                              \
        asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL  \
                     "\n\tv_lshlrev_b32_e32 %0, 4, %0"                                                             \
                     : "=&v"(ad_)                                                                                  \
                     : "v"(blk16), "v"(rp));                                                                       \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad_));                                                \
    } while (0)
#define PH_PK_ROW(I, W)                                    \
    do {                                                   \
        const int i_ = (I);                                \
        const uint32_t left = H[i_];                       \
        const uint32_t h0 = pk_max(pk_add(pdiag, (W).x), pk_subsat(pk_max(pr0, left), gap2)); \
        const uint32_t h1 = pk_max(pk_add(pr0, (W).y), pk_subsat(pk_max(pr1, h0), gap2));     \
        const uint32_t h2 = pk_max(pk_add(pr1, (W).z), pk_subsat(pk_max(pr2, h1), gap2));     \
        const uint32_t h3 = pk_max(pk_add(pr2, (W).w), pk_subsat(pk_max(pr3, h2), gap2));     \
        /* (round 5) the block maximum through gfx950's three-operand half-float maximum: every H here is an integer in  \
           [0, 30000) (packed_plan), and non-negative halves -- denormals included, the kernels run with them preserved -- \
           order like their bit patterns; 0x7C00 (infinity, NaN) is never reached.  Two instructions instead of five.    \
           (Not for the cell itself: diag + score can be a small negative integer, which is a NaN pattern, and this       \
           maximum propagates NaNs.) */                                                      \
        bm = pkf_max3(pkf_max3(bm, h0, h1), h2, h3);                                          \
        pdiag = left;                                      \
        pr0 = h0;                                          \
        pr1 = h1;                                          \
        pr2 = h2;                                          \
        pr3 = h3;                                          \
        H[i_] = h3;                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// byte code of A[i] of one pair, pad code where the pair has no such row or the byte is not in FirstAlphabet
// ---- what a packed kernel keeps per pair (one per 16-bit half of its registers) -----------------------------------------------
// the running maximum `best`, the FIRST 4-column block that reached it `bestq`, and in `ties` bit 0 = "a later block reached it
// again" with, in bits 1..15, the low 15 bits of the LAST such block (round 6: nearly every tie of configs[3] is two or three
// neighbouring blocks -- the same alignment path coming back to its maximum --, and the locate step then only has to look at
// those blocks instead of handing the pair to the full sweep).  No extra register, and the tie branch stays ONE instruction:
// the block number is wave-uniform, so what goes into the field is built on the scalar unit.
// infoQ word of a pair: bits 0..15 first block, bits 16..23 span = last - first (saturating at 255), bit 31 tie.
__device__ __forceinline__ void track_block_max(uint32_t bm, uint32_t q, uint32_t &best, uint32_t &bestq, uint32_t &ties)
{
    const uint32_t blo = bm & 0xFFFFu, bhi = bm >> 16, mlo = best & 0xFFFFu, mhi = best >> 16;
    const uint32_t tv = ((q & 0x7FFFu) << 1) | 1u; // (scalar)
    if (blo > mlo) {
        best = (best & 0xFFFF0000u) | blo;
        bestq = (bestq & 0xFFFF0000u) | (q & 0xFFFFu);
        ties &= 0xFFFF0000u;
    } else if (blo == mlo && blo != 0u) {
        ties = (ties & 0xFFFF0000u) | tv;
    }
    if (bhi > mhi) {
        best = (best & 0xFFFFu) | (bhi << 16);
        bestq = (bestq & 0xFFFFu) | (q << 16);
        ties &= 0xFFFFu;
    } else if (bhi == mhi && bhi != 0u) {
        ties = (ties & 0xFFFFu) | (tv << 16);
    }
}
// (the span is exact while the reference has at most 32,768 blocks; sw_locate16_kernel only trusts it there)
__device__ __forceinline__ uint32_t info_q_word(uint32_t firstq, uint32_t tfield)
{
    const uint32_t span = min((((tfield >> 1) & 0x7FFFu) - firstq) & 0x7FFFu, 255u);
    return firstq | ((tfield & 1u) ? span << 16 : 0u) | ((tfield & 1u) << 31);
}
__device__ __forceinline__ uint32_t info_q_lo(uint32_t bestq, uint32_t ties) { return info_q_word(bestq & 0xFFFFu, ties & 0xFFFFu); }
__device__ __forceinline__ uint32_t info_q_hi(uint32_t bestq, uint32_t ties) { return info_q_word(bestq >> 16, ties >> 16); }

__device__ __forceinline__ uint32_t row_code(const uint8_t *__restrict__ ap, uint32_t lenA, int i,
                                             const uint8_t *__restrict__ codeL, uint32_t pad)
{
    if ((uint32_t)i >= lenA)
        return pad;
    const uint32_t c = codeL[ap[i]];
    return c == 0xFFu ? pad : c;
}

// SKIP: the wave's longest read decides how many row groups of the unrolled sweep run (a uniform branch per group);
// batches whose longest read nearly fills RA take the plain instantiation (the branches cost 3 % there)
template <int RA, bool SKIP, bool F16>
__global__ __launch_bounds__(THREADS, 2) void sw_pk_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                       uint64_t npairs, const uint32_t *__restrict__ prof2,
                                                       uint32_t nq, uint32_t jcb, uint32_t tab_bytes, int ncp,
                                                       const uint8_t *__restrict__ codeA, int ncodes, int gapabs,
                                                       uint32_t *__restrict__ infoM, uint32_t *__restrict__ infoQ)
{
    static_assert(RA % 4 == 0 && RA <= 152, "RA"); // two workgroups per CU: 256 VGPRs hold H[RA] + RA/4 code registers
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_pk[];
    uint8_t *codeL = lds_pk + (size_t)jcb * tab_bytes;
    const int tid = threadIdx.x;
    codeL[tid] = codeA[tid];
    __syncthreads();

    // my two pairs
    const uint64_t base = (uint64_t)blockIdx.x * (2 * THREADS);
    const uint64_t p0 = base + tid, p1 = base + THREADS + tid;
    const uint8_t *ap0 = A, *ap1 = A;
    uint32_t len0 = 0, len1 = 0;
    if (p0 < npairs) {
        const uint64_t o = offA[p0], l = offA[p0 + 1] - o;
        ap0 = A + o;
        len0 = l > (uint64_t)RA ? 0u : (uint32_t)l; // too long: no score here, the locate kernel reports it
    }
    if (p1 < npairs) {
        const uint64_t o = offA[p1], l = offA[p1 + 1] - o;
        ap1 = A + o;
        len1 = l > (uint64_t)RA ? 0u : (uint32_t)l;
    }
    // index of row i's code pair inside a block's table, four rows per register
    uint32_t rpk[RA / 4];
#pragma unroll
    for (int w = 0; w < RA / 4; ++w) {
        uint32_t pk = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int i = 4 * w + h;
            const uint32_t c0 = row_code(ap0, len0, i, codeL, (uint32_t)ncodes);
            const uint32_t c1 = row_code(ap1, len1, i, codeL, (uint32_t)ncodes);
            pk |= (c0 * (uint32_t)ncp + c1) << (8 * h);
        }
        rpk[w] = pk;
    }

    uint32_t H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;
    // row groups this wave needs (its longest read): the rest of the unrolled sweep is skipped, so 100-bp reads
    // cost their own rows, not RA
    int ng = RA / 4;
    if (SKIP) {
        uint32_t wl = max(len0, len1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
        ng = __builtin_amdgcn_readfirstlane((int)((wl + 3u) >> 2));
    }
    uint32_t gap2 = (uint32_t)gapabs | ((uint32_t)gapabs << 16);
    if (F16) { // -|gap| * 2^-11 in both halves
        const uint32_t g = half_bits(-gapabs);
        gap2 = g | (g << 16);
    }
    uint32_t best = 0, bestq = 0, ties = 0; // packed halves: maximum, its first block, bit 0 / bit 16 = tie
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_pk));

    for (uint32_t q0 = 0; q0 < nq; q0 += jcb) {
        const uint32_t nb = min(jcb, nq - q0);
        __syncthreads(); // previous chunk fully consumed
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(prof2) + (size_t)q0 * tab_bytes);
            uint4 *dst = reinterpret_cast<uint4 *>(lds_pk);
            const uint32_t nvec = nb * tab_bytes / 16;
            for (uint32_t v = tid; v < nvec; v += THREADS)
                dst[v] = src[v];
        }
        __syncthreads();
        for (uint32_t t = 0; t < nb; ++t) {
            const uint32_t blk16 = (lds_base + t * tab_bytes) >> 4; // lds_pk and tab_bytes are multiples of 16
            uint32_t pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pdiag = F16 ? gap2 : 0u, bm = 0; // F16: 0 - |gap| (PH_PKF_ROW)
            uint32_t pg0 = 0, pg1 = 0, pg2 = 0, pg3 = 0; // F16: the row above, gap already taken
            (void)pg0, (void)pg1, (void)pg2, (void)pg3, (void)pr3;
            u32x4 wa, wb;
#define PH_PK_STEP(I, W)       \
    do {                       \
        if constexpr (F16)     \
            PH_PKF_ROW(I, W);  \
        else                   \
            PH_PK_ROW(I, W);   \
    } while (0)
            PH_PK_ISSUE(wa, rpk[0], "BYTE_0");
#pragma unroll
            for (int g = 0; g < RA / 4; ++g) {
                if (!SKIP || g < ng) { // wave-uniform
                    PH_PK_ISSUE(wb, rpk[g], "BYTE_1");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                    PH_PK_STEP(4 * g, wa);
                    PH_PK_ISSUE(wa, rpk[g], "BYTE_2");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                    PH_PK_STEP(4 * g + 1, wb);
                    PH_PK_ISSUE(wb, rpk[g], "BYTE_3");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                    PH_PK_STEP(4 * g + 2, wa);
                    if (g + 1 < RA / 4) {
                        PH_PK_ISSUE(wa, rpk[g + 1], "BYTE_0");
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wb));
                    }
                    PH_PK_STEP(4 * g + 3, wb);
                }
            }
            if (SKIP)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa)); // the read issued ahead of a skipped group
            // block maximum against the running one, per half; a block reaching the maximum AGAIN is a tie
            const uint32_t q = q0 + t;
            track_block_max(bm, q, best, bestq, ties);
        }
    }
#undef PH_PK_STEP
    uint32_t m0 = best & 0xFFFFu, m1 = best >> 16;
    if (F16) {
        m0 = half_score(m0);
        m1 = half_score(m1);
    }
    if (p0 < npairs) {
        infoM[p0] = m0;
        infoQ[p0] = info_q_lo(bestq, ties);
    }
    if (p1 < npairs) {
        infoM[p1] = m1;
        infoQ[p1] = info_q_hi(bestq, ties);
    }
}

// ---- the same sweep with ONE wave per workgroup and the current block's table at a FIXED LDS address --------------
// A row's table entry then sits at (code-pair index << 4) + an immediate: ONE vector instruction (an SDWA shift of the
// packed index byte) instead of two (SDWA add of the block base, shift), 16 instead of 17 per row and 4-column block.
// A workgroup of one wave owns its LDS allocation, so the slot is LDS address 0 for every wave and no barrier guards
// it: when a block starts the wave sends the next block's table (at most 64 entries of 16 bytes, one per lane,
// L2-resident) from global memory straight into a staging area of LDS (global_load_lds: no register is held under the
// rows, where the allocator is at its limit), and copies it over the slot when the block's last row has read its entry --
// LDS runs a wave's instructions in order, so the next block's first read sees the new table.  No chunk staging, no
// __syncthreads in the sweep.  Half-float cells only (F16), tables of up to 64 code pairs (ncp <= 8).
#define PH_PK1_ISSUE(dst, rp, SEL)                                                                                   \
    do {                                                                                                           \
        uint32_t ad_;                                                                                              \
        asm volatile("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL \
                     : "=v"(ad_)                                                                                   \
                     : "v"(rp));                                                                                   \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad_));                                                \
    } while (0)

template <int RA, bool SKIP>
__global__ __launch_bounds__(64, 2) void sw_pk1_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                      uint64_t npairs, const uint32_t *__restrict__ prof2, uint32_t nq,
                                                      uint32_t tab_bytes, int ncp, const uint8_t *__restrict__ codeA,
                                                      int ncodes, int gapabs, uint32_t *__restrict__ infoM,
                                                      uint32_t *__restrict__ infoQ)
{
    static_assert(RA % 4 == 0 && RA <= 152, "RA");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_pk[]; // [0, 1024) the table slot, [1024, 1280) the code bytes,
    uint8_t *codeL = lds_pk + 1024;                                  // [1280, 2304) the next block's table arriving
    const uint32_t lane = threadIdx.x;
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)lds_pk)) != 0u)
        __builtin_trap(); // the sweep addresses the slot by immediate
#pragma unroll
    for (int u = 0; u < 4; ++u)
        codeL[lane + 64 * u] = codeA[lane + 64 * u];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // my two pairs
    const uint64_t base = (uint64_t)blockIdx.x * 128;
    const uint64_t p0 = base + lane, p1 = base + 64 + lane;
    uint64_t o0 = 0, o1 = 0;
    uint32_t len0 = 0, len1 = 0;
    if (p0 < npairs) {
        o0 = offA[p0];
        const uint64_t l = offA[p0 + 1] - o0;
        len0 = l > (uint64_t)RA ? 0u : (uint32_t)l; // too long: no score here, the locate kernel reports it
    }
    if (p1 < npairs) {
        o1 = offA[p1];
        const uint64_t l = offA[p1 + 1] - o1;
        len1 = l > (uint64_t)RA ? 0u : (uint32_t)l;
    }
    // index of row i's code pair inside a block's table, four rows per register.  The reads' bytes come as RA / 4 + 1
    // ALIGNED dwords per pair, all in flight at once, funnelled to the read's own alignment -- a byte at a time the prologue
    // was 2 x RA dependent round trips, ~2 % of a wave's life with both waves of a SIMD in it together.  A buffer resource
    // over the whole packed batch (rounded out to whole dwords: an aligned dword never crosses a page) bounds the loads;
    // what lies beyond the batch reads as zero, what lies beyond the read is never looked at.  Batches of 4 GB and more
    // keep the byte loads.
    uint32_t rpk[RA / 4];
    const uint64_t totalA = offA[npairs];
    if (totalA < 0xFFFFFFF0ull) {
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 3u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A) - misA, 0,
                                                                            (int)(((uint32_t)totalA + misA + 3u) & ~3u), 0x00020000);
        uint32_t d0[RA / 4], d1[RA / 4];
        {
            const uint32_t b0 = (uint32_t)o0 + misA, b1 = (uint32_t)o1 + misA;
            uint32_t a0[RA / 4 + 1], a1[RA / 4 + 1];
#pragma unroll
            for (int w = 0; w <= RA / 4; ++w) {
                a0[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b0 & ~3u) + 4u * w), 0, 0);
                a1[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b1 & ~3u) + 4u * w), 0, 0);
            }
#pragma unroll
            for (int w = 0; w < RA / 4; ++w) {
                d0[w] = __builtin_amdgcn_alignbyte(a0[w + 1], a0[w], b0 & 3u);
                d1[w] = __builtin_amdgcn_alignbyte(a1[w + 1], a1[w], b1 & 3u);
            }
        }
#pragma unroll
        for (int w = 0; w < RA / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t i = 4u * w + h;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes; // pad: no such row, or a byte outside FirstAlphabet
                if (i < len0) {
                    const uint32_t c = codeL[(d0[w] >> (8 * h)) & 0xFFu];
                    c0 = c == 0xFFu ? c0 : c;
                }
                if (i < len1) {
                    const uint32_t c = codeL[(d1[w] >> (8 * h)) & 0xFFu];
                    c1 = c == 0xFFu ? c1 : c;
                }
                pk |= (c0 * (uint32_t)ncp + c1) << (8 * h);
            }
            rpk[w] = pk;
        }
    } else {
        const uint8_t *ap0 = A + o0, *ap1 = A + o1;
#pragma unroll
        for (int w = 0; w < RA / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int i = 4 * w + h;
                const uint32_t c0 = row_code(ap0, len0, i, codeL, (uint32_t)ncodes);
                const uint32_t c1 = row_code(ap1, len1, i, codeL, (uint32_t)ncodes);
                pk |= (c0 * (uint32_t)ncp + c1) << (8 * h);
            }
            rpk[w] = pk;
        }
    }
    uint32_t H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;
    int ng = RA / 4;
    if (SKIP) {
        uint32_t wl = max(len0, len1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
        ng = __builtin_amdgcn_readfirstlane((int)((wl + 3u) >> 2));
    }
    const uint32_t gh = half_bits(-gapabs);
    const uint32_t gap2 = gh | (gh << 16); // -|gap| * 2^-11 in both halves
    uint32_t best = 0, bestq = 0, ties = 0;

    // table of block 0 into the slot; lane l owns entry l
    const uint32_t nent = tab_bytes >> 4;
    const bool mine = lane < nent;
    const uint32_t slot = lane << 4;
    const uint8_t *tabp = reinterpret_cast<const uint8_t *>(prof2) + slot; // my entry of block t
    if (mine)
        reinterpret_cast<uint4 *>(lds_pk)[lane] = *reinterpret_cast<const uint4 *>(tabp);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = 0; t < nq; ++t) {
        tabp += tab_bytes;
        // the next block's table: global memory -> the staging area behind the codes, no register held under the rows
        if (mine && t + 1 < nq)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)tabp,
                                             (__attribute__((address_space(3))) void *)((__attribute__((address_space(3))) uint8_t *)lds_pk + 1280), 16, 0, 0);
        // (the LDS pointer stays in its own address space: through a generic pointer hipcc 7.2 once stopped here with "Illegal
        // instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" -- the null check of the cast back)
        uint32_t pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pdiag = gap2, bm = 0; // 0 - |gap| (PH_PKF_ROW)
        uint32_t pg0 = 0, pg1 = 0, pg2 = 0, pg3 = 0;
        (void)pr3;
        u32x4 wa, wb;
        PH_PK1_ISSUE(wa, rpk[0], "BYTE_0");
#pragma unroll
        for (int g = 0; g < RA / 4; ++g) {
            if (!SKIP || g < ng) { // wave-uniform
                PH_PK1_ISSUE(wb, rpk[g], "BYTE_1");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PKF_ROW(4 * g, wa);
                PH_PK1_ISSUE(wa, rpk[g], "BYTE_2");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                PH_PKF_ROW(4 * g + 1, wb);
                PH_PK1_ISSUE(wb, rpk[g], "BYTE_3");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PKF_ROW(4 * g + 2, wa);
                if (g + 1 < RA / 4) {
                    PH_PK1_ISSUE(wa, rpk[g + 1], "BYTE_0");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wb));
                }
                PH_PKF_ROW(4 * g + 3, wb);
            }
        }
        if (SKIP)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa)); // the read issued ahead of a skipped group
        if (mine && t + 1 < nq) { // every row of this block has its entry: the slot takes the next table
            u32x4 tmp;
            asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b128 %0, %1 offset:1280\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %1, %0"
                         : "=&v"(tmp)
                         : "v"(slot)
                         : "memory");
        }
        // block maximum against the running one, per half; a block reaching the maximum AGAIN is a tie
        track_block_max(bm, t, best, bestq, ties);
    }
    const uint32_t m0 = half_score(best & 0xFFFFu), m1 = half_score(best >> 16);
    if (p0 < npairs) {
        infoM[p0] = m0;
        infoQ[p0] = info_q_lo(bestq, ties);
    }
    if (p1 < npairs) {
        infoM[p1] = m1;
        infoQ[p1] = info_q_hi(bestq, ties);
    }
}

__device__ __forceinline__ uint32_t from_lane_above(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
}

// ---- sw_pk1_kernel's sweep with every lane's rows split over TWO lanes (round 6) ------------------------------------
// 152 packed H rows + the row codes pin sw_pk1_kernel at 256 registers, two waves per SIMD, and two waves cannot hide the
// packed chain's own dependency (hipcc pads five s_nop per row; VALU issue 0.82 of the busy cycles).  Here lanes 2d and
// 2d + 1 share the two read pairs of duo d: the even lane holds rows [0, RB), the odd lane rows [RB, 2 RB) and works ONE
// 4-column block behind, so what it needs from above -- the even lane's last row of the block before (four H, the
// pre-subtracted diagonal, the block maximum) -- is in the even lane's registers when the step starts: six DPP row_shr:1
// moves per step, each of which also resets the even lane to the sweep's top-of-column constants (the select takes the
// constant on even lanes).  76 H rows + 19 code registers: 128 registers, four waves per SIMD.
// Two tables live in LDS at fixed addresses: block t's at 0 for the even lanes, block t-1's at 1024 for the odd lanes,
// whose code bytes carry + 64 -- so the address stays ONE SDWA shift.  When a step ends, slot 0 moves to slot 1 and the
// staged next table to slot 0 (LDS runs a wave's instructions in order; one wave per workgroup, no barrier).  Step 0's
// odd lanes see an all-pad table (H stays 0), the last step's even lanes sweep a stale table and nothing reads them.
// SKIP (batches whose longest read leaves four or more row groups of the 152-row tile unused): the split between the two lanes
// follows the WAVE's longest read -- each lane takes ng = ceil(longest / 8) groups of four rows, the odd lane's rows start at
// 4 ng -- so reads of 65..136 bp run as tight as a tile of their own would (a uniform branch per group, as sw_pk1_kernel<.., true>).
template <int RB, bool SKIP>
__global__ __launch_bounds__(64, 4) void sw_pk1x2_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                        uint64_t npairs, const uint32_t *__restrict__ prof2, uint32_t nq,
                                                        uint32_t tab_bytes, int ncp, const uint8_t *__restrict__ codeA,
                                                        int ncodes, int gapabs, uint32_t *__restrict__ infoM,
                                                        uint32_t *__restrict__ infoQ)
{
    static_assert(RB % 4 == 0 && RB <= 76, "RB");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_pk[]; // [0, 1024) slot 0, [1024, 2048) slot 1,
    uint8_t *codeL = lds_pk + 2048;                                  // [2048, 2304) the code bytes, [2304, 3328) staging
    const uint32_t lane = threadIdx.x;
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)lds_pk)) != 0u)
        __builtin_trap(); // the sweep addresses the slots by immediate
#pragma unroll
    for (int u = 0; u < 4; ++u)
        codeL[lane + 64 * u] = codeA[lane + 64 * u];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const uint32_t band = lane & 1u, duo = lane >> 1;
    const bool odd = band != 0u;
    const uint64_t base = (uint64_t)blockIdx.x * 64;
    const uint64_t p0 = base + duo, p1 = base + 32 + duo;
    uint64_t o0 = 0, o1 = 0;
    uint32_t len0 = 0, len1 = 0;
    if (p0 < npairs) {
        o0 = offA[p0];
        const uint64_t l = offA[p0 + 1] - o0;
        len0 = l > (uint64_t)(2 * RB) ? 0u : (uint32_t)l; // too long: no score here, the locate kernel reports it
    }
    if (p1 < npairs) {
        o1 = offA[p1];
        const uint64_t l = offA[p1 + 1] - o1;
        len1 = l > (uint64_t)(2 * RB) ? 0u : (uint32_t)l;
    }
    int ng = RB / 4;
    if (SKIP) {
        uint32_t wl = max(len0, len1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
        ng = __builtin_amdgcn_readfirstlane((int)((wl + 7u) >> 3)); // <= RB / 4: no read is longer than 2 RB
    }
    // my rows' code pairs, four per register (+ 64 on the odd lanes: slot 1); the bytes as aligned dwords through a
    // bounded buffer resource, as sw_pk1_kernel
    const uint32_t row0 = band * 4u * (uint32_t)ng;
    const uint32_t slot_bias = odd ? 64u : 0u;
    uint32_t rpk[RB / 4];
    const uint64_t totalA = offA[npairs];
    if (totalA < 0xFFFFFFF0ull) {
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 3u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A) - misA, 0,
                                                                            (int)(((uint32_t)totalA + misA + 3u) & ~3u), 0x00020000);
        uint32_t d0[RB / 4], d1[RB / 4];
        {
            const uint32_t b0 = (uint32_t)o0 + misA + row0, b1 = (uint32_t)o1 + misA + row0;
            uint32_t a0[RB / 4 + 1], a1[RB / 4 + 1];
#pragma unroll
            for (int w = 0; w <= RB / 4; ++w) {
                a0[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b0 & ~3u) + 4u * w), 0, 0);
                a1[w] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((b1 & ~3u) + 4u * w), 0, 0);
            }
#pragma unroll
            for (int w = 0; w < RB / 4; ++w) {
                d0[w] = __builtin_amdgcn_alignbyte(a0[w + 1], a0[w], b0 & 3u);
                d1[w] = __builtin_amdgcn_alignbyte(a1[w + 1], a1[w], b1 & 3u);
            }
        }
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t i = row0 + 4u * w + h;
                uint32_t c0 = (uint32_t)ncodes, c1 = (uint32_t)ncodes; // pad: no such row, or a byte outside FirstAlphabet
                if (i < len0) {
                    const uint32_t c = codeL[(d0[w] >> (8 * h)) & 0xFFu];
                    c0 = c == 0xFFu ? c0 : c;
                }
                if (i < len1) {
                    const uint32_t c = codeL[(d1[w] >> (8 * h)) & 0xFFu];
                    c1 = c == 0xFFu ? c1 : c;
                }
                pk |= (c0 * (uint32_t)ncp + c1 + slot_bias) << (8 * h);
            }
            rpk[w] = pk;
        }
    } else {
        const uint8_t *ap0 = A + o0, *ap1 = A + o1;
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int i = (int)row0 + 4 * w + h;
                const uint32_t c0 = row_code(ap0, len0, i, codeL, (uint32_t)ncodes);
                const uint32_t c1 = row_code(ap1, len1, i, codeL, (uint32_t)ncodes);
                pk |= (c0 * (uint32_t)ncp + c1 + slot_bias) << (8 * h);
            }
            rpk[w] = pk;
        }
    }
    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;
    const uint32_t gh = half_bits(-gapabs);
    const uint32_t gap2 = gh | (gh << 16); // -|gap| * 2^-11 in both halves
    uint32_t best = 0, bestq = 0, ties = 0;

    // block 0's table into slot 0, an all-pad table (column 0 with its + |gap| bias, as profile2_kernel) into slot 1
    const uint32_t nent = tab_bytes >> 4;
    const bool mine = lane < nent;
    const uint32_t slot = lane << 4;
    const uint8_t *tabp = reinterpret_cast<const uint8_t *>(prof2) + slot; // my entry of block t
    if (mine)
        reinterpret_cast<uint4 *>(lds_pk)[lane] = *reinterpret_cast<const uint4 *>(tabp);
    {
        const uint32_t pd = half_bits(PADS), pd0 = half_bits(PADS + gapabs);
        reinterpret_cast<uint4 *>(lds_pk + 1024)[lane] = make_uint4(pd0 | (pd0 << 16), pd | (pd << 16), pd | (pd << 16), pd | (pd << 16));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pdiag = gap2, bm = 0;
    for (uint32_t t = 0; t <= nq; ++t) { // even lanes: block t (t < nq); odd lanes: block t - 1
        tabp += tab_bytes;
        if (mine && t + 1 < nq)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)tabp,
                                             (__attribute__((address_space(3))) void *)((__attribute__((address_space(3))) uint8_t *)lds_pk + 2304), 16, 0, 0);
        // the band above hands its last row of block t - 1 down (odd lanes); the even lanes start at the top of the column
        {
            const uint32_t u0 = from_lane_above(pr0), u1 = from_lane_above(pr1), u2 = from_lane_above(pr2),
                           u3 = from_lane_above(pr3), ud = from_lane_above(pdiag), ub = from_lane_above(bm);
            pr0 = odd ? u0 : 0u;
            pr1 = odd ? u1 : 0u;
            pr2 = odd ? u2 : 0u;
            pr3 = odd ? u3 : 0u;
            pdiag = odd ? ud : gap2; // 0 - |gap| (PH_PKF_ROW)
            bm = odd ? ub : 0u;
        }
        uint32_t pg0 = pkf_addc(pr0, gap2), pg1 = pkf_addc(pr1, gap2), pg2 = pkf_addc(pr2, gap2), pg3 = pkf_addc(pr3, gap2);
        u32x4 wa, wb;
        PH_PK1_ISSUE(wa, rpk[0], "BYTE_0");
#pragma unroll
        for (int g = 0; g < RB / 4; ++g) {
            if (!SKIP || g < ng) { // wave-uniform
                PH_PK1_ISSUE(wb, rpk[g], "BYTE_1");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PKF_ROW(4 * g, wa);
                PH_PK1_ISSUE(wa, rpk[g], "BYTE_2");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                PH_PKF_ROW(4 * g + 1, wb);
                PH_PK1_ISSUE(wb, rpk[g], "BYTE_3");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PKF_ROW(4 * g + 2, wa);
                if (g + 1 < RB / 4) {
                    PH_PK1_ISSUE(wa, rpk[g + 1], "BYTE_0");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wb));
                }
                PH_PKF_ROW(4 * g + 3, wb);
            }
        }
        if (SKIP)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa)); // the read issued ahead of a skipped group
        if (mine) { // every row of this step has its entry: slot 1 takes block t's table, slot 0 the staged block t + 1
            u32x4 cur, nxt;
            asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2304\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "ds_write_b128 %2, %0 offset:1024\n\tds_write_b128 %2, %1"
                         : "=&v"(cur), "=&v"(nxt)
                         : "v"(slot)
                         : "memory");
        }
        // the pair's block maximum (all 2 RB rows) is the odd lane's; a block reaching the maximum AGAIN is a tie
        const uint32_t tb = t - 1u;
        track_block_max(bm, tb, best, bestq, ties);
    }
    if (odd) {
        const uint32_t m0 = half_score(best & 0xFFFFu), m1 = half_score(best >> 16);
        if (p0 < npairs) {
            infoM[p0] = m0;
            infoQ[p0] = info_q_lo(bestq, ties);
        }
        if (p1 < npairs) {
            infoM[p1] = m1;
            infoQ[p1] = info_q_hi(bestq, ties);
        }
    }
}

// ---- reads of 153 .. 256 rows: K lanes per pair ------------------------------------------------------
// One lane cannot hold more than 152 packed rows at two workgroups per CU, and at one workgroup per CU the
// dependent packed chain stands exposed (measured: 256 rows in one lane run no faster than the 32-bit kernel).
// So a pair of reads is spread over K neighbouring lanes, RB rows each, as a short systolic array: the lane
// of band b works on 4-column block t - b in step t, and its last row (four values + the diagonal one of the
// block before) reaches the lane of band b + 1 by one DPP row shift per step, together with the running
// block maximum of the bands above.  The last band's lane sees the pair's block maxima and keeps M / first
// block / tie exactly as sw_pk_kernel does.  prof2 carries K - 1 all-pad blocks on either side (lanes ahead
// of / behind the reference see pad columns, in which H only decays), an LDS chunk K - 1 extra blocks.

template <int RB, int K, bool F16>
__global__ __launch_bounds__(THREADS, RB <= 64 ? 4 : 2) void sw_pkb_kernel(const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA,
                                                        uint64_t npairs, const uint32_t *__restrict__ prof2,
                                                        uint32_t nq, uint32_t jcb, uint32_t tab_bytes, int ncp,
                                                        const uint8_t *__restrict__ codeA, int ncodes, int gapabs,
                                                        uint32_t *__restrict__ infoM, uint32_t *__restrict__ infoQ)
{
    static_assert(RB % 4 == 0 && RB <= 152 && K >= 2 && K <= 16 && (K & (K - 1)) == 0, "RB, K"); // a DPP row is 16 lanes
    constexpr int G = THREADS / K; // lane groups per workgroup, two pairs each
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_pk[];
    uint8_t *codeL = lds_pk + (size_t)(jcb + K - 1) * tab_bytes;
    const int tid = threadIdx.x;
    codeL[tid] = codeA[tid];
    __syncthreads();

    const int grp = tid / K, band = tid % K;
    const uint64_t base = (uint64_t)blockIdx.x * (2 * G);
    const uint64_t p0 = base + grp, p1 = base + G + grp;
    const uint8_t *ap0 = A, *ap1 = A;
    uint32_t len0 = 0, len1 = 0;
    if (p0 < npairs) {
        const uint64_t o = offA[p0], l = offA[p0 + 1] - o;
        ap0 = A + o;
        len0 = l > (uint64_t)(RB * K) ? 0u : (uint32_t)l; // too long: no score here, the locate kernel reports it
    }
    if (p1 < npairs) {
        const uint64_t o = offA[p1], l = offA[p1 + 1] - o;
        ap1 = A + o;
        len1 = l > (uint64_t)(RB * K) ? 0u : (uint32_t)l;
    }
    uint32_t rpk[RB / 4]; // my band's rows
#pragma unroll
    for (int w = 0; w < RB / 4; ++w) {
        uint32_t pk = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int i = band * RB + 4 * w + h;
            const uint32_t c0 = row_code(ap0, len0, i, codeL, (uint32_t)ncodes);
            const uint32_t c1 = row_code(ap1, len1, i, codeL, (uint32_t)ncodes);
            pk |= (c0 * (uint32_t)ncp + c1) << (8 * h);
        }
        rpk[w] = pk;
    }

    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;
    uint32_t gap2 = (uint32_t)gapabs | ((uint32_t)gapabs << 16);
    if (F16) {
        const uint32_t g = half_bits(-gapabs);
        gap2 = g | (g << 16);
    }
    const uint32_t inner = band ? 0xFFFFFFFFu : 0u; // band 0 has zeros above it
    uint32_t best = 0, bestq = 0, ties = 0;
    uint32_t out0 = 0, out1 = 0, out2 = 0, out3 = 0, outd = 0, outm = 0, last3 = 0; // what the band below reads next step
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_pk));
    const uint32_t steps = nq + (K - 1); // nq real blocks, the last band K - 1 steps behind the first

    for (uint32_t s0 = 0; s0 < steps; s0 += jcb) {
        const uint32_t ns = min(jcb, steps - s0);
        __syncthreads(); // previous chunk fully consumed
        {
            // extended blocks [s0, s0 + ns + K - 1): extended block e = real block e - (K - 1)
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(prof2) + (size_t)s0 * tab_bytes);
            uint4 *dst = reinterpret_cast<uint4 *>(lds_pk);
            const uint32_t nvec = (ns + K - 1) * tab_bytes / 16;
            for (uint32_t v = tid; v < nvec; v += THREADS)
                dst[v] = src[v];
        }
        __syncthreads();
        for (uint32_t t = 0; t < ns; ++t) {
            // my block: real t + s0 - band = extended slot t + (K - 1 - band) of this chunk
            const uint32_t blk16 = (lds_base + (t + (uint32_t)(K - 1 - band)) * tab_bytes) >> 4;
            uint32_t pr0 = from_lane_above(out0) & inner, pr1 = from_lane_above(out1) & inner;
            uint32_t pr2 = from_lane_above(out2) & inner, pr3 = from_lane_above(out3) & inner;
            uint32_t pdiag = from_lane_above(outd) & inner;
            if (F16)
                pdiag = pkf_add(pdiag, gap2); // the diagonal travels with the gap taken (PH_PKF_ROW)
            const uint32_t m_in = from_lane_above(outm) & inner;
            uint32_t bm = 0;
            uint32_t pg0 = 0, pg1 = 0, pg2 = 0, pg3 = 0; // F16: the row above, gap already taken
            if (F16) {
                pg0 = pkf_addc(pr0, gap2);
                pg1 = pkf_addc(pr1, gap2);
                pg2 = pkf_addc(pr2, gap2);
                pg3 = pkf_addc(pr3, gap2);
            }
            u32x4 wa, wb;
#define PH_PK_STEP(I, W)       \
    do {                       \
        if constexpr (F16)     \
            PH_PKF_ROW(I, W);  \
        else                   \
            PH_PK_ROW(I, W);   \
    } while (0)
            PH_PK_ISSUE(wa, rpk[0], "BYTE_0");
#pragma unroll
            for (int g = 0; g < RB / 4; ++g) {
                PH_PK_ISSUE(wb, rpk[g], "BYTE_1");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PK_STEP(4 * g, wa);
                PH_PK_ISSUE(wa, rpk[g], "BYTE_2");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                PH_PK_STEP(4 * g + 1, wb);
                PH_PK_ISSUE(wb, rpk[g], "BYTE_3");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa));
                PH_PK_STEP(4 * g + 2, wa);
                if (g + 1 < RB / 4) {
                    PH_PK_ISSUE(wa, rpk[g + 1], "BYTE_0");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wb));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wb));
                }
                PH_PK_STEP(4 * g + 3, wb);
            }
            // my last row and the block maximum so far, for the band below
            out0 = pr0;
            out1 = pr1;
            out2 = pr2;
            out3 = pr3;
            outd = last3;
            last3 = pr3;
            bm = pk_max(bm, m_in); // non-negative halves order like their bit patterns
            outm = bm;
            // last band: bm is the pair's maximum over block s0 + t - (K - 1) (all-pad blocks give 0: no effect)
            const uint32_t q = s0 + t - (uint32_t)(K - 1);
            track_block_max(bm, q, best, bestq, ties);
        }
    }
#undef PH_PK_STEP
    if (band == K - 1) {
        uint32_t m0 = best & 0xFFFFu, m1 = best >> 16;
        if (F16) {
            m0 = half_score(m0);
            m1 = half_score(m1);
        }
        if (p0 < npairs) {
            infoM[p0] = m0;
            infoQ[p0] = info_q_lo(bestq, ties);
        }
        if (p1 < npairs) {
            infoM[p1] = m1;
            infoQ[p1] = info_q_hi(bestq, ties);
        }
    }
}
#undef PH_PKF_ROW
#undef PH_PK_ROW
#undef PH_PK_ISSUE

#define PH_LC_CELL(S, DIAG, UP, LEFT, HOUT, C)                                    \
    do {                                                                          \
        HOUT = max(max((DIAG) + (S), 0), max((UP), (LEFT)) + gap);                \
        if (FIND) /* the least (row, block of the window, column) worth M */       \
            key = min(key, HOUT == M ? ((uint32_t)i_ << 10) | kbv | (uint32_t)(C) : 0xFFFFFFFFu); \
    } while (0)
#define PH_LC_ROW(I, W)                                   \
    do {                                                  \
        const int i_ = (I);                               \
        const uint32_t w_ = (W);                          \
        const int s0 = (int)(int8_t)(w_);                 \
        const int s1 = (int)(int8_t)(w_ >> 8);            \
        const int s2 = (int)(int8_t)(w_ >> 16);           \
        const int s3 = (int)w_ >> 24;                     \
        const int left = H[i_];                           \
        int h0, h1, h2, h3;                               \
        PH_LC_CELL(s0, pdiag, pr0, left, h0, 0);          \
        PH_LC_CELL(s1, pr0, pr1, h0, h1, 1);              \
        PH_LC_CELL(s2, pr1, pr2, h1, h2, 2);              \
        PH_LC_CELL(s3, pr2, pr3, h2, h3, 3);              \
        pdiag = left;                                     \
        pr0 = h0;                                         \
        pr1 = h1;                                         \
        pr2 = h2;                                         \
        pr3 = h3;                                         \
        H[i_] = h3;                                       \
    } while (0)

template <int RA, int CP>
__global__ __launch_bounds__(THREADS) void sw_locate_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    uint32_t lenB, uint32_t lenB_pad, const int8_t *__restrict__ prof, const uint8_t *__restrict__ codeA,
    const uint32_t *__restrict__ binfo, int ncodes, int gap, int smax, const uint32_t *__restrict__ infoM,
    const uint32_t *__restrict__ infoQ, uint32_t *__restrict__ list, uint32_t *__restrict__ count,
    int64_t *__restrict__ score, uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err,
    int defer)
{
    static_assert(RA % 4 == 0 && RA <= 256, "RA");
    extern __shared__ __attribute__((aligned(16))) int8_t lds_lc[];
    int8_t *P = lds_lc;
    uint8_t *codeL = reinterpret_cast<uint8_t *>(lds_lc + (size_t)lenB_pad * CP);
    const int tid = threadIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(prof);
        uint4 *dst = reinterpret_cast<uint4 *>(P);
        const uint32_t nvec = lenB_pad * CP / 16;
        for (uint32_t v = tid; v < nvec; v += THREADS)
            dst[v] = src[v];
    }
    codeL[tid] = codeA[tid];
    __syncthreads();

    const uint64_t pair = (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < npairs;
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
    const uint8_t *ap = A + o0;
    // packed row codes (as sw_shared_kernel) and the first byte of A outside FirstAlphabet
    uint32_t apk[RA / 4];
    int firstbad = -1;
    uint32_t badsym = 0, a0sym = 0;
#pragma unroll
    for (int w = 0; w < RA / 4; ++w) {
        uint32_t pk = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = 4 * w + b;
            uint32_t code = (uint32_t)ncodes;
            if ((uint32_t)i < lenA) {
                const uint32_t sym = ap[i];
                if (i == 0)
                    a0sym = sym;
                code = codeL[sym];
                if (code == 0xFFu) {
                    if (firstbad < 0) {
                        firstbad = i;
                        badsym = sym;
                    }
                    code = (uint32_t)ncodes;
                }
            }
            pk |= (code * 4u) << (8 * b);
        }
        apk[w] = pk;
    }
    uint32_t e = 0;
    if (too_long) {
        e = 0xFFFFFFFFu;
    } else if (lenA > 0 && lenB > 0) { // align.go:189-191 + matrix.go:29-36: row-major first failing cell
        const uint32_t bbad = binfo[0];
        if (firstbad == 0)
            e = (1u << 8) | a0sym;
        else if (bbad != 0xFFFFFFFFu)
            e = (2u << 8) | B[bbad];
        else if (firstbad > 0)
            e = (1u << 8) | badsym;
    }
    const int M = active ? (int)infoM[pair] : 0;
    const uint32_t iq = active ? infoQ[pair] : 0u;
    const bool tie = (iq >> 31) != 0u;
    const uint32_t q = iq & 0xFFFFu;
    // columns that can feed a cell worth M in block q: lenA + (smax*lenA - M)/|gap| before its last column.  A NEAR tie (every
    // block worth M within PH_SW_NEAR_SPAN blocks of the first: sw_locate16_kernel has the argument) stretches the window to
    // the last such block and searches its last span + 1 blocks; other ties go to the full sweep.
    uint32_t span = (iq >> 16) & 0xFFu;
    uint32_t jb0 = 0, nblk = 0;
    bool near = false;
    if (active && e == 0u && M > 0 && !defer && (!tie || span <= (uint32_t)PH_SW_NEAR_SPAN)) {
        if (!tie)
            span = 0;
        const uint32_t g = (uint32_t)(-gap), top = (uint32_t)smax * lenA;
        const uint32_t need = lenA + (top > (uint32_t)M ? (top - (uint32_t)M) / g : 0u) + 4u + 4u * span;
        const uint32_t jend = 4u * (q + span) + 4u; // one past the last block's last column
        jb0 = (jend > need ? jend - need : 0u) & ~3u;
        nblk = (jend - jb0) >> 2;
        near = tie && nblk <= 255u && lenB_pad <= 131072u; // (eight bits of the key number a window's blocks; the span is exact)
        if (tie && !near)
            nblk = 0;
    }
    const bool work = active && e == 0u && M > 0 && (!tie || near);
    if (!work)
        nblk = 0;

    int H[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i)
        H[i] = 0;
    uint32_t key = 0xFFFFFFFFu;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P));
    // Lanes sweep different numbers of blocks; they are aligned at the END (a lane with fewer blocks idles first), so the
    // wave's last iteration is every lane's last block -- the only one that can hold M (earlier ones stayed below it) --
    // and the search for the first cell worth M is a second instantiation of the block body, run once per wave: the other
    // iterations pay 5 instructions per cell instead of 8.
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    auto sweep = [&](uint32_t bt, auto find_tag) {
        constexpr bool FIND = decltype(find_tag)::value;
        const uint32_t kbv = min(bt, 255u) << 2;
        (void)kbv;
        const uint32_t blk = lds_base + ((jb0 >> 2) + bt) * (CP * 4);
        int pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0, pdiag = 0;
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
            PH_LC_ROW(4 * g + 0, wa0);
            PH_LC_ROW(4 * g + 1, wa1);
            PH_LC_ROW(4 * g + 2, wa2);
            PH_LC_ROW(4 * g + 3, wa3);
            wa0 = wb0;
            wa1 = wb1;
            wa2 = wb2;
            wa3 = wb3;
        }
    };
    uint32_t wspan = near ? span : 0u; // the wave's widest near tie: its last wspan + 1 iterations carry the search
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        wspan = max(wspan, (uint32_t)__shfl_xor((int)wspan, d, 64));
    for (uint32_t t = 0; t < nmax; ++t)
        if (t >= lag) { // (nblk == 0: lag == nmax, never)
            if (t + 1u + wspan >= nmax)
                sweep(t - lag, std::true_type{});
            else
                sweep(t - lag, std::false_type{});
        }

    if (!active)
        return;
    // a tie that is not a near one, or (never expected) no cell found: the exact kernel decides
    if (e == 0u && M > 0 && ((tie && !near) || (!defer && key == 0xFFFFFFFFu))) {
        list[atomicAdd(count, 1u)] = (uint32_t)pair;
        return;
    }
    const bool hit = e == 0u && M > 0;
    score[pair] = hit ? (int64_t)M : 0;
    if (defer) { // the traceback kernel locates the cell inside block q
        endA[pair] = hit ? SW_END_DEFERRED : 0u;
        endB[pair] = hit ? 4u * q + 4u : 0u;
    } else { // key = row << 10 | block of the window << 2 | column (without a tie the block is q)
        endA[pair] = hit ? (key >> 10) + 1u : 0u;
        endB[pair] = hit ? (near ? jb0 + 4u * ((key >> 2) & 0xFFu) : 4u * q) + (key & 3u) + 1u : 0u;
    }
    err[pair] = e;
}
#undef PH_LC_ROW
#undef PH_LC_CELL

// ---- the locate step on packed half-floats, two bands of rows per lane (gfx950) ----------------------------
// Same job and outputs as sw_locate_kernel. Under the half-float condition of the packed pass (every H < 2048) the
// windowed recurrence runs on halves: one pair per lane as before (lanes sit in different windows), but rows [0, RB) in
// the low halves of the packed registers and rows [RB, 2 RB) in the high halves, the lower band one 4-column block
// behind -- band 0's last row moves from the low to the high halves between two blocks. Three instructions per cell
// PAIR (add, v_pk_maximum3_f16, clamped add) against five per cell; the profile is a table of halves P16[block][code][4]
// (two ds_read_b64 per row pair, interleaved by four v_perm_b32). Lanes are aligned at the end of their windows; the
// first cell worth M in row-major order is searched in the wave's last two iterations (band 0's last block, then band 1's).
__global__ __launch_bounds__(256) void profile16_kernel(const uint8_t *__restrict__ B, uint32_t lenB, uint32_t lenB_pad,
                                                       const int8_t *__restrict__ lutc, int ncodes, int ncp,
                                                       uint2 *__restrict__ prof16)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nq = lenB_pad / 4 + 1; // the last block is all pad: what a band reads where it has no block
    if (e >= nq * (uint32_t)ncp)
        return;
    const uint32_t q = e / (uint32_t)ncp;
    const int c = (int)(e % (uint32_t)ncp);
    uint32_t h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t j = 4 * q + u;
        int sv = -128; // pad columns / pad code: keeps every H at 0
        if (j < lenB && c < ncodes)
            sv = lutc[c * 256 + B[j]];
        h[u] = half_bits(sv);
    }
    prof16[e] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
}

#define PH_L16_CELL(W, DIAG, UPG, LEFTG, H, GOUT, C)                                          \
    do {                                                                                      \
        H = pkf_max3(pkf_add((DIAG), (W)), (UPG), (LEFTG));                                   \
        GOUT = pkf_addc(H, gap2);                                                             \
        if (FIND == 1 || FIND == 2)                                                           \
            key = (key == 0xFFFFFFFFu && ((H >> (FIND == 2 ? 16 : 0)) & 0xFFFFu) == Mh)        \
                      ? (uint32_t)(((r_ + (FIND == 2 ? RB : 0)) << 2) | (C))                  \
                      : key;                                                                  \
        if (FIND == 3) { /* several blocks may hold a cell worth M: the least (row, block, column) of both bands */ \
            key = min(key, (H & 0xFFFFu) == Mh ? ((uint32_t)r_ << 10) | kb0 | (uint32_t)(C) : 0xFFFFFFFFu);         \
            key = min(key, (H >> 16) == Mh ? ((uint32_t)(r_ + RB) << 10) | kb1 | (uint32_t)(C) : 0xFFFFFFFFu);      \
        }                                                                                     \
    } while (0)
#define PH_L16_ROW(R, X0, X1, Y0, Y1)                                       \
    do {                                                                    \
        const int r_ = (R);                                                 \
        const uint32_t w0 = __builtin_amdgcn_perm((Y0), (X0), 0x05040100u); \
        const uint32_t w1 = __builtin_amdgcn_perm((Y0), (X0), 0x07060302u); \
        const uint32_t w2 = __builtin_amdgcn_perm((Y1), (X1), 0x05040100u); \
        const uint32_t w3 = __builtin_amdgcn_perm((Y1), (X1), 0x07060302u); \
        const uint32_t left = H[r_];                                        \
        const uint32_t gl = pkf_addc(left, gap2);                           \
        uint32_t h0, h1, h2, h3, g0, g1, g2, g3;                            \
        PH_L16_CELL(w0, pdiag, pg0, gl, h0, g0, 0);                         \
        PH_L16_CELL(w1, pr0, pg1, g0, h1, g1, 1);                           \
        PH_L16_CELL(w2, pr1, pg2, g1, h2, g2, 2);                           \
        PH_L16_CELL(w3, pr2, pg3, g2, h3, g3, 3);                           \
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
    } while (0)
#define PH_L16_ADDR(dst, base, pk, SEL)                                                                    \
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL \
                 : "=v"(dst)                                                                               \
                 : "v"(base), "v"(pk))
#define PH_L16_ISSUE(pk0, pk1, X, Y)                                          \
    do {                                                                      \
        uint32_t a_[8];                                                       \
        PH_L16_ADDR(a_[0], base0, pk0, "BYTE_0");                             \
        PH_L16_ADDR(a_[1], base1, pk1, "BYTE_0");                             \
        PH_L16_ADDR(a_[2], base0, pk0, "BYTE_1");                             \
        PH_L16_ADDR(a_[3], base1, pk1, "BYTE_1");                             \
        PH_L16_ADDR(a_[4], base0, pk0, "BYTE_2");                             \
        PH_L16_ADDR(a_[5], base1, pk1, "BYTE_2");                             \
        PH_L16_ADDR(a_[6], base0, pk0, "BYTE_3");                             \
        PH_L16_ADDR(a_[7], base1, pk1, "BYTE_3");                             \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[0]) : "v"(a_[0]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[0]) : "v"(a_[1]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[1]) : "v"(a_[2]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[1]) : "v"(a_[3]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[2]) : "v"(a_[4]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[2]) : "v"(a_[5]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(X[3]) : "v"(a_[6]));         \
        asm volatile("ds_read_b64 %0, %1" : "=v"(Y[3]) : "v"(a_[7]));         \
    } while (0)

typedef uint32_t l16_u32x2 __attribute__((ext_vector_type(2)));

template <int RB>
__global__ __launch_bounds__(THREADS, 2) void sw_locate16_kernel(
    const uint8_t *__restrict__ A, const uint64_t *__restrict__ offA, uint64_t npairs, const uint8_t *__restrict__ B,
    uint32_t lenB, uint32_t lenB_pad, const uint2 *__restrict__ prof16, const uint8_t *__restrict__ codeA,
    const uint32_t *__restrict__ binfo, int ncodes, int gap, int smax, const uint32_t *__restrict__ infoM,
    const uint32_t *__restrict__ infoQ, uint32_t *__restrict__ list, uint32_t *__restrict__ count,
    int64_t *__restrict__ score, uint32_t *__restrict__ endA, uint32_t *__restrict__ endB, uint32_t *__restrict__ err,
    int defer)
{
    static_assert(RB % 4 == 0 && RB <= 76, "RB");
    constexpr int RA = 2 * RB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_l16[];
    const int ncp = ncodes + 1;
    const uint32_t nqB = lenB_pad / 4, pstride = (uint32_t)ncp * 8u;
    uint2 *P = reinterpret_cast<uint2 *>(lds_l16);
    uint8_t *codeL = lds_l16 + (size_t)(nqB + 1) * pstride;
    const int tid = threadIdx.x;
    for (uint32_t v = tid; v < (nqB + 1) * (uint32_t)ncp; v += THREADS)
        P[v] = prof16[v];
    codeL[tid] = codeA[tid];
    __syncthreads();

    const uint64_t pair = (uint64_t)blockIdx.x * THREADS + tid;
    const bool active = pair < npairs;
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
    const uint8_t *ap = A + o0;
    // byte offsets (code * 8) of my rows inside a profile block, four rows per register and band; and the first byte of
    // A outside FirstAlphabet (as sw_locate_kernel)
    uint32_t apk0[RB / 4], apk1[RB / 4];
    int firstbad = -1;
    uint32_t badsym = 0, a0sym = 0;
#pragma unroll
    for (int band = 0; band < 2; ++band) {
#pragma unroll
        for (int w = 0; w < RB / 4; ++w) {
            uint32_t pk = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = band * RB + 4 * w + b;
                uint32_t code = (uint32_t)ncodes;
                if ((uint32_t)i < lenA) {
                    const uint32_t sym = ap[i];
                    if (i == 0)
                        a0sym = sym;
                    code = codeL[sym];
                    if (code == 0xFFu) {
                        if (firstbad < 0) {
                            firstbad = i;
                            badsym = sym;
                        }
                        code = (uint32_t)ncodes;
                    }
                }
                pk |= (code * 8u) << (8 * b);
            }
            if (band == 0)
                apk0[w] = pk;
            else
                apk1[w] = pk;
        }
    }
    uint32_t e = 0;
    if (too_long) {
        e = 0xFFFFFFFFu;
    } else if (lenA > 0 && lenB > 0) { // align.go:189-191 + matrix.go:29-36: row-major first failing cell
        const uint32_t bbad = binfo[0];
        if (firstbad == 0)
            e = (1u << 8) | a0sym;
        else if (bbad != 0xFFFFFFFFu)
            e = (2u << 8) | B[bbad];
        else if (firstbad > 0)
            e = (1u << 8) | badsym;
    }
    const int M = active ? (int)infoM[pair] : 0;
    const uint32_t iq = active ? infoQ[pair] : 0u;
    const bool tie = (iq >> 31) != 0u;
    const uint32_t q = iq & 0xFFFFu;
    // A NEAR tie (round 6): every block worth M lies in [q, q + span] with span <= NEAR_SPAN (track_block_max) -- at configs[3]
    // all ties are (one alignment path coming back to its maximum a block or two later).  The window then ends behind block
    // q + span and begins where block q's window would, every cell worth M inside it is computed exactly, and the least
    // (row, column) among them is align.go:197's answer: no full sweep.  (Not when the end cell is left to the traceback.)
    constexpr uint32_t NEAR_SPAN = PH_SW_NEAR_SPAN;
    uint32_t span = (iq >> 16) & 0xFFu;
    uint32_t jb0 = 0, nblk = 0;
    bool near = false;
    if (active && e == 0u && M > 0 && !defer && (!tie || span <= NEAR_SPAN)) {
        if (!tie)
            span = 0;
        const uint32_t g = (uint32_t)(-gap), top = (uint32_t)smax * lenA;
        const uint32_t need = lenA + (top > (uint32_t)M ? (top - (uint32_t)M) / g : 0u) + 4u + 4u * span;
        const uint32_t jend = 4u * (q + span) + 4u; // one past the last block's last column
        jb0 = (jend > need ? jend - need : 0u) & ~3u;
        nblk = (jend - jb0) >> 2;
        near = tie && nblk <= 255u && nqB <= 32768u; // (the key numbers a window's blocks in eight bits; the span is exact)
        if (tie && !near)
            nblk = 0;
    }
    const bool work = active && e == 0u && M > 0 && (!tie || near);
    if (!work)
        nblk = 0;

    uint32_t H[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
        H[i] = 0;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(P));
    const uint32_t pad_base = lds_base + nqB * pstride;
    uint32_t nmax = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    const uint32_t lag = nmax - nblk;
    const uint32_t gh = half_bits(gap); // gap < 0
    const uint32_t gap2 = gh | (gh << 16);
    const uint32_t Mh = half_bits(M);
    uint32_t key = 0xFFFFFFFFu;
    // band 0's last row of the block before, already in the high halves: what band 1 finds above its first row
    uint32_t hh0 = 0, hh1 = 0, hh2 = 0, hh3 = 0, hg0 = 0, hg1 = 0, hg2 = 0, hg3 = 0, hd = 0;
    auto sweep = [&](uint32_t bt, auto find_tag) { // bt = band 0's block; band 1 works on bt - 1
        constexpr int FIND = decltype(find_tag)::value; // 0, 1 = search band 0's cells, 2 = band 1's, 3 = both, the block in the key
        const uint32_t kb0 = min(bt, 255u) << 2, kb1 = (bt >= 1u ? min(bt - 1u, 255u) : 255u) << 2;
        (void)kb0;
        (void)kb1;
        const uint32_t base0 = bt < nblk ? lds_base + ((jb0 >> 2) + bt) * pstride : pad_base;
        const uint32_t base1 = bt >= 1u ? lds_base + ((jb0 >> 2) + bt - 1u) * pstride : pad_base;
        uint32_t pr0 = hh0, pr1 = hh1, pr2 = hh2, pr3 = hh3, pg0 = hg0, pg1 = hg1, pg2 = hg2, pg3 = hg3, pdiag = hd;
        l16_u32x2 xa[4], ya[4], xb[4], yb[4];
        PH_L16_ISSUE(apk0[0], apk1[0], xa, ya);
#pragma unroll
        for (int g = 0; g < RB / 4; ++g) {
            if (g + 1 < RB / 4) {
                PH_L16_ISSUE(apk0[g + 1], apk1[g + 1], xb, yb);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]),
                               "+v"(ya[3]));
            }
            PH_L16_ROW(4 * g + 0, xa[0].x, xa[0].y, ya[0].x, ya[0].y);
            PH_L16_ROW(4 * g + 1, xa[1].x, xa[1].y, ya[1].x, ya[1].y);
            PH_L16_ROW(4 * g + 2, xa[2].x, xa[2].y, ya[2].x, ya[2].y);
            PH_L16_ROW(4 * g + 3, xa[3].x, xa[3].y, ya[3].x, ya[3].y);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[u] = xb[u];
                ya[u] = yb[u];
            }
        }
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
    // iterations 0 .. nmax (a lane runs nblk + 1 of them); the last two carry the search -- or, in a wave with a near tie, the last
    // wspan + 2 in the form that looks at both bands and keeps the least (row, block, column): a lane without a tie finds its one
    // block's cell that way too (nothing before its block q is worth M)
    uint32_t wspan = near ? span : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        wspan = max(wspan, (uint32_t)__shfl_xor((int)wspan, d, 64));
    const bool wave_near = __builtin_amdgcn_readfirstlane((int)wspan) != 0;
    for (uint32_t t = 0; t <= nmax; ++t) {
        if (nblk == 0u || t < lag)
            continue;
        if (wave_near) {
            if (t + 1u + wspan >= nmax)
                sweep(t - lag, std::integral_constant<int, 3>{});
            else
                sweep(t - lag, std::integral_constant<int, 0>{});
        } else if (t + 1u == nmax)
            sweep(t - lag, std::integral_constant<int, 1>{});
        else if (t == nmax)
            sweep(t - lag, std::integral_constant<int, 2>{});
        else
            sweep(t - lag, std::integral_constant<int, 0>{});
    }

    if (!active)
        return;
    // a tie that is not a near one, or (never expected) no cell found: the exact kernel decides
    if (e == 0u && M > 0 && ((tie && !near) || (!defer && key == 0xFFFFFFFFu))) {
        list[atomicAdd(count, 1u)] = (uint32_t)pair;
        return;
    }
    const bool hit = e == 0u && M > 0;
    score[pair] = hit ? (int64_t)M : 0;
    if (defer) { // the traceback kernel locates the cell inside block q
        endA[pair] = hit ? SW_END_DEFERRED : 0u;
        endB[pair] = hit ? 4u * q + 4u : 0u;
    } else if (wave_near) { // key = row << 10 | block of the window << 2 | column (a lane without a tie: its block is q)
        endA[pair] = hit ? (key >> 10) + 1u : 0u;
        endB[pair] = hit ? (near ? jb0 + 4u * ((key >> 2) & 0xFFu) : 4u * q) + (key & 3u) + 1u : 0u;
    } else {
        endA[pair] = hit ? (key >> 2) + 1u : 0u;
        endB[pair] = hit ? 4u * q + (key & 3u) + 1u : 0u;
    }
    err[pair] = e;
}
#undef PH_L16_ISSUE
#undef PH_L16_ADDR
#undef PH_L16_ROW
#undef PH_L16_CELL

// ---- host side ---------------------------------------------------------------------------------------
bool packed_plan(const polyhip_scoring *sc, uint64_t npairs, uint32_t max_lenA, uint64_t lenB, PackedPlan *out)
{
    PackedPlan p{};
    if (env_is("POLYHIP_SW_PACKED", '0')) // testing aid
        return false;
    const uint64_t minlen = std::min<uint64_t>(max_lenA, lenB);
    if (!(sc->int8_ok && sc->gap <= -1 && -sc->gap < 16384 && sc->smax > 0 && sc->cp <= 8 &&
          lenB > 0 && lenB < (1ull << 18) && (uint64_t)sc->smax * minlen < 30000ull))
        return false;
    if (max_lenA > 2048 || npairs >= (1ull << 32))
        return false;
    // rows per lane x lanes per pair (sw_pkb_kernel above 152 rows): the smallest tile that holds the longest read
    static const int tiles[][2] = {{64, 1}, {152, 1}, {128, 2}, {152, 2}, {128, 4}, {152, 4}, {128, 8}, {152, 8}, {128, 16}};
    // (round 6) 64 rows per lane on twice the lanes: the H column and the row codes fit 128 registers, four waves per SIMD
    // instead of two (what sw_pk1x2_kernel does for 152 rows).  POLYHIP_SW_TILE64=0: the 128-row tiles.
    static const int tiles64[][2] = {{64, 1}, {152, 1}, {64, 4}, {152, 2}, {64, 8}, {152, 4}, {64, 16}, {152, 8}, {128, 16}};
    for (const auto &t : PH_SW_TILE64_DEFAULT != env_is("POLYHIP_SW_TILE64", PH_SW_TILE64_DEFAULT ? '0' : '1') ? tiles64 : tiles)
        if ((uint32_t)(t[0] * t[1]) >= max_lenA) {
            p.rb = t[0];
            p.k = t[1];
            break;
        }
    p.ra = p.rb * p.k;
    p.skip_rows = p.k == 1 && max_lenA + 16 <= (uint32_t)p.ra; // at least four row groups to save
    // every H below 2048: the three-instruction half-float cell (POLYHIP_SW_F16=0: the int16 one, testing aid)
    // (column 0 of the profile carries score + |gap|: that sum has to be a half-float integer too)
    p.f16 = (uint64_t)sc->smax * minlen <= 2047ull && (int64_t)sc->smax - sc->gap <= 2048 && !env_is("POLYHIP_SW_F16", '0');
    p.ncp = sc->ncodes + 1;
    p.tab_bytes = (uint32_t)(p.ncp * p.ncp * 16);
    p.lenB_pad = (uint32_t)align_up(lenB, 4);
    p.nq = p.lenB_pad / 4;
    p.jcb = std::max<uint32_t>(1, std::min<uint32_t>(64, 36864u / p.tab_bytes));
    p.pk_smem = (size_t)(p.jcb + p.k - 1) * p.tab_bytes + 256;
    // one wave per workgroup, the block's table at a fixed LDS address (POLYHIP_SW_PK1=0: the chunk-staged kernel)
    p.pk1 = p.f16 && p.k == 1 && p.ncp * p.ncp <= 64 && !env_is("POLYHIP_SW_PK1", '0');
    // the two-lane form of that kernel (round 6: 128 registers, four waves per SIMD; POLYHIP_SW_PK1X2=0: one lane per two pairs)
    p.x2_rb = 0;
    if (p.pk1 && p.ra == 152 && !env_is("POLYHIP_SW_PK1X2", '0')) // (up to 64 rows the one-lane kernel holds four waves itself)
        p.x2_rb = 76;
    p.locate_smem = (size_t)p.lenB_pad * 8 + 256;
    if (p.ra <= 256 && p.locate_smem > 160 * 1024)
        return false; // the byte profile of the reference has to sit whole in LDS for step 2
    p.prof2_bytes = align_up((size_t)(p.nq + 2 * (p.k - 1)) * p.tab_bytes, 256);
    p.info_bytes = align_up((size_t)npairs * 4, 256);
    const size_t tab16 = ((size_t)p.lenB_pad / 4 + 1) * (size_t)p.ncp * 8;
    p.locate16_smem = tab16 + 256;
    p.locate16 = p.f16 && p.k == 1 && p.locate16_smem <= 79 * 1024 && !env_is("POLYHIP_SW_LOCATE16", '0');
    p.prof16_bytes = p.locate16 ? align_up(tab16, 256) : 0;
    p.work_bytes = p.prof2_bytes + 3 * p.info_bytes + 256 + p.prof16_bytes;
    *out = p;
    return true;
}

template <int RA, int K>
static int launch_packed(const polyhip_scoring *sc, const PackedPlan &p, const uint8_t *d_A, const uint64_t *d_offA,
                         uint64_t npairs, const uint8_t *d_B, uint32_t lenB, const int8_t *prof, const uint32_t *binfo,
                         uint32_t *prof2, uint32_t *infoM, uint32_t *infoQ, uint32_t *list, uint32_t *count,
                         int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, hipStream_t st, int defer)
{
    if (!p.reuse_profiles) {
        const uint32_t nqe = p.nq + 2 * (K - 1); // K - 1 all-pad blocks on either side
        const uint32_t n = nqe * (uint32_t)(p.ncp * p.ncp);
        hipLaunchKernelGGL(profile2_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_B, lenB, nqe, (uint32_t)(K - 1),
                           sc->d_lutc, sc->ncodes, p.ncp, prof2, (int)p.f16, (int)(-sc->gap));
        PH_HIP(hipGetLastError());
    }
    if (K == 1 && p.pk1 && p.x2_rb != 0 && 2 * p.x2_rb <= RA) {
        // two lanes per lane's worth of rows, four waves per SIMD (POLYHIP_SW_PK1X2=0: one lane, two waves)
        if constexpr (K == 1) {
            const uint64_t blocks = (npairs + 63) / 64;
            auto kern = p.skip_rows ? sw_pk1x2_kernel<76, true> : sw_pk1x2_kernel<76, false>;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), 2048 + 256 + 1024, st, d_A, d_offA, npairs, prof2, p.nq,
                               p.tab_bytes, p.ncp, sc->d_codeA, sc->ncodes, (int)(-sc->gap), infoM, infoQ);
            PH_HIP(hipGetLastError());
        }
    } else if (K == 1 && p.pk1) {
        if constexpr (K == 1) {
            auto kern = p.skip_rows ? sw_pk1_kernel<RA, true> : sw_pk1_kernel<RA, false>;
            const uint64_t blocks = (npairs + 127) / 128;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), 1024 + 256 + 1024, st, d_A, d_offA, npairs, prof2, p.nq,
                               p.tab_bytes, p.ncp, sc->d_codeA, sc->ncodes, (int)(-sc->gap), infoM, infoQ);
            PH_HIP(hipGetLastError());
        }
    } else if constexpr (K == 1) {
        auto kern = p.f16 ? (p.skip_rows ? sw_pk_kernel<RA, true, true> : sw_pk_kernel<RA, false, true>)
                          : (p.skip_rows ? sw_pk_kernel<RA, true, false> : sw_pk_kernel<RA, false, false>);
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)p.pk_smem));
        const uint64_t blocks = (npairs + 2 * THREADS - 1) / (2 * THREADS);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), p.pk_smem, st, d_A, d_offA, npairs, prof2, p.nq,
                           p.jcb, p.tab_bytes, p.ncp, sc->d_codeA, sc->ncodes, (int)(-sc->gap), infoM, infoQ);
        PH_HIP(hipGetLastError());
    } else {
        static_assert(RA % K == 0, "RA, K");
        auto kern = p.f16 ? sw_pkb_kernel<RA / K, K, true> : sw_pkb_kernel<RA / K, K, false>;
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)p.pk_smem));
        constexpr uint64_t per_block = 2 * THREADS / K;
        const uint64_t blocks = (npairs + per_block - 1) / per_block;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), p.pk_smem, st, d_A, d_offA, npairs, prof2, p.nq,
                           p.jcb, p.tab_bytes, p.ncp, sc->d_codeA, sc->ncodes, (int)(-sc->gap), infoM, infoQ);
        PH_HIP(hipGetLastError());
    }
    if constexpr (RA == 64 || RA == 152) {
        if (p.locate16 && !defer) { // (a deferred end cell needs no sweep: the 32-bit kernel's bookkeeping does)
            uint2 *prof16 = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(list) + p.info_bytes);
            const uint32_t n16 = (p.lenB_pad / 4 + 1) * (uint32_t)p.ncp;
            if (!p.reuse_profiles)
                hipLaunchKernelGGL(profile16_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, d_B, lenB, p.lenB_pad,
                                   sc->d_lutc, sc->ncodes, p.ncp, prof16);
            auto kern16 = sw_locate16_kernel<RA / 2>;
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)p.locate16_smem));
            const uint64_t blocks = (npairs + THREADS - 1) / THREADS;
            hipLaunchKernelGGL(kern16, dim3((unsigned)blocks), dim3(THREADS), p.locate16_smem, st, d_A, d_offA, npairs, d_B,
                               lenB, p.lenB_pad, prof16, sc->d_codeA, binfo, sc->ncodes, (int)sc->gap, (int)sc->smax, infoM,
                               infoQ, list, count, d_score, d_endA, d_endB, d_err, defer);
            PH_HIP(hipGetLastError());
            return POLYHIP_OK;
        }
    }
    if constexpr (RA <= 256) {
        auto kern = sw_locate_kernel<RA, 8>;
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)p.locate_smem));
        const uint64_t blocks = (npairs + THREADS - 1) / THREADS;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), p.locate_smem, st, d_A, d_offA, npairs, d_B, lenB,
                           p.lenB_pad, prof, sc->d_codeA, binfo, sc->ncodes, (int)sc->gap, (int)sc->smax, infoM, infoQ,
                           list, count, d_score, d_endA, d_endB, d_err, defer);
        PH_HIP(hipGetLastError());
    }
    return POLYHIP_OK;
}

// the tables packed_run builds at the front of its workspace (prof2, and the locate step's table of halves), on their own:
// a caller that runs several sub-batches through one workspace slice builds them once and sets p.reuse_profiles
int packed_profiles(const polyhip_scoring *sc, const PackedPlan &p, const uint8_t *d_B, uint32_t lenB, void *d_work,
                    hipStream_t st)
{
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint32_t *prof2 = reinterpret_cast<uint32_t *>(w + 256);
    const uint32_t nqe = p.nq + 2 * (uint32_t)(p.k - 1);
    const uint32_t n = nqe * (uint32_t)(p.ncp * p.ncp);
    hipLaunchKernelGGL(profile2_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_B, lenB, nqe, (uint32_t)(p.k - 1), sc->d_lutc,
                       sc->ncodes, p.ncp, prof2, (int)p.f16, (int)(-sc->gap));
    if (p.locate16) {
        uint2 *prof16 = reinterpret_cast<uint2 *>(w + 256 + p.prof2_bytes + 3 * p.info_bytes);
        const uint32_t n16 = (p.lenB_pad / 4 + 1) * (uint32_t)p.ncp;
        hipLaunchKernelGGL(profile16_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, d_B, lenB, p.lenB_pad, sc->d_lutc,
                           sc->ncodes, p.ncp, prof16);
    }
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

int packed_run(const polyhip_scoring *sc, const PackedPlan &p, const uint8_t *d_A, const uint64_t *d_offA,
               uint64_t npairs, const uint8_t *d_B, uint32_t lenB, const int8_t *prof, const uint32_t *binfo,
               void *d_work, int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err,
               uint32_t **list_out, uint32_t **count_out, hipStream_t st, const uint32_t **infoM_out,
               const uint32_t **infoQ_out, int defer)
{
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint32_t *count = reinterpret_cast<uint32_t *>(w);
    uint32_t *prof2 = reinterpret_cast<uint32_t *>(w + 256);
    uint32_t *infoM = reinterpret_cast<uint32_t *>(w + 256 + p.prof2_bytes);
    uint32_t *infoQ = reinterpret_cast<uint32_t *>(w + 256 + p.prof2_bytes + p.info_bytes);
    uint32_t *list = reinterpret_cast<uint32_t *>(w + 256 + p.prof2_bytes + 2 * p.info_bytes);
    PH_HIP(hipMemsetAsync(count, 0, 256, st));
    *list_out = list;
    *count_out = count;
    if (infoM_out)
        *infoM_out = infoM;
    if (infoQ_out)
        *infoQ_out = infoQ;
#define PH_PKB_CASE(RB_, K_)                                                                                              \
    if (p.rb == RB_ && p.k == K_)                                                                                         \
        return launch_packed<RB_ * K_, K_>(sc, p, d_A, d_offA, npairs, d_B, lenB, prof, binfo, prof2, infoM, infoQ, list, \
                                           count, d_score, d_endA, d_endB, d_err, st, defer);
    PH_PKB_CASE(64, 4)
    PH_PKB_CASE(64, 8)
    PH_PKB_CASE(64, 16)
    PH_PKB_CASE(152, 2)
    PH_PKB_CASE(128, 4)
    PH_PKB_CASE(152, 4)
    PH_PKB_CASE(128, 8)
    PH_PKB_CASE(152, 8)
    PH_PKB_CASE(128, 16)
#undef PH_PKB_CASE
    if (p.ra == 64)
        return launch_packed<64, 1>(sc, p, d_A, d_offA, npairs, d_B, lenB, prof, binfo, prof2, infoM, infoQ, list, count,
                                 d_score, d_endA, d_endB, d_err, st, defer);
    if (p.ra == 152)
        return launch_packed<152, 1>(sc, p, d_A, d_offA, npairs, d_B, lenB, prof, binfo, prof2, infoM, infoQ, list, count,
                                  d_score, d_endA, d_endB, d_err, st, defer);
    return launch_packed<256, 2>(sc, p, d_A, d_offA, npairs, d_B, lenB, prof, binfo, prof2, infoM, infoQ, list, count,
                              d_score, d_endA, d_endB, d_err, st, defer);
}

} // namespace k3p
} // namespace polyhip
