// primers.hip -- K4: SantaLucia nearest-neighbour Tm (scan + batch) and
// Marmur-Doty Tm for gfx950.
//
// Replaces primers.SantaLucia (primers/primers.go:70-105), MarmurDoty
// (:108-118) and MeltingTemp (:121-128) for
//   * a SCAN: every window of every length Lmin..Lmax at every start of one
//     long sequence (BASELINE config 5: all 18..30-mers of a 5 Mb genome), and
//   * a packed BATCH of independent primers.
//
// Bit-level parity with the Go code is kept, not just the 1e-6 C tolerance:
//   - fp64 throughout, compiled with -ffp-contract=off (Go/amd64 never fuses);
//   - every addition happens in the reference's order: init (+0.2, -5.7),
//     symmetry (-1.4), terminal A/T (+2.2, +6.9), salt term, then the
//     nearest-neighbour terms left to right.  The windows that share a start
//     share their nearest-neighbour TERMS but not their partial sums (the
//     start values differ per length), so the scan keeps one running (dH, dS)
//     pair per length in registers and feeds all of them from ONE table
//     lookup per dinucleotide: 29 LDS lookups + 598 fp64 adds per start for
//     18..30 instead of 299 lookups;
//   - the two logarithms are evaluated once per call on the host with the
//     algorithm Go's math.Log uses (FreeBSD e_log.c), so no device libm
//     rounding enters.
//
// The scan is HBM-WRITE bound: 24 B of (Tm, dH, dS) per window against
// ~0.08 B of genome read.  One thread per start, outputs in
// per-length planes so that a wave's 64 lanes store 512 contiguous bytes.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "host_pipeline.h"

namespace polyhip {
namespace k4 {

constexpr int THREADS = 256;
constexpr int NL_MAX = 16;    // lengths per launch (longer ranges are split by the host loop)
constexpr int LMAX_MAX = 1024; // longest window the scan stages

// ---- Go's math.Log (src/math/log.go = FreeBSD e_log.c) -----------------------
// log(x) = k*ln2 + log(1+f), 1+f in [sqrt2/2, sqrt2), log(1+f) = f - f^2/2 + s*(f^2/2 + R(s^2)).
static double go_log(double x)
{
    if (std::isnan(x) || (std::isinf(x) && x > 0))
        return x;
    if (x < 0)
        return std::nan("");
    if (x == 0)
        return -INFINITY;
    int e = 0;
    double m = std::frexp(x, &e); // [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2;
        --e;
    }
    const double f = m - 1;
    const double k = (double)e;
    const double s = f / (2 + f);
    const double z = s * s;
    const double w = z * z;
    const double odd = z * (6.666666666666735130e-01 +
                            w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double even = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double R = odd + even;
    const double hfsq = 0.5 * f * f;
    return k * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + k * 1.90821492927058770002e-10)) - f);
}

// per-call constants (kernel argument, lands in SGPRs / the scalar cache)
struct Consts {
    double rlog_sym;   // 1.9872 * ln(conc / 1)   self-complementary        primers.go:85,103
    double rlog_non;   // 1.9872 * ln(conc / 4)                              primers.go:87,103
    double ln_salt;    // ln(Na + 140 * Mg)                                  primers.go:94-95
    double salt[NL_MAX]; // (0.368 * float64(L-1)) * ln_salt for L = Lmin + l (scan only)
};

static Consts make_consts(double conc, double na, double mg, uint32_t Lmin, uint32_t nl)
{
    Consts c;
    const double gas = 1.9872;
    c.rlog_sym = gas * go_log(conc / 1.0);
    c.rlog_non = gas * go_log(conc / 4.0);
    const double saltEffect = na + (mg * 140);
    c.ln_salt = go_log(saltEffect);
    for (uint32_t l = 0; l < NL_MAX; ++l)
        c.salt[l] = l < nl ? (0.368 * (double)((int64_t)(Lmin + l) - 1)) * c.ln_salt : 0.0;
    return c;
}

// nearest-neighbour table indexed by code(x)*5 + code(y), codes A C G T other = 0..4
// (primers.go:42-59; a key that is not in the map reads {0, 0}, primers.go:98)
__constant__ double2 c_nn[25] = {
    /*AA*/ {-7.6, -21.3}, /*AC*/ {-8.4, -22.4}, /*AG*/ {-7.8, -21.0}, /*AT*/ {-7.2, -20.4}, {0.0, 0.0},
    /*CA*/ {-8.5, -22.7}, /*CC*/ {-8.0, -19.9}, /*CG*/ {-10.6, -27.2}, /*CT*/ {-7.8, -21.0}, {0.0, 0.0},
    /*GA*/ {-8.2, -22.2}, /*GC*/ {-9.8, -24.4}, /*GG*/ {-8.0, -19.9}, /*GT*/ {-8.4, -22.4}, {0.0, 0.0},
    /*TA*/ {-7.2, -21.3}, /*TC*/ {-8.2, -22.2}, /*TG*/ {-8.5, -22.7}, /*TT*/ {-7.6, -21.3}, {0.0, 0.0},
    {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};

__device__ __forceinline__ uint32_t ascii_upper(uint32_t b) { return (b - 'a' < 26u) ? b - 32u : b; }

__device__ __forceinline__ uint32_t nt_code(uint32_t up)
{
    return up == 'A' ? 0u : up == 'C' ? 1u : up == 'G' ? 2u : up == 'T' ? 3u : 4u;
}

// transform.complementTable restricted to upper case (the input has been upper-cased, primers.go:71): IUPAC pairs,
// everything else -> 0x00 -- dna_complement_upper, common.h

__device__ __forceinline__ double melting(double dH, double dS, double rlog)
{
    return dH * 1000 / (dS + rlog) - 273.15; // primers.go:103
}

// ---- scan ---------------------------------------------------------------------
// LMIN_CT > 0: lengths LMIN_CT..LMAX_CT known at compile time (everything unrolls,
// no predicated adds); LMIN_CT == 0: runtime Lmin, nl <= NL_MAX lengths.
//
// Round 4: who writes which 128-byte line.  The scan is bound by its write stream (24 B per window), and what that
// stream sustains depends on ONE thing (scripts/ubench/write_bw.hip, profiles/r04_write_bw.md): whether a line is written
// by one workgroup or by two.  Workgroups land on the eight XCDs in turn, each XCD has its own L2, so a line whose two
// parts come from neighbouring workgroups never merges on the chip: both parts go to memory as partial writes.  39 planes
// written in blocks that start wherever 256 * blockIdx falls reach 3.5 TB/s; the same bytes in blocks that own whole lines
// reach 5.4 TB/s -- 8 or 16 bytes per lane makes no difference, and lines shared by the waves of ONE workgroup (one CU, one
// L2) cost nothing.  The round-3 kernel's time was EXACTLY the time of its store pattern alone.
// So a workgroup computes 256 consecutive starts but OWNS, plane by plane, the 240 of them (15 lines) that begin at that
// plane's own line boundary (ld, the plane number and the caller's pointer decide the phase); its neighbour computes the
// other 16 again.  1/16 more arithmetic, and every line of every plane is written by exactly one workgroup (the first
// and last line of a plane excepted).
constexpr int SPL = 1;                     // starts per lane
constexpr int BLOCK_STARTS = THREADS * SPL; // starts a workgroup computes
constexpr int OWN_STARTS = BLOCK_STARTS - 16; // ... and owns in every plane: whole 128-byte lines of doubles

// cols [lo, hi) of plane `O + plane offset` that workgroup `b` of `nb` writes: from the first line boundary at or after
// b * OWN_STARTS to the first one at or after (b + 1) * OWN_STARTS (the plane's head and tail go to the first / last one)
__device__ __forceinline__ void owned_range(const double *plane, uint32_t b, uint32_t nb, uint64_t nstarts, uint64_t &lo, uint64_t &hi)
{
    const uint64_t phase = (reinterpret_cast<uintptr_t>(plane) >> 3) & 15u; // doubles past a line boundary at col 0
    const uint64_t r = (16u - phase) & 15u;                                 // cols = r (mod 16) begin a line
    const uint64_t T = (uint64_t)b * OWN_STARTS;
    lo = b == 0 ? 0 : T + r;
    hi = b + 1 == nb ? nstarts : T + OWN_STARTS + r;
    if (hi > nstarts)
        hi = nstarts;
}

#ifndef PH_K4_WPE
#define PH_K4_WPE 4
#endif
template <int LMIN_CT, int LMAX_CT>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(PH_K4_WPE, 8))) void scan_kernel(const uint8_t *__restrict__ seq, uint64_t len, uint64_t start0,
                                                      uint64_t nstarts, uint32_t Lmin_rt, uint32_t nl_rt, Consts cst,
                                                      double *__restrict__ tm, double *__restrict__ dHo,
                                                      double *__restrict__ dSo, uint64_t ld, double target,
                                                      uint16_t *__restrict__ firstL, double *__restrict__ firstTm)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t Lmin = LMIN_CT > 0 ? (uint32_t)LMIN_CT : Lmin_rt;
    const uint32_t nl = LMIN_CT > 0 ? (uint32_t)(LMAX_CT - LMIN_CT + 1) : nl_rt;
    const uint32_t Lmax = Lmin + nl - 1;
    const uint32_t span = BLOCK_STARTS + Lmax - 1;  // bytes this block looks at
    const uint32_t span_pad = (span + 15u) & ~15u;
    double2 *nn = reinterpret_cast<double2 *>(lds);  // 25 entries (400 B, padded to 512)
    uint8_t *up = lds + 512;                         // upper-cased bytes
    uint8_t *cp = up + span_pad;                     // complement of the upper-cased byte
    uint8_t *cd = cp + span_pad;                     // nearest-neighbour code 0..4
    uint16_t *rad = reinterpret_cast<uint16_t *>(cd + span_pad); // palindromic radius per double-centre (2 * span_pad)

    const int tid = threadIdx.x;
    const uint64_t b0 = start0 + (uint64_t)blockIdx.x * OWN_STARTS;
    if (tid < 25)
        nn[tid] = c_nn[tid];
    for (uint32_t t = tid; t < span; t += THREADS) {
        const uint64_t g = b0 + t;
        const uint32_t u = g < len ? ascii_upper(seq[g]) : 0xFFu;
        up[t] = (uint8_t)u;
        cp[t] = (uint8_t)dna_complement_upper(u);
        cd[t] = (uint8_t)nt_code(u);
    }
    __syncthreads();
    // seq == ReverseComplement(seq) (primers.go:81) for the window [a, a + L) compares the pairs (a + t, a + L - 1 - t), all
    // of which share the DOUBLE-CENTRE 2a + L - 1.  So one radius per double-centre -- how many pairs match, counted from
    // the innermost outwards, capped at what the longest window needs -- answers every (start, length) with one lookup:
    // palindrome <=> radius >= ceil(L / 2).  (Round 3 walked every window's pairs: 13 data-dependent loops of dependent
    // LDS round trips per start; a radius loop ends after 1.3 steps on average and there are two per start.)  A pair
    // matches when BOTH directions of the reference's comparison hold (the complement table is not an involution on
    // bytes it does not know).
    {
        const uint32_t rmax = (Lmax + 1) / 2;
        for (uint32_t c = tid; c + 1 < 2 * span; c += THREADS) {
            int x = (int)(c >> 1), y = (int)((c + 1) >> 1); // innermost pair: x == y (the centre base itself) for even c
            uint32_t r = 0;
            while (r < rmax && x >= 0 && y < (int)span && up[x] == cp[y] && up[y] == cp[x]) {
                ++r;
                --x;
                ++y;
            }
            rad[c] = (uint16_t)r;
        }
    }
    __syncthreads();

    const uint64_t end = start0 + nstarts;
    const uint64_t g0 = b0 + (uint64_t)SPL * tid; // my start
    const uint64_t col = g0 - start0;

    double aH[SPL][NL_MAX], aS[SPL][NL_MAX];
    bool sym[SPL][NL_MAX];
    bool endAT[NL_MAX];
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
        const uint32_t at = SPL * tid + q; // this start's first byte in the staged span
#pragma unroll
        for (int l = 0; l < NL_MAX; ++l) {
            if (LMIN_CT > 0 && l >= LMAX_CT - LMIN_CT + 1)
                break;
            const uint32_t L = Lmin + l;
            const bool pal = rad[2 * at + L - 1] >= (L + 1) / 2; // seq == ReverseComplement(seq), primers.go:81
            sym[q][l] = pal;
            double h = 0.0, s = 0.0;
            h += 0.2; // primers.go:78-79
            s += -5.7;
            if (pal) { // :82-83
                h += 0.0;
                s += -1.4;
            }
            const uint32_t last = cd[at + L - 1];
            const bool at_end = last == 0u || last == 3u;
            if (at_end) { // :89-92, 3' end is A or T
                h += 2.2;
                s += 6.9;
            }
            s += cst.salt[l]; // :95
            aH[q][l] = h;
            aS[q][l] = s;
            endAT[l] = at_end;
        }

        // nearest-neighbour terms, left to right (:97-101): one lookup per
        // dinucleotide, added to every length that still contains it
        if (LMIN_CT > 0) {
            // dH before the nearest-neighbour terms is one of TWO values whatever the length -- 0.2 (+ 0.0 for a
            // palindrome: the same double) or 0.2 + 2.2 -- and the terms are added left to right, so every length's dH is
            // a PREFIX of one of two running sums (bit for bit the reference's additions, :97-101): 2 adds per
            // dinucleotide instead of one per length that contains it (299 -> 58 per start).  dS cannot share: its
            // start value carries the length's own salt term (:95).
            double pa = 0.0, pb = 0.0;
            pa += 0.2;
            pb += 0.2;
            pb += 2.2;
            uint32_t c0 = cd[at];
#pragma unroll
            for (int j = 0; j + 1 < LMAX_CT; ++j) {
                const uint32_t c1 = cd[at + j + 1];
                const double2 t = nn[c0 * 5u + c1];
                c0 = c1;
                pa += t.x;
                pb += t.x;
                if (j + 2 >= LMIN_CT) // dinucleotide j is the last one of length j + 2
                    aH[q][j + 2 - LMIN_CT] = endAT[j + 2 - LMIN_CT] ? pb : pa;
#pragma unroll
                for (int l = 0; l < LMAX_CT - LMIN_CT + 1; ++l)
                    if (j + 1 < LMIN_CT + l)
                        aS[q][l] += t.y;
            }
        } else {
            uint32_t c0 = cd[at];
            for (uint32_t j = 0; j + 1 < Lmax; ++j) {
                const uint32_t c1 = cd[at + j + 1];
                const double2 t = nn[c0 * 5u + c1];
                c0 = c1;
#pragma unroll
                for (int l = 0; l < NL_MAX; ++l) {
                    // x + 0.0 == x for every value an accumulator can hold (never -0.0)
                    const bool in = (uint32_t)l < nl && j + 1 < Lmin + (uint32_t)l;
                    aH[q][l] += in ? t.x : 0.0;
                    aS[q][l] += in ? t.y : 0.0;
                }
            }
        }
    }

    if (firstL) {
        // the grow loop of primers/pcr (pcr.go:47-53: lengthen the primer while MeltingTemp < targetTm) as a reduction:
        // only the first length that is no longer below the target leaves the chip, 2 + 8 bytes per start instead of
        // 24 per window.  Lengths are scanned in ascending order, also across the launches of a long range.
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const uint64_t g = g0 + q;
            if (g >= end || SPL * tid + q >= OWN_STARTS || firstL[col + q] != 0) // the last 16 starts are the next workgroup's
                continue;
            bool found = false;
#pragma unroll
            for (int l = 0; l < NL_MAX; ++l) {
                if (LMIN_CT > 0 && l >= LMAX_CT - LMIN_CT + 1)
                    break;
                const uint32_t L = Lmin + l;
                if (!found && (uint32_t)l < nl && g + L <= len) {
                    const double t = melting(aH[q][l], aS[q][l], sym[q][l] ? cst.rlog_sym : cst.rlog_non);
                    if (!(t < target)) {
                        found = true;
                        firstL[col + q] = (uint16_t)L;
                        if (firstTm)
                            firstTm[col + q] = t;
                    }
                }
            }
        }
        return;
    }
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
#pragma unroll
    for (int l = 0; l < NL_MAX; ++l) {
        if (LMIN_CT > 0 && l >= LMAX_CT - LMIN_CT + 1)
            break;
        if ((uint32_t)l < nl) {
            const uint32_t L = Lmin + l;
            const uint64_t po = (uint64_t)l * ld;
            const bool fits = g0 + L <= len; // else the window runs off the end of the sequence
            const double t = fits ? melting(aH[0][l], aS[0][l], sym[0][l] ? cst.rlog_sym : cst.rlog_non) : qnan;
            const double h = fits ? aH[0][l] : qnan;
            const double e = fits ? aS[0][l] : qnan;
            uint64_t lo, hi;
            owned_range(tm + po, blockIdx.x, gridDim.x, nstarts, lo, hi);
            if (col >= lo && col < hi)
                tm[po + col] = t;
            owned_range(dHo + po, blockIdx.x, gridDim.x, nstarts, lo, hi);
            if (col >= lo && col < hi)
                dHo[po + col] = h;
            owned_range(dSo + po, blockIdx.x, gridDim.x, nstarts, lo, hi);
            if (col >= lo && col < hi)
                dSo[po + col] = e;
        }
    }
}

// ---- batch: one primer per lane, the reference's loop ---------------------------
__global__ __launch_bounds__(THREADS) void batch_kernel(const uint8_t *__restrict__ seqs,
                                                       const uint64_t *__restrict__ offs, uint64_t n, Consts cst,
                                                       double *__restrict__ tm, double *__restrict__ dHo,
                                                       double *__restrict__ dSo)
{
    __shared__ double2 nn[25];
    if (threadIdx.x < 25)
        nn[threadIdx.x] = c_nn[threadIdx.x];
    __syncthreads();
    const uint64_t p = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= n)
        return;
    const uint8_t *s = seqs + offs[p];
    const uint64_t L = offs[p + 1] - offs[p];
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    if (L == 0) { // the reference indexes sequence[-1] and panics (primers.go:89)
        tm[p] = qnan;
        dHo[p] = qnan;
        dSo[p] = qnan;
        return;
    }
    bool pal = true;
    for (uint64_t t = 0; t < L && pal; ++t)
        pal = ascii_upper(s[t]) == dna_complement_upper(ascii_upper(s[L - 1 - t]));
    double h = 0.0, e = 0.0;
    h += 0.2;
    e += -5.7;
    if (pal) {
        h += 0.0;
        e += -1.4;
    }
    const uint32_t last = ascii_upper(s[L - 1]);
    if (last == 'A' || last == 'T') {
        h += 2.2;
        e += 6.9;
    }
    e += (0.368 * (double)((int64_t)L - 1)) * cst.ln_salt;
    uint32_t c0 = nt_code(ascii_upper(s[0]));
    for (uint64_t i = 0; i + 1 < L; ++i) {
        const uint32_t c1 = nt_code(ascii_upper(s[i + 1]));
        const double2 t = nn[c0 * 5u + c1];
        c0 = c1;
        h += t.x;
        e += t.y;
    }
    tm[p] = melting(h, e, pal ? cst.rlog_sym : cst.rlog_non);
    dHo[p] = h;
    dSo[p] = e;
}

// MarmurDoty, primers.go:108-118: 2(A+T) + 4(C+G) - 7 on the upper-cased sequence
__global__ __launch_bounds__(THREADS) void marmur_doty_kernel(const uint8_t *__restrict__ seqs,
                                                             const uint64_t *__restrict__ offs, uint64_t n,
                                                             double *__restrict__ tm)
{
    const uint64_t p = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= n)
        return;
    const uint8_t *s = seqs + offs[p];
    const uint64_t L = offs[p + 1] - offs[p];
    uint64_t at = 0, cg = 0;
    for (uint64_t i = 0; i < L; ++i) {
        const uint32_t c = nt_code(ascii_upper(s[i]));
        at += (c == 0u) | (c == 3u);
        cg += (c == 1u) | (c == 2u);
    }
    // aCount..gCount are exact small integers in float64, so is the result
    tm[p] = 2 * (double)at + 4 * (double)cg - 7.0;
}

static size_t scan_smem(uint32_t Lmax)
{
    const uint32_t span_pad = (BLOCK_STARTS + Lmax - 1 + 15u) & ~15u;
    return 512 + 3 * (size_t)span_pad + 4 * (size_t)span_pad; // nn | up, cp, cd | rad (u16 per double-centre)
}

} // namespace k4
} // namespace polyhip

using namespace polyhip;

static int validate_ascii(const uint8_t *p, uint64_t n, const char *who)
{
    const uint64_t i = first_non_ascii(p, n);
    if (i < n)
        return set_error(POLYHIP_ERR_INVALID, "%s: byte 0x%02x at %llu is not ASCII (Go would case-fold it as UTF-8)", who, p[i],
                         (unsigned long long)(i + md::base().byte));
    return POLYHIP_OK;
}

// validation shared by the two packed-batch host entry points
static int check_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, const char *who, bool empty_panics)
{
    PH_REQUIRE(seqs && offsets, "%s: null pointer", who);
    for (uint64_t i = 0; i < n; ++i) {
        PH_REQUIRE(offsets[i] <= offsets[i + 1], "%s: offsets not ascending at %llu", who,
                   (unsigned long long)(i + md::base().item));
        if (empty_panics && offsets[i] == offsets[i + 1])
            return set_error(POLYHIP_ERR_PANIC, "%s: sequence %llu is empty; primers.SantaLucia(\"\") panics (primers.go:89)",
                             who, (unsigned long long)(i + md::base().item));
    }
    return validate_ascii(seqs + offsets[0], offsets[n] - offsets[0], who);
}

// Both host flavours: chunks of ~64 MB through two slots on the calling thread's two streams (host_pipeline.h); NOUT
// planes of one double per sequence come back per chunk.
template <int NOUT, class Launch>
static int run_batch_host(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, double *const (&outs)[NOUT], Launch launch)
{
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    const Chunks ch = cut_packed(offsets, n, 8 * NOUT, HOST_CHUNK_BYTES);
    struct Slot {
        PackedSlot in;
        DevBuf d[NOUT];
    } slot[2];
    for (size_t q = 0; q < std::min<size_t>(2, ch.count()); ++q) {
        PH_HIP(slot[q].in.alloc(ch, hs.s[q]));
        for (int o = 0; o < NOUT; ++o)
            PH_HIP(slot[q].d[o].alloc(ch.max_items * 8));
    }
    Duplex dx; // (as many bytes come back as go up: no helper thread, see least_rotation.hip)
    PH_HIP(dx.init(0));
    for (size_t c = 0; c < ch.count(); ++c) {
        Slot &S = slot[c & 1];
        const uint64_t i0 = ch.cut[c], m = ch.cut[c + 1] - i0;
        PH_HIP(dx.slot_free(c, S.in.st)); // chunk c-2 has left this slot
        PH_HIP(S.in.upload(seqs, offsets, i0, m));
        double *dd[NOUT];
        for (int o = 0; o < NOUT; ++o)
            dd[o] = S.d[o].template as<double>();
        const int rc = launch(S.in.dseq.template as<uint8_t>(), S.in.doff.template as<uint64_t>(), m, dd, S.in.st);
        if (rc != POLYHIP_OK) {
            (void)dx.finish();
            (void)hs.sync_both();
            return rc;
        }
        struct Planes {
            double *dst[NOUT];
            const double *src[NOUT];
        } pl;
        for (int o = 0; o < NOUT; ++o) {
            pl.dst[o] = outs[o] + i0;
            pl.src[o] = dd[o];
        }
        PH_HIP(dx.download(c, S.in.st, [=](hipStream_t st) -> hipError_t {
            hipError_t e = hipSuccess;
            for (int o = 0; o < NOUT && e == hipSuccess; ++o)
                e = hipMemcpyAsync(pl.dst[o], pl.src[o], m * 8, hipMemcpyDeviceToHost, st);
            return e;
        }));
    }
    PH_HIP(dx.finish());
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

extern "C" {

static int scan_impl(const uint8_t *d_seq, uint64_t len, uint64_t start0, uint64_t nstarts, uint32_t Lmin, uint32_t Lmax,
                     double primer_conc, double salt_conc, double mg_conc, double *d_tm, double *d_dH, double *d_dS,
                     uint64_t ld, double target, uint16_t *d_first_len, double *d_first_tm, polyhip_stream_t stream)
{
    if (Lmin == 0)
        return set_error(POLYHIP_ERR_PANIC, "primers.SantaLucia(\"\") indexes sequence[-1] (primers.go:89): the reference panics");
    PH_REQUIRE(Lmin <= Lmax, "polyhip_santalucia_scan: Lmin %u > Lmax %u", Lmin, Lmax);
    PH_REQUIRE(Lmax <= (uint32_t)k4::LMAX_MAX, "polyhip_santalucia_scan: Lmax %u > %d is not implemented", Lmax,
               k4::LMAX_MAX);
    if (nstarts == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seq && ((d_tm && d_dH && d_dS) || d_first_len), "polyhip_santalucia_scan: null pointer");
    PH_REQUIRE(d_first_len || ld >= nstarts, "polyhip_santalucia_scan: plane stride %llu < nstarts %llu", (unsigned long long)ld,
               (unsigned long long)nstarts);
    PH_REQUIRE(start0 <= len && nstarts <= len - start0 + 0, "polyhip_santalucia_scan: starts [%llu, +%llu) outside the sequence",
               (unsigned long long)start0, (unsigned long long)nstarts);
    hipStream_t st = as_stream(stream);
    const uint64_t blocks = (nstarts + k4::OWN_STARTS - 1) / k4::OWN_STARTS;
    PH_REQUIRE(blocks < (1ull << 31), "polyhip_santalucia_scan: too many starts for one call");
    if (d_first_len)
        PH_HIP(hipMemsetAsync(d_first_len, 0, nstarts * sizeof(uint16_t), st)); // 0 = no length reaches the target
    if (d_first_len && d_first_tm)
        PH_HIP(hipMemsetAsync(d_first_tm, 0xFF, nstarts * sizeof(double), st)); // all ones: a NaN where nothing is found
    for (uint32_t L0 = Lmin; L0 <= Lmax; L0 += k4::NL_MAX) {
        const uint32_t nl = Lmax - L0 + 1 < (uint32_t)k4::NL_MAX ? Lmax - L0 + 1 : (uint32_t)k4::NL_MAX;
        const k4::Consts c = k4::make_consts(primer_conc, salt_conc, mg_conc, L0, nl);
        const uint64_t plane0 = d_first_len ? 0 : (uint64_t)(L0 - Lmin) * ld;
        const size_t smem = k4::scan_smem(L0 + nl - 1);
        if (L0 == 18 && nl == 13) {
            hipLaunchKernelGGL((k4::scan_kernel<18, 30>), dim3((unsigned)blocks), dim3(k4::THREADS), smem, st, d_seq,
                               len, start0, nstarts, L0, nl, c, d_tm + plane0, d_dH + plane0, d_dS + plane0, ld, target,
                               d_first_len, d_first_tm);
        } else {
            hipLaunchKernelGGL((k4::scan_kernel<0, 0>), dim3((unsigned)blocks), dim3(k4::THREADS), smem, st, d_seq,
                               len, start0, nstarts, L0, nl, c, d_tm + plane0, d_dH + plane0, d_dS + plane0, ld, target,
                               d_first_len, d_first_tm);
        }
        PH_HIP(hipGetLastError());
    }
    return POLYHIP_OK;
}

int polyhip_santalucia_scan_dev(const uint8_t *d_seq, uint64_t len, uint64_t start0, uint64_t nstarts, uint32_t Lmin,
                                uint32_t Lmax, double primer_conc, double salt_conc, double mg_conc, double *d_tm,
                                double *d_dH, double *d_dS, uint64_t ld, polyhip_stream_t stream)
{
    PH_REQUIRE(nstarts == 0 || Lmin == 0 || (d_tm && d_dH && d_dS), "polyhip_santalucia_scan: null pointer");
    return scan_impl(d_seq, len, start0, nstarts, Lmin, Lmax, primer_conc, salt_conc, mg_conc, d_tm, d_dH, d_dS, ld, 0.0, nullptr,
                     nullptr, stream);
}

int polyhip_santalucia_scan_first_dev(const uint8_t *d_seq, uint64_t len, uint64_t start0, uint64_t nstarts, uint32_t Lmin,
                                      uint32_t Lmax, double primer_conc, double salt_conc, double mg_conc, double target_tm,
                                      uint16_t *d_first_len, double *d_first_tm, polyhip_stream_t stream)
{
    PH_REQUIRE(nstarts == 0 || Lmin == 0 || d_first_len, "polyhip_santalucia_scan_first: null pointer");
    return scan_impl(d_seq, len, start0, nstarts, Lmin, Lmax, primer_conc, salt_conc, mg_conc, nullptr, nullptr, nullptr, 0,
                     target_tm, d_first_len, d_first_tm, stream);
}

// Starts [a, b) of the host flavour's scan on the calling thread's current device: the slice of the sequence those
// windows touch (an (Lmax - 1)-byte halo on the right) goes up, the planes come back into columns [a, b) of
// the caller's planes (row stride ld = the whole scan's number of starts).
static int scan_range_one(const uint8_t *seq, uint64_t len, uint64_t a, uint64_t b, uint32_t Lmin, uint32_t Lmax,
                          double primer_conc, double salt_conc, double mg_conc, double *tm, double *dH, double *dS, uint64_t ld)
{
    if (a >= b)
        return POLYHIP_OK;
    const uint64_t nst = b - a, end = std::min<uint64_t>(len, b - 1 + Lmax), slen = end - a;
    const uint32_t nl = Lmax - Lmin + 1;
    int rc;
    {
        md::BaseScope pos(0, a);
        rc = validate_ascii(seq + a, slen, "polyhip_santalucia_scan");
    }
    if (rc != POLYHIP_OK)
        return rc;
    const uint64_t nout = nst * (uint64_t)nl;
    DevBuf dseq, dtm, ddh, dds;
    PH_HIP(dseq.alloc(slen));
    PH_HIP(dtm.alloc(nout * 8));
    PH_HIP(ddh.alloc(nout * 8));
    PH_HIP(dds.alloc(nout * 8));
    HostStreams &hs = host_streams(); // the calling thread's own stream, not the null stream
    PH_HIP(hs.init());
    PH_HIP(hipMemcpyAsync(dseq.p, seq + a, slen, hipMemcpyHostToDevice, hs.s[0]));
    rc = polyhip_santalucia_scan_dev(dseq.as<uint8_t>(), slen, 0, nst, Lmin, Lmax, primer_conc, salt_conc, mg_conc,
                                     dtm.as<double>(), ddh.as<double>(), dds.as<double>(), nst, hs.s[0]);
    if (rc != POLYHIP_OK) {
        (void)hipStreamSynchronize(hs.s[0]);
        return rc;
    }
    PH_HIP(hipStreamSynchronize(hs.s[0]));
    // three outputs, three streams: one pageable download does not fill the link (a copy engine each does).  The whole
    // scan on one device (nst == ld) is one contiguous copy per output, a shard's columns one copy per length.
    hipStream_t cs[3] = {nullptr, nullptr, nullptr};
    double *dstp[3] = {tm, dH, dS};
    const double *srcp[3] = {dtm.as<double>(), ddh.as<double>(), dds.as<double>()};
    hipError_t e = hipSuccess;
    for (int q = 0; q < 3 && e == hipSuccess; ++q) {
        e = hipStreamCreateWithFlags(&cs[q], hipStreamNonBlocking);
        if (e != hipSuccess)
            break;
        if (nst == ld) {
            e = hipMemcpyAsync(dstp[q], srcp[q], nout * 8, hipMemcpyDeviceToHost, cs[q]);
        } else {
            for (uint32_t l = 0; l < nl && e == hipSuccess; ++l)
                e = hipMemcpyAsync(dstp[q] + (uint64_t)l * ld + a, srcp[q] + (uint64_t)l * nst, nst * 8, hipMemcpyDeviceToHost, cs[q]);
        }
    }
    for (int q = 0; q < 3; ++q)
        if (cs[q]) {
            const hipError_t e2 = hipStreamSynchronize(cs[q]);
            if (e == hipSuccess)
                e = e2;
            (void)hipStreamDestroy(cs[q]);
        }
    PH_HIP(e);
    return POLYHIP_OK;
}

int polyhip_santalucia_scan(const uint8_t *seq, uint64_t len, uint32_t Lmin, uint32_t Lmax, double primer_conc,
                            double salt_conc, double mg_conc, double *tm, double *dH, double *dS)
{
    if (Lmin == 0)
        return polyhip_santalucia_scan_dev(nullptr, 0, 0, 0, 0, Lmax, primer_conc, salt_conc, mg_conc, nullptr, nullptr,
                                           nullptr, 0, nullptr);
    PH_REQUIRE(Lmin <= Lmax, "polyhip_santalucia_scan: Lmin %u > Lmax %u", Lmin, Lmax);
    PH_REQUIRE(Lmax <= (uint32_t)k4::LMAX_MAX, "polyhip_santalucia_scan: Lmax %u > %d is not implemented", Lmax, k4::LMAX_MAX);
    if (len < Lmin)
        return POLYHIP_OK; // no window fits
    PH_REQUIRE(seq && tm && dH && dS, "polyhip_santalucia_scan: null pointer");
    const uint64_t nstarts = len - Lmin + 1;
    std::shared_ptr<md::Pool> P = md::pool();
    if (!P)
        return scan_range_one(seq, len, 0, nstarts, Lmin, Lmax, primer_conc, salt_conc, mg_conc, tm, dH, dS, nstarts);
    // windows are independent -- contiguous ranges of starts, each with its halo
    const size_t nsh = md::size(*P);
    return md::run(*P, [&](size_t q) {
        const uint64_t a = (uint64_t)(((unsigned __int128)nstarts * q) / nsh), b = (uint64_t)(((unsigned __int128)nstarts * (q + 1)) / nsh);
        return scan_range_one(seq, len, a, b, Lmin, Lmax, primer_conc, salt_conc, mg_conc, tm, dH, dS, nstarts);
    });
}

static int scan_first_range_one(const uint8_t *seq, uint64_t len, uint64_t a, uint64_t b, uint32_t Lmin, uint32_t Lmax,
                                double primer_conc, double salt_conc, double mg_conc, double target_tm, uint16_t *first_len,
                                double *first_tm)
{
    if (a >= b)
        return POLYHIP_OK;
    const uint64_t nst = b - a, end = std::min<uint64_t>(len, b - 1 + Lmax), slen = end - a;
    int rc;
    {
        md::BaseScope pos(0, a);
        rc = validate_ascii(seq + a, slen, "polyhip_santalucia_scan_first");
    }
    if (rc != POLYHIP_OK)
        return rc;
    DevBuf dseq, dlen, dtm;
    PH_HIP(dseq.alloc(slen));
    PH_HIP(dlen.alloc(nst * 2));
    if (first_tm)
        PH_HIP(dtm.alloc(nst * 8));
    HostStreams &hs = host_streams(); // the calling thread's own stream, not the null stream
    PH_HIP(hs.init());
    hipStream_t st = hs.s[0];
    PH_HIP(hipMemcpyAsync(dseq.p, seq + a, slen, hipMemcpyHostToDevice, st));
    rc = polyhip_santalucia_scan_first_dev(dseq.as<uint8_t>(), slen, 0, nst, Lmin, Lmax, primer_conc, salt_conc, mg_conc,
                                           target_tm, dlen.as<uint16_t>(), first_tm ? dtm.as<double>() : nullptr, st);
    if (rc != POLYHIP_OK) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    PH_HIP(hipMemcpyAsync(first_len + a, dlen.p, nst * 2, hipMemcpyDeviceToHost, st));
    if (first_tm)
        PH_HIP(hipMemcpyAsync(first_tm + a, dtm.p, nst * 8, hipMemcpyDeviceToHost, st));
    PH_HIP(hipStreamSynchronize(st));
    return POLYHIP_OK;
}

int polyhip_santalucia_scan_first(const uint8_t *seq, uint64_t len, uint32_t Lmin, uint32_t Lmax, double primer_conc,
                                  double salt_conc, double mg_conc, double target_tm, uint16_t *first_len, double *first_tm)
{
    if (Lmin == 0)
        return polyhip_santalucia_scan_dev(nullptr, 0, 0, 0, 0, Lmax, primer_conc, salt_conc, mg_conc, nullptr, nullptr,
                                           nullptr, 0, nullptr);
    PH_REQUIRE(Lmin <= Lmax, "polyhip_santalucia_scan_first: Lmin %u > Lmax %u", Lmin, Lmax);
    PH_REQUIRE(Lmax <= (uint32_t)k4::LMAX_MAX, "polyhip_santalucia_scan: Lmax %u > %d is not implemented", Lmax, k4::LMAX_MAX);
    if (len < Lmin)
        return POLYHIP_OK; // no window fits
    PH_REQUIRE(seq && first_len, "polyhip_santalucia_scan_first: null pointer");
    const uint64_t nstarts = len - Lmin + 1;
    std::shared_ptr<md::Pool> P = md::pool();
    if (!P)
        return scan_first_range_one(seq, len, 0, nstarts, Lmin, Lmax, primer_conc, salt_conc, mg_conc, target_tm, first_len,
                                    first_tm);
    const size_t nsh = md::size(*P);
    return md::run(*P, [&](size_t q) {
        const uint64_t a = (uint64_t)(((unsigned __int128)nstarts * q) / nsh), b = (uint64_t)(((unsigned __int128)nstarts * (q + 1)) / nsh);
        return scan_first_range_one(seq, len, a, b, Lmin, Lmax, primer_conc, salt_conc, mg_conc, target_tm, first_len, first_tm);
    });
}

int polyhip_santalucia_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, double primer_conc,
                                 double salt_conc, double mg_conc, double *d_tm, double *d_dH, double *d_dS,
                                 polyhip_stream_t stream)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && d_tm && d_dH && d_dS, "polyhip_santalucia_batch: null pointer");
    const uint64_t blocks = (n + k4::THREADS - 1) / k4::THREADS;
    PH_REQUIRE(blocks < (1ull << 31), "polyhip_santalucia_batch: too many sequences for one call");
    const k4::Consts c = k4::make_consts(primer_conc, salt_conc, mg_conc, 1, 0);
    hipLaunchKernelGGL(k4::batch_kernel, dim3((unsigned)blocks), dim3(k4::THREADS), 0, as_stream(stream), d_seqs,
                       d_offsets, n, c, d_tm, d_dH, d_dS);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

static int santalucia_batch_one(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, double primer_conc, double salt_conc,
                                double mg_conc, double *tm, double *dH, double *dS)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(tm && dH && dS, "polyhip_santalucia_batch: null pointer");
    if (int rc = check_batch(seqs, offsets, n, "polyhip_santalucia_batch", true))
        return rc;
    double *const outs[3] = {tm, dH, dS};
    return run_batch_host<3>(seqs, offsets, n, outs,
                             [&](const uint8_t *ds, const uint64_t *dof, uint64_t m, double *(&d)[3], hipStream_t st) {
                                 return polyhip_santalucia_batch_dev(ds, dof, m, primer_conc, salt_conc, mg_conc, d[0], d[1], d[2], st);
                             });
}

int polyhip_santalucia_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, double primer_conc,
                             double salt_conc, double mg_conc, double *tm, double *dH, double *dS)
{
    std::shared_ptr<md::Pool> P = n ? md::pool() : nullptr;
    if (!P)
        return santalucia_batch_one(seqs, offsets, n, primer_conc, salt_conc, mg_conc, tm, dH, dS);
    PH_REQUIRE(seqs && offsets && tm && dH && dS, "polyhip_santalucia_batch: null pointer");
    const std::vector<uint64_t> cut = md::split(n, md::size(*P), [&](uint64_t i) { return offsets[i] - offsets[0] + i * 24; });
    return md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, offsets[i0] - offsets[0]);
        return santalucia_batch_one(seqs, offsets + i0, m, primer_conc, salt_conc, mg_conc, tm + i0, dH + i0, dS + i0);
    });
}

int polyhip_marmurdoty_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, double *d_tm,
                                 polyhip_stream_t stream)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && d_tm, "polyhip_marmurdoty_batch: null pointer");
    const uint64_t blocks = (n + k4::THREADS - 1) / k4::THREADS;
    PH_REQUIRE(blocks < (1ull << 31), "polyhip_marmurdoty_batch: too many sequences for one call");
    hipLaunchKernelGGL(k4::marmur_doty_kernel, dim3((unsigned)blocks), dim3(k4::THREADS), 0, as_stream(stream), d_seqs,
                       d_offsets, n, d_tm);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

static int marmurdoty_batch_one(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, double *tm)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(tm, "polyhip_marmurdoty_batch: null pointer");
    if (int rc = check_batch(seqs, offsets, n, "polyhip_marmurdoty_batch", false))
        return rc;
    double *const outs[1] = {tm};
    return run_batch_host<1>(seqs, offsets, n, outs,
                             [&](const uint8_t *ds, const uint64_t *dof, uint64_t m, double *(&d)[1], hipStream_t st) {
                                 return polyhip_marmurdoty_batch_dev(ds, dof, m, d[0], st);
                             });
}

int polyhip_marmurdoty_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, double *tm)
{
    std::shared_ptr<md::Pool> P = n ? md::pool() : nullptr;
    if (!P)
        return marmurdoty_batch_one(seqs, offsets, n, tm);
    PH_REQUIRE(seqs && offsets && tm, "polyhip_marmurdoty_batch: null pointer");
    const std::vector<uint64_t> cut = md::split(n, md::size(*P), [&](uint64_t i) { return offsets[i] - offsets[0] + i * 8; });
    return md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, offsets[i0] - offsets[0]);
        return marmurdoty_batch_one(seqs, offsets + i0, m, tm + i0);
    });
}

} // extern "C"
