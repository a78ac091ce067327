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
//           h = rotl(h ^ P, 13) * 5 + c chain, tail bytes, fmix32
//   select  hashes <= tau are appended (wave ballot + one LDS atomic per
//           wave) to a candidate buffer; tau starts from the count a uniform
//           hash would need (s + 6 sqrt(s) + 16 expected survivors) and the
//           result is VERIFIED: fewer than s survivors -> the sequence is
//           redone accepting everything, so the output is exact for any input.
//   shrink  when the buffer fills, and at the end: exact bottom-s of the
//           buffer by LDS counting sort on the top bits (2048 bins) + in-bin
//           ranking; duplicates keep distinct ranks, as the reference keeps
//           duplicate hashes.
//
// Integer ALU/LDS bound (no MFMA: this is hashing, not a contraction).
// Algorithmic HBM bytes per sequence: len + 4*s (read once, sketch written once).
#include <hip/hip_runtime.h>

#include "common.h"

namespace polyhip {
namespace k1 {

constexpr int THREADS = 256;
constexpr int TW = 2048;  // windows per tile (8 per lane = 2 groups of 4)
constexpr int GROUPS = TW / (4 * THREADS);
constexpr int NB = 2048;  // counting-sort bins
constexpr uint32_t C1 = 0xcc9e2d51u, C2 = 0x1b873593u;

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t premix(uint32_t k)
{
    k *= C1;
    k = rotl32(k, 15);
    k *= C2;
    return k;
}

__device__ __forceinline__ uint32_t chain(uint32_t h)
{
    h = rotl32(h, 13);
    return h * 5u + 0xe6546b64u;
}

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

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

struct Smem {
    uint32_t *seqb;   // tile bytes, window 0 at byte 0
    uint32_t *P;      // premixed blocks per byte position; aliased by `bins`
    uint32_t *cand;   // candidate hashes (<= tau)
    uint32_t *binned; // counting-sort scratch
    uint32_t *misc;   // [0] count  [1] tau  [2] restart flag  [4..8) wave totals
};

// Exact bottom-s of cand[0..C): leaves cand[0..s) ascending, count = s,
// tau = cand[s-1].  Requires C >= s.  All threads of the block call it.
__device__ void shrink(const Smem &sm, uint32_t s)
{
    const int tid = threadIdx.x;
    uint32_t *bins = sm.P;
    const uint32_t C = sm.misc[0];
    const uint32_t tau = sm.misc[1];
    // every candidate is <= tau: pick the shift that spreads [0, tau] over <= NB bins
    const int sig = 32 - __builtin_clz(tau | 1u); // significant bits of tau
    const int shift = sig > 11 ? sig - 11 : 0;
    __syncthreads(); // everyone has read misc / finished with P before it becomes bins

    for (int b = tid; b < NB; b += THREADS)
        bins[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < C; i += THREADS)
        atomicAdd(&bins[sm.cand[i] >> shift], 1u);
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
        }
    }
    __syncthreads();
    // scatter: afterwards bins[b] = end of bin b, start of bin b = bins[b-1]
    for (uint32_t i = tid; i < C; i += THREADS) {
        uint32_t h = sm.cand[i];
        uint32_t slot = atomicAdd(&bins[h >> shift], 1u);
        sm.binned[slot] = h;
    }
    __syncthreads();
    // rank inside the bin; ties broken by slot so duplicates get distinct ranks
    for (uint32_t j = tid; j < C; j += THREADS) {
        uint32_t h = sm.binned[j];
        uint32_t b = h >> shift;
        uint32_t start = b ? bins[b - 1] : 0u;
        if (start >= s)
            continue;
        uint32_t end = bins[b];
        uint32_t rank = 0;
        for (uint32_t x = start; x < end; ++x) {
            uint32_t o = sm.binned[x];
            rank += (o < h) || (o == h && x < j);
        }
        uint32_t pos = start + rank;
        if (pos < s)
            sm.cand[pos] = h;
    }
    __syncthreads();
    if (tid == 0) {
        sm.misc[0] = s;
        sm.misc[1] = sm.cand[s - 1];
    }
    __syncthreads();
}

// KS > 0: k known at compile time (full unroll of the block chain); KS == 0: runtime k.
template <int KS>
__global__ __launch_bounds__(THREADS) void sketch_kernel(const uint8_t *__restrict__ seqs,
                                                        const uint64_t *__restrict__ offs, uint32_t k_rt,
                                                        uint32_t s, uint32_t *__restrict__ out,
                                                        uint32_t n_seq_dw, uint32_t n_P, uint32_t cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_raw[];
    Smem sm;
    sm.seqb = smem_raw;
    sm.P = sm.seqb + n_seq_dw;
    sm.cand = sm.P + n_P;
    sm.binned = sm.cand + cap;
    sm.misc = sm.binned + cap;

    const uint32_t k = KS > 0 ? (uint32_t)KS : k_rt;
    const int nblk = (int)(k >> 2);
    const int tail = (int)(k & 3);
    const uint32_t tailmask = tail == 0 ? 0u : (0xFFFFFFFFu >> (32 - 8 * tail));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t r = blockIdx.x;
    const uint64_t o0 = offs[r], o1 = offs[r + 1];
    const int64_t n = (int64_t)(o1 - o0);
    const int64_t nwin = n - (int64_t)k; // mash.go:73: len-k windows, last k-mer skipped
    if (nwin <= 0)
        return;
    uint32_t *__restrict__ outp = out + r * (uint64_t)s;
    const bool positional = nwin < (int64_t)s; // mash.go:81-84: sketch never fills, never sorted

    // global source, dword-aligned view
    const uintptr_t g0 = (uintptr_t)(seqs + o0);
    const uint32_t gsh = (uint32_t)(g0 & 3);
    const uint32_t *__restrict__ gdw = (const uint32_t *)(g0 - gsh);
    const int64_t gbytes = n + gsh; // bytes of the aligned view that may be touched

    // initial threshold: uniform-hash estimate with a 6-sigma margin (verified below)
    uint32_t tau0 = 0xFFFFFFFFu;
    {
        uint64_t target = (uint64_t)s + 6ull * (uint64_t)__builtin_sqrtf((float)s) + 16ull;
        if ((int64_t)target < nwin)
            tau0 = (uint32_t)((target << 32) / (uint64_t)nwin);
    }

    for (int attempt = 0; attempt < 2; ++attempt) {
        if (tid == 0) {
            sm.misc[0] = 0;
            sm.misc[1] = attempt == 0 ? tau0 : 0xFFFFFFFFu;
        }
        __syncthreads();

        for (int64_t t0 = 0; t0 < nwin; t0 += TW) {
            const int64_t wleft = nwin - t0; // windows from t0 on
            // ---- make room: the tile may append up to TW candidates
            if (!positional) {
                if (sm.misc[0] + (uint32_t)TW > cap)
                    shrink(sm, s);
            }
            const uint32_t tau = sm.misc[1];

            // ---- stage: tile bytes [t0, t0 + TW + k + 4) -> seqb
            {
                const int64_t dbase = t0 >> 2; // t0 is a multiple of TW, so of 4
                for (uint32_t d = tid; d < n_seq_dw; d += THREADS) {
                    const int64_t gi = dbase + d;
                    uint32_t lo = (gi * 4 < gbytes) ? gdw[gi] : 0u;
                    uint32_t v = lo;
                    if (gsh) {
                        uint32_t hi = ((gi + 1) * 4 < gbytes) ? gdw[gi + 1] : 0u;
                        v = funnel_bytes(hi, lo, gsh);
                    }
                    sm.seqb[d] = v;
                }
            }
            __syncthreads();

            // ---- premix: P[p..p+3] for p = 4*q
            {
                const int nq = (TW >> 2) + nblk; // quads of byte positions needed
                for (int q = tid; q < nq; q += THREADS) {
                    const uint32_t d0 = sm.seqb[q], d1 = sm.seqb[q + 1];
                    uint4 p;
                    p.x = premix(d0);
                    p.y = premix(funnel_bytes(d1, d0, 1));
                    p.z = premix(funnel_bytes(d1, d0, 2));
                    p.w = premix(funnel_bytes(d1, d0, 3));
                    reinterpret_cast<uint4 *>(sm.P)[q] = p;
                }
            }
            __syncthreads();

            // ---- hash + select
            const uint4 *P4 = reinterpret_cast<const uint4 *>(sm.P);
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const int wq = tid + THREADS * g; // quad of windows inside the tile
                uint32_t h[4] = {0u, 0u, 0u, 0u};
                if (KS > 0) {
#pragma unroll
                    for (int j = 0; j < (KS >> 2); ++j) {
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
                if (tail) {
                    const uint32_t d0 = sm.seqb[wq + nblk], d1 = sm.seqb[wq + nblk + 1];
                    h[0] ^= premix(d0 & tailmask);
                    h[1] ^= premix(funnel_bytes(d1, d0, 1) & tailmask);
                    h[2] ^= premix(funnel_bytes(d1, d0, 2) & tailmask);
                    h[3] ^= premix(funnel_bytes(d1, d0, 3) & tailmask);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    h[c] = fmix32(h[c] ^ k);

                const int64_t w0 = (int64_t)4 * wq; // first window of the quad, tile-relative
                if (positional) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (w0 + c < wleft)
                            outp[t0 + w0 + c] = h[c];
                } else {
                    bool a[4];
                    uint64_t m[4];
                    uint32_t tot = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        a[c] = (w0 + c < wleft) && (h[c] <= tau);
                        m[c] = __ballot(a[c]);
                        tot += (uint32_t)__popcll(m[c]);
                    }
                    if (tot) { // wave-uniform
                        uint32_t base = 0;
                        if (lane == 0)
                            base = atomicAdd(&sm.misc[0], tot);
                        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (a[c])
                                sm.cand[base + lane_rank(m[c])] = h[c];
                            base += (uint32_t)__popcll(m[c]);
                        }
                    }
                }
            }
            __syncthreads(); // seqb / P are rewritten by the next tile
        }

        if (positional)
            return;
        if (sm.misc[0] >= s)
            break;
        // the threshold guess kept fewer than s hashes: redo, accepting everything
        __syncthreads();
    }

    shrink(sm, s);
    for (uint32_t i = tid; i < s; i += THREADS)
        outp[i] = sm.cand[i];
}

struct Launch {
    uint32_t n_seq_dw, n_P, cap;
    size_t smem_bytes;
};

static Launch plan(uint32_t k, uint32_t s)
{
    Launch L;
    const uint32_t nblk = k / 4;
    L.n_seq_dw = ((TW / 4 + nblk + 2) + 3u) & ~3u;
    L.n_P = TW + 4 * nblk;
    const uint32_t s4 = (s + 3u) & ~3u;
    L.cap = s4 + ((s / 2 + 64u + 3u) & ~3u) + TW;
    L.smem_bytes = (size_t)(L.n_seq_dw + L.n_P + 2 * L.cap + 16) * 4;
    return L;
}

template <int KS>
static int launch(const uint8_t *d_seqs, const uint64_t *d_offs, uint64_t n, uint32_t k, uint32_t s,
                  uint32_t *d_out, const Launch &L, hipStream_t st)
{
    auto kern = sketch_kernel<KS>;
    if (L.smem_bytes > 48 * 1024) {
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.smem_bytes));
    }
    // grid.x is limited to 2^31-1 blocks; batches beyond that are split by the caller loop below
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(THREADS), L.smem_bytes, st, d_seqs, d_offs, k, s, d_out,
                       L.n_seq_dw, L.n_P, L.cap);
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
    if (s < 2)
        return set_error(POLYHIP_ERR_PANIC,
                         "mash.Sketch with SketchSize %u indexes Sketches[-1] (mash.go:96,98): the reference panics",
                         s);
    PH_REQUIRE(s <= 8192, "polyhip_mash_sketch_batch: SketchSize %u > 8192 is not implemented", s);
    PH_REQUIRE(k <= 4096, "polyhip_mash_sketch_batch: KmerSize %u > 4096 is not implemented", k);
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && d_out, "polyhip_mash_sketch_batch: null pointer");
    const k1::Launch L = k1::plan(k, s);
    if (L.smem_bytes > 160 * 1024)
        return set_error(POLYHIP_ERR_UNSUPPORTED, "polyhip_mash_sketch_batch: k=%u s=%u needs %zu B of LDS", k, s,
                         L.smem_bytes);
    hipStream_t st = as_stream(stream);
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

int polyhip_mash_sketch_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s,
                              uint32_t *out)
{
    if (s < 2)
        return polyhip_mash_sketch_batch_dev(nullptr, nullptr, n, k, s, nullptr, nullptr);
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets && out, "polyhip_mash_sketch_batch: null pointer");
    for (uint64_t i = 0; i < n; ++i)
        PH_REQUIRE(offsets[i] <= offsets[i + 1], "polyhip_mash_sketch_batch: offsets not ascending at %llu",
                   (unsigned long long)i);
    const uint64_t b0 = offsets[0], nbytes = offsets[n] - b0;
    DevBuf dseq, doff, dout;
    PH_HIP(dseq.alloc(nbytes + 16));
    PH_HIP(doff.alloc((n + 1) * sizeof(uint64_t)));
    PH_HIP(dout.alloc(n * (uint64_t)s * sizeof(uint32_t)));
    // offsets rebased so the device copy starts at byte 0
    {
        uint64_t *tmp = new uint64_t[n + 1];
        for (uint64_t i = 0; i <= n; ++i)
            tmp[i] = offsets[i] - b0;
        hipError_t e = hipMemcpy(doff.p, tmp, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
        delete[] tmp;
        PH_HIP(e);
    }
    PH_HIP(hipMemcpy(dseq.p, seqs + b0, nbytes, hipMemcpyHostToDevice));
    // `out` is in/out (prior Sketches survive where the reference leaves them)
    PH_HIP(hipMemcpy(dout.p, out, n * (uint64_t)s * sizeof(uint32_t), hipMemcpyHostToDevice));
    int rc = polyhip_mash_sketch_batch_dev(dseq.as<uint8_t>(), doff.as<uint64_t>(), n, k, s, dout.as<uint32_t>(),
                                           nullptr);
    if (rc != POLYHIP_OK)
        return rc;
    PH_HIP(hipStreamSynchronize(nullptr));
    PH_HIP(hipMemcpy(out, dout.p, n * (uint64_t)s * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return POLYHIP_OK;
}

} // extern "C"
