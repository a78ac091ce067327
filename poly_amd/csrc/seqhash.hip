// seqhash.hip -- S2: batched seqhash.Hash for gfx950.
//
// Replaces seqhash.Hash (seqhash/seqhash.go:141-224) for a packed batch that shares one
// (sequenceType, circular, doubleStranded) triple -- the shape of clone's dedup loop
// (clone/clone.go:269-320) and of bulk database hashing:
//   normalise  strings.ToUpper (:143), RNA: U -> T (:146-148), alphabet check with the first
//              offending letter (:156-176)                                   [prepare_kernel]
//   strands    transform.ReverseComplement (transform.go:15-23,78-109) when doubleStranded
//   rotate     RotateSequence (:127-138) of each strand when circular       [K5, least_rotation.hip]
//   choose     sort.Strings(...)[0]: the bytewise smaller candidate (:180-193)
//   hash       BLAKE3-256 (lukechampine.com/blake3 v1.1.5 Sum256, :221): 1 KiB chunks of 16
//              64-byte blocks, chaining values merged pairwise level by level (with the odd one
//              promoted this IS BLAKE3's left-full tree), ROOT flag on the last compression
//   format     "v1_" + {D,R,P}{C,L}{D,S} + "_" + 64 hex digits (:196-222), 71 characters
//
// One workgroup per sequence for the byte passes, one 64-thread workgroup per sequence for
// BLAKE3 (a lane per chunk, then a lane per parent).  Integer/byte work, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "host_pipeline.h"

#ifndef PH_S2_PAIR_BLOCKS
#define PH_S2_PAIR_BLOCKS 0 // 1: chunk_cv takes two blocks per round of loads (round 6: measured SLOWER, 1.06 against 0.93 ms -- 86 registers, 5 waves per SIMD)
#endif

namespace polyhip {
namespace s2 {

constexpr int THREADS = 256;
constexpr int BTHREADS = 64;
constexpr uint32_t CHUNK = 1024;
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

__constant__ uint32_t c_iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

#define PH_G(a, b, c, d, mx, my)   \
    do {                           \
        a = a + b + (mx);          \
        d = rotr(d ^ a, 16);       \
        c = c + d;                 \
        b = rotr(b ^ c, 12);       \
        a = a + b + (my);          \
        d = rotr(d ^ a, 8);        \
        c = c + d;                 \
        b = rotr(b ^ c, 7);        \
    } while (0)

// BLAKE3 compression function; out[0..8) = the new chaining value (first 8 output words)
__device__ void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                         uint32_t flags, uint32_t out[8])
{
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        m[i] = block[i];
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
    uint32_t s8 = c_iv[0], s9 = c_iv[1], s10 = c_iv[2], s11 = c_iv[3];
    uint32_t s12 = (uint32_t)counter, s13 = (uint32_t)(counter >> 32), s14 = block_len, s15 = flags;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        PH_G(s0, s4, s8, s12, m[0], m[1]);
        PH_G(s1, s5, s9, s13, m[2], m[3]);
        PH_G(s2, s6, s10, s14, m[4], m[5]);
        PH_G(s3, s7, s11, s15, m[6], m[7]);
        PH_G(s0, s5, s10, s15, m[8], m[9]);
        PH_G(s1, s6, s11, s12, m[10], m[11]);
        PH_G(s2, s7, s8, s13, m[12], m[13]);
        PH_G(s3, s4, s9, s14, m[14], m[15]);
        if (r < 6) { // message permutation 2 6 3 10 7 0 4 13 1 11 12 5 9 14 15 8
            uint32_t t[16];
            t[0] = m[2]; t[1] = m[6]; t[2] = m[3]; t[3] = m[10]; t[4] = m[7]; t[5] = m[0]; t[6] = m[4]; t[7] = m[13];
            t[8] = m[1]; t[9] = m[11]; t[10] = m[12]; t[11] = m[5]; t[12] = m[9]; t[13] = m[14]; t[14] = m[15];
            t[15] = m[8];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                m[i] = t[i];
        }
    }
    out[0] = s0 ^ s8;
    out[1] = s1 ^ s9;
    out[2] = s2 ^ s10;
    out[3] = s3 ^ s11;
    out[4] = s4 ^ s12;
    out[5] = s5 ^ s13;
    out[6] = s6 ^ s14;
    out[7] = s7 ^ s15;
}

// A candidate of seqhash.go:180-193 as it is READ (round 5): the normalised strand itself, or -- rc -- its reverse complement,
// byte p of which is cmpT[data[len - 1 - p]].  Round 4 wrote the second strand out (0.5 GB per 100k x 5 kb, a kernel's worth
// of traffic) only for K5, the comparison and BLAKE3 to read it back; now all three read it through this view.
struct Strand {
    const uint8_t *data; // the normalised sequence
    uint64_t len;
    bool rc;
    const uint8_t *cmpT; // complement of every byte value (LDS), used when rc
    __device__ __forceinline__ uint32_t byte(uint64_t p) const { return rc ? cmpT[data[len - 1 - p]] : data[p]; }
};

// 64 bytes from an arbitrary address: aligned dwords funnelled to the data's own alignment (the 17th dword is only touched
// when it holds bytes of the block)
// (the pointer reaches here through `Strand`, where the compiler loses sight of its address space and would emit FLAT loads:
// those count on the LDS counter as well, so every wait for a byte-map lookup would also wait for the block's global loads.
// PH_S2_GLOBAL_LOADS=0 keeps the flat form -- measured in profiles/r06_seqhash_global_loads.log)
#ifndef PH_S2_GLOBAL_LOADS
#define PH_S2_GLOBAL_LOADS 1
#endif
#if PH_S2_GLOBAL_LOADS
typedef const __attribute__((address_space(1))) uint32_t *gptr32_t;
#else
typedef const uint32_t *gptr32_t;
#endif
__device__ __forceinline__ void load16(const uint8_t *p, uint32_t (&o)[16])
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t sh = (uint32_t)(addr & 3u);
    gptr32_t gd = (gptr32_t)(addr - sh);
    uint32_t d[17];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        d[i] = gd[i];
    d[16] = sh ? gd[16] : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        o[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
}

// bytes [p, p + 64) of the strand as 16 little-endian words; p may be negative or run past the end by up to 63 bytes (the
// bytes there are the neighbouring sequence's or the workspace's padding: the caller masks them)
__device__ __forceinline__ void strand_block(const Strand &S, int64_t p, uint32_t (&w)[16])
{
    if (!S.rc) {
        load16(S.data + p, w);
    } else {
        // strand bytes p .. p+63 are the complements of data[len-1-p], data[len-2-p], ...: the 64 bytes that END at
        // data[len-1-p], reversed
        uint32_t d[16];
        load16(S.data + ((int64_t)S.len - 64 - p), d);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t x = d[15 - i];
            w[i] = (uint32_t)S.cmpT[x >> 24] | ((uint32_t)S.cmpT[(x >> 16) & 0xFFu] << 8) | ((uint32_t)S.cmpT[(x >> 8) & 0xFFu] << 16) |
                   ((uint32_t)S.cmpT[x & 0xFFu] << 24);
        }
    }
}

// 128 bytes at once (round 6): two consecutive blocks of a chunk from ONE round of loads.  A thread's blocks are 64 bytes
// apart and a compression (~1,500 instructions) apart in time, so a 128-byte line used to be asked for two or three times
// -- by then it had often left the caches (FETCH_SIZE 0.93 GB for 0.5 GB of sequence).
__device__ __forceinline__ void load32(const uint8_t *p, uint32_t (&o)[32])
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t sh = (uint32_t)(addr & 3u);
    gptr32_t gd = (gptr32_t)(addr - sh);
    uint32_t d[33];
#pragma unroll
    for (int i = 0; i < 32; ++i)
        d[i] = gd[i];
    d[32] = sh ? gd[32] : 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        o[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
}
// bytes [p, p + 128) of the strand, 0 <= p, p + 128 <= len
__device__ __forceinline__ void strand_block2(const Strand &S, int64_t p, uint32_t (&w)[32])
{
    if (!S.rc) {
        load32(S.data + p, w);
    } else {
        uint32_t d[32];
        load32(S.data + ((int64_t)S.len - 128 - p), d);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t x = d[31 - i];
            w[i] = (uint32_t)S.cmpT[x >> 24] | ((uint32_t)S.cmpT[(x >> 16) & 0xFFu] << 8) | ((uint32_t)S.cmpT[(x >> 8) & 0xFFu] << 16) |
                   ((uint32_t)S.cmpT[x & 0xFFu] << 24);
        }
    }
}

// chaining value of chunk `idx` of the string S[rot..len) + S[0..rot) -- the rotation `rot` of the strand, never
// materialised (round 3 wrote both strands' rotated copies out, 1 GB per 100k x 5 kb, only to read them back here) --;
// root = this chunk is the whole input
__device__ void chunk_cv(const Strand &S, uint64_t rot, uint64_t idx, bool root, uint32_t cv[8])
{
    const uint64_t len = S.len;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        cv[i] = c_iv[i];
    const uint64_t base = idx * CHUNK;
    const uint64_t clen = len - base < CHUNK ? len - base : CHUNK; // 0 only for the empty input
    const uint32_t nblocks = clen == 0 ? 1u : (uint32_t)((clen + 63) / 64);
    for (uint32_t b = 0; b < nblocks; ++b) {
        const uint64_t off = base + (uint64_t)b * 64;
        const uint32_t blen = (uint32_t)(clen - (uint64_t)b * 64 < 64 ? clen - (uint64_t)b * 64 : 64);
        const uint64_t src = off + rot >= len ? off + rot - len : off + rot; // where the block starts in the strand (rot < len, off < len)
#if PH_S2_PAIR_BLOCKS
        if (clen - (uint64_t)b * 64 >= 128 && src + 128 <= len) { // two whole blocks that lie inside the strand: one round of loads
            uint32_t w2[32], wa[16], wb[16], o[8];
            strand_block2(S, (int64_t)src, w2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                wa[i] = w2[i];
                wb[i] = w2[16 + i];
            }
            compress(cv, wa, idx, 64u, b == 0 ? (uint32_t)CHUNK_START : 0u, o);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                cv[i] = o[i];
            ++b;
            uint32_t flags = 0;
            if (b == nblocks - 1)
                flags |= CHUNK_END | (root ? (uint32_t)ROOT : 0u);
            compress(cv, wb, idx, 64u, flags, o);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                cv[i] = o[i];
            continue;
        }
#endif
        uint32_t w[16];
        if (blen == 64) {
            strand_block(S, (int64_t)src, w); // (past the end of the strand these are foreign bytes: masked below)
            if (src + 64 > len) {
                // the one block per rotated sequence that runs over the end of the strand: its first m bytes are the string's
                // last ones, the rest its first ones -- a second funnelled load, placed so that byte i of the block is byte
                // i of both, and a select per dword (a byte-by-byte path here stalled the whole wave behind the one lane
                // in sixty-four that needed it: 0.20 -> 0.45 ms for the chunk kernel)
                const uint32_t m = (uint32_t)(len - src); // 1 .. 63
                uint32_t v[16];
                strand_block(S, -(int64_t)m, v); // >= 63 bytes in front of / behind the sequence: a neighbour or the padding
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t lo = 4u * (uint32_t)i;
                    if (lo >= m)
                        w[i] = v[i];
                    else if (lo + 4u > m) {
                        const uint32_t keep = (1u << (8u * (m - lo))) - 1u;
                        w[i] = (w[i] & keep) | (v[i] & ~keep);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t p = 4 * i + j;
                    if (p < blen)
                        x |= S.byte(src + p >= len ? src + p - len : src + p) << (8 * j); // the string's last, short block
                }
                w[i] = x;
            }
        }
        uint32_t flags = 0;
        if (b == 0)
            flags |= CHUNK_START;
        if (b == nblocks - 1) {
            flags |= CHUNK_END;
            if (root)
                flags |= ROOT;
        }
        uint32_t o[8];
        compress(cv, w, idx, blen, flags, o);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            cv[i] = o[i];
    }
}

__device__ __forceinline__ uint32_t ascii_upper(uint32_t b) { return (b - 'a' < 26u) ? b - 32u : b; }

__device__ __forceinline__ bool in_set(uint32_t c, const char *set)
{
    for (const char *p = set; *p; ++p)
        if ((uint32_t)(uint8_t)*p == c)
            return true;
    return false;
}

// 4 bytes at an arbitrary address: one or two aligned dwords and a funnel shift (the second dword is
// only touched when it holds some of the 4 bytes)
__device__ __forceinline__ uint32_t load4(const uint8_t *p)
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t sh = (uint32_t)(addr & 3u);
    const uint32_t *gd = reinterpret_cast<const uint32_t *>(addr - sh);
    const uint32_t d0 = gd[0], d1 = sh ? gd[1] : 0u;
    return __builtin_amdgcn_alignbyte(d1, d0, sh);
}

// normalise + validate (+ reverse complement).  err[q] = 0 or (code << 8) | first offending letter.
// Byte maps live in LDS (normalised letter, its complement, membership in the alphabet); the sequence is
// walked in whole OUTPUT dwords (unaligned 4-byte loads on the input side), the few bytes in front of the
// first and behind the last aligned dword singly.
__global__ __launch_bounds__(THREADS) void prepare_kernel(const uint8_t *__restrict__ seqs,
                                                         const uint64_t *__restrict__ offs, uint64_t n, int seq_type,
                                                         uint8_t *__restrict__ norm, uint32_t *__restrict__ err,
                                                         unsigned long long *__restrict__ first_non_ascii,
                                                         const uint32_t *__restrict__ any_bad)
{
    // (the reverse complement is not written any more: K5, the comparison and BLAKE3 read it through `Strand`)
    if (any_bad && *any_bad == 0u)
        return; // normalise_stream_kernel did the whole batch and met no letter outside the alphabet: err[] stays zero
    __shared__ unsigned long long first_bad; // (position << 8) | letter
    __shared__ unsigned long long first_hi;  // position of the sequence's first byte >= 0x80 (the host flavour refuses those)
    __shared__ uint8_t upL[256], okL[256];
    {
        const uint32_t b = threadIdx.x; // THREADS == 256
        uint32_t c = ascii_upper(b);
        if (seq_type == 1 && c == 'U') // seqhash.go:146-148
            c = 'T';
        upL[b] = (uint8_t)c;
        okL[b] = seq_type == 2 ? in_set(c, "ACDEFGHIKLMNPQRSTVWYUO*BXZ") : in_set(c, "ATUGCYRSWKMBDHVNZ");
    }
    for (uint64_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint64_t o0 = offs[q], len = offs[q + 1] - o0;
        const uint8_t *src = seqs + o0;
        __syncthreads();
        if (threadIdx.x == 0) {
            first_bad = ~0ull;
            first_hi = ~0ull;
        }
        __syncthreads();
        auto flag = [&](uint64_t t, uint32_t b) { // a letter outside the alphabet (rare: off the fast path)
            atomicMin(&first_bad, ((unsigned long long)t << 8) | upL[b]);
            if (b & 0x80u)
                atomicMin(&first_hi, (unsigned long long)t);
        };
        auto one = [&](uint64_t t) { // byte t of the sequence -> norm[t]
            const uint32_t b = src[t], c = upL[b];
            if (!okL[b])
                flag(t, b);
            norm[o0 + t] = (uint8_t)c;
        };
        if (len < 16) {
            for (uint64_t t = threadIdx.x; t < len; t += THREADS)
                one(t);
        } else {
            // norm: aligned output dwords [head, head + 4*nd)
            {
                uint8_t *dst = norm + o0;
                const uint64_t head = (4 - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u;
                const uint64_t nd = (len - head) >> 2, tail0 = head + 4 * nd;
                if (threadIdx.x < head || (threadIdx.x >= 4 && threadIdx.x - 4 < len - tail0)) {
                    const uint64_t t = threadIdx.x < 4 ? threadIdx.x : tail0 + (threadIdx.x - 4);
                    const uint32_t b = src[t];
                    if (!okL[b])
                        flag(t, b);
                    dst[t] = upL[b];
                }
                uint32_t *od = reinterpret_cast<uint32_t *>(dst + head);
                for (uint64_t d = threadIdx.x; d < nd; d += THREADS) {
                    const uint64_t t = head + 4 * d;
                    const uint32_t w = load4(src + t);
                    uint32_t o = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t b = (w >> (8 * k)) & 0xFFu;
                        if (!okL[b])
                            flag(t + k, b);
                        o |= (uint32_t)upL[b] << (8 * k);
                    }
                    od[d] = o;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            err[q] = first_bad == ~0ull ? 0u : (((seq_type == 2 ? 3u : 2u) << 8) | (uint32_t)(first_bad & 0xFF));
            if (first_hi != ~0ull) // byte offset inside the batch: the host flavour names it and refuses the call
                atomicMin(first_non_ascii, (unsigned long long)(o0 - offs[0]) + first_hi);
        }
    }
}

// The same normalisation as prepare_kernel for a batch WITHOUT a letter outside its alphabet -- the common case -- as one
// streaming pass over the batch's bytes, 16 per thread, whatever sequence they belong to (the normalised copy has the batch's
// own byte offsets): strings.ToUpper (:143) and U -> T (:146-148) through one 16-bit table entry per byte, bit 8 of which
// says "not in the alphabet".  Such a byte only raises *any_bad: prepare_kernel -- a workgroup per sequence, two barriers and
// three atomics per sequence: 0.33 ms per 100k x 5 kb, latency-bound -- then runs after all and reports the first offending
// letter of every sequence exactly as before; without one it returns at once and err[] stays zero.
__global__ __launch_bounds__(THREADS) void normalise_stream_kernel(const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ offs,
                                                                  uint64_t n, int seq_type, uint8_t *__restrict__ norm,
                                                                  uint32_t *__restrict__ any_bad)
{
    const uint64_t b0 = offs[0], b1 = offs[n]; // the batch's bytes
    __shared__ uint16_t upT[256];
    {
        const uint32_t b = threadIdx.x; // THREADS == 256
        uint32_t c = ascii_upper(b);
        if (seq_type == 1 && c == 'U')
            c = 'T';
        const bool ok = seq_type == 2 ? in_set(c, "ACDEFGHIKLMNPQRSTVWYUO*BXZ") : in_set(c, "ATUGCYRSWKMBDHVNZ");
        upT[b] = (uint16_t)(c | (ok ? 0u : 0x100u));
    }
    __syncthreads();
    // whole 16-byte pieces of the normalised copy [a0, a1) inside [b0, b1); the few bytes in front and behind singly
    const uint64_t a0 = (b0 + 15) & ~15ull, a1 = b1 & ~15ull;
    uint32_t bad = 0;
    const uint64_t gtid = (uint64_t)blockIdx.x * THREADS + threadIdx.x, nth = (uint64_t)gridDim.x * THREADS;
    if (a0 < a1) {
        for (uint64_t p = gtid; p < (a1 - a0) >> 4; p += nth) {
            uint4 v;
            __builtin_memcpy(&v, seqs + a0 + 16 * p, 16); // (the caller's buffer may sit at any byte address)
            uint32_t w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t e0 = upT[w[i] & 0xFFu], e1 = upT[(w[i] >> 8) & 0xFFu], e2 = upT[(w[i] >> 16) & 0xFFu], e3 = upT[w[i] >> 24];
                bad |= e0 | e1 | e2 | e3;
                o[i] = (e0 & 0xFFu) | ((e1 & 0xFFu) << 8) | ((e2 & 0xFFu) << 16) | (e3 << 24);
            }
            *reinterpret_cast<uint4 *>(norm + a0 + 16 * p) = make_uint4(o[0], o[1], o[2], o[3]); // (norm is 256-byte aligned)
        }
    }
    if (a0 < a1) {
        if (gtid < 32) { // up to 15 bytes in front of a0 and up to 15 behind a1
            const uint64_t t = gtid < 16 ? b0 + gtid : a1 + (gtid - 16);
            if (gtid < 16 ? t < a0 : t < b1) {
                const uint32_t e = upT[seqs[t]];
                bad |= e;
                norm[t] = (uint8_t)e;
            }
        }
    } else { // no whole aligned piece (a batch of up to 30 bytes): every byte singly
        for (uint64_t t = b0 + gtid; t < b1; t += nth) {
            const uint32_t e = upT[seqs[t]];
            bad |= e;
            norm[t] = (uint8_t)e;
        }
    }
    if (__ballot((bad & 0x100u) != 0u) != 0ull && (threadIdx.x & 63) == 0)
        atomicOr(any_bad, 1u);
}

// sort.Strings(...)[0] (seqhash.go:180-193): sel[q] = 1 when the second candidate -- the (rotated) reverse complement, read
// through the complement table -- is the bytewise smaller one.  One wave per sequence, 64 bytes per step, out at the first
// difference.
__global__ __launch_bounds__(THREADS) void select_kernel(const uint8_t *__restrict__ norm, const uint64_t *__restrict__ offs, uint64_t n,
                                                        const uint64_t *__restrict__ rot0, const uint64_t *__restrict__ rot1,
                                                        uint32_t *__restrict__ sel)
{
    __shared__ uint8_t cmpT[256];
    cmpT[threadIdx.x] = (uint8_t)dna_complement_upper(threadIdx.x); // THREADS == 256
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * (THREADS / 64);
    for (uint64_t q = (uint64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6); q < n; q += nw) {
        const uint64_t o0 = offs[q], len = offs[q + 1] - o0;
        const uint64_t r0 = rot0 ? rot0[q] : 0, r1 = rot1 ? rot1[q] : 0; // the candidates are rotations of the two strands
        uint32_t pick = 0;
        for (uint64_t t0 = 0; t0 < len; t0 += 64) {
            const uint64_t t = t0 + lane;
            const uint64_t ta = t + r0 >= len ? t + r0 - len : t + r0, tb = t + r1 >= len ? t + r1 - len : t + r1;
            const uint32_t a = t < len ? norm[o0 + ta] : 0u, b = t < len ? cmpT[norm[o0 + (len - 1 - tb)]] : 0u;
            const uint64_t ne = __ballot(a != b);
            if (ne) {
                const int f = __builtin_ctzll(ne);
                pick = (uint32_t)__builtin_amdgcn_readlane((int)b, f) < (uint32_t)__builtin_amdgcn_readlane((int)a, f);
                break;
            }
        }
        if (lane == 0)
            sel[q] = pick;
    }
}

// Where sequence q's chaining values live: slot cv_base(q) = (bytes in front of it) / 1024 + q of the value buffer.  No
// scan: consecutive slots are at least ceil(len / 1024) apart (floor((a + len) / 1024) - floor(a / 1024) >= ceil(len / 1024)
// - 1), and the last one stays below total_bytes / 1024 + n, the buffer's size.  (Round 3 ran a one-workgroup scan of the
// chunk counts in front of every batch: 0.16 ms of the 1.95 ms a 100k x 5 kb batch took.)
__device__ __forceinline__ uint64_t cv_base(uint64_t bytes_before, uint64_t q) { return bytes_before / CHUNK + q; }

// chaining values of every chunk of every multi-chunk sequence, ONE THREAD PER CHUNK over the whole batch
// (a chunk's 16 blocks chain, so the chunk is the unit of parallelism): slot g belongs to the sequence q
// with cv_base(q) <= g < cv_base(q + 1) (binary search), its value goes to level A of that sequence's tree.
__global__ __launch_bounds__(THREADS) void chunk_kernel(const uint8_t *__restrict__ norm, int two_strands,
                                                       const uint64_t *__restrict__ offs, uint64_t n, uint64_t max_chunks,
                                                       const uint64_t *__restrict__ rot0, const uint64_t *__restrict__ rot1,
                                                       const uint32_t *__restrict__ sel, const uint32_t *__restrict__ err,
                                                       uint32_t *__restrict__ cvbuf)
{
    __shared__ uint8_t cmpT[256];
    cmpT[threadIdx.x] = (uint8_t)dna_complement_upper(threadIdx.x); // THREADS == 256
    __syncthreads();
    const uint64_t g = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (g >= max_chunks)
        return;
    const uint64_t first = offs[0];
    uint64_t lo = 0, hi = n; // last q with cv_base(q) <= g
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (cv_base(offs[mid] - first, mid) <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint64_t q = lo, o0 = offs[q], len = offs[q + 1] - o0;
    const uint64_t nchunks = len == 0 ? 1 : (len + CHUNK - 1) / CHUNK;
    const uint64_t qbase = cv_base(o0 - first, q);
    const uint64_t c = g - qbase;
    if (c >= nchunks || nchunks == 1 || err[q] != 0u)
        return; // behind the batch's last chunk; single-chunk sequences are the root compression's business
    const bool second = two_strands && sel[q];
    const Strand S{norm + o0, len, second, cmpT};
    const uint64_t rot = second ? (rot1 ? rot1[q] : 0) : (rot0 ? rot0[q] : 0);
    uint32_t cv[8];
    chunk_cv(S, rot, c, false, cv);
    uint32_t *A = cvbuf + qbase * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        A[c * 8 + i] = cv[i];
}

// "v1_" + flags + "_" + hex(root), seqhash.go:221-222
__device__ __forceinline__ void format_hash(const uint32_t (&root)[8], uint32_t prefix_letters, char *__restrict__ o)
{
    o[0] = 'v';
    o[1] = '1';
    o[2] = '_';
    o[3] = (char)(prefix_letters & 0xFF);
    o[4] = (char)((prefix_letters >> 8) & 0xFF);
    o[5] = (char)((prefix_letters >> 16) & 0xFF);
    o[6] = '_';
    const char *hex = "0123456789abcdef";
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) {
            const uint32_t byte = (root[i] >> (8 * j)) & 0xFF; // little-endian words
            o[7 + (4 * i + j) * 2] = hex[byte >> 4];
            o[7 + (4 * i + j) * 2 + 1] = hex[byte & 15];
        }
    o[71] = 0;
}

// The tree above the chunks for sequences of up to SMALL_CHUNKS chunks (64 kB), ONE THREAD PER SEQUENCE: a 5 kb plasmid
// has five chunk values and four parent compressions, one after the other -- a workgroup per sequence (round 3) kept one
// or two of its 64 lanes busy through three barriers and cost 0.44 ms per 100k sequences, the lanes of this kernel are all
// busy and it costs a twentieth of that.  Level-by-level pairing with the odd value promoted (= BLAKE3's left-full tree),
// ping-pong between the two halves of the sequence's slot range.
constexpr uint64_t SMALL_CHUNKS = 64;
__global__ __launch_bounds__(THREADS) void hash_small_kernel(const uint8_t *__restrict__ norm, int two_strands,
                                                            const uint64_t *__restrict__ offs, uint64_t n,
                                                            const uint64_t *__restrict__ rot0, const uint64_t *__restrict__ rot1,
                                                            uint32_t *__restrict__ cvbuf, const uint32_t *__restrict__ sel,
                                                            const uint32_t *__restrict__ err, uint32_t prefix_letters,
                                                            char *__restrict__ out)
{
    __shared__ uint8_t cmpT[256];
    cmpT[threadIdx.x] = (uint8_t)dna_complement_upper(threadIdx.x); // THREADS == 256
    __syncthreads();
    const uint64_t q = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (q >= n)
        return;
    char *o = out + q * 72;
    if (err[q] != 0u) {
        o[0] = 0;
        return;
    }
    const uint64_t o0 = offs[q], len = offs[q + 1] - o0;
    const uint64_t nchunks = len == 0 ? 1 : (len + CHUNK - 1) / CHUNK;
    if (nchunks > SMALL_CHUNKS)
        return; // hash_kernel's
    uint32_t root[8];
    if (nchunks == 1) {
        const bool second = two_strands && sel[q];
        const Strand S{norm + o0, len, second, cmpT};
        chunk_cv(S, len ? (second ? (rot1 ? rot1[q] : 0) : (rot0 ? rot0[q] : 0)) : 0, 0, true, root);
    } else {
        uint32_t *A = cvbuf + cv_base(o0 - offs[0], q) * 16, *B = A + nchunks * 8; // level A was filled by chunk_kernel
        uint64_t m = nchunks;
        uint32_t *src = A, *dst = B;
        uint32_t iv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            iv[i] = c_iv[i];
        while (m > 2) {
            const uint64_t pairs = m / 2;
            for (uint64_t p = 0; p < pairs; ++p) {
                uint32_t blk[16], cv[8];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    blk[i] = src[p * 16 + i];
                compress(iv, blk, 0, 64, PARENT, cv);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    dst[p * 8 + i] = cv[i];
            }
            if (m & 1) // the odd one is promoted unchanged
                for (int i = 0; i < 8; ++i)
                    dst[pairs * 8 + i] = src[(m - 1) * 8 + i];
            m = pairs + (m & 1);
            uint32_t *t = src;
            src = dst;
            dst = t;
        }
        uint32_t blk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            blk[i] = src[i];
        compress(iv, blk, 0, 64, PARENT | ROOT, root);
    }
    format_hash(root, prefix_letters, o);
}

// the same for sequences of MORE than SMALL_CHUNKS chunks: a workgroup per sequence, a level's pairs side by side
__global__ __launch_bounds__(BTHREADS) void hash_kernel(const uint64_t *__restrict__ offs, uint64_t n,
                                                       uint32_t *__restrict__ cvbuf, const uint32_t *__restrict__ sel,
                                                       const uint32_t *__restrict__ err, uint32_t prefix_letters,
                                                       char *__restrict__ out)
{
    const int tid = threadIdx.x;
    for (uint64_t q = blockIdx.x; q < n; q += gridDim.x) {
        char *o = out + q * 72;
        const uint64_t o0 = offs[q], len = offs[q + 1] - o0;
        const uint64_t nchunks = len == 0 ? 1 : (len + CHUNK - 1) / CHUNK;
        if (nchunks <= SMALL_CHUNKS || err[q] != 0u)
            continue; // hash_small_kernel's
        uint32_t *A = cvbuf + cv_base(o0 - offs[0], q) * 16, *B = A + nchunks * 8; // two levels of chaining values
        uint32_t root[8];
        {
            // level A was filled by chunk_kernel
            uint64_t m = nchunks;
            uint32_t *src = A, *dst = B;
            while (m > 2) {
                __threadfence_block();
                __syncthreads();
                const uint64_t pairs = m / 2;
                for (uint64_t p = tid; p < pairs; p += BTHREADS) {
                    uint32_t blk[16], cv[8], iv[8];
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        blk[i] = src[p * 16 + i];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        iv[i] = c_iv[i];
                    compress(iv, blk, 0, 64, PARENT, cv);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        dst[p * 8 + i] = cv[i];
                }
                if ((m & 1) && tid == 0) // the odd one is promoted unchanged
                    for (int i = 0; i < 8; ++i)
                        dst[pairs * 8 + i] = src[(m - 1) * 8 + i];
                m = pairs + (m & 1);
                uint32_t *t = src;
                src = dst;
                dst = t;
            }
            __threadfence_block();
            __syncthreads();
            if (tid == 0) {
                uint32_t blk[16], iv[8];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    blk[i] = src[i];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    iv[i] = c_iv[i];
                compress(iv, blk, 0, 64, PARENT | ROOT, root);
            }
        }
        if (tid == 0)
            format_hash(root, prefix_letters, o);
    }
}

struct Layout {
    size_t off_norm, off_rc, off_rot0, off_rot1, off_rotidx, off_cvoff, off_sel, off_cv, total;
};

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static Layout layout(uint64_t n, uint64_t total_bytes, int circular, int ds)
{
    Layout L;
    size_t o = 256; // front padding: a block that wraps around a rotated sequence loads up to 63 bytes in front of it
    L.off_norm = o; o += al(total_bytes + 16);
    L.off_rc = o; // (round 5: the reverse complement is read through the complement table, never written)
    (void)ds;
    L.off_rot0 = o; // (the rotated copies of round 3 are gone: a rotation is an index, applied where the bytes are read)
    L.off_rot1 = o;
    L.off_rotidx = o; o += circular ? al(2 * n * 8) : 0; // least-rotation index of the strand and of its reverse complement
    L.off_cvoff = o; o += al((n + 1) * 8);
    L.off_sel = o; o += al(n * 4);
    // two levels of 8-word chaining values per chunk; chunks <= total_bytes / 1024 + n
    L.off_cv = o; o += al((total_bytes / s2::CHUNK + n + 1) * 16 * 4);
    L.total = o;
    return L;
}

} // namespace s2
} // namespace polyhip

using namespace polyhip;

extern "C" {

int polyhip_least_rotation_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint64_t max_len,
                                     uint64_t *d_rot_index, uint8_t *d_rotated, polyhip_stream_t stream);

size_t polyhip_seqhash_workspace_bytes(uint64_t n, uint64_t total_bytes, int circular, int double_stranded)
{
    return s2::layout(n, total_bytes, circular, double_stranded).total;
}

int polyhip_seqhash_batch_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint64_t total_bytes,
                              uint64_t max_len, int seq_type, int circular, int double_stranded, char *d_out,
                              uint32_t *d_err, void *d_work, size_t work_bytes, polyhip_stream_t stream)
{
    PH_REQUIRE(seq_type >= 0 && seq_type <= 2,
               "Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType code: %d", seq_type); // seqhash.go:152
    PH_REQUIRE(!(seq_type == 2 && double_stranded), "Proteins cannot be double stranded");                 // seqhash.go:175
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(d_seqs && d_offsets && d_out && d_err && d_work, "polyhip_seqhash_batch: null pointer");
    const s2::Layout L = s2::layout(n, total_bytes, circular, double_stranded);
    PH_REQUIRE(work_bytes >= L.total, "polyhip_seqhash_batch: workspace too small (%zu < %zu)", work_bytes, L.total);
    hipStream_t st = as_stream(stream);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint8_t *norm = w + L.off_norm;
    uint64_t *rotidx = reinterpret_cast<uint64_t *>(w + L.off_rotidx);
    const uint64_t *r0 = nullptr, *r1 = nullptr;
    uint32_t *cvbuf = reinterpret_cast<uint32_t *>(w + L.off_cv);
    uint32_t *sel = reinterpret_cast<uint32_t *>(w + L.off_sel);

    const unsigned blocks = (unsigned)std::min<uint64_t>(n, 256ull * 32ull);
    // the workspace's first word: batch offset of the first byte >= 0x80, all ones if there is none (what the reference
    // would treat as UTF-8; polyhip_seqhash_batch reads it back and refuses the call, a device-pointer caller may)
    unsigned long long *non_ascii = reinterpret_cast<unsigned long long *>(w);
    PH_HIP(hipMemsetAsync(non_ascii, 0xFF, 8, st));
    // the streaming pass first (POLYHIP_S2_STREAM=0: the per-sequence pass alone, testing aid); the per-sequence pass behind it
    // only works when a letter outside the alphabet was met
    uint32_t *any_bad = reinterpret_cast<uint32_t *>(w + 8);
    const bool streaming = !env_is("POLYHIP_S2_STREAM", '0');
    // (round 6) circular sequences that a wave of K5 takes alone: K5 normalises the bytes while it stages them and writes the
    // normalised copy -- no streaming pass in front of it (one read of the batch and one launch less).  POLYHIP_S2_FOLD=0: off.
    const bool fold = streaming && circular && k5_wave_takes_all(max_len) && !env_is("POLYHIP_S2_FOLD", '0');
    if (streaming) {
        PH_HIP(hipMemsetAsync(any_bad, 0, 4, st));
        PH_HIP(hipMemsetAsync(d_err, 0, n * 4, st));
        const uint64_t pieces = (total_bytes >> 4) + 2;
        if (!fold)
            hipLaunchKernelGGL(s2::normalise_stream_kernel, dim3((unsigned)std::min<uint64_t>((pieces + s2::THREADS - 1) / s2::THREADS, 256ull * 16ull)),
                               dim3(s2::THREADS), 0, st, d_seqs, d_offsets, n, seq_type, norm, any_bad);
    }
    if (!fold)
        hipLaunchKernelGGL(s2::prepare_kernel, dim3(blocks), dim3(s2::THREADS), 0, st, d_seqs, d_offsets, n, seq_type, norm, d_err,
                           non_ascii, streaming ? any_bad : (uint32_t *)nullptr);
    PH_HIP(hipGetLastError());
    if (circular) { // the indexes only: neither the rotated strings nor the second strand are ever written -- ONE K5 pass
                    // stages every sequence once and searches it in both reading directions
        const int r = fold ? k5_least_rotation_strands_dev(d_seqs, d_offsets, n, max_len, rotidx, nullptr,
                                                           double_stranded ? rotidx + n : nullptr, stream, seq_type, norm, any_bad)
                           : k5_least_rotation_strands_dev(norm, d_offsets, n, max_len, rotidx, nullptr,
                                                           double_stranded ? rotidx + n : nullptr, stream);
        if (r != POLYHIP_OK)
            return r;
        if (fold) { // a letter outside the alphabet was met: the per-sequence pass names it (and the first byte >= 0x80)
            hipLaunchKernelGGL(s2::prepare_kernel, dim3(blocks), dim3(s2::THREADS), 0, st, d_seqs, d_offsets, n, seq_type, norm, d_err,
                               non_ascii, any_bad);
            PH_HIP(hipGetLastError());
        }
        r0 = rotidx;
        if (double_stranded)
            r1 = rotidx + n;
    }
    const uint32_t letters = (uint32_t)(seq_type == 0 ? 'D' : seq_type == 1 ? 'R' : 'P') |
                             ((uint32_t)(circular ? 'C' : 'L') << 8) | ((uint32_t)(double_stranded ? 'D' : 'S') << 16);
    const int two = double_stranded ? 1 : 0;
    if (two) {
        const unsigned sblocks = (unsigned)std::min<uint64_t>((n + 3) / 4, 256ull * 32ull);
        hipLaunchKernelGGL(s2::select_kernel, dim3(sblocks), dim3(s2::THREADS), 0, st, norm, d_offsets, n, r0, r1, sel);
    }
    {
        const uint64_t max_chunks = total_bytes / s2::CHUNK + n + 1; // >= the batch's chunk count
        hipLaunchKernelGGL(s2::chunk_kernel, dim3((unsigned)((max_chunks + s2::THREADS - 1) / s2::THREADS)),
                           dim3(s2::THREADS), 0, st, norm, two, d_offsets, n, max_chunks, r0, r1, sel, d_err, cvbuf);
    }
    hipLaunchKernelGGL(s2::hash_small_kernel, dim3((unsigned)((n + s2::THREADS - 1) / s2::THREADS)), dim3(s2::THREADS), 0, st, norm, two,
                       d_offsets, n, r0, r1, cvbuf, sel, d_err, letters, d_out);
    if (max_len > s2::SMALL_CHUNKS * s2::CHUNK) // some sequence has more chunks than one thread should merge
        hipLaunchKernelGGL(s2::hash_kernel, dim3(blocks), dim3(s2::BTHREADS), 0, st, d_offsets, n, cvbuf, sel, d_err, letters, d_out);
    PH_HIP(hipGetLastError());
    return POLYHIP_OK;
}

static int seqhash_batch_one(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, int seq_type, int circular,
                             int double_stranded, char *out, uint32_t *err)
{
    PH_REQUIRE(seq_type >= 0 && seq_type <= 2,
               "Only sequenceTypes of DNA, RNA, or PROTEIN allowed. Got sequenceType code: %d", seq_type);
    PH_REQUIRE(!(seq_type == 2 && double_stranded), "Proteins cannot be double stranded");
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets && out && err, "polyhip_seqhash_batch: null pointer");
    uint64_t max_len = 0;
    for (uint64_t i = 0; i < n; ++i) {
        PH_REQUIRE(offsets[i] <= offsets[i + 1], "polyhip_seqhash_batch: offsets not ascending at %llu",
                   (unsigned long long)(i + md::base().item));
        max_len = std::max(max_len, offsets[i + 1] - offsets[i]);
    }
    (void)max_len;
    // Bytes >= 0x80 (refused: Go would treat them as UTF-8) are found by the DEVICE's normalising pass, which reads every
    // byte anyway, and reported through the first word of the workspace -- the host scan of round 3 was 11 of the 27 ms
    // a 100k x 5 kb batch took from host memory.
    // Chunks of ~64 MB through two slots on the calling thread's two streams: chunk c is uploaded and hashed while the
    // seqhashes of chunk c-1 travel back (its download is issued AFTER chunk c's kernels: a download into pageable memory
    // holds the host until the data is there).
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    const Chunks ch = cut_packed(offsets, n, 76, HOST_CHUNK_BYTES);
    struct Slot {
        PackedSlot in;
        DevBuf dout, derr, dwork;
    } slot[2];
    const size_t wb = polyhip_seqhash_workspace_bytes(ch.max_items, ch.max_bytes, circular, double_stranded);
    for (size_t q = 0; q < std::min<size_t>(2, ch.count()); ++q) {
        PH_HIP(slot[q].in.alloc(ch, hs.s[q]));
        PH_HIP(slot[q].dout.alloc(ch.max_items * 72));
        PH_HIP(slot[q].derr.alloc(ch.max_items * 4));
        PH_HIP(slot[q].dwork.alloc(wb));
    }
    std::vector<unsigned long long> hi(ch.count(), ~0ull); // per chunk: offset of its first byte >= 0x80
    SyncOnExit wait0(hs.s[0]), wait1(hs.s[1]); // (downloads into `hi` and the caller's buffers may be in flight on any way out)
    auto download = [&](size_t c) -> hipError_t {
        Slot &S = slot[c & 1];
        const uint64_t i0 = ch.cut[c], m = ch.cut[c + 1] - i0;
        hipError_t e = hipMemcpyAsync(out + i0 * 72, S.dout.p, m * 72, hipMemcpyDeviceToHost, S.in.st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(err + i0, S.derr.p, m * 4, hipMemcpyDeviceToHost, S.in.st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(&hi[c], S.dwork.p, 8, hipMemcpyDeviceToHost, S.in.st);
        return e;
    };
    for (size_t c = 0; c < ch.count(); ++c) {
        Slot &S = slot[c & 1];
        const uint64_t i0 = ch.cut[c], m = ch.cut[c + 1] - i0, cb = offsets[i0 + m] - offsets[i0];
        PH_HIP(hipStreamSynchronize(S.in.st)); // chunk c-2 has left this slot
        PH_HIP(S.in.upload(seqs, offsets, i0, m));
        uint64_t ml = 0;
        for (uint64_t i = 0; i < m; ++i)
            ml = std::max(ml, offsets[i0 + i + 1] - offsets[i0 + i]);
        const int rc = polyhip_seqhash_batch_dev(S.in.dseq.as<uint8_t>(), S.in.doff.as<uint64_t>(), m, cb, ml, seq_type, circular,
                                                 double_stranded, S.dout.as<char>(), S.derr.as<uint32_t>(), S.dwork.p, wb, S.in.st);
        if (rc != POLYHIP_OK) {
            (void)hs.sync_both();
            return rc;
        }
        if (c > 0)
            PH_HIP(download(c - 1));
    }
    PH_HIP(download(ch.count() - 1));
    PH_HIP(hs.sync_both());
    for (size_t c = 0; c < ch.count(); ++c)
        if (hi[c] != ~0ull) {
            const uint64_t at = offsets[ch.cut[c]] - offsets[0] + hi[c];
            return set_error(POLYHIP_ERR_INVALID, "polyhip_seqhash_batch: byte 0x%02x at %llu is not ASCII (Go would case-fold it as UTF-8)",
                             seqs[offsets[0] + at], (unsigned long long)(at + md::base().byte));
        }
    return POLYHIP_OK;
}

int polyhip_seqhash_batch(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, int seq_type, int circular,
                          int double_stranded, char *out, uint32_t *err)
{
    std::shared_ptr<md::Pool> P = n ? md::pool() : nullptr;
    if (!P || seq_type < 0 || seq_type > 2 || (seq_type == 2 && double_stranded)) // argument errors: the one-device path words them
        return seqhash_batch_one(seqs, offsets, n, seq_type, circular, double_stranded, out, err);
    PH_REQUIRE(seqs && offsets && out && err, "polyhip_seqhash_batch: null pointer");
    const std::vector<uint64_t> cut = md::split(n, md::size(*P), [&](uint64_t i) { return offsets[i] - offsets[0] + i * 76; });
    return md::run(*P, [&](size_t q) {
        const uint64_t i0 = cut[q], m = cut[q + 1] - i0;
        md::BaseScope pos(i0, offsets[i0] - offsets[0]);
        return seqhash_batch_one(seqs, offsets + i0, m, seq_type, circular, double_stranded, out + i0 * 72, err + i0);
    });
}

} // extern "C"
