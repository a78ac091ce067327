// common.h -- shared host-side plumbing for libpolyhip.so (error reporting,
// HIP call checking, scoped device buffers for the host-pointer entry points).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "polyhip.h"

// ---- ablation probes ------------------------------------------------------------------------------------------------------
// The kernels carry compile-time probes (PH_ABL, PH_K2_NOWALK, PH_K2_J_*, PH_K2_F_*, PH_SELBINS, PH_SEL_ATOMIC ...) that take a
// phase out or swap an experiment in; most of them produce WRONG results by construction.  They exist for the timing harnesses
// under scripts/ (ubench/build_k1_variants.sh, build_variant.sh), which say so with -DPH_ABLATION_BUILD; a stray -D in the
// product build stops here instead of shipping a library that computes something else (round-5 advice).
#if !defined(PH_ABLATION_BUILD)
#if (defined(PH_ABL) && PH_ABL != 0) || (defined(PH_SELBINS) && PH_SELBINS != 0) || (defined(PH_SEL_ATOMIC) && PH_SEL_ATOMIC != 0) || \
    defined(PH_LDS_PAD) || defined(PH_K2_NO_SWZ) || defined(PH_K2_C4_NOPOS) || defined(PH_K2_F4_NORANK) || defined(PH_K2_F4_RANKED) ||   \
    defined(PH_K2_F4_NOSTORE) || defined(PH_K2_F4_LINSTORE) || defined(PH_K2_ZA_NOSTORE) || defined(PH_K2_J_NOCONSUME) ||                  \
    defined(PH_K2_J_NOATOM) || defined(PH_K2_J_L1) || defined(PH_K2_J_X4A) || defined(PH_K2_J_X2A) || defined(PH_K2_NOWALK) || defined(PH_K2_NOFLUSH) || defined(PH_K2_F_NOLDS) ||         \
    defined(PH_K2_F_NOSTORE) || defined(PH_K2_F_PLAIN) || defined(PH_TB_BITS_ALIGN)
#error "an ablation probe is defined in a product build: probes are for scripts/ (pass -DPH_ABLATION_BUILD there)"
#endif
#endif

namespace polyhip {

// thread-local message behind polyhip_last_error()
int set_error(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define PH_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess)                                                         \
            return ::polyhip::set_error(POLYHIP_ERR_HIP, "%s: %s (%s:%d)", #expr,     \
                                        hipGetErrorString(_e), __FILE__, __LINE__);   \
    } while (0)

#define PH_REQUIRE(cond, ...)                                                         \
    do {                                                                              \
        if (!(cond))                                                                  \
            return ::polyhip::set_error(POLYHIP_ERR_INVALID, __VA_ARGS__);            \
    } while (0)

// Device allocation that frees itself (host-pointer entry points only; the
// _dev entry points never allocate).
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf()
    {
        if (p)
            (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    void reset()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// hipMallocAsync'd scratch of a _dev entry point: freed on the stream when the entry point is through with it
// (release()), or on whichever early return comes first
struct StreamFree {
    void *p;
    hipStream_t st;
    hipError_t release()
    {
        void *q = p;
        p = nullptr;
        return q ? hipFreeAsync(q, st) : hipSuccess;
    }
    ~StreamFree()
    {
        if (p)
            (void)hipFreeAsync(p, st);
    }
};

// transform.complementTable (transform/transform.go:78-109) for UPPER-case letters; unmapped bytes -> 0x00.  (One copy for
// primers.hip, seqhash.hip and least_rotation.hip's reverse-complement view.)
__device__ __forceinline__ uint32_t dna_complement_upper(uint32_t up)
{
    switch (up) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'B': return 'V';
    case 'V': return 'B';
    case 'D': return 'H';
    case 'H': return 'D';
    case 'K': return 'M';
    case 'M': return 'K';
    case 'R': return 'Y';
    case 'Y': return 'R';
    case 'N': return 'N';
    case 'S': return 'S';
    case 'W': return 'W';
    default: return 0;
    }
}

// Waits for a stream on EVERY way out of a scope that has asynchronous device-to-host copies into its own locals (or into
// the caller's buffers) in flight: a failed HIP call further down would otherwise return while those copies are still
// pending and turn into writes to freed host memory (round-4 advice).  On the normal path the scope has synchronised
// already and this costs one more, immediate, wait.
struct SyncOnExit {
    hipStream_t st;
    explicit SyncOnExit(hipStream_t s) : st(s) {}
    SyncOnExit(const SyncOnExit &) = delete;
    SyncOnExit &operator=(const SyncOnExit &) = delete;
    ~SyncOnExit() { (void)hipStreamSynchronize(st); }
};

// Testing aids: NAME=0 (or =1) in the environment, read per call, switches one kernel variant off (or on) so that
// the parity tests can cross-check the variants; never needed in production.
inline bool env_is(const char *name, char value)
{
    const char *e = getenv(name);
    return e && e[0] == value;
}

inline hipStream_t as_stream(polyhip_stream_t s) { return static_cast<hipStream_t>(s); }

// Index of the first byte >= 0x80 in p[0, n), or n.  The host flavours refuse such input where the Go code would treat it
// as UTF-8; a byte-by-byte loop with an early exit does not vectorise (0.2 s per GB), this OR-reduction over 4 KB blocks
// does (the compiler turns it into SIMD ors) and only a failing block is searched byte by byte.
inline uint64_t first_non_ascii(const uint8_t *p, uint64_t n)
{
    uint64_t i = 0;
    for (; i + 4096 <= n; i += 4096) {
        uint8_t acc = 0;
        for (int j = 0; j < 4096; ++j)
            acc |= p[i + j];
        if (acc & 0x80)
            break;
    }
    for (; i < n; ++i)
        if (p[i] >= 0x80)
            return i;
    return n;
}

// A second stream the library keeps per calling thread, for entry points that run sub-batches of one call side by side
// (the end of one sub-batch's kernels -- waves finish at different times -- overlaps the start of the next).  fork():
// the aux stream waits for everything enqueued on the caller's stream so far; join(): the caller's stream waits for the
// aux stream.  Everything stays ordered on the caller's stream as if it had run there.
struct AuxStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int dev = -1; // the device it was made on: a thread that moves to another device gets a new one
    AuxStream() = default;
    AuxStream(const AuxStream &) = delete;
    AuxStream &operator=(const AuxStream &) = delete;
    // the thread's four cached entries go with the thread (a worker of a device list that is replaced: round-4 advice -- every
    // polyhip_set_devices leaked up to 4 streams and 8 events per worker).  A process that is exiting has lost its HIP
    // runtime already; its objects are left alone (hipStreamQuery then fails, and nothing is destroyed).
    ~AuxStream()
    {
        if (!s)
            return;
        const hipError_t q = hipStreamQuery(s);
        if (q != hipSuccess && q != hipErrorNotReady)
            return;
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
        for (int q2 = 0; q2 < 2; ++q2)
            if (ev[q2])
                (void)hipEventDestroy(ev[q2]);
    }
    hipError_t init()
    {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess)
            return e;
        if (s && ev[0] && ev[1] && dev == cur)
            return hipSuccess;
        // another device, or an earlier creation that failed half way: drop what is there and start over; `dev` marks
        // the set as usable only once the stream AND both events exist
        if (s) {
            (void)hipStreamDestroy(s);
            s = nullptr;
        }
        for (int q = 0; q < 2; ++q)
            if (ev[q]) {
                (void)hipEventDestroy(ev[q]);
                ev[q] = nullptr;
            }
        dev = -1;
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        if (e == hipSuccess)
            dev = cur;
        return e;
    }
    hipError_t fork(hipStream_t caller)
    {
        hipError_t e = init();
        if (e == hipSuccess)
            e = hipEventRecord(ev[0], caller);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s, ev[0], 0);
        return e;
    }
    hipError_t join(hipStream_t caller)
    {
        hipError_t e = hipEventRecord(ev[1], s);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(caller, ev[1], 0);
        return e;
    }
};
// one per (calling thread, caller stream): the two slots of a host-pointer pipeline run on two streams of one thread,
// and a single aux stream for both would make each slot's join wait for the other slot's forked work as well (round-3
// advice).  Four entries per thread, recycled round robin (a recycled stream simply keeps its order: fork() waits).
inline AuxStream &aux_stream(hipStream_t caller)
{
    struct Entry {
        hipStream_t key = nullptr;
        bool used = false;
        AuxStream a;
    };
    static thread_local Entry e[4];
    static thread_local unsigned next = 0;
    for (Entry &x : e)
        if (x.used && x.key == caller)
            return x.a;
    Entry &x = e[next++ & 3];
    x.used = true;
    x.key = caller;
    return x.a;
}

// value of the lane below (lane 0 gets 0): one DPP move (wave_shr:1, a GFX9 control gfx950 still has) instead of
// the ds_bpermute a __shfl_up costs -- the systolic one-wave-per-pair kernels do this every step
__device__ __forceinline__ int from_lane_below(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}

// Inclusive sum over the 64 lanes of a wave and the largest value of a wave (in every lane) on DPP moves: no ds_bpermute, and
// -- what matters in kernels at 64 registers -- no per-step lane-index registers for the compiler to hoist out of loops
__device__ __forceinline__ uint32_t dpp_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t dpp_wave_max(uint32_t v)
{
#define PH_DPP_MAX(ctrl, rows) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false))
    PH_DPP_MAX(0xb1, 0xf);  // quad_perm [1,0,3,2]
    PH_DPP_MAX(0x4e, 0xf);  // quad_perm [2,3,0,1]
    PH_DPP_MAX(0x124, 0xf); // row_ror 4
    PH_DPP_MAX(0x128, 0xf); // row_ror 8
    PH_DPP_MAX(0x142, 0xa); // row_bcast15 into rows 1, 3
    PH_DPP_MAX(0x143, 0xc); // row_bcast31 into rows 2, 3
#undef PH_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

} // namespace polyhip
