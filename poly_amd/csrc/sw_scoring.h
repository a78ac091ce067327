// sw_scoring.h -- the align.Scoring handle shared by the SW score pass
// (sw_batch.hip) and the traceback (sw_traceback.hip).
#pragma once

// The byte-profile kernels (score pass, locate, traceback) fetch the profile dwords of four rows at once: address of
// row r's dword = block base `blk` (a uint32_t LDS byte address in scope) + code * 4, the code byte picked out of the
// packed row register by one SDWA add; hand-placed so that the loads run a group of rows ahead of their use.
#define PH_PROF_ADDR(dst, pk, SEL)                                                                         \
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL \
                 : "=v"(dst)                                                                               \
                 : "v"(blk), "v"(pk))
#define PH_PROF_ISSUE(pk, w0, w1, w2, w3)                         \
    do {                                                          \
        uint32_t a0_, a1_, a2_, a3_;                              \
        PH_PROF_ADDR(a0_, pk, "BYTE_0");                          \
        PH_PROF_ADDR(a1_, pk, "BYTE_1");                          \
        PH_PROF_ADDR(a2_, pk, "BYTE_2");                          \
        PH_PROF_ADDR(a3_, pk, "BYTE_3");                          \
        asm volatile("ds_read_b32 %0, %1" : "=v"(w0) : "v"(a0_)); \
        asm volatile("ds_read_b32 %0, %1" : "=v"(w1) : "v"(a1_)); \
        asm volatile("ds_read_b32 %0, %1" : "=v"(w2) : "v"(a2_)); \
        asm volatile("ds_read_b32 %0, %1" : "=v"(w3) : "v"(a3_)); \
    } while (0)

// w = 2 * w + (x > y): the compare's carry shifted into a word (direction bits of the tracebacks; two instructions)
#define PH_CARRY_BIT(w, x, y)                                                           \
    asm volatile("v_cmp_gt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" \
                 : "+v"(w)                                                              \
                 : "v"(x), "v"(y)                                                       \
                 : "vcc")

#include <cstdint>
#include <mutex>
#include <vector>

struct polyhip_scoring {
    int64_t gap;
    int32_t lut[65536];
    uint8_t validA[256], validB[256];
    int ncodes;          // valid A symbols
    uint8_t codeA[256];  // byte -> code, 0xFF = not in FirstAlphabet
    int ncodesB;         // valid B symbols
    uint8_t codeB[256];  // byte -> code, 0xFF = not in SecondAlphabet
    int cp;              // profile bytes per column (>= ncodes + 1, multiple of 4)
    int32_t smin, smax;  // over valid (a, b) pairs
    int32_t absmax;      // max(|smin|, |smax|, |gap|)
    bool int8_ok;
    int device;
    // device tables
    int8_t *d_lutc;      // [ncodes][256] int8 (only if int8_ok)
    uint8_t *d_codeA;    // [256]
    uint8_t *d_codeB;    // [256]
    int32_t *d_lutcc;    // [ncodes + 1][ncodesB + 1] compact int32 table (last row/col: zeros for pad codes)
    int32_t *d_lut;      // [256][256]
    uint8_t *d_validA, *d_validB;
    // copies of this handle on the other devices of the library's device list (multi_device.h): made by the first
    // fan-out worker that needs one, owned and destroyed by this handle
    std::mutex *rep_m;
    std::vector<polyhip_scoring *> *rep;
};

namespace polyhip {
// the handle (or its copy) whose tables live on the calling thread's current device; null + polyhip_last_error() on failure
const polyhip_scoring *scoring_here(const polyhip_scoring *sc);
// polyhip_sw_last_path & friends are per-thread; a fan-out runs the kernels on worker threads, so the wrapper carries
// the first non-empty shard's answers back to the caller's thread (sw_batch.hip / sw_traceback.hip own the variables)
struct KernelChoice {
    int sw_path = 0, sw_half = 0, tb_path = 0, tb_half = 0, nw_path = 0;
};
KernelChoice kernel_choice_get();            // this thread's
void kernel_choice_set(const KernelChoice &); // ... becomes this
namespace k3 {
void score_choice(int *path, int *half, bool set); // sw_batch.hip's two variables
}
} // namespace polyhip

// ---- packed score pass (sw_packed.hip), driven from polyhip_sw_batch_dev (sw_batch.hip) ----------------
#include <hip/hip_runtime.h>

namespace polyhip {
namespace k3p {

struct PackedPlan {
    int ra, ncp;                    // rows per pair (template: rb * k), codes incl. pad
    int rb, k;                      // rows per lane; lanes per pair: 1, or 2..16 above 152 rows (sw_pkb_kernel)
    bool skip_rows;                 // the longest read leaves >= 16 rows of the tile unused: sw_pk_kernel<RA, true>
    bool pk1;                       // sw_pk1_kernel: one wave per workgroup, the block's table at a fixed LDS address
    int x2_rb;                      // sw_pk1x2_kernel<76, skip_rows>: the two-lane form is taken (65..152 rows), 0: not
    bool f16;                       // every H < 2048: sw_pk_kernel's half-float cells (3 instructions instead of 4)
    uint32_t tab_bytes;             // bytes of one block's table: ncp * ncp * 16
    uint32_t lenB_pad, nq, jcb;     // columns (multiple of 4), 4-column blocks, blocks per LDS chunk
    size_t pk_smem, locate_smem;
    size_t prof2_bytes, info_bytes; // workspace pieces
    bool locate16;                  // sw_locate16_kernel: half-float, two bands of rows per lane (f16, k == 1, table fits twice per CU)
    size_t prof16_bytes, locate16_smem;
    bool reuse_profiles;            // the tables at the front of the workspace are there already (packed_profiles)
    size_t work_bytes;              // 256 (tie counter) + prof2 + infoM + infoQ + tie list
};

// false: the batch does not qualify (see sw_packed.hip) or POLYHIP_SW_PACKED=0
bool packed_plan(const polyhip_scoring *sc, uint64_t npairs, uint32_t max_lenA, uint64_t lenB, PackedPlan *out);

// profile2 + packed pass + locate.  Pairs the exact kernel has to redo (ties) are left on *list_out
// (count at *count_out, both inside d_work); every other pair has its four outputs written.
// Above 256 rows (p.ra > 256) there is no locate step here: *infoM_out / *infoQ_out (maximum, its block, tie bit)
// are handed to the one-wave-per-pair kernel's locate mode (wave_run) and no output is written yet.
int packed_profiles(const polyhip_scoring *sc, const PackedPlan &p, const uint8_t *d_B, uint32_t lenB, void *d_work,
                    hipStream_t st);
int packed_run(const polyhip_scoring *sc, const PackedPlan &p, const uint8_t *d_A, const uint64_t *d_offA,
               uint64_t npairs, const uint8_t *d_B, uint32_t lenB, const int8_t *prof, const uint32_t *binfo,
               void *d_work, int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err,
               uint32_t **list_out, uint32_t **count_out, hipStream_t st, const uint32_t **infoM_out = nullptr,
               const uint32_t **infoQ_out = nullptr, int defer = 0);
// defer != 0 (p.ra <= 256 only): the locate step does not run its DP.  A pair whose maximum sits in ONE block gets
// score = M, endA = SW_END_DEFERRED, endB = the 1-based last column of that block; the traceback kernel, which sweeps
// those columns anyway, finds the row-major-first cell worth M in its last block (sw_traceback.hip).  Ties still go
// on the list for the exact kernel, errors are reported as usual.
constexpr uint32_t SW_END_DEFERRED = 0xFFFFFFFFu;

} // namespace k3p
} // namespace polyhip

namespace polyhip {
namespace k3 {
// polyhip_sw_batch_dev with the option of leaving end cells to the traceback kernel (sw_batch.hip)
int score_pass(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs, uint32_t max_lenA,
               const uint8_t *d_B, const uint64_t *d_offB, uint64_t lenB, int64_t *d_score, uint32_t *d_endA,
               uint32_t *d_endB, uint32_t *d_err, void *d_work, size_t work_bytes, polyhip_stream_t stream, int defer,
               int *deferred);
} // namespace k3
} // namespace polyhip

// ---- one-wave-per-pair exact score pass (sw_wave.hip): the packed pass's tie list and small batches ----
namespace polyhip {
namespace k3w {

// pairs [0, npairs) if list == nullptr, else the *count pairs on `list`; max_items bounds the grid;
// d_offB == nullptr: one shared B of lenB bytes (binfo = its first invalid byte), else per-pair B
constexpr uint32_t WAVE_MAX_LENA = 4096; // 64 lanes x 64 rows
int wave_run(const polyhip_scoring *sc, const uint8_t *d_A, const uint64_t *d_offA, uint64_t npairs, uint32_t max_lenA,
             const uint8_t *d_B, const uint64_t *d_offB, uint32_t lenB, const uint32_t *binfo, const uint32_t *list,
             const uint32_t *count,
             uint64_t max_items, int64_t *d_score, uint32_t *d_endA, uint32_t *d_endB, uint32_t *d_err, hipStream_t st,
             const uint32_t *infoM = nullptr, const uint32_t *infoQ = nullptr, // locate mode: see sw_wave.hip
             int defer = 0); // ... and leave the end cell of a pair with one block to the traceback kernel (wave8_ok only)
// locate mode on a byte profile of the pair (sw_wave8_kernel): reads of 257..1024 rows, score - gap in a byte, the planes of a
// workgroup's four pairs within 64 KB of LDS (POLYHIP_SW_WAVE8=0: never)
bool wave8_ok(const polyhip_scoring *sc, uint32_t max_lenA);

} // namespace k3w
} // namespace polyhip

// ---- host-pointer flavours: packed A / B batches copied to the device -------------------------------------
#include <algorithm>
#include <vector>

#include "common.h"
#include "multi_device.h"

namespace polyhip {

// Shards of a batch of pairs for the device list (pairs are independent): balanced by sequence bytes -- with
// one shared reference a pair's cells are proportional to its read's length -- plus what comes back per pair.
inline std::vector<uint64_t> split_pairs(const md::Pool &P, const uint64_t *offA, const uint64_t *offB, uint64_t npairs,
                                         uint64_t out_per_pair)
{
    return md::split(npairs, md::size(P), [&](uint64_t i) {
        return offA[i] - offA[0] + (offB ? offB[i] - offB[0] : 0) + i * out_per_pair;
    });
}

// Validates a packed batch of pairs (offsets ascending, buffers present), copies it to the device with offsets
// rebased to 0, and reports the longest A / B.  Shared by polyhip_sw_batch, polyhip_sw_align_batch, polyhip_nw_align_batch.
struct PairStage {
    DevBuf dA, doA, dB, doB;
    std::vector<uint64_t> hoA, hoB; // rebased offsets: they must outlive the asynchronous uploads
    uint64_t maxA = 0, maxB = 0;
    bool per_pair_B = false;
    hipStream_t up = nullptr; // the stream the uploads went to
    // an early return of the entry point must not free the staging vectors and device buffers under uploads in flight
    ~PairStage()
    {
        if (up)
            (void)hipStreamSynchronize(up);
    }

    const uint8_t *A() const { return dA.as<uint8_t>(); }
    const uint64_t *offA() const { return doA.as<uint64_t>(); }
    const uint8_t *B() const { return dB.as<uint8_t>(); }
    const uint64_t *offB() const { return per_pair_B ? doB.as<uint64_t>() : nullptr; }

    // uploads are enqueued on `st` (one of the calling thread's own streams, host_pipeline.h -- never the null stream);
    // the caller synchronises `st` before this object goes away
    int load(const char *who, const uint8_t *A_, const uint64_t *offA_, uint64_t npairs, const uint8_t *B_,
             const uint64_t *offB_, uint64_t lenB, hipStream_t st)
    {
        per_pair_B = offB_ != nullptr;
        up = st;
        maxA = 0;
        maxB = offB_ ? 0 : lenB;
        for (uint64_t i = 0; i < npairs; ++i) {
            PH_REQUIRE(offA_[i] <= offA_[i + 1], "%s: offA not ascending at %llu", who, (unsigned long long)(i + md::base().item));
            maxA = std::max(maxA, offA_[i + 1] - offA_[i]);
            if (offB_) {
                PH_REQUIRE(offB_[i] <= offB_[i + 1], "%s: offB not ascending at %llu", who, (unsigned long long)(i + md::base().item));
                maxB = std::max(maxB, offB_[i + 1] - offB_[i]);
            }
        }
        PH_REQUIRE(maxA < 0xFFFFFFFFull && maxB < 0xFFFFFFFFFFFFull, "%s: sequence too long", who);
        const uint64_t a0 = offA_[0], abytes = offA_[npairs] - a0;
        const uint64_t b0 = offB_ ? offB_[0] : 0, bbytes = offB_ ? offB_[npairs] - b0 : lenB;
        PH_REQUIRE((A_ || abytes == 0) && (B_ || bbytes == 0), "%s: null sequence buffer", who);
        PH_HIP(dA.alloc(abytes + 16));
        PH_HIP(doA.alloc((npairs + 1) * 8));
        PH_HIP(dB.alloc(bbytes + 16));
        hoA.resize(npairs + 1);
        for (uint64_t i = 0; i <= npairs; ++i)
            hoA[i] = offA_[i] - a0;
        PH_HIP(hipMemcpyAsync(doA.p, hoA.data(), (npairs + 1) * 8, hipMemcpyHostToDevice, st));
        if (abytes)
            PH_HIP(hipMemcpyAsync(dA.p, A_ + a0, abytes, hipMemcpyHostToDevice, st));
        if (bbytes)
            PH_HIP(hipMemcpyAsync(dB.p, B_ + b0, bbytes, hipMemcpyHostToDevice, st));
        if (offB_) {
            PH_HIP(doB.alloc((npairs + 1) * 8));
            hoB.resize(npairs + 1);
            for (uint64_t i = 0; i <= npairs; ++i)
                hoB[i] = offB_[i] - b0;
            PH_HIP(hipMemcpyAsync(doB.p, hoB.data(), (npairs + 1) * 8, hipMemcpyHostToDevice, st));
        }
        return POLYHIP_OK;
    }
};

} // namespace polyhip
