// mash_pipeline.hip -- BASELINE configs[2] as ONE host call: reads in, all-vs-all matrix out.
//
//   reference: for every sequence mash.New(k, s).Sketch(seq) (search/mash/mash.go:59-104), then for every ordered pair
//   X_i.Similarity(X_j) / .Distance(X_j) (mash.go:107-140) -- two nested loops over one slice of *Mash in the caller.
//
// The sketches never visit the host between the two steps unless the caller asks for them.  On a device list
// (multi_device.h) this is SURVEY 8e's configs[2] flow inside one process: the reads shard by bytes, every device
// sketches its shard into its own copy of the full sketch array; then (round 4) the devices build ONE index together
// without gathering the sketches -- level 1 on a device's own rows, index items pulled by value range
// (hipMemcpyPeerAsync over xGMI: no RCCL, no one-rank-per-device rule, a device may appear twice in the list), level 2
// on 1/N of the range, finished parts pulled (k2_exchange_index, mash_distance.hip) -- and every device joins the rows it
// sketched, which go straight to the caller's matrix.  Where the merge needs raw sketches (an irregular sketch), or
// with POLYHIP_K2_EXCHANGE=0, the devices PULL the other shards' rows instead (the in-process all-gather) and each
// builds the whole index.  The matrix stays sharded by rows all the way: no second collective.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"
#include "host_pipeline.h"
#include "multi_device.h"

using namespace polyhip;

namespace {

thread_local int g_last_path = 0;

// one device's part: its copy of the sketch array, the reads it sketched
struct SketchShard {
    uint64_t i0 = 0, i1 = 0; // reads [i0, i1)
    int dev = -1;
    DevBuf dSk;              // n * s hashes: rows [i0, i1) after round A, everything after the pull
    int panic = POLYHIP_OK;  // SketchSize < 2: the first panicking sequence is named, the rest is still sketched
    std::string panic_text;
};

// Round A: sketch reads [i0, i1) into rows [i0, i1) of this device's sketch array, chunk by chunk through two slots
// (the same pipeline as polyhip_mash_sketch_batch, minus the download unless the caller wants the sketches).
int sketch_shard(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s, uint32_t *sketches,
                 SketchShard &sh)
{
    PH_HIP(hipGetDevice(&sh.dev));
    PH_HIP(sh.dSk.alloc(n * (size_t)s * 4));
    const uint64_t m_all = sh.i1 - sh.i0;
    if (m_all == 0)
        return POLYHIP_OK;
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    const uint64_t row = (uint64_t)s * 4;
    uint32_t *d_rows = sh.dSk.as<uint32_t>() + sh.i0 * (uint64_t)s;
    // a row of a read with fewer than s windows keeps (part of) its prior state (mash.go:81-84): the caller's rows if it
    // passed any, zeros (= mash.New) otherwise
    bool need_prior = false;
    const uint64_t *off = offsets + sh.i0;
    for (uint64_t i = 0; i < m_all; ++i) {
        PH_REQUIRE(off[i] <= off[i + 1], "polyhip_mash_sketch_distance_matrix: offsets not ascending at %llu",
                   (unsigned long long)(i + sh.i0));
        need_prior |= off[i + 1] - off[i] < (uint64_t)k + s;
    }
    if (need_prior) {
        if (sketches)
            PH_HIP(hipMemcpyAsync(d_rows, sketches + sh.i0 * (uint64_t)s, m_all * row, hipMemcpyHostToDevice, hs.s[0]));
        else
            PH_HIP(hipMemsetAsync(d_rows, 0, m_all * row, hs.s[0]));
        PH_HIP(hipStreamSynchronize(hs.s[0])); // both slots' streams write these rows next
    }
    const Chunks ch = cut_packed(off, m_all, row, 256ull << 20);
    PackedSlot slot[2];
    for (size_t q = 0; q < std::min<size_t>(2, ch.count()); ++q)
        PH_HIP(slot[q].alloc(ch, hs.s[q]));
    for (size_t c = 0; c < ch.count(); ++c) {
        PackedSlot &S = slot[c & 1];
        const uint64_t j0 = ch.cut[c], m = ch.cut[c + 1] - j0;
        PH_HIP(hipStreamSynchronize(S.st)); // chunk c-2 has left this slot (its staged offsets included)
        PH_HIP(S.upload(seqs, off, j0, m));
        int rc;
        {
            md::BaseScope pos(sh.i0 + j0, 0);
            rc = polyhip_mash_sketch_batch_dev(S.dseq.as<uint8_t>(), S.doff.as<uint64_t>(), m, k, s, d_rows + j0 * (uint64_t)s, S.st);
        }
        if (rc == POLYHIP_ERR_PANIC) {
            if (sh.panic == POLYHIP_OK) {
                sh.panic = rc;
                sh.panic_text = polyhip_last_error();
            }
        } else if (rc != POLYHIP_OK) {
            (void)hs.sync_both();
            return rc;
        }
        if (sketches)
            PH_HIP(hipMemcpyAsync(sketches + (sh.i0 + j0) * (uint64_t)s, d_rows + j0 * (uint64_t)s, m * row, hipMemcpyDeviceToHost, S.st));
    }
    PH_HIP(hs.sync_both());
    return POLYHIP_OK;
}

// Round B: pull the other shards' rows from their devices, then this device's block of matrix rows
int join_shard(std::vector<SketchShard> &all, size_t me, uint64_t n, uint32_t s, uint64_t r0, uint64_t r1, uint16_t *counts,
               double *dist)
{
    SketchShard &sh = all[me];
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    for (size_t q = 0; q < all.size(); ++q) {
        const SketchShard &o = all[q];
        if (q == me || o.i1 == o.i0)
            continue;
        if (o.dev != sh.dev) {
            int can = 0;
            PH_HIP(hipDeviceCanAccessPeer(&can, sh.dev, o.dev));
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess(o.dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    PH_HIP(e);
                (void)hipGetLastError(); // "already enabled" is sticky otherwise
            } // without peer access the runtime stages the copy through the host: slower, still correct
        }
        const uint64_t a = o.i0 * (uint64_t)s;
        PH_HIP(hipMemcpyPeerAsync(sh.dSk.as<uint32_t>() + a, sh.dev, o.dSk.as<uint32_t>() + a, o.dev, (o.i1 - o.i0) * (uint64_t)s * 4,
                                  hs.s[0]));
    }
    if (r0 == r1 || !(counts || dist)) {
        PH_HIP(hipStreamSynchronize(hs.s[0]));
        return POLYHIP_OK;
    }
    // ordered behind the pulls on the thread's first stream
    return k2_rows_to_host(sh.dSk.as<uint32_t>() + r0 * (uint64_t)s, r1 - r0, s, sh.dSk.as<uint32_t>(), n, s,
                           counts ? counts + r0 * n : nullptr, dist ? dist + r0 * n : nullptr);
}

} // namespace

extern "C" {

int polyhip_mash_sketch_distance_matrix(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s,
                                        uint32_t *sketches, uint16_t *counts, double *dist)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets, "polyhip_mash_sketch_distance_matrix: null pointer");
    PH_REQUIRE(s <= 65535, "polyhip_mash_sketch_distance_matrix: SketchSize %u > 65535 (the matrix holds 16-bit counts)", s);
    if (s == 0 && (counts || dist)) // Similarity indexes Sketches[-1] (mash.go:117) whatever Sketch did before
        return polyhip_mash_shared_counts_dev(nullptr, n, 0, nullptr, n, 0, nullptr, 0, nullptr, 0, nullptr);
    std::shared_ptr<md::Pool> P = md::pool();
    const size_t nsh = P ? md::size(*P) : 1;
    g_last_path = P ? 2 : 0;
    const uint64_t row = (uint64_t)s * 4;
    const std::vector<uint64_t> cut = md::split(n, nsh, [&](uint64_t i) { return offsets[i] - offsets[0] + i * row; });
    std::vector<SketchShard> sh(nsh);
    for (size_t q = 0; q < nsh; ++q) {
        sh[q].i0 = cut[q];
        sh[q].i1 = cut[q + 1];
    }
    auto round_a = [&](size_t q) { return sketch_shard(seqs, offsets, n, k, s, sketches, sh[q]); };
    auto round_b = [&](size_t q) {
        // matrix rows are cut evenly, whatever the reads' sizes were: every device holds every sketch by now
        const uint64_t r0 = (uint64_t)(((unsigned __int128)n * q) / nsh), r1 = (uint64_t)(((unsigned __int128)n * (q + 1)) / nsh);
        const int rc = join_shard(sh, q, n, s, r0, r1, counts, dist);
        return rc;
    };
    int rc = P ? md::run(*P, round_a) : round_a(0);
    if (rc != POLYHIP_OK)
        return rc;
    for (size_t q = 0; q < nsh; ++q) // SketchSize 1: the reference panics in Sketch, before any distance is asked for
        if (sh[q].panic != POLYHIP_OK)
            return set_error(sh[q].panic, "%s", sh[q].panic_text.c_str());
    // Round B without gathering the sketches (round 4; POLYHIP_K2_EXCHANGE=0: the gather, testing aid and the way back):
    // the devices build ONE index together -- level 1 on their own rows, items exchanged by value range, level 2 on 1/N of the
    // range, finished parts exchanged (k2_exchange_index, mash_distance.hip) -- and every device joins the rows it sketched.
    if (P && nsh >= 2 && (counts || dist) && !env_is("POLYHIP_K2_EXCHANGE", '0')) {
        std::vector<K2XShard> xs(nsh);
        uint64_t rows_blk = 1;
        for (size_t q = 0; q < nsh; ++q) {
            xs[q].dev = sh[q].dev;
            xs[q].i0 = sh[q].i0;
            xs[q].i1 = sh[q].i1;
            xs[q].sk = sh[q].dSk.as<uint32_t>();
            rows_blk = std::max(rows_blk, k2_rows_per_block(sh[q].i1 - sh[q].i0, n, counts != nullptr, dist != nullptr));
        }
        bool built = false;
        rc = k2_exchange_index(*P, xs, n, s, rows_blk, &built);
        if (rc != POLYHIP_OK)
            return rc;
        if (built)
            g_last_path = 1;
        if (built)
            return md::run(*P, [&](size_t q) {
                const uint64_t i0 = sh[q].i0, m = sh[q].i1 - i0;
                if (m == 0)
                    return (int)POLYHIP_OK;
                const uint32_t *sk = sh[q].dSk.as<uint32_t>();
                return k2_rows_to_host(sk + i0 * (uint64_t)s, m, s, sk, n, s, counts ? counts + i0 * n : nullptr,
                                       dist ? dist + i0 * n : nullptr, xs[q].work.p, xs[q].work_bytes);
            });
        // (an irregular sketch, a join that is not the dense one, or an input the merge would take: gather after all)
    }
    rc = P ? md::run(*P, round_b) : round_b(0);
    // the sketch arrays are freed by whoever drops `sh` (hipFree takes a pointer of any device)
    return rc;
}

int polyhip_mash_sketch_distance_matrix_last_path(void) { return g_last_path; }

} // extern "C"
