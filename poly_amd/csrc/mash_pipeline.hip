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
#include <chrono>
#include <string>
#include <vector>

#include "common.h"
#include "host_pipeline.h"
#include "multi_device.h"

using namespace polyhip;

namespace {

thread_local int g_last_path = 0;
thread_local polyhip_matrix_info g_last_info = {};

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// one device's part: its copy of the sketch array, the reads it sketched
struct SketchShard {
    uint64_t i0 = 0, i1 = 0; // reads [i0, i1)
    int dev = -1;
    DevBuf dSk;              // the device's sketch rows: ITS OWN rows [i0, i1) on a device list -- the item exchange never needs
                             // another device's rows, and the gather re-allocates (widen) --, all n on one device
    uint64_t row0 = 0;       // dSk's first row (sk() is the array's virtual base: row r lives at sk() + r * s)
    uint32_t *sk(uint32_t s) const { return dSk.as<uint32_t>() - row0 * (uint64_t)s; }
    int panic = POLYHIP_OK;  // SketchSize < 2: the first panicking sequence is named, the rest is still sketched
    std::string panic_text;
};

// Round A: sketch reads [i0, i1) into rows [i0, i1) of this device's sketch array, chunk by chunk through two slots
// (the same pipeline as polyhip_mash_sketch_batch, minus the download unless the caller wants the sketches).
int sketch_shard(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s, uint32_t *sketches,
                 SketchShard &sh)
{
    PH_HIP(hipGetDevice(&sh.dev));
    const uint64_t m_all = sh.i1 - sh.i0;
    // own rows only (round-4 verdict: every device used to allocate all n rows even where it never reads the others')
    sh.row0 = sh.i0;
    PH_HIP(sh.dSk.alloc(std::max<uint64_t>(m_all, 1) * (size_t)s * 4));
    (void)n;
    if (m_all == 0)
        return POLYHIP_OK;
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    const uint64_t row = (uint64_t)s * 4;
    uint32_t *d_rows = sh.dSk.as<uint32_t>(); // = sk(s) + i0 * s
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
// (two md::run rounds: every device first widens its array to all n rows, keeping its own -- only then may anybody pull)
int widen_shard(SketchShard &sh, uint64_t n, uint32_t s)
{
    if (sh.row0 == 0 && sh.i0 == 0 && sh.i1 == n)
        return POLYHIP_OK; // one shard holds everything already
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    DevBuf full;
    PH_HIP(full.alloc(n * (size_t)s * 4));
    if (sh.i1 > sh.i0)
        PH_HIP(hipMemcpyAsync(full.as<uint32_t>() + sh.i0 * (uint64_t)s, sh.dSk.p, (sh.i1 - sh.i0) * (uint64_t)s * 4,
                              hipMemcpyDeviceToDevice, hs.s[0]));
    PH_HIP(hipStreamSynchronize(hs.s[0]));
    std::swap(sh.dSk.p, full.p); // (`full` now frees the own-rows array)
    sh.row0 = 0;
    return POLYHIP_OK;
}

int join_shard(std::vector<SketchShard> &all, size_t me, uint64_t n, uint32_t s, uint64_t r0, uint64_t r1, uint16_t *counts,
               double *dist, K2XferStats *stats)
{
    SketchShard &sh = all[me];
    HostStreams &hs = host_streams();
    PH_HIP(hs.init());
    for (size_t q = 0; q < all.size(); ++q) {
        const SketchShard &o = all[q];
        if (q == me || o.i1 == o.i0)
            continue;
        int tr = 0;
        if (int e = k2_enable_peer(sh.dev, o.dev, &tr))
            return e;
        const uint64_t a = o.i0 * (uint64_t)s, bytes = (o.i1 - o.i0) * (uint64_t)s * 4;
        PH_HIP(hipMemcpyPeerAsync(sh.sk(s) + a, sh.dev, o.sk(s) + a, o.dev, bytes, hs.s[0]));
        if (stats)
            stats->count(tr, bytes);
    }
    if (r0 == r1 || !(counts || dist)) {
        PH_HIP(hipStreamSynchronize(hs.s[0]));
        return POLYHIP_OK;
    }
    // ordered behind the pulls on the thread's first stream
    return k2_rows_to_host(sh.sk(s) + r0 * (uint64_t)s, r1 - r0, s, sh.sk(s), n, s, counts ? counts + r0 * n : nullptr,
                           dist ? dist + r0 * n : nullptr);
}

} // namespace

extern "C" {

int polyhip_mash_sketch_distance_matrix(const uint8_t *seqs, const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t s,
                                        uint32_t *sketches, uint16_t *counts, double *dist)
{
    if (n == 0)
        return POLYHIP_OK;
    PH_REQUIRE(seqs && offsets, "polyhip_mash_sketch_distance_matrix: null pointer");
    // the matrix's limits apply where a matrix is asked for: with counts == dist == NULL the call only sketches, as
    // polyhip_mash_sketch_batch does for any SketchSize (round-4 advice)
    PH_REQUIRE(!(counts || dist) || s <= 65535,
               "polyhip_mash_sketch_distance_matrix: SketchSize %u > 65535 (the matrix holds 16-bit counts)", s);
    PH_REQUIRE(!(counts || dist) || n < (1ull << 31), "polyhip_mash_sketch_distance_matrix: %llu sketches, the matrix takes fewer than 2^31",
               (unsigned long long)n);
    g_last_info = polyhip_matrix_info{};
    K2XferStats stats;
    auto publish = [&](int rc) { // the transports and the path of this call, whatever way it ends
        g_last_info.path = g_last_path;
        g_last_info.peer_copies = stats.peer.load();
        g_last_info.staged_copies = stats.staged.load();
        g_last_info.local_copies = stats.local.load();
        g_last_info.bytes_peer = stats.bytes_peer.load();
        g_last_info.bytes_staged = stats.bytes_staged.load();
        g_last_info.bytes_local = stats.bytes_local.load();
        return rc;
    };
    if (s == 0 && (counts || dist)) // Similarity indexes Sketches[-1] (mash.go:117) whatever Sketch did before
        return polyhip_mash_shared_counts_dev(nullptr, n, 0, nullptr, n, 0, nullptr, 0, nullptr, 0, nullptr);
    std::shared_ptr<md::Pool> P = md::pool();
    const size_t nsh = P ? md::size(*P) : 1;
    g_last_path = P ? 2 : 0;
    g_last_info.devices = (int32_t)nsh;
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
        const int rc = join_shard(sh, q, n, s, r0, r1, counts, dist, &stats);
        return rc;
    };
    auto t0 = std::chrono::steady_clock::now();
    int rc = P ? md::run(*P, round_a) : round_a(0);
    g_last_info.ms_sketch = ms_since(t0);
    if (rc != POLYHIP_OK)
        return publish(rc);
    for (size_t q = 0; q < nsh; ++q) // SketchSize 1: the reference panics in Sketch, before any distance is asked for
        if (sh[q].panic != POLYHIP_OK)
            return publish(set_error(sh[q].panic, "%s", sh[q].panic_text.c_str()));
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
            xs[q].sk = sh[q].sk(s); // (virtual base: the exchange reads rows [i0, i1) only)
            xs[q].stats = &stats;
            rows_blk = std::max(rows_blk, k2_rows_per_block(sh[q].i1 - sh[q].i0, n, counts != nullptr, dist != nullptr));
        }
        bool built = false;
        t0 = std::chrono::steady_clock::now();
        rc = k2_exchange_index(*P, xs, n, s, rows_blk, &built);
        g_last_info.ms_index = ms_since(t0);
        if (rc != POLYHIP_OK)
            return publish(rc);
        if (built) {
            g_last_path = 1;
            t0 = std::chrono::steady_clock::now();
            rc = md::run(*P, [&](size_t q) {
                const uint64_t i0 = sh[q].i0, m = sh[q].i1 - i0;
                if (m == 0)
                    return (int)POLYHIP_OK;
                // Y = NULL: a device holds ITS rows only, and the join reads Y's raw sketches nowhere but in the merge
                // (irregular sketches, overflow rows, "merge everything"), which k2_exchange_index ruled out before it reported
                // the index as built and which returns at once without Y.  That exclusion is re-checked on what the join DID:
                const uint32_t *sk = sh[q].sk(s);
                if (int e = k2_rows_to_host(sk + i0 * (uint64_t)s, m, s, nullptr, n, s, counts ? counts + i0 * n : nullptr,
                                            dist ? dist + i0 * n : nullptr, xs[q].work.p, xs[q].work_bytes))
                    return e;
                uint32_t mode = 0, nirrx = 0, nirry = 0, novf = 0;
                if (int e = polyhip_mash_shared_counts_mode_dev(xs[q].work.p, &mode, &nirrx, &nirry, &novf, nullptr))
                    return e;
                if (mode != 0 || nirrx || nirry || novf)
                    return set_error(POLYHIP_ERR_HIP, "polyhip_mash_sketch_distance_matrix: the join behind the item exchange took the merge "
                                     "(mode %u, irregular %u / %u, overflow rows %u), which reads sketches this device does not hold -- "
                                     "a bug in k2_exchange_index's conditions; POLYHIP_K2_EXCHANGE=0 gathers the sketches instead",
                                     mode, nirrx, nirry, novf);
                return (int)POLYHIP_OK;
            });
            g_last_info.ms_join = ms_since(t0);
            return publish(rc);
        }
        // (an irregular sketch, a join that is not the dense one, or an input the merge would take: gather after all)
    }
    t0 = std::chrono::steady_clock::now();
    if (P) { // every device widens its array to all n rows before anybody pulls
        rc = md::run(*P, [&](size_t q) { return widen_shard(sh[q], n, s); });
        if (rc != POLYHIP_OK)
            return publish(rc);
    }
    rc = P ? md::run(*P, round_b) : round_b(0);
    g_last_info.ms_join = ms_since(t0);
    // the sketch arrays are freed by whoever drops `sh` (hipFree takes a pointer of any device)
    return publish(rc);
}

int polyhip_mash_sketch_distance_matrix_last_path(void) { return g_last_path; }

int polyhip_mash_sketch_distance_matrix_last_info(polyhip_matrix_info *info)
{
    PH_REQUIRE(info, "polyhip_mash_sketch_distance_matrix_last_info: null pointer");
    *info = g_last_info;
    return POLYHIP_OK;
}

} // extern "C"
