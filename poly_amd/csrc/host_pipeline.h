// host_pipeline.h -- what the host-pointer (cgo) entry points share: two streams per calling thread and a packed batch
// cut into chunks that alternate between two device slots, so that the upload and the kernels of chunk c overlap the
// download of chunk c-1.  Nothing here touches the null stream: concurrent goroutines (cgo calls arrive on arbitrary OS
// threads) do not serialise on it.
#pragma once
#include <atomic>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "multi_device.h"

namespace polyhip {

// two non-blocking streams + one event per calling thread and device, made on first use and kept (a stream costs tens
// of microseconds to create: too much for a single-sequence call)
struct HostStreams {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t ev = nullptr;
    int dev = -1;
    hipError_t init()
    {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess)
            return e;
        if (s[0] && s[1] && ev && dev == cur)
            return hipSuccess;
        for (int q = 0; q < 2; ++q)
            if (s[q]) {
                (void)hipStreamDestroy(s[q]);
                s[q] = nullptr;
            }
        if (ev) {
            (void)hipEventDestroy(ev);
            ev = nullptr;
        }
        dev = -1;
        for (int q = 0; q < 2 && e == hipSuccess; ++q)
            e = hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess)
            dev = cur;
        return e;
    }
    // a fan-out worker that ends (its device list was replaced) gives its streams back; the calling threads of a host
    // program keep theirs for the life of the process
    void destroy()
    {
        for (int q = 0; q < 2; ++q)
            if (s[q]) {
                (void)hipStreamSynchronize(s[q]);
                (void)hipStreamDestroy(s[q]);
                s[q] = nullptr;
            }
        if (ev) {
            (void)hipEventDestroy(ev);
            ev = nullptr;
        }
        dev = -1;
    }
    hipError_t sync_both()
    {
        hipError_t e = hipStreamSynchronize(s[0]);
        const hipError_t e1 = hipStreamSynchronize(s[1]);
        return e != hipSuccess ? e : e1;
    }
};
inline HostStreams &host_streams()
{
    static thread_local HostStreams h;
    return h;
}

// A packed batch cut into chunks of about `target` bytes (sequence bytes + out_per_item per sequence); items cut[c] ..
// cut[c + 1] form chunk c.
struct Chunks {
    std::vector<uint64_t> cut{0};
    uint64_t max_bytes = 0, max_items = 0;
    size_t count() const { return cut.size() - 1; }
};
inline Chunks cut_packed(const uint64_t *offsets, uint64_t n, uint64_t out_per_item, uint64_t target)
{
    Chunks c;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i, sz = 0;
        do {
            sz += offsets[j + 1] - offsets[j] + out_per_item;
            ++j;
        } while (j < n && sz < target);
        c.cut.push_back(j);
        c.max_bytes = std::max(c.max_bytes, offsets[j] - offsets[i]);
        c.max_items = std::max(c.max_items, j - i);
        i = j;
    }
    return c;
}

// the input side of one slot: sequence bytes + rebased offsets of the chunk it currently holds
struct PackedSlot {
    DevBuf dseq, doff;
    std::vector<uint64_t> hoff;
    hipStream_t st = nullptr;
    hipError_t alloc(const Chunks &c, hipStream_t stream)
    {
        st = stream;
        hoff.resize(c.max_items + 1);
        hipError_t e = dseq.alloc(c.max_bytes + 16);
        if (e == hipSuccess)
            e = doff.alloc((c.max_items + 1) * sizeof(uint64_t));
        return e;
    }
    // chunk [i0, i0 + m): offsets rebased so that the device copy starts at byte 0.  The caller has synchronised `st`
    // since this slot's previous chunk (hoff is reused).
    hipError_t upload(const uint8_t *seqs, const uint64_t *offsets, uint64_t i0, uint64_t m)
    {
        const uint64_t b0 = offsets[i0];
        for (uint64_t i = 0; i <= m; ++i)
            hoff[i] = offsets[i0 + i] - b0;
        hipError_t e = hipMemcpyAsync(doff.p, hoff.data(), (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && offsets[i0 + m] > b0)
            e = hipMemcpyAsync(dseq.p, seqs + b0, offsets[i0 + m] - b0, hipMemcpyHostToDevice, st);
        return e;
    }
};

// A helper thread per host call that runs the call's DOWNLOADS.  A copy between the device and pageable host memory holds
// the calling thread until the bytes are across, so one thread alone takes uploads and downloads in turns and uses half of
// the duplex link; with the downloads of finished chunks on this thread, chunk c + 1 goes up while chunk c comes down
// (a Go slice is pageable memory: this is the case that matters for cgo callers).  Jobs run in the order they were pushed.
class Downloader {
  public:
    explicit Downloader(int device) : dev_(device), th_([this] { loop(); }) {}
    ~Downloader()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void push(std::function<hipError_t()> job)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(job));
        }
        cv_.notify_all();
    }
    // until `count` jobs have finished; the first error any job returned
    hipError_t wait(size_t count)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return done_ >= count; });
        return err_;
    }

  private:
    void loop()
    {
        (void)hipSetDevice(dev_);
        for (;;) {
            std::function<hipError_t()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || next_ < q_.size(); });
                if (next_ >= q_.size())
                    return;
                job = std::move(q_[next_++]);
            }
            const hipError_t e = job();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (e != hipSuccess && err_ == hipSuccess)
                    err_ = e;
                ++done_;
            }
            cv_.notify_all();
        }
    }
    int dev_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::function<hipError_t()>> q_;
    size_t next_ = 0, done_ = 0;
    hipError_t err_ = hipSuccess;
    bool stop_ = false;
    std::thread th_; // last: the thread starts when everything above exists
};

// The two-slot pipelines' download side: chunk c's results travel back on a helper thread and a stream of their own while
// the calling thread uploads chunk c + 1 (both directions of the link at once).  With a single chunk there is nothing to
// overlap and no thread is started: the copies run on the slot's own stream.
//     Duplex dx;  dx.init(nchunks);
//     for c:  dx.slot_free(c);  upload + launch on st;  dx.download(c, st, [=](hipStream_t s) { return hipMemcpyAsync(.., s); });
//     dx.finish();
class Duplex {
  public:
    ~Duplex()
    {
        dl_.reset(); // drains the queue and joins
        for (hipEvent_t e : ev_)
            if (e)
                (void)hipEventDestroy(e);
        if (ds_) {
            (void)hipStreamSynchronize(ds_);
            (void)hipStreamDestroy(ds_);
        }
    }
    hipError_t init(size_t nchunks)
    {
        if (nchunks < 2)
            return hipSuccess;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        for (int q = 0; q < 2 && e == hipSuccess; ++q)
            e = hipEventCreateWithFlags(&ev_[q], hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipStreamCreateWithFlags(&ds_, hipStreamNonBlocking);
        if (e == hipSuccess)
            dl_.reset(new Downloader(dev));
        return e;
    }
    // the slot of chunk c is free again: chunk c - 2's download (and so its kernels and uploads) is through
    hipError_t slot_free(size_t c, hipStream_t slot_stream)
    {
        if (!dl_)
            return hipStreamSynchronize(slot_stream);
        return c >= 2 ? dl_->wait(c - 1) : hipSuccess;
    }
    // everything enqueued on `slot_stream` so far produces chunk c's results; `copies(stream)` enqueues their downloads
    hipError_t download(size_t c, hipStream_t slot_stream, std::function<hipError_t(hipStream_t)> copies)
    {
        if (!dl_)
            return copies(slot_stream);
        hipEvent_t ev = ev_[c & 1];
        const hipError_t e = hipEventRecord(ev, slot_stream);
        if (e != hipSuccess)
            return e;
        hipStream_t ds = ds_;
        dl_->push([=]() -> hipError_t {
            hipError_t x = hipStreamWaitEvent(ds, ev, 0);
            if (x == hipSuccess)
                x = copies(ds);
            if (x == hipSuccess)
                x = hipStreamSynchronize(ds);
            return x;
        });
        ++pushed_;
        return hipSuccess;
    }
    hipError_t finish() { return dl_ ? dl_->wait(pushed_) : hipSuccess; }

  private:
    std::unique_ptr<Downloader> dl_;
    hipEvent_t ev_[2] = {nullptr, nullptr};
    hipStream_t ds_ = nullptr;
    size_t pushed_ = 0;
};

// chunk size of the two-slot host pipelines; POLYHIP_HOST_CHUNK_MB=<n> overrides it (testing / tuning aid)
inline uint64_t host_chunk_bytes()
{
    if (const char *e = getenv("POLYHIP_HOST_CHUNK_MB")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 1 && v <= 65536)
            return v << 20;
    }
    return 64ull << 20;
}
#define HOST_CHUNK_BYTES (::polyhip::host_chunk_bytes())

// mash_distance.hip: row blocks of the shared-count / distance matrix from device-resident sketches to host buffers
// (index_work: a workspace that already holds the index of Y -- sized with k2_rows_per_block's rows on its X side --, or
// null: the call builds the index in a workspace of its own)
int k2_rows_to_host(const uint32_t *dX, uint64_t nx, uint32_t sx, const uint32_t *dY, uint64_t ny, uint32_t sy,
                    uint16_t *counts, double *dist, void *index_work = nullptr, size_t index_work_bytes = 0);
uint64_t k2_rows_per_block(uint64_t nx, uint64_t ny, bool counts, bool dist);

// mash_distance.hip: the index of a sketch set that is SPREAD over the devices of a list, built without gathering the
// sketches (every device runs level 1 on its own rows, the 8-byte items travel by value range,
// level 2 runs on 1/N of the range, the finished parts are exchanged).  Shard q lives on worker q of the pool: `sk` is
// that device's n x s array of which rows [i0, i1) are valid.  On return *built says whether every device's `work` holds
// the whole index (false: the set has an irregular sketch, the geometry is not the dense join's, or the input is so
// dense that the merge would take it -- the caller gathers the sketches instead).
// device-to-device copies of one polyhip_mash_sketch_distance_matrix call, by transport (polyhip_matrix_info); the workers
// of a device list count into one of these
struct K2XferStats {
    std::atomic<int> peer{0}, staged{0}, local{0};
    std::atomic<uint64_t> bytes_peer{0}, bytes_staged{0}, bytes_local{0};
    // 0 = local (one device), 1 = peer access on, 2 = no peer access: staged through the host by the runtime
    void count(int transport, uint64_t bytes)
    {
        (transport == 0 ? local : transport == 1 ? peer : staged).fetch_add(1, std::memory_order_relaxed);
        (transport == 0 ? bytes_local : transport == 1 ? bytes_peer : bytes_staged).fetch_add(bytes, std::memory_order_relaxed);
    }
};
// makes `other`'s memory reachable from device `me` (the calling thread's current device) if the hardware allows it;
// *transport as K2XferStats::count's
int k2_enable_peer(int me, int other, int *transport);

struct K2XShard {
    int dev = -1;
    K2XferStats *stats = nullptr;
    uint64_t i0 = 0, i1 = 0;
    const uint32_t *sk = nullptr;
    DevBuf work;
    size_t work_bytes = 0;
    // (scratch of the exchange)
    std::vector<uint32_t> h_gcount, h_cstart;
    uint32_t h_maxval = 0, h_maxmult = 0, h_fmt = 0;
    uint64_t h_nirr = 0, h_est = 0;
    DevBuf segtab, segptr;
};
// raw_type >= 0 (seqhash.Hash; only when k5_wave_takes_all(max_len)): d_seqs are the caller's bytes; the wave kernel normalises
// them while it stages them, writes the normalised copy to d_norm_out (indexed like d_seqs) and raises *d_any_bad on a letter
// outside the alphabet of sequence type raw_type
int k5_least_rotation_strands_dev(const uint8_t *d_seqs, const uint64_t *d_offsets, uint64_t n, uint64_t max_len,
                                  uint64_t *d_rot_index, uint8_t *d_rotated, uint64_t *d_rot_rc, polyhip_stream_t stream,
                                  int raw_type = -1, uint8_t *d_norm_out = nullptr, uint32_t *d_any_bad = nullptr);
bool k5_wave_takes_all(uint64_t max_len); // every sequence of the batch gets a wave of least_rotation_wave_kernel
int k2_exchange_index(md::Pool &P, std::vector<K2XShard> &sh, uint64_t n, uint32_t s, uint64_t rows_blk, bool *built);

} // namespace polyhip
