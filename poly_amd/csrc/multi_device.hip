// multi_device.hip -- the process-wide device list and its worker threads (see multi_device.h).
#include "multi_device.h"
#include "host_pipeline.h"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

namespace polyhip {
namespace md {

namespace {

thread_local Base g_base;
thread_local bool g_in_worker = false;

// one call's shards: the caller waits on `cv` until `pending` reaches zero
struct Job {
    const std::function<int(size_t)> *fn = nullptr;
    std::mutex m;
    std::condition_variable cv;
    size_t pending = 0;
    std::vector<int> status;
    std::vector<std::string> msg;
};

struct Worker {
    int dev = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::pair<Job *, size_t>> q;
    bool stop = false;

    void loop()
    {
        g_in_worker = true;
        for (;;) {
            std::pair<Job *, size_t> t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) { // stop, and nothing left to do
                    host_streams().destroy();
                    return;
                }
                t = q.front();
                q.pop_front();
            }
            Job &j = *t.first;
            clear_error();
            g_base = Base();
            int rc;
            // the device is per-thread state: set on every task (cheap) rather than trusted from the last one
            const hipError_t e = hipSetDevice(dev);
            if (e != hipSuccess)
                rc = set_error(POLYHIP_ERR_HIP, "hipSetDevice(%d): %s", dev, hipGetErrorString(e));
            else
                rc = (*j.fn)(t.second);
            std::string text = rc != POLYHIP_OK ? polyhip_last_error() : "";
            {
                std::lock_guard<std::mutex> lk(j.m);
                j.status[t.second] = rc;
                j.msg[t.second] = std::move(text);
                if (--j.pending == 0)
                    j.cv.notify_all();
            }
        }
    }
};

} // namespace

struct Pool {
    std::vector<std::unique_ptr<Worker>> w;
    std::mutex submit; // a job's shards enter every queue in one step: all workers see the jobs in the same order
    explicit Pool(const std::vector<int> &ids)
    {
        for (int id : ids) {
            w.emplace_back(new Worker());
            w.back()->dev = id;
        }
        for (auto &x : w)
            x->th = std::thread([p = x.get()] { p->loop(); });
    }
    ~Pool()
    {
        for (auto &x : w) {
            {
                std::lock_guard<std::mutex> lk(x->m);
                x->stop = true;
            }
            x->cv.notify_all();
        }
        for (auto &x : w)
            if (x->th.joinable())
                x->th.join();
    }
};

namespace {

// never destroyed: at process exit the workers simply end with the process (joining them from a static destructor
// would race the HIP runtime's own teardown)
struct Global {
    std::mutex m;
    std::shared_ptr<Pool> cur;
    std::vector<int> ids;
    bool configured = false; // set_devices was called, or the environment has been read
};
Global &global()
{
    static Global *g = new Global();
    return *g;
}

// "0,1,2" / "0,0,0" / "all"; anything else is reported, not guessed at
int parse_list(const char *text, int ndev, std::vector<int> *out)
{
    out->clear();
    if (!strcmp(text, "all")) {
        for (int i = 0; i < ndev; ++i)
            out->push_back(i);
        return POLYHIP_OK;
    }
    const char *p = text;
    while (*p) {
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v >= ndev)
            return set_error(POLYHIP_ERR_INVALID, "POLYHIP_DEVICES=\"%s\": expected device ids below %d separated by commas", text,
                             ndev);
        out->push_back((int)v);
        p = end;
        if (*p == ',')
            ++p;
        else if (*p)
            return set_error(POLYHIP_ERR_INVALID, "POLYHIP_DEVICES=\"%s\": expected device ids below %d separated by commas", text,
                             ndev);
    }
    return POLYHIP_OK;
}

int install(Global &g, const std::vector<int> &ids)
{
    // the old pool ends when its last running call lets go of it (the calls hold a shared_ptr)
    g.cur = ids.empty() ? nullptr : std::make_shared<Pool>(ids);
    g.ids = ids;
    g.configured = true;
    return POLYHIP_OK;
}

} // namespace

Base &base() { return g_base; }

std::shared_ptr<Pool> pool()
{
    if (g_in_worker)
        return nullptr;
    Global &g = global();
    std::lock_guard<std::mutex> lk(g.m);
    if (!g.configured) {
        g.configured = true;
        if (const char *e = getenv("POLYHIP_DEVICES"))
            if (*e) {
                int ndev = 0;
                std::vector<int> ids;
                // a bad list is not fatal here (this is the middle of some compute call): it is ignored, with the reason in
                // polyhip_last_error() should that call fail for want of a device; polyhip_set_devices reports it properly
                if (hipGetDeviceCount(&ndev) == hipSuccess && parse_list(e, ndev, &ids) == POLYHIP_OK)
                    install(g, ids);
            }
    }
    return g.cur;
}

size_t size(const Pool &p) { return p.w.size(); }
int device(const Pool &p, size_t worker) { return p.w[worker]->dev; }

std::vector<uint64_t> split(uint64_t n, size_t nshards, const std::function<uint64_t(uint64_t)> &prefix)
{
    std::vector<uint64_t> cut(nshards + 1, n);
    cut[0] = 0;
    const uint64_t total = n ? prefix(n) : 0;
    for (size_t q = 1; q < nshards; ++q) {
        // the smallest i whose prefix reaches q / nshards of the total (128-bit product: byte counts times shard counts)
        const uint64_t want = (uint64_t)(((unsigned __int128)total * q) / nshards);
        uint64_t lo = cut[q - 1], hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (prefix(mid) < want)
                lo = mid + 1;
            else
                hi = mid;
        }
        cut[q] = lo;
    }
    return cut;
}

int run(Pool &p, const std::function<int(size_t)> &fn)
{
    const size_t n = p.w.size();
    Job j;
    j.fn = &fn;
    j.pending = n;
    j.status.assign(n, POLYHIP_OK);
    j.msg.resize(n);
    {
        std::lock_guard<std::mutex> lk(p.submit);
        for (size_t q = 0; q < n; ++q) {
            {
                std::lock_guard<std::mutex> lw(p.w[q]->m);
                p.w[q]->q.emplace_back(&j, q);
            }
            p.w[q]->cv.notify_one();
        }
    }
    {
        std::unique_lock<std::mutex> lk(j.m);
        j.cv.wait(lk, [&] { return j.pending == 0; });
    }
    for (size_t q = 0; q < n; ++q)
        if (j.status[q] != POLYHIP_OK)
            return set_error(j.status[q], "%s", j.msg[q].c_str());
    return POLYHIP_OK;
}

} // namespace md
} // namespace polyhip

using namespace polyhip;

extern "C" {

int polyhip_set_devices(const int *ids, int n)
{
    PH_REQUIRE(n >= 0 && n <= 64 && (ids || n == 0), "polyhip_set_devices: expected 0..64 device ids");
    int ndev = 0;
    if (n > 0) {
        PH_HIP(hipGetDeviceCount(&ndev));
        for (int i = 0; i < n; ++i)
            PH_REQUIRE(ids[i] >= 0 && ids[i] < ndev, "polyhip_set_devices: device %d is not one of the %d visible", ids[i], ndev);
    }
    md::Global &g = md::global();
    std::shared_ptr<md::Pool> old;
    {
        std::lock_guard<std::mutex> lk(g.m);
        old = g.cur; // joined outside the lock, once its running calls are through
        md::install(g, std::vector<int>(ids, ids + n));
    }
    return POLYHIP_OK;
}

int polyhip_get_devices(int *ids, int capacity)
{
    (void)md::pool(); // reads POLYHIP_DEVICES if nothing has been configured yet
    md::Global &g = md::global();
    std::lock_guard<std::mutex> lk(g.m);
    const int n = (int)g.ids.size();
    for (int i = 0; i < n && i < capacity && ids; ++i)
        ids[i] = g.ids[i];
    return n;
}

int polyhip_init(int n_devices)
{
    int ndev = 0;
    PH_HIP(hipGetDeviceCount(&ndev));
    PH_REQUIRE(n_devices <= ndev, "polyhip_init: %d devices asked for, %d visible", n_devices, ndev);
    if (n_devices <= 0)
        n_devices = ndev; // "the node's GPUs"
    std::vector<int> ids(n_devices);
    for (int i = 0; i < n_devices; ++i)
        ids[i] = i;
    return polyhip_set_devices(ids.data(), n_devices);
}

int polyhip_shutdown(void) { return polyhip_set_devices(nullptr, 0); }

} // extern "C"
