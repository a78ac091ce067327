// multi_device.h -- one host-pointer (cgo) call spread over several GPUs of the node (polyhip_init(n_devices),
// 8e: reads / pairs / starts / rows shard with no data-path collective).
//
// The library keeps ONE device list per process (polyhip_set_devices / polyhip_init / POLYHIP_DEVICES in the
// environment).  While it is empty every host-pointer entry point runs on the calling thread's current device, as it
// always did.  With a list of n entries the entry points cut their batch into n contiguous shards balanced by bytes and
// hand shard q to worker thread q, which lives on device ids[q] for the life of the list (its own two streams, its own
// two-slot pipeline: host_pipeline.h) and writes its results straight into the caller's slices.  The list may name a
// device more than once ("0,0,0"): the shards then share that GPU -- that is how the fan-out is tested on a one-GPU box.
// Everything stays inside the process: no RCCL, no one-rank-per-device rule; where devices exchange data (the sketches
// in front of an all-vs-all distance matrix) it is hipMemcpyPeerAsync.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "common.h"

namespace polyhip {
namespace md {

struct Pool; // the worker threads of one device list

// The pool a host-pointer entry point should fan out over, or null: no list is configured, or the caller IS one of the
// pool's workers (the shards call the single-device bodies, which never come back here).
std::shared_ptr<Pool> pool();
size_t size(const Pool &p);
int device(const Pool &p, size_t worker);

// nshards + 1 ascending cut points over n items; prefix(i) = the (non-decreasing) cost of items [0, i).  Shard q = items
// [cut[q], cut[q + 1]) holds about 1 / nshards of the cost; shards may be empty.
std::vector<uint64_t> split(uint64_t n, size_t nshards, const std::function<uint64_t(uint64_t)> &prefix);

// fn(q) on worker q (its device current) for every q < size(p); returns when all are done.  POLYHIP_OK, or the status
// of the LOWEST failing shard with its message as the calling thread's polyhip_last_error() -- shards are contiguous
// and in order, so that is the failure the single-device call would have reported first.
int run(Pool &p, const std::function<int(size_t)> &fn);

// What a shard adds to the item / byte positions its error messages name, so that they read as positions of the whole
// batch.  Thread-local; zero outside a shard.
struct Base {
    uint64_t item = 0, byte = 0;
};
Base &base();
struct BaseScope {
    Base saved;
    BaseScope(uint64_t item, uint64_t byte) : saved(base())
    {
        base().item = saved.item + item;
        base().byte = saved.byte + byte;
    }
    ~BaseScope() { base() = saved; }
};

} // namespace md
} // namespace polyhip
