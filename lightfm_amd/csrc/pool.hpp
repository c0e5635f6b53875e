// pool.hpp -- process-wide device-memory pool of liblfm_hip.so.
//
// Every device allocation of the library (weight tables, COO / CSR arrays, shuffle slots, the
// rocPRIM temporaries of csr_build.hip, per-call scratch) comes from here and goes back here; blocks
// are handed to the HIP runtime again only by lfm_device_trim() or when an allocation fails.
//
// Why (round 3, DESIGN.md "Root cause of the round-2 process abort"): on the MI355X boxes a buffer
// obtained from hipMalloc shortly after other buffers had been hipFree'd -- in particular memory that had
// been mapped UNCACHED (hipDeviceMallocUncached, the weight tables of large models) or had held the
// rocPRIM sort's key buffers -- could keep serving OLD contents to the wavefronts of some XCDs long
// after a kernel of the same stream had rewritten it (in-kernel probe: cached, cache-bypassing and
// post-buffer_inv loads all returned the old words, the host's read-back the new ones).  A shuffle
// entry read that way indexed the COO out of range: "Memory access fault by GPU", SIGABRT of the host
// process.  Blocks that are never returned to the runtime keep their mapping and their memory type
// for the life of the process: the cached and the uncached pool never exchange memory.
//
// A block is reused only for requests of its own allocation flags (memory type) and of its size class or a
// smaller one down to half its size.  Size classes are eighths of powers of two (at most 12.5 % of slack), 4 KiB
// at least.  Releasing a block does NOT
// synchronise anything (hipFree did): owners release after the work that uses the block has been
// waited for (lfm_session's destructor and every entry point with per-call buffers drain their
// stream first).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace lfm {

class DevPool {
public:
    static DevPool &instance()
    {
        static DevPool *p = new DevPool();  // never destroyed: the HIP runtime may already be gone at exit
        return *p;
    }
    // LIGHTFM_AMD_POOL=0 (experiments only): plain hipMalloc / hipFree
    static bool enabled()
    {
        static const bool on = [] {
            const char *e = getenv("LIGHTFM_AMD_POOL");
            return !(e && atoi(e) == 0);
        }();
        return on;
    }
    static size_t size_class(size_t bytes)
    {
        if (bytes <= 4096) return 4096;
        size_t p2 = 4096;
        while (p2 < bytes) p2 <<= 1;       // smallest power of two >= bytes
        const size_t step = p2 >> 4;        // sixteenth of it = eighth of the power of two below
        return (bytes + step - 1) / step * step;
    }
    // flags: 0 = hipMalloc, otherwise hipExtMallocWithFlags flags
    hipError_t alloc(void **out, size_t bytes, int flags)
    {
        *out = nullptr;
        if (bytes == 0) return hipSuccess;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!enabled()) return raw_alloc(out, bytes, flags);
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> lk(mu_);
            // the smallest cached block of this device and memory type that fits, up to twice the request (calls
            // of varying size -- predict, predict_rank -- then share blocks instead of leaving one per class behind)
            for (auto it = free_.lower_bound(Key{dev, flags, cls}); it != free_.end(); ++it) {
                const Key &k = it->first;
                if (k.dev != dev || k.flags != flags || k.cls > 2 * cls) break;
                if (it->second.empty()) continue;
                *out = it->second.back();
                it->second.pop_back();
                cached_bytes_ -= k.cls;
                live_[*out] = k;  // the block keeps its own class
                return hipSuccess;
            }
        }
        hipError_t e = raw_alloc(out, cls, flags);
        if (e != hipSuccess) {  // give everything back to the runtime and try once more
            (void)hipGetLastError();
            trim();
            e = raw_alloc(out, cls, flags);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(mu_);
            live_[*out] = Key{dev, flags, cls};
            reserved_bytes_ += cls;
        }
        return e;
    }
    void release(void *p)
    {
        if (!p) return;
        if (!enabled()) {
            (void)hipFree(p);
            return;
        }
        std::lock_guard<std::mutex> lk(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) {  // not ours (cannot happen): hand it to the runtime
            (void)hipFree(p);
            return;
        }
        free_[it->second].push_back(p);
        cached_bytes_ += it->second.cls;
        live_.erase(it);
    }
    // Returns every cached (unused) block to the HIP runtime; the bytes released.
    size_t trim()
    {
        std::vector<std::pair<int, void *>> blocks;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto &kv : free_) {
                for (void *p : kv.second) blocks.emplace_back(kv.first.dev, p);
                bytes += kv.first.cls * kv.second.size();
                kv.second.clear();
            }
            cached_bytes_ = 0;
            reserved_bytes_ -= bytes;
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto &b : blocks) {
            (void)hipSetDevice(b.first);
            (void)hipFree(b.second);
        }
        (void)hipSetDevice(cur);
        return bytes;
    }
    void stats(size_t *reserved, size_t *cached)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (reserved) *reserved = reserved_bytes_;
        if (cached) *cached = cached_bytes_;
    }

private:
    struct Key {
        int dev, flags;
        size_t cls;
        bool operator<(const Key &o) const
        {
            if (dev != o.dev) return dev < o.dev;
            if (flags != o.flags) return flags < o.flags;
            return cls < o.cls;
        }
    };
    static hipError_t raw_alloc(void **out, size_t bytes, int flags)
    {
        return flags ? hipExtMallocWithFlags(out, bytes, (unsigned)flags) : hipMalloc(out, bytes);
    }
    std::mutex mu_;
    std::map<Key, std::vector<void *>> free_;
    std::map<void *, Key> live_;
    size_t reserved_bytes_ = 0, cached_bytes_ = 0;
};

inline hipError_t pool_alloc(void **out, size_t bytes, int flags = 0) { return DevPool::instance().alloc(out, bytes, flags); }
inline void pool_free(void *p) { DevPool::instance().release(p); }

}  // namespace lfm
