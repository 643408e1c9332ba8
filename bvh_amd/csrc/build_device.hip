// Dispatch of DefaultBuilder's modes (reference default_builder.h:33-62) onto the device builders.
#include "common.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace bvh_amd {

SahParams& ambient_sah() {
    static thread_local SahParams params;
    return params;
}

hipStream_t& ambient_stream() {
    static thread_local hipStream_t stream = nullptr;
    return stream;
}

bool scratch_pool_enabled() {
    static const bool wanted = !(std::getenv("BVH_AMD_POOL") && std::atoi(std::getenv("BVH_AMD_POOL")) == 0);
    if (!wanted) return false;
    static std::mutex m;
    static bool configured[64] = {};
    static bool usable[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(m);
    if (!configured[dev]) {
        configured[dev] = true;
        hipMemPool_t pool = nullptr;
        int supported = 0;
        if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, dev) == hipSuccess && supported &&
            hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
            uint64_t keep = ~uint64_t{0};                      // freed blocks stay with the pool until bvh_amd_release_cached_memory()
            usable[dev] = hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess;
        }
        (void)hipGetLastError();
    }
    return usable[dev];
}

namespace {

// Freed scratch blocks by (device, stream), each list ordered by capacity; `clock` orders evictions (oldest first).
struct CachedBlock { void* p; uint64_t age; };
struct ScratchCache {
    std::mutex m;
    std::map<std::pair<int, hipStream_t>, std::multimap<size_t, CachedBlock>> lists;
    size_t cached_bytes[64] = {};
    uint64_t clock = 0;
    size_t limit = size_t{8192} << 20;
    ScratchCache() {
        if (const char* e = std::getenv("BVH_AMD_CACHE_MB")) limit = static_cast<size_t>(std::max(0ll, std::atoll(e))) << 20;
    }
};
ScratchCache& scratch_cache() { static ScratchCache* c = new ScratchCache; return *c; }      // outlives the runtime's own teardown

} // namespace

hipError_t scratch_alloc(void** p, size_t bytes, ScratchTag* tag) {
    tag->pooled = scratch_pool_enabled();
    tag->stream = ambient_stream();
    tag->capacity = bytes;
    if (!tag->pooled) return hipMalloc(p, bytes);
    ScratchCache& c = scratch_cache();
    int dev = 0;
    if (c.limit && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lock(c.m);
        auto l = c.lists.find({ dev, tag->stream });
        if (l != c.lists.end()) {
            auto it = l->second.lower_bound(bytes);
            if (it != l->second.end() && it->first <= bytes + bytes / 4 + 4096) {       // near fit only: a big block must not serve small requests
                *p = it->second.p;
                tag->capacity = it->first;
                c.cached_bytes[dev] -= it->first;
                l->second.erase(it);
                return hipSuccess;
            }
        }
    }
    return hipMallocAsync(p, bytes, tag->stream);
}

void scratch_free(void* p, const ScratchTag& tag) {
    if (!p) return;
    if (!tag.pooled) { (void)hipFree(p); return; }
    ScratchCache& c = scratch_cache();
    int dev = 0;
    if (c.limit && tag.capacity <= c.limit && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lock(c.m);
        c.lists[{ dev, tag.stream }].emplace(tag.capacity, CachedBlock{ p, ++c.clock });
        c.cached_bytes[dev] += tag.capacity;
        while (c.cached_bytes[dev] > c.limit) {              // over the bound: the oldest block of this device goes back to the pool
            std::multimap<size_t, CachedBlock>* from = nullptr;
            std::multimap<size_t, CachedBlock>::iterator oldest;
            hipStream_t on = nullptr;
            for (auto& l : c.lists) {
                if (l.first.first != dev) continue;
                for (auto it = l.second.begin(); it != l.second.end(); ++it)
                    if (!from || it->second.age < oldest->second.age) { from = &l.second; oldest = it; on = l.first.second; }
            }
            if (!from) break;
            // (a stream destroyed since: the free is ordered on the null stream instead — any stream may free stream-ordered memory; a
            //  plain hipFree would have the runtime look at the allocating stream again)
            if (hipFreeAsync(oldest->second.p, on) != hipSuccess) { (void)hipGetLastError(); if (hipFreeAsync(oldest->second.p, nullptr) != hipSuccess) (void)hipGetLastError(); }
            c.cached_bytes[dev] -= oldest->first;
            from->erase(oldest);
        }
        return;
    }
    (void)hipFreeAsync(p, tag.stream);
}

void scratch_cache_flush() {
    ScratchCache& c = scratch_cache();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(c.m);
    for (auto l = c.lists.begin(); l != c.lists.end();) {
        if (l->first.first != dev) { ++l; continue; }
        for (auto& b : l->second)
            if (hipFreeAsync(b.second.p, l->first.second) != hipSuccess) { (void)hipGetLastError(); if (hipFreeAsync(b.second.p, nullptr) != hipSuccess) (void)hipGetLastError(); }
        l = c.lists.erase(l);
    }
    if (dev >= 0 && dev < 64) c.cached_bytes[dev] = 0;
}

// A stream the library itself owns (the mini-tree builder's worker stream) is about to be destroyed: nothing may stay cached under
// its handle — a later eviction would hand the runtime a dead stream (seen as a hang of the GPU suite, round 4: the first eviction
// after host threads with worker streams of their own had ended).
void scratch_cache_drop_stream(hipStream_t s) {
    ScratchCache& c = scratch_cache();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(c.m);
    auto l = c.lists.find({ dev, s });
    if (l == c.lists.end()) return;
    for (auto& b : l->second) {
        if (hipFreeAsync(b.second.p, s) != hipSuccess) { (void)hipGetLastError(); if (hipFreeAsync(b.second.p, nullptr) != hipSuccess) (void)hipGetLastError(); }
        if (dev >= 0 && dev < 64) c.cached_bytes[dev] -= b.first;
    }
    c.lists.erase(l);
}

namespace {

constexpr size_t kReadbackWords = 64;

__global__ void __launch_bounds__(64) k_readback(const uint32_t* src, uint32_t words, uint32_t* host, uint32_t seq) {
    if (threadIdx.x == 0) {
        for (uint32_t w = 0; w < words; ++w) host[1 + w] = src[w];
        // the payload before the sequence number, both visible to the polling host while the kernel is still retiring
        __hip_atomic_store(&host[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Pinned words are handed from thread to thread: a thread that ends returns its words here and the next new thread of the device takes
// them over (the mini-tree builder's top-level worker is a thread per build: a hipHostMalloc per build would cost more than the
// worker saves). The words of the pool live as long as the process.
struct ReadbackPool {
    std::mutex m;
    std::multimap<int, uint32_t*> idle;
};
ReadbackPool& readback_pool() { static ReadbackPool* p = new ReadbackPool; return *p; }

struct ReadbackSlot {                          // one per calling thread and device
    int device = -1;
    uint32_t* pinned = nullptr;
    uint32_t seq = 0;
    bool broken = false;
    void give_back() {
        if (!pinned) return;
        ReadbackPool& pool = readback_pool();
        std::lock_guard<std::mutex> lock(pool.m);
        pool.idle.emplace(device, pinned);
        pinned = nullptr; device = -1;
    }
    bool take(int dev) {
        ReadbackPool& pool = readback_pool();
        std::lock_guard<std::mutex> lock(pool.m);
        auto it = pool.idle.find(dev);
        if (it == pool.idle.end()) return false;
        pinned = it->second; device = dev;
        seq = pinned[0];                        // continue the sequence its last owner left (the arrival test compares with the word)
        pool.idle.erase(it);
        return true;
    }
    ~ReadbackSlot() { give_back(); }
};

} // namespace

int readback(void* dst, const void* d_src, size_t bytes, hipStream_t stream) {
    static const bool blocking = std::getenv("BVH_AMD_READBACK") && std::strcmp(std::getenv("BVH_AMD_READBACK"), "sync") == 0;
    static thread_local ReadbackSlot slot;
    int dev = -1;
    BVH_HIP_TRY(hipGetDevice(&dev), BVH_AMD_ERR_HIP);
    const bool fits = bytes % 4 == 0 && bytes / 4 <= kReadbackWords && reinterpret_cast<uintptr_t>(d_src) % 4 == 0;
    if (!blocking && fits && !slot.broken) {
        if (slot.device != dev) {
            slot.give_back();
            if (!slot.take(dev)) {
                if (hipHostMalloc(reinterpret_cast<void**>(&slot.pinned), (kReadbackWords + 1) * sizeof(uint32_t), hipHostMallocCoherent) != hipSuccess) {
                    (void)hipGetLastError();
                    slot.pinned = nullptr;
                    slot.broken = true;
                } else {
                    slot.pinned[0] = 0; slot.seq = 0; slot.device = dev;
                }
            }
        }
        if (!slot.broken) {
            const uint32_t seq = ++slot.seq ? slot.seq : ++slot.seq;          // never 0
            hipLaunchKernelGGL(k_readback, dim3(1), dim3(64), 0, stream, static_cast<const uint32_t*>(d_src), static_cast<uint32_t>(bytes / 4), slot.pinned, seq);
            BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
            bool arrived = false;
            for (uint32_t spin = 0;; ++spin) {
                if (__atomic_load_n(&slot.pinned[0], __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() > give_up) break;
            }
            if (!arrived) {                                   // long-running work ahead of us on the stream: block like everybody else
                BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
                if (__atomic_load_n(&slot.pinned[0], __ATOMIC_ACQUIRE) != seq) return fail(BVH_AMD_ERR_HIP, "readback: the copy kernel did not run");
            }
            std::memcpy(dst, slot.pinned + 1, bytes);
            return BVH_AMD_OK;
        }
    }
    BVH_HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template <typename T>
int build_binned_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                        hipStream_t stream);

template <typename T>
int build_sweep_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                       bool optimize, hipStream_t stream);

template <typename T>
int build_minitree_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg, hipStream_t stream);

template <typename T>
int build_on_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                    bvh_amd_builder builder, hipStream_t stream)
{
    StreamScope scratch_on(stream);
    // default_builder.h:39-40: the parallel overload falls back to the serial one below the threshold
    if (builder == BVH_AMD_BUILDER_DEFAULT_PARALLEL && n < cfg.parallel_threshold) builder = BVH_AMD_BUILDER_DEFAULT_SERIAL;
    if (builder == BVH_AMD_BUILDER_BINNED || (builder == BVH_AMD_BUILDER_DEFAULT_SERIAL && cfg.quality == BVH_BUILD_QUALITY_LOW))
        return build_binned_device<T>(out, d_bboxes, d_centers, n, cfg, stream);
    if (builder == BVH_AMD_BUILDER_SWEEP) return build_sweep_device<T>(out, d_bboxes, d_centers, n, cfg, false, stream);
    if (builder == BVH_AMD_BUILDER_DEFAULT_SERIAL)            // Medium: sweep; High: sweep + reinsertion (default_builder.h:56-60)
        return build_sweep_device<T>(out, d_bboxes, d_centers, n, cfg, cfg.quality == BVH_BUILD_QUALITY_HIGH, stream);
    if (builder == BVH_AMD_BUILDER_DEFAULT_PARALLEL) {
        // 2D: the reference's mini-tree builder reads p[2] of a 2D vector (mini_tree_builder.h:183), undefined behaviour
        // that no implementation can reproduce; refused loudly instead of guessed
        if (out.dim == 2)
            return fail(BVH_AMD_ERR_UNSUPPORTED, "build: 2D BVHs have no defined thread-pool build at or above parallel_threshold "
                                                  "(the reference reads the third component of 2D points); pass pool = NULL");
        return build_minitree_device<T>(out, d_bboxes, d_centers, n, cfg, stream);
    }
    return fail(BVH_AMD_ERR_ARG, "build: unknown builder");
}

template int build_on_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, bvh_amd_builder, hipStream_t);
template int build_on_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, bvh_amd_builder, hipStream_t);

} // namespace bvh_amd
