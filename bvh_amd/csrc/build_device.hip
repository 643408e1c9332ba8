// Dispatch of DefaultBuilder's modes (reference default_builder.h:33-62) onto the device builders.
#include "common.h"

#include <cstdlib>

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

hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t* stream_used, bool* pooled) {
    *pooled = scratch_pool_enabled();
    *stream_used = ambient_stream();
    return *pooled ? hipMallocAsync(p, bytes, *stream_used) : hipMalloc(p, bytes);
}

void scratch_free(void* p, hipStream_t stream, bool pooled) {
    if (!p) return;
    if (pooled) (void)hipFreeAsync(p, stream); else (void)hipFree(p);
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
