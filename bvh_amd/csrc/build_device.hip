// Dispatch of DefaultBuilder's modes (reference default_builder.h:33-62) onto the device builders.
#include "common.h"

namespace bvh_amd {

SahParams& ambient_sah() {
    static thread_local SahParams params;
    return params;
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
