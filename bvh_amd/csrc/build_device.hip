// Device builders (placeholder dispatch until build_binned.hip lands).
#include "common.h"

namespace bvh_amd {

template <typename T>
int build_on_device(BvhImpl<T>&, const T*, const T*, size_t, const bvh_build_config&, bvh_amd_builder, hipStream_t) {
    return fail(BVH_AMD_ERR_UNSUPPORTED, "build: this builder mode is not implemented on the device yet");
}

template int build_on_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, bvh_amd_builder, hipStream_t);
template int build_on_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, bvh_amd_builder, hipStream_t);

} // namespace bvh_amd
