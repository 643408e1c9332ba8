// Caller-side steps of the reference's benchmark renderer (test/benchmark.cpp:340-393) as device kernels, so that the
// build -> permute -> trace -> shade pipeline stays in HBM end to end: primary rays of its pinhole camera (:343-359) and the
// eyelight shading of the hits (:363-371). Same float expressions in the same order (no contraction), so the PPM the
// reference writes for the Cornell box (md5 96f6bbdc...) is reproduced byte for byte (tests/test_gpu_traverse.py).
#include "common.h"

#include <cfloat>
#include <cmath>

namespace bvh_amd {

namespace {

template <typename T> struct Lim;
template <> struct Lim<float>  { static constexpr float max = FLT_MAX; };
template <> struct Lim<double> { static constexpr double max = DBL_MAX; };

template <typename T> struct V3 { T x, y, z; };
template <typename T> __host__ __device__ inline T dot3(V3<T> a, V3<T> b) { return ((T(0) + a.x * b.x) + a.y * b.y) + a.z * b.z; }   // vec.h:98-100
template <typename T> __host__ __device__ inline V3<T> cross3(V3<T> a, V3<T> b) {
    return V3<T>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> __host__ __device__ inline V3<T> normalize3(V3<T> a) {          // v * (1 / length(v))
    const T inv = T(1) / std::sqrt(dot3(a, a));
    return V3<T>{a.x * inv, a.y * inv, a.z * inv};
}

template <typename T>
__global__ void __launch_bounds__(256) pinhole_kernel(V3<T> eye, V3<T> dir, V3<T> right, V3<T> up, uint32_t width, uint32_t height, T* rays) {
    const size_t i = size_t{blockIdx.x} * 256 + threadIdx.x;
    if (i >= size_t{width} * height) return;
    const uint32_t x = static_cast<uint32_t>(i % width), y = static_cast<uint32_t>(i / width);
    const T u = T(2) * static_cast<T>(x) / static_cast<T>(width) - T(1);
    const T v = T(2) * static_cast<T>(y) / static_cast<T>(height) - T(1);
    T* r = rays + 8 * i;                                      // Ray(eye, dir + u * right + v * up): tmin 0, tmax max (ray.h:20-27)
    r[0] = eye.x; r[1] = eye.y; r[2] = eye.z;
    r[3] = (dir.x + u * right.x) + v * up.x;
    r[4] = (dir.y + u * right.y) + v * up.y;
    r[5] = (dir.z + u * right.z) + v * up.z;
    r[6] = T(0);
    r[7] = Lim<T>::max;
}

template <typename T>
__global__ void __launch_bounds__(256) eyelight_kernel(const T* tris12, const T* rays, const typename HitOf<T>::Type* hits, size_t n, uint8_t* rgb) {
    const size_t i = size_t{blockIdx.x} * 256 + threadIdx.x;
    if (i >= n) return;
    T intensity = T(0);
    const uint32_t prim = hits[i].prim;
    if (prim != BVH_AMD_INVALID) {
        const T* t = tris12 + 12 * size_t{prim};
        const V3<T> nrm = normalize3(V3<T>{t[9], t[10], t[11]});
        const T d = dot3(nrm, V3<T>{rays[8 * i + 3], rays[8 * i + 4], rays[8 * i + 5]});
        intensity = d < T(0) ? -d : d;
    }
    const int k = static_cast<int>(intensity * T(256));
    const uint8_t pixel = static_cast<uint8_t>(k < 0 ? 0 : (k > 255 ? 255 : k));
    rgb[3 * i + 0] = pixel; rgb[3 * i + 1] = pixel; rgb[3 * i + 2] = pixel;
}

} // namespace

template <typename T>
int launch_pinhole_rays(const T eye[3], const T dir[3], const T up[3], size_t width, size_t height, T* d_rays, hipStream_t stream) {
    if (!eye || !dir || !up || (!d_rays && width * height)) return fail(BVH_AMD_ERR_ARG, "pinhole_rays: null argument");
    if (width == 0 || height == 0) return BVH_AMD_OK;
    if (width > 0xffffffffull || height > 0xffffffffull) return fail(BVH_AMD_ERR_ARG, "pinhole_rays: image too large");
    const V3<T> d = normalize3(V3<T>{dir[0], dir[1], dir[2]});                         // benchmark.cpp:343-345
    const V3<T> right = normalize3(cross3(d, V3<T>{up[0], up[1], up[2]}));
    const V3<T> upv = cross3(right, d);
    const size_t n = width * height;
    hipLaunchKernelGGL(pinhole_kernel<T>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, V3<T>{eye[0], eye[1], eye[2]}, d, right, upv,
                       static_cast<uint32_t>(width), static_cast<uint32_t>(height), d_rays);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template <typename T>
int launch_shade_eyelight(const T* d_tris12, const T* d_rays, const typename HitOf<T>::Type* d_hits, size_t n, uint8_t* d_rgb, hipStream_t stream) {
    if (n == 0) return BVH_AMD_OK;
    if (!d_tris12 || !d_rays || !d_hits || !d_rgb) return fail(BVH_AMD_ERR_ARG, "shade_eyelight: null argument");
    hipLaunchKernelGGL(eyelight_kernel<T>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, d_tris12, d_rays, d_hits, n, d_rgb);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template int launch_pinhole_rays<float>(const float*, const float*, const float*, size_t, size_t, float*, hipStream_t);
template int launch_pinhole_rays<double>(const double*, const double*, const double*, size_t, size_t, double*, hipStream_t);
template int launch_shade_eyelight<float>(const float*, const float*, const bvh_hit3f*, size_t, uint8_t*, hipStream_t);
template int launch_shade_eyelight<double>(const double*, const double*, const bvh_hit3d*, size_t, uint8_t*, hipStream_t);

} // namespace bvh_amd
