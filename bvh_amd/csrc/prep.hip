// K0 — caller-side primitive preparation as kernels (reference test/benchmark.cpp:205-225, tri.h:24-37,
// sphere.h:24-27). Pure streaming: AoS in, AoS out, one primitive per lane.
#include "common.h"

namespace bvh_amd {

namespace {

template <typename T> __device__ inline T pick_min(T a, T b) { return a < b ? a : b; }   // utils.h:41-43
template <typename T> __device__ inline T pick_max(T a, T b) { return a > b ? a : b; }

template <typename T>
__global__ void __launch_bounds__(256) tri_bounds_kernel(const T* tris, size_t n, T* bb, T* cc) {
    size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    T p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = tris[9 * i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T lo = p[k], hi = p[k];                              // BBox(p0).extend(p1).extend(p2), tri.h:24
        lo = pick_min(lo, p[3 + k]); hi = pick_max(hi, p[3 + k]);
        lo = pick_min(lo, p[6 + k]); hi = pick_max(hi, p[6 + k]);
        bb[6 * i + k] = lo;
        bb[6 * i + 3 + k] = hi;
        cc[3 * i + k] = (p[k] + p[3 + k] + p[6 + k]) * static_cast<T>(1. / 3.);   // tri.h:25
    }
}

template <typename T>
__global__ void __launch_bounds__(256) precompute_kernel(const T* tris, const uint32_t* perm, size_t n, T* out) {
    size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    size_t j = perm ? perm[i] : i;
    T p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = tris[9 * j + k];
    T e1[3], e2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e1[k] = p[k] - p[3 + k]; e2[k] = p[6 + k] - p[k]; }   // tri.h:36
    T* o = out + 12 * i;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = p[k]; o[3 + k] = e1[k]; o[6 + k] = e2[k]; }
    o[9]  = e1[1] * e2[2] - e1[2] * e2[1];                   // vec.h:103-108
    o[10] = e1[2] * e2[0] - e1[0] * e2[2];
    o[11] = e1[0] * e2[1] - e1[1] * e2[0];
}

template <typename T>
__global__ void __launch_bounds__(256) sphere_bounds_kernel(const T* sph, size_t n, T* bb, T* cc) {
    size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    T r = sph[4 * i + 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T c = sph[4 * i + k];
        bb[6 * i + k] = c - r;
        bb[6 * i + 3 + k] = c + r;
        cc[3 * i + k] = c;
    }
}

__global__ void __launch_bounds__(256) gather_kernel(const uint32_t* in, const uint32_t* perm, size_t n, uint32_t words, uint32_t* out) {
    size_t g = blockIdx.x * size_t{256} + threadIdx.x;
    size_t total = n * words;
    for (; g < total; g += size_t{gridDim.x} * 256) {
        size_t i = g / words, w = g - i * words;
        out[g] = in[size_t{perm[i]} * words + w];
    }
}

// ---- 2D families: circles, and the three-wide staging of 2D inputs (z = 0) the builders run on ------------------------------
template <typename T>
__global__ void __launch_bounds__(256) circle_bounds_kernel(const T* c3, size_t n, T* bb4, T* cc2) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    const T x = c3[3 * i], y = c3[3 * i + 1], r = c3[3 * i + 2];          // Sphere<T, 2>::get_bbox / get_center (sphere.h:24-27)
    bb4[4 * i + 0] = x - r; bb4[4 * i + 1] = y - r; bb4[4 * i + 2] = x + r; bb4[4 * i + 3] = y + r;
    cc2[2 * i + 0] = x; cc2[2 * i + 1] = y;
}

template <typename T>
__global__ void __launch_bounds__(256) widen_inputs_kernel(const T* bb4, const T* cc2, size_t n, T* bb6, T* cc3) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    bb6[6 * i + 0] = bb4[4 * i + 0]; bb6[6 * i + 1] = bb4[4 * i + 1]; bb6[6 * i + 2] = T(0);    // {min.x, min.y, 0, max.x, max.y, 0}
    bb6[6 * i + 3] = bb4[4 * i + 2]; bb6[6 * i + 4] = bb4[4 * i + 3]; bb6[6 * i + 5] = T(0);
    cc3[3 * i + 0] = cc2[2 * i + 0]; cc3[3 * i + 1] = cc2[2 * i + 1]; cc3[3 * i + 2] = T(0);
}

inline unsigned blocks_for(size_t n) { return static_cast<unsigned>((n + 255) / 256); }

} // namespace

template <typename T>
int launch_circle_bounds(const T* d_circles3, size_t n, T* d_bb4, T* d_cc2, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    hipLaunchKernelGGL(circle_bounds_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, s, d_circles3, n, d_bb4, d_cc2);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}
template <typename T>
int launch_widen_inputs(const T* d_bb4, const T* d_cc2, size_t n, T* d_bb6, T* d_cc3, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    hipLaunchKernelGGL(widen_inputs_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, s, d_bb4, d_cc2, n, d_bb6, d_cc3);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}
template int launch_circle_bounds<float>(const float*, size_t, float*, float*, hipStream_t);
template int launch_circle_bounds<double>(const double*, size_t, double*, double*, hipStream_t);
template int launch_widen_inputs<float>(const float*, const float*, size_t, float*, float*, hipStream_t);
template int launch_widen_inputs<double>(const double*, const double*, size_t, double*, double*, hipStream_t);

template <typename T>
int launch_tri_bounds(const T* d_tris9, size_t n, T* d_bb, T* d_cc, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    hipLaunchKernelGGL(tri_bounds_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, s, d_tris9, n, d_bb, d_cc);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template <typename T>
int launch_precompute_tris(const T* d_tris9, const uint32_t* d_perm, size_t n, T* d_out, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    hipLaunchKernelGGL(precompute_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, s, d_tris9, d_perm, n, d_out);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template <typename T>
int launch_sphere_bounds(const T* d_sph4, size_t n, T* d_bb, T* d_cc, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    hipLaunchKernelGGL(sphere_bounds_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, s, d_sph4, n, d_bb, d_cc);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

int launch_gather(const void* d_in, const uint32_t* d_perm, size_t n, size_t stride, void* d_out, hipStream_t s) {
    if (!n) return BVH_AMD_OK;
    if (stride % 4) return fail(BVH_AMD_ERR_ARG, "gather: stride must be a multiple of 4 bytes");
    uint32_t words = static_cast<uint32_t>(stride / 4);
    size_t total = n * words;
    unsigned grid = static_cast<unsigned>(std::min<size_t>((total + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, s, static_cast<const uint32_t*>(d_in), d_perm, n, words,
                       static_cast<uint32_t*>(d_out));
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template int launch_tri_bounds<float>(const float*, size_t, float*, float*, hipStream_t);
template int launch_tri_bounds<double>(const double*, size_t, double*, double*, hipStream_t);
template int launch_precompute_tris<float>(const float*, const uint32_t*, size_t, float*, hipStream_t);
template int launch_precompute_tris<double>(const double*, const uint32_t*, size_t, double*, hipStream_t);
template int launch_sphere_bounds<float>(const float*, size_t, float*, float*, hipStream_t);
template int launch_sphere_bounds<double>(const double*, size_t, double*, double*, hipStream_t);

} // namespace bvh_amd
