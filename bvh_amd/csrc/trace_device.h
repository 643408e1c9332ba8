// Device-side helpers of the batch traversal kernels (traverse.hip): per-type arithmetic, record / ray / primitive loads, the
// kernel argument block. Kept apart from traverse.hip so that tests/cpp/trace_body_host.cpp can compile the kernel BODY
// (trace_body.inc) for the host with one emulated lane (test infrastructure; the product only ever runs it on the device).
// Expects common.h (or the test's stand-ins for it) to have been included: PairNode, HitOf, kCountBits, kCountMask, kWave, LEAF_*.
#pragma once

// An empty asm the value must pass through: stops the compiler from fusing the loads on either side of it.
#if defined(__HIPCC__)
#define BVH_AMD_KEEP_APART(v) asm volatile("" : "+v"(v))
#else
#define BVH_AMD_KEEP_APART(v) asm volatile("" : "+r"(v))
#endif

namespace bvh_amd {

namespace {

constexpr int kBlock = 256;
constexpr int kLdsDepth = 20;
// Refill when at least this many lanes of the wave are idle / run the leaf code when at least this many lanes wait at a leaf.
// Swept on the device (profiles/README.md): a walk served by the L1 / L2s (small trees, coherent or coherence-sorted batches) is
// bound by issue slots and wants its lanes refilled early (soup_1m sorted: 1744 Mrays/s at 54 / 8, 1970 at 36 / 12; Sponza proxy
// 5.07 -> 5.77 Grays/s, 1M-triangle terrain 5.35 -> 6.59). Only a walk that misses the L2s all the time (unsorted uniform rays on
// the 1M soup) would rather have rare, full refills, and by little: 1404 Mrays/s at 54 / 8, 1382 at 36 / 12.
constexpr int kRefillThreshold = 36;
constexpr int kLeafThreshold = 12;


template <typename T> struct Num;
template <> struct Num<float> {
    static constexpr float kMax = 3.402823466e+38f, kEps = 1.1920928955078125e-07f;
    __device__ static bool finite(float x) { return (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u; }
    __device__ static float bump2(float x) { return __uint_as_float(__float_as_uint(x) + 2u); }
    __device__ static bool sign(float x) { return (__float_as_uint(x) >> 31) != 0; }
    __device__ static float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    __device__ static float sqrt_(float x) { return __builtin_sqrtf(x); }
    __device__ static float abs_(float x) { return __builtin_fabsf(x); }
    __device__ static float copysign_(float a, float b) { return __builtin_copysignf(a, b); }
    __device__ static float max_num(float a, float b) { return __builtin_fmaxf(a, b); }      // v_max_f32: the non-NaN operand wins
    __device__ static float min_num(float a, float b) { return __builtin_fminf(a, b); }
};
template <> struct Num<double> {
    static constexpr double kMax = 1.7976931348623157e+308, kEps = 2.220446049250313e-16;
    __device__ static bool finite(double x) {
        return (static_cast<uint32_t>(__double_as_longlong(x) >> 32) & 0x7FF00000u) != 0x7FF00000u;
    }
    __device__ static double bump2(double x) { return __longlong_as_double(__double_as_longlong(x) + 2ll); }
    __device__ static bool sign(double x) { return (__double_as_longlong(x) >> 63) != 0; }
    __device__ static double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ static double sqrt_(double x) { return __builtin_sqrt(x); }
    __device__ static double abs_(double x) { return __builtin_fabs(x); }
    __device__ static double copysign_(double a, double b) { return __builtin_copysign(a, b); }
    __device__ static double max_num(double a, double b) { return __builtin_fmax(a, b); }
    __device__ static double min_num(double a, double b) { return __builtin_fmin(a, b); }
};

// utils.h:41-43 — must stay compare+select: the second operand wins on NaN and on equality.
template <typename T> __device__ inline T pick_min(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ inline T pick_max(T a, T b) { return a > b ? a : b; }

template <typename T> __device__ inline T dot3(T a0, T a1, T a2, T b0, T b1, T b2) {   // vec.h:98-100
    return ((T(0) + a0 * b0) + a1 * b1) + a2 * b2;
}
// A ds_read_b32 the optimiser may neither drop nor merge with a load from another address space.
__device__ inline uint32_t lds_word(const uint32_t* shared) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const volatile uint32_t LdsWord;
    return *(LdsWord*)shared;
#else
    return *shared;                                // (tests/cpp/trace_body_host.cpp compiles the body for the host)
#endif
}

template <typename T> __device__ inline T dot2(T a0, T a1, T b0, T b1) { return (T(0) + a0 * b0) + a1 * b1; }   // Vec<T, 2>

constexpr uint32_t kTicketStride = 16;       // ticket counters 128 bytes apart

template <typename T>
struct TraceArgs {
    const PairNode<T>* pairs;
    const T* prims;
    const T* rays;
    typename HitOf<T>::Type* hits;
    unsigned long long n;
    unsigned long long* work;                  // next ray ticket of part p of this launch at work[p * kTicketStride]
    unsigned long long part_size;              // tickets [p * part_size, min(n, (p + 1) * part_size)) form part p (a multiple of 64)
    uint32_t parts;                            // 1, or one part per XCD (see trace_body.inc: refill)
    bvh_amd_counters* counters;
    const uint32_t* order;                     // optional: ticket -> ray index (coherence sort); results are unaffected
    uint32_t* deep;                            // stack entries beyond 64, deep_cap per resident lane (trees deeper than 64 levels only)
    uint32_t deep_cap;
    uint32_t root_index;
    int refill_threshold;                      // refill when at least this many lanes are idle
    int leaf_threshold;                        // leave the inner-node loop when this many lanes wait at a leaf
};

__device__ inline void load_pair(const PairNode<float>* p, float (&lb)[6], float (&rb)[6], uint32_t& li, uint32_t& ri) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1], c = q[2];
    uint2 d = reinterpret_cast<const uint2*>(p)[6];
    lb[0] = a.x; lb[1] = a.y; lb[2] = a.z; lb[3] = a.w; lb[4] = b.x; lb[5] = b.y;
    rb[0] = b.z; rb[1] = b.w; rb[2] = c.x; rb[3] = c.y; rb[4] = c.z; rb[5] = c.w;
    li = d.x; ri = d.y;
}
__device__ inline void load_pair(const PairNode<double>* p, double (&lb)[6], double (&rb)[6], uint32_t& li, uint32_t& ri) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 v = q[i]; lb[2 * i] = v.x; lb[2 * i + 1] = v.y; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 v = q[3 + i]; rb[2 * i] = v.x; rb[2 * i + 1] = v.y; }
    uint2 d = reinterpret_cast<const uint2*>(p)[12];
    li = d.x; ri = d.y;
}

__device__ inline void load_prim12(const float* p, float (&v)[12]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ inline void load_prim12(const double* p, double (&v)[12]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 6; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
__device__ inline void load_prim4(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ inline void load_prim4(const double* p, double (&v)[4]) {
    const double2* q = reinterpret_cast<const double2*>(p);
    double2 a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ inline void load_ray(const float* p, float (&v)[8]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void load_ray(const double* p, double (&v)[8]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
// Ray<T, 2>: {org[2], dir[2], tmin, tmax}
__device__ inline void load_ray2(const float* p, float (&v)[6]) {
    const float2* q = reinterpret_cast<const float2*>(p);
    float2 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}
__device__ inline void load_ray2(const double* p, double (&v)[6]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
__device__ inline void store_hit(bvh_hit3f* out, uint32_t prim, float t, float u, float v) {
    *reinterpret_cast<float4*>(out) = make_float4(__uint_as_float(prim), t, u, v);
}
__device__ inline void store_hit(bvh_hit3d* out, uint32_t prim, double t, double u, double v) {
    double2* q = reinterpret_cast<double2*>(out);
    q[0] = make_double2(__longlong_as_double(static_cast<long long>(prim)), t);
    q[1] = make_double2(u, v);
}

} // namespace

} // namespace bvh_amd
