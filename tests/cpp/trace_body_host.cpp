// TEST INFRASTRUCTURE (CPU, no GPU): compiles the text of the batch traversal kernels — bvh_amd/csrc/trace_body.inc with the device
// helpers of bvh_amd/csrc/trace_device.h — for the HOST and runs it with ONE emulated lane (refill and leaf thresholds 1, the
// wave intrinsics reduced to their single-lane meaning). What it can show: the per-ray logic of the very source the device runs —
// record addressing, push / pop, the leaf loop — gives the oracle's hits and counters (and, with -DBVH_HOST_WAVE64, the wave-level
// protocol: 64 fibers switching at the wave intrinsics). What it cannot show: anything that needs the hardware (LDS layout across
// lanes, occupancy, speed). tests/test_kernel_body_host.py drives it. Nothing here is shipped; the product runs this body on the device only.
//
// Built by the test with: g++ -std=c++20 -O1 -mavx2 -mfma -ffp-contract=off -fno-strict-aliasing -shared -fPIC.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "../../include/bvh_amd.h"                               // bvh_hit3f / bvh_hit3d, bvh_amd_counters, BVH_AMD_INVALID

// ---- single-lane stand-ins for what hip_runtime.h provides ------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __shared__ static

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline double2 make_double2(double x, double y) { return {x, y}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
inline uint32_t __float_as_uint(float x) { return __builtin_bit_cast(uint32_t, x); }
inline float __uint_as_float(uint32_t x) { return __builtin_bit_cast(float, x); }
inline long long __double_as_longlong(double x) { return __builtin_bit_cast(long long, x); }
inline double __longlong_as_double(long long x) { return __builtin_bit_cast(double, x); }
#if !defined(BVH_HOST_WAVE64)
// one lane: lane 0 of a wavefront whose other 63 lanes hold nothing
inline uint64_t __ballot(bool p) { return p ? 1ull : 0ull; }
inline int __popcll(uint64_t m) { return __builtin_popcountll(m); }
template <typename V> inline V __shfl(V v, int) { return v; }
template <typename V> inline V __shfl_down(V, int) { return V(0); }
template <typename V> inline V atomicAdd(V* p, V v) { V old = *p; *p = old + v; return old; }
template <typename V> inline V atomicOr(V* p, V v) { V old = *p; *p = old | v; return old; }
static const struct { unsigned x; } threadIdx = {0};
static struct { unsigned x; } blockIdx = {0}, gridDim = {1};     // trace_body_host_set_grid: the emulated blocks run one after another
#else
// BVH_HOST_WAVE64: one wavefront of 64 lanes = 64 fibers (ucontext) run round-robin by on_all_lanes(), switching at the wave
// intrinsics — every one of them sits in wave-uniform control flow in trace_body.inc, so after one round all lanes stand at the
// same intrinsic. Real refill / leaf-parking thresholds. Single OS thread: the "atomics" need no atomicity.
#include <ucontext.h>
static struct { unsigned x; } threadIdx = {0};                   // set by the scheduler before a lane resumes
static struct { unsigned x; } blockIdx = {0}, gridDim = {1};     // trace_body_host_set_grid: the emulated blocks run one after another
static ucontext_t g_main, g_lane[64];
static uint64_t g_slot[2][64];                                   // double buffered: a lane may run ahead to the next intrinsic
static unsigned g_phase[64];
inline void lane_yield() { const unsigned t = threadIdx.x; swapcontext(&g_lane[t], &g_main); threadIdx.x = t; }
inline int __popcll(uint64_t m) { return __builtin_popcountll(m); }
inline const uint64_t* wave_exchange(uint64_t mine) {
    const unsigned t = threadIdx.x, ph = g_phase[t]++ & 1u;
    g_slot[ph][t] = mine;
    lane_yield();                                                // ... until every lane has deposited its value
    return g_slot[ph];
}
inline uint64_t __ballot(bool p) {
    const uint64_t* s = wave_exchange(p ? 1 : 0);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) m |= s[i] << i;
    return m;
}
template <typename V> inline V wave_read(V v, int src, bool valid) {
    static_assert(sizeof(V) <= 8);
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(V));
    const uint64_t* s = wave_exchange(bits);
    V out = v;
    if (valid) std::memcpy(&out, &s[src], sizeof(V));
    return out;
}
template <typename V> inline V __shfl(V v, int src) { return wave_read(v, src, true); }
template <typename V> inline V __shfl_down(V v, int off) { const int src = int(threadIdx.x) + off; return wave_read(v, src & 63, src < 64); }
template <typename V> inline V atomicAdd(V* p, V v) { V old = *p; *p = old + v; return old; }
template <typename V> inline V atomicOr(V* p, V v) { V old = *p; *p = old | v; return old; }
#endif
using std::min;

// ---- stand-ins for bvh_amd/csrc/common.h (which needs the HIP headers) ------------------------------------------------------
namespace bvh_amd {
constexpr unsigned kCountBits = 4;
constexpr uint32_t kCountMask = 15u;
constexpr int kWave = 64;
template <typename T> struct PairNode;
template <> struct PairNode<float> { float lb[6], rb[6]; uint32_t li, ri; uint32_t pad[2]; };
template <> struct PairNode<double> { double lb[6], rb[6]; uint32_t li, ri; uint32_t pad[6]; };
template <typename T> struct HitOf;
template <> struct HitOf<float> { using Type = bvh_hit3f; };
template <> struct HitOf<double> { using Type = bvh_hit3d; };
enum { LEAF_TRIANGLE = 0, LEAF_SPHERE = 1 };
} // namespace bvh_amd

#include "../../bvh_amd/csrc/trace_device.h"

#if defined(BVH_HOST_WAVE64)
#include <functional>
#include <vector>
constexpr int kHostRefill = 54, kHostLeaf = 8;                   // traverse.hip: kRefillThreshold, kLeafThreshold
static std::function<void()>* g_lane_fn = nullptr;
static bool g_lane_done[64];
static void lane_entry() { (*g_lane_fn)(); g_lane_done[threadIdx.x] = true; swapcontext(&g_lane[threadIdx.x], &g_main); }
template <typename F> void on_all_lanes(F f) {
    std::function<void()> fn = f;
    g_lane_fn = &fn;
    constexpr size_t kStack = 1 << 20;
    std::vector<std::vector<char>> stacks(64, std::vector<char>(kStack));
    for (unsigned t = 0; t < 64; ++t) {
        getcontext(&g_lane[t]);
        g_lane[t].uc_stack.ss_sp = stacks[t].data();
        g_lane[t].uc_stack.ss_size = kStack;
        g_lane[t].uc_link = &g_main;
        makecontext(&g_lane[t], lane_entry, 0);
        g_lane_done[t] = false; g_phase[t] = 0;
    }
    for (;;) {                                                   // one round = every live lane runs to its next wave intrinsic
        unsigned live = 0;
        for (unsigned t = 0; t < 64; ++t) {
            if (g_lane_done[t]) continue;
            threadIdx.x = t;
            swapcontext(&g_main, &g_lane[t]);
            live += g_lane_done[t] ? 0 : 1;
        }
        if (live == 0) break;
    }
}
#else
constexpr int kHostRefill = 1, kHostLeaf = 1;                    // one lane: refill as soon as it is idle, leaf code as soon as it waits
template <typename F> void on_all_lanes(F f) { f(); }
#endif

namespace bvh_amd {
namespace {

template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D, bool Deep>
void host_trace(TraceArgs<T> a) {
#include "../../bvh_amd/csrc/trace_body.inc"
}

#if defined(BVH_HOST_WAVE64)
// The body with the quad-cooperative record fetch (trace_kernel_coop / trace_kernel_coop_nd: every record family, no deep stack), 64
// fibers: coop_load_pair runs as the device's own text, its two quad primitives emulated by meaning (trace_device.h).
template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D>
void host_trace_coop(TraceArgs<T> a) {
    constexpr bool Deep = false;
#undef BVH_TRACE_COOP
#define BVH_TRACE_COOP true
#include "../../bvh_amd/csrc/trace_body.inc"
#undef BVH_TRACE_COOP
}
#endif

} // namespace
} // namespace bvh_amd

namespace {

uint32_t g_parts = 1;                          // ticket ranges of the emulated launch (trace_body_host_set_parts)
uint32_t g_blocks = 1, g_stagger = 0;          // blocks of the emulated grid and its staggered drain (trace_body_host_set_grid)

// The blocks of the emulated grid, the LAST one first: a block of a higher drain class stops drawing tickets early and leaves what is
// left of its range to the blocks after it; block 0 (always class 0) never stops early and runs last, so every ticket must be drawn.
template <typename F> void on_all_blocks(F&& run_block) {
    gridDim.x = g_blocks;
    for (uint32_t b = g_blocks; b-- > 0;) { blockIdx.x = b; run_block(); }
    blockIdx.x = 0; gridDim.x = 1;
}

template <typename T, int Leaf, int D, bool Deep>
void run_variant(const bvh_amd::TraceArgs<T>& a, int any, int robust) {
    using namespace bvh_amd;
    on_all_blocks([&] {
        on_all_lanes([&] {
            if (any) { if (robust) host_trace<T, true, true, Leaf, true, D, Deep>(a); else host_trace<T, true, false, Leaf, true, D, Deep>(a); }
            else { if (robust) host_trace<T, false, true, Leaf, true, D, Deep>(a); else host_trace<T, false, false, Leaf, true, D, Deep>(a); }
        });
    });
}

template <typename T>
int run_any(const void* pairs, uint32_t root_index, const void* prims, const void* rays, size_t n_rays, int dim, int leaf, int any, int robust,
            uint32_t* deep, uint32_t deep_cap, void* hits, unsigned long long* counters3) {
    using namespace bvh_amd;
    unsigned long long work[8 * kTicketStride] = {};
    bvh_amd_counters cnt = {0, 0, 0};
    TraceArgs<T> a;
    a.pairs = static_cast<const PairNode<T>*>(pairs);
    a.prims = static_cast<const T*>(prims); a.rays = static_cast<const T*>(rays);
    a.hits = static_cast<typename HitOf<T>::Type*>(hits);
    a.n = n_rays; a.work = work; a.parts = g_parts; a.part_size = g_parts > 1 ? ((n_rays + g_parts - 1) / g_parts + 63) / 64 * 64 : n_rays; a.counters = &cnt; a.order = nullptr; a.deep = deep; a.deep_cap = deep_cap;
    a.root_index = root_index;
    a.refill_threshold = kHostRefill; a.leaf_threshold = kHostLeaf;
    a.coop = 0; a.prim_stride = 12; a.stream_hints = 0; a.stagger = g_stagger; a.one_shot = 0; a.wave_times = nullptr;
    if (dim == 2) { if (deep) run_variant<T, LEAF_SPHERE, 2, true>(a, any, robust); else run_variant<T, LEAF_SPHERE, 2, false>(a, any, robust); }
    else if (leaf == LEAF_SPHERE) { if (deep) run_variant<T, LEAF_SPHERE, 3, true>(a, any, robust); else run_variant<T, LEAF_SPHERE, 3, false>(a, any, robust); }
    else { if (deep) run_variant<T, LEAF_TRIANGLE, 3, true>(a, any, robust); else run_variant<T, LEAF_TRIANGLE, 3, false>(a, any, robust); }
    counters3[0] = cnt.node_pairs; counters3[1] = cnt.prim_tests; counters3[2] = cnt.leaves;
    return 0;
}

} // namespace

#if defined(BVH_HOST_WAVE64)
namespace {
template <typename T, int Leaf, int D>
int run_coop(const void* pairs, uint32_t root_index, const void* prims, const void* rays, size_t n_rays, int any, int robust, int refill, int leaf_threshold,
             void* hits, unsigned long long* counters3) {
    using namespace bvh_amd;
    unsigned long long work[8 * kTicketStride] = {};
    bvh_amd_counters cnt = {0, 0, 0};
    TraceArgs<T> a;
    a.pairs = static_cast<const PairNode<T>*>(pairs);
    a.prims = static_cast<const T*>(prims); a.rays = static_cast<const T*>(rays); a.hits = static_cast<typename HitOf<T>::Type*>(hits);
    a.n = n_rays; a.work = work; a.parts = g_parts; a.part_size = g_parts > 1 ? ((n_rays + g_parts - 1) / g_parts + 63) / 64 * 64 : n_rays;
    a.counters = &cnt; a.order = nullptr; a.deep = nullptr; a.deep_cap = 0; a.root_index = root_index;
    a.refill_threshold = refill; a.leaf_threshold = leaf_threshold;
    a.coop = 1; a.prim_stride = Leaf == LEAF_SPHERE ? 4 : 12; a.stream_hints = 0; a.stagger = 0; a.one_shot = 0; a.wave_times = nullptr;
    on_all_lanes([&] {
        if (any) { if (robust) host_trace_coop<T, true, true, Leaf, true, D>(a); else host_trace_coop<T, true, false, Leaf, true, D>(a); }
        else { if (robust) host_trace_coop<T, false, true, Leaf, true, D>(a); else host_trace_coop<T, false, false, Leaf, true, D>(a); }
    });
    counters3[0] = cnt.node_pairs; counters3[1] = cnt.prim_tests; counters3[2] = cnt.leaves;
    return 0;
}
} // namespace

// The quad-cooperative kernels' body as a full wavefront, thresholds as given (the device runs it with 12 / 12 and 20 / 20): float /
// double, dim 3 (triangles leaf = 0, spheres leaf = 1) or dim 2 (circles). Returns 0.
extern "C" int trace_body_host_coop(int is_double, const void* pairs, uint32_t root_index, const void* prims, const void* rays, size_t n_rays, int dim, int leaf,
                                    int any, int robust, int refill, int leaf_threshold, void* hits, unsigned long long* counters3) {
    using namespace bvh_amd;
    if (is_double) {
        if (dim == 2) return run_coop<double, LEAF_SPHERE, 2>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
        if (leaf == LEAF_SPHERE) return run_coop<double, LEAF_SPHERE, 3>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
        return run_coop<double, LEAF_TRIANGLE, 3>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
    }
    if (dim == 2) return run_coop<float, LEAF_SPHERE, 2>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
    if (leaf == LEAF_SPHERE) return run_coop<float, LEAF_SPHERE, 3>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
    return run_coop<float, LEAF_TRIANGLE, 3>(pairs, root_index, prims, rays, n_rays, any, robust, refill, leaf_threshold, hits, counters3);
}
#endif

extern "C" {

// Number of ticket ranges (1..8) the following emulated launches cut their rays into.
void trace_body_host_set_parts(int parts) { g_parts = parts < 1 ? 1u : parts > 8 ? 8u : static_cast<uint32_t>(parts); }

// Blocks of the emulated grid (one wavefront each, run one after another, the last block first) and the staggered drain's tickets per
// class and range (TraceArgs::stagger; 0 = off) of the following trace_body_host_any launches.
void trace_body_host_set_grid(int blocks, int stagger) { g_blocks = blocks < 1 ? 1u : static_cast<uint32_t>(blocks); g_stagger = stagger < 0 ? 0u : static_cast<uint32_t>(stagger); }

// The float / triangle / 3D body with counters on (`unused` keeps the historical signature). Returns 0.
int trace_body_host(const void* pairs64, const void* unused, uint32_t root_index, const float* tris12, const float* rays8, size_t n_rays,
                    int any, int robust, void* hits16, unsigned long long* counters3) {
    (void)unused;
    return run_any<float>(pairs64, root_index, tris12, rays8, n_rays, 3, bvh_amd::LEAF_TRIANGLE, any, robust, nullptr, 0, hits16, counters3);
}

// Every PairNode variant: is_double, dim 2 / 3, leaf 0 = triangles (12 values) / 1 = spheres (4 values; circles of 3 in 2D), deep = a
// spill buffer of deep_cap words for stack entries beyond 64 (or NULL). pairs: 64-byte (float) / 128-byte (double) records.
int trace_body_host_any(int is_double, const void* pairs, uint32_t root_index, const void* prims, const void* rays, size_t n_rays, int dim, int leaf,
                        int any, int robust, uint32_t* deep, uint32_t deep_cap, void* hits, unsigned long long* counters3) {
    if (is_double) return run_any<double>(pairs, root_index, prims, rays, n_rays, dim, leaf, any, robust, deep, deep_cap, hits, counters3);
    return run_any<float>(pairs, root_index, prims, rays, n_rays, dim, leaf, any, robust, deep, deep_cap, hits, counters3);
}

} // extern "C"
