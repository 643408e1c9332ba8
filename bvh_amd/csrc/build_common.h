// Shared pieces of the device builders (binned, sweep, mini-tree): order-preserving float keys, the
// reference's box arithmetic, the Phase A tree node, libstdc++ heap replay, and Phase C (numbering + emission).
// See build_binned.hip for the three-phase scheme.
#pragma once

#include "common.h"

#include <algorithm>
#include <cfloat>

namespace bvh_amd {

template <typename T> int relayout_on_device(BvhImpl<T>& b, const HostNode<T>* d_nodes, hipStream_t stream);

namespace bld {

constexpr int kSmall = 64;              // Phase B threshold: one primitive per lane
constexpr int kChunk = 2048;            // primitives per block in Phase A passes
constexpr int kBins = 8;                // binned_sah_builder.h:19
constexpr uint32_t kNone = 0xFFFFFFFFu;

enum : uint32_t { KIND_BIG = 0, KIND_INNER = 1, KIND_SMALL = 2 };
enum : uint32_t { MODE_PARTITION = 0, MODE_FALLBACK = 1 };

// ---- order-preserving integer image of a float (for atomic min/max) ---------------------------------
template <typename T> struct Ord;
template <> struct Ord<float> {
    using U = uint32_t;
    static constexpr float kMax = FLT_MAX;
    __device__ static U enc(float f) { U u = __float_as_uint(f); return u ^ (static_cast<U>(static_cast<int32_t>(u) >> 31) | 0x80000000u); }
    __device__ static float dec(U k) { U u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k; return __uint_as_float(u); }
    __device__ static float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    __device__ static uint32_t sign(float f) { return __float_as_uint(f) >> 31; }
    __device__ static float zero(uint32_t negative) { return __uint_as_float(negative << 31); }
};
template <> struct Ord<double> {
    using U = unsigned long long;
    static constexpr double kMax = DBL_MAX;
    __device__ static U enc(double f) {
        U u = static_cast<U>(__double_as_longlong(f));
        return u ^ (static_cast<U>(static_cast<long long>(u) >> 63) | 0x8000000000000000ull);
    }
    __device__ static double dec(U k) {
        U u = (k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k;
        return __longlong_as_double(static_cast<long long>(u));
    }
    __device__ static double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ static uint32_t sign(double f) { return static_cast<uint32_t>(static_cast<U>(__double_as_longlong(f)) >> 63); }
    __device__ static double zero(uint32_t negative) { return __longlong_as_double(static_cast<long long>(static_cast<U>(negative) << 63)); }
};

template <typename T> __device__ inline T pick_min(T a, T b) { return a < b ? a : b; }    // utils.h:41-43
template <typename T> __device__ inline T pick_max(T a, T b) { return a > b ? a : b; }

template <typename T> __device__ inline T half_area(const T* lo, const T* hi) {            // bbox.h:32-38
    T d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    return (d0 + d1) * d2 + d0 * d1;
}
template <typename T> __device__ inline int widest_axis(const T* lo, const T* hi) {        // vec.h:23-33
    T d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    int axis = 0;
    if (d[1] > d[axis]) axis = 1;
    if (d[2] > d[axis]) axis = 2;
    return axis;
}
// min(BinCount - 1, size_t(max(pos, 0)))  (binned_sah_builder.h:94-95); NaN -> 0, +inf saturates to 7.
template <typename T> __device__ inline uint32_t bin_of(T pos) {
    T v = pick_max(pos, T(0));
    return v >= T(kBins - 1) ? uint32_t(kBins - 1) : static_cast<uint32_t>(v);
}

// ---- data structures ---------------------------------------------------------------------------------
template <typename T>
struct ANode {                           // Phase A tree node (BFS allocation order)
    T lo[3], hi[3];
    uint32_t begin, end;
    uint32_t child;                      // A id of the first child (KIND_INNER)
    uint32_t parent;                     // parent A id | which << 31; root: kNone
    uint32_t kind;
    uint32_t ic;                         // inner nodes in this subtree (incl. itself)
    uint32_t rank;                       // pre-order rank among inner nodes, in the reference's processing order
    uint32_t pad;
};

template <typename T>
struct SlotBins {                        // 3 axes x 8 bins of {box, count}
    typename Ord<T>::U lo[3][kBins][3];
    typename Ord<T>::U hi[3][kBins][3];
    uint32_t cnt[3][kBins];
};

template <typename T>
struct SlotState {
    T split_pos;
    typename Ord<T>::U clo[2][3], chi[2][3];   // child boxes: [0] = positions < split, [1] = the rest
    // robust_min/max return their SECOND argument on equality (utils.h:41-43), so when a bound is zero its
    // sign is that of the LAST zero in position order. (position << 1 | sign) of the last zero-valued
    // contribution per component; consulted only when the decoded bound compares equal to zero.
    uint32_t zlo[2][3], zhi[2][3];
    uint32_t axis, wide, mode;
    uint32_t m;                          // #primitives satisfying the partition predicate
    uint32_t nviol;                      // #misplaced pairs (Hoare swaps)
    uint32_t split;                      // absolute split index
    uint32_t node, task0, ntasks;
};

struct Task { uint32_t slot, begin, end; };

struct Counters {
    uint32_t n_nodes, n_active_next, n_tasks_next, n_small, error, pad[3];
};

template <typename T>
struct BuildCtx {
    const T* bboxes;                     // n x {min xyz, max xyz}
    const T* centers;                    // n x 3
    uint32_t* ids;
    uint32_t n;
    uint32_t min_leaf, max_leaf;
    ANode<T>* nodes;
    uint32_t node_cap;
    SlotBins<T>* bins;
    SlotState<T>* state;
    SlotState<T>* state_next;
    uint32_t slot_cap;
    Task* tasks;
    Task* tasks_next;
    uint32_t task_cap;
    uint32_t* chunk_true;
    uint32_t* ltab;
    uint32_t* rtab;
    uint32_t* small_list;
    HostNode<T>* stage;                  // 2n staged nodes of the Phase B subtrees
    Counters* counters;
};

// ---- libstdc++ std::partial_sort, replayed by one lane (stl_heap.h / stl_algo.h:1912-1919)
// Operates on `ids[0..)` with keys key(id); comp(a, b) = key(a) < key(b).
template <typename KeyFn>
__device__ void heap_push(uint32_t* a, long hole, long top, uint32_t value, KeyFn key) {
    long parent = (hole - 1) / 2;
    while (hole > top && key(a[parent]) < key(value)) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
}
template <typename KeyFn>
__device__ void heap_adjust(uint32_t* a, long hole, long len, uint32_t value, KeyFn key) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (key(a[second]) < key(a[second - 1])) second--;
        a[hole] = a[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        a[hole] = a[second - 1];
        hole = second - 1;
    }
    heap_push(a, hole, top, value, key);
}
template <typename KeyFn>
__device__ void partial_sort_replay(uint32_t* a, long middle, long last, KeyFn key) {
    if (middle >= 2) {                                       // __make_heap(first, middle)
        long parent = (middle - 2) / 2;
        for (;;) {
            uint32_t v = a[parent];
            heap_adjust(a, parent, middle, v, key);
            if (parent == 0) break;
            parent--;
        }
    }
    for (long i = middle; i < last; ++i) {                   // __heap_select
        if (key(a[i]) < key(a[0])) {                         // __pop_heap(first, middle, i)
            uint32_t v = a[i];
            a[i] = a[0];
            heap_adjust(a, 0, middle, v, key);
        }
    }
    for (long end = middle; end > 1;) {                      // __sort_heap(first, middle)
        --end;
        uint32_t v = a[end];
        a[end] = a[0];
        heap_adjust(a, 0, end, v, key);
    }
}

// ---- SAH candidate sweep over one axis (binned_sah_builder.h:101-116) -----------------------------------
// Starts from (FLT_MAX, -) and reports the first strict minimum; combining axes 0,1,2 with strict `<`
// afterwards equals the reference's carried `best_split`.
template <typename T, typename LoadBin>
__device__ inline void sweep_axis(LoadBin load, T& best_cost, uint32_t& best_bin) {
    T right_cost[kBins];
    {
        T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
        uint32_t cnt = 0;
#pragma unroll
        for (int i = kBins - 1; i > 0; --i) {
            T blo[3], bhi[3]; uint32_t bc;
            load(i, blo, bhi, bc);
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = pick_min(lo[k], blo[k]); hi[k] = pick_max(hi[k], bhi[k]); }
            cnt += bc;
            right_cost[i] = half_area(lo, hi) * static_cast<T>(cnt);
        }
    }
    best_cost = Ord<T>::kMax;
    best_bin = kBins / 2;
    T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < kBins - 1; ++i) {
        T blo[3], bhi[3]; uint32_t bc;
        load(i, blo, bhi, bc);
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = pick_min(lo[k], blo[k]); hi[k] = pick_max(hi[k], bhi[k]); }
        cnt += bc;
        T cost = half_area(lo, hi) * static_cast<T>(cnt) + right_cost[i + 1];
        if (cost < best_cost) { best_cost = cost; best_bin = i + 1; }
    }
}

template <typename T>
__device__ inline T decode_bound(typename Ord<T>::U key, uint32_t ztrack) {
    T v = Ord<T>::dec(key);
    if (v == T(0)) v = Ord<T>::zero(ztrack & 1u);
    return v;
}
template <typename T>
__device__ inline void track_zero(uint32_t* slot, T value, uint32_t pos) {
    if (value == T(0)) atomicMax(slot, (pos << 1) | Ord<T>::sign(value));
}

__device__ inline uint32_t pack_item(uint32_t node, uint32_t b, uint32_t e) { return node | (b << 8) | (e << 16); }

// Lanes of one wavefront communicate through LDS without a workgroup barrier (waves of a block run
// independent subtrees). LDS operations of a wave execute in order; this only stops the compiler from
// moving memory operations across the hand-off.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================================
// Phase C: numbering and emission
// =====================================================================================================

template <typename T>
__global__ void __launch_bounds__(256) k_count_inner(BuildCtx<T> c, uint32_t first, uint32_t last) {
    const uint32_t v = first + blockIdx.x * 256 + threadIdx.x;
    if (v >= last) return;
    ANode<T>& nd = c.nodes[v];
    if (nd.kind == KIND_INNER) nd.ic = 1 + c.nodes[nd.child].ic + c.nodes[nd.child + 1].ic;
}

template <typename T>
__global__ void __launch_bounds__(256) k_assign_ranks(BuildCtx<T> c, uint32_t first, uint32_t last) {
    const uint32_t v = first + blockIdx.x * 256 + threadIdx.x;
    if (v >= last) return;
    const ANode<T>& nd = c.nodes[v];
    if (nd.kind != KIND_INNER) return;
    ANode<T>& c0 = c.nodes[nd.child];
    ANode<T>& c1 = c.nodes[nd.child + 1];
    // the item with fewer primitives is popped first; on a tie the second child (top_down_sah_builder.h:116-121)
    const bool c0_first = (c0.end - c0.begin) < (c1.end - c1.begin);
    if (c0_first) { c0.rank = nd.rank + 1; c1.rank = nd.rank + 1 + c0.ic; }
    else          { c1.rank = nd.rank + 1; c0.rank = nd.rank + 1 + c1.ic; }
}

template <typename T>
__device__ inline uint32_t final_id(const BuildCtx<T>& c, const ANode<T>& nd) {
    if (nd.parent == kNone) return 0;
    return 1 + 2 * c.nodes[nd.parent & 0x7FFFFFFFu].rank + (nd.parent >> 31);
}

template <typename T>
__global__ void __launch_bounds__(256) k_emit_tree(BuildCtx<T> c, uint32_t n_nodes, HostNode<T>* out) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n_nodes) return;
    const ANode<T>& nd = c.nodes[v];
    using I = typename IndexOf<T>::Type;
    HostNode<T> rec;
    for (int k = 0; k < 3; ++k) { rec.bounds[2 * k] = nd.lo[k]; rec.bounds[2 * k + 1] = nd.hi[k]; }
    if (nd.kind == KIND_INNER) {
        rec.index = static_cast<I>(1 + 2 * nd.rank) << kCountBits;
    } else {                                                  // KIND_SMALL: the staged subtree root
        const HostNode<T>& sr = c.stage[2ull * nd.begin];
        rec.index = (sr.index & kCountMask) ? sr.index : ((sr.index >> kCountBits) + 2 * I(nd.rank)) << kCountBits;
    }
    out[final_id(c, nd)] = rec;
}

template <typename T>
__global__ void __launch_bounds__(256) k_emit_small(BuildCtx<T> c, uint32_t n_small, HostNode<T>* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    const ANode<T>& nd = c.nodes[c.small_list[w]];
    using I = typename IndexOf<T>::Type;
    const uint32_t count = 2 * nd.ic;                         // staged nodes besides the root
    const HostNode<T>* stage = c.stage + 2ull * nd.begin;
    for (uint32_t j = 1 + lane; j <= count; j += 64) {
        HostNode<T> rec = stage[j];
        if ((rec.index & kCountMask) == 0) rec.index = ((rec.index >> kCountBits) + 2 * I(nd.rank)) << kCountBits;
        out[2ull * nd.rank + j] = rec;
    }
}

template <typename T> struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)); }
};


} // namespace bld
} // namespace bvh_amd
