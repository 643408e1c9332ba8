// Shared pieces of the device builders (binned, sweep, mini-tree): order-preserving float keys, the
// reference's box arithmetic, the Phase A tree node, libstdc++ heap replay, and Phase C (numbering + emission).
// See build_binned.hip for the three-phase scheme.
#pragma once

#include "common.h"

#include <algorithm>
#include <cfloat>
#include <functional>
#include <vector>

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

// Block-wide min / max of per-thread bounds into six LDS words: each wave folds its 64 values with xor-shuffles first and ONE lane
// per wave touches the LDS word instead of 256 threads contending for six LDS addresses. Keys are the order-preserving integer
// images (Ord<T>::enc), so integer min / max is the float min / max.
template <typename U> __device__ inline U wave_min_key(U v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const U o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
template <typename U> __device__ inline U wave_max_key(U v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const U o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

template <typename T> __device__ inline T pick_min(T a, T b) { return a < b ? a : b; }    // utils.h:41-43
template <typename T> __device__ inline T pick_max(T a, T b) { return a > b ? a : b; }

// 2D builds (`2f` / `2d` families, Node<T, 2>) run the same kernels on three-wide data with z = 0 everywhere: the z lanes of
// every box stay (+0, +0) and are inert; only the DECISIONS know the dimension: the half area (bbox.h:32-38: d0 + d1 in
// 2D), the widest axis, and which axes are candidates for a split.
// SplitHeuristic (split_heuristic.h:17-38): a range of `size` primitives counts as ceil(size / 2^log_cluster) clusters
// (get_prim_count, :25-27), as the scalar the costs are computed in; the defaults {0, 1} make it the plain count.
template <typename T> __device__ inline T sah_prims(uint32_t size, uint32_t log_cluster) {
    return static_cast<T>((static_cast<unsigned long long>(size) + ((1ull << log_cluster) - 1ull)) >> log_cluster);
}

template <typename T> __device__ inline T half_area(const T* lo, const T* hi, int dim = 3) {   // bbox.h:32-38
    T d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    if (dim == 2) return d0 + d1;
    return (d0 + d1) * d2 + d0 * d1;
}
template <typename T> __device__ inline int widest_axis(const T* lo, const T* hi, int dim = 3) {   // vec.h:23-33
    T d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    int axis = 0;
    if (d[1] > d[axis]) axis = 1;
    if (dim > 2 && d[2] > d[axis]) axis = 2;
    return axis;
}
// min(BinCount - 1, size_t(max(pos, 0)))  (binned_sah_builder.h:94-95); NaN -> 0, +inf saturates to 7.
template <int NB, typename T> __device__ inline uint32_t bin_of_n(T pos) {
    T v = pick_max(pos, T(0));
    return v >= T(NB - 1) ? uint32_t(NB - 1) : static_cast<uint32_t>(v);
}
template <typename T> __device__ inline uint32_t bin_of(T pos) { return bin_of_n<kBins>(pos); }

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
    uint32_t tree;                       // which tree of a forest build (mini-trees); 0 for single-tree builds
};

// BinCount (binned_sah_builder.h:18) is a template parameter of the reference's builder: NB = 8 is its default and what DefaultBuilder
// and the mini-tree builder instantiate; 4 / 16 / 32 are served by the plain Phase A + node-by-node Phase B kernels (build_binned.hip).
template <typename T, int NB = kBins>
struct SlotBins {                        // 3 axes x NB bins of {box, count}
    typename Ord<T>::U lo[3][NB][3];
    typename Ord<T>::U hi[3][NB][3];
    uint32_t cnt[3][NB];
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
    uint32_t n_nodes, n_active_next, n_tasks_next, n_big_next, n_small, error, pad[2];   // (the three *_next words are reset together per level)
    uint32_t n_medium[4];                // segments handed to k_medium, by size class (<= 256 / 512 / 1024 / 2048 primitives)
};

// Round 4: segments of 65 .. medium_cap primitives leave the level-synchronous Phase A and are split down to <= 64-primitive subtrees by
// ONE BLOCK each, primitives resident in LDS (build_binned.hip: k_medium). The nodes such a block creates are appended to c.nodes in
// its own level order; MedInfo keeps what the numbering (Phase C) needs to walk them level by level.
constexpr int kMedLevels = 40;
struct MedInfo {
    uint32_t base;                       // A id of local node 1 (local node 0 is the segment's own node)
    uint32_t count;                      // local nodes, the segment's own node included
    uint32_t n_levels;
    uint16_t level_start[kMedLevels + 1];   // local ids: level l = [level_start[l], level_start[l + 1])
};

template <typename T>
struct BuildCtx {
    const T* bboxes;                     // n x {min xyz, max xyz}
    const T* centers;                    // n x 3
    uint32_t* ids;
    uint32_t n;
    uint32_t min_leaf, max_leaf;
    int dim = 3;                         // 2: Node<T, 2> semantics on z = 0 data (see half_area)
    uint32_t sah_log = 0;                // SplitHeuristic: log2 of the cluster size ...
    T sah_ratio = T(1);                  // ... and the node / primitive cost ratio (split_heuristic.h:17-23)
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
    // forest builds (mini-trees): every tree is emitted standalone (local node ids, local leaf first_id) at
    // tree_node_off[tree]; tree_begin[tree] = first position of the tree's primitives. Null for single trees.
    const uint32_t* tree_node_off = nullptr;
    const uint32_t* tree_begin = nullptr;
    // segments of at most medium_cap primitives (0: none) are finished by k_medium instead of further Phase A levels
    uint32_t big_threshold = 0;              // > 0: segments of more primitives are counted in counters->n_big_next (sweep builder: multi-block scans)
    uint32_t medium_cap = 0;
    uint32_t medium_slots = 0;               // entries per size class in medium_list / med_info
    uint32_t medium_min_class = 0;           // segments of more than 256 primitives are listed under at least this size class
    uint32_t med_cum[4] = {0, 0, 0, 0};      // a launch over several class lists at once: running block counts (all 0: one list, already offset)
    uint32_t* medium_list = nullptr;
    MedInfo* med_info = nullptr;
    unsigned long long* med_prof = nullptr;  // developer knob BVH_AMD_MED_PROF=1: reference-clock ticks per phase of k_medium, summed over blocks
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
template <typename T, int NB, typename LoadBin>
__device__ inline void sweep_axis_n(LoadBin load, T& best_cost, uint32_t& best_bin, int dim, uint32_t sah_log) {
    constexpr int kBins = NB;            // (shadows the default: BinCount of this instantiation)
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
            right_cost[i] = half_area(lo, hi, dim) * sah_prims<T>(cnt, sah_log);
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
        T cost = half_area(lo, hi, dim) * sah_prims<T>(cnt, sah_log) + right_cost[i + 1];
        if (cost < best_cost) { best_cost = cost; best_bin = i + 1; }
    }
}

template <typename T, typename LoadBin>
__device__ inline void sweep_axis(LoadBin load, T& best_cost, uint32_t& best_bin, int dim, uint32_t sah_log) {
    sweep_axis_n<T, kBins>(load, best_cost, best_bin, dim, sah_log);
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
// Hand-off between lanes of DIFFERENT workgroups through an arrival ticket (the bottom-up climbs of refit / subtree counts): the data
// is written and read with agent-scope relaxed atomics (sc1 accesses: coherent across the XCDs' L2s by themselves), the ticket is an
// agent-scope atomic add, and between them only the wave's OWN accesses have to be complete — a wait, not cache maintenance.
// __threadfence() (agent-scope fence) additionally writes back and invalidates the XCD's whole L2 on gfx950, per call: two per climbed
// node made a 1.9M-node climb cost 6 ms instead of 0.1 (measured round 4, profiles/r04_build_extract_ab.txt). Every location
// exchanged this way must be accessed with agent-scope atomics on both sides.
// The wait is written out (round 5): a workgroup-scope release fence emits NO s_waitcnt on gfx950, so the sc1 stores were
// ordered before the ticket only by an unrelated dependent load that happened to sit between them. `s_waitcnt vmcnt(0)` = every
// store / load this wave has issued has been acknowledged at its scope (sc1 stores: past the XCD's L2); the fences around it keep
// the compiler from moving accesses across. tests/test_host_logic.py greps the ISA of the climbing kernels for the wait.
__device__ inline void ticket_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ inline void ticket_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

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
    const uint32_t off = c.tree_node_off ? c.tree_node_off[nd.tree] : 0;
    if (nd.parent == kNone) return off;
    return off + 1 + 2 * c.nodes[nd.parent & 0x7FFFFFFFu].rank + (nd.parent >> 31);
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
        const I rebase = static_cast<I>(c.tree_begin ? c.tree_begin[nd.tree] : 0) << kCountBits;
        rec.index = (sr.index & kCountMask) ? sr.index - rebase : ((sr.index >> kCountBits) + 2 * I(nd.rank)) << kCountBits;
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
    const size_t off = c.tree_node_off ? c.tree_node_off[nd.tree] : 0;
    const I rebase = static_cast<I>(c.tree_begin ? c.tree_begin[nd.tree] : 0) << kCountBits;
    for (uint32_t j = 1 + lane; j <= count; j += 64) {
        HostNode<T> rec = stage[j];
        if ((rec.index & kCountMask) == 0) rec.index = ((rec.index >> kCountBits) + 2 * I(nd.rank)) << kCountBits;
        else rec.index -= rebase;
        out[off + 2ull * nd.rank + j] = rec;
    }
}

// Scratch buffer from the stream-ordered pool (common.h: scratch_alloc): allocated on, and released in the order of, the stream
// of the operation in progress on this thread. A buffer whose pointer is taken over by a BvhImpl (p = nullptr here) is later
// released with hipFree, which accepts pool memory.
// A forest caller may move Phase B (k_small_levels) to a stream of its own — the mini-tree builder runs it on a stream whose CU mask
// leaves a few CUs to the top-level build beside it: `start` is recorded on the build's stream by the caller, `done` is recorded
// behind Phase B and the build's stream waits for it.
struct PhaseB { hipStream_t stream = nullptr; hipEvent_t start = nullptr, done = nullptr; };

template <typename T> struct DevBuf {
    T* p = nullptr;
    ScratchTag tag;
    bool pooled = false;                       // mirrors tag.pooled (read by the callers that must synchronise before a plain hipFree)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { scratch_free(p, tag); }
    hipError_t alloc(size_t count) {
        scratch_free(p, tag);
        p = nullptr;
        const hipError_t e = scratch_alloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T), &tag);
        pooled = tag.pooled;
        return e;
    }
};


// ---- Phase A pieces shared by the binned and sweep builders -------------------------------------------

template <typename T>
__global__ void __launch_bounds__(256) k_init_root(BuildCtx<T> c, bool init_iota) {
    // ids = iota (binned_sah_builder.h:77) or the caller's order; root box = compute_bbox(0, n) in position order
    // (top_down_sah_builder.h:80, :133-139)
    __shared__ typename Ord<T>::U slo[3], shi[3];
    if (threadIdx.x < 3) { slo[threadIdx.x] = Ord<T>::enc(Ord<T>::kMax); shi[threadIdx.x] = Ord<T>::enc(-Ord<T>::kMax); }
    __syncthreads();
    T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
    for (size_t i = blockIdx.x * size_t{256} + threadIdx.x; i < c.n; i += size_t{gridDim.x} * 256) {
        size_t id = i;
        if (init_iota) c.ids[i] = static_cast<uint32_t>(i);
        else id = c.ids[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T a = c.bboxes[6 * id + k], b = c.bboxes[6 * id + 3 + k];
            lo[k] = pick_min(lo[k], a);
            hi[k] = pick_max(hi[k], b);
            track_zero(&c.state[0].zlo[0][k], a, static_cast<uint32_t>(i));
            track_zero(&c.state[0].zhi[0][k], b, static_cast<uint32_t>(i));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], klo); atomicMax(&shi[k], khi); }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        // the root's key box lives in state[0].clo[0]/chi[0] until k_make_root decodes it
        atomicMin(&c.state[0].clo[0][threadIdx.x], slo[threadIdx.x]);
        atomicMax(&c.state[0].chi[0][threadIdx.x], shi[threadIdx.x]);
    }
}

template <typename T>
__global__ void k_prepare_root(BuildCtx<T> c) {
    for (int k = 0; k < 3; ++k) {
        c.state[0].clo[0][k] = Ord<T>::enc(Ord<T>::kMax); c.state[0].chi[0][k] = Ord<T>::enc(-Ord<T>::kMax);
        c.state[0].zlo[0][k] = 0; c.state[0].zhi[0][k] = 0;
    }
    Counters z = {};
    *c.counters = z;
}

template <typename T>
__device__ void emit_child(const BuildCtx<T>& c, uint32_t id) {
    // registers a freshly created node either as a big segment of the next level or as a Phase B root
    ANode<T>& nd = c.nodes[id];
    const uint32_t size = nd.end - nd.begin;
    if (size > kSmall && size <= c.medium_cap) {              // one block finishes it in LDS (k_medium)
        nd.kind = KIND_BIG;
        uint32_t cls = size <= 256 ? 0u : size <= 512 ? 1u : size <= 1024 ? 2u : 3u;           // k_medium<T, 256 << cls>
        if (cls > 0 && cls < c.medium_min_class) cls = c.medium_min_class;
        const uint32_t m = atomicAdd(&c.counters->n_medium[cls], 1u);
        c.medium_list[cls * c.medium_slots + m] = id;
    } else if (size > kSmall) {
        nd.kind = KIND_BIG;
        if (c.big_threshold && size > c.big_threshold) atomicAdd(&c.counters->n_big_next, 1u);
        const uint32_t slot = atomicAdd(&c.counters->n_active_next, 1u);
        const uint32_t nt = (size + kChunk - 1) / kChunk;
        const uint32_t t0 = atomicAdd(&c.counters->n_tasks_next, nt);
        if (slot >= c.slot_cap || t0 + nt > c.task_cap) { atomicOr(&c.counters->error, 1u); return; }
        SlotState<T>& st = c.state_next[slot];
        st.node = id; st.task0 = t0; st.ntasks = nt;
        for (uint32_t t = 0; t < nt; ++t) {
            Task tk;
            tk.slot = slot;
            tk.begin = nd.begin + t * kChunk;
            tk.end = min(nd.end, tk.begin + kChunk);
            c.tasks_next[t0 + t] = tk;
        }
    } else {
        nd.kind = KIND_SMALL;
        const uint32_t s = atomicAdd(&c.counters->n_small, 1u);
        c.small_list[s] = id;
    }
}

template <typename T>
__global__ void k_make_root(BuildCtx<T> c) {
    ANode<T>& r = c.nodes[0];
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = decode_bound<T>(c.state[0].clo[0][k], c.state[0].zlo[0][k]);
        r.hi[k] = decode_bound<T>(c.state[0].chi[0][k], c.state[0].zhi[0][k]);
    }
    r.begin = 0; r.end = c.n; r.child = kNone; r.parent = kNone; r.ic = 0; r.rank = 0; r.tree = 0;
    c.counters->n_nodes = 1;
    emit_child(c, 0);
}

// One block per active slot: reset its accumulators.
template <typename T, int NB = kBins>
__global__ void __launch_bounds__(64) k_init_slots(BuildCtx<T> c) {
    const uint32_t slot = blockIdx.x;
    const auto lo0 = Ord<T>::enc(Ord<T>::kMax), hi0 = Ord<T>::enc(-Ord<T>::kMax);
    if (c.bins) {
        SlotBins<T, NB>& b = reinterpret_cast<SlotBins<T, NB>*>(c.bins)[slot];
        for (int w = threadIdx.x; w < 3 * NB * 3; w += 64) { (&b.lo[0][0][0])[w] = lo0; (&b.hi[0][0][0])[w] = hi0; }
        for (int w = threadIdx.x; w < 3 * NB; w += 64) (&b.cnt[0][0])[w] = 0;
    }
    SlotState<T>& st = c.state[slot];
    if (threadIdx.x < 6) {
        st.clo[threadIdx.x / 3][threadIdx.x % 3] = lo0; st.chi[threadIdx.x / 3][threadIdx.x % 3] = hi0;
        st.zlo[threadIdx.x / 3][threadIdx.x % 3] = 0;   st.zhi[threadIdx.x / 3][threadIdx.x % 3] = 0;
    }
    if (threadIdx.x == 0) { st.m = 0; st.nviol = 0; st.mode = MODE_PARTITION; st.split = 0; }
}

// compute_bbox of both children (top_down_sah_builder.h:96-97, :133-139).
template <typename T>
__global__ void __launch_bounds__(256) k_child_bounds(BuildCtx<T> c) {
    __shared__ typename Ord<T>::U slo[2][3], shi[2][3];
    const Task tk = c.tasks[blockIdx.x];
    SlotState<T>& st = c.state[tk.slot];
    if (threadIdx.x < 6) { slo[threadIdx.x / 3][threadIdx.x % 3] = Ord<T>::enc(Ord<T>::kMax); shi[threadIdx.x / 3][threadIdx.x % 3] = Ord<T>::enc(-Ord<T>::kMax); }
    __syncthreads();
    const uint32_t split = st.split;
    // the chunk lies on one side unless it straddles the split: reduce in registers per side first
    T lo[2][3], hi[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[s][k] = Ord<T>::kMax; hi[s][k] = -Ord<T>::kMax; }
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) {
        const uint32_t id = c.ids[pos];
        const bool right = pos >= split;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T a = c.bboxes[6ull * id + k], b = c.bboxes[6ull * id + 3 + k];
            track_zero(&st.zlo[right ? 1 : 0][k], a, pos);
            track_zero(&st.zhi[right ? 1 : 0][k], b, pos);
            if (right) { lo[1][k] = pick_min(lo[1][k], a); hi[1][k] = pick_max(hi[1][k], b); }
            else       { lo[0][k] = pick_min(lo[0][k], a); hi[0][k] = pick_max(hi[0][k], b); }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T a = lo[s][k], b = hi[s][k];
            for (int off = 32; off > 0; off >>= 1) { a = pick_min(a, __shfl_xor(a, off)); b = pick_max(b, __shfl_xor(b, off)); }
            if ((threadIdx.x & 63) == 0) { atomicMin(&slo[s][k], Ord<T>::enc(a)); atomicMax(&shi[s][k], Ord<T>::enc(b)); }
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int s = threadIdx.x / 3, k = threadIdx.x % 3;
        atomicMin(&st.clo[s][k], slo[s][k]);
        atomicMax(&st.chi[s][k], shi[s][k]);
    }
}

// Child creation with SATO order (top_down_sah_builder.h:91-113).
template <typename T>
__global__ void __launch_bounds__(64) k_finalize(BuildCtx<T> c, uint32_t n_active) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    const SlotState<T>& st = c.state[slot];
    ANode<T>& nd = c.nodes[st.node];
    T lo[2][3], hi[2][3];
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k) { lo[s][k] = decode_bound<T>(st.clo[s][k], st.zlo[s][k]); hi[s][k] = decode_bound<T>(st.chi[s][k], st.zhi[s][k]); }
    uint32_t rb[2] = { nd.begin, st.split }, re[2] = { st.split, nd.end };
    int first = 0;
    if (half_area(lo[0], hi[0], c.dim) < half_area(lo[1], hi[1], c.dim)) first = 1;        // :105-108
    const uint32_t child = atomicAdd(&c.counters->n_nodes, 2u);
    if (child + 2 > c.node_cap) { atomicOr(&c.counters->error, 2u); return; }
    nd.child = child;
    nd.kind = KIND_INNER;
    for (int w = 0; w < 2; ++w) {
        const int s = w == 0 ? first : 1 - first;
        ANode<T>& ch = c.nodes[child + w];
        for (int k = 0; k < 3; ++k) { ch.lo[k] = lo[s][k]; ch.hi[k] = hi[s][k]; }
        ch.begin = rb[s]; ch.end = re[s];
        ch.child = kNone; ch.parent = st.node | (uint32_t(w) << 31); ch.ic = 0; ch.rank = 0; ch.tree = nd.tree;
        emit_child(c, child + w);
    }
}


// ---- host side shared by the builders ----------------------------------------------------------------------

// Phase C: inner counts bottom-up, ranks top-down, then scatter of the Phase A nodes and the staged subtrees.
// the medium segments' own nodes (k_medium): inner counts bottom-up / ranks top-down inside every segment, one wave each
template <typename T>
__global__ void __launch_bounds__(64) k_medium_count(BuildCtx<T> c) {
    __shared__ uint32_t ic[256];
    __shared__ uint16_t ch[256];
    const int lane = threadIdx.x;
    const MedInfo mi = c.med_info[blockIdx.x];
    const uint32_t root = c.medium_list[blockIdx.x];
    if (mi.count == 0) return;                                // (the block gave up: the build is retried without k_medium)
    for (uint32_t t = lane; t < mi.count; t += 64) {
        const ANode<T>& nd = c.nodes[t == 0 ? root : mi.base + t - 1];
        const bool inner = nd.kind == KIND_INNER;
        ch[t] = inner ? static_cast<uint16_t>(nd.child - mi.base + 1) : uint16_t{0};
        ic[t] = inner ? 0u : nd.ic;                           // KIND_SMALL: Phase B has counted its subtree
    }
    wave_sync();
    for (uint32_t lv = mi.n_levels; lv-- > 0;) {
        for (uint32_t t = mi.level_start[lv] + lane; t < mi.level_start[lv + 1]; t += 64)
            if (ch[t]) ic[t] = 1 + ic[ch[t]] + ic[ch[t] + 1];
        wave_sync();
    }
    for (uint32_t t = lane; t < mi.count; t += 64)
        if (ch[t]) c.nodes[t == 0 ? root : mi.base + t - 1].ic = ic[t];
}
template <typename T>
__global__ void __launch_bounds__(64) k_medium_rank(BuildCtx<T> c) {
    __shared__ uint32_t ic[256], rank[256], size[256];
    __shared__ uint16_t ch[256];
    const int lane = threadIdx.x;
    const MedInfo mi = c.med_info[blockIdx.x];
    const uint32_t root = c.medium_list[blockIdx.x];
    if (mi.count == 0) return;
    for (uint32_t t = lane; t < mi.count; t += 64) {
        const ANode<T>& nd = c.nodes[t == 0 ? root : mi.base + t - 1];
        ch[t] = nd.kind == KIND_INNER ? static_cast<uint16_t>(nd.child - mi.base + 1) : uint16_t{0};
        ic[t] = nd.ic; size[t] = nd.end - nd.begin; rank[t] = nd.rank;     // (rank: only the segment's own node has one yet)
    }
    wave_sync();
    for (uint32_t lv = 0; lv < mi.n_levels; ++lv) {
        for (uint32_t t = mi.level_start[lv] + lane; t < mi.level_start[lv + 1]; t += 64) {
            const uint32_t a = ch[t];
            if (!a) continue;
            // the item with fewer primitives is popped first; on a tie the second child (top_down_sah_builder.h:116-121)
            if (size[a] < size[a + 1]) { rank[a] = rank[t] + 1; rank[a + 1] = rank[t] + 1 + ic[a]; }
            else                       { rank[a + 1] = rank[t] + 1; rank[a] = rank[t] + 1 + ic[a + 1]; }
        }
        wave_sync();
    }
    for (uint32_t t = 1 + lane; t < mi.count; t += 64) c.nodes[mi.base + t - 1].rank = rank[t];
}

template <typename T>
int number_nodes(const BuildCtx<T>& c, const std::vector<uint32_t>& level_start, hipStream_t stream, const uint32_t* n_medium = nullptr)
{
    const size_t levels = level_start.size() - 1;
    auto per_class = [&](auto kernel) {
        for (uint32_t cls = 0; n_medium && cls < 4; ++cls) {
            if (!n_medium[cls]) continue;
            BuildCtx<T> cc = c;
            cc.medium_list = c.medium_list + size_t{cls} * c.medium_slots;
            cc.med_info = c.med_info + size_t{cls} * c.medium_slots;
            hipLaunchKernelGGL(kernel, dim3(n_medium[cls]), dim3(64), 0, stream, cc);
        }
    };
    per_class(k_medium_count<T>);
    for (size_t l = levels; l-- > 0;) {
        const uint32_t a = level_start[l], b = level_start[l + 1];
        if (b > a) hipLaunchKernelGGL(k_count_inner<T>, dim3((b - a + 255) / 256), dim3(256), 0, stream, c, a, b);
    }
    for (size_t l = 0; l < levels; ++l) {
        const uint32_t a = level_start[l], b = level_start[l + 1];
        if (b > a) hipLaunchKernelGGL(k_assign_ranks<T>, dim3((b - a + 255) / 256), dim3(256), 0, stream, c, a, b);
    }
    per_class(k_medium_rank<T>);
    return BVH_AMD_OK;
}

template <typename T>
int number_and_emit(BvhImpl<T>& out, const BuildCtx<T>& c, const std::vector<uint32_t>& level_start, uint32_t n_nodes_a,
                    uint32_t n_small, DevBuf<HostNode<T>>& final_nodes, hipStream_t stream, const uint32_t* n_medium = nullptr)
{
    int rc = number_nodes<T>(c, level_start, stream, n_medium);
    if (rc) return rc;
    ANode<T> root;
    { int rb_ = readback(&root, c.nodes, sizeof(root), stream); if (rb_) return rb_; }
    const size_t total_nodes = 2 * size_t{root.ic} + 1;
    BVH_HIP_TRY(final_nodes.alloc(total_nodes), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_emit_tree<T>, dim3((n_nodes_a + 255) / 256), dim3(256), 0, stream, c, n_nodes_a, final_nodes.p);
    if (n_small) hipLaunchKernelGGL(k_emit_small<T>, dim3((n_small + 3) / 4), dim3(256), 0, stream, c, n_small, final_nodes.p);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    out.node_count = total_nodes;
    return BVH_AMD_OK;
}

// Results: the device copy (reference-layout nodes, pair records, prim ids). The host mirror is filled lazily
// (BvhImpl::sync_host). `d_ids` holds n prim ids in BVH order; with take_ids the buffer itself becomes out.d_prim_ids
// (the caller must then forget it). final_nodes is always taken over.
template <typename T>
int finish_build(BvhImpl<T>& out, DevBuf<HostNode<T>>& final_nodes, uint32_t* d_ids, size_t n, hipStream_t stream, bool take_ids) {
    out.nodes.clear();                                        // out.node_count was set by the caller
    out.prim_ids.clear();
    out.prim_count = n;
    out.host_valid = false;
    int rc = relayout_on_device(out, final_nodes.p, stream);
    if (rc) return rc;
    if (!out.d_work) BVH_HIP_TRY(hipMalloc(&out.d_work, size_t{BvhImpl<T>::kWorkSlots} * BvhImpl<T>::kWorkStride * sizeof(unsigned long long)), BVH_AMD_ERR_HIP);
    if (out.d_prim_ids) { scratch_forget(out.d_prim_ids); (void)hipFree(out.d_prim_ids); out.d_prim_ids = nullptr; }
    if (take_ids) out.d_prim_ids = d_ids;
    else {
        BVH_HIP_TRY(hipMalloc(&out.d_prim_ids, std::max<size_t>(n, 1) * sizeof(uint32_t)), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipMemcpyAsync(out.d_prim_ids, d_ids, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
    }
    HostNode<T> root;
    { int rb_ = readback(&root, final_nodes.p, sizeof(root), stream); if (rb_) return rb_; }
    if (out.d_nodes) { scratch_forget(out.d_nodes); (void)hipFree(out.d_nodes); }
    out.d_nodes = final_nodes.p;
    out.d_nodes_count = out.node_count;
    final_nodes.p = nullptr;
    out.root_index = static_cast<uint32_t>(root.index);
    for (int k = 0; k < 6; ++k) out.root_bounds[k] = root.bounds[k];
    return BVH_AMD_OK;
}

} // namespace bld
} // namespace bvh_amd
