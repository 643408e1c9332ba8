// K1-K4, K7-K9 — MiniTreeBuilder (mini_tree_builder.h:47-310) on gfx950, bit-exact with the reference:
//
//   build_mini_trees (:160-205): centroid bounds -> (2^log2_grid_dim)^3 Morton grid cell per primitive (16^3 by default) -> greedy merge of adjacent
//       cells up to parallel_threshold (when pruning is on) -> one BinnedSahBuilder tree per group over the group's
//       ids in ascending order (:124). The reference's per-thread bin vectors + std::sort become one histogram,
//       a one-wavefront merge over the cells and ONE stable radix sort by group id; all groups are then built
//       simultaneously by the forest variant of the binned builder (build_binned.hip).
//   prune_mini_trees (:207-247): area threshold from the serially summed root areas; per tree, the reference's
//       explicit-stack DFS (second child first) cuts at nodes with half_area < threshold or leaves; each cut subtree
//       is re-laid out by extract_bvh (bvh.h:92-122; right child first, children allocated at visit time, leaves'
//       primitives re-packed in visit order). Here: one lane per tree / per cut replays those DFS orders, twice
//       (count, then write) around exclusive scans that give every cut its node and primitive offsets.
//   build_top_bvh (:249-310): SweepSahBuilder with leaf size 1 over the cut roots (build_sweep.hip), then the splice:
//       top leaves become copies of the cut roots, tree i's nodes 1.. go to node_offsets[i] + j, its primitives to
//       prim_offsets[i], indices rebased. The count pass already knows both offsets (a leaf-size-1 tree over m
//       items has exactly 2m - 1 nodes), so the write pass stores straight into the final arrays.

#include "build_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <system_error>
#include <condition_variable>
#include <thread>

namespace bvh_amd {

using namespace bld;

template <typename T>
int build_binned_forest_device(const T* d_bboxes, const T* d_centers, uint32_t* d_ids, uint32_t n, const uint32_t* d_group_begin,
                               uint32_t n_groups, const bvh_build_config& cfg, DevBuf<HostNode<T>>& trees,
                               DevBuf<uint32_t>& tree_node_off, uint32_t& total_nodes, hipStream_t stream,
                               const std::function<int(const ANode<T>*, PhaseB*)>& roots_ready);
template <typename T>
int sweep_core(const T* d_bboxes, const T* d_centers, size_t n, uint32_t min_leaf, uint32_t max_leaf,
               DevBuf<HostNode<T>>& final_nodes, DevBuf<uint32_t>& ord, size_t& total_nodes, hipStream_t stream, int dim);

namespace {

constexpr uint32_t kDefaultLog2Grid = 4;                     // Config::log2_grid_dim (:42); at most 10: MortonCode has 32 bits (:169)
constexpr int kWalkStack = 160;

template <typename T> struct Eps;
template <> struct Eps<float>  { static constexpr float v = 1.1920928955078125e-07f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

template <typename T> __device__ inline T guarded_inverse(T x) {            // utils.h:59-63
    return __builtin_fabs(x) <= Eps<T>::v ? static_cast<T>(__builtin_copysign(static_cast<double>(Ord<T>::kMax), static_cast<double>(x))) : T(1) / x;
}

struct MtScalars { uint32_t n_groups, n_cuts, error, pad; };

template <typename T>
__global__ void k_mt_prepare(typename Ord<T>::U* keybox, MtScalars* sc) {
    for (int k = threadIdx.x; k < 3; k += blockDim.x) { keybox[k] = Ord<T>::enc(Ord<T>::kMax); keybox[3 + k] = Ord<T>::enc(-Ord<T>::kMax); }
    if (threadIdx.x == 0) { MtScalars z = {}; *sc = z; }
}

// center_bbox (:162-167)
template <typename T>
__global__ void __launch_bounds__(256) k_center_bounds(const T* centers, uint32_t n, typename Ord<T>::U* keybox) {
    __shared__ typename Ord<T>::U slo[3], shi[3];
    if (threadIdx.x < 3) { slo[threadIdx.x] = Ord<T>::enc(Ord<T>::kMax); shi[threadIdx.x] = Ord<T>::enc(-Ord<T>::kMax); }
    __syncthreads();
    T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
    for (size_t i = blockIdx.x * size_t{256} + threadIdx.x; i < n; i += size_t{gridDim.x} * 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T v = centers[3 * i + k]; lo[k] = pick_min(lo[k], v); hi[k] = pick_max(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], klo); atomicMax(&shi[k], khi); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&keybox[threadIdx.x], slo[threadIdx.x]); atomicMax(&keybox[3 + threadIdx.x], shi[threadIdx.x]); }
}

__device__ inline uint32_t spread3(uint32_t v) {             // utils.h:104-115 for inputs below 2^10: bit k -> bit 3k
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// grid cell of every primitive (:170-185) + cell histogram
template <typename T>
__global__ void __launch_bounds__(256) k_cells(const T* centers, uint32_t n, const typename Ord<T>::U* keybox, uint32_t grid_dim, uint32_t cells,
                                               uint32_t* codes, uint32_t* hist) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T lo = Ord<T>::dec(keybox[k]), hi = Ord<T>::dec(keybox[3 + k]);
        const T scale = static_cast<T>(grid_dim) * guarded_inverse(hi - lo);
        const T shift = (-lo) * scale;
        const T p = pick_max(Ord<T>::fma_(centers[3ull * i + k], scale, shift), T(0));
        g[k] = p >= static_cast<T>(grid_dim - 1) ? grid_dim - 1 : static_cast<uint32_t>(p);
    }
    const uint32_t code = (spread3(g[0]) | (spread3(g[1]) << 1) | (spread3(g[2]) << 2)) & (cells - 1);
    codes[i] = code;
    atomicAdd(&hist[code], 1u);
}

// The same for grids of at most 4096 cells (the default 16^3), round 4: the histogram of a block's chunk is built in LDS and every
// non-empty cell is flushed with ONE global atomic. k_cells above issues a device-scope atomic per primitive — 10M of them onto 4096
// words took 0.68 ms, a twentieth of a 10M-triangle Low build — which this cuts by the chunk's primitives per non-empty cell
// (uniformly random input: chunk / 4096; a mesh in a coherent order: far more).
constexpr uint32_t kCellsLds = 4096;
template <typename T>
__global__ void __launch_bounds__(1024) k_cells_lds(const T* centers, uint32_t n, const typename Ord<T>::U* keybox, uint32_t grid_dim, uint32_t cells,
                                                    uint32_t* codes, uint32_t* hist, uint32_t chunk) {
    __shared__ uint32_t local[kCellsLds];
    for (uint32_t q = threadIdx.x; q < cells; q += 1024) local[q] = 0;
    T scale[3], shift[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T lo = Ord<T>::dec(keybox[k]), hi = Ord<T>::dec(keybox[3 + k]);
        scale[k] = static_cast<T>(grid_dim) * guarded_inverse(hi - lo);
        shift[k] = (-lo) * scale[k];
    }
    __syncthreads();
    const uint32_t begin = blockIdx.x * chunk, end = min(n, begin + chunk);
    for (uint32_t i = begin + threadIdx.x; i < end; i += 1024) {
        uint32_t g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T p = pick_max(Ord<T>::fma_(centers[3ull * i + k], scale[k], shift[k]), T(0));
            g[k] = p >= static_cast<T>(grid_dim - 1) ? grid_dim - 1 : static_cast<uint32_t>(p);
        }
        const uint32_t code = (spread3(g[0]) | (spread3(g[1]) << 1) | (spread3(g[2]) << 2)) & (cells - 1);
        codes[i] = code;
        atomicAdd(&local[code], 1u);
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < cells; q += 1024) { const uint32_t v = local[q]; if (v) atomicAdd(&hist[q], v); }
}

// merge_small_bins (:84-91) + remove_empty_bins (:93-96) by one wavefront, 64 cells per step. The reference's loop starts a bin
// at cell i and absorbs the following cells while the running size stays <= threshold; empty cells never change the outcome
// (they join anything, or start a bin that the next non-empty cell either joins or replaces), so the greedy runs over the
// non-empty cells only: `acc` = size of the open bin (0 = none). Steps = cells / 64 + bins closed.
__global__ void __launch_bounds__(64) k_merge_cells(const uint32_t* hist, uint32_t cells, int merge, uint32_t threshold, uint32_t* group_of,
                                                    uint32_t* group_begin, MtScalars* sc) {
    const uint32_t lane = threadIdx.x;
    uint32_t groups = 0, run = 0, acc = 0;                   // wave-uniform: bins opened, primitives in closed bins, open bin's size
    for (uint32_t base = 0; base < cells; base += 64) {
        const uint32_t cell = base + lane;
        const uint32_t c = cell < cells ? hist[cell] : 0;
        uint64_t todo = __ballot(c != 0);
        if (!todo) continue;
        uint32_t incl = c;                                    // inclusive prefix sum of the 64 counts
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= static_cast<uint32_t>(d)) incl += o; }
        uint32_t consumed = 0;                                // counts of this step already placed in closed bins / before the open bin's part
        uint32_t mine = 0;
        while (todo) {
            if (acc == 0) {                                   // open a bin at the first unplaced non-empty cell
                const int first = __builtin_ctzll(todo);
                if (lane == 0) group_begin[groups] = run;
                if (lane == static_cast<uint32_t>(first)) mine = groups;
                ++groups;
                acc = __shfl(c, first);
                consumed = __shfl(incl, first);
                todo &= todo - 1;
                continue;
            }
            const bool joins = merge && acc + (incl - consumed) <= threshold;     // monotone over the unplaced lanes
            const uint64_t rest = todo & ~__ballot(joins);
            const uint64_t joined = todo & ~rest;
            if ((joined >> lane) & 1) mine = groups - 1;
            if (!rest) { acc += __shfl(incl, 63) - consumed; todo = 0; break; }
            const int stop = __builtin_ctzll(rest);           // this cell does not fit: close the bin in front of it
            const uint32_t before = __shfl(incl, stop) - __shfl(c, stop);
            run += acc + (before - consumed);
            acc = 0;
            consumed = before;
            todo = rest;
        }
        if (c != 0) group_of[cell] = mine;
    }
    run += acc;
    if (lane == 0) { group_begin[groups] = run; sc->n_groups = groups; }
}

// The same for grids of at most kMergeBlockCells cells (log2_grid_dim <= 4, the default 16^3 included) by ONE block: with P =
// exclusive prefix sums of the cell counts, the bin the reference opens at cell k absorbs cell j > k exactly while
// P[j + 1] - P[k] <= threshold (:87-88: the running size plus the next cell's), so next(k) = the first j > k with
// P[j + 1] > P[k] + threshold is a binary search per cell, all cells at once; the bins are the orbit of cell 0 under next, marked by
// pointer doubling (reach <- reach | J(reach), J <- J o J: log2(cells) rounds); empty bins are dropped (:93-96) and the surviving
// bins are numbered by a scan. Without merging (pruning off) every cell is its own bin. 1.39 ms -> ~0.03 ms at the default grid.
constexpr uint32_t kMergeBlockCells = 4096;
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* warp_sums, uint32_t& total) {      // 1024 threads
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= static_cast<uint32_t>(d)) incl += o; }
    if (lane == 63) warp_sums[wave] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = warp_sums[w]; if (w < wave) base += x; all += x; }
    __syncthreads();
    total = all;
    return base + incl - v;
}
__global__ void __launch_bounds__(1024) k_merge_cells_block(const uint32_t* hist, uint32_t cells, int merge, uint32_t threshold, uint32_t* group_of,
                                                            uint32_t* group_begin, MtScalars* sc) {
    __shared__ uint32_t P[kMergeBlockCells + 1];             // P[k] = primitives in cells [0, k)
    __shared__ uint32_t nxt[kMergeBlockCells];               // next(k); bit 31: a non-empty bin starts at k
    __shared__ uint32_t jmp[2][kMergeBlockCells];            // next^(2^d), double-buffered
    __shared__ uint8_t reach[kMergeBlockCells];              // k lies on the orbit of cell 0
    __shared__ uint32_t warp_sums[16];
    constexpr uint32_t kPer = kMergeBlockCells / 1024;       // consecutive cells per thread
    const uint32_t t = threadIdx.x, first = t * kPer;
    uint32_t cnt[kPer], local = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) { cnt[q] = first + q < cells ? hist[first + q] : 0u; local += cnt[q]; }
    uint32_t total = 0;
    uint32_t run = block_exclusive_scan(local, warp_sums, total);
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) { if (first + q <= cells) P[first + q] = run; run += cnt[q]; }
    if (t == 0) P[cells] = total;
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t k = first + q;
        if (k >= cells) break;
        uint32_t j = k + 1;
        if (merge) {
            const uint64_t limit = uint64_t{P[k]} + threshold;
            uint32_t lo = k + 1, hi = cells;                  // smallest j in [k + 1, cells] with j == cells or P[j + 1] > limit
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (P[mid + 1] > limit) hi = mid; else lo = mid + 1; }
            j = lo;
        }
        nxt[k] = j;
    }
    __syncthreads();
    // the bins: 0, next(0), next(next(0)), ...; `cells` acts as the end marker (a fixed point of next)
    for (uint32_t k = t; k < cells; k += 1024) { jmp[0][k] = nxt[k]; reach[k] = k == 0 ? 1 : 0; }
    __syncthreads();
    if (merge) {
        int cur = 0;
        for (uint32_t span = 1; span < cells; span <<= 1) {   // after the round: reach = {next^i(0) : i < 2 span}, jmp[cur] = next^(2 span)
            for (uint32_t k = t; k < cells; k += 1024) {
                const uint32_t j = jmp[cur][k];
                if (reach[k] && j < cells) reach[j] = 1;      // (benign race: every writer stores 1)
                jmp[cur ^ 1][k] = j < cells ? jmp[cur][j] : cells;
            }
            cur ^= 1;
            __syncthreads();
        }
    } else {
        for (uint32_t k = t; k < cells; k += 1024) reach[k] = 1;
        __syncthreads();
    }
    for (uint32_t k = t; k < cells; k += 1024) {
        const uint32_t j = nxt[k];
        if (reach[k] && P[j] != P[k]) nxt[k] = 0x80000000u | j;          // a surviving (non-empty) bin starts here
    }
    __syncthreads();
    uint32_t starts[kPer], nstart = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t k = first + q;
        // only cells on the orbit carry the flag; off-orbit cells still hold a plain next() below 2^31
        starts[q] = k < cells && (nxt[k] & 0x80000000u) ? 1u : 0u;
        nstart += starts[q];
    }
    uint32_t groups = 0;
    uint32_t g = block_exclusive_scan(nstart, warp_sums, groups);     // bins that start before this thread's cells
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t k = first + q;
        if (k >= cells) break;
        if (starts[q]) { group_begin[g] = P[k]; ++g; }
        if (cnt[q] != 0) group_of[k] = g - 1;                 // the latest bin started at or before k (a non-empty cell always lies in one)
    }
    if (t == 0) { group_begin[groups] = total; sc->n_groups = groups; }
}

__global__ void __launch_bounds__(256) k_group_keys(const uint32_t* codes, const uint32_t* group_of, uint32_t n, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = group_of[codes[i]];
    vals[i] = i;
}

template <typename T> __device__ inline T node_half_area(const HostNode<T>& nd) {
    const T d0 = nd.bounds[1] - nd.bounds[0], d1 = nd.bounds[3] - nd.bounds[2], d2 = nd.bounds[5] - nd.bounds[4];
    return (d0 + d1) * d2 + d0 * d1;
}
template <typename T> __device__ inline bool node_is_leaf(const HostNode<T>& nd) { return (nd.index & kCountMask) != 0; }

// avg_area summed in tree order (:209-213): the SUM is serial (its rounding depends on the order), the loads are not — the block
// fetches the root areas of 1024 trees at a time into LDS, one lane adds them in order (0.29 ms -> ~0.02 ms for 724 trees)
template <typename T>
__global__ void __launch_bounds__(1024) k_prune_threshold(const HostNode<T>* trees, const uint32_t* tree_off, uint32_t n_trees, T ratio, T* threshold) {
    __shared__ T area[1024];
    T avg = T(0);
    for (uint32_t base = 0; base < n_trees; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        if (t < n_trees) area[threadIdx.x] = node_half_area(trees[tree_off[t]]);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t m = min(1024u, n_trees - base);
            for (uint32_t q = 0; q < m; ++q) avg += area[q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        avg /= static_cast<T>(n_trees);
        *threshold = avg * ratio;
    }
}

// The cut DFS (:216-232): pass 0 counts the cuts of each tree, pass 1 writes them at cut_off[tree].
template <typename T>
__global__ void __launch_bounds__(64) k_prune_walk(const HostNode<T>* trees, const uint32_t* tree_off, uint32_t n_trees, const T* threshold,
                                                   int pass, uint32_t* cut_count, const uint32_t* cut_off, uint2* cuts, MtScalars* sc) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_trees) return;
    const HostNode<T>* tree = trees + tree_off[t];
    const T thr = *threshold;
    uint32_t stack[kWalkStack];
    int sp = 0;
    stack[sp++] = 0;
    uint32_t count = 0;
    const uint32_t base = pass ? cut_off[t] : 0;
    while (sp) {
        const uint32_t id = stack[--sp];
        const HostNode<T>& nd = tree[id];
        if (node_half_area(nd) < thr || node_is_leaf(nd)) {
            if (pass) cuts[base + count] = make_uint2(t, id);
            ++count;
        } else {
            if (sp + 2 > kWalkStack) { atomicOr(&sc->error, 1u); break; }
            const uint32_t f = static_cast<uint32_t>(nd.index >> kCountBits);
            stack[sp++] = f;
            stack[sp++] = f + 1;
        }
    }
    if (!pass) cut_count[t] = count;
}

// Without pruning every mini-tree is one cut, and what the extraction's count pass and its two scans would find is known: tree t moves
// nodes(t) - 1 nodes and all of its primitives, so the offsets are differences of the forest's own prefix sums.
__global__ void __launch_bounds__(256) k_whole_trees_as_cuts(uint32_t n_trees, const uint32_t* tree_off, const uint32_t* group_begin, uint2* cuts,
                                                             uint32_t* node_off, uint32_t* prim_off) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_trees) return;
    cuts[t] = make_uint2(t, 0);
    node_off[t] = tree_off[t] - tree_off[0] - t;
    prim_off[t] = group_begin[t] - group_begin[0];
}

// The top-level builder's inputs (:251-256) straight from the forest's working roots (without pruning the cut roots ARE the tree roots,
// and their boxes are final long before the trees below them are): box and (max + min) * 0.5 (bbox.h:30) per tree.
template <typename T>
__global__ void __launch_bounds__(256) k_top_inputs(const ANode<T>* roots, uint32_t n_trees, T* top_boxes, T* top_centers) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_trees) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T lo = roots[t].lo[k], hi = roots[t].hi[k];
        top_boxes[6ull * t + k] = lo;
        top_boxes[6ull * t + 3 + k] = hi;
        top_centers[3ull * t + k] = (hi + lo) * T(0.5);
    }
}

// extract_bvh (bvh.h:92-122) per cut. pass 0: sizes (nodes - 1, prims); pass 1: write into the final arrays.
template <typename T>
struct ExtractArgs {
    const HostNode<T>* trees; const uint32_t* tree_off; const uint32_t* group_begin; const uint32_t* ids;
    const uint2* cuts; uint32_t n_cuts;
    uint32_t* nodes_minus1; uint32_t* prims;                  // pass 0 outputs / pass 1 inputs are their exclusive scans
    const uint32_t* node_off; const uint32_t* prim_off;
    HostNode<T>* out_nodes; uint32_t* out_ids; HostNode<T>* cut_roots; T* top_boxes; T* top_centers;
    uint32_t top_nodes;                                       // 2 * n_cuts - 1
    MtScalars* sc;
};

template <typename T>
__global__ void __launch_bounds__(64) k_extract(ExtractArgs<T> a, int pass) {
    using I = typename IndexOf<T>::Type;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_cuts) return;
    const uint2 cut = a.cuts[i];
    const uint32_t t = cut.x, r = cut.y;
    const HostNode<T>* tree = a.trees + a.tree_off[t];
    const uint32_t gb = a.group_begin[t];
    const uint32_t node_base = pass ? a.top_nodes - 1 + a.node_off[i] : 0;     // node_offsets[i] (:268)
    const uint32_t prim_base = pass ? a.prim_off[i] : 0;                        // prim_offsets[i] (:269)
    auto rebased = [&](HostNode<T> nd, uint32_t local_first) {                 // copy_node (:275-279)
        const uint32_t cnt = static_cast<uint32_t>(nd.index & kCountMask);
        nd.index = cnt ? ((static_cast<I>(prim_base + local_first) << kCountBits) | cnt) : (static_cast<I>(node_base + local_first) << kCountBits);
        return nd;
    };
    uint32_t n_nodes = 1, n_prims = 0;
    HostNode<T> root_rec;
    if (r == 0) {                                             // the whole mini-tree moves (:239-240)
        n_nodes = a.tree_off[t + 1] - a.tree_off[t];
        n_prims = a.group_begin[t + 1] - gb;
        // (the nodes and ids of a whole tree are copied by k_extract_whole, one block per cut: a tree of 12 k primitives
        //  copied by this one lane was 3.5 ms of the 10M-triangle Low build)
        if (pass) root_rec = rebased(tree[0], static_cast<uint32_t>(tree[0].index >> kCountBits));
    } else {
        uint2 stack[kWalkStack];
        int sp = 0;
        stack[sp++] = make_uint2(r, 0);
        while (sp) {
            const uint2 top = stack[--sp];
            const HostNode<T> src = tree[top.x];
            HostNode<T> rec;
            if (node_is_leaf(src)) {
                const uint32_t first = static_cast<uint32_t>(src.index >> kCountBits), cnt = static_cast<uint32_t>(src.index & kCountMask);
                if (pass) {
                    rec = rebased(src, n_prims);
                    for (uint32_t q = 0; q < cnt; ++q) a.out_ids[prim_base + n_prims + q] = a.ids[gb + first + q];
                }
                n_prims += cnt;
            } else {
                if (sp + 2 > kWalkStack) { atomicOr(&a.sc->error, 1u); break; }
                const uint32_t first = static_cast<uint32_t>(src.index >> kCountBits);
                if (pass) rec = rebased(src, n_nodes);
                stack[sp++] = make_uint2(first, n_nodes);
                stack[sp++] = make_uint2(first + 1, n_nodes + 1);
                n_nodes += 2;
            }
            if (pass) { if (top.y == 0) root_rec = rec; else a.out_nodes[node_base + top.y] = rec; }
        }
    }
    if (!pass) { a.nodes_minus1[i] = n_nodes - 1; a.prims[i] = n_prims; return; }
    a.cut_roots[i] = root_rec;
    if (!a.top_boxes) return;                                 // (already written from the forest's working roots)
    // the top-level builder's inputs (:251-256)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T lo = root_rec.bounds[2 * k], hi = root_rec.bounds[2 * k + 1];
        a.top_boxes[6ull * i + k] = lo;
        a.top_boxes[6ull * i + 3 + k] = hi;
        a.top_centers[3ull * i + k] = (hi + lo) * T(0.5);    // bbox.h:30
    }
}

// ---- extract_bvh per cut without a walk per cut (round 4) -------------------------------------------------------------------------
// k_extract above replays extract_bvh's explicit stack with ONE LANE per cut: a cut of a 1M-triangle Medium build has ~100 nodes, of a
// 10M-triangle terrain ~1000 (up to 3500) — hundreds to thousands of dependent loads per lane, twice (count, then write), 0.36 ms of a
// 3.1 ms build. The layout extract_bvh produces is a function of subtree sizes (extract.hip's header): with inner(.) / prims(.) the
// inner nodes / primitives of a subtree, a node X processed after r(X) inner nodes and p(X) primitives puts its children at
// 1 + 2 r(X), 2 + 2 r(X), and r(second) = r(X) + 1, p(second) = p(X), r(first) = r(X) + 1 + inner(second), p(first) = p(X) +
// prims(second). So:
//   k_forest_parents  parent of every node (tree-local), one lane per node;
//   k_cut_counts      inner / prims of every node at or below a cut root: one lane per leaf climbs, the second child to arrive at a
//                     node sums its children (arrival tickets), and the climb ends at the cut root — this also IS the count pass:
//                     a cut holds 1 + 2 inner(root) nodes and prims(root) primitives;
//   k_cut_assign      one lane per node climbs to its cut root adding up the r / p steps on the way (a first child, at its odd index,
//                     adds its sibling's counts) and writes the node where that puts it; nodes above the cuts find no cut and are dropped.
// Every lane's chain is the node's depth inside its cut (~7 for 100 nodes, ~12 for 3500), all nodes at once, no stacks to overflow.
template <typename T>
__global__ void __launch_bounds__(256) k_forest_parents(const HostNode<T>* trees, const uint32_t* tree_off, uint32_t* parent) {
    const uint32_t base = tree_off[blockIdx.x], nn = tree_off[blockIdx.x + 1] - base;
    for (uint32_t j = threadIdx.x; j < nn; j += 256) {
        const HostNode<T> nd = trees[base + j];
        if (node_is_leaf(nd)) continue;
        const uint32_t f = static_cast<uint32_t>(nd.index >> kCountBits);
        parent[base + f] = j; parent[base + f + 1] = j;
    }
}

__global__ void __launch_bounds__(256) k_cut_flags(const uint2* cuts, uint32_t n_cuts, const uint32_t* tree_off, uint32_t* cut_of) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_cuts) cut_of[tree_off[cuts[i].x] + cuts[i].y] = i + 1;      // 0: not the root of a cut
}

template <typename T>
__global__ void __launch_bounds__(256) k_cut_counts(const HostNode<T>* trees, const uint32_t* tree_off, const uint32_t* parent, const uint32_t* cut_of,
                                                    uint32_t* arrived, uint32_t* inner, uint32_t* prims) {
    const uint32_t base = tree_off[blockIdx.x], nn = tree_off[blockIdx.x + 1] - base;
    const HostNode<T>* tree = trees + base;
    // One block owns one tree, so every exchange is between waves of ONE workgroup: workgroup-scope release / acquire on the ticket is
    // all the ordering needed — waits for the wave's own stores, no cache maintenance. (The first version used __threadfence() =
    // agent scope, which on gfx950 writes back and invalidates the XCD's L2 per call: two per climbed node made this kernel 6 ms.)
    auto put = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto get = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    for (uint32_t j = threadIdx.x; j < nn; j += 256) {
        const HostNode<T> nd = tree[j];
        if (!node_is_leaf(nd)) continue;
        put(&inner[base + j], 0u);
        put(&prims[base + j], static_cast<uint32_t>(nd.index & kCountMask));
        uint32_t cur = j;
        while (cur != 0 && cut_of[base + cur] == 0) {         // (every leaf is a cut or below one: the climb ends at a cut root)
            const uint32_t p = parent[base + cur];
            // release (this subtree's counts before the ticket) + acquire (the sibling's counts after it)
            if (__hip_atomic_fetch_add(&arrived[base + p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) break;   // first to arrive: the sibling's lane finishes this node
            const uint32_t f = static_cast<uint32_t>(tree[p].index >> kCountBits);
            put(&inner[base + p], 1u + get(&inner[base + f]) + get(&inner[base + f + 1]));
            put(&prims[base + p], get(&prims[base + f]) + get(&prims[base + f + 1]));
            cur = p;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_cut_sizes(ExtractArgs<T> a, const uint32_t* inner, const uint32_t* prims) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_cuts) return;
    const uint32_t t = a.cuts[i].x, r = a.cuts[i].y;
    if (r == 0) {                                             // the whole mini-tree moves (:239-240)
        a.nodes_minus1[i] = a.tree_off[t + 1] - a.tree_off[t] - 1;
        a.prims[i] = a.group_begin[t + 1] - a.group_begin[t];
    } else {
        a.nodes_minus1[i] = 2 * inner[a.tree_off[t] + r];
        a.prims[i] = prims[a.tree_off[t] + r];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_cut_assign(ExtractArgs<T> a, const uint32_t* parent, const uint32_t* cut_of, const uint32_t* inner,
                                                    const uint32_t* prims) {
    using I = typename IndexOf<T>::Type;
    const uint32_t t = blockIdx.x, base = a.tree_off[t], nn = a.tree_off[t + 1] - base, gb = a.group_begin[t];
    const HostNode<T>* tree = a.trees + base;
    for (uint32_t j = threadIdx.x; j < nn; j += 256) {
        uint32_t r = 0, p = 0, own_step = 0, cut = cut_of[base + j];
        const bool is_root = cut != 0;
        if (!is_root) {
            uint32_t at = j;
            while (at != 0) {
                const bool first = (at & 1u) != 0;            // the first child of a pair sits at the odd index (bvh.h:34)
                const uint32_t step_r = 1u + (first ? inner[base + at + 1] : 0u), step_p = first ? prims[base + at + 1] : 0u;
                if (at == j) own_step = step_r;
                r += step_r; p += step_p;
                at = parent[base + at];
                cut = cut_of[base + at];
                if (cut) break;
            }
            if (!cut) continue;                               // above the cuts: the top-level build replaces these nodes
        }
        const uint32_t r_parent = r - own_step;               // the node's place follows from its PARENT's r: 1 + 2 r (first child), 2 + 2 r
        const uint32_t c = cut - 1;
        const bool whole = a.cuts[c].y == 0;
        if (whole && !is_root) continue;                      // copied as it lies by k_extract_whole
        const uint32_t node_base = a.top_nodes - 1 + a.node_off[c], prim_base = a.prim_off[c];
        HostNode<T> nd = tree[j];
        const uint32_t cnt = static_cast<uint32_t>(nd.index & kCountMask), first_id = static_cast<uint32_t>(nd.index >> kCountBits);
        if (whole) {                                          // copy_node (:275-279) of the root of a tree that moves whole
            nd.index = cnt ? ((static_cast<I>(prim_base + first_id) << kCountBits) | cnt) : (static_cast<I>(node_base + first_id) << kCountBits);
        } else if (cnt) {
            for (uint32_t q = 0; q < cnt; ++q) a.out_ids[prim_base + p + q] = a.ids[gb + first_id + q];
            nd.index = (static_cast<I>(prim_base + p) << kCountBits) | cnt;
        } else {
            nd.index = static_cast<I>(node_base + 1 + 2 * r) << kCountBits;
        }
        if (!is_root) { a.out_nodes[node_base + ((j & 1u) ? 1u : 2u) + 2 * r_parent] = nd; continue; }
        a.cut_roots[c] = nd;
#pragma unroll
        for (int k = 0; k < 3; ++k) {                         // the top-level builder's inputs (:251-256)
            const T lo = nd.bounds[2 * k], hi = nd.bounds[2 * k + 1];
            a.top_boxes[6ull * c + k] = lo;
            a.top_boxes[6ull * c + 3 + k] = hi;
            a.top_centers[3ull * c + k] = (hi + lo) * T(0.5);    // bbox.h:30
        }
    }
}

// the copy of a cut that is a whole mini-tree (r == 0, :239-240 + copy_node :275-279), one block per cut
template <typename T>
__global__ void __launch_bounds__(256) k_extract_whole(ExtractArgs<T> a) {
    using I = typename IndexOf<T>::Type;
    const uint32_t i = blockIdx.x;
    const uint2 cut = a.cuts[i];
    if (cut.y != 0) return;
    const uint32_t t = cut.x;
    const HostNode<T>* tree = a.trees + a.tree_off[t];
    const uint32_t gb = a.group_begin[t];
    const uint32_t node_base = a.top_nodes - 1 + a.node_off[i], prim_base = a.prim_off[i];
    const uint32_t n_nodes = a.tree_off[t + 1] - a.tree_off[t], n_prims = a.group_begin[t + 1] - gb;
    for (uint32_t j = 1 + threadIdx.x; j < n_nodes; j += 256) {
        HostNode<T> nd = tree[j];
        const uint32_t cnt = static_cast<uint32_t>(nd.index & kCountMask), first = static_cast<uint32_t>(nd.index >> kCountBits);
        nd.index = cnt ? ((static_cast<I>(prim_base + first) << kCountBits) | cnt) : (static_cast<I>(node_base + first) << kCountBits);
        a.out_nodes[node_base + j] = nd;
    }
    for (uint32_t p = threadIdx.x; p < n_prims; p += 256) a.out_ids[prim_base + p] = a.ids[gb + p];
}

// top nodes into the final array; top leaves become the cut roots (:282-288)
template <typename T>
__global__ void __launch_bounds__(256) k_splice_top(const HostNode<T>* top, const uint32_t* top_ids, uint32_t top_nodes, const HostNode<T>* cut_roots,
                                                    HostNode<T>* out) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= top_nodes) return;
    HostNode<T> nd = top[v];
    if (node_is_leaf(nd)) nd = cut_roots[top_ids[static_cast<uint32_t>(nd.index >> kCountBits)]];
    out[v] = nd;
}

// ---- the top-level build beside the forest (round 4) ----------------------------------------------------------------------------
// Without pruning (Quality::Low, or MiniTreeBuilder with pruning off) the top-level BVH (:249-310) is built over the mini-trees' ROOT
// boxes, and those are final after k_forest_roots / k_medium — a third of the way into the forest build. The top level is a chain of
// small launches and host round trips (the std::sort replay, one sweep level, one k_sweep_medium block ...: 0.9 ms for the 4096 roots
// of a 16^3 grid, 40 % of a 1M-triangle Low build) that leaves the chip idle, while Phase B of the forest (k_small_levels) fills it
// without needing the host. So the top level runs on a stream of its own, driven by a worker thread (both pipelines block on
// readbacks, one host thread cannot drive two), between two events: `roots` (recorded on the caller's stream behind the kernel that
// writes the top-level inputs) and `done` (recorded on the worker's stream; the splice waits for it). The worker is ONE persistent
// thread per calling thread (round 5; rounds 4's builds spawned a std::thread each and that thread spun until the roots were
// announced): idle it sleeps on a condition variable; handed a job while the grid / radix-sort kernels run, it sets its device and
// scopes up and waits for the roots — a bounded spin first (the announcement is ~0.3 ms away and a futex wake-up costs 30-60 us of a
// 1.5 ms build), then the condition variable — so a build that takes long to reach its roots, or eight host threads building at
// once, do not burn a core each.
struct TopHelper {                                            // per calling thread: stream + events of its top-level worker
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t roots = nullptr, done = nullptr, spliced = nullptr;
    // Phase B of the forest beside the worker: a stream whose CU mask leaves `reserved` CUs alone. k_small_levels has thousands of blocks
    // waiting for every slot that frees, and a worker launch of 1024-thread blocks (the LDS sort, the sweep's scans, k_sweep_medium)
    // never finds a CU with room for one — measured: the worker's first such kernel sat 570 us, until Phase B had drained, whatever
    // the stream priority. On the reserved CUs it starts at once; Phase B loses reserved / 256 of the chip.
    hipStream_t phase_b = nullptr;
    hipEvent_t phase_b_done = nullptr;
    // the persistent worker: `work` is the posted job (empty when idle)
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> work;
    bool busy = false, quit = false;
    bool post(std::function<void()> fn) {                     // false: no thread to be had (the top level then follows the forest)
        std::unique_lock<std::mutex> lock(m);
        if (!worker.joinable()) {
            try {
                worker = std::thread([this] {
                    std::unique_lock<std::mutex> l(m);
                    for (;;) {
                        cv.wait(l, [this] { return quit || static_cast<bool>(work); });
                        if (quit) return;
                        std::function<void()> fn = std::move(work);
                        work = nullptr;
                        l.unlock();
                        fn();
                        l.lock();
                        busy = false;
                        cv.notify_all();
                    }
                });
            } catch (const std::system_error&) { return false; }
        }
        cv.wait(lock, [this] { return !busy; });
        work = std::move(fn);
        busy = true;
        cv.notify_all();
        return true;
    }
    void wait_idle() { std::unique_lock<std::mutex> lock(m); cv.wait(lock, [this] { return !busy; }); }
    void stop_worker() {
        if (!worker.joinable()) return;
        { std::lock_guard<std::mutex> lock(m); quit = true; }
        cv.notify_all();
        worker.join();
        quit = false;
    }
    void release() {
        wait_idle();
        if (stream) {                                         // (scratch cached by the worker hangs on fences, not on this handle: build_device.hip)
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (phase_b) (void)hipStreamDestroy(phase_b);
        for (hipEvent_t* e : { &roots, &done, &spliced, &phase_b_done }) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        stream = nullptr; phase_b = nullptr; device = -1;
    }
    int prepare(int dev) {
        if (device == dev) return BVH_AMD_OK;
        release();
        // the highest priority the device offers: the worker's launches are one or a few blocks each and sit on the critical path, while
        // Phase B on the caller's stream has thousands of blocks waiting for every slot that frees
        int least = 0, greatest = 0;
        BVH_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest), BVH_AMD_ERR_HIP);
        static const bool flat = BVH_DEV_INT("BVH_AMD_TOP_PRIORITY", 1) == 0;      // A/B runs
        BVH_HIP_TRY(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, flat ? least : greatest), BVH_AMD_ERR_HIP);
        for (hipEvent_t* e : { &roots, &done, &spliced, &phase_b_done }) BVH_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming), BVH_AMD_ERR_HIP);
        // 32 = bits 0..31 of the mask = one CU of every shader engine of every XCD (the mask's bits go round the XCDs first, then round
        // an XCD's four engines): 8 CUs (one engine per XCD short of a CU) cost Phase B the same 8-12 % and serve the worker worse
        static const int reserve = BVH_DEV_INT("BVH_AMD_TOP_RESERVE", 32);            // A/B runs (0: off)
        int cus = 0;
        if (reserve > 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 4 * reserve) {
            std::vector<uint32_t> mask((cus + 31) / 32, 0u);
            for (int cu = reserve; cu < cus; ++cu) mask[cu / 32] |= 1u << (cu % 32);
            if (hipExtStreamCreateWithCUMask(&phase_b, static_cast<uint32_t>(mask.size()), mask.data()) != hipSuccess) { (void)hipGetLastError(); phase_b = nullptr; }
        }
        device = dev;
        return BVH_AMD_OK;
    }
    ~TopHelper() { stop_worker(); release(); }
};

template <typename T>
struct TopJob {
    TopHelper* helper = nullptr;                              // set by start(): the job was posted to this helper's worker
    std::atomic<int> go{0};                                   // 0 wait, 1 the roots event is recorded, -1 not needed
    std::mutex go_m;
    std::condition_variable go_cv;
    int rc = BVH_AMD_OK;
    std::string error;
    DevBuf<HostNode<T>> top;
    DevBuf<uint32_t> top_ord;
    size_t top_count = 0;
    bool announced = false;
    bool joined = false;                                      // the caller's stream waits for (or the host has waited for) the worker's stream
    const T* boxes = nullptr; const T* centers = nullptr; uint32_t n_roots = 0;      // set before `go`
    bool started() const { return helper != nullptr; }
    void start(int dev, TopHelper& h, SahParams sah) {
        const bool posted = h.post([this, dev, &h, sah] {
            if (hipSetDevice(dev) != hipSuccess) { rc = BVH_AMD_ERR_HIP; error = "build: hipSetDevice on the top-level worker"; return; }
            StreamScope scratch_on(h.stream);
            SahScope heuristic(sah);
            int state = 0;
            for (uint32_t spin = 0; spin < 200000u && (state = go.load(std::memory_order_acquire)) == 0; ++spin)      // ~0.5 ms
                if ((spin & 63u) == 63u) std::this_thread::yield();
            if (state == 0) {
                std::unique_lock<std::mutex> lock(go_m);
                go_cv.wait(lock, [this] { return go.load(std::memory_order_acquire) != 0; });
                state = go.load(std::memory_order_acquire);
            }
            if (state < 0) return;
            if (hipStreamWaitEvent(h.stream, h.roots, 0) != hipSuccess) { rc = BVH_AMD_ERR_HIP; error = "build: hipStreamWaitEvent on the top-level worker"; return; }
            rc = sweep_core<T>(boxes, centers, n_roots, 1, 1, top, top_ord, top_count, h.stream, 3);
            if (rc) { error = current_error(); return; }
            if (hipEventRecord(h.done, h.stream) != hipSuccess) { rc = BVH_AMD_ERR_HIP; error = "build: hipEventRecord on the top-level worker"; }
        });
        if (posted) helper = &h;
    }
    void signal(int state) {
        { std::lock_guard<std::mutex> lock(go_m); go.store(state, std::memory_order_release); }
        go_cv.notify_all();
    }
    // the worker is through with this job (its host side); a job whose roots were never announced is called off
    void finish() {
        if (!helper) return;
        if (go.load(std::memory_order_acquire) == 0) signal(-1);
        helper->wait_idle();
    }
    // Every exit: the worker's KERNELS may still be reading the inputs (top_boxes / top_centers) and writing `top` when the buffers
    // are released in the order of the caller's stream. The success path makes the caller's stream wait for `done`
    // (joined = true); any other exit after the announcement waits for the worker's stream here.
    ~TopJob() {
        finish();
        if (helper && announced && !joined) (void)hipStreamSynchronize(helper->stream);
    }
};

constexpr uint32_t kTopReserveMaxPrims = 4u << 20;

bool top_beside_enabled() {
    static const bool off = BVH_DEV_INT("BVH_AMD_TOP_BESIDE", 1) == 0;      // A/B runs
    return !off;
}

} // namespace

// MiniTreeBuilder::build on the device: final nodes (reference layout) + prim ids, both resident.
template <typename T>
int minitree_core(const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg, bool prune, T prune_ratio, uint32_t log2_grid,
                  DevBuf<HostNode<T>>& final_nodes, DevBuf<uint32_t>& final_ids, size_t& total_nodes, hipStream_t stream)
{
    if (n >= (size_t{1} << 28)) return fail(BVH_AMD_ERR_UNSUPPORTED, "build: more than 2^28 primitives");
    if (log2_grid < 1 || log2_grid > 10) return fail(BVH_AMD_ERR_ARG, "build: log2_grid_dim must be in [1, 10] (three coordinates in a 32-bit Morton code)");
    const uint32_t n32 = static_cast<uint32_t>(n);
    const uint32_t grid_dim = 1u << log2_grid, cells = 1u << (3 * log2_grid);
    const size_t max_groups = std::min<size_t>(cells, n);
    using U = typename Ord<T>::U;
    DevBuf<U> keybox;
    DevBuf<uint32_t> hist, codes, group_of, group_begin, keys, ids, keys_tmp, vals_tmp;
    DevBuf<MtScalars> scalars;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(keybox.alloc(6)); A(hist.alloc(cells)); A(codes.alloc(n)); A(group_of.alloc(cells)); A(group_begin.alloc(max_groups + 1));
    A(keys.alloc(n)); A(ids.alloc(n)); A(keys_tmp.alloc(n)); A(vals_tmp.alloc(n)); A(scalars.alloc(1));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));

    // ---- build_mini_trees
    BVH_HIP_TRY(hipMemsetAsync(hist.p, 0, size_t{cells} * sizeof(uint32_t), stream), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_mt_prepare<T>, dim3(1), dim3(64), 0, stream, keybox.p, scalars.p);
    // (every block ends with six global atomics on the same six words: 2048 blocks spent 60 us of this kernel's 61 on them at 1M
    //  primitives — and 78 us at 10M; 512 blocks stream the centres just as well)
    const unsigned red_grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 512));
    hipLaunchKernelGGL(k_center_bounds<T>, dim3(red_grid), dim3(256), 0, stream, d_centers, n32, keybox.p);
    if (cells <= kCellsLds) {
        // chunks of at least 8192 primitives, at most ~1024 blocks (four per CU)
        const uint32_t chunk = std::max<uint32_t>(8192u, (n32 + 1023u) / 1024u);
        hipLaunchKernelGGL(k_cells_lds<T>, dim3((n32 + chunk - 1) / chunk), dim3(1024), 0, stream, d_centers, n32, keybox.p, grid_dim, cells, codes.p, hist.p, chunk);
    } else
        hipLaunchKernelGGL(k_cells<T>, dim3((n32 + 255) / 256), dim3(256), 0, stream, d_centers, n32, keybox.p, grid_dim, cells, codes.p,
                           hist.p);
    const uint32_t merge_threshold = static_cast<uint32_t>(std::min<size_t>(cfg.parallel_threshold, 0x7fffffffu));   // counts stay below 2^28
    if (cells <= kMergeBlockCells)
        hipLaunchKernelGGL(k_merge_cells_block, dim3(1), dim3(1024), 0, stream, hist.p, cells, prune ? 1 : 0, merge_threshold, group_of.p, group_begin.p,
                           scalars.p);
    else
        hipLaunchKernelGGL(k_merge_cells, dim3(1), dim3(64), 0, stream, hist.p, cells, prune ? 1 : 0, merge_threshold, group_of.p, group_begin.p,
                           scalars.p);
    hipLaunchKernelGGL(k_group_keys, dim3((n32 + 255) / 256), dim3(256), 0, stream, codes.p, group_of.p, n32, keys.p, ids.p);
    int key_bits = 1;                                         // group ids are below min(cells, n)
    while (key_bits < 32 && (size_t{1} << key_bits) < max_groups) ++key_bits;
    int rc = radix_sort_pairs<uint32_t>(keys.p, ids.p, keys_tmp.p, vals_tmp.p, n32, 1, key_bits, stream);   // stable: ids ascending per group (:124)
    if (rc) return rc;
    DevBuf<T> top_boxes, top_centers;
    TopJob<T> job;                                            // (declared after what its worker reads: joined before those are released)
    static thread_local TopHelper helper;
    const bool beside = !prune && max_groups > static_cast<size_t>(kSmall) && top_beside_enabled() && scratch_pool_enabled();
    if (beside) {                                             // the worker starts while the kernels above run (the host would only wait for them)
        int dev = 0;
        BVH_HIP_TRY(hipGetDevice(&dev), BVH_AMD_ERR_HIP);
        rc = helper.prepare(dev);
        if (rc) return rc;
        job.start(dev, helper, ambient_sah());             // (no thread to be had: the top level follows the forest)
    }
    MtScalars hs;
    { int rb_ = readback(&hs, scalars.p, sizeof(hs), stream); if (rb_) return rb_; }
    const uint32_t n_trees = hs.n_groups;

    DevBuf<HostNode<T>> trees;
    DevBuf<uint32_t> tree_off;
    uint32_t forest_nodes = 0;
    std::function<int(const ANode<T>*, PhaseB*)> roots_ready;
    if (beside && job.started() && n_trees > static_cast<uint32_t>(kSmall)) {
        A(top_boxes.alloc(6 * size_t{n_trees})); A(top_centers.alloc(3 * size_t{n_trees}));
        if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));
        job.boxes = top_boxes.p; job.centers = top_centers.p; job.n_roots = n_trees;
        roots_ready = [&](const ANode<T>* roots, PhaseB* lane) -> int {
            hipLaunchKernelGGL(k_top_inputs<T>, dim3((n_trees + 255) / 256), dim3(256), 0, stream, roots, n_trees, top_boxes.p, top_centers.p);
            BVH_HIP_TRY(hipEventRecord(helper.roots, stream), BVH_AMD_ERR_HIP);
            job.announced = true;
            job.signal(1);
            // (the masked stream costs Phase B ~8-12 % whatever the number of CUs left out; beyond a few million primitives that is more
            //  than the top level's 0.5 ms, which then simply runs in the shadow of the forest's numbering / emit passes)
            if (helper.phase_b && n32 <= kTopReserveMaxPrims) { lane->stream = helper.phase_b; lane->start = helper.roots; lane->done = helper.phase_b_done; }
            return BVH_AMD_OK;
        };
    }
    rc = build_binned_forest_device<T>(d_bboxes, d_centers, ids.p, n32, group_begin.p, n_trees, cfg, trees, tree_off, forest_nodes, stream, roots_ready);
    if (rc) return rc;
    if (!job.announced) job.finish();                         // (the forest never reached the point where it announces its roots)

    // ---- prune_mini_trees: the list of cuts (tree, node) in the reference's order
    DevBuf<uint2> cuts;
    uint32_t n_cuts = n_trees;
    if (prune) {
        DevBuf<T> threshold;
        DevBuf<uint32_t> cut_count, cut_off;
        BVH_HIP_TRY(threshold.alloc(1), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(cut_count.alloc(n_trees), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(cut_off.alloc(n_trees), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_prune_threshold<T>, dim3(1), dim3(1024), 0, stream, trees.p, tree_off.p, n_trees, prune_ratio, threshold.p);
        const unsigned tg = (n_trees + 63) / 64;
        hipLaunchKernelGGL(k_prune_walk<T>, dim3(tg), dim3(64), 0, stream, trees.p, tree_off.p, n_trees, threshold.p, 0, cut_count.p,
                           static_cast<const uint32_t*>(nullptr), static_cast<uint2*>(nullptr), scalars.p);
        rc = exclusive_scan_u32(cut_count.p, cut_off.p, n_trees, &n_cuts, stream);
        if (rc) return rc;
        BVH_HIP_TRY(cuts.alloc(n_cuts), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_prune_walk<T>, dim3(tg), dim3(64), 0, stream, trees.p, tree_off.p, n_trees, threshold.p, 1, cut_count.p,
                           cut_off.p, cuts.p, scalars.p);
        if (!scratch_pool_enabled()) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // plain hipFree of the workspace on return (the pool frees in stream order)
    }

    // ---- extract (sizes -> offsets -> write) and the top-level builder's inputs
    DevBuf<uint32_t> nm1, np, node_off, prim_off;
    DevBuf<HostNode<T>> cut_roots;
    if (prune) { A(nm1.alloc(n_cuts)); A(np.alloc(n_cuts)); } else A(cuts.alloc(n_cuts));
    A(node_off.alloc(n_cuts)); A(prim_off.alloc(n_cuts)); A(cut_roots.alloc(n_cuts));
    if (!job.announced) { A(top_boxes.alloc(6 * size_t{n_cuts})); A(top_centers.alloc(3 * size_t{n_cuts})); }
    A(final_ids.alloc(n));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));
    ExtractArgs<T> ea;
    ea.trees = trees.p; ea.tree_off = tree_off.p; ea.group_begin = group_begin.p; ea.ids = ids.p; ea.cuts = cuts.p; ea.n_cuts = n_cuts;
    ea.nodes_minus1 = nm1.p; ea.prims = np.p; ea.node_off = node_off.p; ea.prim_off = prim_off.p;
    ea.out_nodes = nullptr; ea.out_ids = final_ids.p; ea.cut_roots = cut_roots.p;
    ea.top_boxes = job.announced ? nullptr : top_boxes.p; ea.top_centers = job.announced ? nullptr : top_centers.p;
    ea.top_nodes = 2 * n_cuts - 1; ea.sc = scalars.p;
    const unsigned cg = (n_cuts + 63) / 64;
    uint32_t below = 0;
    static const bool walk_per_cut = BVH_DEV_IS("BVH_AMD_EXTRACT", "walk");    // A/B runs
    const bool per_node = prune && !walk_per_cut;
    DevBuf<uint32_t> x_parent, x_cut_of, x_arrived, x_inner, x_prims;
    if (per_node) {
        A(x_parent.alloc(forest_nodes)); A(x_cut_of.alloc(forest_nodes)); A(x_arrived.alloc(forest_nodes)); A(x_inner.alloc(forest_nodes)); A(x_prims.alloc(forest_nodes));
        if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));
        BVH_HIP_TRY(hipMemsetAsync(x_cut_of.p, 0, size_t{forest_nodes} * 4, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipMemsetAsync(x_arrived.p, 0, size_t{forest_nodes} * 4, stream), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_forest_parents<T>, dim3(n_trees), dim3(256), 0, stream, trees.p, tree_off.p, x_parent.p);
        hipLaunchKernelGGL(k_cut_flags, dim3((n_cuts + 255) / 256), dim3(256), 0, stream, cuts.p, n_cuts, tree_off.p, x_cut_of.p);
        hipLaunchKernelGGL(k_cut_counts<T>, dim3(n_trees), dim3(256), 0, stream, trees.p, tree_off.p, x_parent.p, x_cut_of.p, x_arrived.p, x_inner.p, x_prims.p);
    }
    if (prune) {
        uint32_t prim_total = 0;
        if (per_node) hipLaunchKernelGGL(k_cut_sizes<T>, dim3((n_cuts + 255) / 256), dim3(256), 0, stream, ea, x_inner.p, x_prims.p);
        else hipLaunchKernelGGL(k_extract<T>, dim3(cg), dim3(64), 0, stream, ea, 0);
        rc = exclusive_scan_u32(nm1.p, node_off.p, n_cuts, &below, stream);
        if (rc) return rc;
        rc = exclusive_scan_u32(np.p, prim_off.p, n_cuts, &prim_total, stream);
        if (rc) return rc;
        if (prim_total != n32) return fail(BVH_AMD_ERR_OVERFLOW, "build: mini-tree extraction lost primitives (walk stack overflow?)");
        if (BVH_DEV_STR("BVH_AMD_CUT_STATS")) {                // developer knob: sizes of the cut subtrees
            std::vector<uint32_t> hn(n_cuts);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(hn.data(), nm1.p, size_t{n_cuts} * 4, hipMemcpyDeviceToHost);
            uint32_t bins[12] = {}; uint64_t sum = 0; uint32_t mx = 0;
            for (uint32_t v : hn) { uint32_t b = 0; while ((2u << b) <= v + 1 && b < 11) ++b; bins[b]++; sum += v + 1; mx = std::max(mx, v + 1); }
            fprintf(stderr, "[cuts] trees %u cuts %u mean nodes %.1f max %u; by nodes <2,<4,..:", n_trees, n_cuts, double(sum) / n_cuts, mx);
            for (int b = 0; b < 12; ++b) fprintf(stderr, " %u", bins[b]);
            fprintf(stderr, "\n");
        }
    } else {
        // every mini-tree moves whole (:239-240): the count pass, its two scans and their host round trips have known results
        hipLaunchKernelGGL(k_whole_trees_as_cuts, dim3((n_trees + 255) / 256), dim3(256), 0, stream, n_trees, tree_off.p, group_begin.p, cuts.p,
                           node_off.p, prim_off.p);
        below = forest_nodes - n_trees;
    }
    total_nodes = size_t{ea.top_nodes} + below;
    BVH_HIP_TRY(final_nodes.alloc(total_nodes), BVH_AMD_ERR_HIP);
    ea.out_nodes = final_nodes.p;
    if (per_node) hipLaunchKernelGGL(k_cut_assign<T>, dim3(n_trees), dim3(256), 0, stream, ea, x_parent.p, x_cut_of.p, x_inner.p, x_prims.p);
    else hipLaunchKernelGGL(k_extract<T>, dim3(cg), dim3(64), 0, stream, ea, 1);
    hipLaunchKernelGGL(k_extract_whole<T>, dim3(n_cuts), dim3(256), 0, stream, ea);
    if (prune) {                                              // (only the cut walks have a stack to overflow)
        { int rb_ = readback(&hs, scalars.p, sizeof(hs), stream); if (rb_) return rb_; }
        if (hs.error) return fail(BVH_AMD_ERR_OVERFLOW, "build: mini-tree deeper than the pruning walk stack");
    }

    // ---- build_top_bvh: sweep SAH with one cut root per leaf, then the splice
    DevBuf<HostNode<T>> top_here;
    DevBuf<uint32_t> top_ord_here;
    size_t top_count = 0;
    if (job.announced) {                                      // built beside the forest: wait for the worker, then for its stream
        job.finish();
        if (job.rc) return fail(job.rc, job.error);
        BVH_HIP_TRY(hipStreamWaitEvent(stream, helper.done, 0), BVH_AMD_ERR_HIP);
        job.joined = true;
        top_count = job.top_count;
    } else {
        rc = sweep_core<T>(top_boxes.p, top_centers.p, n_cuts, 1, 1, top_here, top_ord_here, top_count, stream, 3);
        if (rc) return rc;
    }
    const DevBuf<HostNode<T>>& top = job.announced ? job.top : top_here;
    const DevBuf<uint32_t>& top_ord = job.announced ? job.top_ord : top_ord_here;
    if (top_count != ea.top_nodes) return fail(BVH_AMD_ERR_OVERFLOW, "build: unexpected top-level node count");
    hipLaunchKernelGGL(k_splice_top<T>, dim3((ea.top_nodes + 255) / 256), dim3(256), 0, stream, top.p, top_ord.p, ea.top_nodes, cut_roots.p,
                       final_nodes.p);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    if (job.announced) {
        // the worker's stream owns `top` / `top_ord` (they return to its cache): nothing it runs later may start before the splice has read them
        BVH_HIP_TRY(hipEventRecord(helper.spliced, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamWaitEvent(helper.stream, helper.spliced, 0), BVH_AMD_ERR_HIP);
    }
    if (!scratch_pool_enabled()) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // plain hipFree of the workspace on return (the pool frees in stream order)
    return BVH_AMD_OK;
}

template <typename T> int reinsertion_optimize_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim);

// DefaultBuilder::build(pool, ...): mini-trees, plus the reinsertion pass at Quality::High (default_builder.h:41-44, :65-73).
// MiniTreeBuilder::build(pool, bboxes, centers, config) itself (mini_tree_builder.h:29-58) with its own knobs; DefaultBuilder's
// three qualities are three settings of them (default_builder.h:65-73) plus the reinsertion pass at High.
template <typename T>
int build_minitree_explicit(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg, bool prune, T ratio,
                            bool optimize, uint32_t log2_grid, hipStream_t stream) {
    StreamScope scratch_on(stream);
    BVH_HIP_TRY(hipGetDevice(&out.device), BVH_AMD_ERR_HIP);
    DevBuf<HostNode<T>> final_nodes;
    DevBuf<uint32_t> final_ids;
    size_t total_nodes = 0;
    int rc = minitree_core<T>(d_bboxes, d_centers, n, cfg, prune, ratio, log2_grid, final_nodes, final_ids, total_nodes, stream);
    if (rc) return rc;
    if (optimize) {
        rc = reinsertion_optimize_device<T>(final_nodes.p, total_nodes, stream, 3);
        if (rc) return rc;
    }
    out.node_count = total_nodes;
    rc = finish_build<T>(out, final_nodes, final_ids.p, n, stream, /*take_ids=*/true);
    if (rc) return rc;
    final_ids.p = nullptr;
    return BVH_AMD_OK;
}

template <typename T>
int build_minitree_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg, hipStream_t stream) {
    return build_minitree_explicit<T>(out, d_bboxes, d_centers, n, cfg, cfg.quality != BVH_BUILD_QUALITY_LOW,
                                      cfg.quality == BVH_BUILD_QUALITY_HIGH ? T(0.01) : T(0.1), cfg.quality == BVH_BUILD_QUALITY_HIGH, kDefaultLog2Grid,
                                      stream);
}

template int build_minitree_explicit<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, bool, float, bool, uint32_t, hipStream_t);
template int build_minitree_explicit<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, bool, double, bool, uint32_t, hipStream_t);
template int build_minitree_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, hipStream_t);
template int build_minitree_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, hipStream_t);

} // namespace bvh_amd
