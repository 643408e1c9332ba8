// K10 — ReinsertionOptimizer (reinsertion_optimizer.h:19-267) on gfx950, bit-exact with the reference.
//
// Per iteration (3 by default, batch = 5 % of the nodes):
//   find_candidates (:88-105)  top-k nodes by half-area through a min-heap whose ARRAY LAYOUT (not just its set) is the
//                              input order of everything downstream, so libstdc++'s make_heap / pop_heap / push_heap are
//                              replayed exactly (stl_heap.h:134-148, :223-266, :339-362). It is inherently
//                              sequential: one wavefront runs it — 64 lanes stream and pre-filter the costs (only values
//                              above the current heap minimum can enter, and the minimum only grows), lane 0 sifts; the top
//                              16 K entries of the heap live in LDS, the rest in HBM.
//   find_reinsertion (:107-188) one lane per candidate: the reference's branch-and-bound walk with its explicit stack
//                              (same push order, same strict comparisons), read-only on the tree.
//   remove_if + std::sort by gain, descending (:254-256): stable compaction (scan) + the exact std::sort emulation on the
//                              negated gains (sort_emul.hip) — equal gains are common and their order decides conflicts.
//   greedy apply (:258-265)    sequential by nature (live parents, touched flags, refits up both paths): one lane.
#include "build_common.h"

namespace bvh_amd {

using namespace bld;

namespace {

constexpr int kSearchStack = 96;
template <typename T> struct HeapCap;                       // heap entries kept in LDS (cost + id)
template <> struct HeapCap<float>  { static constexpr uint32_t v = 16384; };
template <> struct HeapCap<double> { static constexpr uint32_t v = 8192; };   // 16-byte entries

struct Move { uint32_t from, to; };
struct ReScalars { uint32_t n_moves, error, pad[2]; };

template <typename T> __device__ inline T ha6(const T* b) {                  // bounds = {minx,maxx,miny,maxy,minz,maxz}
    const T d0 = b[1] - b[0], d1 = b[3] - b[2], d2 = b[5] - b[4];
    return (d0 + d1) * d2 + d0 * d1;
}
template <typename T> __device__ inline bool is_leaf(const HostNode<T>& n) { return (n.index & kCountMask) != 0; }
template <typename T> __device__ inline uint32_t first_of(const HostNode<T>& n) { return static_cast<uint32_t>(n.index >> kCountBits); }
__device__ inline uint32_t sibling_of(uint32_t id) { return (id & 1u) ? id + 1 : id - 1; }      // bvh.h:34-39
__device__ inline uint32_t left_of(uint32_t id) { return (id & 1u) ? id : id - 1; }              // bvh.h:43-45

// compute_parents (:72-86) + half-area of every node
template <typename T>
__global__ void __launch_bounds__(256) k_parents_costs(const HostNode<T>* nodes, uint32_t n, uint32_t* parent, T* cost, int with_parents) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HostNode<T> nd = nodes[i];
    cost[i] = ha6(nd.bounds);
    if (with_parents) {
        if (i == 0) parent[0] = 0;
        if (!is_leaf(nd)) { parent[first_of(nd)] = i; parent[first_of(nd) + 1] = i; }
    }
}

// ---- find_candidates: libstdc++ heap algorithms with comp = std::greater on cost ---------------------------------
// The whole wavefront works on ONE heap operation at a time, with wave-uniform control flow:
//   * __adjust_heap (stl_heap.h:223-248) = walk the min-child path from the hole to a leaf, shift that path up one
//     level, then __push_heap the saved value back up along the same path. The path is discovered FIVE LEVELS PER
//     MEMORY ROUND-TRIP: the 62 descendants of the current hole are loaded by 62 lanes at once and the five child
//     choices are made with readlane on registers. Path entry j stays in lane j, so the final layout
//     (entries 1..m* move up, the value lands at path[m*], m* = deepest entry with !(cost > value)) is ONE parallel
//     write round.
//   * __push_heap (stl_heap.h:134-148) from the last position: the ancestor chain of a FIXED position is loaded by
//     ~17 lanes at once; the shift is again one parallel write round.
// Heap entries {cost, id}: the first 16 K (float) / 8 K (double) in LDS, the rest in HBM.
template <typename T> struct Ent { T cost; uint32_t id; };

template <typename T>
struct WaveHeap {
    Ent<T>* lds; Ent<T>* glob; uint32_t cap;
    int cap_level;                                            // deepest level that lies completely in LDS
    __device__ Ent<T> get(uint32_t i) const { return i < cap ? lds[i] : glob[i]; }
    __device__ void set(uint32_t i, Ent<T> e) { if (i < cap) lds[i] = e; else glob[i] = e; }
};

// Stores of this wave before its later loads. Memory instructions of ONE wave are processed in program order by the
// LDS and by the vector memory pipeline (that is what makes store-then-load through a may-alias pointer work for a single
// lane); lanes of the wave hand entries to each other through memory, so only the COMPILER must be kept from reordering
// across the hand-off. No s_waitcnt is needed, which keeps store round-trips off the critical path.
__device__ inline void heap_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline double lane_value(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), l), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ inline uint32_t lane_value(uint32_t v, int l) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l)); }

// __adjust_heap(first, hole0, len, value) with comp(a, b) = a.cost > b.cost. All lanes call it with identical arguments.
template <typename T>
__device__ void wave_adjust_heap(WaveHeap<T>& h, uint32_t hole0, uint32_t len, Ent<T> value, int lane) {
    uint32_t my_pos = hole0;                                  // lane j: position of path entry j (lane 0: the hole itself)
    Ent<T> my_ent = value;                                    // lane j >= 1: the entry found there
    uint32_t cur = hole0;
    int depth = 0;                                            // path length so far
    const uint32_t limit = (len - 1) / 2;                     // nodes below `limit` have two children
    while (cur < limit) {
        // lanes 1..62 load the descendants of `cur` down five levels (BFS order inside the subtree)
        const int t = lane;
        const int d = 31 - __clz(t + 1);
        const uint32_t node = ((cur + 1) << d) - 1 + (static_cast<uint32_t>(t + 1) - (1u << d));
        Ent<T> e; e.cost = T(0); e.id = 0;
        if (t >= 1 && t <= 62 && node < len) e = h.get(node);
        int r = 0;                                            // BFS index of the current hole inside the loaded subtree
        // a round stops at the last LDS-resident level so that only one round per operation touches HBM
        const int level = 31 - __clz(cur + 1);
        const int steps = (level < h.cap_level && level + 5 > h.cap_level) ? h.cap_level - level : 5;   // >= 1
        for (int step = 0; step < steps && cur < limit; ++step) {
            const int c1 = 2 * r + 1, c2 = 2 * r + 2;
            const T k1 = lane_value(e.cost, c1), k2 = lane_value(e.cost, c2);
            const bool left = k2 > k1;                        // comp(second, second - 1): take the left child
            r = left ? c1 : c2;
            cur = left ? 2 * cur + 1 : 2 * cur + 2;
            ++depth;
            const T pc = lane_value(e.cost, r);
            const uint32_t pi = lane_value(e.id, r);
            if (lane == depth) { my_pos = cur; my_ent.cost = pc; my_ent.id = pi; }
        }
    }
    if ((len & 1u) == 0 && cur == (len - 2) / 2) {            // a last node with a single (left) child
        const uint32_t child = 2 * cur + 1;
        const Ent<T> e = h.get(child);
        cur = child;
        ++depth;
        if (lane == depth) { my_pos = cur; my_ent = e; }
    }
    // __push_heap(first, hole = path[depth], top = hole0, value): entries below m* stay, 1..m* move up, value at path[m*]
    const bool stays = lane >= 1 && lane <= depth && !(my_ent.cost > value.cost);
    const uint64_t mask = __ballot(stays);
    const int mstar = mask ? 63 - __clzll(static_cast<long long>(mask)) : 0;
    const uint32_t up_pos = __shfl_up(my_pos, 1);             // position of path entry j - 1
    if (lane >= 1 && lane <= mstar) h.set(up_pos, my_ent);
    if (lane == mstar) h.set(my_pos, value);
    heap_sync();
}

// __push_heap(first, hole = len - 1, top = 0, value)
template <typename T>
__device__ void wave_push_heap(WaveHeap<T>& h, uint32_t len, Ent<T> value, int lane) {
    const uint32_t pos = lane == 0 ? len - 1 : (len >> lane) - 1;       // lane j: j-th ancestor of len - 1
    const bool valid = lane >= 1 && (len >> lane) >= 1;
    Ent<T> e; e.cost = T(0); e.id = 0;
    if (valid) e = h.get(pos);
    const uint64_t above = __ballot(valid && e.cost > value.cost) >> 1;   // bit j-1: ancestor j is moved down
    const int moves = above == ~uint64_t{0} ? 64 : __ffsll(static_cast<long long>(~above)) - 1;   // leading run of ones
    const uint32_t below_pos = __shfl_up(pos, 1);
    if (lane >= 1 && lane <= moves) h.set(below_pos, e);
    if (lane == moves) h.set(pos, value);
    heap_sync();
}

template <typename T>
__global__ void __launch_bounds__(64) k_heap_select(const T* cost, uint32_t n_nodes, uint32_t target, Ent<T>* glob, uint32_t* out_ids) {
    extern __shared__ unsigned char heap_lds[];
    WaveHeap<T> h;
    h.lds = reinterpret_cast<Ent<T>*>(heap_lds);
    h.glob = glob;
    h.cap = HeapCap<T>::v;
    h.cap_level = 30 - __clz(static_cast<int>(HeapCap<T>::v));  // cap = 2^m entries: levels 0 .. m-1 are complete
    const int lane = threadIdx.x;
    const uint32_t head = min(n_nodes, target + 1);
    const uint32_t k = head - 1;                              // candidates 1 .. head-1  (:93-94)
    for (uint32_t j = lane; j < k; j += 64) { Ent<T> e; e.cost = cost[j + 1]; e.id = j + 1; h.set(j, e); }
    heap_sync();
    if (k == 0) return;
    if (k >= 2) {                                             // __make_heap (stl_heap.h:339-362)
        for (uint32_t parent = (k - 2) / 2;; --parent) {
            wave_adjust_heap(h, parent, k, h.get(parent), lane);
            if (parent == 0) break;
        }
    }
    for (uint32_t base = head; base < n_nodes; base += 64) {  // :96-103
        const uint32_t i = base + lane;
        const bool in = i < n_nodes;
        const T c = in ? cost[i] : T(0);
        uint64_t mask = __ballot(in && h.get(0).cost < c);    // the heap minimum only grows: a failed test stays failed
        while (mask) {
            const int j = __ffsll(static_cast<long long>(mask)) - 1;
            mask &= mask - 1;
            const T cj = lane_value(c, j);
            if (h.get(0).cost < cj) {
                if (k > 1) wave_adjust_heap(h, 0u, k - 1, h.get(k - 1), lane);   // std::pop_heap
                Ent<T> w; w.cost = cj; w.id = base + j;
                wave_push_heap(h, k, w, lane);                // back() = {i, cost}; std::push_heap
            }
        }
    }
    for (uint32_t j = lane; j < k; j += 64) out_ids[j] = h.get(j).id;
}

// ---- find_reinsertion (:107-188), one lane per candidate ----------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) k_search(const HostNode<T>* nodes, const uint32_t* parent, const uint32_t* cand, uint32_t k,
                                               Move* moves, T* gains, uint32_t* keep, ReScalars* sc) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= k) return;
    const uint32_t id = cand[c];
    T s_bound[kSearchStack]; uint32_t s_node[kSearchStack];
    int sp = 0;
    uint32_t best_to = 0; T best_gain = T(0);
    T self[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) self[q] = nodes[id].bounds[q];
    const T self_area = ha6(self);
    const uint32_t first_parent = parent[id];
    T gain_so_far = ha6(nodes[first_parent].bounds);
    uint32_t sib = sibling_of(id);
    T pivot_box[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) pivot_box[q] = nodes[sib].bounds[q];
    uint32_t pivot = first_parent;
    bool overflow = false;
    do {
        s_bound[sp] = gain_so_far; s_node[sp] = sib; ++sp;
        while (sp) {
            --sp;
            const T bound = s_bound[sp]; const uint32_t dst = s_node[sp];
            if (bound - self_area <= best_gain) continue;
            const HostNode<T> dn = nodes[dst];
            T merged[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {                     // dst.get_bbox().extend(node.get_bbox())
                merged[2 * q] = pick_min(dn.bounds[2 * q], self[2 * q]);
                merged[2 * q + 1] = pick_max(dn.bounds[2 * q + 1], self[2 * q + 1]);
            }
            const T gain = bound - ha6(merged);
            if (gain > best_gain) { best_to = dst; best_gain = gain; }
            if (!is_leaf(dn)) {
                const T child_bound = gain + ha6(dn.bounds);
                if (sp + 2 > kSearchStack) { overflow = true; break; }
                s_bound[sp] = child_bound; s_node[sp] = first_of(dn); ++sp;
                s_bound[sp] = child_bound; s_node[sp] = first_of(dn) + 1; ++sp;
            }
        }
        if (overflow) break;
        if (pivot != first_parent) {                          // :177-180
            const HostNode<T> sn = nodes[sib];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                pivot_box[2 * q] = pick_min(pivot_box[2 * q], sn.bounds[2 * q]);
                pivot_box[2 * q + 1] = pick_max(pivot_box[2 * q + 1], sn.bounds[2 * q + 1]);
            }
            gain_so_far += ha6(nodes[pivot].bounds) - ha6(pivot_box);
        }
        sib = sibling_of(pivot);
        pivot = parent[pivot];
    } while (pivot != 0);
    if (overflow) atomicOr(&sc->error, 1u);
    uint32_t from = id;
    if (best_to == sibling_of(id) || best_to == first_parent) { from = 0; best_to = 0; best_gain = T(0); }   // :184-186
    Move m; m.from = from; m.to = best_to;
    moves[c] = m;
    gains[c] = best_gain;
    keep[c] = best_gain <= T(0) ? 0u : 1u;                    // remove_if(area_diff <= 0), :254-255
}

template <typename T>
__global__ void __launch_bounds__(256) k_compact(const Move* moves, const T* gains, const uint32_t* keep, const uint32_t* off, uint32_t k,
                                                 Move* out_moves, T* out_neg_gain) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k || !keep[c]) return;
    out_moves[off[c]] = moves[c];
    out_neg_gain[off[c]] = -gains[c];                         // sort(greater) on gain == sort(less) on -gain
}

// ---- greedy apply (:258-265) with reinsert_node (:190-213) and refit_from (:215-225) ------------------------------------
template <typename T>
__device__ void refit_upwards(HostNode<T>* nodes, const uint32_t* parent, uint32_t i) {
    do {
        HostNode<T>& nd = nodes[i];
        if (!is_leaf(nd)) {
            const HostNode<T>& l = nodes[first_of(nd)];
            const HostNode<T>& r = nodes[first_of(nd) + 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) {                     // left.get_bbox().extend(right.get_bbox())
                nd.bounds[2 * q] = pick_min(l.bounds[2 * q], r.bounds[2 * q]);
                nd.bounds[2 * q + 1] = pick_max(l.bounds[2 * q + 1], r.bounds[2 * q + 1]);
            }
        }
        i = parent[i];
    } while (i != 0);
}

template <typename T>
__global__ void k_apply(HostNode<T>* nodes, uint32_t* parent, unsigned char* touched, const Move* moves, const uint32_t* order, uint32_t m) {
    using I = typename IndexOf<T>::Type;
    for (uint32_t j = 0; j < m; ++j) {
        const Move mv = moves[order[j]];
        const uint32_t from = mv.from, to = mv.to;
        const uint32_t hot[5] = { to, from, sibling_of(from), parent[to], parent[from] };     // get_conflicts (:227-234)
        bool clash = false;
        for (int q = 0; q < 5; ++q) clash = clash || touched[hot[q]];
        if (clash) continue;
        for (int q = 0; q < 5; ++q) touched[hot[q]] = 1;
        const uint32_t sib = sibling_of(from), par = parent[from];
        const HostNode<T> sib_node = nodes[sib], dst_node = nodes[to];
        nodes[to].index = static_cast<I>(left_of(from)) << kCountBits;
        nodes[sib] = dst_node;
        nodes[par] = sib_node;
        if (!is_leaf(dst_node)) { parent[first_of(dst_node)] = sib; parent[first_of(dst_node) + 1] = sib; }
        if (!is_leaf(sib_node)) { parent[first_of(sib_node)] = par; parent[first_of(sib_node) + 1] = par; }
        parent[sib] = to;
        parent[from] = to;
        refit_upwards(nodes, parent, to);
        refit_upwards(nodes, parent, par);
    }
}

// Bvh::refit (bvh.h:185-218): every inner box = left.bbox.extend(right.bbox), children before parents. One lane per
// leaf climbs; the second child to arrive at a node computes it (agent-scope fences order the hand-off across XCDs).
template <typename T>
__global__ void __launch_bounds__(256) k_refit(HostNode<T>* nodes, const uint32_t* parent, uint32_t* arrived, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || i == 0 || !is_leaf(nodes[i])) return;
    uint32_t cur = parent[i];
    for (;;) {
        __threadfence();                                      // release: this lane's box is visible before the ticket
        if (atomicAdd(&arrived[cur], 1u) == 0) return;        // first child: the sibling's lane finishes this node
        __threadfence();                                      // acquire: see the sibling's box
        HostNode<T>& nd = nodes[cur];
        const uint32_t f = first_of(nd);
        const T* l = nodes[f].bounds;
        const T* r = nodes[f + 1].bounds;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const T lo_l = __hip_atomic_load(&l[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), lo_r = __hip_atomic_load(&r[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const T hi_l = __hip_atomic_load(&l[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi_r = __hip_atomic_load(&r[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nd.bounds[2 * q], pick_min(lo_l, lo_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nd.bounds[2 * q + 1], pick_max(hi_l, hi_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cur == 0) return;
        cur = parent[cur];
    }
}

} // namespace

template <typename T>
int refit_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream) {
    const uint32_t n = static_cast<uint32_t>(node_count);
    if (n < 3) return BVH_AMD_OK;
    DevBuf<uint32_t> parent, arrived;
    DevBuf<T> cost;
    hipError_t e = parent.alloc(n);
    if (e == hipSuccess) e = arrived.alloc(n);
    if (e == hipSuccess) e = cost.alloc(n);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("refit: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(arrived.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_parents_costs<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p, cost.p, 1);
    hipLaunchKernelGGL(k_refit<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, parent.p, arrived.p, n);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}
template int refit_device<float>(HostNode<float>*, size_t, hipStream_t);
template int refit_device<double>(HostNode<double>*, size_t, hipStream_t);

// ReinsertionOptimizer::optimize on device-resident nodes (reference layout), in place.
template <typename T>
int reinsertion_optimize_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream) {
    const uint32_t n = static_cast<uint32_t>(node_count);
    if (n < 2) return BVH_AMD_OK;
    const T ratio = static_cast<T>(0.05);                     // Config (:19-25)
    const size_t iterations = 3;
    const uint32_t batch = static_cast<uint32_t>(std::max<size_t>(1, static_cast<size_t>(static_cast<T>(node_count) * ratio)));   // :238-239
    const uint32_t head = std::min<uint32_t>(n, batch + 1), k = head - 1;
    if (k == 0) return BVH_AMD_OK;

    DevBuf<uint32_t> parent, cand, keep, off, order;
    DevBuf<Ent<T>> heap_g;
    DevBuf<T> cost, gains, neg_gain;
    DevBuf<Move> moves, kept;
    DevBuf<unsigned char> touched;
    DevBuf<ReScalars> scalars;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(parent.alloc(n)); A(heap_g.alloc(k)); A(cand.alloc(k)); A(keep.alloc(k)); A(off.alloc(k)); A(order.alloc(k));
    A(cost.alloc(n)); A(gains.alloc(k)); A(neg_gain.alloc(k)); A(moves.alloc(k)); A(kept.alloc(k));
    A(touched.alloc(n)); A(scalars.alloc(1));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("optimize: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(scalars.p, 0, sizeof(ReScalars), stream), BVH_AMD_ERR_HIP);

    const size_t heap_lds = size_t{HeapCap<T>::v} * sizeof(Ent<T>);
    BVH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_heap_select<T>), hipFuncAttributeMaxDynamicSharedMemorySize, int(heap_lds)),
                BVH_AMD_ERR_HIP);
    for (size_t it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_parents_costs<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p, cost.p, it == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_heap_select<T>, dim3(1), dim3(64), heap_lds, stream, cost.p, n, batch, heap_g.p, cand.p);
        BVH_HIP_TRY(hipMemsetAsync(touched.p, 0, n, stream), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_search<T>, dim3((k + 63) / 64), dim3(64), 0, stream, d_nodes, parent.p, cand.p, k, moves.p, gains.p, keep.p, scalars.p);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        uint32_t m = 0;
        int rc = exclusive_scan_u32(keep.p, off.p, k, &m, stream);
        if (rc) return rc;
        if (m == 0) continue;
        hipLaunchKernelGGL(k_compact<T>, dim3((k + 255) / 256), dim3(256), 0, stream, moves.p, gains.p, keep.p, off.p, k, kept.p, neg_gain.p);
        rc = std_sort_ids<T>(order.p, neg_gain.p, m, 1, 0, 1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_apply<T>, dim3(1), dim3(1), 0, stream, d_nodes, parent.p, touched.p, kept.p, order.p, m);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    }
    ReScalars hs;
    BVH_HIP_TRY(hipMemcpyAsync(&hs, scalars.p, sizeof(hs), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    if (hs.error) return fail(BVH_AMD_ERR_OVERFLOW, "optimize: reinsertion search stack exceeded 96 entries");
    return BVH_AMD_OK;
}

template int reinsertion_optimize_device<float>(HostNode<float>*, size_t, hipStream_t);
template int reinsertion_optimize_device<double>(HostNode<double>*, size_t, hipStream_t);

} // namespace bvh_amd
