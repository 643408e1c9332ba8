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
template <> struct HeapCap<double> { static constexpr uint32_t v = 8192; };

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
template <typename T>
struct HeapRef {
    T* lc; uint32_t* li;                                     // LDS part
    T* gc; uint32_t* gi;                                     // HBM part (indexed by absolute heap position)
    uint32_t cap;
    __device__ T cost(uint32_t i) const { return i < cap ? lc[i] : gc[i]; }
    __device__ uint32_t id(uint32_t i) const { return i < cap ? li[i] : gi[i]; }
    __device__ void set(uint32_t i, T c, uint32_t v) { if (i < cap) { lc[i] = c; li[i] = v; } else { gc[i] = c; gi[i] = v; } }
    __device__ void move(uint32_t dst, uint32_t src) { set(dst, cost(src), id(src)); }
};

template <typename T>
__device__ void heap_push_up(HeapRef<T>& h, long hole, long top, T vc, uint32_t vi) {          // __push_heap
    long parent = (hole - 1) / 2;
    while (hole > top && h.cost(parent) > vc) {              // comp(parent, value) = parent.cost > value.cost
        h.move(hole, parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h.set(hole, vc, vi);
}
template <typename T>
__device__ void heap_sift(HeapRef<T>& h, long hole, long len, T vc, uint32_t vi) {              // __adjust_heap
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (h.cost(second) > h.cost(second - 1)) second--;  // comp(second, second - 1)
        h.move(hole, second);
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        h.move(hole, second - 1);
        hole = second - 1;
    }
    heap_push_up(h, hole, top, vc, vi);
}

template <typename T>
__global__ void __launch_bounds__(64) k_heap_select(const T* cost, uint32_t n_nodes, uint32_t target, T* gc, uint32_t* gi, uint32_t* out_ids) {
    extern __shared__ unsigned char heap_lds[];
    const uint32_t cap = HeapCap<T>::v;
    HeapRef<T> h;
    h.lc = reinterpret_cast<T*>(heap_lds);
    h.li = reinterpret_cast<uint32_t*>(heap_lds + size_t{cap} * sizeof(T));
    h.gc = gc; h.gi = gi; h.cap = cap;
    const int lane = threadIdx.x;
    const uint32_t head = min(n_nodes, target + 1);
    const uint32_t k = head - 1;                              // candidates 1 .. head-1  (:93-94)
    for (uint32_t j = lane; j < k; j += 64) h.set(j, cost[j + 1], j + 1);
    __syncthreads();
    if (k == 0) return;
    if (lane == 0 && k >= 2) {                                // __make_heap
        for (long parent = (long(k) - 2) / 2;; --parent) {
            heap_sift(h, parent, long(k), h.cost(parent), h.id(parent));
            if (parent == 0) break;
        }
    }
    __syncthreads();
    for (uint32_t base = head; base < n_nodes; base += 64) {  // :96-103
        const uint32_t i = base + lane;
        const bool in = i < n_nodes;
        const T c = in ? cost[i] : T(0);
        T hmin = h.cost(0);
        hmin = __shfl(hmin, 0);
        uint64_t mask = __ballot(in && hmin < c);             // the heap minimum only grows: a failed test stays failed
        while (mask) {
            const int j = __ffsll(static_cast<long long>(mask)) - 1;
            mask &= mask - 1;
            const T cj = __shfl(c, j);
            if (lane == 0 && h.cost(0) < cj) {
                if (k > 1) {                                   // std::pop_heap: value = heap[k-1]; heap[k-1] = heap[0]; sift
                    const T vc = h.cost(k - 1); const uint32_t vi = h.id(k - 1);
                    h.move(k - 1, 0);
                    heap_sift(h, 0, long(k) - 1, vc, vi);
                }
                heap_push_up(h, long(k) - 1, 0, cj, base + j); // back() = {i, cost}; std::push_heap
            }
        }
    }
    __syncthreads();
    for (uint32_t j = lane; j < k; j += 64) out_ids[j] = h.id(j);
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

} // namespace

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

    DevBuf<uint32_t> parent, heap_i, cand, keep, off, order;
    DevBuf<T> cost, heap_c, gains, neg_gain;
    DevBuf<Move> moves, kept;
    DevBuf<unsigned char> touched;
    DevBuf<ReScalars> scalars;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(parent.alloc(n)); A(heap_i.alloc(k)); A(cand.alloc(k)); A(keep.alloc(k)); A(off.alloc(k)); A(order.alloc(k));
    A(cost.alloc(n)); A(heap_c.alloc(k)); A(gains.alloc(k)); A(neg_gain.alloc(k)); A(moves.alloc(k)); A(kept.alloc(k));
    A(touched.alloc(n)); A(scalars.alloc(1));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("optimize: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(scalars.p, 0, sizeof(ReScalars), stream), BVH_AMD_ERR_HIP);

    const size_t heap_lds = size_t{HeapCap<T>::v} * (sizeof(T) + sizeof(uint32_t));
    BVH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_heap_select<T>), hipFuncAttributeMaxDynamicSharedMemorySize, int(heap_lds)),
                BVH_AMD_ERR_HIP);
    for (size_t it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_parents_costs<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p, cost.p, it == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_heap_select<T>, dim3(1), dim3(64), heap_lds, stream, cost.p, n, batch, heap_c.p, heap_i.p, cand.p);
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
