// Bvh::extract_bvh (bvh.h:92-122) on gfx950: the subtree under `root_id` re-laid out as its own BVH, bit-identical with
// the reference's sequential replay.
//
// The reference walks the subtree with an explicit stack, pushing (first, size) then (first + 1, size + 1): the RIGHT child
// is processed first, a node's children are allocated at nodes.size() when the node is processed, and a leaf's primitives
// are appended when it is processed. So for a node X processed after r(X) inner nodes and p(X) primitives:
//     children of X            -> destination ids 1 + 2 r(X), 2 + 2 r(X)
//     leaf X                   -> first_id = p(X)
//     right child R of X       -> r(R) = r(X) + 1,                  p(R) = p(X)
//     left  child L of X       -> r(L) = r(X) + 1 + inner(R),       p(L) = p(X) + prims(R)
// with inner(.) / prims(.) the number of inner nodes / primitives of a subtree. Those come from one bottom-up pass
// (arrival tickets, like k_refit); the assignment is a level-synchronous top-down sweep over the subtree.
#include "build_common.h"

namespace bvh_amd {

using namespace bld;

namespace {

constexpr uint32_t kNoParent = 0xFFFFFFFFu;

template <typename T> __device__ inline bool leaf_node(const HostNode<T>& n) { return (n.index & kCountMask) != 0; }
template <typename T> __device__ inline uint32_t first_id(const HostNode<T>& n) { return static_cast<uint32_t>(n.index >> kCountBits); }

template <typename T>
__global__ void __launch_bounds__(256) k_parents(const HostNode<T>* nodes, uint32_t n, uint32_t* parent) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == 0) parent[0] = 0;
    const HostNode<T> nd = nodes[i];
    if (!leaf_node(nd)) { parent[first_id(nd)] = i; parent[first_id(nd) + 1] = i; }
}

// inner[i], prims[i] for every node: one lane per leaf climbs, the second child to arrive at a node sums its children
template <typename T>
__global__ void __launch_bounds__(256) k_subtree_counts(const HostNode<T>* nodes, const uint32_t* parent, uint32_t n, uint32_t* arrived,
                                                        uint32_t* inner, uint32_t* prims) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HostNode<T> nd = nodes[i];
    if (!leaf_node(nd)) return;
    __hip_atomic_store(&inner[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&prims[i], static_cast<uint32_t>(nd.index & kCountMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i == 0) return;
    // (parent[] starts out as kNoParent everywhere: a node no inner node references — tolerated by the validation like the reference
    //  tolerates it — is the top of its own subtree and the climb ends there instead of following a stale word)
    uint32_t cur = parent[i];
    if (cur == kNoParent) return;
    for (;;) {
        ticket_release();                                     // this subtree's counts before the ticket (build_common.h)
        if (atomicAdd(&arrived[cur], 1u) == 0) return;        // first child: the sibling's lane finishes this node
        ticket_acquire();
        const uint32_t f = first_id(nodes[cur]);
        const uint32_t in = 1u + __hip_atomic_load(&inner[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               + __hip_atomic_load(&inner[f + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t pr = __hip_atomic_load(&prims[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          + __hip_atomic_load(&prims[f + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&inner[cur], in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&prims[cur], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) return;
        cur = parent[cur];
        if (cur == kNoParent) return;
    }
}

struct Item { uint32_t src, dst, r, p; };

template <typename T>
__global__ void __launch_bounds__(256) k_extract_level(const HostNode<T>* nodes, const uint32_t* ids, const uint32_t* inner, const uint32_t* prims,
                                                       const Item* in, uint32_t n_in, Item* out, uint32_t* n_out,
                                                       HostNode<T>* out_nodes, uint32_t* out_ids) {
    using I = typename IndexOf<T>::Type;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const Item it = in[i];
    HostNode<T> nd = nodes[it.src];
    if (leaf_node(nd)) {
        const uint32_t first = first_id(nd), cnt = static_cast<uint32_t>(nd.index & kCountMask);
        for (uint32_t q = 0; q < cnt; ++q) out_ids[it.p + q] = ids[first + q];
        nd.index = (static_cast<I>(it.p) << kCountBits) | cnt;
    } else {
        const uint32_t first = first_id(nd), kids = 1 + 2 * it.r;
        const uint32_t slot = atomicAdd(n_out, 2u);
        Item right; right.src = first + 1; right.dst = kids + 1; right.r = it.r + 1; right.p = it.p;
        Item left; left.src = first; left.dst = kids; left.r = it.r + 1 + inner[first + 1]; left.p = it.p + prims[first + 1];
        out[slot] = left;
        out[slot + 1] = right;
        nd.index = static_cast<I>(kids) << kCountBits;
    }
    out_nodes[it.dst] = nd;
}

} // namespace

// d_nodes / d_ids: the source BVH resident on the device. Fills `out` (resident; host mirror lazy).
template <typename T>
int extract_device(BvhImpl<T>& out, const HostNode<T>* d_nodes, size_t node_count, const uint32_t* d_ids, size_t root_id, hipStream_t stream) {
    StreamScope scratch_on(stream);
    if (root_id >= node_count) return fail(BVH_AMD_ERR_ARG, "extract: root_id out of range");
    const uint32_t n = static_cast<uint32_t>(node_count);
    BVH_HIP_TRY(hipGetDevice(&out.device), BVH_AMD_ERR_HIP);
    DevBuf<uint32_t> parent, arrived, inner, prims, counter, new_ids;
    DevBuf<Item> fa, fb;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(parent.alloc(n)); A(arrived.alloc(n)); A(inner.alloc(n)); A(prims.alloc(n)); A(counter.alloc(1));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("extract: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(arrived.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemsetAsync(parent.p, 0xFF, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);     // kNoParent: see k_subtree_counts
    BVH_HIP_TRY(hipMemsetAsync(inner.p, 0xFF, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);      // "never counted": nodes of a cycle no leaf pass completes
    hipLaunchKernelGGL(k_parents<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p);
    hipLaunchKernelGGL(k_subtree_counts<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, parent.p, n, arrived.p, inner.p, prims.p);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    uint32_t counts[2] = {0, 0};
    BVH_HIP_TRY(hipMemcpyAsync(&counts[0], inner.p + root_id, 4, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemcpyAsync(&counts[1], prims.p + root_id, 4, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    if (counts[0] == kNoParent) return fail(BVH_AMD_ERR_ARG, "extract: root_id is not the root of a subtree (its nodes form a cycle no leaf pass completes)");
    const size_t out_nodes_n = 1 + 2 * size_t{counts[0]}, out_prims = counts[1];
    DevBuf<HostNode<T>> new_nodes;
    A(new_nodes.alloc(out_nodes_n)); A(new_ids.alloc(out_prims));
    A(fa.alloc(out_nodes_n)); A(fb.alloc(out_nodes_n));       // a level never holds more items than the subtree has nodes
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("extract: hipMalloc: ") + hipGetErrorString(e));
    const Item first{static_cast<uint32_t>(root_id), 0u, 0u, 0u};
    BVH_HIP_TRY(hipMemcpyAsync(fa.p, &first, sizeof(Item), hipMemcpyHostToDevice, stream), BVH_AMD_ERR_HIP);
    uint32_t width = 1;
    Item* cur = fa.p; Item* nxt = fb.p;
    while (width) {
        BVH_HIP_TRY(hipMemsetAsync(counter.p, 0, 4, stream), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_extract_level<T>, dim3((width + 255) / 256), dim3(256), 0, stream, d_nodes, d_ids, inner.p, prims.p, cur, width, nxt,
                           counter.p, new_nodes.p, new_ids.p);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipMemcpyAsync(&width, counter.p, 4, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        std::swap(cur, nxt);
    }
    out.node_count = out_nodes_n;
    int rc = finish_build<T>(out, new_nodes, new_ids.p, out_prims, stream, /*take_ids=*/true);
    if (rc) return rc;
    new_ids.p = nullptr;
    return BVH_AMD_OK;
}

// How many nodes of the array the tree below node 0 holds (1 + 2 x its inner nodes). Equal to the node count for every BVH a builder
// made; smaller when the array also holds nodes nothing reachable references (tolerated on the way in, wire.hip).
template <typename T>
int reachable_node_count(const HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, size_t* out) {
    StreamScope scratch_on(stream);
    const uint32_t n = static_cast<uint32_t>(node_count);
    DevBuf<uint32_t> parent, arrived, inner, prims;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(parent.alloc(n)); A(arrived.alloc(n)); A(inner.alloc(n)); A(prims.alloc(n));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("reachable_node_count: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(arrived.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemsetAsync(parent.p, 0xFF, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemsetAsync(inner.p, 0xFF, 4, stream), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_parents<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p);
    hipLaunchKernelGGL(k_subtree_counts<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, parent.p, n, arrived.p, inner.p, prims.p);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    uint32_t inner_root = 0;
    BVH_HIP_TRY(hipMemcpyAsync(&inner_root, inner.p, 4, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    *out = inner_root == kNoParent ? 0 : 1 + 2 * size_t{inner_root};
    return BVH_AMD_OK;
}
template int reachable_node_count<float>(const HostNode<float>*, size_t, hipStream_t, size_t*);
template int reachable_node_count<double>(const HostNode<double>*, size_t, hipStream_t, size_t*);

template int extract_device<float>(BvhImpl<float>&, const HostNode<float>*, size_t, const uint32_t*, size_t, hipStream_t);
template int extract_device<double>(BvhImpl<double>&, const HostNode<double>*, size_t, const uint32_t*, size_t, hipStream_t);

} // namespace bvh_amd
