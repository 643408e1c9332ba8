// Bvh::serialize / Bvh::deserialize (reference bvh.h:221-243 over Node::serialize / deserialize, node.h:90-102) with the byte
// stream in DEVICE memory: [node_count][prim_count] as Index::Type, node_count x {bounds[2 dim], index}, prim_count x Index::Type.
// This is the payload of the one RCCL broadcast of the multi-GPU path: the building rank writes it straight from
// its resident nodes, every other rank turns the received buffer into resident nodes + traversal records — no byte of it visits
// the host (only the 16 + 56 bytes of header and root node that size the allocations).
#include "common.h"

#include <memory>

namespace bvh_amd {

namespace {

template <typename T> constexpr size_t node_bytes(int dim) { return 2 * size_t(dim) * sizeof(T) + sizeof(typename IndexOf<T>::Type); }

template <typename T>
__global__ void __launch_bounds__(64) k_wire_header(typename IndexOf<T>::Type* out, uint64_t nn, uint64_t np) {
    if (threadIdx.x == 0) { out[0] = static_cast<typename IndexOf<T>::Type>(nn); out[1] = static_cast<typename IndexOf<T>::Type>(np); }
}

// resident (three-wide) nodes <-> Node<T, 2> records of the stream; the stream is only aligned to sizeof(Index)
template <typename T>
__global__ void __launch_bounds__(256) k_wire_narrow_nodes(const HostNode<T>* nodes, size_t n, HostNode2<T>* out) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    HostNode2<T> r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.bounds[k] = nodes[i].bounds[k];
    r.index = nodes[i].index;
    out[i] = r;
}
template <typename T>
__global__ void __launch_bounds__(256) k_wire_widen_nodes(const HostNode2<T>* in, size_t n, HostNode<T>* nodes) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    HostNode<T> r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.bounds[k] = in[i].bounds[k];
    r.bounds[4] = r.bounds[5] = T(0);
    r.index = in[i].index;
    nodes[i] = r;
}
template <typename I>
__global__ void __launch_bounds__(256) k_wire_ids_out(const uint32_t* ids, size_t n, I* out) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i < n) out[i] = static_cast<I>(ids[i]);
}
template <typename I>
__global__ void __launch_bounds__(256) k_wire_ids_in(const I* in, size_t n, uint32_t* ids, uint32_t* bad) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    const I v = in[i];
    if (v >= (I{1} << 28)) atomicOr(bad, 4u);
    ids[i] = static_cast<uint32_t>(v);
}

// What the traversal records rely on (upload.hip: pair p = nodes[2p + 1], nodes[2p + 2]): every inner node's children are an
// adjacent pair that starts at an odd id inside the array, every leaf's range lies inside prim_ids. A stream from an untrusted
// source that breaks this would be walked with misaligned pairs / out-of-bounds reads; it is refused instead.
// `refs[p]` counts the inner nodes whose children are pair p. Node 0 belongs to no pair, so a cycle reachable from the root
// needs a pair with two parents: with no pair referenced more than once, whatever is reachable from the root is a tree, a walk
// from the root cannot run in a cycle, and the depth computed for the traversal stack bounds it. Pairs nobody references
// (nodes left unused after append_node / remove_last_node edits, hand-built arrays) are never visited and are tolerated, as
// the reference tolerates them.
template <typename T>
__global__ void __launch_bounds__(256) k_validate_nodes(const HostNode<T>* nodes, size_t nn, size_t np, uint32_t* refs, uint32_t* bad) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= nn) return;
    const auto index = nodes[i].index;
    const uint64_t first = static_cast<uint64_t>(index) >> kCountBits, count = static_cast<uint64_t>(index) & kCountMask;
    if (count == 0) {
        if ((first & 1u) == 0 || first + 1 >= nn) atomicOr(bad, 1u);
        else atomicAdd(&refs[first >> 1], 1u);
    } else if (first + count > np) atomicOr(bad, 2u);
}
__global__ void __launch_bounds__(256) k_validate_refs(const uint32_t* refs, size_t n_pairs, uint32_t* bad) {
    const size_t p = blockIdx.x * size_t{256} + threadIdx.x;
    if (p < n_pairs && refs[p] > 1u) atomicOr(bad, 8u);
}

template <typename T>
int check_nodes(const HostNode<T>* d_nodes, size_t nn, size_t np, uint32_t* d_bad, hipStream_t stream, const char* who) {
    const size_t n_pairs = (nn - 1) / 2;
    uint32_t* refs = nullptr;
    BVH_HIP_TRY(hipMalloc(&refs, std::max<size_t>(n_pairs, 1) * sizeof(uint32_t)), BVH_AMD_ERR_HIP);
    hipError_t e = hipMemsetAsync(refs, 0, std::max<size_t>(n_pairs, 1) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_validate_nodes<T>, dim3(static_cast<unsigned>((nn + 255) / 256)), dim3(256), 0, stream, d_nodes, nn, np, refs, d_bad);
    if (n_pairs) hipLaunchKernelGGL(k_validate_refs, dim3(static_cast<unsigned>((n_pairs + 255) / 256)), dim3(256), 0, stream, refs, n_pairs, d_bad);
    if (e == hipSuccess) e = hipGetLastError();
    uint32_t bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(refs);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    if (bad & 1u) return fail(BVH_AMD_ERR_ARG, std::string(who) + ": an inner node's children are not an adjacent pair at an odd index inside the node array");
    if (bad & 2u) return fail(BVH_AMD_ERR_ARG, std::string(who) + ": a leaf's primitive range lies outside prim_ids");
    if (bad & 8u) return fail(BVH_AMD_ERR_ARG, std::string(who) + ": some sibling pair is the child of several inner nodes (not a tree)");
    if (bad & 4u) return fail(BVH_AMD_ERR_UNSUPPORTED, std::string(who) + ": primitive id beyond 2^28 (32-bit device indices)");
    return BVH_AMD_OK;
}

} // namespace

template <typename T>
size_t wire_size(const BvhImpl<T>& b) {
    return 2 * sizeof(typename IndexOf<T>::Type) + b.node_count * node_bytes<T>(b.dim) + b.prim_count * sizeof(typename IndexOf<T>::Type);
}

template <typename T>
int validate_resident_nodes(const HostNode<T>* d_nodes, size_t nn, size_t np, hipStream_t stream, const char* who) {
    uint32_t* d_bad = nullptr;
    BVH_HIP_TRY(hipMalloc(&d_bad, sizeof(uint32_t)), BVH_AMD_ERR_HIP);
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
    int rc = e == hipSuccess ? check_nodes<T>(d_nodes, nn, np, d_bad, stream, who) : fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d_bad);
    return rc;
}

// `b.d_nodes` must be resident and current (capi.hip: make_nodes_resident). Returns the stream size; writes only if cap suffices.
template <typename T>
size_t serialize_to_device(const BvhImpl<T>& b, void* d_out, size_t cap, hipStream_t stream) {
    using I = typename IndexOf<T>::Type;
    const size_t need = wire_size(b);
    if (!d_out || cap < need) return need;
    if (!b.d_nodes || !b.d_prim_ids) { set_error("serialize_device: the BVH has no resident nodes"); return 0; }
    auto* p = static_cast<uint8_t*>(d_out);
    if (reinterpret_cast<uintptr_t>(p) % sizeof(I)) { set_error("serialize_device: the buffer must be aligned to the index type"); return 0; }
    hipLaunchKernelGGL(k_wire_header<T>, dim3(1), dim3(64), 0, stream, reinterpret_cast<I*>(p), uint64_t{b.node_count}, uint64_t{b.prim_count});
    p += 2 * sizeof(I);
    hipError_t e = hipSuccess;
    if (b.dim == 3) e = hipMemcpyAsync(p, b.d_nodes, b.node_count * sizeof(HostNode<T>), hipMemcpyDeviceToDevice, stream);
    else hipLaunchKernelGGL(k_wire_narrow_nodes<T>, dim3(static_cast<unsigned>((b.node_count + 255) / 256)), dim3(256), 0, stream, b.d_nodes, b.node_count,
                            reinterpret_cast<HostNode2<T>*>(p));
    p += b.node_count * node_bytes<T>(b.dim);
    if (e == hipSuccess && b.prim_count) {
        if constexpr (sizeof(I) == sizeof(uint32_t)) e = hipMemcpyAsync(p, b.d_prim_ids, b.prim_count * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        else hipLaunchKernelGGL(k_wire_ids_out<I>, dim3(static_cast<unsigned>((b.prim_count + 255) / 256)), dim3(256), 0, stream, b.d_prim_ids, b.prim_count,
                                reinterpret_cast<I*>(p));
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("serialize_device: ") + hipGetErrorString(e)); return 0; }
    return need;
}

template <typename T>
BvhImpl<T>* deserialize_from_device(const void* d_bytes, size_t size, int dim, hipStream_t stream) {
    using I = typename IndexOf<T>::Type;
    if (!d_bytes || size < 2 * sizeof(I)) { set_error("deserialize_device: truncated stream"); return nullptr; }
    if (reinterpret_cast<uintptr_t>(d_bytes) % sizeof(I)) { set_error("deserialize_device: the buffer must be aligned to the index type"); return nullptr; }
    const size_t nb = node_bytes<T>(dim);
    // header + root node: the only bytes of the stream the host looks at
    alignas(8) uint8_t head[2 * sizeof(I) + sizeof(HostNode<T>)] = {};
    const size_t head_bytes = std::min(size, 2 * sizeof(I) + nb);
    BVH_HIP_TRY_PTR(hipMemcpyAsync(head, d_bytes, head_bytes, hipMemcpyDeviceToHost, stream));
    BVH_HIP_TRY_PTR(hipStreamSynchronize(stream));
    I hdr[2];
    std::memcpy(hdr, head, sizeof(hdr));
    const size_t nn = hdr[0], np = hdr[1];
    if (nn == 0) { set_error("deserialize_device: empty BVH"); return nullptr; }
    if (nn % 2 == 0) { set_error("deserialize_device: node count must be odd (root + sibling pairs)"); return nullptr; }
    if (nn >= (size_t{1} << 28) || np >= (size_t{1} << 28)) { set_error("deserialize_device: more than 2^28 nodes/primitives (32-bit device indices)"); return nullptr; }
    if (size < 2 * sizeof(I) + nn * nb + np * sizeof(I)) { set_error("deserialize_device: truncated stream"); return nullptr; }
    auto b = std::make_unique<BvhImpl<T>>();
    b->dim = dim;
    b->node_count = nn; b->prim_count = np;
    b->host_valid = false;
    if (hipGetDevice(&b->device) != hipSuccess) { set_error("deserialize_device: no current device"); return nullptr; }
    const auto* p = static_cast<const uint8_t*>(d_bytes) + 2 * sizeof(I);
    BVH_HIP_TRY_PTR(hipMalloc(&b->d_nodes, nn * sizeof(HostNode<T>)));
    b->d_nodes_count = nn;
    if (dim == 3) BVH_HIP_TRY_PTR(hipMemcpyAsync(b->d_nodes, p, nn * sizeof(HostNode<T>), hipMemcpyDeviceToDevice, stream));
    else hipLaunchKernelGGL(k_wire_widen_nodes<T>, dim3(static_cast<unsigned>((nn + 255) / 256)), dim3(256), 0, stream,
                            reinterpret_cast<const HostNode2<T>*>(p), nn, b->d_nodes);
    p += nn * nb;
    BVH_HIP_TRY_PTR(hipMalloc(&b->d_prim_ids, std::max<size_t>(np, 1) * sizeof(uint32_t)));
    uint32_t* d_bad = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_bad, sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess && np)
        hipLaunchKernelGGL(k_wire_ids_in<I>, dim3(static_cast<unsigned>((np + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const I*>(p), np, b->d_prim_ids, d_bad);
    int rc = e == hipSuccess ? check_nodes<T>(b->d_nodes, nn, np, d_bad, stream, "deserialize_device") : fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d_bad);
    if (rc) return nullptr;
    if (relayout_on_device<T>(*b, b->d_nodes, stream)) return nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&b->d_work, size_t{BvhImpl<T>::kWorkSlots} * BvhImpl<T>::kWorkStride * sizeof(unsigned long long)));
    // root node from the bytes read above (a Node<T, 2> record has its index after four bounds)
    I root_index;
    std::memcpy(&root_index, head + 2 * sizeof(I) + 2 * size_t(dim) * sizeof(T), sizeof(I));
    b->root_index = static_cast<uint32_t>(root_index);
    T rb[6] = {0, 0, 0, 0, 0, 0};
    std::memcpy(rb, head + 2 * sizeof(I), 2 * size_t(dim) * sizeof(T));
    for (int k = 0; k < 6; ++k) b->root_bounds[k] = rb[k];
    BVH_HIP_TRY_PTR(hipStreamSynchronize(stream));            // the caller may free / reuse d_bytes when this returns
    return b.release();
}

template size_t wire_size<float>(const BvhImpl<float>&);
template size_t wire_size<double>(const BvhImpl<double>&);
template int validate_resident_nodes<float>(const HostNode<float>*, size_t, size_t, hipStream_t, const char*);
template int validate_resident_nodes<double>(const HostNode<double>*, size_t, size_t, hipStream_t, const char*);
template size_t serialize_to_device<float>(const BvhImpl<float>&, void*, size_t, hipStream_t);
template size_t serialize_to_device<double>(const BvhImpl<double>&, void*, size_t, hipStream_t);
template BvhImpl<float>* deserialize_from_device<float>(const void*, size_t, int, hipStream_t);
template BvhImpl<double>* deserialize_from_device<double>(const void*, size_t, int, hipStream_t);

} // namespace bvh_amd
