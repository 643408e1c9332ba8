// Device copy of a reference-layout BVH: sibling pairs re-laid out as aligned PairNode records.
#include "common.h"

#include <mutex>

namespace bvh_amd {

namespace {

// nodes: reference layout (28/56 B per node, node.h:31-37). Pair p holds nodes[2p+1], nodes[2p+2]
// (children are always allocated as adjacent pairs after the root, top_down_sah_builder.h:91-94, and the
// reinsertion optimizer keeps siblings adjacent, reinsertion_optimizer.h:190-213).
template <typename T>
__global__ void __launch_bounds__(256) relayout_pairs(const HostNode<T>* nodes, size_t pair_count, PairNode<T>* out) {
    size_t p = blockIdx.x * size_t{256} + threadIdx.x;
    if (p >= pair_count) return;
    const HostNode<T>& l = nodes[2 * p + 1];
    const HostNode<T>& r = nodes[2 * p + 2];
    PairNode<T> rec;
#pragma unroll
    for (int k = 0; k < 6; ++k) { rec.lb[k] = l.bounds[k]; rec.rb[k] = r.bounds[k]; }
    rec.li = static_cast<uint32_t>(l.index);
    rec.ri = static_cast<uint32_t>(r.index);
#pragma unroll
    for (size_t k = 0; k < sizeof(rec.pad) / 4; ++k) rec.pad[k] = 0;
    out[p] = rec;
}

// depth of every pair record by pointer jumping over the parent links: O(log depth) rounds, no host round trips
template <typename T>
__global__ void __launch_bounds__(256) k_depth_init(const PairNode<T>* pairs, uint32_t n_pairs, uint32_t root_pair, uint32_t* anc, uint32_t* dist) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    // (the root's children, level 1, keep distance 0; they are not necessarily pair 0: reinsertion may move them)
    const uint32_t li = pairs[p].li, ri = pairs[p].ri;
    const uint32_t lp = li >> (kCountBits + 1), rp = ri >> (kCountBits + 1);
    if ((li & kCountMask) == 0 && lp < n_pairs && lp != root_pair) { anc[lp] = p; dist[lp] = 1; }
    if ((ri & kCountMask) == 0 && rp < n_pairs && rp != root_pair) { anc[rp] = p; dist[rp] = 1; }
}
// every pair starts as its own ancestor at distance 0, so that pairs no parent points to (only possible in a malformed tree
// handed in through from_nodes / deserialize) never leave an uninitialised link behind
__global__ void __launch_bounds__(256) k_depth_identity(uint32_t n, uint32_t* anc, uint32_t* dist) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { anc[i] = i; dist[i] = 0; }
}
// (the last round takes the maximum over the pairs whose chain ends at the ROOT's pair only: pairs nobody reachable references —
//  tolerated by the validation, wire.hip — and cycles among them are never walked and must not deepen the traversal stack)
__global__ void __launch_bounds__(256) k_depth_jump(const uint32_t* anc, const uint32_t* dist, uint32_t n, uint32_t* anc_out, uint32_t* dist_out,
                                                    uint32_t* max_out, uint32_t root_pair) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = anc[i], d = dist[i] + dist[a], top = anc[a];
    anc_out[i] = top;
    dist_out[i] = d;
    if (max_out && top == root_pair) atomicMax(max_out, d);
    if (max_out && anc[top] != top) atomicOr(max_out + 1, 1u);      // a chain longer than the rounds so far: more rounds are needed
}

// Sum over the inner nodes of half-area(node) / half-area(root), in units of 2^-16 (integer, so that the sum does not depend on
// the order of the atomics): the number of pair records a random line through the root box is expected to fetch.
template <typename T>
__global__ void __launch_bounds__(256) k_expected_visits(const PairNode<T>* pairs, uint32_t n_pairs, double inv_root_area, unsigned long long* sum) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (p < n_pairs) {
        const PairNode<T>& q = pairs[p];
        auto share = [&](const T* b, uint32_t index) -> unsigned long long {
            if ((index & kCountMask) != 0) return 0;                              // a leaf fetches no further record
            const double x = double(b[1]) - double(b[0]), y = double(b[3]) - double(b[2]), z = double(b[5]) - double(b[4]);
            const double r = (x * y + y * z + z * x) * inv_root_area;
            return r > 0 ? static_cast<unsigned long long>(fmin(r, 1.0) * 65536.0) : 0;   // (NaN / empty boxes count nothing)
        };
        mine = share(q.lb, q.li) + share(q.rb, q.ri);
    }
    __shared__ unsigned long long part[4];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

} // namespace

template <typename T>
int tree_depth(const BvhImpl<T>& b, hipStream_t stream) {
    if (b.max_depth.load() >= 0) return BVH_AMD_OK;
    static std::mutex once;                                   // concurrent first traversals compute it once
    std::lock_guard<std::mutex> lock(once);
    if (b.max_depth.load() >= 0) return BVH_AMD_OK;
    if (b.pair_count == 0) { b.max_depth = 0; return BVH_AMD_OK; }
    const uint32_t n = static_cast<uint32_t>(b.pair_count);
    // (round 5: from the scratch block cache like every other temporary — a plain hipMalloc + hipFree pair, the latter a device
    //  synchronisation, was a good part of what the FIRST batch through a fresh tree paid over a settled one)
    StreamScope scratch_on(stream);
    ScratchTag buf_tag;
    uint32_t* buf = nullptr;
    BVH_HIP_TRY(scratch_alloc(reinterpret_cast<void**>(&buf), (size_t{4} * n + 4) * sizeof(uint32_t), &buf_tag), BVH_AMD_ERR_HIP);
    uint32_t *anc = buf, *dist = buf + n, *anc2 = buf + 2 * size_t{n}, *dist2 = buf + 3 * size_t{n};
    uint32_t* d_max = buf + 4 * size_t{n};                    // {max depth, pad, 64-bit sum of expected visits}: 16-byte aligned
    unsigned long long* d_visits = reinterpret_cast<unsigned long long*>(d_max + 2);
    hipError_t e = hipMemsetAsync(d_max, 0, 16, stream);
    const unsigned grid = (n + 255) / 256;
    {
        const double x = double(b.root_bounds[1]) - double(b.root_bounds[0]), y = double(b.root_bounds[3]) - double(b.root_bounds[2]),
                     z = b.dim == 3 ? double(b.root_bounds[5]) - double(b.root_bounds[4]) : 0.0;
        const double area = x * y + y * z + z * x;
        hipLaunchKernelGGL(k_expected_visits<T>, dim3(grid), dim3(256), 0, stream, b.d_pairs, n, area > 0 ? 1.0 / area : 0.0, d_visits);
    }
    hipLaunchKernelGGL(k_depth_identity, dim3(grid), dim3(256), 0, stream, n, anc, dist);
    hipLaunchKernelGGL(k_depth_init<T>, dim3(grid), dim3(256), 0, stream, b.d_pairs, n, b.root_index >> (kCountBits + 1), anc, dist);
    int rounds = 1;
    while ((uint64_t{1} << rounds) < uint64_t{n} + 1) ++rounds;        // after r rounds every chain of length <= 2^r is resolved
    // Builders make trees of a few dozen levels: seven rounds resolve every chain of up to 128 pairs, and the last of them reports
    // whether any chain is still unresolved (word 1); only then do the remaining rounds run (round 4: the first traversal of a tree
    // paid ~20 launches over all pairs for nothing)
    uint32_t words[4] = {0, 0, 0, 0};
    int done = 0;
    for (const int upto : {std::min(rounds, 7), rounds}) {
        if (done >= upto) break;
        if (done) e = hipMemsetAsync(d_max, 0, 8, stream);
        for (int r = done; r < upto; ++r) {
            hipLaunchKernelGGL(k_depth_jump, dim3(grid), dim3(256), 0, stream, anc, dist, n, anc2, dist2, r == upto - 1 ? d_max : nullptr,
                               b.root_index >> (kCountBits + 1));
            std::swap(anc, anc2); std::swap(dist, dist2);
        }
        done = upto;
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(words, d_max, 16, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess || words[1] == 0) break;                   // every chain ends at a fixed point: the depth is final
    }
    scratch_free(buf, buf_tag);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("tree_depth: ") + hipGetErrorString(e));
    const uint32_t deepest = words[0];
    b.expected_visits = static_cast<float>(static_cast<double>((static_cast<unsigned long long>(words[3]) << 32) | words[2]) / 65536.0);
    b.max_depth = static_cast<int>(deepest) + 1;              // pairs at distance d from the root pair hold nodes of level d + 1
    return BVH_AMD_OK;
}
template int tree_depth<float>(const BvhImpl<float>&, hipStream_t);
template int tree_depth<double>(const BvhImpl<double>&, hipStream_t);

template <typename T>
BvhImpl<T>::~BvhImpl() {
    if (device >= 0) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        if (d_pairs) { scratch_forget(d_pairs); (void)hipFree(d_pairs); }
        if (d_prim_ids) { scratch_forget(d_prim_ids); (void)hipFree(d_prim_ids); }
        if (d_work) (void)hipFree(d_work);
        for (hipEvent_t& e : work_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (PlanSearch& ps : plan_search) { if (ps.start) (void)hipEventDestroy(ps.start); if (ps.stop) (void)hipEventDestroy(ps.stop); ps.start = ps.stop = nullptr; }
        if (d_nodes) { scratch_forget(d_nodes); (void)hipFree(d_nodes); }
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
}

template <typename T>
int BvhImpl<T>::sync_host() const {
    if (host_valid.load(std::memory_order_acquire)) return BVH_AMD_OK;
    std::lock_guard<std::mutex> lock(host_mutex);
    if (host_valid.load(std::memory_order_acquire)) return BVH_AMD_OK;
    int cur = -1;
    BVH_HIP_TRY(hipGetDevice(&cur), BVH_AMD_ERR_HIP);
    if (cur != device) BVH_HIP_TRY(hipSetDevice(device), BVH_AMD_ERR_HIP);
    nodes.resize(node_count);
    std::vector<uint32_t> ids(prim_count);
    hipError_t e = hipMemcpy(nodes.data(), d_nodes, node_count * sizeof(HostNode<T>), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ids.data(), d_prim_ids, prim_count * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("sync_host: ") + hipGetErrorString(e));
    prim_ids.assign(ids.begin(), ids.end());
    nodes2_valid.store(false);                                // a 2D mirror, if any, is stale now
    host_valid.store(true, std::memory_order_release);
    return BVH_AMD_OK;
}

template <typename T>
int relayout_on_device(BvhImpl<T>& b, const HostNode<T>* d_nodes, hipStream_t stream) {
    b.pair_count = (b.node_count - 1) / 2;
    b.max_depth = -1;
    { std::lock_guard<std::mutex> lock(b.plan_mutex); for (int k = 0; k < 2; ++k) { b.launch_plan[k] = 0; b.plan_search[k].index = 0; b.plan_search[k].pending = false; b.plan_search[k].trying = -1; b.plan_search[k].dropped = 0; for (auto& c : b.plan_search[k].count) c = 0; for (auto& t : b.plan_search[k].ns_per_ray) t = 0.0f; for (auto& r : b.plan_search[k].rays_of) r = 0; } }
    if (b.d_pairs) { scratch_forget(b.d_pairs); (void)hipFree(b.d_pairs); b.d_pairs = nullptr; }
    if (b.pair_count) {
        // (from the stream-ordered pool when it is on: a plain hipMalloc of the 10M-triangle scene's 0.5 GB of records costs ~2 ms;
        //  BvhImpl releases it with hipFree, which accepts pool memory)
        StreamScope scratch_on(stream);
        ScratchTag tag;
        BVH_HIP_TRY(scratch_alloc(reinterpret_cast<void**>(&b.d_pairs), b.pair_count * sizeof(PairNode<T>), &tag), BVH_AMD_ERR_HIP);
        unsigned grid = static_cast<unsigned>((b.pair_count + 255) / 256);
        hipLaunchKernelGGL(relayout_pairs<T>, dim3(grid), dim3(256), 0, stream, d_nodes, b.pair_count, b.d_pairs);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    }
    return BVH_AMD_OK;
}

template <typename T>
int BvhImpl<T>::sync_host2() const {
    int rc = sync_host();
    if (rc) return rc;
    if (nodes2_valid.load(std::memory_order_acquire)) return BVH_AMD_OK;
    std::lock_guard<std::mutex> lock(host_mutex);
    if (!nodes2_valid.load(std::memory_order_acquire)) {
        nodes2.resize(nodes.size());
        for (size_t i = 0; i < nodes.size(); ++i) {
            for (int k = 0; k < 4; ++k) nodes2[i].bounds[k] = nodes[i].bounds[k];
            nodes2[i].index = nodes[i].index;
        }
        nodes2_valid.store(true, std::memory_order_release);
    }
    return BVH_AMD_OK;
}

template <typename T>
void BvhImpl<T>::widen_host() const {
    nodes.resize(nodes2.size());
    for (size_t i = 0; i < nodes2.size(); ++i) {
        for (int k = 0; k < 4; ++k) nodes[i].bounds[k] = nodes2[i].bounds[k];
        nodes[i].bounds[4] = nodes[i].bounds[5] = T(0);
        nodes[i].index = nodes2[i].index;
    }
}

// Uploads the host mirror (b.nodes, b.prim_ids) to the current device.
template <typename T>
int upload_bvh(BvhImpl<T>& b, hipStream_t stream) {
    if (b.nodes.empty()) return fail(BVH_AMD_ERR_ARG, "upload: empty BVH");
    b.node_count = b.nodes.size(); b.prim_count = b.prim_ids.size(); b.host_valid = true;
    for (int k = 0; k < 6; ++k) b.root_bounds[k] = b.nodes[0].bounds[k];
    if (b.nodes.size() % 2 == 0) return fail(BVH_AMD_ERR_ARG, "upload: node count must be odd (root + sibling pairs)");
    if (b.nodes.size() >= (size_t{1} << 28) || b.prim_ids.size() >= (size_t{1} << 28))
        return fail(BVH_AMD_ERR_UNSUPPORTED, "upload: more than 2^28 nodes/primitives (32-bit device indices)");
    BVH_HIP_TRY(hipGetDevice(&b.device), BVH_AMD_ERR_HIP);
    if (!b.d_work) BVH_HIP_TRY(hipMalloc(&b.d_work, size_t{BvhImpl<T>::kWorkSlots} * BvhImpl<T>::kWorkStride * sizeof(unsigned long long)), BVH_AMD_ERR_HIP);
    b.root_index = static_cast<uint32_t>(b.nodes[0].index);

    HostNode<T>* d_nodes = nullptr;
    BVH_HIP_TRY(hipMalloc(&d_nodes, b.nodes.size() * sizeof(HostNode<T>)), BVH_AMD_ERR_HIP);
    hipError_t e = hipMemcpyAsync(d_nodes, b.nodes.data(), b.nodes.size() * sizeof(HostNode<T>), hipMemcpyHostToDevice, stream);
    int rc = e == hipSuccess ? validate_resident_nodes<T>(d_nodes, b.nodes.size(), b.prim_ids.size(), stream, "upload") : fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    if (rc == BVH_AMD_OK) rc = relayout_on_device(b, d_nodes, stream);
    if (rc == BVH_AMD_OK) {
        std::vector<uint32_t> ids(b.prim_ids.size());
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = static_cast<uint32_t>(b.prim_ids[i]);
        if (b.d_prim_ids) { (void)hipFree(b.d_prim_ids); b.d_prim_ids = nullptr; }
        e = hipMalloc(&b.d_prim_ids, std::max<size_t>(ids.size(), 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(b.d_prim_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // `ids` and the staging copy die here
        if (e != hipSuccess) rc = fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(stream);
    (void)hipFree(d_nodes);
    return rc;
}

template struct BvhImpl<float>;
template struct BvhImpl<double>;
template int upload_bvh<float>(BvhImpl<float>&, hipStream_t);
template int upload_bvh<double>(BvhImpl<double>&, hipStream_t);
template int relayout_on_device<float>(BvhImpl<float>&, const HostNode<float>*, hipStream_t);
template int relayout_on_device<double>(BvhImpl<double>&, const HostNode<double>*, hipStream_t);

} // namespace bvh_amd
