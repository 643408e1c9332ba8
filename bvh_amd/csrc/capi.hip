// The C-ABI of libbvh_amd.so (include/bvh_amd.h). Thin: argument checks, handle ownership, host mirror
// bookkeeping; all compute is in the HIP kernels (traverse.hip, prep.hip, build_*.hip). No CPU fallback.
#include "common.h"

#include <cstring>
#include <memory>

namespace bvh_amd {

template <typename T> int reinsertion_optimize_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim);
template <typename T>
int reinsertion_optimize_config_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim, double batch_size_ratio, size_t iterations);
void reinsertion_stats(unsigned out[2]);
void last_optimize_profile(bvh_amd_optimize_profile* out);
template <typename T>
int build_minitree_explicit(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg, bool prune, T ratio,
                            bool optimize, uint32_t log2_grid, hipStream_t stream);
template <typename T>
int extract_device(BvhImpl<T>& out, const HostNode<T>* d_nodes, size_t node_count, const uint32_t* d_ids, size_t root_id, hipStream_t stream);
template <typename T> int refit_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream);
template <typename T> int reachable_node_count(const HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, size_t* out);
template <typename T> int relayout_on_device(BvhImpl<T>& b, const HostNode<T>* d_nodes, hipStream_t stream);

namespace {
thread_local std::string g_error;
}

void set_error(const std::string& msg) { g_error = msg; }
int fail(int code, const std::string& msg) { g_error = msg; return code; }
std::string current_error() { return g_error; }

namespace {

struct ThreadPoolTag { size_t thread_count; };   // opaque `bvh_thread_pool`: only its presence matters

template <typename T> struct CTypes;
template <> struct CTypes<float>  { using Bvh = bvh3f; using Node = bvh_node3f; using BBox = bvh_bbox3f; using Vec = bvh_vec3f; using Ray = bvh_ray3f; };
template <> struct CTypes<double> { using Bvh = bvh3d; using Node = bvh_node3d; using BBox = bvh_bbox3d; using Vec = bvh_vec3d; using Ray = bvh_ray3d; };

template <typename T> BvhImpl<T>* impl(typename CTypes<T>::Bvh* b) { return reinterpret_cast<BvhImpl<T>*>(b); }
template <typename T> const BvhImpl<T>* impl(const typename CTypes<T>::Bvh* b) { return reinterpret_cast<const BvhImpl<T>*>(b); }
template <typename T> typename CTypes<T>::Bvh* handle(BvhImpl<T>* b) { return reinterpret_cast<typename CTypes<T>::Bvh*>(b); }

bvh_build_config default_config() {               // default_builder.h:23-30, top_down_sah_builder.h:27-40
    bvh_build_config c;
    c.quality = BVH_BUILD_QUALITY_HIGH;
    c.min_leaf_size = 1;
    c.max_leaf_size = 8;
    c.parallel_threshold = 1024;
    return c;
}

// Runs `build` with the caller's SplitHeuristic in force (NULL = the reference's default {0, 1}).
template <typename Build>
auto with_sah(const bvh_amd_sah_config* sah, Build&& build) -> decltype(build()) {
    SahParams p;
    if (sah) {
        if (sah->log_cluster_size >= 64) { set_error("build: sah.log_cluster_size must be below 64 (split_heuristic.h:21, make_bitmask<size_t>)"); return nullptr; }
        p.log_cluster = static_cast<uint32_t>(sah->log_cluster_size);
        p.cost_ratio = sah->cost_ratio;
    }
    SahScope scope(p);
    return build();
}

// BinnedSahBuilder<Node, BinCount>::build (binned_sah_builder.h:18, :32-38) with a BinCount other than the reference's default
template <typename Build>
auto with_bins(const bvh_amd_sah_config* sah, size_t bin_count, Build&& build) -> decltype(build()) {
    if (bin_count != 4 && bin_count != 8 && bin_count != 16 && bin_count != 32) {
        set_error("build: bin_count must be 4, 8, 16 or 32 (BinnedSahBuilder's BinCount, binned_sah_builder.h:18)");
        return nullptr;
    }
    return with_sah(sah, [&] { ambient_sah().bin_count = static_cast<uint32_t>(bin_count); return build(); });
}

template <typename T>
typename CTypes<T>::Bvh* build_device(const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config* config,
                                      bvh_amd_builder builder, void* stream)
{
    if (!d_bboxes || !d_centers || n == 0) { set_error("build: empty input (the reference's behaviour is undefined for 0 primitives)"); return nullptr; }
    bvh_build_config cfg = config ? *config : default_config();
    if (cfg.min_leaf_size < 1 || cfg.min_leaf_size > cfg.max_leaf_size || cfg.max_leaf_size > 15) {
        set_error("build: need 1 <= min_leaf_size <= max_leaf_size <= 15 (4-bit primitive count, index.h:38)");
        return nullptr;
    }
    auto b = std::make_unique<BvhImpl<T>>();
    if (build_on_device<T>(*b, d_bboxes, d_centers, n, cfg, builder, static_cast<hipStream_t>(stream)) != BVH_AMD_OK)
        return nullptr;
    return handle<T>(b.release());
}

template <typename T>
typename CTypes<T>::Bvh* build_minitree(const T* d_bboxes, const T* d_centers, size_t n, const bvh_amd_minitree_config* config, void* stream) {
    if (!d_bboxes || !d_centers || n == 0) { set_error("build: empty input"); return nullptr; }
    bvh_amd_minitree_config c = config ? *config : bvh_amd_minitree_config{1, 8, 1, 0.01, 1024, 4, 0, 1.0};
    if (c.min_leaf_size < 1 || c.min_leaf_size > c.max_leaf_size || c.max_leaf_size > 15) {
        set_error("build: need 1 <= min_leaf_size <= max_leaf_size <= 15 (4-bit primitive count, index.h:38)");
        return nullptr;
    }
    if (c.log2_grid_dim < 1 || c.log2_grid_dim > 10) {       // mini_tree_builder.h:169 asserts <= digits(MortonCode) / 3; 0 would be one cell
        set_error("build_minitree: log2_grid_dim must be in [1, 10] (three coordinates in a 32-bit Morton code, mini_tree_builder.h:169)");
        return nullptr;
    }
    bvh_build_config cfg = default_config();
    cfg.min_leaf_size = c.min_leaf_size; cfg.max_leaf_size = c.max_leaf_size; cfg.parallel_threshold = c.parallel_threshold;
    if (c.log_cluster_size >= 64) { set_error("build: log_cluster_size must be below 64 (split_heuristic.h:21)"); return nullptr; }
    SahParams sah;
    sah.log_cluster = static_cast<uint32_t>(c.log_cluster_size); sah.cost_ratio = c.cost_ratio;
    SahScope scope(sah);
    auto b = std::make_unique<BvhImpl<T>>();
    if (build_minitree_explicit<T>(*b, d_bboxes, d_centers, n, cfg, c.enable_pruning != 0, static_cast<T>(c.pruning_area_ratio), false,
                                   static_cast<uint32_t>(c.log2_grid_dim), static_cast<hipStream_t>(stream)) != BVH_AMD_OK)
        return nullptr;
    return handle<T>(b.release());
}

template <typename T>
typename CTypes<T>::Bvh* build_host(bvh_thread_pool* pool, const typename CTypes<T>::BBox* bboxes,
                                    const typename CTypes<T>::Vec* centers, size_t n, const bvh_build_config* config)
{
    if (!bboxes || !centers || n == 0) { set_error("build: empty input"); return nullptr; }
    static_assert(sizeof(typename CTypes<T>::BBox) == 6 * sizeof(T) && sizeof(typename CTypes<T>::Vec) == 3 * sizeof(T));
    T *d_bb = nullptr, *d_cc = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_bb, n * 6 * sizeof(T)));
    hipError_t e = hipMalloc(&d_cc, n * 3 * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(d_bb, bboxes, n * 6 * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cc, centers, n * 3 * sizeof(T), hipMemcpyHostToDevice);
    typename CTypes<T>::Bvh* out = nullptr;
    if (e == hipSuccess)
        out = build_device<T>(d_bb, d_cc, n, config, pool ? BVH_AMD_BUILDER_DEFAULT_PARALLEL : BVH_AMD_BUILDER_DEFAULT_SERIAL, nullptr);
    else
        set_error(std::string("build: ") + hipGetErrorString(e));
    (void)hipDeviceSynchronize();
    (void)hipFree(d_bb);
    if (d_cc) (void)hipFree(d_cc);
    return out;
}

template <typename T>
typename CTypes<T>::Bvh* from_nodes(const void* nodes, size_t nn, const size_t* prim_ids, size_t np) {
    if (!nodes || nn == 0 || (!prim_ids && np)) { set_error("from_nodes: null/empty input"); return nullptr; }
    auto b = std::make_unique<BvhImpl<T>>();
    b->nodes.resize(nn);
    std::memcpy(b->nodes.data(), nodes, nn * sizeof(HostNode<T>));
    b->prim_ids.assign(prim_ids, prim_ids + np);
    if (upload_bvh<T>(*b, nullptr) != BVH_AMD_OK) return nullptr;
    return handle<T>(b.release());
}

// Byte stream of Bvh::serialize (bvh.h:221-229): [node_count][prim_count] in Index::Type, nodes, prim ids.
template <typename T>
size_t stream_size(const BvhImpl<T>& b) {
    using I = typename IndexOf<T>::Type;
    return 2 * sizeof(I) + b.node_count * sizeof(HostNode<T>) + b.prim_count * sizeof(I);
}

template <typename T>
size_t serialize(const BvhImpl<T>& b, void* out, size_t cap) {
    using I = typename IndexOf<T>::Type;
    size_t need = stream_size(b);
    if (!out || cap < need) return need;
    if (b.sync_host() != BVH_AMD_OK) return 0;
    auto p = static_cast<uint8_t*>(out);
    I hdr[2] = { static_cast<I>(b.nodes.size()), static_cast<I>(b.prim_ids.size()) };
    std::memcpy(p, hdr, sizeof(hdr)); p += sizeof(hdr);
    std::memcpy(p, b.nodes.data(), b.nodes.size() * sizeof(HostNode<T>)); p += b.nodes.size() * sizeof(HostNode<T>);
    for (size_t id : b.prim_ids) { I v = static_cast<I>(id); std::memcpy(p, &v, sizeof(v)); p += sizeof(v); }
    return need;
}

template <typename T>
typename CTypes<T>::Bvh* deserialize(const void* bytes, size_t size) {
    using I = typename IndexOf<T>::Type;
    if (!bytes || size < 2 * sizeof(I)) { set_error("deserialize: truncated stream"); return nullptr; }
    auto p = static_cast<const uint8_t*>(bytes);
    I hdr[2];
    std::memcpy(hdr, p, sizeof(hdr)); p += sizeof(hdr);
    size_t nn = hdr[0], np = hdr[1];
    if (nn > size / sizeof(HostNode<T>) || np > size / sizeof(I) ||           // (bounded first: a crafted header must not overflow the sum)
        size < 2 * sizeof(I) + nn * sizeof(HostNode<T>) + np * sizeof(I)) { set_error("deserialize: truncated stream"); return nullptr; }
    auto b = std::make_unique<BvhImpl<T>>();
    b->nodes.resize(nn);
    std::memcpy(b->nodes.data(), p, nn * sizeof(HostNode<T>)); p += nn * sizeof(HostNode<T>);
    b->prim_ids.resize(np);
    for (size_t i = 0; i < np; ++i) { I v; std::memcpy(&v, p, sizeof(v)); p += sizeof(v); b->prim_ids[i] = static_cast<size_t>(v); }
    if (upload_bvh<T>(*b, nullptr) != BVH_AMD_OK) return nullptr;
    return handle<T>(b.release());
}

template <typename T>
void save(const BvhImpl<T>& b, FILE* f) {
    std::vector<uint8_t> buf(stream_size(b));
    serialize(b, buf.data(), buf.size());
    fwrite(buf.data(), 1, buf.size(), f);
}

template <typename T>
typename CTypes<T>::Bvh* load(FILE* f) {
    using I = typename IndexOf<T>::Type;
    I hdr[2] = {0, 0};
    if (fread(hdr, sizeof(I), 2, f) != 2) { set_error("load: truncated stream"); return nullptr; }
    std::vector<uint8_t> buf;
    try { buf.resize(2 * sizeof(I) + size_t(hdr[0]) * sizeof(HostNode<T>) + size_t(hdr[1]) * sizeof(I)); }
    catch (const std::exception&) { set_error("load: the header asks for more memory than there is"); return nullptr; }   // (never across the C ABI)
    std::memcpy(buf.data(), hdr, sizeof(hdr));
    size_t rest = buf.size() - sizeof(hdr);
    if (fread(buf.data() + sizeof(hdr), 1, rest, f) != rest) { set_error("load: truncated stream"); return nullptr; }
    return deserialize<T>(buf.data(), buf.size());
}

// The reference-layout nodes resident on the device, with possible host-side edits pushed.
template <typename T>
int make_nodes_resident(BvhImpl<T>& b) {
    if (b.node_count == 0) return fail(BVH_AMD_ERR_ARG, "empty bvh");
    if (b.dim == 2 && b.host_valid && b.nodes2_valid) b.widen_host();    // 2D: the caller edits the narrow mirror
    const size_t bytes = b.node_count * sizeof(HostNode<T>);
    if (b.d_nodes && b.d_nodes_count != b.node_count) {        // nodes were appended / removed on the host
        (void)hipFree(b.d_nodes);
        b.d_nodes = nullptr;
    }
    if (!b.d_nodes) {                                          // BVH came from the host (from_nodes / load): make it resident
        if (!b.host_valid || b.nodes.size() != b.node_count) return fail(BVH_AMD_ERR_ARG, "the BVH has neither a resident nor a host copy of its nodes");
        BVH_HIP_TRY(hipMalloc(&b.d_nodes, bytes), BVH_AMD_ERR_HIP);
        b.d_nodes_count = b.node_count;
        BVH_HIP_TRY(hipMemcpy(b.d_nodes, b.nodes.data(), bytes, hipMemcpyHostToDevice), BVH_AMD_ERR_HIP);
    } else if (b.host_valid) {                                 // the host mirror may have been edited through bvh_node* pointers
        BVH_HIP_TRY(hipMemcpy(b.d_nodes, b.nodes.data(), bytes, hipMemcpyHostToDevice), BVH_AMD_ERR_HIP);
    } else {
        return BVH_AMD_OK;                                     // resident nodes written by the device builders / a validated upload
    }
    // whatever came from the host mirror may have been edited by the caller (bvh_node*_set_*): the kernels that walk these nodes
    // (refit, optimize, extract, the traversal records built from them) rely on the structure wire.hip checks
    return validate_resident_nodes<T>(b.d_nodes, b.node_count, b.prim_count, nullptr, "sync (host mirror -> device)");
}

// Bvh::serialize into / Bvh::deserialize out of DEVICE memory (wire.hip): the broadcast payload never visits the host.
template <typename T>
size_t serialize_device(BvhImpl<T>* pb, void* d_out, size_t cap, void* stream) {
    if (!pb) { set_error("serialize_device: null bvh"); return 0; }
    const size_t need = wire_size<T>(*pb);
    if (!d_out || cap < need) return need;
    if (make_nodes_resident<T>(*pb)) return 0;
    return serialize_to_device<T>(*pb, d_out, cap, static_cast<hipStream_t>(stream));
}

// Runs `op` (optimize / refit) on the resident reference-layout nodes, then refreshes the traversal records and (if it was
// valid) the host mirror.
template <typename T, typename Op>
int on_resident_nodes(BvhImpl<T>* pb, Op op) {
    if (!pb) return fail(BVH_AMD_ERR_ARG, "null bvh");
    BvhImpl<T>& b = *pb;
    int rc = make_nodes_resident<T>(b);
    if (rc) return rc;
    const size_t bytes = b.node_count * sizeof(HostNode<T>);
    rc = op(b.d_nodes, b.node_count);
    if (rc) return rc;
    rc = relayout_on_device<T>(b, b.d_nodes, nullptr);
    if (rc) return rc;
    HostNode<T> root;
    BVH_HIP_TRY(hipMemcpy(&root, b.d_nodes, sizeof(root), hipMemcpyDeviceToHost), BVH_AMD_ERR_HIP);
    b.root_index = static_cast<uint32_t>(root.index);
    for (int k = 0; k < 6; ++k) b.root_bounds[k] = root.bounds[k];
    if (b.host_valid) BVH_HIP_TRY(hipMemcpy(b.nodes.data(), b.d_nodes, bytes, hipMemcpyDeviceToHost), BVH_AMD_ERR_HIP);
    b.nodes2_valid = false;
    return BVH_AMD_OK;
}

template <typename T> BvhImpl<T>* extract(BvhImpl<T>* pb, size_t root_id) {
    if (!pb) { set_error("extract: null bvh"); return nullptr; }
    BvhImpl<T>& b = *pb;
    if (make_nodes_resident<T>(b)) return nullptr;
    if (!b.d_prim_ids) { set_error("extract: the BVH has no device prim ids"); return nullptr; }
    auto out = std::make_unique<BvhImpl<T>>();
    out->dim = b.dim;
    if (extract_device<T>(*out, b.d_nodes, b.node_count, b.d_prim_ids, root_id, nullptr)) return nullptr;
    return out.release();
}

// The ReinsertionOptimizer walks parent links from every candidate to the root (reinsertion_optimizer.h:107-188). On an array that
// also holds nodes nothing reachable references — tolerated on the way in (wire.hip), like the reference tolerates them — the
// reference's zero-initialised parents_ (:74) would make such a node a child of the root and corrupt the tree, and a cycle among
// unreachable nodes would never reach the root at all. Refused here, before anything is touched: optimize wants a proper tree.
template <typename T> int whole_array_is_one_tree(const HostNode<T>* d, size_t n) {
    size_t reachable = 0;
    const int rc = reachable_node_count<T>(d, n, nullptr, &reachable);
    if (rc) return rc;
    if (reachable != n)
        return fail(BVH_AMD_ERR_UNSUPPORTED, "optimize: " + std::to_string(n - reachable) + " of the " + std::to_string(n) +
                    " nodes are not reachable from the root (unused sibling pairs); extract_bvh(0) first, or remove them");
    return BVH_AMD_OK;
}
template <typename T> int optimize(BvhImpl<T>* b) {
    const int dim = b ? b->dim : 3;
    return on_resident_nodes<T>(b, [dim](HostNode<T>* d, size_t n) {
        const int rc = whole_array_is_one_tree<T>(d, n);
        return rc ? rc : reinsertion_optimize_device<T>(d, n, nullptr, dim);
    });
}
template <typename T> int optimize_config(BvhImpl<T>* b, const bvh_amd_optimize_config* config) {
    const int dim = b ? b->dim : 3;
    const bvh_amd_optimize_config c = config ? *config : bvh_amd_optimize_config{0.05, 3};
    return on_resident_nodes<T>(b, [dim, c](HostNode<T>* d, size_t n) {
        const int rc = whole_array_is_one_tree<T>(d, n);
        return rc ? rc : reinsertion_optimize_config_device<T>(d, n, nullptr, dim, c.batch_size_ratio, c.max_iter_count);
    });
}
template <typename T> int refit(BvhImpl<T>* b) {
    return on_resident_nodes<T>(b, [](HostNode<T>* d, size_t n) { return refit_device<T>(d, n, nullptr); });
}

// Host-side edits (bvh_node* setters, append/remove) live in the mirror; this pushes them to the device copy.
template <typename T> int sync_device(BvhImpl<T>* pb) {
    if (!pb) return fail(BVH_AMD_ERR_ARG, "sync_device: null bvh");
    BvhImpl<T>& b = *pb;
    int rc = b.dim == 2 ? b.sync_host2() : b.sync_host();
    if (rc) return rc;
    if (b.dim == 2) b.widen_host();
    if (b.d_nodes) { (void)hipFree(b.d_nodes); b.d_nodes = nullptr; }
    return upload_bvh<T>(b, nullptr);
}

// ---- the 2D families (c_api/bvh.cpp:7-10): the same BvhImpl with dim = 2; inputs are widened to z = 0 on the device, the
// host mirror the caller sees is BvhImpl::nodes2 in the reference's 20/40-byte layout ------------------------------------------
template <typename T> struct CTypes2;
template <> struct CTypes2<float>  { using Bvh = bvh2f; using Node = bvh_node2f; using BBox = bvh_bbox2f; using Vec = bvh_vec2f; using Ray = bvh_ray2f; };
template <> struct CTypes2<double> { using Bvh = bvh2d; using Node = bvh_node2d; using BBox = bvh_bbox2d; using Vec = bvh_vec2d; using Ray = bvh_ray2d; };
template <typename T> BvhImpl<T>* impl2(typename CTypes2<T>::Bvh* b) { return reinterpret_cast<BvhImpl<T>*>(b); }
template <typename T> const BvhImpl<T>* impl2(const typename CTypes2<T>::Bvh* b) { return reinterpret_cast<const BvhImpl<T>*>(b); }
template <typename T> typename CTypes2<T>::Bvh* handle2(BvhImpl<T>* b) { return reinterpret_cast<typename CTypes2<T>::Bvh*>(b); }

template <typename T>
typename CTypes2<T>::Bvh* build2_device(const T* d_bb4, const T* d_cc2, size_t n, const bvh_build_config* config, bvh_amd_builder builder, void* stream) {
    if (!d_bb4 || !d_cc2 || n == 0) { set_error("build: empty input (the reference's behaviour is undefined for 0 primitives)"); return nullptr; }
    bvh_build_config cfg = config ? *config : default_config();
    if (cfg.min_leaf_size < 1 || cfg.min_leaf_size > cfg.max_leaf_size || cfg.max_leaf_size > 15) {
        set_error("build: need 1 <= min_leaf_size <= max_leaf_size <= 15 (4-bit primitive count, index.h:38)");
        return nullptr;
    }
    T *d_bb6 = nullptr, *d_cc3 = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_bb6, n * 6 * sizeof(T)));
    hipError_t e = hipMalloc(&d_cc3, n * 3 * sizeof(T));
    auto b = std::make_unique<BvhImpl<T>>();
    b->dim = 2;
    int rc = e == hipSuccess ? launch_widen_inputs<T>(d_bb4, d_cc2, n, d_bb6, d_cc3, static_cast<hipStream_t>(stream))
                             : fail(BVH_AMD_ERR_HIP, std::string("build: ") + hipGetErrorString(e));
    if (rc == BVH_AMD_OK) rc = build_on_device<T>(*b, d_bb6, d_cc3, n, cfg, builder, static_cast<hipStream_t>(stream));
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    (void)hipFree(d_bb6);
    if (d_cc3) (void)hipFree(d_cc3);
    return rc == BVH_AMD_OK ? handle2<T>(b.release()) : nullptr;
}

template <typename T>
typename CTypes2<T>::Bvh* build2_host(bvh_thread_pool* pool, const typename CTypes2<T>::BBox* bboxes, const typename CTypes2<T>::Vec* centers, size_t n,
                                      const bvh_build_config* config) {
    if (!bboxes || !centers || n == 0) { set_error("build: empty input"); return nullptr; }
    static_assert(sizeof(typename CTypes2<T>::BBox) == 4 * sizeof(T) && sizeof(typename CTypes2<T>::Vec) == 2 * sizeof(T));
    T *d_bb = nullptr, *d_cc = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_bb, n * 4 * sizeof(T)));
    hipError_t e = hipMalloc(&d_cc, n * 2 * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(d_bb, bboxes, n * 4 * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cc, centers, n * 2 * sizeof(T), hipMemcpyHostToDevice);
    typename CTypes2<T>::Bvh* out = nullptr;
    if (e == hipSuccess)
        out = build2_device<T>(d_bb, d_cc, n, config, pool ? BVH_AMD_BUILDER_DEFAULT_PARALLEL : BVH_AMD_BUILDER_DEFAULT_SERIAL, nullptr);
    else
        set_error(std::string("build: ") + hipGetErrorString(e));
    (void)hipDeviceSynchronize();
    (void)hipFree(d_bb);
    if (d_cc) (void)hipFree(d_cc);
    return out;
}

template <typename T>
BvhImpl<T>* adopt_nodes2(const HostNode2<T>* nodes, size_t nn, std::vector<size_t>&& ids) {
    auto b = std::make_unique<BvhImpl<T>>();
    b->dim = 2;
    b->nodes2.assign(nodes, nodes + nn);
    b->widen_host();
    b->prim_ids = std::move(ids);
    if (upload_bvh<T>(*b, nullptr) != BVH_AMD_OK) return nullptr;
    b->nodes2_valid = true;
    return b.release();
}

template <typename T>
typename CTypes2<T>::Bvh* from_nodes2(const void* nodes, size_t nn, const size_t* prim_ids, size_t np) {
    if (!nodes || nn == 0 || (!prim_ids && np)) { set_error("from_nodes: null/empty input"); return nullptr; }
    return handle2<T>(adopt_nodes2<T>(static_cast<const HostNode2<T>*>(nodes), nn, std::vector<size_t>(prim_ids, prim_ids + np)));
}

template <typename T>
size_t stream_size2(const BvhImpl<T>& b) {
    using I = typename IndexOf<T>::Type;
    return 2 * sizeof(I) + b.node_count * sizeof(HostNode2<T>) + b.prim_count * sizeof(I);
}

template <typename T>
size_t serialize2(const BvhImpl<T>& b, void* out, size_t cap) {           // bvh.h:221-229 with Node<T, 2> records
    using I = typename IndexOf<T>::Type;
    const size_t need = stream_size2(b);
    if (!out || cap < need) return need;
    if (b.sync_host2() != BVH_AMD_OK) return 0;
    auto p = static_cast<uint8_t*>(out);
    I hdr[2] = { static_cast<I>(b.nodes2.size()), static_cast<I>(b.prim_ids.size()) };
    std::memcpy(p, hdr, sizeof(hdr)); p += sizeof(hdr);
    std::memcpy(p, b.nodes2.data(), b.nodes2.size() * sizeof(HostNode2<T>)); p += b.nodes2.size() * sizeof(HostNode2<T>);
    for (size_t id : b.prim_ids) { I v = static_cast<I>(id); std::memcpy(p, &v, sizeof(v)); p += sizeof(v); }
    return need;
}

template <typename T>
typename CTypes2<T>::Bvh* deserialize2(const void* bytes, size_t size) {
    using I = typename IndexOf<T>::Type;
    if (!bytes || size < 2 * sizeof(I)) { set_error("deserialize: truncated stream"); return nullptr; }
    auto p = static_cast<const uint8_t*>(bytes);
    I hdr[2];
    std::memcpy(hdr, p, sizeof(hdr)); p += sizeof(hdr);
    const size_t nn = hdr[0], np = hdr[1];
    if (size < 2 * sizeof(I) + nn * sizeof(HostNode2<T>) + np * sizeof(I)) { set_error("deserialize: truncated stream"); return nullptr; }
    std::vector<HostNode2<T>> nodes(nn);
    std::memcpy(nodes.data(), p, nn * sizeof(HostNode2<T>)); p += nn * sizeof(HostNode2<T>);
    std::vector<size_t> ids(np);
    for (size_t i = 0; i < np; ++i) { I v; std::memcpy(&v, p, sizeof(v)); p += sizeof(v); ids[i] = static_cast<size_t>(v); }
    return handle2<T>(adopt_nodes2<T>(nodes.data(), nn, std::move(ids)));
}

template <typename T>
typename CTypes2<T>::Bvh* load2(FILE* f) {
    using I = typename IndexOf<T>::Type;
    I hdr[2] = {0, 0};
    if (fread(hdr, sizeof(I), 2, f) != 2) { set_error("load: truncated stream"); return nullptr; }
    std::vector<uint8_t> buf;
    try { buf.resize(2 * sizeof(I) + size_t(hdr[0]) * sizeof(HostNode2<T>) + size_t(hdr[1]) * sizeof(I)); }
    catch (const std::exception&) { set_error("load: the header asks for more memory than there is"); return nullptr; }
    std::memcpy(buf.data(), hdr, sizeof(hdr));
    const size_t rest = buf.size() - sizeof(hdr);
    if (fread(buf.data() + sizeof(hdr), 1, rest, f) != rest) { set_error("load: truncated stream"); return nullptr; }
    return deserialize2<T>(buf.data(), buf.size());
}

template <typename T>
int intersect2(const typename CTypes2<T>::Bvh* bvh, const T* d_circles3, const typename CTypes2<T>::Ray* d_rays, size_t n, unsigned flags,
               typename HitOf<T>::Type* d_hits, bvh_amd_counters* d_counters, void* stream) {
    if (!bvh) return fail(BVH_AMD_ERR_ARG, "intersect_rays: null bvh");
    static_assert(sizeof(typename CTypes2<T>::Ray) == 6 * sizeof(T));
    const BvhImpl<T>& b = *impl2<T>(bvh);
    int cur = -1;
    BVH_HIP_TRY(hipGetDevice(&cur), BVH_AMD_ERR_HIP);
    if (cur != b.device) return fail(BVH_AMD_ERR_ARG, "intersect_rays: BVH lives on another device than the current one");
    return launch_traverse<T>(b, LEAF_SPHERE, d_circles3, reinterpret_cast<const T*>(d_rays), n, flags, d_hits, d_counters,
                              static_cast<hipStream_t>(stream));
}

template <typename T>
int intersect(const typename CTypes<T>::Bvh* bvh, int leaf, const T* d_prims, const typename CTypes<T>::Ray* d_rays, size_t n,
              unsigned flags, typename HitOf<T>::Type* d_hits, bvh_amd_counters* d_counters, void* stream)
{
    if (!bvh) return fail(BVH_AMD_ERR_ARG, "intersect_rays: null bvh");
    static_assert(sizeof(typename CTypes<T>::Ray) == 8 * sizeof(T));
    const BvhImpl<T>& b = *impl<T>(bvh);
    int cur = -1;
    BVH_HIP_TRY(hipGetDevice(&cur), BVH_AMD_ERR_HIP);
    if (cur != b.device) return fail(BVH_AMD_ERR_ARG, "intersect_rays: BVH lives on another device than the current one");
    return launch_traverse<T>(b, leaf, d_prims, reinterpret_cast<const T*>(d_rays), n, flags, d_hits, d_counters,
                              static_cast<hipStream_t>(stream));
}

// bvhXX_intersect_ray{,_any}{,_robust} (c_api/bvh.h:277-295 over bvh_impl.h:235-250): one ray, the leaves go to the caller's
// function. The walk runs on the device (traverse.hip, ray_step_kernel); `ray` is the family's own struct.
template <typename T, int D>
int intersect_ray_visit(const BvhImpl<T>* b, const void* ray, size_t start, unsigned flags, bool (*leaf_fn)(void*, T*, size_t, size_t),
                        void (*inner_fn)(void*, size_t), void* user) {
    if (!b || !ray) return fail(BVH_AMD_ERR_ARG, "intersect_ray: null bvh or ray");
    if (b->dim != D) return fail(BVH_AMD_ERR_ARG, "intersect_ray: dimension mismatch");
    int cur = -1;
    BVH_HIP_TRY(hipGetDevice(&cur), BVH_AMD_ERR_HIP);
    if (cur != b->device) return fail(BVH_AMD_ERR_ARG, "intersect_ray: BVH lives on another device than the current one");
    if (start == BVH_AMD_START_AT_ROOT) start = b->root_index;
    const T* r = static_cast<const T*>(ray);
    T ray8[8];
    if (D == 3) { for (int k = 0; k < 8; ++k) ray8[k] = r[k]; }
    else { ray8[0] = r[0]; ray8[1] = r[1]; ray8[2] = T(0); ray8[3] = r[2]; ray8[4] = r[3]; ray8[5] = T(0); ray8[6] = r[4]; ray8[7] = r[5]; }
    return trace_ray_callbacks<T>(*b, ray8, static_cast<uint32_t>(start), (flags & BVH_AMD_RAY_ANY_HIT) != 0, (flags & BVH_AMD_RAY_ROBUST) != 0,
                                  leaf_fn, inner_fn, user);
}

// The reference's bvhXX_build never returns NULL and its callers do not check (test/c_api_example.c:114-120): say why before they trip.
template <typename P> P* loud(P* result, const char* what) {
    if (!result) std::fprintf(stderr, "bvh_amd: %s failed: %s\n", what, g_error.c_str());
    return result;
}

// bvhXX_optimize / bvhXX_refit return void in the reference and cannot fail there; here a HIP error or a search-stack overflow would
// leave the tree un-optimised, un-refitted or half updated with nobody the wiser: say why and stop (callers that want to handle
// the failure use the int-returning bvhXX_optimize_config / bvhXX_refit_status).
inline void loud_or_abort(int rc, const char* what) {
    if (rc != BVH_AMD_OK) {
        std::fprintf(stderr, "bvh_amd: %s failed: %s\n", what, g_error.c_str());
        std::abort();
    }
}

// The reference's functions return void and cannot fail; a failure here would otherwise read as "no intersection".
template <typename T, int D, typename Callback>
void intersect_ray_legacy(const BvhImpl<T>* b, const void* ray, const Callback* callback, unsigned flags) {
    int rc = callback && callback->user_fn && b
                 ? intersect_ray_visit<T, D>(b, ray, b->root_index, flags, callback->user_fn, nullptr, callback->user_data)
                 : fail(BVH_AMD_ERR_ARG, "intersect_ray: null bvh or callback");
    if (rc != BVH_AMD_OK) {
        std::fprintf(stderr, "bvh_amd: bvh_intersect_ray failed: %s\n", g_error.c_str());
        std::abort();
    }
}

} // namespace

template <typename T> int nodes_resident(BvhImpl<T>& b) { return make_nodes_resident<T>(b); }
template int nodes_resident<float>(BvhImpl<float>&);
template int nodes_resident<double>(BvhImpl<double>&);

} // namespace bvh_amd

using namespace bvh_amd;

extern "C" {

const char* bvh_amd_last_error(void) { return g_error.c_str(); }
const char* bvh_amd_version(void) { return "bvh_amd 0.1 (gfx950)"; }
const char* bvh_amd_last_kernel_name(void) { return last_kernel_name(); }
int bvh_amd_last_launch_reordered(void) { return last_launch_reordered() ? 1 : 0; }
void bvh_amd_kernel_timing(int on) { kernel_timing(on != 0); }
int bvh_amd_kernel_times(float* ms_out, size_t capacity, size_t* count_out) {
    if (!ms_out && capacity) return fail(BVH_AMD_ERR_ARG, "bvh_amd_kernel_times: null output");
    return kernel_times(ms_out, capacity, count_out);
}
void bvh_amd_last_optimize_profile(struct bvh_amd_optimize_profile* out) { if (out) last_optimize_profile(out); }
int bvh_amd_experiment(const char* name, int value) { return set_experiment(name, value); }
int bvh_amd_wave_times(unsigned long long* out, size_t capacity_waves, size_t* n_waves) { return wave_times(out, capacity_waves, n_waves); }
void bvh_amd_tuning(int refill_threshold, int leaf_threshold, int coop_fetch, int ticket_ranges) { set_tuning(refill_threshold, leaf_threshold, coop_fetch, ticket_ranges); }
void bvh_amd_last_launch_plan(int out[4]) { if (out) last_launch_plan(out); }
void bvh_amd_last_plan_search(float ns_per_ray[5], int measurements[5], unsigned* dropped_mask) { last_plan_search(ns_per_ray, measurements, dropped_mask); }
int bvh_amd_reorder_times(float* ms_out, size_t capacity, size_t* count_out) {
    if (!ms_out && capacity) return fail(BVH_AMD_ERR_ARG, "bvh_amd_reorder_times: null output");
    return reorder_times(ms_out, capacity, count_out);
}
void bvh_amd_reinsertion_stats(unsigned out[2]) { if (out) reinsertion_stats(out); }

// Scratch blocks of finished builds stay cached in the current device's stream-ordered pool (common.h: scratch_alloc); this hands
// them back to the driver (waits for the device first).
int bvh_amd_release_cached_memory(void) {
    int dev = 0;
    BVH_HIP_TRY(hipGetDevice(&dev), BVH_AMD_ERR_HIP);
    (void)bvh_amd_comm_cache_clear();                         // the communicators bvhXX_replicate keeps (replicate.hip)
    scratch_cache_flush();
    BVH_HIP_TRY(hipDeviceSynchronize(), BVH_AMD_ERR_HIP);
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess) { (void)hipGetLastError(); return BVH_AMD_OK; }
    BVH_HIP_TRY(hipMemPoolTrimTo(pool, 0), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipDeviceSynchronize(), BVH_AMD_ERR_HIP);     // (nothing of the library's is queued between the unmapping and its next allocation)
    return BVH_AMD_OK;
}

size_t bvh_amd_cached_scratch_bytes(void) { return scratch_cache_bytes(); }
size_t bvh_amd_scratch_cache_limit(void) { return scratch_cache_limit(); }

int bvh_amd_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    return n;
}

int bvh_amd_device_name(int device, char* out, size_t cap) {
    hipDeviceProp_t p;
    BVH_HIP_TRY(hipGetDeviceProperties(&p, device), BVH_AMD_ERR_HIP);
    snprintf(out, cap, "%s (%s)", p.name, p.gcnArchName);
    return BVH_AMD_OK;
}

void* bvh_amd_device_alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { set_error(std::string("device_alloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
void bvh_amd_device_free(void* p) { if (p) (void)hipFree(p); }
int bvh_amd_copy_to_device(void* d, const void* h, size_t bytes) { BVH_HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice), BVH_AMD_ERR_HIP); return BVH_AMD_OK; }
int bvh_amd_copy_to_host(void* h, const void* d, size_t bytes) { BVH_HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost), BVH_AMD_ERR_HIP); return BVH_AMD_OK; }
int bvh_amd_synchronize(void* stream) { BVH_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)), BVH_AMD_ERR_HIP); return BVH_AMD_OK; }

bvh_thread_pool* bvh_thread_pool_create(size_t thread_count) {
    return reinterpret_cast<bvh_thread_pool*>(new ThreadPoolTag{thread_count});
}
void bvh_thread_pool_destroy(bvh_thread_pool* p) { delete reinterpret_cast<ThreadPoolTag*>(p); }

#define BVH_AMD_IMPL(T, S)                                                                                          \
    bvh##S* bvh##S##_build(bvh_thread_pool* pool, const bvh_bbox##S* bb, const bvh_vec##S* cc, size_t n,            \
                           const bvh_build_config* cfg) { return loud(build_host<T>(pool, bb, cc, n, cfg), "bvh" #S "_build"); } \
    bvh##S* bvh##S##_build_device(const T* d_bb, const T* d_cc, size_t n, const bvh_build_config* cfg,              \
                                  enum bvh_amd_builder builder, void* stream) {                                     \
        return build_device<T>(d_bb, d_cc, n, cfg, builder, stream); }                                              \
    bvh##S* bvh##S##_build_sah(bvh_thread_pool* pool, const bvh_bbox##S* bb, const bvh_vec##S* cc, size_t n,        \
                               const bvh_build_config* cfg, const bvh_amd_sah_config* sah) {                       \
        return with_sah(sah, [&] { return build_host<T>(pool, bb, cc, n, cfg); }); }                                \
    bvh##S* bvh##S##_build_device_sah(const T* d_bb, const T* d_cc, size_t n, const bvh_build_config* cfg,          \
                                      enum bvh_amd_builder builder, const bvh_amd_sah_config* sah, void* stream) {  \
        return with_sah(sah, [&] { return build_device<T>(d_bb, d_cc, n, cfg, builder, stream); }); }               \
    bvh##S* bvh##S##_build_device_binned(const T* d_bb, const T* d_cc, size_t n, const bvh_build_config* cfg,       \
                                         const bvh_amd_sah_config* sah, size_t bin_count, void* stream) {           \
        return with_bins(sah, bin_count, [&] { return build_device<T>(d_bb, d_cc, n, cfg, BVH_AMD_BUILDER_BINNED, stream); }); } \
    bvh##S* bvh##S##_build_minitree_device(const T* d_bb, const T* d_cc, size_t n, const bvh_amd_minitree_config* cfg, void* stream) { \
        return build_minitree<T>(d_bb, d_cc, n, cfg, stream); }                                                     \
    bvh##S* bvh##S##_extract(bvh##S* b, size_t root_id) { return handle<T>(extract<T>(impl<T>(b), root_id)); }      \
    bvh##S* bvh##S##_from_nodes(const void* nodes, size_t nn, const size_t* ids, size_t np) {                       \
        return from_nodes<T>(nodes, nn, ids, np); }                                                                 \
    void bvh##S##_destroy(bvh##S* b) { delete impl<T>(b); }                                                         \
    void bvh##S##_optimize(bvh_thread_pool*, bvh##S* b) { loud_or_abort(optimize<T>(impl<T>(b)), "bvh" #S "_optimize"); } \
    int bvh##S##_optimize_config(bvh##S* b, const bvh_amd_optimize_config* c) { return optimize_config<T>(impl<T>(b), c); } \
    void bvh##S##_refit(bvh##S* b) { loud_or_abort(refit<T>(impl<T>(b)), "bvh" #S "_refit"); }                     \
    int bvh##S##_refit_status(bvh##S* b) { return refit<T>(impl<T>(b)); }                                           \
    int bvh##S##_sync_device(bvh##S* b) { return sync_device<T>(impl<T>(b)); }                                       \
    void bvh##S##_append_node(bvh##S* b) {                                                                          \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK) return;                                                           \
        impl<T>(b)->nodes.emplace_back(); impl<T>(b)->node_count = impl<T>(b)->nodes.size(); }                       \
    void bvh##S##_remove_last_node(bvh##S* b) {                                                                     \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK || impl<T>(b)->nodes.empty()) return;                              \
        impl<T>(b)->nodes.pop_back(); impl<T>(b)->node_count = impl<T>(b)->nodes.size(); }                           \
    void bvh_node##S##_set_prim_count(bvh_node##S* n, size_t c) {                                                   \
        auto h = reinterpret_cast<HostNode<T>*>(n);                                                                 \
        h->index = (h->index & ~static_cast<IndexOf<T>::Type>(kCountMask)) | (static_cast<IndexOf<T>::Type>(c) & kCountMask); } \
    void bvh_node##S##_set_first_id(bvh_node##S* n, size_t f) {                                                     \
        auto h = reinterpret_cast<HostNode<T>*>(n);                                                                 \
        h->index = (static_cast<IndexOf<T>::Type>(f) << kCountBits) | (h->index & kCountMask); }                    \
    void bvh_node##S##_set_bbox(bvh_node##S* n, const bvh_bbox##S* bb) {                                            \
        auto h = reinterpret_cast<HostNode<T>*>(n);                                                                 \
        h->bounds[0] = bb->min.x; h->bounds[1] = bb->max.x; h->bounds[2] = bb->min.y; h->bounds[3] = bb->max.y;     \
        h->bounds[4] = bb->min.z; h->bounds[5] = bb->max.z; }                                                         \
    void bvh##S##_save(const bvh##S* b, FILE* f) { if (b && f) save<T>(*impl<T>(b), f); }                           \
    bvh##S* bvh##S##_load(FILE* f) { return f ? load<T>(f) : nullptr; }                                             \
    size_t bvh##S##_serialize(const bvh##S* b, void* out, size_t cap) { return b ? serialize<T>(*impl<T>(b), out, cap) : 0; } \
    bvh##S* bvh##S##_deserialize(const void* bytes, size_t size) { return deserialize<T>(bytes, size); }            \
    size_t bvh##S##_serialize_device(bvh##S* b, void* d_out, size_t cap, void* stream) { return serialize_device<T>(impl<T>(b), d_out, cap, stream); } \
    bvh##S* bvh##S##_deserialize_device(const void* d_bytes, size_t size, void* stream) {                           \
        return handle<T>(deserialize_from_device<T>(d_bytes, size, 3, static_cast<hipStream_t>(stream))); }         \
    bvh_node##S* bvh##S##_get_node(bvh##S* b, size_t i) {                                                          \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK) return nullptr;                                                   \
        return reinterpret_cast<bvh_node##S*>(&impl<T>(b)->nodes[i]); }                                              \
    size_t bvh##S##_get_prim_id(const bvh##S* b, size_t i) {                                                        \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK) return BVH_INVALID_PRIM_ID;                                       \
        return impl<T>(b)->prim_ids[i]; }                                                                            \
    size_t bvh##S##_get_prim_count(const bvh##S* b) { return impl<T>(b)->prim_count; }                              \
    size_t bvh##S##_get_node_count(const bvh##S* b) { return impl<T>(b)->node_count; }                              \
    bool bvh_node##S##_is_leaf(const bvh_node##S* n) { return (reinterpret_cast<const HostNode<T>*>(n)->index & kCountMask) != 0; } \
    size_t bvh_node##S##_get_prim_count(const bvh_node##S* n) { return reinterpret_cast<const HostNode<T>*>(n)->index & kCountMask; } \
    size_t bvh_node##S##_get_first_id(const bvh_node##S* n) { return reinterpret_cast<const HostNode<T>*>(n)->index >> kCountBits; } \
    bvh_bbox##S bvh_node##S##_get_bbox(const bvh_node##S* n) {                                                      \
        auto h = reinterpret_cast<const HostNode<T>*>(n);                                                           \
        bvh_bbox##S r; r.min.x = h->bounds[0]; r.max.x = h->bounds[1]; r.min.y = h->bounds[2]; r.max.y = h->bounds[3]; \
        r.min.z = h->bounds[4]; r.max.z = h->bounds[5]; return r; }                                                 \
    void bvh##S##_copy_nodes(const bvh##S* b, void* out) {                                                          \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK) return;                                                           \
        std::memcpy(out, impl<T>(b)->nodes.data(), impl<T>(b)->nodes.size() * sizeof(HostNode<T>)); }               \
    void bvh##S##_copy_prim_ids(const bvh##S* b, size_t* out) {                                                     \
        if (impl<T>(b)->sync_host() != BVH_AMD_OK) return;                                                           \
        std::memcpy(out, impl<T>(b)->prim_ids.data(), impl<T>(b)->prim_ids.size() * sizeof(size_t)); }              \
    const uint32_t* bvh##S##_device_prim_ids(const bvh##S* b) { return impl<T>(b)->d_prim_ids; }                    \
    int bvh_amd_tri_bounds##S(const T* t, size_t n, T* bb, T* cc, void* s) {                                        \
        return launch_tri_bounds<T>(t, n, bb, cc, static_cast<hipStream_t>(s)); }                                   \
    int bvh_amd_precompute_tris##S(const T* t, const uint32_t* perm, size_t n, T* out, void* s) {                   \
        return launch_precompute_tris<T>(t, perm, n, out, static_cast<hipStream_t>(s)); }                           \
    int bvh_amd_sphere_bounds##S(const T* sp, size_t n, T* bb, T* cc, void* s) {                                    \
        return launch_sphere_bounds<T>(sp, n, bb, cc, static_cast<hipStream_t>(s)); }                               \
    int bvh##S##_intersect_rays_tri(const bvh##S* b, const T* prims, const bvh_ray##S* rays, size_t n, unsigned flags, \
                                    bvh_hit##S* hits, bvh_amd_counters* cnt, void* s) {                             \
        return intersect<T>(b, LEAF_TRIANGLE, prims, rays, n, flags, hits, cnt, s); }                               \
    int bvh##S##_intersect_rays_sphere(const bvh##S* b, const T* prims, const bvh_ray##S* rays, size_t n, unsigned flags, \
                                       bvh_hit##S* hits, bvh_amd_counters* cnt, void* s) {                          \
        return intersect<T>(b, LEAF_SPHERE, prims, rays, n, flags, hits, cnt, s); }                                 \
    int bvh##S##_prepare_trace(const bvh##S* b, size_t n_rays_hint, void* s) {                                      \
        if (!b) return fail(BVH_AMD_ERR_ARG, "prepare_trace: null bvh");                                            \
        int cur = -1;                                                                                               \
        BVH_HIP_TRY(hipGetDevice(&cur), BVH_AMD_ERR_HIP);                                                           \
        if (cur != impl<T>(b)->device) return fail(BVH_AMD_ERR_ARG, "prepare_trace: BVH lives on another device than the current one"); \
        return prepare_trace<T>(*impl<T>(b), n_rays_hint, static_cast<hipStream_t>(s)); }

BVH_AMD_IMPL(float, 3f)
BVH_AMD_IMPL(double, 3d)

#define BVH_AMD_IMPL_RAY(T, S, CB, VIS, D, IMPL)                                                                         \
    void bvh##S##_intersect_ray(const bvh##S* b, const bvh_ray##S* r, const CB* cb) { intersect_ray_legacy<T, D>(IMPL<T>(b), r, cb, 0u); } \
    void bvh##S##_intersect_ray_any(const bvh##S* b, const bvh_ray##S* r, const CB* cb) {                           \
        intersect_ray_legacy<T, D>(IMPL<T>(b), r, cb, BVH_AMD_RAY_ANY_HIT); }                                        \
    void bvh##S##_intersect_ray_robust(const bvh##S* b, const bvh_ray##S* r, const CB* cb) {                        \
        intersect_ray_legacy<T, D>(IMPL<T>(b), r, cb, BVH_AMD_RAY_ROBUST); }                                         \
    void bvh##S##_intersect_ray_any_robust(const bvh##S* b, const bvh_ray##S* r, const CB* cb) {                    \
        intersect_ray_legacy<T, D>(IMPL<T>(b), r, cb, BVH_AMD_RAY_ANY_HIT | BVH_AMD_RAY_ROBUST); }                   \
    int bvh##S##_intersect_ray_visit(const bvh##S* b, const bvh_ray##S* r, size_t start, unsigned flags, const VIS* v) { \
        if (!v || !v->leaf_fn) return fail(BVH_AMD_ERR_ARG, "intersect_ray_visit: null visitor");                   \
        return intersect_ray_visit<T, D>(IMPL<T>(b), r, start, flags, v->leaf_fn, v->inner_fn, v->user_data); }

BVH_AMD_IMPL_RAY(float, 3f, bvh_intersect_callbackf, bvh_amd_ray_visitorf, 3, impl)
BVH_AMD_IMPL_RAY(double, 3d, bvh_intersect_callbackd, bvh_amd_ray_visitord, 3, impl)

#define BVH_AMD_IMPL2(T, S)                                                                                         \
    bvh##S* bvh##S##_build(bvh_thread_pool* pool, const bvh_bbox##S* bb, const bvh_vec##S* cc, size_t n,            \
                           const bvh_build_config* cfg) { return loud(build2_host<T>(pool, bb, cc, n, cfg), "bvh" #S "_build"); } \
    bvh##S* bvh##S##_build_device(const T* d_bb4, const T* d_cc2, size_t n, const bvh_build_config* cfg,            \
                                  enum bvh_amd_builder builder, void* stream) {                                     \
        return build2_device<T>(d_bb4, d_cc2, n, cfg, builder, stream); }                                           \
    bvh##S* bvh##S##_build_sah(bvh_thread_pool* pool, const bvh_bbox##S* bb, const bvh_vec##S* cc, size_t n,        \
                               const bvh_build_config* cfg, const bvh_amd_sah_config* sah) {                       \
        return with_sah(sah, [&] { return build2_host<T>(pool, bb, cc, n, cfg); }); }                               \
    bvh##S* bvh##S##_build_device_sah(const T* d_bb4, const T* d_cc2, size_t n, const bvh_build_config* cfg,        \
                                      enum bvh_amd_builder builder, const bvh_amd_sah_config* sah, void* stream) {  \
        return with_sah(sah, [&] { return build2_device<T>(d_bb4, d_cc2, n, cfg, builder, stream); }); }            \
    bvh##S* bvh##S##_build_device_binned(const T* d_bb4, const T* d_cc2, size_t n, const bvh_build_config* cfg,     \
                                         const bvh_amd_sah_config* sah, size_t bin_count, void* stream) {           \
        return with_bins(sah, bin_count, [&] { return build2_device<T>(d_bb4, d_cc2, n, cfg, BVH_AMD_BUILDER_BINNED, stream); }); } \
    bvh##S* bvh##S##_extract(bvh##S* b, size_t root_id) { return handle2<T>(extract<T>(impl2<T>(b), root_id)); }    \
    bvh##S* bvh##S##_from_nodes(const void* nodes, size_t nn, const size_t* ids, size_t np) {                       \
        return from_nodes2<T>(nodes, nn, ids, np); }                                                                \
    void bvh##S##_destroy(bvh##S* b) { delete impl2<T>(b); }                                                        \
    void bvh##S##_optimize(bvh_thread_pool*, bvh##S* b) { loud_or_abort(optimize<T>(impl2<T>(b)), "bvh" #S "_optimize"); } \
    int bvh##S##_optimize_config(bvh##S* b, const bvh_amd_optimize_config* c) { return optimize_config<T>(impl2<T>(b), c); } \
    void bvh##S##_refit(bvh##S* b) { loud_or_abort(refit<T>(impl2<T>(b)), "bvh" #S "_refit"); }                    \
    int bvh##S##_refit_status(bvh##S* b) { return refit<T>(impl2<T>(b)); }                                          \
    int bvh##S##_sync_device(bvh##S* b) { return sync_device<T>(impl2<T>(b)); }                                      \
    void bvh##S##_append_node(bvh##S* b) {                                                                          \
        if (impl2<T>(b)->sync_host2() != BVH_AMD_OK) return;                                                         \
        impl2<T>(b)->nodes2.emplace_back(); impl2<T>(b)->node_count = impl2<T>(b)->nodes2.size(); }                  \
    void bvh##S##_remove_last_node(bvh##S* b) {                                                                     \
        if (impl2<T>(b)->sync_host2() != BVH_AMD_OK || impl2<T>(b)->nodes2.empty()) return;                          \
        impl2<T>(b)->nodes2.pop_back(); impl2<T>(b)->node_count = impl2<T>(b)->nodes2.size(); }                      \
    void bvh_node##S##_set_prim_count(bvh_node##S* n, size_t c) {                                                   \
        auto h = reinterpret_cast<HostNode2<T>*>(n);                                                                \
        h->index = (h->index & ~static_cast<IndexOf<T>::Type>(kCountMask)) | (static_cast<IndexOf<T>::Type>(c) & kCountMask); } \
    void bvh_node##S##_set_first_id(bvh_node##S* n, size_t f) {                                                     \
        auto h = reinterpret_cast<HostNode2<T>*>(n);                                                                \
        h->index = (static_cast<IndexOf<T>::Type>(f) << kCountBits) | (h->index & kCountMask); }                    \
    void bvh_node##S##_set_bbox(bvh_node##S* n, const bvh_bbox##S* bb) {                                            \
        auto h = reinterpret_cast<HostNode2<T>*>(n);                                                                \
        h->bounds[0] = bb->min.x; h->bounds[1] = bb->max.x; h->bounds[2] = bb->min.y; h->bounds[3] = bb->max.y; }   \
    void bvh##S##_save(const bvh##S* b, FILE* f) {                                                                  \
        if (!b || !f) return;                                                                                        \
        std::vector<uint8_t> buf(stream_size2(*impl2<T>(b)));                                                        \
        serialize2(*impl2<T>(b), buf.data(), buf.size());                                                            \
        fwrite(buf.data(), 1, buf.size(), f); }                                                                      \
    bvh##S* bvh##S##_load(FILE* f) { return f ? load2<T>(f) : nullptr; }                                            \
    size_t bvh##S##_serialize(const bvh##S* b, void* out, size_t cap) { return b ? serialize2<T>(*impl2<T>(b), out, cap) : 0; } \
    bvh##S* bvh##S##_deserialize(const void* bytes, size_t size) { return deserialize2<T>(bytes, size); }           \
    size_t bvh##S##_serialize_device(bvh##S* b, void* d_out, size_t cap, void* stream) { return serialize_device<T>(impl2<T>(b), d_out, cap, stream); } \
    bvh##S* bvh##S##_deserialize_device(const void* d_bytes, size_t size, void* stream) {                           \
        return handle2<T>(deserialize_from_device<T>(d_bytes, size, 2, static_cast<hipStream_t>(stream))); }        \
    bvh_node##S* bvh##S##_get_node(bvh##S* b, size_t i) {                                                          \
        if (impl2<T>(b)->sync_host2() != BVH_AMD_OK) return nullptr;                                                 \
        return reinterpret_cast<bvh_node##S*>(&impl2<T>(b)->nodes2[i]); }                                            \
    size_t bvh##S##_get_prim_id(const bvh##S* b, size_t i) {                                                        \
        if (impl2<T>(b)->sync_host() != BVH_AMD_OK) return BVH_INVALID_PRIM_ID;                                      \
        return impl2<T>(b)->prim_ids[i]; }                                                                           \
    size_t bvh##S##_get_prim_count(const bvh##S* b) { return impl2<T>(b)->prim_count; }                             \
    size_t bvh##S##_get_node_count(const bvh##S* b) { return impl2<T>(b)->node_count; }                             \
    bool bvh_node##S##_is_leaf(const bvh_node##S* n) { return (reinterpret_cast<const HostNode2<T>*>(n)->index & kCountMask) != 0; } \
    size_t bvh_node##S##_get_prim_count(const bvh_node##S* n) { return reinterpret_cast<const HostNode2<T>*>(n)->index & kCountMask; } \
    size_t bvh_node##S##_get_first_id(const bvh_node##S* n) { return reinterpret_cast<const HostNode2<T>*>(n)->index >> kCountBits; } \
    bvh_bbox##S bvh_node##S##_get_bbox(const bvh_node##S* n) {                                                      \
        auto h = reinterpret_cast<const HostNode2<T>*>(n);                                                          \
        bvh_bbox##S r; r.min.x = h->bounds[0]; r.max.x = h->bounds[1]; r.min.y = h->bounds[2]; r.max.y = h->bounds[3]; return r; } \
    void bvh##S##_copy_nodes(const bvh##S* b, void* out) {                                                          \
        if (impl2<T>(b)->sync_host2() != BVH_AMD_OK) return;                                                         \
        std::memcpy(out, impl2<T>(b)->nodes2.data(), impl2<T>(b)->nodes2.size() * sizeof(HostNode2<T>)); }          \
    void bvh##S##_copy_prim_ids(const bvh##S* b, size_t* out) {                                                     \
        if (impl2<T>(b)->sync_host() != BVH_AMD_OK) return;                                                          \
        std::memcpy(out, impl2<T>(b)->prim_ids.data(), impl2<T>(b)->prim_ids.size() * sizeof(size_t)); }            \
    const uint32_t* bvh##S##_device_prim_ids(const bvh##S* b) { return impl2<T>(b)->d_prim_ids; }                   \
    int bvh_amd_sphere_bounds##S(const T* c3, size_t n, T* bb4, T* cc2, void* s) {                                  \
        return launch_circle_bounds<T>(c3, n, bb4, cc2, static_cast<hipStream_t>(s)); }                             \
    int bvh##S##_intersect_rays_sphere(const bvh##S* b, const T* circles, const bvh_ray##S* rays, size_t n, unsigned flags, \
                                       HitOf<T>::Type* hits, bvh_amd_counters* cnt, void* s) {                      \
        return intersect2<T>(b, circles, rays, n, flags, hits, cnt, s); }

BVH_AMD_IMPL2(float, 2f)
BVH_AMD_IMPL2(double, 2d)
BVH_AMD_IMPL_RAY(float, 2f, bvh_intersect_callbackf, bvh_amd_ray_visitorf, 2, impl2)
BVH_AMD_IMPL_RAY(double, 2d, bvh_intersect_callbackd, bvh_amd_ray_visitord, 2, impl2)

int bvh_amd_std_sort_ids3f(const float* d_keys, size_t n, uint32_t* d_ids_out, void* stream) {
    return std_sort_ids<float>(d_ids_out, d_keys, static_cast<uint32_t>(n), 1, 0, 1, static_cast<hipStream_t>(stream));
}
int bvh_amd_std_sort_ids3d(const double* d_keys, size_t n, uint32_t* d_ids_out, void* stream) {
    return std_sort_ids<double>(d_ids_out, d_keys, static_cast<uint32_t>(n), 1, 0, 1, static_cast<hipStream_t>(stream));
}
int bvh_amd_radix_sort_pairs_u32(uint32_t* d_keys, uint32_t* d_vals, size_t n, int bits, void* stream) {
    uint32_t *kt = nullptr, *vt = nullptr;
    BVH_HIP_TRY(hipMalloc(&kt, std::max<size_t>(n, 1) * 4), BVH_AMD_ERR_HIP);
    hipError_t e = hipMalloc(&vt, std::max<size_t>(n, 1) * 4);
    int rc = e == hipSuccess ? radix_sort_pairs<uint32_t>(d_keys, d_vals, kt, vt, static_cast<uint32_t>(n), 1, bits, static_cast<hipStream_t>(stream))
                             : fail(BVH_AMD_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(kt);
    if (vt) (void)hipFree(vt);
    return rc;
}

int bvh_amd_pinhole_rays3f(const float eye[3], const float dir[3], const float up[3], size_t w, size_t h, bvh_ray3f* d_rays, void* stream) {
    return launch_pinhole_rays<float>(eye, dir, up, w, h, reinterpret_cast<float*>(d_rays), static_cast<hipStream_t>(stream));
}
int bvh_amd_pinhole_rays3d(const double eye[3], const double dir[3], const double up[3], size_t w, size_t h, bvh_ray3d* d_rays, void* stream) {
    return launch_pinhole_rays<double>(eye, dir, up, w, h, reinterpret_cast<double*>(d_rays), static_cast<hipStream_t>(stream));
}
int bvh_amd_shade_eyelight3f(const float* d_tris12, const bvh_ray3f* d_rays, const bvh_hit3f* d_hits, size_t n, uint8_t* d_rgb, void* stream) {
    return launch_shade_eyelight<float>(d_tris12, reinterpret_cast<const float*>(d_rays), d_hits, n, d_rgb, static_cast<hipStream_t>(stream));
}
int bvh_amd_shade_eyelight3d(const double* d_tris12, const bvh_ray3d* d_rays, const bvh_hit3d* d_hits, size_t n, uint8_t* d_rgb, void* stream) {
    return launch_shade_eyelight<double>(d_tris12, reinterpret_cast<const double*>(d_rays), d_hits, n, d_rgb, static_cast<hipStream_t>(stream));
}

int bvh_amd_gather(const void* d_in, const uint32_t* d_perm, size_t n, size_t stride, void* d_out, void* stream) {
    return launch_gather(d_in, d_perm, n, stride, d_out, static_cast<hipStream_t>(stream));
}

} // extern "C"
