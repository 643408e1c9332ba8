// Internal definitions shared by the host layer and the HIP kernels of libbvh_amd.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/bvh_amd.h"

namespace bvh_amd {

constexpr unsigned kCountBits = 4;            // reference node.h:22 (PrimCountBits)
constexpr uint32_t kCountMask = 15u;
constexpr int kWave = 64;

// ---- host-visible node in the reference's layout (node.h:31-37, index.h:74-81) ----------------
template <typename T> struct IndexOf;
template <> struct IndexOf<float>  { using Type = uint32_t; };
template <> struct IndexOf<double> { using Type = uint64_t; };

template <typename T>
struct HostNode {
    T bounds[6];                               // {minx,maxx,miny,maxy,minz,maxz}
    typename IndexOf<T>::Type index;           // first_id << 4 | prim_count
};
static_assert(sizeof(HostNode<float>) == 28 && sizeof(HostNode<double>) == 56);

template <typename T>
struct HostNode2 {                             // Node<T, 2> (node.h:31-37): what bvh2f_get_node / _save / _serialize expose
    T bounds[4];                               // {minx,maxx,miny,maxy}
    typename IndexOf<T>::Type index;
};
static_assert(sizeof(HostNode2<float>) == 20 && sizeof(HostNode2<double>) == 40);

// ---- device traversal record: both children of one inner node in one aligned line ------------
// The reference fetches nodes[first_id] and nodes[first_id + 1] (bvh.h:133-134), 2 x 28 B that start at
// byte 28 * odd and straddle cache lines. On the device the sibling pair p = (first_id - 1) / 2 is one
// 64-byte (float) / 128-byte (double) record; child indices are narrowed to 32 bits (checked on upload).
template <typename T> struct PairNode;
template <> struct alignas(64) PairNode<float> {
    float lb[6], rb[6];
    uint32_t li, ri;
    uint32_t pad[2];
};
template <> struct alignas(128) PairNode<double> {
    double lb[6], rb[6];
    uint32_t li, ri;
    uint32_t pad[6];
};
static_assert(sizeof(PairNode<float>) == 64 && sizeof(PairNode<double>) == 128);


template <typename T> struct HitOf;
template <> struct HitOf<float>  { using Type = bvh_hit3f; };
template <> struct HitOf<double> { using Type = bvh_hit3d; };

// ---- error plumbing --------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
std::string current_error();                   // capi.hip: the calling thread's message (a worker thread hands its own to the thread that started it)

#define BVH_HIP_TRY(expr, code)                                                                     \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return ::bvh_amd::fail((code), std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

#define BVH_HIP_TRY_PTR(expr)                                                                       \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            ::bvh_amd::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                \
            return nullptr;                                                                         \
        }                                                                                           \
    } while (0)

enum : int { BVH_AMD_OK = 0, BVH_AMD_ERR_HIP = -1, BVH_AMD_ERR_ARG = -2, BVH_AMD_ERR_UNSUPPORTED = -3,
             BVH_AMD_ERR_OVERFLOW = -4 };

// ---- the object behind `struct bvh3f` / `struct bvh3d` ---------------------------------------------
template <typename T>
struct BvhImpl {
    // host mirror, reference layout (what the accessor API and serialize() read). After a device build it is filled
    // lazily from d_nodes / d_prim_ids by sync_host(): GPU-resident workflows never pay the device-to-host copy.
    mutable std::vector<HostNode<T>> nodes;
    mutable std::vector<size_t> prim_ids;
    mutable std::atomic<bool> host_valid{true};
    mutable std::mutex host_mutex;             // the lazy fill below may be triggered by concurrent const accessors (bvhXX_get_prim_id in callbacks)
    size_t node_count = 0, prim_count = 0;
    // 3, or 2 for the `2f` / `2d` families (Node<T, 2>): everything on the device stays three wide with z = 0 (inert), only the
    // decisions (half area, widest axis, split candidates, slab test, circle test) know the dimension. The host mirror of
    // a 2D BVH is `nodes2` in the reference's 20/40-byte layout; `nodes` then only stages transfers.
    int dim = 3;
    mutable std::vector<HostNode2<T>> nodes2;
    mutable std::atomic<bool> nodes2_valid{false};         // nodes2 mirrors `nodes` (it goes stale whenever `nodes` is refreshed from the device)
    int sync_host2() const;                    // sync_host() + nodes -> nodes2
    void widen_host() const;                   // nodes2 -> nodes (z = 0): the caller may have edited nodes2 through bvh_node2X pointers
    HostNode<T>* d_nodes = nullptr;            // reference-layout nodes resident in HBM (device builds)
    size_t d_nodes_count = 0;                  // its capacity in nodes (append/remove_last_node change node_count on the host)
    T root_bounds[6] = {0, 0, 0, 0, 0, 0};
    int sync_host() const;
    // device copy
    int device = -1;
    PairNode<T>* d_pairs = nullptr;            // (node_count - 1) / 2 records
    size_t pair_count = 0;
    uint32_t* d_prim_ids = nullptr;            // prim_count
    uint32_t root_index = 0;                   // nodes[0].index narrowed to 32 bits
    // Deepest level of the tree (root = 0), computed lazily on the device by the first batch traversal after a (re)layout.
    // The traversal stack never holds more entries than that: trees up to 64 levels use LDS + per-lane scratch (the
    // reference's SmallStack<Index, 64>), deeper ones additionally spill to d_deep (its GrowingStack, stack.h:34-46).
    // (the spill buffer and the scratch of the optional ray-coherence sort are stream-ordered allocations of each launch)
    mutable std::atomic<int> max_depth{-1};
    // Sum over the inner nodes of area(node) / area(root) = the pair records a random line through the scene is expected to fetch
    // (upload.hip: tree_depth fills it together with max_depth, before max_depth). Low for scenes a ray crosses quickly, in the
    // hundreds for a soup; decides whether reordering a ray batch is worth its fixed cost per ray (traverse.hip).
    mutable std::atomic<float> expected_visits{0.0f};
    // How large batches are traced through THIS tree, found by measurement (traverse.hip: calibrate): index 0 closest-hit, 1 any-hit.
    // 0 = not measured yet; otherwise 1 | reorder << 1 | coop << 2 | refill << 8 | leaf << 16. Reset whenever the tree is re-laid out.
    mutable std::atomic<uint32_t> launch_plan[2] = {};
    struct PlanSearch {                        // the measurement in progress: one candidate per large batch, timed by events on its stream
        int index = 0;                         // measurements made so far (the predictor's plan goes first)
        static constexpr int kCandidates = 5;
        int trying = -1;                       // the candidate the pending measurement belongs to
        uint8_t count[kCandidates] = {};       // measurements of each candidate (at most two; the better time counts)
        uint8_t dropped = 0;                   // candidates out of the race (bit c): > 10 % behind the best after a measurement, or of a
                                               // family (reordered / as given) that lost by > 25 % — they are not traced again
        bool pending = false;                  // a thread has claimed the events for candidate `index`
        bool recorded = false;                 // ... and both of them are on its stream: only then may another call read them
        hipEvent_t start = nullptr, stop = nullptr;
        size_t rays = 0;
        float ns_per_ray[kCandidates] = {};
        size_t rays_of[kCandidates] = {};      // the batch size the kept time of each candidate was measured on (times of very different batches are not compared)
    };
    mutable PlanSearch plan_search[2];
    mutable std::mutex plan_mutex;
    // Batch launches are re-entrant like the reference's Bvh::intersect on a const Bvh: every launch takes the next of
    // kWorkSlots {ray ticket counter, status word} slots, so launches of one BVH issued from several threads / on several streams
    // do not share a counter (up to kWorkSlots of them in flight at a time).
    // A slot is claimed under work_mutex and stays busy until its launch has recorded the slot's event on its stream; the next
    // launch that takes the slot is ordered behind that event, so launches in flight never share a counter however many host
    // threads are between their claim and their record (a 65th concurrent claimant waits for a slot to be released).
    static constexpr uint32_t kWorkSlots = 64, kWorkStride = 4096;     // slots of 32 KB: up to 256 ticket counters 128 bytes apart
    unsigned long long* d_work = nullptr;      // kWorkSlots x kWorkStride words: the ticket counters of the launch that holds the slot
    mutable std::atomic<uint32_t> work_next{0};
    mutable hipEvent_t work_done[kWorkSlots] = {};      // recorded behind the slot's latest launch (created on first use)
    mutable bool work_busy[kWorkSlots] = {};            // claimed by a launch that has not recorded work_done yet (under work_mutex)
    mutable std::mutex work_mutex;
    ~BvhImpl();
};

// Environment switches. A RELEASE library reads exactly the variables documented in include/bvh_amd.h ("Environment": BVH_AMD_CACHE_MB,
// BVH_AMD_POOL, BVH_AMD_CALIBRATE, BVH_AMD_REINSERT, BVH_AMD_RCCL_LIB and the platform's ROCM_PATH). A/B switches, profiling aids and
// fault injection exist only in a library compiled with -DBVH_AMD_DEVELOPER (`python -m bvh_amd.build --developer` ->
// bvh_amd/lib/libbvh_amd_dev.so, what the fault-injection tests load): in a release build the macros below drop the variable's name
// at preprocessing time, so that neither the switch nor its string exists in the product.
#if defined(BVH_AMD_DEVELOPER)
inline int dev_env_int_(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
inline bool dev_env_is_(const char* name, const char* value) { const char* v = std::getenv(name); return v && std::strcmp(v, value) == 0; }
#define BVH_DEV_INT(name, dflt) (::bvh_amd::dev_env_int_(name, dflt))
#define BVH_DEV_STR(name) (static_cast<const char*>(std::getenv(name)))
#define BVH_DEV_IS(name, value) (::bvh_amd::dev_env_is_(name, value))
#else
#define BVH_DEV_INT(name, dflt) (dflt)
#define BVH_DEV_STR(name) (static_cast<const char*>(nullptr))
#define BVH_DEV_IS(name, value) (false)
#endif

// The SplitHeuristic (reference split_heuristic.h:17-23) of the build in progress on the calling thread. The reference's C
// struct bvh_build_config has no room for it, so the C-ABI entry points that accept one (bvh_amd_sah_config) set it for the
// duration of the call and the builders read it where they fill their kernel arguments; the default is the reference's {0, 1}.
// bin_count: BinnedSahBuilder's BinCount template argument (binned_sah_builder.h:18), read by the explicit binned builder only — the
// mini-tree builder and DefaultBuilder instantiate the default (mini_tree_builder.h:129, default_builder.h:52) and keep 8.
struct SahParams { uint32_t log_cluster = 0; double cost_ratio = 1.0; uint32_t bin_count = 8; };
SahParams& ambient_sah();                     // build_device.hip
struct SahScope {
    SahParams saved;
    explicit SahScope(const SahParams& p) : saved(ambient_sah()) { ambient_sah() = p; }
    ~SahScope() { ambient_sah() = saved; }
    SahScope(const SahScope&) = delete;
    SahScope& operator=(const SahScope&) = delete;
};

// Scratch memory of the builders / optimizer / sorts (build_common.h: DevBuf) comes from the device's stream-ordered pool
// (hipMallocAsync / hipFreeAsync on the stream of the operation in progress on the calling thread, release threshold raised so
// freed blocks stay cached): a build makes dozens of allocations, and hipMalloc / hipFree cost 50-300 us each and hipFree
// synchronises the device (measured: ~2.4 ms of a 10.9 ms 1M-triangle build). bvh_amd_release_cached_memory() trims the pool;
// BVH_AMD_POOL=0 restores plain hipMalloc / hipFree.
hipStream_t& ambient_stream();                 // build_device.hip
// The stream of the API call in progress on this thread. The OUTERMOST scope of a call on its stream is also what the scratch cache
// hangs its bookkeeping on: if the call freed scratch, one event (the scope's fence) is recorded on the stream when the scope ends.
struct ScratchScope;                           // build_device.hip
void scratch_forget(void* p);                  // build_device.hip (developer builds: bookkeeping of the overlap check; otherwise nothing)
struct StreamScope {
    hipStream_t saved;
    ScratchScope* scope;                       // nullptr: shares the enclosing scope's (same stream)
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
};
bool scratch_pool_enabled();                   // build_device.hip: configures the current device's default pool on first use
// Scratch comes from the runtime's stream-ordered pool, but only ever through ONE library-owned stream per device — the pool never
// sees a caller's stream, so nothing the library caches or a Bvh owns can outlive a handle it depends on (build_device.hip, round 5).
// On top of the pool sits a small block cache: hipMallocAsync / hipFreeAsync still cost ~5 / ~13 us of host time per call (measured,
// tools/src/alloc_bench.hip), ~1 ms over the ~60 scratch buffers of a build, so a freed block is kept together with the fence of the
// API call that freed it and handed to the next request of about the same size, behind that fence. A block whose pointer is taken
// over by a BvhImpl simply never comes back to the cache; the BvhImpl releases it with hipFree, which accepts pool memory.
// BVH_AMD_CACHE_MB bounds the cached bytes per device (default min(1024, 5 % of the free HBM); 0 turns the cache off).
struct ScratchTag {                            // what scratch_free needs to know about a block
    size_t capacity = 0;                       // bytes actually behind the pointer (>= the request)
    bool pooled = false;
};
hipError_t scratch_alloc(void** p, size_t bytes, ScratchTag* tag);
void scratch_free(void* p, const ScratchTag& tag);
void scratch_cache_flush();                    // hands every cached block of the current device back to the pool
size_t scratch_cache_bytes();                  // bytes the cache holds on the current device / its bound
size_t scratch_cache_limit();

// A few bytes (<= 256, 4-byte aligned) from device memory to the host, in stream order, WITHOUT a stream synchronisation: a
// one-lane kernel copies them into coherent pinned host memory and publishes a sequence number, the host spins on it (falls back to
// hipMemcpyAsync + hipStreamSynchronize after a timeout or when pinned memory is unavailable). When it returns, everything queued on
// the stream before it has run. The builders size their next launch ~20 times per build from such a read; a blocking
// synchronisation costs 30-60 us of idle GPU each time, the poll a fraction of that. BVH_AMD_READBACK=sync restores the blocking form.
int readback(void* dst, const void* d_src, size_t bytes, hipStream_t stream);     // build_device.hip

// upload.hip
template <typename T> int upload_bvh(BvhImpl<T>& b, hipStream_t stream);
template <typename T> int tree_depth(const BvhImpl<T>& b, hipStream_t stream);   // fills b.max_depth (cached)
template <typename T> int prepare_trace(const BvhImpl<T>& b, size_t n_rays_hint, hipStream_t stream);   // traverse.hip

template <typename T> int relayout_on_device(BvhImpl<T>& b, const HostNode<T>* d_nodes, hipStream_t stream);

// wire.hip: the Bvh::serialize byte stream in device memory, and the structural checks the traversal records rely on
template <typename T> size_t wire_size(const BvhImpl<T>& b);
template <typename T> size_t serialize_to_device(const BvhImpl<T>& b, void* d_out, size_t cap, hipStream_t stream);
template <typename T> BvhImpl<T>* deserialize_from_device(const void* d_bytes, size_t size, int dim, hipStream_t stream);
template <typename T> int nodes_resident(BvhImpl<T>& b);   // capi.hip: reference-layout nodes resident on the device, host edits pushed and re-validated
template <typename T> int validate_resident_nodes(const HostNode<T>* d_nodes, size_t nn, size_t np, hipStream_t stream, const char* who);

// traverse.hip
template <typename T>
int launch_traverse(const BvhImpl<T>& b, int leaf_kind, const T* d_prims, const T* d_rays, size_t n, unsigned flags,
                    typename HitOf<T>::Type* d_hits, bvh_amd_counters* d_counters, hipStream_t stream);
void last_launch_plan(int out[4]);                            // traverse.hip: {reordered, coop, refill, leaf} of the calling thread's latest launch
int wave_times(unsigned long long* out, size_t capacity_waves, size_t* n_waves);   // traverse.hip: developer library only
void last_plan_search(float ns_per_ray[5], int measurements[5], unsigned* dropped);   // traverse.hip
int set_experiment(const char* name, int value);                             // traverse.hip: developer experiments of the calling thread
void set_tuning(int refill, int leaf, int coop, int parts);               // traverse.hip: per-thread overrides for A/B runs (< 0: default)
const char* last_kernel_name();
bool last_launch_reordered();
void kernel_timing(bool on);                                  // traverse.hip
int kernel_times(float* ms_out, size_t capacity, size_t* count_out);
int reorder_times(float* ms_out, size_t capacity, size_t* count_out);
template <typename T>
int trace_ray_callbacks(const BvhImpl<T>& b, const T ray8[8], uint32_t start, bool any, bool robust,
                        bool (*leaf_fn)(void*, T*, size_t, size_t), void (*inner_fn)(void*, size_t), void* user);

// prep.hip
template <typename T> int launch_tri_bounds(const T* d_tris9, size_t n, T* d_bb, T* d_cc, hipStream_t s);
template <typename T> int launch_precompute_tris(const T* d_tris9, const uint32_t* d_perm, size_t n, T* d_out, hipStream_t s);
template <typename T> int launch_sphere_bounds(const T* d_sph4, size_t n, T* d_bb, T* d_cc, hipStream_t s);
int launch_gather(const void* d_in, const uint32_t* d_perm, size_t n, size_t stride, void* d_out, hipStream_t s);
template <typename T> int launch_circle_bounds(const T* d_circles3, size_t n, T* d_bb4, T* d_cc2, hipStream_t s);
template <typename T> int launch_widen_inputs(const T* d_bb4, const T* d_cc2, size_t n, T* d_bb6, T* d_cc3, hipStream_t s);

// render.hip
template <typename T> int launch_pinhole_rays(const T eye[3], const T dir[3], const T up[3], size_t width, size_t height, T* d_rays, hipStream_t s);
template <typename T>
int launch_shade_eyelight(const T* d_tris12, const T* d_rays, const typename HitOf<T>::Type* d_hits, size_t n, uint8_t* d_rgb, hipStream_t s);

// sort_emul.hip
template <typename K>
int radix_sort_pairs(K* keys, uint32_t* vals, K* keys_tmp, uint32_t* vals_tmp, uint32_t n, uint32_t batch, int bits, hipStream_t stream,
                     uint32_t* hist_buf = nullptr, bool iota_vals = false, bool keys_wanted = true, uint32_t** vals_result = nullptr,
                     bool first_hist_done = false);
// (first_hist_done: the caller's key kernel has already left the first pass's per-tile digit histogram in hist_buf — tiles of kRadixTileU32
//  keys, digit-major / tile-minor, batch == 1 — so the first k_radix_hist is not launched; traverse.hip: ray_keys_kernel)
constexpr int kRadixTileU32 = 8192;
// (iota_vals: `vals` need not be initialised, the values are the input positions 0..n-1 of each array; keys_wanted = false: only
//  the values are sorted on return, `keys` is left in an unspecified state; vals_result != nullptr: exactly ceil(bits / 8) passes,
//  *vals_result = whichever of vals / vals_tmp holds the sorted values, keys unspecified)
size_t radix_sort_hist_words(uint32_t n, uint32_t batch);
template <typename T>
int std_sort_ids(uint32_t* d_ids, const T* d_keys, uint32_t n, uint32_t batch, uint32_t astride, uint32_t istride, hipStream_t stream);

// out[i] = sum of in[0..i); the total is optionally returned to the host (synchronises the stream)
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_host, hipStream_t stream);

// build_*.hip
template <typename T>
int build_on_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                    bvh_amd_builder builder, hipStream_t stream);

enum { LEAF_TRIANGLE = 0, LEAF_SPHERE = 1 };

} // namespace bvh_amd
