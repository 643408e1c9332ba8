/* bvh_amd — MI355X-native BVH construction and traversal behind madmann91/bvh's C bindings.
 *
 * This header is the C-ABI boundary of libbvh_amd.so. It has two parts:
 *
 *  (1) the reference's own C API (reference src/bvh/v2/c_api/bvh.h), re-declared here with the same
 *      names, argument meaning and ownership rules so that code written against libbvh_c.so links
 *      against libbvh_amd.so unchanged. Each declaration cites the reference line it replaces.
 *      `struct bvh3f` etc. stay opaque; ours additionally owns the device-resident copy.
 *
 *  (2) additive, callback-free batch entry points (`*_device`, `*_rays_*`). The reference's
 *      per-ray `bvhXX_intersect_ray(bvh, ray, callback)` (c_api/bvh.h:277-295) takes a host function
 *      pointer per leaf: it is served (the walk runs on the device, the leaves come back to the host
 *      callback in the reference's order), but it costs kernel launches per ray; the batch family is what
 *      a maintainer binds for the hot path (INTEGRATION.md shows the binding). Buffers named d_* are DEVICE pointers
 *      (HBM of the current HIP device); all others are host pointers. `stream` is a hipStream_t
 *      passed as void* (NULL = the default stream); batch calls are asynchronous on it.
 *
 * Error behaviour: the reference reports no errors at all. Functions here that
 * return int return 0 on success and a negative code on failure; pointer-returning functions return
 * NULL on failure; `bvh_amd_last_error()` gives the thread's last message. There is NO CPU fallback:
 * if no MI355X/HIP device is usable the calls fail.
 */
#ifndef BVH_AMD_H
#define BVH_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVH_AMD_API __attribute__((visibility("default")))

#define BVH_ROOT_INDEX 0                     /* c_api/bvh.h:32 */
#define BVH_INVALID_PRIM_ID SIZE_MAX         /* c_api/bvh.h:33 */

/* ---- types shared with the reference (c_api/bvh.h:35-83) ---------------------------------- */
struct bvh3f;
struct bvh3d;
struct bvh_node3f;
struct bvh_node3d;
struct bvh2f;                                   /* c_api/bvh.h:35-43: the 2D families (Node<T, 2>, 20/40-byte nodes) */
struct bvh2d;
struct bvh_node2f;
struct bvh_node2d;
struct bvh_thread_pool;

enum bvh_build_quality { BVH_BUILD_QUALITY_LOW, BVH_BUILD_QUALITY_MEDIUM, BVH_BUILD_QUALITY_HIGH };

struct bvh_build_config {                    /* c_api/bvh.h:53-58 */
    enum bvh_build_quality quality;
    size_t min_leaf_size;
    size_t max_leaf_size;
    size_t parallel_threshold;
};

struct bvh_vec2f { float x, y; };
struct bvh_vec2d { double x, y; };
struct bvh_bbox2f { struct bvh_vec2f min, max; };
struct bvh_bbox2d { struct bvh_vec2d min, max; };
struct bvh_ray2f { struct bvh_vec2f org, dir; float tmin, tmax; };
struct bvh_ray2d { struct bvh_vec2d org, dir; double tmin, tmax; };
struct bvh_vec3f { float x, y, z; };
struct bvh_vec3d { double x, y, z; };
struct bvh_bbox3f { struct bvh_vec3f min, max; };
struct bvh_bbox3d { struct bvh_vec3d min, max; };
struct bvh_ray3f { struct bvh_vec3f org, dir; float tmin, tmax; };
struct bvh_ray3d { struct bvh_vec3d org, dir; double tmin, tmax; };

/* c_api/bvh.h:75-83: leaf callback of the per-ray entry points. user_fn(user_data, &tmax, begin, end) intersects the
 * primitives [begin, end) (BVH order), returns true on a hit and may shorten the ray through the pointer. */
struct bvh_intersect_callbackf {
    void* user_data;
    bool (*user_fn)(void*, float*, size_t begin, size_t end);
};
struct bvh_intersect_callbackd {
    void* user_data;
    bool (*user_fn)(void*, double*, size_t begin, size_t end);
};

/* Additive: the same with the reference's optional InnerFn (bvh.h:72-73): inner_fn(user_data, first_child_id) is called for
 * every visited pair of siblings (nodes first_child_id and first_child_id + 1) before its boxes are tested; may be NULL. */
struct bvh_amd_ray_visitorf {
    void* user_data;
    bool (*leaf_fn)(void*, float*, size_t begin, size_t end);
    void (*inner_fn)(void*, size_t first_child_id);
};
struct bvh_amd_ray_visitord {
    void* user_data;
    bool (*leaf_fn)(void*, double*, size_t begin, size_t end);
    void (*inner_fn)(void*, size_t first_child_id);
};

/* ---- additive types --------------------------------------------------------------------------- */

/* One record per ray. prim = BVH-order primitive index (the `i` the reference's leaf callback
 * receives, bvh.h:152; original id = bvhXX_get_prim_id(bvh, prim)), or BVH_AMD_INVALID on a miss, in
 * which case t = the ray's input tmax. Triangles: (t,u,v) of PrecomputedTri::intersect (tri.h:56-74).
 * Spheres: t = t0, u = t1 of Sphere::intersect (sphere.h:32-49), v = 0. */
#define BVH_AMD_INVALID 0xFFFFFFFFu
struct bvh_hit3f { uint32_t prim; float t, u, v; };
struct bvh_hit3d { uint32_t prim; uint32_t pad; double t, u, v; };

/* Traversal statistics (the reference's optional InnerFn/leaf counters, test/benchmark.cpp:258-296).
 * Sums over the batch: inner-node pair visits, primitive tests, leaf visits. */
struct bvh_amd_counters { unsigned long long node_pairs, prim_tests, leaves; };

enum bvh_amd_ray_flags {
    BVH_AMD_RAY_ANY_HIT = 1u,  /* Bvh::intersect<IsAnyHit = true>: no near/far reordering, stop at first hit */
    BVH_AMD_RAY_ROBUST  = 2u,  /* Bvh::intersect<IsRobust = true>: Ize's robust slab test (node.h:68-77)      */
    BVH_AMD_RAY_SORTED  = 4u,  /* always reorder the batch internally for coherence (a 24-bit origin-cell / direction-octant key,
                                  three radix passes, ~0.45 ms per 16M rays); per-ray results are unchanged. With neither this
                                  flag nor BVH_AMD_RAY_UNSORTED the library decides: it reorders batches of >= 1M rays on trees
                                  whose node records exceed the 32 MB of L2 and through which a random line is expected to fetch
                                  >= 100 records (sum of area(node) / area(root) over the inner nodes): +40..50 % there      */
    BVH_AMD_RAY_UNSORTED = 16u, /* never reorder (rays already coherent, or no scratch memory to spare: 16 bytes per ray)  */
    BVH_AMD_RAY_ORIGINAL_IDS = 8u  /* hit.prim = the ORIGINAL primitive id bvh.prim_ids[i] (what c_api_example.c:265-268 looks up per
                                      hit) instead of the BVH-order index i: one more pass over the hit records, on the device    */
};

/* Which reference builder a device build reproduces (bit-exact node/prim order). */
enum bvh_amd_builder {
    BVH_AMD_BUILDER_DEFAULT_SERIAL   = 0, /* DefaultBuilder::build(bboxes, centers, cfg), default_builder.h:49  */
    BVH_AMD_BUILDER_DEFAULT_PARALLEL = 1, /* DefaultBuilder::build(pool, ...), default_builder.h:33             */
    BVH_AMD_BUILDER_BINNED           = 2, /* BinnedSahBuilder::build, binned_sah_builder.h:32                   */
    BVH_AMD_BUILDER_SWEEP            = 3  /* SweepSahBuilder::build, sweep_sah_builder.h:30                     */
};

/* ---- library / device ------------------------------------------------------------------------- */
BVH_AMD_API const char* bvh_amd_last_error(void);
BVH_AMD_API const char* bvh_amd_version(void);
BVH_AMD_API int bvh_amd_device_count(void);            /* < 0 on HIP failure */
BVH_AMD_API int bvh_amd_device_name(int device, char* out, size_t cap);

/* Device memory helpers (thin wrappers over hipMalloc / hipMemcpy) so that C and C++ callers need no HIP headers. */
BVH_AMD_API void* bvh_amd_device_alloc(size_t bytes);                         /* NULL on failure */
BVH_AMD_API void bvh_amd_device_free(void* d_ptr);
BVH_AMD_API int bvh_amd_copy_to_device(void* d_dst, const void* h_src, size_t bytes);
BVH_AMD_API int bvh_amd_copy_to_host(void* h_dst, const void* d_src, size_t bytes);
BVH_AMD_API int bvh_amd_synchronize(void* stream);
/* Builds, optimize, the sorts and reordered ray batches take their scratch from the device's stream-ordered memory pool — always
 * through a stream the LIBRARY owns, never the caller's — and keep freed blocks in a small cache for the next call, each with an
 * event that marks the end of the call that freed it. Consequences a caller can rely on (round 5; c_api/bvh.h:129-132 has no
 * lifetime rule beyond _destroy, and neither has this library): a `stream` argument is only used during the call it is passed to;
 * it may be destroyed afterwards, whatever was built or cached while it lived (tests/c/stream_lifetime.c).
 * The cache holds at most BVH_AMD_CACHE_MB megabytes per device (default: min(1024, 5 % of the HBM free at first use); 0 = no
 * cache). Scratch the cache does not keep goes back to the device's default memory pool and STAYS MAPPED there for the next
 * request: the pool's release threshold is set to "never on its own", because memory that the pool unmaps at some synchronisation
 * and maps again right afterwards was read stale by compute kernels on this platform (profiles/r05_pool_trim_stale_reads.txt). So
 * a process keeps reserved what its largest call needed (a 10M-triangle build cycles ~3 GB of scratch) until it calls
 * bvh_amd_release_cached_memory(), which returns cache and pool to the driver with the device idle before and after.
 * BVH_AMD_POOL=0 disables pool and cache (plain hipMalloc / hipFree).
 *
 * Environment variables a release library reads — all of them: BVH_AMD_CACHE_MB, BVH_AMD_POOL (above); BVH_AMD_CALIBRATE=0 (no
 * measured launch-plan search, the predictor's plan always); BVH_AMD_REINSERT=exact (ReinsertionOptimizer: replay the reference's
 * candidate heap in every iteration instead of only where the heap-free path cannot prove the same result); BVH_AMD_RCCL_LIB (path of librccl for the multi-GPU
 * entry points; else the loader's search path, $ROCM_PATH/lib, /opt/rocm/lib). A/B switches, profiling aids and fault injection
 * exist only in the developer build (python -m bvh_amd.build --developer -> libbvh_amd_dev.so, -DBVH_AMD_DEVELOPER).          */
BVH_AMD_API int bvh_amd_release_cached_memory(void);
BVH_AMD_API size_t bvh_amd_cached_scratch_bytes(void);     /* bytes the block cache holds on the current device right now */
BVH_AMD_API size_t bvh_amd_scratch_cache_limit(void);      /* its bound on the current device */
/* Measurement aid for bench.py (csrc/probe.hip): mean launch time of a dependent walk over a table of 64-byte records (word 0 of
 * a record = index of the next one), one record in flight per lane — the rate the memory system gives the traversal's access
 * pattern when nothing else is in the way. Not used by any product path. */
BVH_AMD_API int bvh_amd_probe_record_walk(const void* d_table, uint32_t n_records, uint32_t steps, int blocks_per_cu, int reps,
                                          float* ms_out, unsigned long long* records_out, void* stream);
/* The same walk with the way a record reaches its lane selectable (csrc/probe.hip): mode 0 per-lane loads (as above), 1 quad-
 * cooperative loads + LDS transpose, 2 / 3 one chain per quad loaded by the quad / by its first lane; `active` = lanes of a wave
 * that own a chain (1..64); 4 quad-cooperative loads + in-register transpose (the shipped fetch); 5 / 6 records of 128 bytes, one chain per
 * octet of lanes loaded by the octet / by its first lane (n_records then counts 128-byte records). A table of a few KB measures the L1,
 * a few MB the L2, beyond 32 MB the fabric side. */
/* One chain per lane that alternates between a small (L2-resident) and a big (beyond the L2s) table, one fetch in flight per lane:
 * do the times of the two levels add (one shared resource: the CU's outstanding lines) or overlap? (csrc/probe.hip) */
BVH_AMD_API int bvh_amd_probe_mixed_walk(const void* d_small, uint32_t n_small, const void* d_big, uint32_t n_big, uint32_t steps, int blocks_per_cu, int reps,
                                         int coop, float* ms_out, unsigned long long* records_out, void* stream);
BVH_AMD_API int bvh_amd_probe_record_walk_ex(const void* d_table, uint32_t n_records, uint32_t steps, int blocks_per_cu, int reps, int mode, int active,
                                             float* ms_out, unsigned long long* records_out, void* stream);

/* ---- multi-GPU (the "required extension" of 8b: device select + replicate) ------------------------------------
 * Ray batches shard embarrassingly (a const Bvh + per-ray state, reference bvh.h:160-182 touches nothing shared); the ONE
 * exchange of the path is a root-to-all broadcast of the scene over RCCL / xGMI: the Bvh::serialize byte stream (reference
 * bvh.h:221-229) written into HBM straight from the resident nodes, and the BVH-ordered primitive array. It runs inside this
 * library (csrc/replicate.hip; librccl is opened on first use); no payload byte visits the host on any rank. A BVH is bound to the device
 * it was built / received on; bvhXX_intersect_rays_* must be called with that device current (bvh_amd_device_select).
 *   one process per GPU: rank 0 calls bvh_amd_comm_unique_id, the 128 bytes travel by any side channel (a file, MPI, torch's
 *     store), every rank calls bvh_amd_comm_create on ITS device, then bvhXX_broadcast; or wrap an ncclComm_t you already have
 *     with bvh_amd_comm_adopt;
 *   one process, N GPUs (what a C caller of the reference API would do): bvhXX_replicate.                                        */
#define BVH_AMD_COMM_ID_BYTES 128
struct bvh_amd_comm;                                   /* an RCCL communicator + its rank / size / device */
BVH_AMD_API int bvh_amd_device_select(int device);     /* hipSetDevice for the calling thread: following builds / uploads live there */
BVH_AMD_API int bvh_amd_device_current(void);          /* < 0 on HIP failure */
BVH_AMD_API int bvh_amd_comm_unique_id(void* id_out /* BVH_AMD_COMM_ID_BYTES */);                   /* ncclGetUniqueId */
BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_create(const void* id, int n_ranks, int rank);        /* ncclCommInitRank on the current device; NULL on failure */
BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_adopt(void* nccl_comm /* ncclComm_t, stays yours */);
BVH_AMD_API void bvh_amd_comm_destroy(struct bvh_amd_comm*);
BVH_AMD_API int bvh_amd_comm_rank(const struct bvh_amd_comm*);
BVH_AMD_API int bvh_amd_comm_size(const struct bvh_amd_comm*);
BVH_AMD_API void* bvh_amd_comm_handle(const struct bvh_amd_comm*);                                  /* the ncclComm_t */
/* ncclBroadcast of a raw device buffer, in place, on `stream` (e.g. the scene box the ray generators need) */
BVH_AMD_API int bvh_amd_comm_broadcast(struct bvh_amd_comm*, void* d_buf, size_t bytes, int root, void* stream);
/* bvhXX_replicate keeps its communicators (one set per list of devices: ncclCommInitAll is paid once, not per scene); this destroys
 * them (also done by bvh_amd_release_cached_memory) and returns the number of sets there were.                                   */
BVH_AMD_API int bvh_amd_comm_cache_clear(void);
/* The path librccl was opened from ("" until the first multi-GPU call, or when it could not be opened: bvh_amd_last_error). RCCL is
 * loaded on first use of the entry points of this section — a single-GPU program never needs it to be installed.                 */
BVH_AMD_API const char* bvh_amd_rccl_library(void);
/* Collective: EVERY rank of the communicator calls it. The root passes its BVH and the BVH-ordered primitive array
 * (prim_bytes bytes of device memory); the others pass NULL / NULL / 0. Returns this rank's device-resident BVH — the root's own
 * object on the root, a new one elsewhere (bvhXX_destroy) — and in *d_prims_out this rank's primitive array (the root's own
 * pointer on the root, otherwise a new buffer to release with bvh_amd_device_free); *prim_bytes_out (may be NULL) its size.
 * NULL on failure (bvh_amd_last_error). No rank is left waiting by a peer's failure: a root that has nothing valid to send says so
 * in the header, and after the header every rank prepares its side (family check, receive buffers, the root's serialization) and
 * the ranks agree on one status word (ncclAllReduce, min) BEFORE any payload is posted — if any rank failed, all return NULL. When it returns, the BVH is ready on `stream`'s device and the buffers may be used from any stream.                 */
BVH_AMD_API struct bvh3f* bvh3f_broadcast(struct bvh_amd_comm*, int root, struct bvh3f* bvh, const void* d_prims, size_t prim_bytes,
                                          void** d_prims_out, size_t* prim_bytes_out, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_broadcast(struct bvh_amd_comm*, int root, struct bvh3d* bvh, const void* d_prims, size_t prim_bytes,
                                          void** d_prims_out, size_t* prim_bytes_out, void* stream);
/* One process, several GPUs: copies the scene from the BVH's own device to every other device of `devices` (NULL: 0 ..
 * n_devices - 1; the BVH's device must be among them) with ncclCommInitAll + one grouped ncclBroadcast. bvhs_out[i] /
 * d_prims_out[i] belong to devices[i]; the entry of the BVH's own device holds the original pointers, the others are new
 * (bvhXX_destroy / bvh_amd_device_free with that device current). The calling thread's current device is left unchanged.      */
BVH_AMD_API int bvh3f_replicate(struct bvh3f* bvh, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices,
                                struct bvh3f** bvhs_out, void** d_prims_out);
BVH_AMD_API int bvh3d_replicate(struct bvh3d* bvh, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices,
                                struct bvh3d** bvhs_out, void** d_prims_out);

/* ---- thread pool (c_api/bvh.h:90-91). Kept for signature compatibility; the GPU grid replaces it.
 * A non-NULL pool selects the reference's *parallel* builder semantics (mini-trees). ------------- */
BVH_AMD_API struct bvh_thread_pool* bvh_thread_pool_create(size_t thread_count);
BVH_AMD_API void bvh_thread_pool_destroy(struct bvh_thread_pool*);

/* ---- construction ------------------------------------------------------------------------------ */
/* c_api/bvh.h:106-118. Host arrays in; the build runs on the current HIP device; the returned object
 * holds the reference-layout host mirror (for the accessors below) and the device-resident copy. */
BVH_AMD_API struct bvh3f* bvh3f_build(struct bvh_thread_pool*, const struct bvh_bbox3f* bboxes,
    const struct bvh_vec3f* centers, size_t prim_count, const struct bvh_build_config* config);
BVH_AMD_API struct bvh3d* bvh3d_build(struct bvh_thread_pool*, const struct bvh_bbox3d* bboxes,
    const struct bvh_vec3d* centers, size_t prim_count, const struct bvh_build_config* config);

/* Additive: inputs already resident in HBM (n x {min,max} and n x center, tightly packed). */
BVH_AMD_API struct bvh3f* bvh3f_build_device(const float* d_bboxes, const float* d_centers, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_build_device(const double* d_bboxes, const double* d_centers, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, void* stream);

/* Additive: TopDownSahBuilder::Config::sah, the SplitHeuristic every builder's Config carries (src/bvh/v2/split_heuristic.h:17-38,
 * top_down_sah_builder.h:27-30) and the reference's C struct bvh_build_config has no field for: a range of `size` primitives
 * costs ceil(size / 2^log_cluster_size) primitive intersections, and a node that is not split saves `cost_ratio` (ratio of the
 * cost of a ray-box test over a ray-primitive test). NULL = the reference's defaults {0, 1}. bvhXX_build_sah is bvhXX_build
 * (DefaultBuilder, pool or not) with it; bvhXX_build_device_sah is bvhXX_build_device with it. */
struct bvh_amd_sah_config {
    size_t log_cluster_size;                   /* default 0 */
    double cost_ratio;                         /* default 1 */
};
BVH_AMD_API struct bvh3f* bvh3f_build_sah(struct bvh_thread_pool*, const struct bvh_bbox3f* bboxes, const struct bvh_vec3f* centers,
    size_t prim_count, const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah);
BVH_AMD_API struct bvh3d* bvh3d_build_sah(struct bvh_thread_pool*, const struct bvh_bbox3d* bboxes, const struct bvh_vec3d* centers,
    size_t prim_count, const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah);
BVH_AMD_API struct bvh3f* bvh3f_build_device_sah(const float* d_bboxes, const float* d_centers, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, const struct bvh_amd_sah_config* sah, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_build_device_sah(const double* d_bboxes, const double* d_centers, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, const struct bvh_amd_sah_config* sah, void* stream);

/* Additive: BinnedSahBuilder<Node, BinCount>::build (src/bvh/v2/binned_sah_builder.h:18, :32-38) with its BinCount template argument
 * as a run-time value: 4, 8 (the reference's default, = bvhXX_build_device_sah with BVH_AMD_BUILDER_BINNED), 16 or 32 bins per axis.
 * Same inputs as bvhXX_build_device_sah; NULL for anything else (bvh_amd_last_error). DefaultBuilder and MiniTreeBuilder always
 * instantiate the default (default_builder.h:52, mini_tree_builder.h:129), so only a direct user of BinnedSahBuilder has this knob. */
BVH_AMD_API struct bvh3f* bvh3f_build_device_binned(const float* d_bboxes, const float* d_centers, size_t prim_count,
    const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah, size_t bin_count, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_build_device_binned(const double* d_bboxes, const double* d_centers, size_t prim_count,
    const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah, size_t bin_count, void* stream);

/* Additive: MiniTreeBuilder::build(pool, bboxes, centers, config) itself (src/bvh/v2/mini_tree_builder.h:29-58) with its own
 * configuration; DefaultBuilder(pool)'s three qualities are three settings of it (default_builder.h:65-73). log2_grid_dim
 * may be 1..10 (three coordinates in the reference's 32-bit Morton code, mini_tree_builder.h:169); the cell histogram takes
 * 8^log2_grid_dim x 8 bytes of HBM. */
struct bvh_amd_minitree_config {
    size_t min_leaf_size, max_leaf_size;       /* TopDownSahBuilder::Config (top_down_sah_builder.h:27-40), defaults 1 / 8 */
    int enable_pruning;                        /* default 1 */
    double pruning_area_ratio;                 /* default 0.01 */
    size_t parallel_threshold;                 /* default 1024 */
    size_t log2_grid_dim;                      /* default 4 */
    size_t log_cluster_size;                   /* SplitHeuristic (split_heuristic.h:17-23), defaults 0 ... */
    double cost_ratio;                         /* ... and 1 */
};
BVH_AMD_API struct bvh3f* bvh3f_build_minitree_device(const float* d_bboxes, const float* d_centers, size_t prim_count,
    const struct bvh_amd_minitree_config* config, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_build_minitree_device(const double* d_bboxes, const double* d_centers, size_t prim_count,
    const struct bvh_amd_minitree_config* config, void* stream);

/* Additive: wrap an existing reference-layout BVH (28/56-byte nodes, node.h:31-37) and upload it. */
BVH_AMD_API struct bvh3f* bvh3f_from_nodes(const void* nodes, size_t node_count, const size_t* prim_ids, size_t prim_count);
BVH_AMD_API struct bvh3d* bvh3d_from_nodes(const void* nodes, size_t node_count, const size_t* prim_ids, size_t prim_count);

/* Additive: Bvh::extract_bvh(root_id) (src/bvh/v2/bvh.h:92-122) on the device: the subtree under `root_id` as its own BVH,
 * node order and re-packed prim ids exactly as the reference lays them out. NULL + bvh_amd_last_error() on failure. */
BVH_AMD_API struct bvh3f* bvh3f_extract(struct bvh3f* bvh, size_t root_id);
BVH_AMD_API struct bvh3d* bvh3d_extract(struct bvh3d* bvh, size_t root_id);

BVH_AMD_API void bvh3f_destroy(struct bvh3f*);                                   /* c_api/bvh.h:130 */
BVH_AMD_API void bvh3d_destroy(struct bvh3d*);                                   /* c_api/bvh.h:132 */

/* ---- optimization (c_api/bvh.h:226-229): ReinsertionOptimizer::optimize on the device, in place. The pool
 * argument is accepted for signature compatibility (the result does not depend on it in the reference either).
 * On failure the BVH is left unchanged and bvh_amd_last_error() is set. */
BVH_AMD_API void bvh3f_optimize(struct bvh_thread_pool*, struct bvh3f*);
BVH_AMD_API void bvh3d_optimize(struct bvh_thread_pool*, struct bvh3d*);
/* Additive: ReinsertionOptimizer::Config (src/bvh/v2/reinsertion_optimizer.h:18-24), which the reference's C API does not
 * expose; NULL = its defaults {0.05, 3}. Returns an error code (the void functions above only set bvh_amd_last_error). */
struct bvh_amd_optimize_config {
    double batch_size_ratio;                   /* fraction of the nodes re-inserted per iteration, default 0.05 */
    size_t max_iter_count;                     /* default 3 */
};
BVH_AMD_API int bvh3f_optimize_config(struct bvh3f*, const struct bvh_amd_optimize_config*);
BVH_AMD_API int bvh3d_optimize_config(struct bvh3d*, const struct bvh_amd_optimize_config*);

/* ---- refit and node editing (c_api/bvh.h:170-229). The setters, append and remove act on the host mirror exactly
 * like the reference (node pointers alias the mirror and are invalidated by append). `bvhXX_refit` pushes the mirror
 * to the device, recomputes every inner box bottom-up there (Bvh::refit, bvh.h:211-218) and refreshes both copies.
 * After editing WITHOUT refit/optimize, call bvhXX_sync_device before tracing (additive; 0 on success). */
BVH_AMD_API void bvh3f_refit(struct bvh3f*);
BVH_AMD_API void bvh3d_refit(struct bvh3d*);
/* The void functions above (c_api/bvh.h:218-229) cannot report a failure: on one (HIP error, reinsertion search-stack overflow)
 * they print the reason and abort rather than leave a half-updated tree behind. Callers that want to handle it use
 * bvhXX_optimize_config (NULL config = the reference's defaults) and bvhXX_refit_status: 0 or a negative code + bvh_amd_last_error(). */
BVH_AMD_API int bvh3f_refit_status(struct bvh3f*);
BVH_AMD_API int bvh3d_refit_status(struct bvh3d*);
BVH_AMD_API int bvh3f_sync_device(struct bvh3f*);
BVH_AMD_API int bvh3d_sync_device(struct bvh3d*);
BVH_AMD_API void bvh3f_append_node(struct bvh3f*);
BVH_AMD_API void bvh3d_append_node(struct bvh3d*);
BVH_AMD_API void bvh3f_remove_last_node(struct bvh3f*);
BVH_AMD_API void bvh3d_remove_last_node(struct bvh3d*);
BVH_AMD_API void bvh_node3f_set_prim_count(struct bvh_node3f*, size_t);
BVH_AMD_API void bvh_node3d_set_prim_count(struct bvh_node3d*, size_t);
BVH_AMD_API void bvh_node3f_set_first_id(struct bvh_node3f*, size_t);
BVH_AMD_API void bvh_node3d_set_first_id(struct bvh_node3d*, size_t);
BVH_AMD_API void bvh_node3f_set_bbox(struct bvh_node3f*, const struct bvh_bbox3f*);
BVH_AMD_API void bvh_node3d_set_bbox(struct bvh_node3d*, const struct bvh_bbox3d*);

/* ---- serialization (c_api/bvh.h:136-144; byte format of Bvh::serialize, bvh.h:221-229) -------- */
BVH_AMD_API void bvh3f_save(const struct bvh3f*, FILE*);
BVH_AMD_API void bvh3d_save(const struct bvh3d*, FILE*);
BVH_AMD_API struct bvh3f* bvh3f_load(FILE*);
BVH_AMD_API struct bvh3d* bvh3d_load(FILE*);
/* Additive: the same byte stream to/from memory (it is the RCCL broadcast payload). Returns the
 * stream size; writes only if cap is large enough. */
BVH_AMD_API size_t bvh3f_serialize(const struct bvh3f*, void* out, size_t cap);
BVH_AMD_API size_t bvh3d_serialize(const struct bvh3d*, void* out, size_t cap);
BVH_AMD_API struct bvh3f* bvh3f_deserialize(const void* bytes, size_t size);
BVH_AMD_API struct bvh3d* bvh3d_deserialize(const void* bytes, size_t size);
/* The same stream in DEVICE memory (bvh.h:221-243, node.h:90-102): written from / turned into the resident nodes by kernels and
 * device-to-device copies, so the RCCL broadcast of a scene moves no payload byte through the host (deserialize reads the
 * 16-byte header and the root node only). d_out / d_bytes must be aligned to the index type (4 bytes float, 8 double).
 * serialize: returns the stream size, writes (asynchronously on `stream`) only if cap suffices, 0 on error.
 * deserialize: validates the structure the traversal relies on (children adjacent at an odd index, leaf ranges inside prim_ids);
 * the buffer may be released when the call returns. */
BVH_AMD_API size_t bvh3f_serialize_device(struct bvh3f*, void* d_out, size_t cap, void* stream);
BVH_AMD_API size_t bvh3d_serialize_device(struct bvh3d*, void* d_out, size_t cap, void* stream);
BVH_AMD_API struct bvh3f* bvh3f_deserialize_device(const void* d_bytes, size_t size, void* stream);
BVH_AMD_API struct bvh3d* bvh3d_deserialize_device(const void* d_bytes, size_t size, void* stream);

/* ---- accessors (c_api/bvh.h:148-203), on the host mirror -------------------------------------- */
BVH_AMD_API struct bvh_node3f* bvh3f_get_node(struct bvh3f*, size_t);
BVH_AMD_API struct bvh_node3d* bvh3d_get_node(struct bvh3d*, size_t);
BVH_AMD_API size_t bvh3f_get_prim_id(const struct bvh3f*, size_t);
BVH_AMD_API size_t bvh3d_get_prim_id(const struct bvh3d*, size_t);
BVH_AMD_API size_t bvh3f_get_prim_count(const struct bvh3f*);
BVH_AMD_API size_t bvh3d_get_prim_count(const struct bvh3d*);
BVH_AMD_API size_t bvh3f_get_node_count(const struct bvh3f*);
BVH_AMD_API size_t bvh3d_get_node_count(const struct bvh3d*);
BVH_AMD_API bool bvh_node3f_is_leaf(const struct bvh_node3f*);
BVH_AMD_API bool bvh_node3d_is_leaf(const struct bvh_node3d*);
BVH_AMD_API size_t bvh_node3f_get_prim_count(const struct bvh_node3f*);
BVH_AMD_API size_t bvh_node3d_get_prim_count(const struct bvh_node3d*);
BVH_AMD_API size_t bvh_node3f_get_first_id(const struct bvh_node3f*);
BVH_AMD_API size_t bvh_node3d_get_first_id(const struct bvh_node3d*);
BVH_AMD_API struct bvh_bbox3f bvh_node3f_get_bbox(const struct bvh_node3f*);
BVH_AMD_API struct bvh_bbox3d bvh_node3d_get_bbox(const struct bvh_node3d*);
/* Additive bulk accessors: copy all nodes (reference layout) / prim ids out of the host mirror. */
BVH_AMD_API void bvh3f_copy_nodes(const struct bvh3f*, void* out_nodes28);
BVH_AMD_API void bvh3d_copy_nodes(const struct bvh3d*, void* out_nodes56);
BVH_AMD_API void bvh3f_copy_prim_ids(const struct bvh3f*, size_t* out);
BVH_AMD_API void bvh3d_copy_prim_ids(const struct bvh3d*, size_t* out);
/* Device-resident prim ids (uint32, BVH order) for device-side permutation of primitives. */
BVH_AMD_API const uint32_t* bvh3f_device_prim_ids(const struct bvh3f*);
BVH_AMD_API const uint32_t* bvh3d_device_prim_ids(const struct bvh3d*);

/* ---- primitive preparation on the device (caller-side loops of test/benchmark.cpp:205-225) ---- */
/* tris9: n x {p0,p1,p2}. Writes Tri::get_bbox / Tri::get_center (tri.h:24-25). */
BVH_AMD_API int bvh_amd_tri_bounds3f(const float* d_tris9, size_t n, float* d_bboxes, float* d_centers, void* stream);
BVH_AMD_API int bvh_amd_tri_bounds3d(const double* d_tris9, size_t n, double* d_bboxes, double* d_centers, void* stream);
/* out[i] = PrecomputedTri(tris[perm ? perm[i] : i]) (tri.h:35-37): n x {p0,e1,e2,n}. */
BVH_AMD_API int bvh_amd_precompute_tris3f(const float* d_tris9, const uint32_t* d_perm, size_t n, float* d_tris12, void* stream);
BVH_AMD_API int bvh_amd_precompute_tris3d(const double* d_tris9, const uint32_t* d_perm, size_t n, double* d_tris12, void* stream);
/* spheres4: n x {center, radius}. Sphere::get_bbox / get_center (sphere.h:24-27). */
BVH_AMD_API int bvh_amd_sphere_bounds3f(const float* d_sph4, size_t n, float* d_bboxes, float* d_centers, void* stream);
BVH_AMD_API int bvh_amd_sphere_bounds3d(const double* d_sph4, size_t n, double* d_bboxes, double* d_centers, void* stream);
/* out[i] = in[perm[i]] for records of `stride` bytes (multiple of 4). */
BVH_AMD_API int bvh_amd_gather(const void* d_in, const uint32_t* d_perm, size_t n, size_t stride, void* d_out, void* stream);

/* The renderer of the reference's benchmark (test/benchmark.cpp:340-371) around the batch traversal: primary rays of its
 * pinhole camera, row-major (y, x): Ray(eye, normalize(dir) + u * right + v * up), u = 2x/w - 1, v = 2y/h - 1 (:343-359);
 * eye/dir/up are HOST float[3]. */
BVH_AMD_API int bvh_amd_pinhole_rays3f(const float eye[3], const float dir[3], const float up[3], size_t width, size_t height,
    struct bvh_ray3f* d_rays, void* stream);
BVH_AMD_API int bvh_amd_pinhole_rays3d(const double eye[3], const double dir[3], const double up[3], size_t width, size_t height,
    struct bvh_ray3d* d_rays, void* stream);
/* ... and its eyelight shading (:363-371): rgb[3i..3i+2] = clamp(int(|dot(normalize(tri.n), ray.dir)| * 256), 0, 255), 0 for a
 * miss. d_tris12 = the BVH-ordered PrecomputedTri array the rays were traced against. */
BVH_AMD_API int bvh_amd_shade_eyelight3f(const float* d_tris12, const struct bvh_ray3f* d_rays, const struct bvh_hit3f* d_hits, size_t n,
    uint8_t* d_rgb, void* stream);
BVH_AMD_API int bvh_amd_shade_eyelight3d(const double* d_tris12, const struct bvh_ray3d* d_rays, const struct bvh_hit3d* d_hits, size_t n,
    uint8_t* d_rgb, void* stream);

/* ---- batched traversal: Bvh::intersect<IsAnyHit, IsRobust> (bvh.h:160-182) for n rays ---------- */
/* d_prims are in BVH order (prims[i] belongs to prim_ids[i]), like the reference's permuted
 * primitives (test/simple_example.cpp:57-65). d_counters may be NULL. Re-entrant like Bvh::intersect on a const Bvh: launches
 * of one BVH may be issued concurrently from several host threads and on several streams (up to 64 in flight per BVH). */
BVH_AMD_API int bvh3f_intersect_rays_tri(const struct bvh3f*, const float* d_tris12, const struct bvh_ray3f* d_rays,
    size_t n, unsigned flags, struct bvh_hit3f* d_hits, struct bvh_amd_counters* d_counters, void* stream);
BVH_AMD_API int bvh3d_intersect_rays_tri(const struct bvh3d*, const double* d_tris12, const struct bvh_ray3d* d_rays,
    size_t n, unsigned flags, struct bvh_hit3d* d_hits, struct bvh_amd_counters* d_counters, void* stream);
BVH_AMD_API int bvh3f_intersect_rays_sphere(const struct bvh3f*, const float* d_sph4, const struct bvh_ray3f* d_rays,
    size_t n, unsigned flags, struct bvh_hit3f* d_hits, struct bvh_amd_counters* d_counters, void* stream);
BVH_AMD_API int bvh3d_intersect_rays_sphere(const struct bvh3d*, const double* d_sph4, const struct bvh_ray3d* d_rays,
    size_t n, unsigned flags, struct bvh_hit3d* d_hits, struct bvh_amd_counters* d_counters, void* stream);

/* Optional, additive: pays NOW what the first large batch through a fresh tree would pay inside its own call — the tree's depth /
 * expected-visits pass (one read-back) and the first allocation of the ray-reordering scratch for batches of `n_rays_hint` rays (kept
 * in the library's block cache for that stream). A single Bvh::intersect on a fresh Bvh is the reference's normal use (bvh.h:160-182);
 * a batch that comes unprepared does the same work itself. */
BVH_AMD_API int bvh3f_prepare_trace(const struct bvh3f*, size_t n_rays_hint, void* stream);
BVH_AMD_API int bvh3d_prepare_trace(const struct bvh3d*, size_t n_rays_hint, void* stream);

/* ---- building blocks exposed for unit tests (order-defining sorts) ----------------------- */
/* d_ids_out[0..n) = the permutation libstdc++ 11's std::sort(iota, [&](i,j){ return keys[i] < keys[j]; }) produces,
 * including the arrangement of equal keys (sweep_sah_builder.h:57-63 depends on it). */
BVH_AMD_API int bvh_amd_std_sort_ids3f(const float* d_keys, size_t n, uint32_t* d_ids_out, void* stream);
BVH_AMD_API int bvh_amd_std_sort_ids3d(const double* d_keys, size_t n, uint32_t* d_ids_out, void* stream);
/* stable LSD radix sort of (key, value) pairs by the low `bits` bits of the key, in place */
BVH_AMD_API int bvh_amd_radix_sort_pairs_u32(uint32_t* d_keys, uint32_t* d_vals, size_t n, int bits, void* stream);

/* Name and average duration source for profiling: the kernel symbol the last intersect call used. */
BVH_AMD_API const char* bvh_amd_last_kernel_name(void);
/* 1 when the calling thread's latest 3D batch traversal reordered its rays internally (BVH_AMD_RAY_SORTED, or the default rule). */
BVH_AMD_API int bvh_amd_last_launch_reordered(void);
/* Measurement aid: with timing on, every batch traversal launched by the calling thread records a pair of events on its stream
 * around the traversal kernel alone (not the optional ray reordering in front of it). bvh_amd_kernel_times waits for and returns
 * the durations (ms) of the latest min(capacity, 256) launches since timing was switched on, oldest first. */
BVH_AMD_API void bvh_amd_kernel_timing(int on);
BVH_AMD_API int bvh_amd_kernel_times(float* ms_out, size_t capacity, size_t* count_out);
/* The same launches: milliseconds from the start of the call on its stream to the start of the traversal kernel = ray keys +
 * radix sort of the optional reordering (0 for a batch traced as given). */
BVH_AMD_API int bvh_amd_reorder_times(float* ms_out, size_t capacity, size_t* count_out);
/* Developer knob for A/B runs inside one process: overrides, for the calling thread's following batch launches, the refill /
 * leaf-parking thresholds of the persistent waves (lanes idle before a wave draws new rays / lanes waiting at a leaf before the
 * leaf code runs), the record fetch of the float 3D kernels (0 per lane, 1 quad-cooperative) and the number of ticket ranges a launch
 * is cut into (1..256; default one per XCD). < 0 restores the default. Results never depend on any of them. */
BVH_AMD_API void bvh_amd_tuning(int refill_threshold, int leaf_threshold, int coop_fetch, int ticket_ranges);
/* Developer experiments of the calling thread, for A/B runs inside one process (tools/r04_experiments.py); results never change.
 * name: "grid_blocks" (cap of the persistent grid), "stream_hints" (1: rays / order / hit records loaded and stored non-temporally),
 * "tri_stride" (floats between PrecomputedTri records of the caller's array: 12, or 16 = padded to a 64-byte line), "key_curve"
 * (0 Morton, 1 Hilbert order of the reordering key), "key_bits" (bits per axis of its grid, 1..8), "step_events" (events the device
 * logs per launch of the per-ray callback walk: a short log makes the tests continue a walk from the device's stack); value < 0 = the default;
 * "one_shot" (1 / 0: force / forbid the one-shot grid of batches of up to 2^18 rays), "stagger" (tickets per eighth of the persistent grid
 * of the staggered drain; 0 = off), "key_class_bits" / "key_class_scale" (chord classes of the reordering key: long rays first);
 * "reset" clears all of them. Returns BVH_AMD_ERR_ARG for an unknown name.                                                       */
BVH_AMD_API int bvh_amd_experiment(const char* name, int value);
/* Developer library only (libbvh_amd_dev.so; the release library returns BVH_AMD_ERR_ARG and its kernels carry no such code): after
 * bvh_amd_experiment("wave_times", 1), every batch launch of the calling thread records, per wavefront of the persistent grid, six
 * 64-bit words {begin, last ticket draw, end} in s_memrealtime ticks (100 MHz), {XCC id << 32 | rays traced}, {ticks spent inside
 * refills (ticket atomic -> order -> ray loads arrived), number of refills}; this copies the latest
 * launch's records out (it waits for the device). The drain-tail study of profiles/r05_tail_timeline_before.txt / _after.txt (tools/tail_timeline.py). */
BVH_AMD_API int bvh_amd_wave_times(unsigned long long* out, size_t capacity_waves, size_t* n_waves);
/* How the calling thread's latest batch launch was traced: out = {0 as given / 1 reordered / 2 reordered with the long rays first,
 * record fetch 0 per lane / 1 quad-cooperative, refill threshold, leaf threshold}. For 3D trees whose traversal records exceed the
 * 32 MB of L2 and batches of >= 2^20 rays the library MEASURES this once per tree and kind of ray (closest / any-hit): the first such
 * batch is traced with the plan a static predictor gives the tree, the following ones with the other candidate plans — {as given,
 * reordered} x {per-lane, cooperative fetch} and, closest-hit, reordered with the long rays first — one WHOLE batch each, timed by
 * events on the launch stream (csrc/traverse.hip: launch_traverse); a candidate that loses clearly is not traced again, survivors are
 * measured twice (3 to 8 batches in all), and the fastest is kept until the tree is re-laid out. Smaller trees and batches use the
 * predictor. BVH_AMD_CALIBRATE=0 in the environment keeps the predictor everywhere. */
BVH_AMD_API void bvh_amd_last_launch_plan(int out[4]);
/* The table of the latest plan search the calling thread FINISHED (the call that settled a tree's plan): nanoseconds per ray of the five
 * candidates {0 as given + per lane, 1 as given + cooperative, 2 reordered + per lane (any-hit: as given + cooperative, heavy thresholds),
 * 3 reordered + cooperative, 4 reordered + cooperative + long rays first (closest-hit only)} — the better of a candidate's measurements,
 * the keys + sort of a reordering candidate charged at their measured rate —, how often each was measured (0: pruned before its turn)
 * and the mask of candidates dropped as clear losers. Measurement aid: bench.py quotes it in `roofline.launch_plan.search`. */
BVH_AMD_API void bvh_amd_last_plan_search(float ns_per_ray[5], int measurements[5], unsigned* dropped_mask);
/* ReinsertionOptimizer iterations run so far in this process: out[0] = through the heap-free fast path, out[1] = through the
 * exact replay of the reference's candidate heap + std::sort (taken when ties make their layout matter).
 * Both produce the reference's result bit for bit; BVH_AMD_REINSERT=exact in the environment forces the replay. */
BVH_AMD_API void bvh_amd_reinsertion_stats(unsigned out[2]);
/* The calling thread's latest ReinsertionOptimizer run (bvhXX_optimize*, or the optimize step of a Quality::High build;
 * reinsertion_optimizer.h:237-267): iterations run, how many of them had to replay the libstdc++ candidate heap of
 * find_candidates exactly (:88-105) instead of the heap-free fast path, the pop_heap + push_heap replacements those replays
 * made, and the GPU time of the heap kernels. */
struct bvh_amd_optimize_profile { unsigned iterations, replayed; unsigned long long replacements; float heap_ms; };
BVH_AMD_API void bvh_amd_last_optimize_profile(struct bvh_amd_optimize_profile* out);


/* ---- one ray, leaves intersected by a HOST callback (c_api/bvh.h:277-295; bvh_impl.h:235-250) --------------------------------
 * Bvh::intersect<IsAnyHit, IsRobust>(ray, root, SmallStack<Index, 64>, leaf_fn): the walk runs on the device and logs the
 * leaves it reaches; the callback is invoked on the calling thread, in the reference's order, with &ray.tmax. Whenever
 * the callback changes tmax the walk restarts from that leaf with the shortened ray, so culling is the reference's.
 * Re-entrant (per-thread stream and buffers). Cost: one or more kernel launches per ray; use the *_intersect_rays_* family for
 * throughput. These return void like the reference's; a device failure aborts with a message instead of reporting "no hit". */
BVH_AMD_API void bvh3f_intersect_ray(const struct bvh3f*, const struct bvh_ray3f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh3d_intersect_ray(const struct bvh3d*, const struct bvh_ray3d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh3f_intersect_ray_any(const struct bvh3f*, const struct bvh_ray3f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh3d_intersect_ray_any(const struct bvh3d*, const struct bvh_ray3d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh3f_intersect_ray_robust(const struct bvh3f*, const struct bvh_ray3f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh3d_intersect_ray_robust(const struct bvh3d*, const struct bvh_ray3d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh3f_intersect_ray_any_robust(const struct bvh3f*, const struct bvh_ray3f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh3d_intersect_ray_any_robust(const struct bvh3d*, const struct bvh_ray3d*, const struct bvh_intersect_callbackd*);
/* Additive: start at any node (`start_index` = the packed index word of Bvh::intersect's `start`, i.e. first_id << 4 |
 * prim_count; BVH_AMD_START_AT_ROOT = the root's), optional inner callback, error code instead of abort. */
#define BVH_AMD_START_AT_ROOT SIZE_MAX
BVH_AMD_API int bvh3f_intersect_ray_visit(const struct bvh3f*, const struct bvh_ray3f*, size_t start_index, unsigned flags,
    const struct bvh_amd_ray_visitorf*);
BVH_AMD_API int bvh3d_intersect_ray_visit(const struct bvh3d*, const struct bvh_ray3d*, size_t start_index, unsigned flags,
    const struct bvh_amd_ray_visitord*);

/* ---- the 2D families `2f` / `2d` (c_api/bvh.cpp:7-10: Bvh<Node<T, 2>>) ---------------------------------------------------
 * Same contracts as the 3D functions above with bvh_bbox2X / bvh_vec2X / bvh_ray2X and 20/40-byte nodes
 * ({minx,maxx,miny,maxy}, index). Builders: serial DefaultBuilder (Low = binned SAH, Medium = sweep SAH, High = sweep +
 * reinsertion), bit-identical with the reference; with a thread pool the reference runs the serial builder below
 * parallel_threshold (default_builder.h:38-39) and so does this library, while AT OR ABOVE the threshold its mini-tree
 * builder reads the third component of 2D points (mini_tree_builder.h:183: undefined behaviour), which is refused here
 * (NULL + bvh_amd_last_error()). Leaf primitive of the batch traversal: circles = Sphere<T, 2> {center.x, center.y, radius}
 * (sphere.h:15-50; tri.h has no 2D intersector). */
BVH_AMD_API struct bvh2f* bvh2f_build(struct bvh_thread_pool*, const struct bvh_bbox2f* bboxes,
    const struct bvh_vec2f* centers, size_t prim_count, const struct bvh_build_config* config);   /* c_api/bvh.h:99-125 */
BVH_AMD_API struct bvh2f* bvh2f_build_device(const float* d_bboxes4, const float* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, void* stream);
BVH_AMD_API struct bvh2f* bvh2f_build_sah(struct bvh_thread_pool*, const struct bvh_bbox2f* bboxes, const struct bvh_vec2f* centers,
    size_t prim_count, const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah);
BVH_AMD_API struct bvh2f* bvh2f_build_device_sah(const float* d_bboxes4, const float* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, const struct bvh_amd_sah_config* sah, void* stream);
BVH_AMD_API struct bvh2f* bvh2f_build_device_binned(const float* d_bboxes4, const float* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah, size_t bin_count, void* stream);
BVH_AMD_API struct bvh2f* bvh2f_from_nodes(const void* nodes, size_t node_count, const size_t* prim_ids, size_t prim_count);
BVH_AMD_API struct bvh2f* bvh2f_extract(struct bvh2f* bvh, size_t root_id);
BVH_AMD_API void bvh2f_destroy(struct bvh2f*);
BVH_AMD_API void bvh2f_optimize(struct bvh_thread_pool*, struct bvh2f*);
BVH_AMD_API int bvh2f_optimize_config(struct bvh2f*, const struct bvh_amd_optimize_config*);
BVH_AMD_API void bvh2f_refit(struct bvh2f*);
BVH_AMD_API int bvh2f_refit_status(struct bvh2f*);
BVH_AMD_API int bvh2f_sync_device(struct bvh2f*);
BVH_AMD_API void bvh2f_append_node(struct bvh2f*);
BVH_AMD_API void bvh2f_remove_last_node(struct bvh2f*);
BVH_AMD_API void bvh_node2f_set_prim_count(struct bvh_node2f*, size_t);
BVH_AMD_API void bvh_node2f_set_first_id(struct bvh_node2f*, size_t);
BVH_AMD_API void bvh_node2f_set_bbox(struct bvh_node2f*, const struct bvh_bbox2f*);
BVH_AMD_API void bvh2f_save(const struct bvh2f*, FILE*);
BVH_AMD_API struct bvh2f* bvh2f_load(FILE*);
BVH_AMD_API size_t bvh2f_serialize(const struct bvh2f*, void* out, size_t capacity);
BVH_AMD_API struct bvh2f* bvh2f_deserialize(const void* bytes, size_t size);
BVH_AMD_API size_t bvh2f_serialize_device(struct bvh2f*, void* d_out, size_t cap, void* stream);
BVH_AMD_API struct bvh2f* bvh2f_deserialize_device(const void* d_bytes, size_t size, void* stream);
BVH_AMD_API struct bvh_node2f* bvh2f_get_node(struct bvh2f*, size_t);
BVH_AMD_API size_t bvh2f_get_prim_id(const struct bvh2f*, size_t);
BVH_AMD_API size_t bvh2f_get_prim_count(const struct bvh2f*);
BVH_AMD_API size_t bvh2f_get_node_count(const struct bvh2f*);
BVH_AMD_API bool bvh_node2f_is_leaf(const struct bvh_node2f*);
BVH_AMD_API size_t bvh_node2f_get_prim_count(const struct bvh_node2f*);
BVH_AMD_API size_t bvh_node2f_get_first_id(const struct bvh_node2f*);
BVH_AMD_API struct bvh_bbox2f bvh_node2f_get_bbox(const struct bvh_node2f*);
BVH_AMD_API void bvh2f_copy_nodes(const struct bvh2f*, void* out);
BVH_AMD_API void bvh2f_copy_prim_ids(const struct bvh2f*, size_t* out);
BVH_AMD_API const uint32_t* bvh2f_device_prim_ids(const struct bvh2f*);
/* circles3: n x {center.x, center.y, radius} -> bboxes n x {min.x,min.y,max.x,max.y}, centers n x 2 (sphere.h:24-27) */
BVH_AMD_API int bvh_amd_sphere_bounds2f(const float* d_circles3, size_t n, float* d_bboxes4, float* d_centers2, void* stream);
/* d_circles3 in BVH order; hit = {prim, t0, t1, 0} like the 3D sphere traversal */
BVH_AMD_API int bvh2f_intersect_rays_sphere(const struct bvh2f* bvh, const float* d_circles3, const struct bvh_ray2f* d_rays,
    size_t n, unsigned flags, struct bvh_hit3f* d_hits, struct bvh_amd_counters* d_counters, void* stream);

BVH_AMD_API struct bvh2d* bvh2d_build(struct bvh_thread_pool*, const struct bvh_bbox2d* bboxes,
    const struct bvh_vec2d* centers, size_t prim_count, const struct bvh_build_config* config);   /* c_api/bvh.h:99-125 */
BVH_AMD_API struct bvh2d* bvh2d_build_device(const double* d_bboxes4, const double* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, void* stream);
BVH_AMD_API struct bvh2d* bvh2d_build_sah(struct bvh_thread_pool*, const struct bvh_bbox2d* bboxes, const struct bvh_vec2d* centers,
    size_t prim_count, const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah);
BVH_AMD_API struct bvh2d* bvh2d_build_device_sah(const double* d_bboxes4, const double* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, enum bvh_amd_builder builder, const struct bvh_amd_sah_config* sah, void* stream);
BVH_AMD_API struct bvh2d* bvh2d_build_device_binned(const double* d_bboxes4, const double* d_centers2, size_t prim_count,
    const struct bvh_build_config* config, const struct bvh_amd_sah_config* sah, size_t bin_count, void* stream);
BVH_AMD_API struct bvh2d* bvh2d_from_nodes(const void* nodes, size_t node_count, const size_t* prim_ids, size_t prim_count);
BVH_AMD_API struct bvh2d* bvh2d_extract(struct bvh2d* bvh, size_t root_id);
BVH_AMD_API void bvh2d_destroy(struct bvh2d*);
BVH_AMD_API void bvh2d_optimize(struct bvh_thread_pool*, struct bvh2d*);
BVH_AMD_API int bvh2d_optimize_config(struct bvh2d*, const struct bvh_amd_optimize_config*);
BVH_AMD_API void bvh2d_refit(struct bvh2d*);
BVH_AMD_API int bvh2d_refit_status(struct bvh2d*);
BVH_AMD_API int bvh2d_sync_device(struct bvh2d*);
BVH_AMD_API void bvh2d_append_node(struct bvh2d*);
BVH_AMD_API void bvh2d_remove_last_node(struct bvh2d*);
BVH_AMD_API void bvh_node2d_set_prim_count(struct bvh_node2d*, size_t);
BVH_AMD_API void bvh_node2d_set_first_id(struct bvh_node2d*, size_t);
BVH_AMD_API void bvh_node2d_set_bbox(struct bvh_node2d*, const struct bvh_bbox2d*);
BVH_AMD_API void bvh2d_save(const struct bvh2d*, FILE*);
BVH_AMD_API struct bvh2d* bvh2d_load(FILE*);
BVH_AMD_API size_t bvh2d_serialize(const struct bvh2d*, void* out, size_t capacity);
BVH_AMD_API struct bvh2d* bvh2d_deserialize(const void* bytes, size_t size);
BVH_AMD_API size_t bvh2d_serialize_device(struct bvh2d*, void* d_out, size_t cap, void* stream);
BVH_AMD_API struct bvh2d* bvh2d_deserialize_device(const void* d_bytes, size_t size, void* stream);
BVH_AMD_API struct bvh_node2d* bvh2d_get_node(struct bvh2d*, size_t);
BVH_AMD_API size_t bvh2d_get_prim_id(const struct bvh2d*, size_t);
BVH_AMD_API size_t bvh2d_get_prim_count(const struct bvh2d*);
BVH_AMD_API size_t bvh2d_get_node_count(const struct bvh2d*);
BVH_AMD_API bool bvh_node2d_is_leaf(const struct bvh_node2d*);
BVH_AMD_API size_t bvh_node2d_get_prim_count(const struct bvh_node2d*);
BVH_AMD_API size_t bvh_node2d_get_first_id(const struct bvh_node2d*);
BVH_AMD_API struct bvh_bbox2d bvh_node2d_get_bbox(const struct bvh_node2d*);
BVH_AMD_API void bvh2d_copy_nodes(const struct bvh2d*, void* out);
BVH_AMD_API void bvh2d_copy_prim_ids(const struct bvh2d*, size_t* out);
BVH_AMD_API const uint32_t* bvh2d_device_prim_ids(const struct bvh2d*);
/* circles3: n x {center.x, center.y, radius} -> bboxes n x {min.x,min.y,max.x,max.y}, centers n x 2 (sphere.h:24-27) */
BVH_AMD_API int bvh_amd_sphere_bounds2d(const double* d_circles3, size_t n, double* d_bboxes4, double* d_centers2, void* stream);
/* d_circles3 in BVH order; hit = {prim, t0, t1, 0} like the 3D sphere traversal */
BVH_AMD_API int bvh2d_intersect_rays_sphere(const struct bvh2d* bvh, const double* d_circles3, const struct bvh_ray2d* d_rays,
    size_t n, unsigned flags, struct bvh_hit3d* d_hits, struct bvh_amd_counters* d_counters, void* stream);

/* c_api/bvh.h:277-295, 2D */
BVH_AMD_API void bvh2f_intersect_ray(const struct bvh2f*, const struct bvh_ray2f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh2d_intersect_ray(const struct bvh2d*, const struct bvh_ray2d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh2f_intersect_ray_any(const struct bvh2f*, const struct bvh_ray2f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh2d_intersect_ray_any(const struct bvh2d*, const struct bvh_ray2d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh2f_intersect_ray_robust(const struct bvh2f*, const struct bvh_ray2f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh2d_intersect_ray_robust(const struct bvh2d*, const struct bvh_ray2d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API void bvh2f_intersect_ray_any_robust(const struct bvh2f*, const struct bvh_ray2f*, const struct bvh_intersect_callbackf*);
BVH_AMD_API void bvh2d_intersect_ray_any_robust(const struct bvh2d*, const struct bvh_ray2d*, const struct bvh_intersect_callbackd*);
BVH_AMD_API int bvh2f_intersect_ray_visit(const struct bvh2f*, const struct bvh_ray2f*, size_t start_index, unsigned flags,
    const struct bvh_amd_ray_visitorf*);
BVH_AMD_API int bvh2d_intersect_ray_visit(const struct bvh2d*, const struct bvh_ray2d*, size_t start_index, unsigned flags,
    const struct bvh_amd_ray_visitord*);
/* multi-GPU for the 2D families: see bvh3f_broadcast / bvh3f_replicate */
BVH_AMD_API struct bvh2f* bvh2f_broadcast(struct bvh_amd_comm*, int root, struct bvh2f* bvh, const void* d_prims, size_t prim_bytes,
                                          void** d_prims_out, size_t* prim_bytes_out, void* stream);
BVH_AMD_API struct bvh2d* bvh2d_broadcast(struct bvh_amd_comm*, int root, struct bvh2d* bvh, const void* d_prims, size_t prim_bytes,
                                          void** d_prims_out, size_t* prim_bytes_out, void* stream);
BVH_AMD_API int bvh2f_replicate(struct bvh2f* bvh, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices,
                                struct bvh2f** bvhs_out, void** d_prims_out);
BVH_AMD_API int bvh2d_replicate(struct bvh2d* bvh, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices,
                                struct bvh2d** bvhs_out, void** d_prims_out);

#ifdef __cplusplus
}
#endif

#endif
