// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// C-ABI harness around the *unmodified* reference library (madmann91/bvh v2), which is header-only
// C++20. This translation unit contains no reference code: it only `#include`s the headers where they
// lie under /root/reference/src (never copied into this repository) and forwards to them. It is built
// by oracle/Makefile into oracle/_ref/libbvh_ref.so (git-ignored, shipped to the GPU box by gpurun)
// and is used
//   * to validate the restatement in oracle/bvh_oracle.cpp (tests/test_oracle_vs_ref.py),
//   * to generate the committed golden fixtures (tests/golden/make_golden.py),
//   * as bench.py's `cpu_baseline` leg of kind "reference".
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Pinned build flags (they are part of the observable behaviour):
//   g++ 11 -std=c++20 -O3 -mavx2 -mfma -ffp-contract=off -DNDEBUG
// -mfma makes <cmath> define FP_FAST_FMAF so that fast_mul_add (utils.h:74-81) is a real fma, and
// -ffp-contract=off forbids any other fusion, i.e. "one rounding per operation, fma exactly at the
// fast_mul_add call sites". (-mavx2 -mfma instead of -march=native so that the .so also runs on the
// GPU box's host CPU; vectorisation cannot change IEEE results without -ffast-math.)

#include <bvh/v2/bvh.h>
#include <bvh/v2/vec.h>
#include <bvh/v2/ray.h>
#include <bvh/v2/node.h>
#include <bvh/v2/default_builder.h>
#include <bvh/v2/binned_sah_builder.h>
#include <bvh/v2/sweep_sah_builder.h>
#include <bvh/v2/mini_tree_builder.h>
#include <bvh/v2/reinsertion_optimizer.h>
#include <bvh/v2/thread_pool.h>
#include <bvh/v2/executor.h>
#include <bvh/v2/stack.h>
#include <bvh/v2/tri.h>
#include <bvh/v2/sphere.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <chrono>
#include <thread>
#include <vector>

#include "oracle_abi.h"

namespace {

using namespace bvh::v2;

template <typename T> using Node3 = Node<T, 3>;
template <typename T> using Bvh3 = Bvh<Node3<T>>;
// the SplitHeuristic (split_heuristic.h:17-23) the following builds use; ref_set_sah() changes it (default {0, 1})
static size_t g_sah_log_cluster = 0;
static double g_sah_cost_ratio = 1.;
// BinnedSahBuilder<Node, BinCount>'s template argument for the ORC_BUILDER_BINNED builds that follow (ref_set_bin_count; default 8)
static size_t g_bin_count = 8;

template <typename T, size_t D> using BvhN = Bvh<Node<T, D>>;   // D = 2: the `2f` / `2d` families of the C API (c_api/bvh.cpp:7-10)

template <typename T, size_t D>
BvhN<T, D>* build(const T* bboxes, const T* centers, size_t n, int builder, int quality,
                  size_t min_leaf, size_t max_leaf, size_t par_threshold, int threads)
{
    using N = Node<T, D>;
    std::vector<BBox<T, D>> bb(n);                            // bboxes: n x {min[D], max[D]}, centers: n x D
    std::vector<Vec<T, D>> cc(n);
    for (size_t i = 0; i < n; ++i) {
        for (size_t k = 0; k < D; ++k) {
            bb[i].min[k] = bboxes[2 * D * i + k];
            bb[i].max[k] = bboxes[2 * D * i + D + k];
            cc[i][k] = centers[D * i + k];
        }
    }
    typename DefaultBuilder<N>::Config cfg;
    cfg.sah = SplitHeuristic<T>(g_sah_log_cluster, static_cast<T>(g_sah_cost_ratio));
    cfg.quality = static_cast<typename DefaultBuilder<N>::Quality>(quality);
    cfg.min_leaf_size = min_leaf;
    cfg.max_leaf_size = max_leaf;
    cfg.parallel_threshold = par_threshold;
    auto out = std::make_unique<BvhN<T, D>>();
    // (with D = 2 the mini-tree builder reads p[2] of a 2D vector, mini_tree_builder.h:183: undefined behaviour. The harness
    //  only lets DefaultBuilder(pool) through below parallel_threshold, where it is the serial builder, default_builder.h:38-39)
    if (D == 2 && builder == ORC_BUILDER_DEFAULT_PARALLEL && n >= par_threshold) return nullptr;
    switch (builder) {
    case ORC_BUILDER_DEFAULT_SERIAL:
        *out = DefaultBuilder<N>::build(bb, cc, cfg);
        break;
    case ORC_BUILDER_DEFAULT_PARALLEL: {
        ThreadPool pool(static_cast<size_t>(threads));
        *out = DefaultBuilder<N>::build(pool, bb, cc, cfg);
        break;
    }
    case ORC_BUILDER_BINNED:
        switch (g_bin_count) {                                // binned_sah_builder.h:18: BinCount is a template argument
        case 4:  *out = BinnedSahBuilder<N, 4>::build(bb, cc, cfg); break;
        case 8:  *out = BinnedSahBuilder<N>::build(bb, cc, cfg); break;
        case 16: *out = BinnedSahBuilder<N, 16>::build(bb, cc, cfg); break;
        case 32: *out = BinnedSahBuilder<N, 32>::build(bb, cc, cfg); break;
        default: return nullptr;
        }
        break;
    case ORC_BUILDER_SWEEP:
        *out = SweepSahBuilder<N>::build(bb, cc, cfg);
        break;
    default:
        return nullptr;
    }
    return out.release();
}

template <typename T, size_t D>
void get_nodes(const BvhN<T, D>* b, void* out) {
    static_assert(sizeof(Node3<float>) == 28 && sizeof(Node3<double>) == 56 && sizeof(Node<float, 2>) == 20 && sizeof(Node<double, 2>) == 40);
    std::memcpy(out, b->nodes.data(), b->nodes.size() * sizeof(Node<T, D>));
}

template <typename T, size_t D>
BvhN<T, D>* from_arrays(const void* nodes, size_t nn, const uint64_t* prim_ids, size_t np) {
    auto b = std::make_unique<BvhN<T, D>>();
    b->nodes.resize(nn);
    std::memcpy(b->nodes.data(), nodes, nn * sizeof(Node<T, D>));
    b->prim_ids.assign(prim_ids, prim_ids + np);
    return b.release();
}

struct VecStream : OutputStream {
    std::vector<uint8_t> bytes;
    bool write_raw(const void* p, size_t n) override {
        auto q = static_cast<const uint8_t*>(p);
        bytes.insert(bytes.end(), q, q + n);
        return true;
    }
};

template <typename T, size_t D>
size_t serialize(const BvhN<T, D>* b, uint8_t* out, size_t cap) {
    VecStream s;
    b->serialize(s);
    if (out && cap >= s.bytes.size())
        std::memcpy(out, s.bytes.data(), s.bytes.size());
    return s.bytes.size();
}

template <typename T, size_t D>
void optimize(BvhN<T, D>* b, int threads) {
    if (threads < 0) {
        ReinsertionOptimizer<Node<T, D>>::optimize(*b);
    } else {
        ThreadPool pool(static_cast<size_t>(threads));
        ReinsertionOptimizer<Node<T, D>>::optimize(pool, *b);
    }
}

template <typename T, size_t D>
void optimize_config(BvhN<T, D>* b, int threads, double ratio, size_t iterations) {
    typename ReinsertionOptimizer<Node<T, D>>::Config config;
    config.batch_size_ratio = static_cast<T>(ratio);
    config.max_iter_count = iterations;
    if (threads < 0) {
        ReinsertionOptimizer<Node<T, D>>::optimize(*b, config);
    } else {
        ThreadPool pool(static_cast<size_t>(threads));
        ReinsertionOptimizer<Node<T, D>>::optimize(pool, *b, config);
    }
}

template <typename T>
void prep_tris(const T* t9, size_t n, T* bboxes, T* centers) {
    for (size_t i = 0; i < n; ++i) {
        Tri<T, 3> tri(Vec<T, 3>(t9[9 * i + 0], t9[9 * i + 1], t9[9 * i + 2]),
                      Vec<T, 3>(t9[9 * i + 3], t9[9 * i + 4], t9[9 * i + 5]),
                      Vec<T, 3>(t9[9 * i + 6], t9[9 * i + 7], t9[9 * i + 8]));
        auto bb = tri.get_bbox();
        auto c = tri.get_center();
        for (int k = 0; k < 3; ++k) {
            bboxes[6 * i + k] = bb.min[k];
            bboxes[6 * i + 3 + k] = bb.max[k];
            centers[3 * i + k] = c[k];
        }
    }
}

template <typename T>
void precompute_tris(const T* t9, const uint64_t* perm, size_t n, T* out12) {
    static_assert(sizeof(PrecomputedTri<T>) == 12 * sizeof(T));
    for (size_t i = 0; i < n; ++i) {
        size_t j = perm ? perm[i] : i;
        PrecomputedTri<T> p(Vec<T, 3>(t9[9 * j + 0], t9[9 * j + 1], t9[9 * j + 2]),
                            Vec<T, 3>(t9[9 * j + 3], t9[9 * j + 4], t9[9 * j + 5]),
                            Vec<T, 3>(t9[9 * j + 6], t9[9 * j + 7], t9[9 * j + 8]));
        std::memcpy(out12 + 12 * i, &p, sizeof(p));
    }
}

template <typename T, size_t D>
void sphere_bboxes(const T* sph, size_t n, T* bboxes, T* centers) {      // sph: n x {center[D], radius}
    for (size_t i = 0; i < n; ++i) {
        Vec<T, D> ctr;
        for (size_t k = 0; k < D; ++k) ctr[k] = sph[(D + 1) * i + k];
        Sphere<T, D> s(ctr, sph[(D + 1) * i + D]);
        auto bb = s.get_bbox();
        auto c = s.get_center();
        for (size_t k = 0; k < D; ++k) {
            bboxes[2 * D * i + k] = bb.min[k];
            bboxes[2 * D * i + D + k] = bb.max[k];
            centers[D * i + k] = c[k];
        }
    }
}

template <typename T> struct HitOf;
template <> struct HitOf<float>  { using Type = orc_hitf; };
template <> struct HitOf<double> { using Type = orc_hitd; };

// Runs `fn(begin, end, thread_slot)` over [0, n) on `threads` std::threads (threads <= 1: inline).
template <typename F>
void parallel_chunks(size_t n, int threads, F&& fn) {
    if (threads <= 1 || n < 1024) { fn(size_t{0}, n, 0); return; }
    std::vector<std::thread> pool;
    size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t b = std::min(n, chunk * t), e = std::min(n, b + chunk);
        pool.emplace_back([=, &fn] { fn(b, e, t); });
    }
    for (auto& t : pool) t.join();
}

// The leaf lambda below is the closest-hit / any-hit pattern of test/benchmark.cpp:277-298 and
// test/simple_example.cpp:81-92 (permuted primitives: the BVH-order index i addresses prims[i]).
// rays [rb, re) through `b`, one after the other (what one worker does)
template <typename T, size_t D, bool Any, bool Robust, typename Prim, typename LeafTest>
void trace_range(const BvhN<T, D>* b, const Prim* prims, const T* rays8, size_t rb, size_t re, typename HitOf<T>::Type* out,
                 LeafTest& leaf_test, uint64_t& pairs, uint64_t& tests, uint64_t& leaves)
{
    for (size_t r = rb; r < re; ++r) {
        const T* q = rays8 + (2 * D + 2) * r;                 // {org[D], dir[D], tmin, tmax}
        Vec<T, D> org, dir;
        for (size_t k = 0; k < D; ++k) { org[k] = q[k]; dir[k] = q[D + k]; }
        Ray<T, D> ray(org, dir, q[2 * D], q[2 * D + 1]);
        typename HitOf<T>::Type h{};
        h.prim = ORC_INVALID; h.t = q[2 * D + 1]; h.u = 0; h.v = 0;
        SmallStack<typename BvhN<T, D>::Index, 64> stack;
        b->template intersect<Any, Robust>(ray, b->get_root().index, stack,
            [&](size_t begin, size_t end) {
                ++leaves;
                for (size_t i = begin; i < end; ++i) {
                    ++tests;
                    leaf_test(prims[i], ray, i, h);
                }
                return h.prim != ORC_INVALID;
            },
            [&](const Node<T, D>&, const Node<T, D>&) { ++pairs; });
        out[r] = h;
    }
}

template <typename T, size_t D, bool Any, bool Robust, typename Prim, typename LeafTest>
void intersect_all(const BvhN<T, D>* b, const Prim* prims, const T* rays8, size_t nrays, int threads,
                   typename HitOf<T>::Type* out, uint64_t* counters, LeafTest&& leaf_test)
{
    std::vector<uint64_t> cnt(3 * std::max(threads, 1), 0);
    parallel_chunks(nrays, threads, [&](size_t rb, size_t re, int slot) {
        uint64_t pairs = 0, tests = 0, leaves = 0;
        trace_range<T, D, Any, Robust>(b, prims, rays8, rb, re, out, leaf_test, pairs, tests, leaves);
        cnt[3 * slot + 0] += pairs; cnt[3 * slot + 1] += tests; cnt[3 * slot + 2] += leaves;
    });
    if (counters) {
        counters[0] = counters[1] = counters[2] = 0;
        for (size_t i = 0; i < cnt.size(); i += 3) {
            counters[0] += cnt[i]; counters[1] += cnt[i + 1]; counters[2] += cnt[i + 2];
        }
    }
}

// The CPU baseline of bench.py: the ray array split by the reference's own ParallelExecutor::for_each
// (executor.h:51-61: one chunk per worker of a persistent ThreadPool), each worker running the benchmark.cpp:277-298 loop;
// `reps` timed passes after one untimed pass, seconds per pass in `seconds`.
template <typename T, bool Any, bool Robust>
void bench_tri(const Bvh3<T>* b, const T* tris12, const T* rays8, size_t nrays, int threads, int reps,
               typename HitOf<T>::Type* out, double* seconds)
{
    auto prims = reinterpret_cast<const PrecomputedTri<T>*>(tris12);
    auto leaf_test = [](const PrecomputedTri<T>& tri, Ray<T, 3>& ray, size_t i, typename HitOf<T>::Type& h) {
        if (auto hit = tri.intersect(ray)) {
            std::tie(ray.tmax, h.u, h.v) = *hit;
            h.t = ray.tmax;
            h.prim = static_cast<decltype(h.prim)>(i);
        }
    };
    ThreadPool pool(static_cast<size_t>(std::max(threads, 0)));
    ParallelExecutor executor(pool);
    for (int rep = -1; rep < reps; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        executor.for_each(0, nrays, [&](size_t rb, size_t re) {
            uint64_t pairs = 0, tests = 0, leaves = 0;
            trace_range<T, 3, Any, Robust>(b, prims, rays8, rb, re, out, leaf_test, pairs, tests, leaves);
        });
        if (rep >= 0) seconds[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
}

template <typename T, bool Any, bool Robust>
void intersect_tri(const Bvh3<T>* b, const T* tris12, const T* rays8, size_t nrays, int threads,
                   typename HitOf<T>::Type* out, uint64_t* counters)
{
    auto prims = reinterpret_cast<const PrecomputedTri<T>*>(tris12);
    intersect_all<T, 3, Any, Robust>(b, prims, rays8, nrays, threads, out, counters,
        [](const PrecomputedTri<T>& tri, Ray<T, 3>& ray, size_t i, typename HitOf<T>::Type& h) {
            if (auto hit = tri.intersect(ray)) {
                std::tie(ray.tmax, h.u, h.v) = *hit;
                h.t = ray.tmax;
                h.prim = static_cast<decltype(h.prim)>(i);
            }
        });
}

template <typename T, size_t D, bool Any, bool Robust>
void intersect_sphere(const BvhN<T, D>* b, const T* sph, const T* rays, size_t nrays, int threads,
                      typename HitOf<T>::Type* out, uint64_t* counters)
{
    auto prims = reinterpret_cast<const Sphere<T, D>*>(sph);
    static_assert(sizeof(Sphere<T, D>) == (D + 1) * sizeof(T));
    intersect_all<T, D, Any, Robust>(b, prims, rays, nrays, threads, out, counters,
        [](const Sphere<T, D>& s, Ray<T, D>& ray, size_t i, typename HitOf<T>::Type& h) {
            if (auto hit = s.intersect(ray)) {
                ray.tmax = hit->first;
                h.t = hit->first;
                h.u = hit->second;
                h.v = 0;
                h.prim = static_cast<decltype(h.prim)>(i);
            }
        });
}

template <typename T, template <typename, bool, bool> class Fn>
struct Dispatch;

#define DISPATCH4(fn, T, any, robust, ...)                                  \
    do {                                                                    \
        if (any) { if (robust) fn<T, true, true>(__VA_ARGS__); else fn<T, true, false>(__VA_ARGS__); } \
        else     { if (robust) fn<T, false, true>(__VA_ARGS__); else fn<T, false, false>(__VA_ARGS__); } \
    } while (0)

/* Node<float, 3, IndexBits, PrimCountBits> with Index parameters other than the defaults (node.h:21-22, index.h:32-41): the serialized
 * stream of one build, for the golden fixture that pins how the C++ mirror re-packs its nodes (tests/golden/make_golden.py).
 * variant 0: Node<float, 3, 32, 2>, SweepSahBuilder, max_leaf_size = max_leaf;  1: Node<float, 3, 64, 6>, DefaultBuilder serial High;
 * 2: variant 1's extract_bvh(root.first_id). Returns the stream's size (written when it fits). */
template <typename N, typename Build>
size_t index_variant_stream(const float* bboxes, const float* centers, size_t n, Build&& build_fn, uint8_t* out, size_t cap) {
    std::vector<BBox<float, 3>> bb(n);
    std::vector<Vec<float, 3>> cc(n);
    for (size_t i = 0; i < n; ++i) for (size_t k = 0; k < 3; ++k) {
        bb[i].min[k] = bboxes[6 * i + k]; bb[i].max[k] = bboxes[6 * i + 3 + k]; cc[i][k] = centers[3 * i + k]; }
    Bvh<N> bvh = build_fn(bb, cc);
    VecStream s;
    bvh.serialize(s);
    if (out && cap >= s.bytes.size()) std::memcpy(out, s.bytes.data(), s.bytes.size());
    return s.bytes.size();
}

} // namespace

extern "C" {

#define DISPATCH_SPHERE(T, D, any, robust, ...)                             \
    do {                                                                    \
        if (any) { if (robust) intersect_sphere<T, D, true, true>(__VA_ARGS__); else intersect_sphere<T, D, true, false>(__VA_ARGS__); } \
        else     { if (robust) intersect_sphere<T, D, false, true>(__VA_ARGS__); else intersect_sphere<T, D, false, false>(__VA_ARGS__); } \
    } while (0)

/* everything that exists for every dimension (D = 2: bboxes n x 4, centers n x 2, nodes 20/40 bytes, spheres n x 3, rays n x 6) */
ORC_EXPORT size_t ref_index_variant_stream(const float* bboxes, const float* centers, size_t n, int variant, size_t max_leaf, uint8_t* out, size_t cap) {
    if (variant == 0) {
        using N = Node<float, 3, 32, 2>;
        return index_variant_stream<N>(bboxes, centers, n, [&](auto& bb, auto& cc) {
            typename SweepSahBuilder<N>::Config cfg; cfg.max_leaf_size = max_leaf; return SweepSahBuilder<N>::build(bb, cc, cfg); }, out, cap);
    }
    using N = Node<float, 3, 64, 6>;
    return index_variant_stream<N>(bboxes, centers, n, [&](auto& bb, auto& cc) {
        typename DefaultBuilder<N>::Config cfg; cfg.quality = DefaultBuilder<N>::Quality::High; cfg.max_leaf_size = max_leaf;
        auto bvh = DefaultBuilder<N>::build(bb, cc, cfg);
        return variant == 2 ? bvh.extract_bvh(bvh.get_root().index.first_id()) : std::move(bvh); }, out, cap);
}

ORC_EXPORT int ref_set_bin_count(size_t bin_count) {
    if (bin_count != 4 && bin_count != 8 && bin_count != 16 && bin_count != 32) return -1;   // the instantiations above
    g_bin_count = bin_count;
    return 0;
}
ORC_EXPORT void ref_set_sah(size_t log_cluster_size, double cost_ratio) { g_sah_log_cluster = log_cluster_size; g_sah_cost_ratio = cost_ratio; }

#define REF_IMPL(T, D, S)                                                                               \
    ORC_EXPORT void* ref_build##S(const T* bboxes, const T* centers, size_t n, int builder,            \
        int quality, size_t min_leaf, size_t max_leaf, size_t par_threshold, int threads) {            \
        return build<T, D>(bboxes, centers, n, builder, quality, min_leaf, max_leaf, par_threshold, threads); } \
    ORC_EXPORT void ref_destroy##S(void* h) { delete static_cast<BvhN<T, D>*>(h); }                     \
    ORC_EXPORT size_t ref_node_count##S(const void* h) { return static_cast<const BvhN<T, D>*>(h)->nodes.size(); } \
    ORC_EXPORT size_t ref_prim_count##S(const void* h) { return static_cast<const BvhN<T, D>*>(h)->prim_ids.size(); } \
    ORC_EXPORT void ref_get_nodes##S(const void* h, void* out) { get_nodes<T, D>(static_cast<const BvhN<T, D>*>(h), out); } \
    ORC_EXPORT void ref_get_prim_ids##S(const void* h, uint64_t* out) {                                \
        auto b = static_cast<const BvhN<T, D>*>(h);                                                     \
        for (size_t i = 0; i < b->prim_ids.size(); ++i) out[i] = b->prim_ids[i]; }                      \
    ORC_EXPORT void* ref_from_arrays##S(const void* nodes, size_t nn, const uint64_t* ids, size_t np) { \
        return from_arrays<T, D>(nodes, nn, ids, np); }                                                 \
    ORC_EXPORT size_t ref_serialize##S(const void* h, uint8_t* out, size_t cap) {                       \
        return serialize<T, D>(static_cast<const BvhN<T, D>*>(h), out, cap); }                          \
    ORC_EXPORT void ref_optimize##S(void* h, int threads) { optimize<T, D>(static_cast<BvhN<T, D>*>(h), threads); } \
    ORC_EXPORT void ref_optimize_config##S(void* h, int threads, double ratio, size_t iterations) {       \
        optimize_config<T, D>(static_cast<BvhN<T, D>*>(h), threads, ratio, iterations); }                  \
    ORC_EXPORT void* ref_extract##S(const void* h, size_t root) {                                       \
        return new BvhN<T, D>(static_cast<const BvhN<T, D>*>(h)->extract_bvh(root)); }                  \
    ORC_EXPORT void ref_refit##S(void* h) { static_cast<BvhN<T, D>*>(h)->refit(); }                     \
    ORC_EXPORT void ref_sphere_bboxes##S(const T* sph, size_t n, T* bb, T* cc) { sphere_bboxes<T, D>(sph, n, bb, cc); } \
    ORC_EXPORT void ref_intersect_sphere##S(const void* h, const T* sph, const T* rays, size_t nrays,   \
        int any, int robust, int threads, HitOf<T>::Type* out, uint64_t* counters) {                    \
        DISPATCH_SPHERE(T, D, any, robust, static_cast<const BvhN<T, D>*>(h), sph, rays, nrays,         \
                        threads, out, counters); }

/* triangles exist in 3D only (tri.h:30-74); MiniTreeBuilder::build itself too (its grid reads three components) */
#define REF_IMPL_TRI(T, S)                                                                              \
    ORC_EXPORT void* ref_build_minitree##S(const T* bboxes, const T* centers, size_t n, size_t min_leaf, size_t max_leaf, \
        int enable_pruning, double pruning_area_ratio, size_t par_threshold, int threads, size_t log2_grid_dim) { \
        std::vector<BBox<T, 3>> bb(n); std::vector<Vec<T, 3>> cc(n);                                    \
        for (size_t i = 0; i < n; ++i) for (size_t k = 0; k < 3; ++k) {                                  \
            bb[i].min[k] = bboxes[6 * i + k]; bb[i].max[k] = bboxes[6 * i + 3 + k]; cc[i][k] = centers[3 * i + k]; } \
        typename MiniTreeBuilder<Node3<T>>::Config cfg;                                                 \
        cfg.sah = SplitHeuristic<T>(g_sah_log_cluster, static_cast<T>(g_sah_cost_ratio));               \
        cfg.min_leaf_size = min_leaf; cfg.max_leaf_size = max_leaf; cfg.enable_pruning = enable_pruning != 0; \
        cfg.pruning_area_ratio = static_cast<T>(pruning_area_ratio); cfg.parallel_threshold = par_threshold; \
        cfg.log2_grid_dim = log2_grid_dim;                                                              \
        ThreadPool pool(static_cast<size_t>(threads));                                                  \
        return new Bvh3<T>(MiniTreeBuilder<Node3<T>>::build(pool, bb, cc, cfg)); }                      \
    ORC_EXPORT void ref_prep_tris##S(const T* t9, size_t n, T* bb, T* cc) { prep_tris<T>(t9, n, bb, cc); } \
    ORC_EXPORT void ref_precompute_tris##S(const T* t9, const uint64_t* perm, size_t n, T* out12) {     \
        precompute_tris<T>(t9, perm, n, out12); }                                                       \
    ORC_EXPORT void ref_intersect_tri##S(const void* h, const T* tris12, const T* rays8, size_t nrays,  \
        int any, int robust, int threads, HitOf<T>::Type* out, uint64_t* counters) {                    \
        DISPATCH4(intersect_tri, T, any, robust, static_cast<const Bvh3<T>*>(h), tris12, rays8, nrays,  \
                  threads, out, counters); }

REF_IMPL(float, 3, 3f)
REF_IMPL(double, 3, 3d)
REF_IMPL(float, 2, 2f)
REF_IMPL(double, 2, 2d)
REF_IMPL_TRI(float, 3f)
REF_IMPL_TRI(double, 3d)

ORC_EXPORT void ref_bench_tri3f(void* h, const float* tris12, const float* rays8, size_t n, int any, int robust, int threads, int reps,
                                orc_hitf* out, double* seconds) {
    DISPATCH4(bench_tri, float, any, robust, static_cast<const Bvh3<float>*>(h), tris12, rays8, n, threads, reps, out, seconds); }
ORC_EXPORT int ref_hardware_threads(void) { return static_cast<int>(std::thread::hardware_concurrency()); }

} // extern "C"
