// bvh::v2 — C++20 host-side mirror of madmann91/bvh's public interface for the hot path, over libbvh_amd.so.
//
// A program written against the reference headers keeps its types and calls:
//
//     using Node = bvh::v2::Node<float, 3>;
//     bvh::v2::ThreadPool pool;
//     auto bvh = bvh::v2::DefaultBuilder<Node>::build(pool, bboxes, centers, config);     // reference default_builder.h:33
//     bvh.nodes, bvh.prim_ids, bvh.get_root().index ...                                     // reference bvh.h:17-89
//
// and runs the build on the MI355X (hand-written HIP kernels behind the C-ABI of include/bvh_amd.h). The reference's per-ray
// `Bvh::intersect(ray, start, stack, leaf_fn, inner_fn)` keeps working (the walk runs on the device, the lambdas on the calling
// thread, in the reference's order), at kernel launches per ray; what to adopt for throughput is its batched equivalent
// `bvh::v2::amd::intersect_batch<IsAnyHit, IsRobust>(bvh, prims, rays, hits)` with the reference's own leaf intersectors
// (PrecomputedTri / Sphere<T, 3> / Sphere<T, 2>). The reference's example programs compile against these headers unchanged
// (oracle/Makefile `refprogs`). Node<T, 2> is served by the `2f` / `2d`
// families of the C-ABI (serial builders; the reference's thread-pool build of 2D data is undefined above parallel_threshold).
// Layouts are bit-compatible with the reference (Vec, BBox, Ray, Node: SURVEY.md §8 sizes), which is what lets the
// C-ABI take these arrays as they are. Errors throw bvh::v2::amd::Error (the reference has no error channel).
#ifndef BVH_V2_BVH_AMD_HPP
#define BVH_V2_BVH_AMD_HPP

#include <algorithm>
#include <array>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <istream>
#include <limits>
#include <memory>
#include <optional>
#include <ostream>
#include <span>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../bvh_amd.h"

namespace bvh::v2 {

// ---- stream.h: the byte-stream interface Bvh::serialize / deserialize are written against (reference stream.h:9-66) --------
class OutputStream {
public:
    virtual ~OutputStream() = default;
    template <typename T> bool write(const T& value) { return write_raw(&value, sizeof(T)); }
protected:
    virtual bool write_raw(const void* bytes, size_t size) = 0;
};
class InputStream {
public:
    virtual ~InputStream() = default;
    template <typename T> T read(T&& fallback = {}) {          // a short read yields the fallback, like the reference
        T value;
        return read_raw(&value, sizeof(T)) == sizeof(T) ? value : std::move(fallback);
    }
protected:
    virtual size_t read_raw(void* bytes, size_t size) = 0;
};
class StdOutputStream : public OutputStream {
public:
    explicit StdOutputStream(std::ostream& os) : os_(os) {}
protected:
    std::ostream& os_;
    bool write_raw(const void* bytes, size_t size) override {
        os_.write(static_cast<const char*>(bytes), static_cast<std::streamsize>(size));
        return os_.good();
    }
};
class StdInputStream : public InputStream {
public:
    explicit StdInputStream(std::istream& is) : is_(is) {}
protected:
    std::istream& is_;
    size_t read_raw(void* bytes, size_t size) override {
        is_.read(static_cast<char*>(bytes), static_cast<std::streamsize>(size));
        return static_cast<size_t>(is_.gcount());
    }
};

// ---- vec.h / bbox.h / ray.h ---------------------------------------------------------------------------------------
template <typename T, size_t N>
struct Vec {
    T values[N];
    Vec() = default;
    template <typename... Rest> Vec(T x, T y, Rest... rest) : values{ x, y, static_cast<T>(rest)... } {}
    explicit Vec(T x) { for (auto& v : values) v = x; }
    T& operator[](size_t i) { return values[i]; }
    T operator[](size_t i) const { return values[i]; }
};
template <typename T, size_t N> Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (size_t i = 0; i < N; ++i) r[i] = a[i] + b[i]; return r; }
template <typename T, size_t N> Vec<T, N> operator-(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (size_t i = 0; i < N; ++i) r[i] = a[i] - b[i]; return r; }
template <typename T, size_t N> Vec<T, N> operator*(const Vec<T, N>& a, T s) { Vec<T, N> r; for (size_t i = 0; i < N; ++i) r[i] = a[i] * s; return r; }
template <typename T, size_t N> Vec<T, N> operator-(const Vec<T, N>& a) { Vec<T, N> r; for (size_t i = 0; i < N; ++i) r[i] = -a[i]; return r; }
template <typename T, size_t N> Vec<T, N> operator*(T s, const Vec<T, N>& a) { return a * s; }
template <typename T, size_t N> T dot(const Vec<T, N>& a, const Vec<T, N>& b) {   // reference vec.h:98-100: ((0 + a0 b0) + a1 b1) + ...
    T sum = T(0);
    for (size_t i = 0; i < N; ++i) sum = sum + a[i] * b[i];
    return sum;
}
template <typename T, size_t N> T length(const Vec<T, N>& v) { return std::sqrt(dot(v, v)); }
template <typename T, size_t N> Vec<T, N> normalize(const Vec<T, N>& v) { return v * (static_cast<T>(1.) / length(v)); }
template <typename T> Vec<T, 3> cross(const Vec<T, 3>& a, const Vec<T, 3>& b) {
    return Vec<T, 3>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

template <typename T, size_t N>
struct BBox {
    Vec<T, N> min, max;
    BBox() = default;
    BBox(const Vec<T, N>& lo, const Vec<T, N>& hi) : min(lo), max(hi) {}
    explicit BBox(const Vec<T, N>& p) : min(p), max(p) {}
    BBox& extend(const BBox& o) {                             // reference bbox.h:22-27 (second operand wins ties)
        for (size_t i = 0; i < N; ++i) { min[i] = min[i] < o.min[i] ? min[i] : o.min[i]; max[i] = max[i] > o.max[i] ? max[i] : o.max[i]; }
        return *this;
    }
    BBox& extend(const Vec<T, N>& p) { return extend(BBox(p)); }
    Vec<T, N> get_diagonal() const { return max - min; }
    Vec<T, N> get_center() const { return (max + min) * static_cast<T>(0.5); }
    T get_half_area() const {
        auto d = get_diagonal();
        if constexpr (N == 3) return (d[0] + d[1]) * d[2] + d[0] * d[1]; else return d[0] + d[1];
    }
    static BBox make_empty() { return BBox(Vec<T, N>(std::numeric_limits<T>::max()), Vec<T, N>(-std::numeric_limits<T>::max())); }
};

template <typename T, size_t N>
struct Ray {
    Vec<T, N> org, dir;
    T tmin, tmax;
    Ray() = default;
    Ray(const Vec<T, N>& o, const Vec<T, N>& d, T t0 = 0, T t1 = std::numeric_limits<T>::max()) : org(o), dir(d), tmin(t0), tmax(t1) {}
};

// ---- index.h / node.h -------------------------------------------------------------------------------------------------
template <size_t Bits, size_t PrimCountBits>
struct Index {
    static_assert(Bits == 32 || Bits == 64, "bvh_amd: Index of 32 or 64 bits (the reference's UnsignedIntType also knows 8 and 16)");
    static_assert(PrimCountBits >= 1 && PrimCountBits < Bits, "index.h:41");
    using Type = std::conditional_t<Bits == 32, uint32_t, uint64_t>;
    static constexpr size_t bits = Bits, prim_count_bits = PrimCountBits, first_id_bits = Bits - PrimCountBits;   // index.h:36-37
    static constexpr Type max_prim_count = (Type{1} << PrimCountBits) - 1;
    static constexpr Type max_first_id = static_cast<Type>(~Type{0}) >> PrimCountBits;                            // index.h:39
    Type value;
    Index() = default;
    explicit Index(Type v) : value(v) {}
    bool operator==(const Index&) const = default;
    Type first_id() const { return value >> PrimCountBits; }
    Type prim_count() const { return value & max_prim_count; }
    bool is_leaf() const { return prim_count() != 0; }
    bool is_inner() const { return !is_leaf(); }
    void set_first_id(size_t first) { *this = prim_count() ? make_leaf(first, prim_count()) : make_inner(first); }      // index.h:55-57
    void set_prim_count(size_t count) { value = (value & ~max_prim_count) | (static_cast<Type>(count) & max_prim_count); }  // index.h:59-61
    static Index make_leaf(size_t first, size_t count) { return Index((static_cast<Type>(first) << PrimCountBits) | static_cast<Type>(count)); }
    static Index make_inner(size_t first) { return Index(static_cast<Type>(first) << PrimCountBits); }
};

template <typename T, size_t Dim, size_t IndexBits = sizeof(T) * CHAR_BIT, size_t PrimCountBits = 4>
struct Node {
    using Scalar = T;
    using Index = bvh::v2::Index<IndexBits, PrimCountBits>;
    static constexpr size_t dimension = Dim;
    std::array<T, Dim * 2> bounds;                            // {min_x, max_x, min_y, max_y, ...}  (reference node.h:31-37)
    Index index;
    bool operator==(const Node&) const = default;
    bool is_leaf() const { return index.is_leaf(); }
    BBox<T, Dim> get_bbox() const {
        BBox<T, Dim> b;
        for (size_t i = 0; i < Dim; ++i) { b.min[i] = bounds[2 * i]; b.max[i] = bounds[2 * i + 1]; }
        return b;
    }
    void set_bbox(const BBox<T, Dim>& b) { for (size_t i = 0; i < Dim; ++i) { bounds[2 * i] = b.min[i]; bounds[2 * i + 1] = b.max[i]; } }
    void serialize(OutputStream& out) const {                 // reference node.h:90-94: the bounds, then the packed index word
        for (const T& b : bounds) out.write(b);
        out.write(index.value);
    }
    [[nodiscard]] static Node deserialize(InputStream& in) {   // reference node.h:96-102
        Node node;
        for (T& b : node.bounds) b = in.template read<T>();
        node.index = Index(in.template read<typename Index::Type>());
        return node;
    }
};
static_assert(sizeof(Node<float, 3>) == 28 && sizeof(Node<double, 3>) == 56, "layout must equal the reference's");
static_assert(sizeof(Node<float, 2>) == 20 && sizeof(Node<double, 2>) == 40 && sizeof(BBox<float, 2>) == sizeof(bvh_bbox2f) &&
              sizeof(Ray<double, 2>) == sizeof(bvh_ray2d), "2D layouts must equal the reference's");
static_assert(sizeof(BBox<float, 3>) == sizeof(bvh_bbox3f) && sizeof(Vec<float, 3>) == sizeof(bvh_vec3f) && sizeof(Ray<float, 3>) == sizeof(bvh_ray3f));
static_assert(sizeof(BBox<double, 3>) == sizeof(bvh_bbox3d) && sizeof(Ray<double, 3>) == sizeof(bvh_ray3d));

// ---- tri.h / sphere.h ----------------------------------------------------------------------------------------------------
template <typename T, size_t N>
struct Tri {
    Vec<T, N> p0, p1, p2;
    Tri() = default;
    Tri(const Vec<T, N>& a, const Vec<T, N>& b, const Vec<T, N>& c) : p0(a), p1(b), p2(c) {}
    BBox<T, N> get_bbox() const { return BBox<T, N>(p0).extend(p1).extend(p2); }
    Vec<T, N> get_center() const { return (p0 + p1 + p2) * static_cast<T>(1. / 3.); }
};
template <typename T>
struct PrecomputedTri {
    Vec<T, 3> p0, e1, e2, n;
    PrecomputedTri() = default;
    PrecomputedTri(const Vec<T, 3>& a, const Vec<T, 3>& b, const Vec<T, 3>& c) : p0(a), e1(a - b), e2(c - a), n(cross(e1, e2)) {}
    PrecomputedTri(const Tri<T, 3>& t) : PrecomputedTri(t.p0, t.p1, t.p2) {}
    Tri<T, 3> convert_to_tri() const { return Tri<T, 3>(p0, p0 - e1, e2 + p0); }
    BBox<T, 3> get_bbox() const { return convert_to_tri().get_bbox(); }
    Vec<T, 3> get_center() const { return convert_to_tri().get_center(); }
    // The leaf test a host callback of Bvh::intersect runs (reference tri.h:56-74); the batch kernels carry the same
    // sequence of operations (traverse.hip). Returns (t, u, v).
    std::optional<std::tuple<T, T, T>> intersect(const Ray<T, 3>& ray) const {
        const auto c = p0 - ray.org;
        const auto r = cross(ray.dir, c);
        const T inv_det = static_cast<T>(1.) / dot(n, ray.dir);
        const T u = dot(r, e2) * inv_det, v = dot(r, e1) * inv_det, w = static_cast<T>(1.) - u - v;
        const T tolerance = -std::numeric_limits<T>::epsilon();
        if (u >= tolerance && v >= tolerance && w >= tolerance) {
            const T t = dot(n, c) * inv_det;
            if (t >= ray.tmin && t <= ray.tmax) return std::make_optional(std::tuple<T, T, T>{ t, u, v });
        }
        return std::nullopt;
    }
};
template <typename T, size_t N>
struct Sphere {
    Vec<T, N> center;
    T radius;
    Vec<T, N> get_center() const { return center; }
    BBox<T, N> get_bbox() const { return BBox<T, N>(center - Vec<T, N>(radius), center + Vec<T, N>(radius)); }
    // reference sphere.h:32-49: the two roots clipped to [tmin, tmax]
    template <bool AssumeNormalized = false>
    std::optional<std::pair<T, T>> intersect(const Ray<T, N>& ray) const {
        const auto oc = ray.org - center;
        const T a = AssumeNormalized ? static_cast<T>(1.) : dot(ray.dir, ray.dir);
        const T b = static_cast<T>(2.) * dot(ray.dir, oc);
        const T c = dot(oc, oc) - radius * radius;
        const T delta = b * b - static_cast<T>(4.) * a * c;
        if (delta >= 0) {
            const T inv = -static_cast<T>(0.5) / a, root = std::sqrt(delta);
            const T x0 = (b + root) * inv, x1 = (b - root) * inv;
            const T t0 = x0 > ray.tmin ? x0 : ray.tmin, t1 = x1 < ray.tmax ? x1 : ray.tmax;
            if (t0 <= t1) return std::make_optional(std::make_pair(t0, t1));
        }
        return std::nullopt;
    }
};

// ---- thread_pool.h / executor.h / stack.h: kept for source compatibility; the GPU grid replaces the pool ----------------
class ThreadPool {
public:
    explicit ThreadPool(size_t thread_count = 0) : count_(thread_count) {}
    size_t get_thread_count() const { return count_; }
private:
    size_t count_;
};
struct SequentialExecutor {
    template <typename Loop> void for_each(size_t begin, size_t end, const Loop& loop) { loop(begin, end); }
};
struct ParallelExecutor {                                     // host-side prep loops of the examples run inline
    ThreadPool& thread_pool;
    explicit ParallelExecutor(ThreadPool& pool, size_t = 1024) : thread_pool(pool) {}
    template <typename Loop> void for_each(size_t begin, size_t end, const Loop& loop) { loop(begin, end); }
};
template <typename T, unsigned Capacity>
struct SmallStack {                                           // reference stack.h:11-30 (the device kernel holds its own)
    static constexpr unsigned capacity = Capacity;
    T elems[Capacity];
    unsigned size = 0;
    bool is_empty() const { return size == 0; }
    void push(const T& t) { elems[size++] = t; }
    T pop() { return elems[--size]; }
};
template <typename T>
struct GrowingStack {                                         // reference stack.h:33-48
    std::vector<T> elems;
    bool is_empty() const { return elems.empty(); }
    void push(const T& t) { elems.push_back(t); }
    T pop() { T top = std::move(elems.back()); elems.pop_back(); return top; }
    void clear() { elems.clear(); }
};

namespace amd {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc, const char* what) { if (rc != 0) throw Error(std::string(what) + ": " + bvh_amd_last_error()); }

template <typename T, size_t N = 3> struct Api;
template <> struct Api<float, 3> {
    using Handle = bvh3f; using CHit = bvh_hit3f;
    static Handle* build(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c) { return bvh3f_build(p, static_cast<const bvh_bbox3f*>(bb), static_cast<const bvh_vec3f*>(cc), n, c); }
    static Handle* build_device(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b) { return bvh3f_build_device(static_cast<const float*>(d_bb), static_cast<const float*>(d_cc), n, c, b, nullptr); }
    static Handle* build_sah(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h) { return bvh3f_build_sah(p, static_cast<const bvh_bbox3f*>(bb), static_cast<const bvh_vec3f*>(cc), n, c, h); }
    static Handle* build_device_sah(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b, const bvh_amd_sah_config* h) { return bvh3f_build_device_sah(static_cast<const float*>(d_bb), static_cast<const float*>(d_cc), n, c, b, h, nullptr); }
    static Handle* build_device_binned(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h, size_t bins) { return bvh3f_build_device_binned(static_cast<const float*>(d_bb), static_cast<const float*>(d_cc), n, c, h, bins, nullptr); }
    static Handle* build_minitree(const void* d_bb, const void* d_cc, size_t n, const bvh_amd_minitree_config* c) { return bvh3f_build_minitree_device(static_cast<const float*>(d_bb), static_cast<const float*>(d_cc), n, c, nullptr); }
    static Handle* from_nodes(const void* nodes, size_t nn, const size_t* ids, size_t np) { return bvh3f_from_nodes(nodes, nn, ids, np); }
    static void destroy(Handle* h) { bvh3f_destroy(h); }
    static size_t node_count(const Handle* h) { return bvh3f_get_node_count(h); }
    static size_t prim_count(const Handle* h) { return bvh3f_get_prim_count(h); }
    static void copy_nodes(const Handle* h, void* out) { bvh3f_copy_nodes(h, out); }
    static void copy_prim_ids(const Handle* h, size_t* out) { bvh3f_copy_prim_ids(h, out); }
    static Handle* extract(Handle* h, size_t root_id) { return bvh3f_extract(h, root_id); }
    static void optimize(Handle* h) { bvh3f_optimize(nullptr, h); }
    static int optimize_config(Handle* h, const bvh_amd_optimize_config* c) { return bvh3f_optimize_config(h, c); }
    static void refit(Handle* h) { bvh3f_refit(h); }
    static const uint32_t* device_prim_ids(const Handle* h) { return bvh3f_device_prim_ids(h); }
    static Handle* broadcast(bvh_amd_comm* c, int root, Handle* h, const void* p, size_t bytes, void** p_out, size_t* bytes_out) { return bvh3f_broadcast(c, root, h, p, bytes, p_out, bytes_out, nullptr); }
    static int replicate(Handle* h, const void* p, size_t bytes, int n, const int* devs, Handle** out, void** p_out) { return bvh3f_replicate(h, p, bytes, n, devs, out, p_out); }
    static int precompute(const void* t9, const uint32_t* perm, size_t n, void* out) { return bvh_amd_precompute_tris3f(static_cast<const float*>(t9), perm, n, static_cast<float*>(out), nullptr); }
    static int trace_tri(const Handle* h, const void* prims, const void* rays, size_t n, unsigned f, void* hits) { return bvh3f_intersect_rays_tri(h, static_cast<const float*>(prims), static_cast<const bvh_ray3f*>(rays), n, f, static_cast<bvh_hit3f*>(hits), nullptr, nullptr); }
    static int trace_sphere(const Handle* h, const void* prims, const void* rays, size_t n, unsigned f, void* hits) { return bvh3f_intersect_rays_sphere(h, static_cast<const float*>(prims), static_cast<const bvh_ray3f*>(rays), n, f, static_cast<bvh_hit3f*>(hits), nullptr, nullptr); }
    static int visit(const Handle* h, const void* ray, size_t start, unsigned f, void* user, bool (*leaf)(void*, float*, size_t, size_t), void (*inner)(void*, size_t)) {
        const bvh_amd_ray_visitorf v{ user, leaf, inner };
        return bvh3f_intersect_ray_visit(h, static_cast<const bvh_ray3f*>(ray), start, f, &v);
    }
};
template <> struct Api<double, 3> {
    using Handle = bvh3d; using CHit = bvh_hit3d;
    static Handle* build(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c) { return bvh3d_build(p, static_cast<const bvh_bbox3d*>(bb), static_cast<const bvh_vec3d*>(cc), n, c); }
    static Handle* build_device(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b) { return bvh3d_build_device(static_cast<const double*>(d_bb), static_cast<const double*>(d_cc), n, c, b, nullptr); }
    static Handle* build_sah(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h) { return bvh3d_build_sah(p, static_cast<const bvh_bbox3d*>(bb), static_cast<const bvh_vec3d*>(cc), n, c, h); }
    static Handle* build_device_sah(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b, const bvh_amd_sah_config* h) { return bvh3d_build_device_sah(static_cast<const double*>(d_bb), static_cast<const double*>(d_cc), n, c, b, h, nullptr); }
    static Handle* build_device_binned(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h, size_t bins) { return bvh3d_build_device_binned(static_cast<const double*>(d_bb), static_cast<const double*>(d_cc), n, c, h, bins, nullptr); }
    static Handle* build_minitree(const void* d_bb, const void* d_cc, size_t n, const bvh_amd_minitree_config* c) { return bvh3d_build_minitree_device(static_cast<const double*>(d_bb), static_cast<const double*>(d_cc), n, c, nullptr); }
    static Handle* from_nodes(const void* nodes, size_t nn, const size_t* ids, size_t np) { return bvh3d_from_nodes(nodes, nn, ids, np); }
    static void destroy(Handle* h) { bvh3d_destroy(h); }
    static size_t node_count(const Handle* h) { return bvh3d_get_node_count(h); }
    static size_t prim_count(const Handle* h) { return bvh3d_get_prim_count(h); }
    static void copy_nodes(const Handle* h, void* out) { bvh3d_copy_nodes(h, out); }
    static void copy_prim_ids(const Handle* h, size_t* out) { bvh3d_copy_prim_ids(h, out); }
    static Handle* extract(Handle* h, size_t root_id) { return bvh3d_extract(h, root_id); }
    static void optimize(Handle* h) { bvh3d_optimize(nullptr, h); }
    static int optimize_config(Handle* h, const bvh_amd_optimize_config* c) { return bvh3d_optimize_config(h, c); }
    static void refit(Handle* h) { bvh3d_refit(h); }
    static const uint32_t* device_prim_ids(const Handle* h) { return bvh3d_device_prim_ids(h); }
    static Handle* broadcast(bvh_amd_comm* c, int root, Handle* h, const void* p, size_t bytes, void** p_out, size_t* bytes_out) { return bvh3d_broadcast(c, root, h, p, bytes, p_out, bytes_out, nullptr); }
    static int replicate(Handle* h, const void* p, size_t bytes, int n, const int* devs, Handle** out, void** p_out) { return bvh3d_replicate(h, p, bytes, n, devs, out, p_out); }
    static int precompute(const void* t9, const uint32_t* perm, size_t n, void* out) { return bvh_amd_precompute_tris3d(static_cast<const double*>(t9), perm, n, static_cast<double*>(out), nullptr); }
    static int trace_tri(const Handle* h, const void* prims, const void* rays, size_t n, unsigned f, void* hits) { return bvh3d_intersect_rays_tri(h, static_cast<const double*>(prims), static_cast<const bvh_ray3d*>(rays), n, f, static_cast<bvh_hit3d*>(hits), nullptr, nullptr); }
    static int trace_sphere(const Handle* h, const void* prims, const void* rays, size_t n, unsigned f, void* hits) { return bvh3d_intersect_rays_sphere(h, static_cast<const double*>(prims), static_cast<const bvh_ray3d*>(rays), n, f, static_cast<bvh_hit3d*>(hits), nullptr, nullptr); }
    static int visit(const Handle* h, const void* ray, size_t start, unsigned f, void* user, bool (*leaf)(void*, double*, size_t, size_t), void (*inner)(void*, size_t)) {
        const bvh_amd_ray_visitord v{ user, leaf, inner };
        return bvh3d_intersect_ray_visit(h, static_cast<const bvh_ray3d*>(ray), start, f, &v);
    }
};


#define BVH_AMD_API_2D(T, S)                                                                                                              \
template <> struct Api<T, 2> {                                                                                                            \
    using Handle = bvh##S; using CHit = std::conditional_t<std::is_same_v<T, float>, bvh_hit3f, bvh_hit3d>;                              \
    static Handle* build(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c) { return bvh##S##_build(p, static_cast<const bvh_bbox##S*>(bb), static_cast<const bvh_vec##S*>(cc), n, c); } \
    static Handle* build_device(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b) { return bvh##S##_build_device(static_cast<const T*>(d_bb), static_cast<const T*>(d_cc), n, c, b, nullptr); } \
    static Handle* build_sah(bvh_thread_pool* p, const void* bb, const void* cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h) { return bvh##S##_build_sah(p, static_cast<const bvh_bbox##S*>(bb), static_cast<const bvh_vec##S*>(cc), n, c, h); } \
    static Handle* build_device_sah(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, bvh_amd_builder b, const bvh_amd_sah_config* h) { return bvh##S##_build_device_sah(static_cast<const T*>(d_bb), static_cast<const T*>(d_cc), n, c, b, h, nullptr); } \
    static Handle* build_device_binned(const void* d_bb, const void* d_cc, size_t n, const bvh_build_config* c, const bvh_amd_sah_config* h, size_t bins) { return bvh##S##_build_device_binned(static_cast<const T*>(d_bb), static_cast<const T*>(d_cc), n, c, h, bins, nullptr); } \
    static Handle* from_nodes(const void* nodes, size_t nn, const size_t* ids, size_t np) { return bvh##S##_from_nodes(nodes, nn, ids, np); } \
    static void destroy(Handle* h) { bvh##S##_destroy(h); }                                                                               \
    static size_t node_count(const Handle* h) { return bvh##S##_get_node_count(h); }                                                      \
    static size_t prim_count(const Handle* h) { return bvh##S##_get_prim_count(h); }                                                      \
    static void copy_nodes(const Handle* h, void* out) { bvh##S##_copy_nodes(h, out); }                                                   \
    static void copy_prim_ids(const Handle* h, size_t* out) { bvh##S##_copy_prim_ids(h, out); }                                           \
    static Handle* extract(Handle* h, size_t root_id) { return bvh##S##_extract(h, root_id); }                                            \
    static void optimize(Handle* h) { bvh##S##_optimize(nullptr, h); }                                                                    \
    static int optimize_config(Handle* h, const bvh_amd_optimize_config* c) { return bvh##S##_optimize_config(h, c); }                    \
    static void refit(Handle* h) { bvh##S##_refit(h); }                                                                                   \
    static const uint32_t* device_prim_ids(const Handle* h) { return bvh##S##_device_prim_ids(h); }                                       \
    static Handle* broadcast(bvh_amd_comm* c, int root, Handle* h, const void* p, size_t bytes, void** p_out, size_t* bytes_out) { return bvh##S##_broadcast(c, root, h, p, bytes, p_out, bytes_out, nullptr); } \
    static int replicate(Handle* h, const void* p, size_t bytes, int n, const int* devs, Handle** out, void** p_out) { return bvh##S##_replicate(h, p, bytes, n, devs, out, p_out); } \
    static int trace_sphere(const Handle* h, const void* prims, const void* rays, size_t n, unsigned f, void* hits) { return bvh##S##_intersect_rays_sphere(h, static_cast<const T*>(prims), static_cast<const bvh_ray##S*>(rays), n, f, static_cast<CHit*>(hits), nullptr, nullptr); } \
    static int visit(const Handle* h, const void* ray, size_t start, unsigned f, void* user, bool (*leaf)(void*, T*, size_t, size_t), void (*inner)(void*, size_t)) { \
        const std::conditional_t<std::is_same_v<T, float>, bvh_amd_ray_visitorf, bvh_amd_ray_visitord> v{ user, leaf, inner };           \
        return bvh##S##_intersect_ray_visit(h, static_cast<const bvh_ray##S*>(ray), start, f, &v); }                                      \
};
BVH_AMD_API_2D(float, 2f)
BVH_AMD_API_2D(double, 2d)
#undef BVH_AMD_API_2D

// RAII device array (HBM of the current HIP device)
template <typename T>
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n) : size_(n), ptr_(static_cast<T*>(bvh_amd_device_alloc(n * sizeof(T)))) { if (!ptr_) throw Error(bvh_amd_last_error()); }
    DeviceArray(std::span<const T> host) : DeviceArray(host.size()) { check(bvh_amd_copy_to_device(ptr_, host.data(), host.size_bytes()), "copy_to_device"); }
    DeviceArray(DeviceArray&& o) noexcept : size_(o.size_), ptr_(o.ptr_) { o.ptr_ = nullptr; o.size_ = 0; }
    DeviceArray& operator=(DeviceArray&& o) noexcept { if (this != &o) { bvh_amd_device_free(ptr_); ptr_ = o.ptr_; size_ = o.size_; o.ptr_ = nullptr; o.size_ = 0; } return *this; }
    ~DeviceArray() { bvh_amd_device_free(ptr_); }
    T* data() const { return ptr_; }
    size_t size() const { return size_; }
    void download(std::span<T> out) const { check(bvh_amd_copy_to_host(out.data(), ptr_, out.size_bytes()), "copy_to_host"); }
private:
    size_t size_ = 0;
    T* ptr_ = nullptr;
};

// One record per ray; prim = BVH-order primitive index (the `i` of the reference's leaf lambda), invalid on a miss.
template <typename T> struct Hit;
template <> struct Hit<float>  { uint32_t prim; float t, u, v; static constexpr uint32_t invalid = BVH_AMD_INVALID; };
template <> struct Hit<double> { uint32_t prim; uint32_t pad; double t, u, v; static constexpr uint32_t invalid = BVH_AMD_INVALID; };
static_assert(sizeof(Hit<float>) == sizeof(bvh_hit3f) && sizeof(Hit<double>) == sizeof(bvh_hit3d));

} // namespace amd

// ---- bvh.h ----------------------------------------------------------------------------------------------------------------
template <typename Node> class ReinsertionOptimizer;

template <typename Node>
struct Bvh {
    using Index = typename Node::Index;
    using Scalar = typename Node::Scalar;
    using Ray = bvh::v2::Ray<Scalar, Node::dimension>;
    static_assert(Node::dimension == 2 || Node::dimension == 3, "bvh_amd: 2D and 3D BVHs");

    std::vector<Node> nodes;                                  // host mirror, reference layout (bvh.h:22-23)
    std::vector<size_t> prim_ids;

    Bvh() = default;
    Bvh(Bvh&&) = default;
    Bvh& operator=(Bvh&&) = default;
    bool operator==(const Bvh& o) const { return nodes == o.nodes && prim_ids == o.prim_ids; }
    const Node& get_root() const { return nodes[0]; }
    static bool is_left_sibling(size_t id) { return id % 2 == 1; }
    static size_t get_sibling_id(size_t id) { return is_left_sibling(id) ? id + 1 : id - 1; }
    static size_t get_left_sibling_id(size_t id) { return is_left_sibling(id) ? id : id - 1; }
    static size_t get_right_sibling_id(size_t id) { return is_left_sibling(id) ? id + 1 : id; }

    // Bvh::extract_bvh (reference bvh.h:92-122) on the device
    [[nodiscard]] Bvh extract_bvh(size_t root_id) const {
        auto* h = amd::Api<Scalar, Node::dimension>::extract(device(), root_id);
        if (!h) throw amd::Error(bvh_amd_last_error());
        Bvh out;
        out.adopt(h);
        return out;
    }

    // Bvh::refit (reference bvh.h:211-218) on the device; `nodes` may have been edited by the caller.
    void refit() { push(); amd::Api<Scalar, Node::dimension>::refit(device_.get()); pull(); }
    // with the reference's leaf callback (bvh.h:211-218 via traverse_bottom_up, :186-208): leaf_fn recomputes each leaf's
    // box on the host, in the reference's order (leaves by descending node index); the inner nodes are refitted on the device
    template <typename LeafFn>
    void refit(LeafFn&& leaf_fn) {
        for (size_t i = nodes.size(); i-- > 0;)
            if (nodes[i].is_leaf()) leaf_fn(nodes[i]);
        refit();
    }

    // Bvh::intersect (reference bvh.h:72-73, :160-182) for one ray with host callbacks: leaf_fn(begin, end) -> bool over
    // BVH-order primitive ranges, optional inner_fn(left, right) per visited pair. The walk runs on the device
    // (bvhXX_intersect_ray_visit); leaf_fn shortens the ray by writing ray.tmax, exactly as with the reference, which is why
    // the ray is re-read after every leaf. The caller's stack object is not used (the device keeps the stack).
    struct IgnoreArgs { template <typename... Args> void operator()(Args&&...) const {} };
    template <bool IsAnyHit, bool IsRobust, typename Stack, typename LeafFn, typename InnerFn = IgnoreArgs>
    void intersect(const Ray& ray, Index start, Stack&, LeafFn&& leaf_fn, InnerFn&& inner_fn = {}) const {
        struct Visit {
            const Bvh* bvh; const Ray* ray; std::remove_reference_t<LeafFn>* leaf; std::remove_reference_t<InnerFn>* inner;
            static bool on_leaf(void* self, Scalar* t, size_t begin, size_t end) {
                auto* v = static_cast<Visit*>(self);
                const bool hit = static_cast<bool>((*v->leaf)(begin, end));
                *t = v->ray->tmax;                            // the reference's traversal reads ray.tmax live (node.h:105-117)
                return hit;
            }
            static void on_inner(void* self, size_t first) {
                auto* v = static_cast<Visit*>(self);
                (*v->inner)(v->bvh->nodes[first], v->bvh->nodes[first + 1]);
            }
        } visit{ this, &ray, &leaf_fn, &inner_fn };
        const unsigned flags = (IsAnyHit ? unsigned(BVH_AMD_RAY_ANY_HIT) : 0u) | (IsRobust ? unsigned(BVH_AMD_RAY_ROBUST) : 0u);
        constexpr bool wants_inner = !std::is_same_v<std::remove_cvref_t<InnerFn>, IgnoreArgs>;
        // (the device twin keeps the default Index packing whatever this Node's: the start index is handed over in that form)
        using DeviceIndex = typename bvh::v2::Node<Scalar, Node::dimension>::Index;
        const auto dev_start = start.is_leaf() ? DeviceIndex::make_leaf(start.first_id(), start.prim_count()) : DeviceIndex::make_inner(start.first_id());
        amd::check(amd::Api<Scalar, Node::dimension>::visit(device(), &ray, static_cast<size_t>(dev_start.value), flags, &visit, &Visit::on_leaf,
                                                            wants_inner ? &Visit::on_inner : nullptr), "intersect_ray_visit");
    }

    // Bvh::traverse_top_down (reference bvh.h:68-70, :125-157) with a user InnerFn that STEERS the descent: inner_fn(left, right) returns
    // {visit left, visit right, right first}; leaf_fn(begin, end) -> bool. Like traverse_bottom_up below this is a host-side utility over
    // the mirror's nodes: user code decides every step, so there is nothing to offload — the walks the LIBRARY defines (Bvh::intersect,
    // amd::intersect_batch) run on the device. Same visit order as the reference: the nearer (or left) child first, the other one on the
    // caller's stack; with IsAnyHit the walk ends at the first leaf_fn that returns true.
    template <bool IsAnyHit, typename Stack, typename LeafFn, typename InnerFn>
    void traverse_top_down(Index start, Stack& stack, LeafFn&& leaf_fn, InnerFn&& inner_fn) const {
        stack.push(start);
        while (!stack.is_empty()) {
            Index at = stack.pop();
            bool dead_end = false;
            while (!dead_end && at.prim_count() == 0) {
                const Node& left = nodes[at.first_id()];
                const Node& right = nodes[at.first_id() + 1];
                const auto [go_left, go_right, right_first] = inner_fn(left, right);
                if (go_left && go_right) {
                    stack.push(right_first ? left.index : right.index);
                    at = right_first ? right.index : left.index;
                } else if (go_left) at = left.index;
                else if (go_right) at = right.index;
                else dead_end = true;
            }
            if (dead_end) continue;
            [[maybe_unused]] const bool was_hit = static_cast<bool>(leaf_fn(at.first_id(), at.first_id() + at.prim_count()));
            if constexpr (IsAnyHit) { if (was_hit) return; }
        }
    }

    // Bvh::traverse_bottom_up (reference bvh.h:76-78, :185-208) with arbitrary host functors: a host-side utility over the
    // mirror's nodes (user code runs per node, so there is nothing to offload; the library's own bottom-up pass, refit, is a
    // device kernel). Leaves are visited by descending node index; an inner node right after its second child.
    template <typename LeafFn = IgnoreArgs, typename InnerFn = IgnoreArgs>
    void traverse_bottom_up(LeafFn&& leaf_fn = {}, InnerFn&& inner_fn = {}) {
        std::vector<size_t> parent_of(nodes.size(), 0);
        std::vector<unsigned char> pending(nodes.size(), 0);   // children of an inner node not visited yet
        for (size_t i = 0; i < nodes.size(); ++i) {
            if (nodes[i].is_leaf()) continue;
            const size_t first = nodes[i].index.first_id();
            parent_of[first] = parent_of[first + 1] = i;
            pending[i] = 2;
        }
        for (size_t i = nodes.size(); i-- > 0;) {
            if (!nodes[i].is_leaf()) continue;
            leaf_fn(nodes[i]);
            for (size_t up = i; up != 0;) {
                up = parent_of[up];
                if (--pending[up] != 0) break;
                inner_fn(nodes[up]);
            }
        }
        device_.reset();                                      // the functors may have edited the nodes
    }

    // Bvh::serialize / deserialize (reference bvh.h:221-243): counts, nodes, primitive ids, all as IndexType; the same byte
    // stream bvhXX_save / bvhXX_serialize of the C-ABI produce
    template <typename IndexType = typename Index::Type>
    void serialize(OutputStream& out) const {
        out.write(static_cast<IndexType>(nodes.size()));
        out.write(static_cast<IndexType>(prim_ids.size()));
        for (const Node& node : nodes) node.serialize(out);
        for (size_t id : prim_ids) out.write(static_cast<IndexType>(id));
    }
    template <typename IndexType = typename Index::Type>
    [[nodiscard]] static Bvh deserialize(InputStream& in) {
        Bvh bvh;
        bvh.nodes.resize(in.template read<IndexType>());
        bvh.prim_ids.resize(in.template read<IndexType>());
        for (Node& node : bvh.nodes) node = Node::deserialize(in);
        for (size_t& id : bvh.prim_ids) id = in.template read<IndexType>();
        return bvh;
    }

    // the device-resident twin (built by DefaultBuilder, or uploaded on demand)
    typename amd::Api<Scalar, Node::dimension>::Handle* device() const {
        if (!device_) const_cast<Bvh*>(this)->push();
        return device_.get();
    }
    void share(const Bvh& o) { nodes = o.nodes; prim_ids = o.prim_ids; device_ = o.device_; }     // the same device twin, one more owner
    void adopt(typename amd::Api<Scalar, Node::dimension>::Handle* h) {
        device_ = std::shared_ptr<typename amd::Api<Scalar, Node::dimension>::Handle>(h, [](auto* p) { amd::Api<Scalar, Node::dimension>::destroy(p); });
        pull();
    }

private:
    std::shared_ptr<typename amd::Api<Scalar, Node::dimension>::Handle> device_;
    // The device twin always holds the reference's DEFAULT node (node.h:20-22: IndexBits = bits of the scalar, PrimCountBits = 4). A Node
    // with other Index parameters (node.h:21-22, index.h:32-41) is the same tree with its index word packed differently: the mirror
    // re-packs (first_id, prim_count) at this boundary and refuses, loudly, what the other side cannot represent.
    using DeviceNode = bvh::v2::Node<Scalar, Node::dimension>;
    static constexpr bool device_layout = std::is_same_v<Node, DeviceNode>;
    void push() {                                             // host mirror -> device
        const void* src = nodes.data();
        std::vector<DeviceNode> packed;
        if constexpr (!device_layout) {
            packed.resize(nodes.size());
            for (size_t i = 0; i < nodes.size(); ++i) {
                const auto first = static_cast<uint64_t>(nodes[i].index.first_id()), count = static_cast<uint64_t>(nodes[i].index.prim_count());
                if (count > static_cast<uint64_t>(DeviceNode::Index::max_prim_count) || first > static_cast<uint64_t>(DeviceNode::Index::max_first_id))
                    throw amd::Error("bvh_amd: a node of this Bvh does not fit the device node (4-bit primitive count, first_id in the remaining bits)");
                packed[i].bounds = nodes[i].bounds;
                packed[i].index = count ? DeviceNode::Index::make_leaf(first, count) : DeviceNode::Index::make_inner(first);
            }
            src = packed.data();
        }
        auto* h = amd::Api<Scalar, Node::dimension>::from_nodes(src, nodes.size(), prim_ids.data(), prim_ids.size());
        if (!h) throw amd::Error(bvh_amd_last_error());
        device_ = std::shared_ptr<typename amd::Api<Scalar, Node::dimension>::Handle>(h, [](auto* p) { amd::Api<Scalar, Node::dimension>::destroy(p); });
    }
    void pull() {                                             // device -> host mirror
        nodes.resize(amd::Api<Scalar, Node::dimension>::node_count(device_.get()));
        prim_ids.resize(amd::Api<Scalar, Node::dimension>::prim_count(device_.get()));
        if constexpr (device_layout) {
            amd::Api<Scalar, Node::dimension>::copy_nodes(device_.get(), nodes.data());
        } else {
            std::vector<DeviceNode> packed(nodes.size());
            amd::Api<Scalar, Node::dimension>::copy_nodes(device_.get(), packed.data());
            constexpr uint64_t first_limit = static_cast<uint64_t>(Index::max_first_id);
            for (size_t i = 0; i < nodes.size(); ++i) {
                const auto first = static_cast<uint64_t>(packed[i].index.first_id()), count = static_cast<uint64_t>(packed[i].index.prim_count());
                if (count > static_cast<uint64_t>(Index::max_prim_count) || first > first_limit)
                    throw amd::Error("bvh_amd: the built tree does not fit this Node's Index (index.h:38-41: PrimCountBits / first_id bits); "
                                     "lower Config::max_leaf_size or widen the Index");
                nodes[i].bounds = packed[i].bounds;
                nodes[i].index = count ? Index::make_leaf(first, count) : Index::make_inner(first);
            }
        }
        amd::Api<Scalar, Node::dimension>::copy_prim_ids(device_.get(), prim_ids.data());
    }
    template <typename N> friend class ReinsertionOptimizer;
};

// ---- split_heuristic.h ----------------------------------------------------------------------------------------------------------
template <typename T>
class SplitHeuristic {                                        // reference split_heuristic.h:11-43; the device builders evaluate it
public:
    SplitHeuristic(size_t log_cluster_size = 0, T cost_ratio = static_cast<T>(1.)) : log_cluster_size_(log_cluster_size), cost_ratio_(cost_ratio) {}
    size_t get_prim_count(size_t size) const { return (size + ((size_t{1} << log_cluster_size_) - 1)) >> log_cluster_size_; }
    template <size_t N> T get_leaf_cost(size_t begin, size_t end, const BBox<T, N>& bbox) const { return bbox.get_half_area() * static_cast<T>(get_prim_count(end - begin)); }
    template <size_t N> T get_non_split_cost(size_t begin, size_t end, const BBox<T, N>& bbox) const {
        return bbox.get_half_area() * (static_cast<T>(get_prim_count(end - begin)) - cost_ratio_);
    }
    bvh_amd_sah_config c_config() const { return bvh_amd_sah_config{ log_cluster_size_, static_cast<double>(cost_ratio_) }; }
private:
    size_t log_cluster_size_;
    T cost_ratio_;
};

// ---- top_down_sah_builder.h / binned_sah_builder.h / sweep_sah_builder.h -------------------------------------
template <typename Node>
class TopDownSahBuilder {
protected:
    using Scalar = typename Node::Scalar;
    using Vec = bvh::v2::Vec<Scalar, Node::dimension>;
    using BBox = bvh::v2::BBox<Scalar, Node::dimension>;
public:
    struct Config {                                           // reference top_down_sah_builder.h:27-40
        SplitHeuristic<Scalar> sah;
        size_t min_leaf_size = 1;
        size_t max_leaf_size = 8;
    };
protected:
    static Bvh<Node> run(bvh_amd_builder which, std::span<const BBox> bboxes, std::span<const Vec> centers, const Config& config, size_t bin_count = 8) {
        if (bboxes.size() != centers.size()) throw amd::Error("bvh_amd: bboxes and centers differ in length");
        bvh_build_config c;
        c.quality = BVH_BUILD_QUALITY_HIGH;                   // (unused by the explicit builders)
        c.min_leaf_size = config.min_leaf_size; c.max_leaf_size = config.max_leaf_size; c.parallel_threshold = 1024;
        const bvh_amd_sah_config sah = config.sah.c_config();
        amd::DeviceArray<BBox> d_bb(bboxes);
        amd::DeviceArray<Vec> d_cc(centers);
        auto* h = which == BVH_AMD_BUILDER_BINNED && bin_count != 8
            ? amd::Api<Scalar, Node::dimension>::build_device_binned(d_bb.data(), d_cc.data(), bboxes.size(), &c, &sah, bin_count)
            : amd::Api<Scalar, Node::dimension>::build_device_sah(d_bb.data(), d_cc.data(), bboxes.size(), &c, which, &sah);
        if (!h) throw amd::Error(bvh_amd_last_error());
        Bvh<Node> bvh;
        bvh.adopt(h);
        return bvh;
    }
};

template <typename Node, size_t BinCount = 8>
class BinnedSahBuilder : public TopDownSahBuilder<Node> {     // reference binned_sah_builder.h:18-38
    // BinCount (reference binned_sah_builder.h:18): the default 8 runs the tuned device path; 4, 16 and 32 run the same builder with
    // their own fill_bins / find_best_split instantiations (bvhXX_build_device_binned), bit-identical to the reference's template
    static_assert(BinCount == 4 || BinCount == 8 || BinCount == 16 || BinCount == 32, "bvh_amd: BinnedSahBuilder is instantiated for BinCount 4, 8, 16 and 32");
    using Base = TopDownSahBuilder<Node>;
public:
    using typename Base::Config;
    [[nodiscard]] static Bvh<Node> build(std::span<const typename Base::BBox> bboxes, std::span<const typename Base::Vec> centers, const Config& config = {}) {
        return Base::run(BVH_AMD_BUILDER_BINNED, bboxes, centers, config, BinCount);
    }
};

template <typename Node>
class SweepSahBuilder : public TopDownSahBuilder<Node> {      // reference sweep_sah_builder.h:30-36
    using Base = TopDownSahBuilder<Node>;
public:
    using typename Base::Config;
    [[nodiscard]] static Bvh<Node> build(std::span<const typename Base::BBox> bboxes, std::span<const typename Base::Vec> centers, const Config& config = {}) {
        return Base::run(BVH_AMD_BUILDER_SWEEP, bboxes, centers, config);
    }
};

// ---- default_builder.h ----------------------------------------------------------------------------------------------------
template <typename Node>
class DefaultBuilder {
    using Scalar = typename Node::Scalar;
    using Vec = bvh::v2::Vec<Scalar, Node::dimension>;
    using BBox = bvh::v2::BBox<Scalar, Node::dimension>;
public:
    enum class Quality { Low, Medium, High };
    struct Config : TopDownSahBuilder<Node>::Config {         // default_builder.h:23-30: sah, min_leaf_size, max_leaf_size, and
        Quality quality = Quality::High;
        size_t parallel_threshold = 1024;
    };
    // with a thread pool: the reference's mini-tree builder semantics (default_builder.h:33-46)
    [[nodiscard]] static Bvh<Node> build(ThreadPool& pool, std::span<const BBox> bboxes, std::span<const Vec> centers, const Config& config = {}) {
        return run(reinterpret_cast<bvh_thread_pool*>(&pool), bboxes, centers, config);
    }
    // without: the serial builders' semantics (default_builder.h:49-62)
    [[nodiscard]] static Bvh<Node> build(std::span<const BBox> bboxes, std::span<const Vec> centers, const Config& config = {}) {
        return run(nullptr, bboxes, centers, config);
    }
private:
    static Bvh<Node> run(bvh_thread_pool* pool, std::span<const BBox> bboxes, std::span<const Vec> centers, const Config& config) {
        bvh_build_config c;
        c.quality = static_cast<bvh_build_quality>(config.quality);
        c.min_leaf_size = config.min_leaf_size; c.max_leaf_size = config.max_leaf_size; c.parallel_threshold = config.parallel_threshold;
        const bvh_amd_sah_config sah = config.sah.c_config();
        auto* h = amd::Api<Scalar, Node::dimension>::build_sah(pool, bboxes.data(), centers.data(), bboxes.size(), &c, &sah);
        if (!h) throw amd::Error(bvh_amd_last_error());
        Bvh<Node> bvh;
        bvh.adopt(h);
        return bvh;
    }
};

// ---- mini_tree_builder.h ------------------------------------------------------------------------------------------------------
// MortonCode: any unsigned integer type. The reference reads it in ONE place, a debug assert on log2_grid_dim (mini_tree_builder.h:171;
// the codes themselves are computed in size_t, :183-186), so it never changes a tree; the device path takes log2_grid_dim 1..10 whatever
// the type (11 would be 2^33 grid bins in the reference too).
template <typename Node, typename MortonCode = uint32_t>
class MiniTreeBuilder {                                       // reference mini_tree_builder.h:24-58 (3D; the grid reads three components)
    static_assert(std::is_unsigned_v<MortonCode>, "MiniTreeBuilder: MortonCode is an unsigned integer type");
    using Scalar = typename Node::Scalar;
    using Vec = bvh::v2::Vec<Scalar, Node::dimension>;
    using BBox = bvh::v2::BBox<Scalar, Node::dimension>;
    static_assert(Node::dimension == 3, "MiniTreeBuilder: 3D only (the reference's own 2D instantiation reads out of bounds)");
public:
    struct Config : TopDownSahBuilder<Node>::Config {
        bool enable_pruning = true;
        Scalar pruning_area_ratio = static_cast<Scalar>(0.01);
        size_t parallel_threshold = 1024;
        size_t log2_grid_dim = 4;
    };
    [[nodiscard]] static Bvh<Node> build(ThreadPool&, std::span<const BBox> bboxes, std::span<const Vec> centers, const Config& config = {}) {
        if (bboxes.size() != centers.size()) throw amd::Error("bvh_amd: bboxes and centers differ in length");
        bvh_amd_minitree_config c;
        c.min_leaf_size = config.min_leaf_size; c.max_leaf_size = config.max_leaf_size; c.enable_pruning = config.enable_pruning ? 1 : 0;
        c.pruning_area_ratio = static_cast<double>(config.pruning_area_ratio); c.parallel_threshold = config.parallel_threshold;
        c.log2_grid_dim = config.log2_grid_dim;
        c.log_cluster_size = config.sah.c_config().log_cluster_size; c.cost_ratio = config.sah.c_config().cost_ratio;
        amd::DeviceArray<BBox> d_bb(bboxes);
        amd::DeviceArray<Vec> d_cc(centers);
        auto* h = amd::Api<Scalar, 3>::build_minitree(d_bb.data(), d_cc.data(), bboxes.size(), &c);
        if (!h) throw amd::Error(bvh_amd_last_error());
        Bvh<Node> bvh;
        bvh.adopt(h);
        return bvh;
    }
};

// ---- reinsertion_optimizer.h ------------------------------------------------------------------------------------------------
template <typename Node>
class ReinsertionOptimizer {
    using Scalar = typename Node::Scalar;
public:
    struct Config {                                           // reference reinsertion_optimizer.h:18-24
        Scalar batch_size_ratio = static_cast<Scalar>(0.05);  // fraction of the nodes re-inserted per iteration
        size_t max_iter_count = 3;
    };
    static void optimize(ThreadPool&, Bvh<Node>& bvh, const Config& config = {}) { optimize(bvh, config); }
    static void optimize(Bvh<Node>& bvh, const Config& config = {}) {
        bvh.push();
        const bvh_amd_optimize_config c{ static_cast<double>(config.batch_size_ratio), config.max_iter_count };
        amd::check(amd::Api<Scalar, Node::dimension>::optimize_config(bvh.device_.get(), &c), "optimize");
        bvh.pull();
    }
};

namespace amd {

// PrecomputedTri of tris[bvh.prim_ids[i]] for every BVH-order slot i, resident in HBM (test/simple_example.cpp:57-65).
template <typename Node>
DeviceArray<PrecomputedTri<typename Node::Scalar>> permuted_triangles(const Bvh<Node>& bvh, std::span<const Tri<typename Node::Scalar, 3>> tris) {
    using T = typename Node::Scalar;
    DeviceArray<Tri<T, 3>> d_tris(tris);
    DeviceArray<PrecomputedTri<T>> out(tris.size());
    check(Api<T, 3>::precompute(d_tris.data(), Api<T, 3>::device_prim_ids(bvh.device()), tris.size(), out.data()), "precompute_tris");
    check(bvh_amd_synchronize(nullptr), "synchronize");
    return out;
}

// Bvh::intersect<IsAnyHit, IsRobust> (reference bvh.h:160-182) for a batch of rays with the closest/any-hit triangle
// leaf loop of test/benchmark.cpp:281-291. Host spans in, host spans out (device-pointer overload below).
template <bool IsAnyHit, bool IsRobust, typename Node>
void intersect_batch(const Bvh<Node>& bvh, const DeviceArray<PrecomputedTri<typename Node::Scalar>>& prims,
                     const DeviceArray<Ray<typename Node::Scalar, 3>>& rays, DeviceArray<Hit<typename Node::Scalar>>& hits) {
    using T = typename Node::Scalar;
    const unsigned flags = (IsAnyHit ? unsigned(BVH_AMD_RAY_ANY_HIT) : 0u) | (IsRobust ? unsigned(BVH_AMD_RAY_ROBUST) : 0u);
    check(Api<T, 3>::trace_tri(bvh.device(), prims.data(), rays.data(), rays.size(), flags, hits.data()), "intersect_rays_tri");
}
template <bool IsAnyHit, bool IsRobust, typename Node>
void intersect_batch(const Bvh<Node>& bvh, const DeviceArray<PrecomputedTri<typename Node::Scalar>>& prims,
                     std::span<const Ray<typename Node::Scalar, 3>> rays, std::span<Hit<typename Node::Scalar>> hits) {
    using T = typename Node::Scalar;
    DeviceArray<Ray<T, 3>> d_rays(rays);
    DeviceArray<Hit<T>> d_hits(rays.size());
    intersect_batch<IsAnyHit, IsRobust>(bvh, prims, d_rays, d_hits);
    d_hits.download(hits);
}
// spheres: hit.t = t0, hit.u = t1 of Sphere::intersect (reference sphere.h:32-49); `spheres` in BVH order
template <bool IsAnyHit, bool IsRobust, typename Node>
void intersect_batch(const Bvh<Node>& bvh, const DeviceArray<Sphere<typename Node::Scalar, Node::dimension>>& spheres,
                     std::span<const Ray<typename Node::Scalar, Node::dimension>> rays, std::span<Hit<typename Node::Scalar>> hits) {
    using T = typename Node::Scalar;
    const unsigned flags = (IsAnyHit ? unsigned(BVH_AMD_RAY_ANY_HIT) : 0u) | (IsRobust ? unsigned(BVH_AMD_RAY_ROBUST) : 0u);
    DeviceArray<Ray<T, Node::dimension>> d_rays(rays);
    DeviceArray<Hit<T>> d_hits(rays.size());
    check(Api<T, Node::dimension>::trace_sphere(bvh.device(), spheres.data(), d_rays.data(), rays.size(), flags, d_hits.data()), "intersect_rays_sphere");
    d_hits.download(hits);
}

// ---- multi-GPU: rays shard, the scene is broadcast once over RCCL inside the library -------------------------------
// A device-resident scene: the BVH (host mirror filled on demand) and its BVH-ordered primitives on ONE device.
template <typename Node, typename Prim>
struct DeviceScene {
    int device = 0;
    Bvh<Node> bvh;
    Prim* prims = nullptr;                                    // device memory of `device`
    size_t prim_count = 0;
    bool owns_prims = false;
    DeviceScene() = default;
    DeviceScene(DeviceScene&& o) noexcept : device(o.device), bvh(std::move(o.bvh)), prims(o.prims), prim_count(o.prim_count), owns_prims(o.owns_prims) { o.prims = nullptr; o.owns_prims = false; }
    DeviceScene& operator=(DeviceScene&&) = delete;
    ~DeviceScene() {
        if (owns_prims && prims) { const int cur = bvh_amd_device_current(); bvh_amd_device_select(device); bvh_amd_device_free(prims); if (cur >= 0) bvh_amd_device_select(cur); }
    }
};

// One process, several GPUs: a copy of (bvh, prims) on every device of `devices` (the BVH's own device among them); entry i
// belongs to devices[i]. Trace shard i with `bvh_amd_device_select(devices[i])` current. (bvhXX_replicate)
template <typename Node, typename Prim>
std::vector<DeviceScene<Node, Prim>> replicate(const Bvh<Node>& bvh, const DeviceArray<Prim>& prims, std::span<const int> devices) {
    using A = Api<typename Node::Scalar, Node::dimension>;
    std::vector<typename A::Handle*> handles(devices.size(), nullptr);
    std::vector<void*> ptrs(devices.size(), nullptr);
    check(A::replicate(bvh.device(), prims.data(), prims.size() * sizeof(Prim), static_cast<int>(devices.size()), devices.data(), handles.data(), ptrs.data()),
          "replicate");
    std::vector<DeviceScene<Node, Prim>> out(devices.size());
    const int cur = bvh_amd_device_current();
    for (size_t i = 0; i < devices.size(); ++i) {
        out[i].device = devices[i];
        out[i].prims = static_cast<Prim*>(ptrs[i]);
        out[i].prim_count = prims.size();
        if (handles[i] == bvh.device()) { out[i].bvh.share(bvh); continue; }                 // the original, not owned twice
        out[i].owns_prims = true;
        bvh_amd_device_select(devices[i]);
        out[i].bvh.adopt(handles[i]);
    }
    if (cur >= 0) bvh_amd_device_select(cur);
    return out;
}

// One process per GPU: every rank calls it with the same `root`; the root passes its scene, the others empty objects.
// `comm` from bvh_amd_comm_create (or bvh_amd_comm_adopt of an ncclComm_t the program already has). (bvhXX_broadcast)
template <typename Node, typename Prim>
DeviceScene<Node, Prim> broadcast(bvh_amd_comm* comm, int root, const Bvh<Node>* bvh, const DeviceArray<Prim>* prims) {
    using A = Api<typename Node::Scalar, Node::dimension>;
    const bool is_root = bvh_amd_comm_rank(comm) == root;
    void* out_prims = nullptr;
    size_t out_bytes = 0;
    typename A::Handle* mine = is_root && bvh ? bvh->device() : nullptr;
    typename A::Handle* h = A::broadcast(comm, root, mine, is_root && prims ? prims->data() : nullptr, is_root && prims ? prims->size() * sizeof(Prim) : 0,
                                         &out_prims, &out_bytes);
    if (!h) throw Error(std::string("broadcast: ") + bvh_amd_last_error());
    DeviceScene<Node, Prim> out;
    out.device = bvh_amd_device_current();
    out.prims = static_cast<Prim*>(out_prims);
    out.prim_count = out_bytes / sizeof(Prim);
    if (h == mine) out.bvh.share(*bvh);
    else { out.owns_prims = true; out.bvh.adopt(h); }
    return out;
}

} // namespace amd
} // namespace bvh::v2

#endif
