// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement of the reference's algorithm (madmann91/bvh v2) for the hot path named by
// BASELINE.json.north_star: top-down SAH builders (binned / sweep / mini-tree), the reinsertion
// optimizer and the stack-based single-ray traverser with the triangle and sphere leaf tests.
// Written from the algorithm's description in SURVEY.md Appendix A with every function citing the
// reference file:line it follows; it shares no code with the reference (flat arrays, no class
// hierarchy, no std::span, one TU). Citations are relative to /root/reference/src/bvh/v2/.
//
// Third-party arithmetic: the reference's results depend on libstdc++'s std::sort / std::partition /
// std::stable_partition / std::partial_sort / heap algorithms (tie behaviour). This
// file calls the same libstdc++ (the image's g++ 11.4) for exactly those, so the dependency is the
// pinned toolchain, not a re-implementation.
//
// Parity status: PINNED. tests/test_oracle_golden.py checks this file bit-for-bit against golden
// vectors produced by the unmodified reference (tests/golden/make_golden.py, via oracle/ref_harness.cpp),
// including the reference's own known answers (test/simple_example.cpp, test/serialize.cpp,
// cornell box node counts); tests/test_oracle_vs_ref.py diffs it against the compiled reference on
// seeded inputs whenever oracle/_ref/libbvh_ref.so is present.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// Build flags are pinned (oracle/Makefile): -O3 -mavx2 -mfma -ffp-contract=off, i.e. every * + - / sqrt
// individually rounded and a fused multiply-add exactly where the reference says fast_mul_add.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

#include "oracle_abi.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Scalars, boxes, nodes
// ---------------------------------------------------------------------------------------------

template <typename T> struct Traits;
template <> struct Traits<float>  { using Index = uint32_t; using Hit = orc_hitf; };
template <> struct Traits<double> { using Index = uint64_t; using Hit = orc_hitd; };

// Dimension of the problem being restated: 3, or 2 for the `2f` / `2d` families (Node<T, 2>, c_api/bvh.cpp:7-10). Storage stays
// three wide (z = 0 everywhere) and every algorithmic loop over axes runs to g_dim, so the 2D entry points below are thin
// wrappers that pad their inputs and narrow their outputs. Test infrastructure: a plain global, set by DimScope for the
// duration of one exported call (worker threads started inside the call see it); calls are not re-entrant across dimensions.
static int g_dim = 3;
struct DimScope { int saved; explicit DimScope(int d) : saved(g_dim) { g_dim = d; } ~DimScope() { g_dim = saved; } };

constexpr unsigned kCountBits = 4;       // node.h:22 PrimCountBits
constexpr size_t   kBins = 8;            // binned_sah_builder.h:19 BinCount (the default; what DefaultBuilder / MiniTreeBuilder instantiate)
constexpr size_t   kMaxBins = 32;        // largest BinCount the explicit binned builder is asked for (orc_set_bin_count)
inline size_t      g_bin_count = kBins;  // BinnedSahBuilder<Node, BinCount>: the template argument of ORC_BUILDER_BINNED builds

// utils.h:41-43: the *second* argument is returned when the first is NaN or when they compare equal.
template <typename T> inline T pick_min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T pick_max(T a, T b) { return a > b ? a : b; }

// utils.h:74-81 under FP_FAST_FMAF (g++ -mfma): a real fused multiply-add.
template <typename T> inline T fused(T a, T b, T c) { return std::fma(a, b, c); }

// utils.h:59-63
template <typename T> inline T guarded_inverse(T x) {
    return std::fabs(x) <= std::numeric_limits<T>::epsilon()
        ? std::copysign(std::numeric_limits<T>::max(), x) : T(1) / x;
}

template <typename T>
struct Box {
    T lo[3], hi[3];
    static Box empty() {                                     // bbox.h:40-44
        Box b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = k < g_dim ? std::numeric_limits<T>::max() : T(0); b.hi[k] = k < g_dim ? -std::numeric_limits<T>::max() : T(0); }
        return b;
    }
    void grow(const Box& o) {                                // bbox.h:22-26: accumulated value is the first operand
        for (int k = 0; k < g_dim; ++k) { lo[k] = pick_min(lo[k], o.lo[k]); hi[k] = pick_max(hi[k], o.hi[k]); }
    }
    void grow_point(const T* p) {                            // bbox.h:18-20
        for (int k = 0; k < g_dim; ++k) { lo[k] = pick_min(lo[k], p[k]); hi[k] = pick_max(hi[k], p[k]); }
    }
    T half_area() const {                                    // bbox.h:32-38
        T d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
        if (g_dim == 2) return d0 + d1;
        return (d0 + d1) * d2 + d0 * d1;
    }
    int largest_axis() const {                               // vec.h:23-33: first maximum wins, NaN never wins
        T d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
        int axis = 0;
        for (int k = 1; k < g_dim; ++k) if (d[k] > d[axis]) axis = k;
        return axis;
    }
};

template <typename T>
struct NodeRec {                                             // node.h:31-37, index.h:74-81
    using Index = typename Traits<T>::Index;
    T bounds[6];                                             // {minx,maxx,miny,maxy,minz,maxz}
    Index index;
    Box<T> box() const {
        Box<T> b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = bounds[2 * k]; b.hi[k] = bounds[2 * k + 1]; }
        return b;
    }
    void set_box(const Box<T>& b) { for (int k = 0; k < 3; ++k) { bounds[2 * k] = b.lo[k]; bounds[2 * k + 1] = b.hi[k]; } }
    size_t first_id() const { return static_cast<size_t>(index >> kCountBits); }
    size_t prim_count() const { return static_cast<size_t>(index & ((Index(1) << kCountBits) - 1)); }
    bool is_leaf() const { return prim_count() != 0; }       // index.h:53
    void set_first_id(size_t id) { index = pack(id, prim_count()); }
    static Index pack(size_t first, size_t count) {
        return (static_cast<Index>(first) << kCountBits) | (static_cast<Index>(count) & ((Index(1) << kCountBits) - 1));
    }
};
static_assert(sizeof(NodeRec<float>) == 28 && sizeof(NodeRec<double>) == 56);

template <typename T>
struct Tree {                                                // bvh.h:17-23
    std::vector<NodeRec<T>> nodes;
    std::vector<size_t> prim_ids;
};

// TopDownSahBuilder::Config (top_down_sah_builder.h:27-40) with its SplitHeuristic (split_heuristic.h:17-38): a range of
// `size` primitives counts as ceil(size / 2^log_cluster) clusters, and a node that stays a leaf saves `cost_ratio`.
inline size_t g_sah_log_cluster = 0;      // set by orc_set_sah(): the SplitHeuristic the following builds use (default {0, 1})
inline double g_sah_cost_ratio = 1.;
struct LeafLimits {
    size_t min_leaf = 1, max_leaf = 8;
    size_t log_cluster = g_sah_log_cluster;
    double cost_ratio = g_sah_cost_ratio;
    template <typename T> T prims(size_t size) const {                      // get_prim_count (:25-27) as the Scalar the costs use
        return static_cast<T>((size + ((size_t{1} << log_cluster) - 1)) >> log_cluster);
    }
    template <typename T> T stay(size_t size) const { return prims<T>(size) - static_cast<T>(cost_ratio); }   // get_non_split_cost (:35-37)
};

// ---------------------------------------------------------------------------------------------
// Top-down driver shared by the binned and sweep builders (top_down_sah_builder.h:74-139)
// ---------------------------------------------------------------------------------------------

template <typename T, typename Splitter>
Tree<T> build_top_down(Splitter& sp, const Box<T>* boxes, size_t n, const LeafLimits& lim) {
    auto range_box = [&](size_t b, size_t e) {               // :133-139
        const std::vector<size_t>& ids = sp.ids();
        Box<T> acc = Box<T>::empty();
        for (size_t i = b; i < e; ++i) acc.grow(boxes[ids[i]]);
        return acc;
    };
    struct Item { size_t node, begin, end; };
    Tree<T> tree;
    tree.nodes.reserve(2 * n / lim.min_leaf);
    tree.nodes.emplace_back();
    tree.nodes[0].set_box(range_box(0, n));
    std::vector<Item> todo{ Item{0, 0, n} };
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        size_t cut = 0;
        if (it.end - it.begin > lim.min_leaf && sp.try_split(tree.nodes[it.node].box(), it.begin, it.end, cut)) {
            size_t child = tree.nodes.size();                // :91-94 children take the next two ids
            tree.nodes[it.node].index = NodeRec<T>::pack(child, 0);
            tree.nodes.resize(child + 2);
            Box<T> ba = range_box(it.begin, cut), bb = range_box(cut, it.end);
            Item ia{child, it.begin, cut}, ib{child + 1, cut, it.end};
            if (ba.half_area() < bb.half_area()) {           // :105-108 SATO: larger half-area first
                std::swap(ba, bb);
                std::swap(ia.begin, ib.begin);
                std::swap(ia.end, ib.end);
            }
            tree.nodes[child].set_box(ba);
            tree.nodes[child + 1].set_box(bb);
            if (ia.end - ia.begin < ib.end - ib.begin)       // :116-120 bigger item pushed first => smaller popped first
                std::swap(ia, ib);
            todo.push_back(ia);
            todo.push_back(ib);
        } else {
            tree.nodes[it.node].index = NodeRec<T>::pack(it.begin, it.end - it.begin);   // :128
        }
    }
    tree.prim_ids = std::move(sp.ids());
    tree.nodes.shrink_to_fit();
    return tree;
}

// ---------------------------------------------------------------------------------------------
// Binned SAH (binned_sah_builder.h)
// ---------------------------------------------------------------------------------------------

template <typename T>
struct BinnedSplitter {
    const Box<T>* boxes;
    const T* centers;                                        // n x 3
    LeafLimits lim;
    std::vector<size_t> order;
    size_t bins;                                             // BinCount (:18)

    BinnedSplitter(const Box<T>* b, const T* c, size_t n, LeafLimits l, size_t bin_count = kBins) : boxes(b), centers(c), lim(l), order(n), bins(bin_count) {
        std::iota(order.begin(), order.end(), size_t{0});    // :77
    }
    std::vector<size_t>& ids() { return order; }

    struct Slot { Box<T> box = Box<T>::empty(); size_t count = 0; };

    size_t median_fallback(int axis, size_t b, size_t e) {  // :118-126
        size_t mid = (b + e + 1) / 2;
        std::partial_sort(order.begin() + b, order.begin() + mid, order.begin() + e,
            [&](size_t i, size_t j) { return centers[3 * i + axis] < centers[3 * j + axis]; });
        return mid;
    }

    bool try_split(const Box<T>& nb, size_t b, size_t e, size_t& cut) {    // :128-156
        const size_t kBins = bins;                           // (shadows the default)
        Slot slots[3][kMaxBins];
        T scale[3], shift[3];
        for (int k = 0; k < g_dim; ++k) {                        // :88-89
            scale[k] = T(kBins) / (nb.hi[k] - nb.lo[k]);
            shift[k] = (-nb.lo[k]) * scale[k];
        }
        for (size_t i = b; i < e; ++i) {                     // :91-98
            size_t p = order[i];
            for (int k = 0; k < g_dim; ++k) {
                T pos = fused(centers[3 * p + k], scale[k], shift[k]);
                size_t s = std::min(kBins - 1, static_cast<size_t>(pick_max(pos, T(0))));
                slots[k][s].box.grow(boxes[p]);
                slots[k][s].count++;
            }
        }
        int wide = nb.largest_axis();
        size_t best_bin = kBins / 2; T best_cost = std::numeric_limits<T>::max(); int best_axis = wide;   // :132-133
        for (int k = 0; k < g_dim; ++k) {                        // :101-116
            Slot acc;
            T right_cost[kMaxBins];
            for (size_t i = kBins - 1; i > 0; --i) {
                acc.box.grow(slots[k][i].box); acc.count += slots[k][i].count;
                right_cost[i] = acc.box.half_area() * lim.template prims<T>(acc.count);
            }
            Slot left;
            for (size_t i = 0; i + 1 < kBins; ++i) {
                left.box.grow(slots[k][i].box); left.count += slots[k][i].count;
                T cost = left.box.half_area() * lim.template prims<T>(left.count) + right_cost[i + 1];
                if (cost < best_cost) { best_bin = i + 1; best_cost = cost; best_axis = k; }
            }
        }
        T stay_cost = nb.half_area() * lim.template stay<T>(e - b);          // split_heuristic.h:35-37
        if (best_cost >= stay_cost) {                        // :138-143
            if (e - b <= lim.max_leaf) return false;
            cut = median_fallback(wide, b, e);
            return true;
        }
        T plane = fused((nb.hi[best_axis] - nb.lo[best_axis]) / T(kBins), static_cast<T>(best_bin), nb.lo[best_axis]);  // :145-148
        size_t idx = std::partition(order.begin() + b, order.begin() + e,
            [&](size_t i) { return centers[3 * i + best_axis] < plane; }) - order.begin();            // :150-151
        if (idx == b || idx == e) idx = median_fallback(wide, b, e);                                  // :152-153
        cut = idx;
        return true;
    }
};

template <typename T>
Tree<T> build_binned(const Box<T>* boxes, const T* centers, size_t n, LeafLimits lim, size_t bin_count = kBins) {
    BinnedSplitter<T> sp(boxes, centers, n, lim, bin_count);
    return build_top_down<T>(sp, boxes, n, lim);
}

// ---------------------------------------------------------------------------------------------
// Sweep SAH (sweep_sah_builder.h)
// ---------------------------------------------------------------------------------------------

template <typename T>
struct SweepSplitter {
    const Box<T>* boxes;
    const T* centers;
    LeafLimits lim;
    std::vector<size_t> sorted[3];
    std::vector<bool> goes_left;
    std::vector<T> suffix_cost;

    SweepSplitter(const Box<T>* b, const T* c, size_t n, LeafLimits l)
        : boxes(b), centers(c), lim(l), goes_left(n), suffix_cost(n)
    {
        for (int k = 0; k < g_dim; ++k) {                        // :57-63
            sorted[k].resize(n);
            std::iota(sorted[k].begin(), sorted[k].end(), size_t{0});
            std::sort(sorted[k].begin(), sorted[k].end(),
                [&, k](size_t i, size_t j) { return centers[3 * i + k] < centers[3 * j + k]; });
        }
    }
    std::vector<size_t>& ids() { return sorted[0]; }         // :66

    void scan_axis(int k, size_t b, size_t e, size_t& best_pos, T& best_cost, int& best_axis) {   // :68-101
        const auto& ord = sorted[k];
        size_t resume = b;
        Box<T> right = Box<T>::empty();
        for (size_t i = e - 1; i > b;) {
            size_t stop = i - std::min(i - b, size_t{32});
            T cost = T(0);
            for (; i > stop; --i) {
                right.grow(boxes[ord[i]]);
                suffix_cost[i] = cost = right.half_area() * lim.template prims<T>(e - i);
            }
            if (cost > best_cost) { resume = i; break; }     // :82-85 chunked early-out
        }
        Box<T> left = Box<T>::empty();
        for (size_t i = b; i < resume; ++i) left.grow(boxes[ord[i]]);
        for (size_t i = resume; i + 1 < e; ++i) {
            left.grow(boxes[ord[i]]);
            T lcost = left.half_area() * lim.template prims<T>(i + 1 - b);
            T cost = lcost + suffix_cost[i + 1];
            if (cost < best_cost) { best_pos = i + 1; best_cost = cost; best_axis = k; }
            else if (lcost > best_cost) break;
        }
    }

    bool try_split(const Box<T>& nb, size_t b, size_t e, size_t& cut) {    // :108-139
        T stay_cost = nb.half_area() * lim.template stay<T>(e - b);
        size_t best_pos = (b + e + 1) / 2; T best_cost = stay_cost; int best_axis = 0;
        for (int k = 0; k < g_dim; ++k) scan_axis(k, b, e, best_pos, best_cost, best_axis);
        if (best_cost >= stay_cost) {
            if (e - b <= lim.max_leaf) return false;
            best_pos = (b + e + 1) / 2;                      // :122-123 median on the widest axis
            best_axis = nb.largest_axis();
        }
        for (size_t i = b; i < best_pos; ++i) goes_left[sorted[best_axis][i]] = true;   // :103-106
        for (size_t i = best_pos; i < e; ++i) goes_left[sorted[best_axis][i]] = false;
        for (int k = 0; k < g_dim; ++k) {                        // :129-136
            if (k == best_axis) continue;
            std::stable_partition(sorted[k].begin() + b, sorted[k].begin() + e,
                [&](size_t i) { return bool(goes_left[i]); });
        }
        cut = best_pos;
        return true;
    }
};

template <typename T>
Tree<T> build_sweep(const Box<T>* boxes, const T* centers, size_t n, LeafLimits lim) {
    SweepSplitter<T> sp(boxes, centers, n, lim);
    return build_top_down<T>(sp, boxes, n, lim);
}

// ---------------------------------------------------------------------------------------------
// Sub-tree extraction (bvh.h:92-122): right-child-first DFS, children allocated at visit time.
// ---------------------------------------------------------------------------------------------

template <typename T>
Tree<T> extract_subtree(const Tree<T>& src, size_t root) {
    Tree<T> out;
    out.nodes.emplace_back();
    std::vector<std::pair<size_t, size_t>> todo{ {root, 0} };
    while (!todo.empty()) {
        auto [s, d] = todo.back();
        todo.pop_back();
        NodeRec<T> node = src.nodes[s];
        if (node.is_leaf()) {
            size_t first = node.first_id(), cnt = node.prim_count();
            node.set_first_id(out.prim_ids.size());
            for (size_t i = 0; i < cnt; ++i) out.prim_ids.push_back(src.prim_ids[first + i]);
            out.nodes[d] = node;
        } else {
            size_t kids = out.nodes.size();
            size_t src_kids = node.first_id();
            node.set_first_id(kids);
            out.nodes[d] = node;
            todo.emplace_back(src_kids, kids);
            todo.emplace_back(src_kids + 1, kids + 1);
            out.nodes.emplace_back();
            out.nodes.emplace_back();
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Mini-tree builder (mini_tree_builder.h). Thread-count independent, so done serially.
// ---------------------------------------------------------------------------------------------

inline size_t spread3(size_t v) {                            // utils.h:104-115 restated for 4-bit inputs and beyond
    size_t r = 0;
    for (unsigned bit = 0; bit < 21; ++bit) r |= ((v >> bit) & size_t{1}) << (3 * bit);
    return r;
}

template <typename T>
struct MiniTreeParams { bool prune; T prune_ratio; size_t par_threshold; unsigned log2_grid = 4; };

template <typename T>
Tree<T> build_mini_trees(const Box<T>* boxes, const T* centers, size_t n, LeafLimits lim, MiniTreeParams<T> mp) {
    // --- build_mini_trees (:160-205)
    Box<T> cbox = Box<T>::empty();
    for (size_t i = 0; i < n; ++i) cbox.grow_point(centers + 3 * i);          // :162-167
    const size_t dim = size_t{1} << mp.log2_grid, cells = size_t{1} << (3 * mp.log2_grid);
    T scale[3], shift[3];
    for (int k = 0; k < 3; ++k) {                                             // :172-173
        scale[k] = static_cast<T>(dim) * guarded_inverse(cbox.hi[k] - cbox.lo[k]);
        shift[k] = (-cbox.lo[k]) * scale[k];
    }
    std::vector<std::vector<size_t>> groups(cells);
    for (size_t i = 0; i < n; ++i) {                                          // :179-185
        size_t g[3];
        for (int k = 0; k < 3; ++k) {
            T p = pick_max(fused(centers[3 * i + k], scale[k], shift[k]), T(0));
            g[k] = std::min(dim - 1, static_cast<size_t>(p));
        }
        size_t code = (spread3(g[0]) | (spread3(g[1]) << 1) | (spread3(g[2]) << 2)) & (cells - 1);
        groups[code].push_back(i);
    }
    if (mp.prune) {                                                           // :84-91 greedy merge on the running size
        for (size_t i = 0; i < groups.size();) {
            size_t j = i + 1;
            for (; j < groups.size() && groups[j].size() + groups[i].size() <= mp.par_threshold; ++j) {
                groups[i].insert(groups[i].end(), groups[j].begin(), groups[j].end());
                groups[j].clear();
            }
            i = j;
        }
    }
    groups.erase(std::remove_if(groups.begin(), groups.end(), [](auto& g) { return g.empty(); }), groups.end());  // :93-96

    std::vector<Tree<T>> trees(groups.size());
    for (size_t t = 0; t < groups.size(); ++t) {                              // BuildTask::run :122-139
        auto& ids = groups[t];
        std::sort(ids.begin(), ids.end());
        std::vector<Box<T>> lb(ids.size());
        std::vector<T> lc(3 * ids.size());
        for (size_t i = 0; i < ids.size(); ++i) {
            lb[i] = boxes[ids[i]];
            for (int k = 0; k < 3; ++k) lc[3 * i + k] = centers[3 * ids[i] + k];
        }
        trees[t] = build_binned<T>(lb.data(), lc.data(), ids.size(), lim);
        for (auto& p : trees[t].prim_ids) p = ids[p];
    }

    // --- prune_mini_trees (:207-247)
    if (mp.prune) {
        T avg = T(0);
        for (auto& tr : trees) avg += tr.nodes[0].box().half_area();          // summed in tree order
        avg /= static_cast<T>(trees.size());
        T threshold = avg * mp.prune_ratio;
        std::vector<std::pair<size_t, size_t>> cuts;
        std::vector<size_t> todo;
        for (size_t t = 0; t < trees.size(); ++t) {
            todo.push_back(0);
            while (!todo.empty()) {
                size_t id = todo.back();
                todo.pop_back();
                const auto& node = trees[t].nodes[id];
                if (node.box().half_area() < threshold || node.is_leaf()) cuts.emplace_back(t, id);
                else { todo.push_back(node.first_id()); todo.push_back(node.first_id() + 1); }   // second child visited first
            }
        }
        std::vector<Tree<T>> pruned(cuts.size());
        for (size_t i = 0; i < cuts.size(); ++i) {
            if (cuts[i].second == 0) pruned[i] = std::move(trees[cuts[i].first]);
            else pruned[i] = extract_subtree(trees[cuts[i].first], cuts[i].second);
        }
        trees = std::move(pruned);
    }

    // --- build_top_bvh (:249-310)
    const size_t m = trees.size();
    std::vector<Box<T>> tb(m);
    std::vector<T> tc(3 * m);
    for (size_t i = 0; i < m; ++i) {
        tb[i] = trees[i].nodes[0].box();
        for (int k = 0; k < 3; ++k) tc[3 * i + k] = (tb[i].hi[k] + tb[i].lo[k]) * T(0.5);   // bbox.h:30
    }
    Tree<T> top = build_sweep<T>(tb.data(), tc.data(), m, LeafLimits{1, 1, lim.log_cluster, lim.cost_ratio});   // :258-260 (the rest of the config is kept)

    std::vector<size_t> node_off(m), prim_off(m);
    size_t node_total = top.nodes.size(), prim_total = 0;
    for (size_t i = 0; i < m; ++i) {                                          // :263-272
        node_off[i] = node_total - 1;
        prim_off[i] = prim_total;
        node_total += trees[i].nodes.size() - 1;
        prim_total += trees[i].prim_ids.size();
    }
    auto rebased = [&](size_t i, const NodeRec<T>& src) {                     // :275-279
        NodeRec<T> d = src;
        d.set_first_id(src.first_id() + (src.is_leaf() ? prim_off[i] : node_off[i]));
        return d;
    };
    for (auto& node : top.nodes) {                                            // :282-288
        if (!node.is_leaf()) continue;
        size_t t = top.prim_ids[node.first_id()];
        node = rebased(t, trees[t].nodes[0]);
    }
    top.nodes.resize(node_total);
    top.prim_ids.resize(prim_total);
    for (size_t i = 0; i < m; ++i) {                                          // :292-307
        for (size_t j = 1; j < trees[i].nodes.size(); ++j) top.nodes[node_off[i] + j] = rebased(i, trees[i].nodes[j]);
        std::copy(trees[i].prim_ids.begin(), trees[i].prim_ids.end(), top.prim_ids.begin() + prim_off[i]);
    }
    return top;
}

// ---------------------------------------------------------------------------------------------
// Reinsertion optimizer (reinsertion_optimizer.h)
// ---------------------------------------------------------------------------------------------

template <typename T>
struct Reinserter {
    Tree<T>& tree;
    std::vector<size_t> parent;

    struct Cand { size_t node; T cost; };
    struct Move { size_t from = 0, to = 0; T gain = T(0); };

    static size_t sibling(size_t id) { return (id % 2 == 1) ? id + 1 : id - 1; }       // bvh.h:34-39
    static size_t left_of(size_t id) { return (id % 2 == 1) ? id : id - 1; }           // bvh.h:43-45

    explicit Reinserter(Tree<T>& t) : tree(t), parent(t.nodes.size()) {                // :72-86
        parent[0] = 0;
        for (size_t i = 0; i < tree.nodes.size(); ++i) {
            if (tree.nodes[i].is_leaf()) continue;
            parent[tree.nodes[i].first_id()] = i;
            parent[tree.nodes[i].first_id() + 1] = i;
        }
    }

    std::vector<Cand> pick_candidates(size_t target) {                                 // :88-105
        auto worse = [](const Cand& a, const Cand& b) { return a.cost > b.cost; };     // std::greater on cost
        const size_t head = std::min(tree.nodes.size(), target + 1);
        std::vector<Cand> heap;
        for (size_t i = 1; i < head; ++i) heap.push_back(Cand{i, tree.nodes[i].box().half_area()});
        std::make_heap(heap.begin(), heap.end(), worse);
        for (size_t i = head; i < tree.nodes.size(); ++i) {
            T cost = tree.nodes[i].box().half_area();
            if (heap.front().cost < cost) {
                std::pop_heap(heap.begin(), heap.end(), worse);
                heap.back() = Cand{i, cost};
                std::push_heap(heap.begin(), heap.end(), worse);
            }
        }
        return heap;
    }

    Move search(size_t id) const {                                                     // :107-188
        Move best;
        best.from = id;
        const Box<T> self = tree.nodes[id].box();
        const T self_area = self.half_area();
        T gain_so_far = tree.nodes[parent[id]].box().half_area();
        size_t sib = sibling(id);
        Box<T> pivot_box = tree.nodes[sib].box();
        const size_t first_parent = parent[id];
        size_t pivot = first_parent;
        std::vector<std::pair<T, size_t>> todo;
        do {
            todo.emplace_back(gain_so_far, sib);
            while (!todo.empty()) {
                auto [bound, dst] = todo.back();
                todo.pop_back();
                if (bound - self_area <= best.gain) continue;
                const NodeRec<T>& dn = tree.nodes[dst];
                Box<T> merged = dn.box();
                merged.grow(self);
                T gain = bound - merged.half_area();
                if (gain > best.gain) { best.to = dst; best.gain = gain; }
                if (!dn.is_leaf()) {
                    T child_bound = gain + dn.box().half_area();
                    todo.emplace_back(child_bound, dn.first_id());
                    todo.emplace_back(child_bound, dn.first_id() + 1);
                }
            }
            if (pivot != first_parent) {                                               // :177-180
                pivot_box.grow(tree.nodes[sib].box());
                gain_so_far += tree.nodes[pivot].box().half_area() - pivot_box.half_area();
            }
            sib = sibling(pivot);
            pivot = parent[pivot];
        } while (pivot != 0);
        if (best.to == sibling(best.from) || best.to == parent[best.from]) best = Move{};   // :184-186
        return best;
    }

    void refit_upwards(size_t i) {                                                     // :215-225
        do {
            NodeRec<T>& node = tree.nodes[i];
            if (!node.is_leaf()) {
                Box<T> b = tree.nodes[node.first_id()].box();
                b.grow(tree.nodes[node.first_id() + 1].box());
                node.set_box(b);
            }
            i = parent[i];
        } while (i != 0);
    }

    void apply(size_t from, size_t to) {                                               // :190-213
        size_t sib = sibling(from), par = parent[from];
        NodeRec<T> sib_node = tree.nodes[sib], dst_node = tree.nodes[to];
        tree.nodes[to].index = NodeRec<T>::pack(left_of(from), 0);
        tree.nodes[sib] = dst_node;
        tree.nodes[par] = sib_node;
        if (!dst_node.is_leaf()) { parent[dst_node.first_id()] = sib; parent[dst_node.first_id() + 1] = sib; }
        if (!sib_node.is_leaf()) { parent[sib_node.first_id()] = par; parent[sib_node.first_id() + 1] = par; }
        parent[sib] = to;
        parent[from] = to;
        refit_upwards(to);
        refit_upwards(par);
    }

    void run(T batch_ratio = T(0.05), size_t iterations = 3) {                         // :237-267, Config :19-25
        size_t batch = std::max(size_t{1}, static_cast<size_t>(static_cast<T>(tree.nodes.size()) * batch_ratio));
        std::vector<bool> touched(tree.nodes.size());
        for (size_t it = 0; it < iterations; ++it) {
            auto cands = pick_candidates(batch);
            std::fill(touched.begin(), touched.end(), false);
            std::vector<Move> moves(cands.size());
            for (size_t i = 0; i < cands.size(); ++i) moves[i] = search(cands[i].node);
            moves.erase(std::remove_if(moves.begin(), moves.end(), [](const Move& m) { return m.gain <= 0; }), moves.end());
            std::sort(moves.begin(), moves.end(), [](const Move& a, const Move& b) { return a.gain > b.gain; });
            for (const Move& m : moves) {
                size_t hot[5] = { m.to, m.from, sibling(m.from), parent[m.to], parent[m.from] };   // :227-234
                bool clash = false;
                for (size_t h : hot) clash = clash || touched[h];
                if (clash) continue;
                for (size_t h : hot) touched[h] = true;
                apply(m.from, m.to);
            }
        }
    }
};

template <typename T>
void optimize_tree(Tree<T>& t) { Reinserter<T>(t).run(); }
template <typename T>
void optimize_tree(Tree<T>& t, double batch_ratio, size_t iterations) { Reinserter<T>(t).run(static_cast<T>(batch_ratio), iterations); }

// ---------------------------------------------------------------------------------------------
// DefaultBuilder dispatch (default_builder.h:33-73)
// ---------------------------------------------------------------------------------------------

template <typename T>
Tree<T> build_dispatch(const T* bboxes6, const T* centers, size_t n, int builder, int quality,
                       LeafLimits lim, size_t par_threshold)
{
    std::vector<Box<T>> boxes(n);
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = bboxes6[6 * i + k]; boxes[i].hi[k] = bboxes6[6 * i + 3 + k]; }
    auto serial = [&] {                                                                // :49-62
        if (quality == ORC_QUALITY_LOW) return build_binned<T>(boxes.data(), centers, n, lim);
        Tree<T> t = build_sweep<T>(boxes.data(), centers, n, lim);
        if (quality == ORC_QUALITY_HIGH) optimize_tree(t);
        return t;
    };
    switch (builder) {
    case ORC_BUILDER_BINNED: return build_binned<T>(boxes.data(), centers, n, lim, g_bin_count);
    case ORC_BUILDER_SWEEP:  return build_sweep<T>(boxes.data(), centers, n, lim);
    case ORC_BUILDER_DEFAULT_PARALLEL: {                                               // :33-46
        if (n < par_threshold) return serial();
        MiniTreeParams<T> mp;                                                          // :65-73
        mp.prune = quality != ORC_QUALITY_LOW;
        mp.prune_ratio = quality == ORC_QUALITY_HIGH ? T(0.01) : T(0.1);
        mp.par_threshold = par_threshold;
        Tree<T> t = build_mini_trees<T>(boxes.data(), centers, n, lim, mp);
        if (quality == ORC_QUALITY_HIGH) optimize_tree(t);
        return t;
    }
    default: return serial();
    }
}

// ---------------------------------------------------------------------------------------------
// Bottom-up refit (bvh.h:185-218)
// ---------------------------------------------------------------------------------------------

template <typename T>
void refit_tree(Tree<T>& t) {
    const size_t n = t.nodes.size();
    std::vector<size_t> parent(n, 0);
    for (size_t i = 0; i < n; ++i) {
        if (t.nodes[i].is_leaf()) continue;
        parent[t.nodes[i].first_id()] = i;
        parent[t.nodes[i].first_id() + 1] = i;
    }
    std::vector<bool> done(n, false);
    for (size_t i = n; i-- > 0;) {
        if (!t.nodes[i].is_leaf()) continue;
        done[i] = true;
        for (size_t j = parent[i];; j = parent[j]) {
            NodeRec<T>& node = t.nodes[j];
            if (done[j] || !done[node.first_id()] || !done[node.first_id() + 1]) break;
            Box<T> b = t.nodes[node.first_id()].box();
            b.grow(t.nodes[node.first_id() + 1].box());
            node.set_box(b);
            done[j] = true;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Primitive preparation (tri.h:24-25, :35-37; sphere.h:24-27)
// ---------------------------------------------------------------------------------------------

template <typename T>
void tri_bounds_and_centers(const T* t9, size_t n, T* bb, T* cc) {
    for (size_t i = 0; i < n; ++i) {
        const T* p = t9 + 9 * i;
        for (int k = 0; k < 3; ++k) {
            T lo = p[k], hi = p[k];                          // BBox(p0).extend(p1).extend(p2)
            lo = pick_min(lo, p[3 + k]); hi = pick_max(hi, p[3 + k]);
            lo = pick_min(lo, p[6 + k]); hi = pick_max(hi, p[6 + k]);
            bb[6 * i + k] = lo; bb[6 * i + 3 + k] = hi;
            cc[3 * i + k] = (p[k] + p[3 + k] + p[6 + k]) * static_cast<T>(1. / 3.);
        }
    }
}

template <typename T>
void tri_precompute(const T* t9, const uint64_t* perm, size_t n, T* out) {
    for (size_t i = 0; i < n; ++i) {
        const T* p = t9 + 9 * (perm ? perm[i] : i);
        T* o = out + 12 * i;
        T e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { o[k] = p[k]; e1[k] = p[k] - p[3 + k]; e2[k] = p[6 + k] - p[k]; }
        for (int k = 0; k < 3; ++k) { o[3 + k] = e1[k]; o[6 + k] = e2[k]; }
        o[9]  = e1[1] * e2[2] - e1[2] * e2[1];               // vec.h:103-108
        o[10] = e1[2] * e2[0] - e1[0] * e2[2];
        o[11] = e1[0] * e2[1] - e1[1] * e2[0];
    }
}

template <typename T>
void sphere_bounds(const T* s4, size_t n, T* bb, T* cc) {
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            bb[6 * i + k] = s4[4 * i + k] - s4[4 * i + 3];
            bb[6 * i + 3 + k] = s4[4 * i + k] + s4[4 * i + 3];
            cc[3 * i + k] = s4[4 * i + k];
        }
}

// ---------------------------------------------------------------------------------------------
// Single-ray traversal (bvh.h:125-182, node.h:59-117, ray.h:29-48) and leaf tests
// ---------------------------------------------------------------------------------------------

template <typename T> inline T dot3(const T* a, const T* b) {   // vec.h:98-100: ((0 + a0 b0) + a1 b1) [+ a2 b2]
    if (g_dim == 2) return (T(0) + a[0] * b[0]) + a[1] * b[1];
    return ((T(0) + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2];
}

template <typename T>
struct RayState {
    T org[3], dir[3], tmin, tmax;
};

// tri.h:56-74
template <typename T>
inline bool hit_triangle(const T* tri /*p0,e1,e2,n*/, const RayState<T>& r, T& t, T& u, T& v) {
    const T* p0 = tri; const T* e1 = tri + 3; const T* e2 = tri + 6; const T* nrm = tri + 9;
    T c[3] = { p0[0] - r.org[0], p0[1] - r.org[1], p0[2] - r.org[2] };
    T rr[3] = { r.dir[1] * c[2] - r.dir[2] * c[1], r.dir[2] * c[0] - r.dir[0] * c[2], r.dir[0] * c[1] - r.dir[1] * c[0] };
    T inv_det = T(1) / dot3(nrm, r.dir);
    u = dot3(rr, e2) * inv_det;
    v = dot3(rr, e1) * inv_det;
    T w = T(1) - u - v;
    const T tol = -std::numeric_limits<T>::epsilon();
    if (u >= tol && v >= tol && w >= tol) {
        t = dot3(nrm, c) * inv_det;
        if (t >= r.tmin && t <= r.tmax) return true;
    }
    return false;
}

// sphere.h:32-49 (AssumeNormalized = false)
template <typename T>
inline bool hit_sphere(const T* s /*center,radius*/, const RayState<T>& r, T& t0, T& t1) {
    T oc[3] = { r.org[0] - s[0], r.org[1] - s[1], r.org[2] - s[2] };
    T a = dot3(r.dir, r.dir);
    T b = T(2) * dot3(r.dir, oc);
    T c = dot3(oc, oc) - s[3] * s[3];
    T delta = b * b - T(4) * a * c;
    if (delta >= 0) {
        T inv = -T(0.5) / a;
        T root = std::sqrt(delta);
        t0 = pick_max((b + root) * inv, r.tmin);
        t1 = pick_min((b - root) * inv, r.tmax);
        if (t0 <= t1) return true;
    }
    return false;
}

template <typename T> inline T nudge_ulps(T x, unsigned k);  // utils.h:47-55
template <> inline float nudge_ulps(float x, unsigned k) {
    if (!std::isfinite(x)) return x;
    uint32_t u; std::memcpy(&u, &x, 4); u += k; std::memcpy(&x, &u, 4); return x;
}
template <> inline double nudge_ulps(double x, unsigned k) {
    if (!std::isfinite(x)) return x;
    uint64_t u; std::memcpy(&u, &x, 8); u += k; std::memcpy(&x, &u, 8); return x;
}

enum class Leaf { Triangle, Sphere };

template <typename T, bool Any, bool Robust, Leaf Kind>
void trace_one(const Tree<T>& tree, const T* prims, RayState<T>& ray, typename Traits<T>::Hit& hit, uint64_t* cnt) {
    using Index = typename Traits<T>::Index;
    T inv[3], inv_org[3], inv_pad[3];
    unsigned oct[3];
    for (int k = 0; k < g_dim; ++k) {                            // bvh.h:162-165
        inv[k] = Robust ? T(1) / ray.dir[k] : guarded_inverse(ray.dir[k]);
        inv_org[k] = (-inv[k]) * ray.org[k];
        inv_pad[k] = nudge_ulps(inv[k], 2);
        oct[k] = std::signbit(ray.dir[k]) ? 1u : 0u;        // ray.h:36-43
    }
    auto slab = [&](const NodeRec<T>& node, T& t0, T& t1) { // node.h:68-88, :105-117
        t0 = ray.tmin; t1 = ray.tmax;
        for (int k = 0; k < g_dim; ++k) {
            T near_b = node.bounds[2 * k + oct[k]], far_b = node.bounds[2 * k + 1 - oct[k]];
            T a, b;
            if (Robust) { a = (near_b - ray.org[k]) * inv[k]; b = (far_b - ray.org[k]) * inv_pad[k]; }
            else        { a = fused(near_b, inv[k], inv_org[k]); b = fused(far_b, inv[k], inv_org[k]); }
            t0 = pick_max(a, t0);
            t1 = pick_min(b, t1);
        }
    };
    // SmallStack<Index, 64> (stack.h:11-30) in the reference's examples and C API; this restatement grows instead
    // (GrowingStack, stack.h:34-46: same push/pop order, no capacity) so that it can also check trees deeper than 64
    Index small[64];
    std::vector<Index> grown;
    Index* stack = small;
    size_t cap = 64;
    auto reserve = [&](size_t need) {
        if (need <= cap) return;
        std::vector<Index> bigger(2 * need);
        std::copy(stack, stack + cap, bigger.begin());
        grown.swap(bigger); stack = grown.data(); cap = grown.size();
    };
    unsigned sp = 0;
    stack[sp++] = tree.nodes[0].index;                       // traversal starts at root.index: the root box is never tested
    const Index count_mask = (Index(1) << kCountBits) - 1;
    while (sp) {
    next_entry:
        if (!sp) break;
        Index top = stack[--sp];
        while ((top & count_mask) == 0) {                    // bvh.h:132-150
            const NodeRec<T>& l = tree.nodes[top >> kCountBits];
            const NodeRec<T>& r = tree.nodes[(top >> kCountBits) + 1];
            cnt[0]++;
            T l0, l1, r0, r1;
            slab(l, l0, l1);
            slab(r, r0, r1);
            bool hl = l0 <= l1, hr = r0 <= r1;
            if (hl) {
                Index near_i = l.index;
                if (hr) {
                    Index far_i = r.index;
                    if (!Any && l0 > r0) std::swap(near_i, far_i);
                    reserve(sp + 1);
                    stack[sp++] = far_i;
                }
                top = near_i;
            } else if (hr) top = r.index;
            else goto next_entry;
        }
        size_t first = static_cast<size_t>(top >> kCountBits), count = static_cast<size_t>(top & count_mask);
        cnt[2]++;
        for (size_t i = first; i < first + count; ++i) {     // leaf lambda of test/benchmark.cpp:281-291
            cnt[1]++;
            if (Kind == Leaf::Triangle) {
                T t, u, v;
                if (hit_triangle(prims + 12 * i, ray, t, u, v)) {
                    ray.tmax = t; hit.t = t; hit.u = u; hit.v = v; hit.prim = static_cast<uint32_t>(i);
                }
            } else {
                T t0, t1;
                if (hit_sphere(prims + 4 * i, ray, t0, t1)) {
                    ray.tmax = t0; hit.t = t0; hit.u = t1; hit.v = 0; hit.prim = static_cast<uint32_t>(i);
                }
            }
        }
        if (Any && hit.prim != ORC_INVALID) return;          // bvh.h:153-155 with "return prim_id != invalid"
    }
}

template <typename T, bool Any, bool Robust, Leaf Kind>
void trace_range(const Tree<T>& tree, const T* prims, const T* rays8, size_t b, size_t e,
                 typename Traits<T>::Hit* out, uint64_t* cnt)
{
    for (size_t r = b; r < e; ++r) {
        const T* q = rays8 + 8 * r;
        RayState<T> ray{ {q[0], q[1], q[2]}, {q[3], q[4], q[5]}, q[6], q[7] };
        typename Traits<T>::Hit h;
        std::memset(&h, 0, sizeof(h));
        h.prim = ORC_INVALID; h.t = q[7]; h.u = 0; h.v = 0;
        trace_one<T, Any, Robust, Kind>(tree, prims, ray, h, cnt);
        out[r] = h;
    }
}

template <typename T, Leaf Kind>
void trace_all(const Tree<T>& tree, const T* prims, const T* rays8, size_t n, int any, int robust, int threads,
               typename Traits<T>::Hit* out, uint64_t* counters)
{
    int nt = (threads <= 1 || n < 1024) ? 1 : threads;
    std::vector<uint64_t> cnt(3 * nt, 0);
    auto work = [&](size_t b, size_t e, int slot) {
        uint64_t* c = cnt.data() + 3 * slot;
        if (any) { if (robust) trace_range<T, true, true, Kind>(tree, prims, rays8, b, e, out, c);
                   else        trace_range<T, true, false, Kind>(tree, prims, rays8, b, e, out, c); }
        else     { if (robust) trace_range<T, false, true, Kind>(tree, prims, rays8, b, e, out, c);
                   else        trace_range<T, false, false, Kind>(tree, prims, rays8, b, e, out, c); }
    };
    if (nt == 1) work(0, n, 0);
    else {
        std::vector<std::thread> pool;
        size_t chunk = (n + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            size_t b = std::min(n, chunk * t), e = std::min(n, b + chunk);
            pool.emplace_back(work, b, e, t);
        }
        for (auto& th : pool) th.join();
    }
    if (counters) {
        counters[0] = counters[1] = counters[2] = 0;
        for (int t = 0; t < nt; ++t) for (int k = 0; k < 3; ++k) counters[k] += cnt[3 * t + k];
    }
}

// bvh.h:221-229 + node.h:90-94: [node_count][prim_count] nodes... prim ids..., all in Index::Type
template <typename T>
size_t serialize_tree(const Tree<T>& t, uint8_t* out, size_t cap) {
    using Index = typename Traits<T>::Index;
    size_t need = 2 * sizeof(Index) + t.nodes.size() * sizeof(NodeRec<T>) + t.prim_ids.size() * sizeof(Index);
    if (!out || cap < need) return need;
    Index hdr[2] = { static_cast<Index>(t.nodes.size()), static_cast<Index>(t.prim_ids.size()) };
    std::memcpy(out, hdr, sizeof(hdr));
    out += sizeof(hdr);
    std::memcpy(out, t.nodes.data(), t.nodes.size() * sizeof(NodeRec<T>));
    out += t.nodes.size() * sizeof(NodeRec<T>);
    for (size_t p : t.prim_ids) { Index v = static_cast<Index>(p); std::memcpy(out, &v, sizeof(v)); out += sizeof(v); }
    return need;
}

} // namespace

extern "C" {

// BinnedSahBuilder's BinCount (binned_sah_builder.h:18) for the ORC_BUILDER_BINNED builds that follow; 2 .. 32
ORC_EXPORT int orc_set_bin_count(size_t bin_count) {
    if (bin_count < 2 || bin_count > kMaxBins) return -1;
    g_bin_count = bin_count;
    return 0;
}
ORC_EXPORT void orc_set_sah(size_t log_cluster_size, double cost_ratio) {
    g_sah_log_cluster = log_cluster_size; g_sah_cost_ratio = cost_ratio;
}

#define ORC_IMPL(T, S)                                                                                  \
    ORC_EXPORT void* orc_build##S(const T* bboxes, const T* centers, size_t n, int builder, int quality, \
        size_t min_leaf, size_t max_leaf, size_t par_threshold, int /*threads*/) {                      \
        if (!n) return nullptr;                                                                          \
        return new Tree<T>(build_dispatch<T>(bboxes, centers, n, builder, quality, LeafLimits{min_leaf, max_leaf}, par_threshold)); } \
    /* MiniTreeBuilder::build(pool, bboxes, centers, config) itself (mini_tree_builder.h:29-58) */                      \
    ORC_EXPORT void* orc_build_minitree##S(const T* bboxes, const T* centers, size_t n, size_t min_leaf, size_t max_leaf, \
        int enable_pruning, double pruning_area_ratio, size_t par_threshold, int /*threads*/, size_t log2_grid_dim) { \
        if (!n) return nullptr;                                                                          \
        std::vector<Box<T>> boxes(n);                                                                    \
        for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = bboxes[6 * i + k]; boxes[i].hi[k] = bboxes[6 * i + 3 + k]; } \
        MiniTreeParams<T> mp; mp.prune = enable_pruning != 0; mp.prune_ratio = static_cast<T>(pruning_area_ratio); mp.par_threshold = par_threshold; \
        mp.log2_grid = static_cast<unsigned>(log2_grid_dim);                                             \
        return new Tree<T>(build_mini_trees<T>(boxes.data(), centers, n, LeafLimits{min_leaf, max_leaf}, mp)); } \
    ORC_EXPORT void orc_destroy##S(void* h) { delete static_cast<Tree<T>*>(h); }                        \
    ORC_EXPORT size_t orc_node_count##S(const void* h) { return static_cast<const Tree<T>*>(h)->nodes.size(); } \
    ORC_EXPORT size_t orc_prim_count##S(const void* h) { return static_cast<const Tree<T>*>(h)->prim_ids.size(); } \
    ORC_EXPORT void orc_get_nodes##S(const void* h, void* out) {                                        \
        auto t = static_cast<const Tree<T>*>(h); std::memcpy(out, t->nodes.data(), t->nodes.size() * sizeof(NodeRec<T>)); } \
    ORC_EXPORT void orc_get_prim_ids##S(const void* h, uint64_t* out) {                                 \
        auto t = static_cast<const Tree<T>*>(h); for (size_t i = 0; i < t->prim_ids.size(); ++i) out[i] = t->prim_ids[i]; } \
    ORC_EXPORT void* orc_from_arrays##S(const void* nodes, size_t nn, const uint64_t* ids, size_t np) { \
        auto t = new Tree<T>; t->nodes.resize(nn); std::memcpy(t->nodes.data(), nodes, nn * sizeof(NodeRec<T>)); \
        t->prim_ids.assign(ids, ids + np); return t; }                                                  \
    ORC_EXPORT size_t orc_serialize##S(const void* h, uint8_t* out, size_t cap) {                       \
        return serialize_tree<T>(*static_cast<const Tree<T>*>(h), out, cap); }                          \
    ORC_EXPORT void orc_optimize##S(void* h, int /*threads*/) { optimize_tree<T>(*static_cast<Tree<T>*>(h)); } \
    ORC_EXPORT void orc_optimize_config##S(void* h, int /*threads*/, double ratio, size_t iterations) {  \
        optimize_tree<T>(*static_cast<Tree<T>*>(h), ratio, iterations); }                                  \
    ORC_EXPORT void* orc_extract##S(const void* h, size_t root) {                                        \
        return new Tree<T>(extract_subtree<T>(*static_cast<const Tree<T>*>(h), root)); }                 \
    ORC_EXPORT void orc_refit##S(void* h) { refit_tree<T>(*static_cast<Tree<T>*>(h)); }                 \
    ORC_EXPORT void orc_prep_tris##S(const T* t9, size_t n, T* bb, T* cc) { tri_bounds_and_centers<T>(t9, n, bb, cc); } \
    ORC_EXPORT void orc_precompute_tris##S(const T* t9, const uint64_t* perm, size_t n, T* out12) {     \
        tri_precompute<T>(t9, perm, n, out12); }                                                        \
    ORC_EXPORT void orc_sphere_bboxes##S(const T* s4, size_t n, T* bb, T* cc) { sphere_bounds<T>(s4, n, bb, cc); } \
    ORC_EXPORT void orc_intersect_tri##S(const void* h, const T* tris12, const T* rays8, size_t nrays,  \
        int any, int robust, int threads, Traits<T>::Hit* out, uint64_t* counters) {                    \
        trace_all<T, Leaf::Triangle>(*static_cast<const Tree<T>*>(h), tris12, rays8, nrays, any, robust, threads, out, counters); } \
    ORC_EXPORT void orc_intersect_sphere##S(const void* h, const T* sph4, const T* rays8, size_t nrays, \
        int any, int robust, int threads, Traits<T>::Hit* out, uint64_t* counters) {                    \
        trace_all<T, Leaf::Sphere>(*static_cast<const Tree<T>*>(h), sph4, rays8, nrays, any, robust, threads, out, counters); }

ORC_IMPL(float, 3f)
ORC_IMPL(double, 3d)

} // extern "C"

// ---- the 2D families: pad to the three-wide internal storage, run with g_dim = 2, narrow the results -----------------------
namespace {

template <typename T> struct Node2Rec { T bounds[4]; typename Traits<T>::Index index; };   // Node<T, 2> (node.h:31-37)
static_assert(sizeof(Node2Rec<float>) == 20 && sizeof(Node2Rec<double>) == 40);

template <typename T>
void narrow_nodes(const Tree<T>& t, Node2Rec<T>* out) {
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        for (int k = 0; k < 4; ++k) out[i].bounds[k] = t.nodes[i].bounds[k];
        out[i].index = t.nodes[i].index;
    }
}

template <typename T>
Tree<T>* build2(const T* bboxes4, const T* centers2, size_t n, int builder, int quality, size_t min_leaf, size_t max_leaf, size_t par_threshold) {
    // DefaultBuilder(pool) above parallel_threshold runs the mini-tree builder, which reads p[2] of a 2D vector
    // (mini_tree_builder.h:183): undefined behaviour in the reference, refused here
    if (!n || (builder == ORC_BUILDER_DEFAULT_PARALLEL && n >= par_threshold)) return nullptr;
    std::vector<T> bb(6 * n), cc(3 * n);
    for (size_t i = 0; i < n; ++i) {                          // {min[2], max[2]} -> {min[3], max[3]} with z = 0
        bb[6 * i + 0] = bboxes4[4 * i + 0]; bb[6 * i + 1] = bboxes4[4 * i + 1]; bb[6 * i + 2] = T(0);
        bb[6 * i + 3] = bboxes4[4 * i + 2]; bb[6 * i + 4] = bboxes4[4 * i + 3]; bb[6 * i + 5] = T(0);
        cc[3 * i + 0] = centers2[2 * i + 0]; cc[3 * i + 1] = centers2[2 * i + 1]; cc[3 * i + 2] = T(0);
    }
    DimScope dim(2);
    return new Tree<T>(build_dispatch<T>(bb.data(), cc.data(), n, builder, quality, LeafLimits{min_leaf, max_leaf}, par_threshold));
}

template <typename T>
size_t serialize2(const Tree<T>& t, uint8_t* out, size_t cap) {           // bvh.h:221-229 with 20/40-byte nodes
    using Index = typename Traits<T>::Index;
    const size_t need = 2 * sizeof(Index) + t.nodes.size() * sizeof(Node2Rec<T>) + t.prim_ids.size() * sizeof(Index);
    if (!out || cap < need) return need;
    Index hdr[2] = { static_cast<Index>(t.nodes.size()), static_cast<Index>(t.prim_ids.size()) };
    std::memcpy(out, hdr, sizeof(hdr));
    out += sizeof(hdr);
    std::vector<Node2Rec<T>> narrow(t.nodes.size());
    narrow_nodes(t, narrow.data());
    std::memcpy(out, narrow.data(), narrow.size() * sizeof(Node2Rec<T>));
    out += narrow.size() * sizeof(Node2Rec<T>);
    for (size_t p : t.prim_ids) { Index v = static_cast<Index>(p); std::memcpy(out, &v, sizeof(v)); out += sizeof(v); }
    return need;
}

template <typename T>
void trace2(const Tree<T>& tree, const T* circles3, const T* rays6, size_t n, int any, int robust, int threads,
            typename Traits<T>::Hit* out, uint64_t* counters) {
    size_t np = tree.prim_ids.size();
    std::vector<T> sph(4 * np), rays(8 * n);
    for (size_t i = 0; i < np; ++i) { sph[4 * i] = circles3[3 * i]; sph[4 * i + 1] = circles3[3 * i + 1]; sph[4 * i + 2] = T(0); sph[4 * i + 3] = circles3[3 * i + 2]; }
    for (size_t i = 0; i < n; ++i) {                           // {org[2], dir[2], tmin, tmax}
        const T* q = rays6 + 6 * i; T* r = rays.data() + 8 * i;
        r[0] = q[0]; r[1] = q[1]; r[2] = T(0); r[3] = q[2]; r[4] = q[3]; r[5] = T(0); r[6] = q[4]; r[7] = q[5];
    }
    DimScope dim(2);
    trace_all<T, Leaf::Sphere>(tree, sph.data(), rays.data(), n, any, robust, threads, out, counters);
}

} // namespace

extern "C" {

#define ORC_IMPL2(T, S)                                                                                 \
    ORC_EXPORT void* orc_build##S(const T* bboxes4, const T* centers2, size_t n, int builder, int quality, \
        size_t min_leaf, size_t max_leaf, size_t par_threshold, int /*threads*/) {                      \
        return build2<T>(bboxes4, centers2, n, builder, quality, min_leaf, max_leaf, par_threshold); }  \
    ORC_EXPORT void orc_destroy##S(void* h) { delete static_cast<Tree<T>*>(h); }                        \
    ORC_EXPORT size_t orc_node_count##S(const void* h) { return static_cast<const Tree<T>*>(h)->nodes.size(); } \
    ORC_EXPORT size_t orc_prim_count##S(const void* h) { return static_cast<const Tree<T>*>(h)->prim_ids.size(); } \
    ORC_EXPORT void orc_get_nodes##S(const void* h, void* out) {                                        \
        narrow_nodes<T>(*static_cast<const Tree<T>*>(h), static_cast<Node2Rec<T>*>(out)); }             \
    ORC_EXPORT void orc_get_prim_ids##S(const void* h, uint64_t* out) {                                 \
        auto t = static_cast<const Tree<T>*>(h); for (size_t i = 0; i < t->prim_ids.size(); ++i) out[i] = t->prim_ids[i]; } \
    ORC_EXPORT void* orc_from_arrays##S(const void* nodes, size_t nn, const uint64_t* ids, size_t np) { \
        auto t = new Tree<T>; t->nodes.resize(nn);                                                       \
        auto src = static_cast<const Node2Rec<T>*>(nodes);                                               \
        for (size_t i = 0; i < nn; ++i) {                                                                \
            for (int k = 0; k < 4; ++k) t->nodes[i].bounds[k] = src[i].bounds[k];                        \
            t->nodes[i].bounds[4] = t->nodes[i].bounds[5] = T(0); t->nodes[i].index = src[i].index; }    \
        t->prim_ids.assign(ids, ids + np); return t; }                                                  \
    ORC_EXPORT size_t orc_serialize##S(const void* h, uint8_t* out, size_t cap) {                       \
        return serialize2<T>(*static_cast<const Tree<T>*>(h), out, cap); }                              \
    ORC_EXPORT void orc_optimize##S(void* h, int /*threads*/) { DimScope dim(2); optimize_tree<T>(*static_cast<Tree<T>*>(h)); } \
    ORC_EXPORT void orc_optimize_config##S(void* h, int /*threads*/, double ratio, size_t iterations) {  \
        DimScope dim(2); optimize_tree<T>(*static_cast<Tree<T>*>(h), ratio, iterations); }                 \
    ORC_EXPORT void* orc_extract##S(const void* h, size_t root) {                                        \
        DimScope dim(2); return new Tree<T>(extract_subtree<T>(*static_cast<const Tree<T>*>(h), root)); } \
    ORC_EXPORT void orc_refit##S(void* h) { DimScope dim(2); refit_tree<T>(*static_cast<Tree<T>*>(h)); } \
    ORC_EXPORT void orc_sphere_bboxes##S(const T* c3, size_t n, T* bb4, T* cc2) {                       \
        for (size_t i = 0; i < n; ++i) for (int k = 0; k < 2; ++k) {      /* sphere.h:24-27 */           \
            bb4[4 * i + k] = c3[3 * i + k] - c3[3 * i + 2]; bb4[4 * i + 2 + k] = c3[3 * i + k] + c3[3 * i + 2]; \
            cc2[2 * i + k] = c3[3 * i + k]; } }                                                          \
    ORC_EXPORT void orc_intersect_sphere##S(const void* h, const T* circles3, const T* rays6, size_t nrays, \
        int any, int robust, int threads, Traits<T>::Hit* out, uint64_t* counters) {                    \
        trace2<T>(*static_cast<const Tree<T>*>(h), circles3, rays6, nrays, any, robust, threads, out, counters); }

ORC_IMPL2(float, 2f)
ORC_IMPL2(double, 2d)

// the order-defining primitive behind SweepSahBuilder's constructor (sweep_sah_builder.h:57-63)
ORC_EXPORT void orc_std_sort_ids3f(const float* keys, size_t n, uint32_t* out) {
    std::vector<size_t> ids(n);
    std::iota(ids.begin(), ids.end(), size_t{0});
    std::sort(ids.begin(), ids.end(), [&](size_t i, size_t j) { return keys[i] < keys[j]; });
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<uint32_t>(ids[i]);
}
ORC_EXPORT void orc_std_sort_ids3d(const double* keys, size_t n, uint32_t* out) {
    std::vector<size_t> ids(n);
    std::iota(ids.begin(), ids.end(), size_t{0});
    std::sort(ids.begin(), ids.end(), [&](size_t i, size_t j) { return keys[i] < keys[j]; });
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<uint32_t>(ids[i]);
}

ORC_EXPORT int orc_hardware_threads(void) { return static_cast<int>(std::thread::hardware_concurrency()); }

} // extern "C"
