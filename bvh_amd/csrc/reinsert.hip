// K10 — ReinsertionOptimizer (reinsertion_optimizer.h:19-267) on gfx950, bit-exact with the reference.
//
// Per iteration (3 by default, batch = 5 % of the nodes):
//   find_candidates (:88-105)  top-k nodes by half-area through a min-heap whose ARRAY LAYOUT (not just its set) is the
//                              input order of everything downstream, so libstdc++'s make_heap / pop_heap / push_heap are
//                              replayed exactly (stl_heap.h:134-148, :223-266, :339-362). make_heap is
//                              level-parallel (one launch per heap level); the replacement loop is inherently sequential
//                              and runs on ONE wavefront, organised around what bounds a lone wave (instruction count and
//                              memory latency): the top 14 heap levels live in LDS, five levels are resolved per LDS round
//                              trip, the ancestor chain of the last position lives in registers, and the HBM part of each
//                              sift-down is deferred and executed 64 at a time (see the comments further down).
//   find_reinsertion (:107-188) one lane per candidate: the reference's branch-and-bound walk with its explicit stack
//                              (same push order, same strict comparisons), read-only on the tree.
//   remove_if + std::sort by gain, descending (:254-256): stable compaction (scan) + the exact std::sort emulation on the
//                              negated gains (sort_emul.hip) — equal gains are common and their order decides conflicts.
//   greedy apply (:258-265)    sequential by nature (live parents, touched flags, refits up both paths): one lane.
//
// FAST PATH (default). The heap layout only matters through (1) which of several equal-cost nodes straddling the
// top-k threshold survive and (2) the order std::sort leaves equal gains in, and (2) only matters when two equal-gain
// reinsertions that are both still applicable touch a common node (non-conflicting reinsertions commute: disjoint
// slots, and every refit path ends consistent bottom-up). So each iteration first runs WITHOUT the replay:
// top-k by a radix sort of the costs, gains sorted by a radix sort, and k_apply verifies, group of equal gains by group,
// that the applicable members have pairwise disjoint conflict sets. The moment (1) or (2) could make the layout
// matter the iteration is rolled back (nodes restored from a device copy) and redone with the exact replay
// (k_heap_select + std::sort emulation). Results are bit-identical to the reference either way. In practice the layout
// matters often: a node X whose best target is its "uncle" Y usually comes with Y -> X at exactly the same gain, and the two
// share nodes, so on the 1M-triangle soup two of three iterations need the replay (~250 ms each; a fast iteration ~4 ms).
// BVH_AMD_REINSERT=exact forces the replay, BVH_AMD_REINSERT_DEBUG=1 reports why an iteration fell back.
#include "build_common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bvh_amd {

using namespace bld;

namespace {

constexpr int kSearchStack = 96;
template <typename T> struct HeapLevels;                    // complete heap levels kept in LDS (cost + id per entry)
template <> struct HeapLevels<float>  { static constexpr int v = 14; };       // 16383 entries of 8 bytes
template <> struct HeapLevels<double> { static constexpr int v = 13; };       //  8191 entries of 16 bytes
template <typename T> struct HeapCap { static constexpr uint32_t v = (1u << HeapLevels<T>::v) - 1; };

struct Move { uint32_t from, to; };
struct ReScalars { uint32_t n_moves, error, ambiguous, replacements; };   // replacements: pop_heap + push_heap pairs of the exact replays (statistics)

template <typename T> __device__ inline T ha6(const T* b, int dim) {         // bounds = {minx,maxx,miny,maxy,minz,maxz}
    const T d0 = b[1] - b[0], d1 = b[3] - b[2], d2 = b[5] - b[4];
    if (dim == 2) return d0 + d1;                            // bbox.h:36 (2D nodes run three wide with z = 0, see build_common.h)
    return (d0 + d1) * d2 + d0 * d1;
}
template <typename T> __device__ inline bool is_leaf(const HostNode<T>& n) { return (n.index & kCountMask) != 0; }
template <typename T> __device__ inline uint32_t first_of(const HostNode<T>& n) { return static_cast<uint32_t>(n.index >> kCountBits); }
__device__ inline uint32_t sibling_of(uint32_t id) { return (id & 1u) ? id + 1 : id - 1; }      // bvh.h:34-39
__device__ inline uint32_t left_of(uint32_t id) { return (id & 1u) ? id : id - 1; }              // bvh.h:43-45

// compute_parents (:72-86) + half-area of every node
template <typename T>
__global__ void __launch_bounds__(256) k_parents_costs(const HostNode<T>* nodes, uint32_t n, uint32_t* parent, T* cost, int with_parents, int dim) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HostNode<T> nd = nodes[i];
    cost[i] = ha6(nd.bounds, dim);
    if (with_parents) {
        if (i == 0) parent[0] = 0;
        if (!is_leaf(nd)) { parent[first_of(nd)] = i; parent[first_of(nd) + 1] = i; }
    }
}

// ---- find_candidates: libstdc++ heap algorithms with comp = std::greater on cost ---------------------------------
// The whole wavefront works on ONE heap operation at a time, with wave-uniform control flow:
//   * __adjust_heap (stl_heap.h:223-248) = walk the min-child path from the hole to a leaf, shift that path up one
//     level, then __push_heap the saved value back up along the same path. The path is discovered FIVE LEVELS PER
//     MEMORY ROUND-TRIP: the 62 descendants of the current hole are loaded by 62 lanes at once and the five child
//     choices are made with readlane on registers. Path entry j stays in lane j, so the final layout
//     (entries 1..m* move up, the value lands at path[m*], m* = deepest entry with !(cost > value)) is ONE parallel
//     write round.
//   * __push_heap (stl_heap.h:134-148) from the last position: the ancestor chain of a FIXED position is loaded by
//     ~17 lanes at once; the shift is again one parallel write round.
// Heap entries {cost, id}: the first 14 (float) / 13 (double) levels in LDS, the rest in HBM.
template <typename T> struct Ent { T cost; uint32_t id; };

// The two halves of the heap are addressed through explicitly address-space-qualified pointers: a generic pointer that
// may be either makes the compiler emit FLAT loads/stores, and flat accesses force an s_waitcnt on every outstanding HBM
// store before the next load (measured: 2.4 us per replacement, most of it waiting for store acknowledgements).
template <typename T>
struct WaveHeap {
    using LdsEnt = __attribute__((address_space(3))) Ent<T>;
    using GlobEnt = __attribute__((address_space(1))) Ent<T>;
    LdsEnt* lds; GlobEnt* glob; uint32_t cap;
    int cap_level;                                            // deepest level that lies completely in LDS
    __device__ Ent<T> get_lds(uint32_t i) const { Ent<T> e; e.cost = lds[i].cost; e.id = lds[i].id; return e; }
    __device__ Ent<T> get_glob(uint32_t i) const { Ent<T> e; e.cost = glob[i].cost; e.id = glob[i].id; return e; }
    __device__ void set_lds(uint32_t i, Ent<T> e) { lds[i].cost = e.cost; lds[i].id = e.id; }
    __device__ void set_glob(uint32_t i, Ent<T> e) { glob[i].cost = e.cost; glob[i].id = e.id; }
    __device__ Ent<T> get(uint32_t i) const { return i < cap ? get_lds(i) : get_glob(i); }
    __device__ void set(uint32_t i, Ent<T> e) { if (i < cap) set_lds(i, e); else set_glob(i, e); }
};

// Stores of this wave before its later loads. Memory instructions of ONE wave are processed in program order by the
// LDS and by the vector memory pipeline (that is what makes store-then-load through a may-alias pointer work for a single
// lane); lanes of the wave hand entries to each other through memory, so only the COMPILER must be kept from reordering
// across the hand-off. No s_waitcnt is needed, which keeps store round-trips off the critical path.
__device__ inline void heap_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline double lane_value(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), l), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ inline uint32_t lane_value(uint32_t v, int l) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l)); }

// lane i <- lane i + 1 (DPP wave_shl:1, a VALU move; __shfl_down would go through the LDS crossbar and its latency)
__device__ inline uint32_t from_next_lane(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x130, 0xf, 0xf, false));
}
__device__ inline float from_next_lane(float v) { return __uint_as_float(from_next_lane(__float_as_uint(v))); }
__device__ inline double from_next_lane(double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    const unsigned long long r = (static_cast<unsigned long long>(from_next_lane(static_cast<uint32_t>(b >> 32))) << 32) | from_next_lane(static_cast<uint32_t>(b));
    return __longlong_as_double(static_cast<long long>(r));
}

// The ancestor chain of the last heap position k-1 (the only positions __push_heap(first, k-1, 0, w) can touch, and where
// std::pop_heap takes its value from). Lane j holds ancestor j: position (k >> j) - 1, lane 0 = k-1 itself. Chain entries
// beyond the LDS part of the heap live in REGISTERS for the whole replacement loop (their HBM copies go stale), so that
// neither the value a pop starts from nor the push that follows waits on HBM.
template <typename T>
struct Chain {
    Ent<T> reg;                                               // authoritative when `bottom`
    uint32_t pos; bool on, bottom;
    uint32_t k; int top_level;                                // level of position k-1
    __device__ void init(const WaveHeap<T>& h, uint32_t k_, int lane) {
        k = k_;
        top_level = 31 - __clz(static_cast<int>(k_));
        on = lane <= top_level;
        pos = on ? (k_ >> lane) - 1 : 0;
        bottom = on && pos >= h.cap;
        reg.cost = T(0); reg.id = 0;
    }
    __device__ void load(const WaveHeap<T>& h) { if (bottom) reg = h.get_glob(pos); }
    __device__ void flush(WaveHeap<T>& h) { if (bottom) h.set_glob(pos, reg); heap_sync(); }
    __device__ bool holds(uint32_t p, int level) const { return level <= top_level && (k >> (top_level - level)) - 1 == p; }   // wave-uniform
};

// ---- __adjust_heap(first, hole0, len, value) with comp(a, b) = a.cost > b.cost, by the whole wavefront ------------------
// All lanes call these with identical (wave-uniform) arguments. The min-child path is discovered FIVE LEVELS PER MEMORY
// ROUND-TRIP: the 62 descendants of the current hole are loaded by 62 lanes at once and the child choices are made with
// readlane on registers. Path entry j is kept by lane j, so the result (entries 1..m* move up one level, the value lands
// at path[m*], m* = deepest entry with !(cost > value)) is one parallel write round.
//
// Because a heap is sorted along every root-to-leaf path, {j : !(cost_j > value)} is a prefix of the path: the upper part
// of a path can be finalised before the lower part is known. The replacement loop uses that to split each pop into
//   top    levels that live in LDS: done at once; if every entry met so far moves up, the hole arrives at the last LDS
//          level and is handed over as a task (position, value); the LDS entry there is marked as an open hole;
//   bottom the rest of the path in HBM. Tasks in different subtrees are independent, so they are DEFERRED: up to 64 of them
//          are collected (one per lane) and then sifted down all at once, one lane per task, so the HBM latency of the
//          deep levels is paid once per batch instead of once per replacement. A later pop that is about to read an open
//          hole (it shows up among the entries of the last LDS level it loads) resolves the batch first.
template <typename T>
struct PathState {
    uint32_t my_pos; Ent<T> my_ent;                           // lane j: position / old entry of path entry j (lane 0: the hole)
    uint32_t cur; int depth;
};

// (a round never straddles the LDS/HBM boundary: it either ends at the last LDS level or starts at or below it, so the
// two kinds of loads never target the same registers under complementary lane masks, which would serialise them)
template <typename T>
__device__ inline Ent<T> load_round(const WaveHeap<T>& h, uint32_t cur, int steps, uint32_t len, bool lds_round, int lane) {
    const int d = 31 - __clz(lane + 1);                       // lanes 1..62: descendants of `cur` in BFS order
    const uint32_t node = ((cur + 1) << d) - 1 + (static_cast<uint32_t>(lane + 1) - (1u << d));
    Ent<T> e; e.cost = T(0); e.id = 0;
    if (lane >= 1 && lane <= 62 && d <= steps && node < len) { if (lds_round) e = h.get_lds(node); else e = h.get_glob(node); }
    return e;
}

template <typename T>
__device__ inline void walk_round(const Ent<T>& e, int steps, uint32_t limit, PathState<T>& p, int lane) {
    int r = 0;                                                // BFS index of the current hole inside the loaded subtree
    for (int step = 0; step < steps && p.cur < limit; ++step) {
        const int c1 = 2 * r + 1, c2 = 2 * r + 2;
        const T k1 = lane_value(e.cost, c1), k2 = lane_value(e.cost, c2);
        const bool left = k2 > k1;                            // comp(second, second - 1): take the left child
        r = left ? c1 : c2;
        p.cur = left ? 2 * p.cur + 1 : 2 * p.cur + 2;
        ++p.depth;
        const T pc = lane_value(e.cost, r);
        const uint32_t pi = lane_value(e.id, r);
        if (lane == p.depth) { p.my_pos = p.cur; p.my_ent.cost = pc; p.my_ent.id = pi; }
    }
}

// entries 1..m* move up; the value lands at path[m*] unless `defer_value` and m* == depth (the hole is handed on). Returns m* == depth.
template <bool LdsOnly, typename T>
__device__ inline bool write_path(WaveHeap<T>& h, const PathState<T>& p, Ent<T> value, bool defer_value, int lane) {
    const bool stays = lane >= 1 && lane <= p.depth && !(p.my_ent.cost > value.cost);
    const uint64_t mask = __ballot(stays);
    const int mstar = mask ? 63 - __clzll(static_cast<long long>(mask)) : 0;
    const uint32_t up_pos = __shfl_up(p.my_pos, 1);           // position of path entry j - 1
    const bool open = mstar == p.depth;
    if constexpr (LdsOnly) {                                  // (the top part never leaves LDS: no HBM instruction at all)
        if (lane >= 1 && lane <= mstar) h.set_lds(up_pos, p.my_ent);
        if (lane == mstar && !(defer_value && open)) h.set_lds(p.my_pos, value);
    } else {
        if (lane >= 1 && lane <= mstar) h.set(up_pos, p.my_ent);
        if (lane == mstar && !(defer_value && open)) h.set(p.my_pos, value);
    }
    heap_sync();
    return open;
}

// The complete operation from `hole0` by the whole wave (used for the rare chain-crossing paths).
// Returns true when the chain registers were flushed because the path ran along the chain into the HBM part of the heap:
// the caller reloads them afterwards.
template <typename T>
__device__ bool wave_adjust_heap(WaveHeap<T>& h, Chain<T>& chain, uint32_t hole0, uint32_t len, Ent<T> value, int lane) {
    PathState<T> p;
    p.my_pos = hole0; p.my_ent = value; p.cur = hole0; p.depth = 0;
    bool flushed = false;
    const uint32_t limit = (len - 1) / 2;                     // nodes below `limit` have two children
    while (p.cur < limit) {
        // a round stops at the last LDS-resident level so that only one round per operation touches HBM
        const int level = 31 - __clz(p.cur + 1);
        const int steps = (level < h.cap_level && level + 5 > h.cap_level) ? h.cap_level - level : 5;   // >= 1
        if (!flushed && level + steps > h.cap_level && chain.holds(p.cur, level)) { chain.flush(h); flushed = true; }
        const Ent<T> e = load_round(h, p.cur, steps, len, level < h.cap_level, lane);
        walk_round(e, steps, limit, p, lane);
    }
    if ((len & 1u) == 0 && p.cur == (len - 2) / 2) {          // a last node with a single (left) child
        const uint32_t child = 2 * p.cur + 1;
        Ent<T> e;
        if (child < h.cap) e = h.get_lds(child); else e = h.get_glob(child);     // wave-uniform
        p.cur = child;
        ++p.depth;
        if (lane == p.depth) { p.my_pos = p.cur; p.my_ent = e; }
    }
    write_path<false>(h, p, value, false, lane);
    return flushed;
}

// Node costs fetched per HBM round trip of the replacement loop. (Round 5 measured a register prefetch of the NEXT chunk, 2048 costs = 32
// loads per lane in flight: the stream's share of wave A fell from 16 % to 12 %, but the scalar registers the loads' bounds handling took
// pushed spills into the replacement itself, +7 % there: 2 % slower in all, profiles/r05_heap_pipe_profile_prefetch.txt. Not kept.)
constexpr uint32_t kStreamChunk = 512;
constexpr uint32_t kOpenHole = 0xffffffffu;                // id of an LDS entry whose content is owed by a deferred task

// Top part of a pop from the root: walks the LDS levels only. Returns true when the hole has to continue below the last LDS
// level; then `hand_pos` is where it stands (all path entries above it have moved up, `value` is not stored yet).
template <typename T, typename Resolve>
__device__ bool top_adjust(WaveHeap<T>& h, uint32_t len, Ent<T> value, int lane, uint32_t& hand_pos, Resolve resolve_tasks) {
    PathState<T> p;
    p.my_pos = 0; p.my_ent = value; p.cur = 0; p.depth = 0;
    const uint32_t limit = (len - 1) / 2;
    int level = 0;
    while (p.cur < limit && level < h.cap_level) {
        const int steps = min(5, h.cap_level - level);
        Ent<T> e = load_round(h, p.cur, steps, len, true, lane);
        if (level + steps == h.cap_level && __ballot(e.id == kOpenHole)) {      // an entry of the last LDS level is still owed
            resolve_tasks();
            e = load_round(h, p.cur, steps, len, true, lane);
        }
        walk_round(e, steps, limit, p, lane);
        level = 31 - __clz(p.cur + 1);
    }
    const bool single = (len & 1u) == 0 && p.cur == (len - 2) / 2;
    const bool below = level >= h.cap_level && (p.cur < limit || single);
    if (!below && single) {                                   // the heap ends inside the LDS levels
        const uint32_t child = 2 * p.cur + 1;
        const Ent<T> e = h.get_lds(child);
        p.cur = child;
        ++p.depth;
        if (lane == p.depth) { p.my_pos = p.cur; p.my_ent = e; }
    }
    const bool open = write_path<true>(h, p, value, below, lane);
    hand_pos = p.cur;
    return below && open;
}

// The same top part for heaps that extend below the LDS levels (len >= cap: every node above the last LDS level has both
// children), written for instruction count, which is what a single wave is bound by (the generic version above spends
// ~1200 clocks per five-level round on readlane chains). BFS node `lane` of the subtree under the hole loads BOTH its
// children with one LDS instruction and chooses between them, all nodes at once; the path is then five dependent readlanes
// ("pointer jumping" through the choices), and every path node itself stores its chosen child's entry (or the value,
// at the first node whose chosen child is greater) into its own position, so no entry travels between lanes.
// (the rounds are unrolled at compile time: 5 + 5 + 3 levels for float, 5 + 5 + 2 for double when the heap extends below LDS;
//  kCapLevel = the number of leading levels whose nodes all have two children, for heaps that end inside LDS)
template <typename T, int Level, int kCapLevel, typename Resolve>
__device__ inline bool fast_top_round(WaveHeap<T>& h, Ent<T> value, int lane, uint32_t cur, int d, uint32_t in_level,
                                      uint32_t& hand_pos, T& root_cost, Resolve& resolve_tasks) {
    constexpr int kSteps = kCapLevel - Level < 5 ? kCapLevel - Level : 5;
    const uint32_t node = ((cur + 1) << d) - 1 + in_level;                // heap position of BFS node `lane`
    const bool inner = lane < (1 << kSteps) - 1;
    Ent<T> c1{}, c2{};
    if (inner) { c1 = h.get_lds(2 * node + 1); c2 = h.get_lds(2 * node + 2); }
    if constexpr (Level + kSteps == kCapLevel) {
        if (__ballot(inner && (c1.id == kOpenHole || c2.id == kOpenHole))) {
            resolve_tasks();                                               // an entry of the last LDS level is still owed
            if (inner) { c1 = h.get_lds(2 * node + 1); c2 = h.get_lds(2 * node + 2); }
        }
    }
    const bool left = c2.cost > c1.cost;                                   // comp(second, second - 1): take the left child
    const Ent<T> chosen = left ? c1 : c2;
    const uint32_t chosen_pos = left ? 2 * node + 1 : 2 * node + 2;
    const int next = left ? 2 * lane + 1 : 2 * lane + 2;
    int r = 0;
    uint64_t path = 1;                                                     // BFS indices of the path nodes of this round
#pragma unroll
    for (int sidx = 1; sidx < kSteps; ++sidx) { r = __builtin_amdgcn_readlane(next, r); path |= uint64_t{1} << r; }
    const bool on_path = (path >> lane) & 1u;
    const uint64_t greater = __ballot(on_path && chosen.cost > value.cost);
    if constexpr (Level == 0) root_cost = (greater & 1u) ? value.cost : lane_value(chosen.cost, 0);
    if (greater) {                                                         // the value lands inside this round
        const int landing = __ffsll(static_cast<long long>(greater)) - 1;  // shallowest such node (BFS order)
        if (on_path && lane < landing) h.set_lds(node, chosen);
        if (lane == landing) h.set_lds(node, value);
        heap_sync();
        return false;
    }
    if (on_path) h.set_lds(node, chosen);
    heap_sync();
    const uint32_t below = lane_value(chosen_pos, r);
    if constexpr (Level + kSteps < kCapLevel)
        return fast_top_round<T, Level + kSteps, kCapLevel>(h, value, lane, below, d, in_level, hand_pos, root_cost, resolve_tasks);
    hand_pos = below;
    return true;
}

// `full_levels` (wave-uniform, 1 .. HeapLevels - 1): every node above that level has both children inside the heap
template <bool BelowLds, typename T, typename Resolve>
__device__ inline bool fast_top_adjust(WaveHeap<T>& h, int full_levels, Ent<T> value, int lane, uint32_t& hand_pos, T& root_cost, Resolve resolve_tasks) {
    const int d = 31 - __clz(lane + 1);
    const uint32_t in_level = static_cast<uint32_t>(lane + 1) - (1u << d);
    if constexpr (BelowLds)                                   // the heap reaches below LDS (large scenes): one instantiation, no dispatch
        return fast_top_round<T, 0, HeapLevels<T>::v - 1>(h, value, lane, 0u, d, in_level, hand_pos, root_cost, resolve_tasks);
#define BVH_FAST_TOP(F) case F: if constexpr (F <= HeapLevels<T>::v - 1) return fast_top_round<T, 0, F>(h, value, lane, 0u, d, in_level, hand_pos, root_cost, resolve_tasks); break;
    switch (full_levels) {
        BVH_FAST_TOP(1) BVH_FAST_TOP(2) BVH_FAST_TOP(3) BVH_FAST_TOP(4) BVH_FAST_TOP(5) BVH_FAST_TOP(6) BVH_FAST_TOP(7)
        BVH_FAST_TOP(8) BVH_FAST_TOP(9) BVH_FAST_TOP(10) BVH_FAST_TOP(11) BVH_FAST_TOP(12) BVH_FAST_TOP(13)
        default: break;
    }
#undef BVH_FAST_TOP
    hand_pos = 0;                                             // (unreachable for valid arguments)
    return true;
}

// __push_heap(first, hole = k - 1, top = 0, w) on the chain: ancestors greater than w move down one place, w lands above them
// Returns how many ancestors moved (== chain.top_level when w became the new root).
// (the chain's LDS part as the push will see it; the two-wave loop issues this read ahead of its token hand-off, whose release
//  store waits for the LDS anyway — nobody but the calling wave writes chain positions, so the values stay current)
template <typename T>
__device__ inline Ent<T> chain_fetch(const WaveHeap<T>& h, const Chain<T>& chain, int lane) {
    Ent<T> e = chain.reg;
    if (chain.on && !chain.bottom && lane >= 1) e = h.get_lds(chain.pos);
    return e;
}
template <typename T>
__device__ int wave_push_chain(WaveHeap<T>& h, Chain<T>& chain, Ent<T> w, int lane, Ent<T> e) {
    const bool valid = chain.on && lane >= 1;
    const uint64_t above = __ballot(valid && e.cost > w.cost) >> 1;       // bit j-1: ancestor j is moved down
    const int moves = above == ~uint64_t{0} ? 64 : __ffsll(static_cast<long long>(~above)) - 1;   // leading run of ones
    Ent<T> below; below.cost = from_next_lane(e.cost); below.id = from_next_lane(e.id);             // ancestor j + 1
    if (chain.on && lane <= moves) {
        const Ent<T> nv = lane == moves ? w : below;
        if (chain.bottom) chain.reg = nv; else h.set_lds(chain.pos, nv);
    }
    heap_sync();
    return moves;
}
template <typename T>
__device__ int wave_push_chain(WaveHeap<T>& h, Chain<T>& chain, Ent<T> w, int lane) { return wave_push_chain(h, chain, w, lane, chain_fetch(h, chain, lane)); }

// literal __adjust_heap by ONE lane on the HBM copy (level-parallel make_heap below)
template <typename T>
__device__ void lane_adjust_heap(Ent<T>* a, uint32_t hole, uint32_t len, Ent<T> value) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (a[child].cost > a[child - 1].cost) --child;
        a[hole] = a[child];
        hole = child;
    }
    if ((len & 1u) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
    }
    while (hole > top) {
        const uint32_t parent = (hole - 1) / 2;
        if (!(a[parent].cost > value.cost)) break;
        a[hole] = a[parent];
        hole = parent;
    }
    a[hole] = value;
}

// the same by one lane on the LDS + HBM heap of the replacement loop (deferred bottom tasks: the hole is an LDS entry,
// everything below it is in HBM)
template <typename T>
__device__ void lane_adjust_heap(WaveHeap<T>& h, uint32_t hole, uint32_t len, Ent<T> value) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        const Ent<T> r = h.get_glob(child), l = h.get_glob(child - 1);
        Ent<T> take = r;
        if (r.cost > l.cost) { --child; take = l; }
        h.set(hole, take);
        hole = child;
    }
    if ((len & 1u) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.set(hole, h.get_glob(child - 1));
        hole = child - 1;
    }
    while (hole > top) {
        const uint32_t parent = (hole - 1) / 2;
        const Ent<T> pe = h.get(parent);
        if (!(pe.cost > value.cost)) break;
        h.set(hole, pe);
        hole = parent;
    }
    h.set(hole, value);
}

template <typename T>
__global__ void __launch_bounds__(256) k_heap_fill(const T* cost, uint32_t k, Ent<T>* glob) {      // candidates 1 .. k  (:93-94)
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < k) { Ent<T> e; e.cost = cost[j + 1]; e.id = j + 1; glob[j] = e; }
}

// __make_heap (stl_heap.h:339-362) calls __adjust_heap for parent = (k-2)/2 down to 0. Calls on one heap level work in
// disjoint subtrees and only depend on the calls below them, so one launch per level (deepest first), one lane per
// node, reproduces the sequential result.
template <typename T>
__global__ void __launch_bounds__(64) k_make_heap_level(Ent<T>* glob, uint32_t k, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < count) lane_adjust_heap(glob, first + i, k, glob[first + i]);
}

// The replacement loop (:96-103). `glob` holds the finished make_heap; the top of the heap is staged into LDS.
// BelowLds: the heap has more entries than fit in LDS (k - 1 >= HeapCap); the two cases are separate kernels so that the
// large-scene loop carries exactly one instantiation of the top phase.
template <typename T, bool BelowLds>
__global__ void __launch_bounds__(64) k_heap_select(const T* cost, uint32_t n_nodes, uint32_t target, Ent<T>* glob, uint32_t* out_ids, ReScalars* sc) {
    extern __shared__ unsigned char heap_lds[];
    WaveHeap<T> h;
    h.lds = (typename WaveHeap<T>::LdsEnt*)heap_lds;
    h.glob = (typename WaveHeap<T>::GlobEnt*)glob;
    auto* stage = (__attribute__((address_space(3))) T*)(heap_lds + size_t{HeapCap<T>::v} * sizeof(Ent<T>));   // costs of the current chunk
    h.cap = HeapCap<T>::v;
    h.cap_level = HeapLevels<T>::v - 1;
    const int lane = threadIdx.x;
    const uint32_t head = min(n_nodes, target + 1);
    const uint32_t k = head - 1;
    if (k == 0) return;
    for (uint32_t j = lane; j < min(k, h.cap); j += 64) h.set_lds(j, h.get_glob(j));
    heap_sync();
    Chain<T> chain;
    chain.init(h, k, lane);
    chain.load(h);
    // deferred bottom tasks: lane i keeps task i
    uint32_t n_tasks = 0; uint32_t task_pos = 0; Ent<T> task_value{};
    auto resolve_tasks = [&]() {
        if (n_tasks) {
            if (static_cast<uint32_t>(lane) < n_tasks) lane_adjust_heap(h, task_pos, k - 1, task_value);
            heap_sync();
            n_tasks = 0;
        }
    };
    T root_cost = h.lds[0].cost;                              // cost of the heap minimum, tracked in a register
    uint32_t n_replaced = 0;                                  // (wave-uniform; reported for the bench's build.high statistics)
    // pops work on positions [0, k - 1): every node above level `full_levels` has both children there
    constexpr bool below_lds = BelowLds;
    int full_levels = below_lds ? h.cap_level : 0;
    if (!below_lds && k >= 4) { full_levels = (31 - __clz(static_cast<int>(k))) - 1; if (full_levels > h.cap_level) full_levels = h.cap_level; }
    const bool last_in_regs = k - 1 >= h.cap;                 // position k-1 (chain lane 0) is register-resident
    for (uint32_t chunk = head; chunk < n_nodes; chunk += kStreamChunk) {   // :96-103, kStreamChunk costs per HBM round trip
        bool any = false;
#pragma unroll
        for (uint32_t q = 0; q < kStreamChunk / 64; ++q) {
            const uint32_t i = chunk + q * 64 + lane;
            const T c_hbm = i < n_nodes ? cost[i] : T(0);
            any = any || (i < n_nodes && root_cost < c_hbm);  // the heap minimum only grows: a failed test stays failed
            stage[q * 64 + lane] = c_hbm;                     // through LDS: the loop below then holds no register that depends
        }                                                     // on an HBM load (the compiler would wait on ALL of them)
        if (!__ballot(any)) continue;
        heap_sync();
        for (uint32_t q = 0; q < kStreamChunk / 64; ++q) {
            const uint32_t base = chunk + q * 64;
            if (base >= n_nodes) break;
            const T c = stage[q * 64 + lane];
            uint64_t mask = __ballot(base + lane < n_nodes && root_cost < c);
            while (mask) {
                const int j = __ffsll(static_cast<long long>(mask)) - 1;
                mask &= mask - 1;
                const T cj = lane_value(c, j);
                if (root_cost < cj) {
                    ++n_replaced;
                    if (k > 1) {                                  // std::pop_heap: the value at k-1 sinks in from the root
                        Ent<T> v;
                        if (last_in_regs) { v.cost = lane_value(chain.reg.cost, 0); v.id = lane_value(chain.reg.id, 0); }
                        else v = h.get_lds(k - 1);
                        uint32_t hand = 0;
                        bool handed;
                        if (full_levels >= 1) handed = fast_top_adjust<BelowLds>(h, full_levels, v, lane, hand, root_cost, resolve_tasks);
                        else { handed = top_adjust(h, k - 1, v, lane, hand, resolve_tasks); root_cost = h.lds[0].cost; }
                        if (handed) {
                            if (!below_lds || chain.holds(hand, h.cap_level)) {
                                // the rest of the path at once, by the wave: a heap that ends inside LDS (its last, incomplete
                                // levels), or a path that follows the chain into HBM (rare)
                                if (wave_adjust_heap(h, chain, hand, k - 1, v, lane)) chain.load(h);
                            } else {
                                if (static_cast<uint32_t>(lane) == n_tasks) { task_pos = hand; task_value = v; }
                                if (lane == 0) h.lds[hand].id = kOpenHole;
                                heap_sync();
                                if (++n_tasks == 64) resolve_tasks();
                            }
                        }
                    }
                    Ent<T> w; w.cost = cj; w.id = base + j;
                    if (wave_push_chain(h, chain, w, lane) == chain.top_level) root_cost = cj;   // back() = {i, cost}; std::push_heap
                }
            }
        }
    }
    resolve_tasks();
    chain.flush(h);
    for (uint32_t j = lane; j < k; j += 64) out_ids[j] = h.get(j).id;
    if (lane == 0) atomicAdd(&sc->replacements, n_replaced);
}

// (Round 2 also built the replacement loop as a systolic pipeline on ONE wavefront — lanes = levels of the ancestor chain for the
//  controller, lanes = heap levels for the off-chain passes, every pass in flight advanced by one shared instruction sequence; lane
//  model tools/heap_sys_sim.cpp, kernel in git history (k_heap_select_sys). Exact on the whole GPU suite, 207 instructions per
//  replacement instead of ~290, but a lone wavefront retires one instruction per ~11 clocks on this kind of VALU <-> SALU
//  ping-pong (ballot, ffs, readlane, DPP): 2050 clocks per replacement against 1860 for the two-wave loop below. Not kept.)

// ---- the replacement loop on TWO wavefronts (heaps that reach below LDS) -----------------------------------------------------
// A lone wave is bound by its instruction count, so the work of one replacement is split in the order it happens:
//   wave A  streams the node costs, does the first five heap levels of every pop and the whole push (the ancestor chain), i.e.
//           everything the NEXT replacement depends on (the root and the value at k-1);
//   wave B  continues each pop from level 5 down to the last LDS level and owns the deferred HBM tasks.
// A hands (position at level 5, value) to B through a ring of tokens in LDS and sets that node's bit in a mask of open holes;
// B clears it when the hole is filled. A samples the mask BEFORE it loads a round: if the two level-5 entries its path then
// depends on were open at that moment, it waits for them and loads again (entries it does not use may be stale or torn). A pop
// whose path follows the ancestor chain past level 5 (a few percent) could meet entries the push keeps in A's registers: A
// walks that subtree alone (B never enters it) with its own list of deferred HBM tasks. Every wait is bounded (kSpinLimit)
// and ends in an error, never a hang.
constexpr uint32_t kQueueCap = 64;
constexpr uint32_t kSpinLimit = 1u << 24;
constexpr int kSplitLevel = 5;

template <typename T> struct PipeToken { uint32_t pos; uint32_t id; T cost; };

// Developer builds only: where the two waves' time goes (clock ticks of s_memtime, printed once per launch by each wave's lane 0).
#if defined(BVH_AMD_DEVELOPER)
#define BVH_PIPE_PROF 1
#else
#define BVH_PIPE_PROF 0
#endif
__device__ inline unsigned long long pipe_clock() { return BVH_PIPE_PROF ? __builtin_readcyclecounter() : 0ull; }
struct PipeCtrl { uint32_t head, tail, a_done, b_done, drain_req, drain_ack, error, open5; };   // open5: bit i = level-5 node i is an open hole

// Control words in LDS with acquire / release at workgroup scope. (Relaxed accesses that rely on the LDS unit performing one
// wave's DS instructions in issue order were tried for the saved s_waitcnt, ~6 %, and produced wrong heaps: not kept.)
__device__ inline uint32_t ctrl_load(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void ctrl_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T>
__global__ void __launch_bounds__(128) k_heap_select_pipe(const T* cost, uint32_t n_nodes, uint32_t target, Ent<T>* glob, uint32_t* out_ids,
                                                          ReScalars* sc) {
    extern __shared__ unsigned char heap_lds[];
    WaveHeap<T> h;
    h.lds = (typename WaveHeap<T>::LdsEnt*)heap_lds;
    h.glob = (typename WaveHeap<T>::GlobEnt*)glob;
    unsigned char* after_heap = heap_lds + size_t{HeapCap<T>::v} * sizeof(Ent<T>);
    auto* stage = (__attribute__((address_space(3))) T*)after_heap;
    auto* queue = reinterpret_cast<PipeToken<T>*>(after_heap + kStreamChunk * sizeof(T));
    auto* ctrl = reinterpret_cast<PipeCtrl*>(after_heap + kStreamChunk * sizeof(T) + kQueueCap * sizeof(PipeToken<T>));
    h.cap = HeapCap<T>::v;
    h.cap_level = HeapLevels<T>::v - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t head = min(n_nodes, target + 1);
    const uint32_t k = head - 1;                              // (k - 1 >= cap: checked by the host)
    for (uint32_t j = threadIdx.x; j < h.cap; j += 128) h.set_lds(j, h.get_glob(j));
    if (threadIdx.x < sizeof(PipeCtrl) / 4) reinterpret_cast<uint32_t*>(ctrl)[threadIdx.x] = 0;
    __syncthreads();
    const int d = 31 - __clz(lane + 1);
    const uint32_t in_level = static_cast<uint32_t>(lane + 1) - (1u << d);
    // bounded wait: returns false (and raises the error flag) instead of hanging
    auto wait_until = [&](auto&& ready) -> bool {
        for (uint32_t spin = 0; spin < kSpinLimit; ++spin) {
            if (ready()) return true;
            if (ctrl_load(&ctrl->error)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        ctrl_store(&ctrl->error, 1u);
        return false;
    };

    if (wave == 1) {
        // ---- wave B: levels 5 .. last LDS level of every handed-over pop, and the deferred HBM tasks -------------------------
        Chain<T> no_chain;                                    // (B never touches the chain; wave_adjust_heap is not used here)
        (void)no_chain;
        uint32_t n_tasks = 0; uint32_t task_pos = 0; Ent<T> task_value{};
        auto resolve_tasks = [&]() {
            if (n_tasks) {
                if (static_cast<uint32_t>(lane) < n_tasks) lane_adjust_heap(h, task_pos, k - 1, task_value);
                heap_sync();
                n_tasks = 0;
            }
        };
        uint32_t done = 0;
        T unused_root = T(0);
        unsigned long long t_idle = 0, t_resolve = 0, n_resolve = 0;
        const unsigned long long t_begin = pipe_clock();
        for (;;) {
            bool stop = false;
            const unsigned long long t_w0 = pipe_clock();
            const bool ok = wait_until([&]() {
                if (ctrl_load(&ctrl->head) != done) return true;
                if (ctrl_load(&ctrl->a_done) && ctrl_load(&ctrl->head) == done) { stop = true; return true; }
                return false;
            });
            t_idle += pipe_clock() - t_w0;
            if (!ok || stop) break;
            const PipeToken<T> tok = queue[done % kQueueCap];
            Ent<T> v; v.cost = tok.cost; v.id = tok.id;
            uint32_t hand = 0;
            if (fast_top_round<T, kSplitLevel, HeapLevels<T>::v - 1>(h, v, lane, tok.pos, d, in_level, hand, unused_root, resolve_tasks)) {
                if (static_cast<uint32_t>(lane) == n_tasks) { task_pos = hand; task_value = v; }
                if (lane == 0) h.lds[hand].id = kOpenHole;
                heap_sync();
                if (++n_tasks == 64) { const unsigned long long t_r0 = pipe_clock(); resolve_tasks(); t_resolve += pipe_clock() - t_r0; ++n_resolve; }
            }
            if (lane == 0) __hip_atomic_fetch_and(&ctrl->open5, ~(1u << (tok.pos - 31u)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++done;
            ctrl_store(&ctrl->tail, done);
        }
        resolve_tasks();
        if (BVH_PIPE_PROF && lane == 0)
            printf("[heap pipe] wave B: %u tokens, %llu ticks in all, %llu idle (waiting for a token), %llu in %llu full HBM batches\n", done,
                   pipe_clock() - t_begin, t_idle, t_resolve, n_resolve);
        __threadfence();
        ctrl_store(&ctrl->b_done, 1u);
        return;
    }

    // ---- wave A -------------------------------------------------------------------------------------------------------------
    Chain<T> chain;
    chain.init(h, k, lane);
    chain.load(h);
    uint32_t n_replaced = 0;
    T root_cost = h.lds[0].cost;
    uint32_t sent = 0, tail_seen = 0;                         // tail_seen: B's progress as last read (only re-read when the ring looks full)
    bool failed = false;
    // A's own deferred HBM tasks: pops whose path stays on the chain at level 5 are walked by A alone (B never enters that
    // subtree and A never shares an entry of it with B), their HBM parts collected and sifted like in the one-wave loop
    unsigned long long ta_ring = 0, ta_open = 0, ta_repl = 0, na_own = 0;
    const unsigned long long ta_begin = pipe_clock();
    uint32_t a_tasks = 0; uint32_t a_task_pos = 0; Ent<T> a_task_value{};
    auto resolve_a = [&]() {
        if (a_tasks) {
            if (static_cast<uint32_t>(lane) < a_tasks) lane_adjust_heap(h, a_task_pos, k - 1, a_task_value);
            heap_sync();
            a_tasks = 0;
        }
    };
    for (uint32_t chunk = head; chunk < n_nodes && !failed; chunk += kStreamChunk) {
        bool any = false;
#pragma unroll
        for (uint32_t q = 0; q < kStreamChunk / 64; ++q) {
            const uint32_t i = chunk + q * 64 + lane;
            const T c_hbm = i < n_nodes ? cost[i] : T(0);
            any = any || (i < n_nodes && root_cost < c_hbm);
            stage[q * 64 + lane] = c_hbm;
        }
        if (!__ballot(any)) continue;
        heap_sync();
        for (uint32_t q = 0; q < kStreamChunk / 64 && !failed; ++q) {
            const uint32_t base = chunk + q * 64;
            if (base >= n_nodes) break;
            const T c = stage[q * 64 + lane];
            uint64_t mask = __ballot(base + lane < n_nodes && root_cost < c);
            while (mask && !failed) {
                const int j = __ffsll(static_cast<long long>(mask)) - 1;
                mask &= mask - 1;
                const T cj = lane_value(c, j);
                if (!(root_cost < cj)) continue;
                ++n_replaced;
                const unsigned long long ta_r0 = pipe_clock();
                Ent<T> v; v.cost = lane_value(chain.reg.cost, 0); v.id = lane_value(chain.reg.id, 0);      // position k-1 lives in registers
                // levels 0 .. 5 (the subtree under the root: heap position == BFS index == lane)
                uint32_t hand = 0;
                bool handed = false;
                Ent<T> chain_e{}; bool chain_ready = false;
                for (;;) {
                    const uint32_t open = ctrl_load(&ctrl->open5);
                    const bool inner = lane < 31;
                    Ent<T> c1{}, c2{};
                    if (inner) { c1 = h.get_lds(2 * lane + 1); c2 = h.get_lds(2 * lane + 2); }
                    const bool left = c2.cost > c1.cost;
                    const Ent<T> chosen = left ? c1 : c2;
                    const int next = left ? 2 * lane + 1 : 2 * lane + 2;
                    int r = 0;
                    uint64_t path = 1;
#pragma unroll
                    for (int sidx = 1; sidx < kSplitLevel; ++sidx) { r = __builtin_amdgcn_readlane(next, r); path |= uint64_t{1} << r; }
                    const uint32_t pair_bits = 3u << (2 * r + 1 - 31);            // the level-5 children of the level-4 path node
                    if (open & pair_bits) {                   // one of them was an open hole when this round was loaded
                        const unsigned long long t_o0 = pipe_clock();
                        if (!wait_until([&]() { return (ctrl_load(&ctrl->open5) & pair_bits) == 0; })) { failed = true; break; }
                        ta_open += pipe_clock() - t_o0;
                        continue;
                    }
                    const bool on_path = (path >> lane) & 1u;
                    const uint64_t greater = __ballot(on_path && chosen.cost > v.cost);
                    root_cost = (greater & 1u) ? v.cost : lane_value(chosen.cost, 0);
                    if (greater) {
                        const int landing = __ffsll(static_cast<long long>(greater)) - 1;
                        if (on_path && lane < landing) h.set_lds(lane, chosen);
                        if (lane == landing) h.set_lds(lane, v);
                    } else {
                        if (on_path) h.set_lds(lane, chosen);
                        hand = static_cast<uint32_t>(__builtin_amdgcn_readlane(next, r));
                        handed = true;
                    }
                    heap_sync();
                    break;
                }
                if (failed) break;
                if (handed) {
                    if (chain.holds(hand, kSplitLevel)) {
                        ++na_own;
                        uint32_t hand13 = 0;
                        T unused_root = T(0);
                        if (fast_top_round<T, kSplitLevel, HeapLevels<T>::v - 1>(h, v, lane, hand, d, in_level, hand13, unused_root, resolve_a)) {
                            if (chain.holds(hand13, h.cap_level)) {          // along the chain into HBM (registers): at once
                                if (wave_adjust_heap(h, chain, hand13, k - 1, v, lane)) chain.load(h);
                            } else {
                                if (static_cast<uint32_t>(lane) == a_tasks) { a_task_pos = hand13; a_task_value = v; }
                                if (lane == 0) h.lds[hand13].id = kOpenHole;
                                heap_sync();
                                if (++a_tasks == 64) resolve_a();
                            }
                        }
                    } else {
                        chain_e = chain_fetch(h, chain, lane);             // (in flight during the hand-off)
                        chain_ready = true;
                        if (sent - tail_seen >= kQueueCap) {
                            const unsigned long long t_q0 = pipe_clock();
                            if (!wait_until([&]() { tail_seen = ctrl_load(&ctrl->tail); return sent - tail_seen < kQueueCap; })) { failed = true; break; }
                            ta_ring += pipe_clock() - t_q0;
                        }
                        if (lane == 0) {
                            PipeToken<T> tok; tok.pos = hand; tok.id = v.id; tok.cost = v.cost;
                            queue[sent % kQueueCap] = tok;
                            __hip_atomic_fetch_or(&ctrl->open5, 1u << (hand - 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        heap_sync();
                        ++sent;
                        ctrl_store(&ctrl->head, sent);
                    }
                }
                if (failed) break;
                Ent<T> w; w.cost = cj; w.id = base + j;
                if (!chain_ready) chain_e = chain_fetch(h, chain, lane);
                if (wave_push_chain(h, chain, w, lane, chain_e) == chain.top_level) root_cost = cj;
                ta_repl += pipe_clock() - ta_r0;
            }
        }
    }
    resolve_a();
    if (BVH_PIPE_PROF && lane == 0)
        printf("[heap pipe] wave A: %u replacements (%llu walked alone along the chain), %u tokens, %llu ticks in all, %llu inside replacements, of which "
               "%llu waiting for ring space and %llu for open level-5 holes\n", n_replaced, na_own, sent, pipe_clock() - ta_begin, ta_repl, ta_ring, ta_open);
    ctrl_store(&ctrl->a_done, 1u);
    if (!wait_until([&]() { return ctrl_load(&ctrl->b_done) != 0; })) failed = true;
    if (failed || ctrl_load(&ctrl->error)) { if (lane == 0) atomicOr(&sc->error, 2u); return; }
    __threadfence();
    chain.flush(h);
    for (uint32_t j = lane; j < k; j += 64) out_ids[j] = h.get(j).id;
    if (lane == 0) atomicAdd(&sc->replacements, n_replaced);
}

#include "heap_head.inc"

// ---- fast path: top-k by cost without the heap (valid when no tie straddles the threshold) --------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_cost_keys(const T* cost, uint32_t n, typename Ord<T>::U* keys, uint32_t* ids) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;        // entry i <-> node i + 1 (the root is never a candidate, :93)
    if (i + 1 >= n) return;
    T c = cost[i + 1];
    if (c == T(0)) c = T(0);                                  // -0 == +0 for operator<
    keys[i] = ~Ord<T>::enc(c);                                // ascending radix sort = descending cost
    ids[i] = i + 1;
}

template <typename U>
__global__ void k_check_threshold(const U* sorted_keys, uint32_t k, uint32_t count, ReScalars* sc) {
    // the k-th and (k+1)-th largest costs are equal: which of them the reference's heap keeps depends on its layout
    if (k < count && sorted_keys[k - 1] == sorted_keys[k]) sc->ambiguous = 1u;
}

template <typename T>
__global__ void __launch_bounds__(256) k_gain_keys(const T* neg_gain, uint32_t m, typename Ord<T>::U* keys, uint32_t* order) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    keys[i] = Ord<T>::enc(neg_gain[i]);                       // gains are > 0: no zero to canonicalise
    order[i] = i;
}

// ---- find_reinsertion (:107-188), one lane per candidate ----------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) k_search(const HostNode<T>* nodes, const uint32_t* parent, const uint32_t* cand, uint32_t k,
                                               Move* moves, T* gains, uint32_t* keep, ReScalars* sc, int dim) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= k) return;
    const uint32_t id = cand[c];
    T s_bound[kSearchStack]; uint32_t s_node[kSearchStack];
    int sp = 0;
    uint32_t best_to = 0; T best_gain = T(0);
    T self[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) self[q] = nodes[id].bounds[q];
    const T self_area = ha6(self, dim);
    const uint32_t first_parent = parent[id];
    T gain_so_far = ha6(nodes[first_parent].bounds, dim);
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
            const T gain = bound - ha6(merged, dim);
            if (gain > best_gain) { best_to = dst; best_gain = gain; }
            if (!is_leaf(dn)) {
                const T child_bound = gain + ha6(dn.bounds, dim);
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
            gain_so_far += ha6(nodes[pivot].bounds, dim) - ha6(pivot_box, dim);
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
__device__ inline void apply_move(HostNode<T>* nodes, uint32_t* parent, unsigned char* touched, const Move mv, uint32_t* dirty, uint32_t& n_dirty) {
    using I = typename IndexOf<T>::Type;
    const uint32_t from = mv.from, to = mv.to;
    const uint32_t hot[5] = { to, from, sibling_of(from), parent[to], parent[from] };     // get_conflicts (:227-234)
    bool clash = false;
    for (int q = 0; q < 5; ++q) clash = clash || touched[hot[q]];
    if (clash) return;
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
    if (dirty) { dirty[n_dirty++] = to; dirty[n_dirty++] = par; return; }      // (refit after the iteration's last move: see k_deferrable)
    refit_upwards(nodes, parent, to);
    refit_upwards(nodes, parent, par);
}

// The two refit_from climbs of every applied move (:211-212, up to the root's children each: 2 x ~22 levels of dependent loads, 16 us
// per move on a lone lane — 87 % of k_apply, 17 ms of a 10M-triangle High build) can be left to ONE parallel bottom-up pass after the
// iteration's last move — over the ancestors, in the tree as the moves leave it, of every move's `to` and old parent (k_dirty_mark,
// k_dirty_refit; the root only when a move from under it was applied, like refit_from): every node whose subtree a move changed is
// among them, and
// recomputing a node the reference did NOT recompute cannot change a bit of it exactly when
//   * every inner node's box equals left.bbox.extend(right.bbox) BITWISE on entry (a builder's tree does; a caller's hand-made tree
//     with loose boxes does not) — by induction every node then does after every move: refit_from recomputes the parent of every slot
//     whose content changed, and all of its ancestors but the root, after the children have their final boxes;
//   * no box holds -0.0 (the root's box is the union of all leaves whichever way it is associated — but `a < b ? a : b` picks between
//     +0.0 and -0.0 by position, and the root is recomputed only by a move whose `from` hangs under it), and no NaN.
// Otherwise (flag raised) the moves refit as they go, like the reference.
template <typename T>
__global__ void __launch_bounds__(256) k_deferrable(const HostNode<T>* nodes, uint32_t n, uint32_t* not_deferrable) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HostNode<T>& nd = nodes[i];
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 6; ++q) bad = bad || !(nd.bounds[q] == nd.bounds[q]) || (nd.bounds[q] == T(0) && signbit(nd.bounds[q]));
    if (!is_leaf(nd)) {
        const HostNode<T>& l = nodes[first_of(nd)];
        const HostNode<T>& r = nodes[first_of(nd) + 1];
#pragma unroll
        for (int q = 0; q < 3; ++q) {                         // (no -0.0 and no NaN anywhere: equal values are equal bits)
            bad = bad || !(nd.bounds[2 * q] == pick_min(l.bounds[2 * q], r.bounds[2 * q]));
            bad = bad || !(nd.bounds[2 * q + 1] == pick_max(l.bounds[2 * q + 1], r.bounds[2 * q + 1]));
        }
    }
    if (bad) *not_deferrable = 1u;
}

// check_ties (fast path): `order` is sorted by gain but equal gains are in an arbitrary order. A group of equal gains is
// order-independent iff its still-applicable members (no clash with what earlier groups touched) have pairwise disjoint
// conflict sets (evaluated on the live parents at the start of the group: applying one such member changes parent[] only
// below nodes of its own conflict set). Otherwise: flag and stop, the host rolls the iteration back.
template <typename T>
__global__ void k_apply(HostNode<T>* nodes, uint32_t* parent, unsigned char* touched, const Move* moves, const uint32_t* order,
                        const T* neg_gain, uint32_t m, int check_ties, uint32_t* group_mark, uint32_t gid_base, ReScalars* sc,
                        uint32_t* dirty, uint32_t* dirty_count) {     // dirty != null: the refits are deferred, the ends of the applied moves are listed
    uint32_t gid = gid_base, n_dirty = 0;
    if (dirty) *dirty_count = 0;
    for (uint32_t j = 0; j < m;) {
        uint32_t e = j + 1;
        if (check_ties) {
            const T g = neg_gain[order[j]];
            while (e < m && neg_gain[order[e]] == g) ++e;
            if (e - j > 1) {
                ++gid;
                for (uint32_t q = j; q < e; ++q) {
                    const Move mv = moves[order[q]];
                    const uint32_t hot[5] = { mv.to, mv.from, sibling_of(mv.from), parent[mv.to], parent[mv.from] };
                    bool clash = false, shared = false;
                    for (int h = 0; h < 5; ++h) { clash = clash || touched[hot[h]]; shared = shared || group_mark[hot[h]] == gid; }
                    if (clash) continue;
                    if (shared) {
#if defined(BVH_AMD_DEVELOPER)
                        printf("[k_apply] tie group of %u moves at rank %u of %u, gain %.9g:\n", e - j, j, m, (double)-g);
                        for (uint32_t qq = j; qq < e && qq < j + 8; ++qq) {
                            const Move mm = moves[order[qq]];
                            printf("    from %u (sibling %u, parent %u) -> to %u (parent %u)\n", mm.from, sibling_of(mm.from), parent[mm.from], mm.to, parent[mm.to]);
                        }
#endif
                        sc->ambiguous = 1u; if (dirty) *dirty_count = n_dirty; return;
                    }
                    for (int h = 0; h < 5; ++h) group_mark[hot[h]] = gid;
                }
            }
        }
        for (uint32_t q = j; q < e; ++q) apply_move(nodes, parent, touched, moves[order[q]], dirty, n_dirty);
        j = e;
    }
    if (dirty) *dirty_count = n_dirty;
}

// The deferred refit. state[] (zero on entry) per node: low byte = children among the nodes to recompute, 0x100 = listed itself,
// 0x10000 x children that have been recomputed. Pass 1: every listed node climbs and announces itself to its parent until it meets a
// node somebody has been at (every edge of the union of the paths is walked once; the root is announced to like any node). Pass 2: the listed nodes nobody announced to
// recompute themselves and climb; a parent is recomputed by the last of its announced children to arrive (tickets like k_refit's).
constexpr uint32_t kDirtyListed = 0x100u, kDirtyDone = 0x10000u;
__global__ void __launch_bounds__(256) k_dirty_mark(const uint32_t* parent, const uint32_t* dirty, const uint32_t* dirty_count, uint32_t* state) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= *dirty_count) return;
    uint32_t i = dirty[t];
    if (atomicOr(&state[i], kDirtyListed) != 0 || i == 0) return;     // (somebody from below has passed already; the root has nobody above it)
    for (;;) {
        const uint32_t p = parent[i];
        if (atomicAdd(&state[p], 1u) != 0 || p == 0) return;
        i = p;
    }
}

template <typename T>
__device__ inline void refit_shared_node(HostNode<T>* nodes, uint32_t cur) {      // left.get_bbox().extend(right.get_bbox()), boxes exchanged between climbers
    HostNode<T>& nd = nodes[cur];
    const uint32_t f = first_of(nd);
    const T* l = nodes[f].bounds;
    const T* r = nodes[f + 1].bounds;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const T lo_l = __hip_atomic_load(&l[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), lo_r = __hip_atomic_load(&r[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const T hi_l = __hip_atomic_load(&l[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi_r = __hip_atomic_load(&r[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&nd.bounds[2 * q], pick_min(lo_l, lo_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&nd.bounds[2 * q + 1], pick_max(hi_l, hi_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_dirty_refit(HostNode<T>* nodes, const uint32_t* parent, const uint32_t* dirty, const uint32_t* dirty_count, uint32_t* state) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= *dirty_count) return;
    uint32_t i = dirty[t];
    if ((state[i] & 0xFFu) != 0) return;                      // nodes to recompute below this one: the last of them to arrive does it
    for (;;) {
        if (!is_leaf(nodes[i])) refit_shared_node(nodes, i);
        if (i == 0) return;
        const uint32_t p = parent[i];
        ticket_release();                                     // this lane's box before the ticket
        const uint32_t old = atomicAdd(&state[p], kDirtyDone);
        if ((old >> 16) + 1 < (old & 0xFFu)) return;          // an announced child is still on its way
        // The root: refit_from stops below it (:224) — except refit_from(0), which a move whose `from` hung under the root makes: that
        // move has put the sibling's node, box included, into slot 0 (:207), so the root MUST be recomputed then (it is listed), and
        // must not be otherwise.
        if (p == 0 && !(old & kDirtyListed)) return;
        ticket_acquire();
        i = p;
    }
}

// Bvh::refit (bvh.h:185-218): every inner box = left.bbox.extend(right.bbox), children before parents. One lane per
// leaf climbs; the second child to arrive at a node computes it (agent-scope fences order the hand-off across XCDs).
template <typename T>
__global__ void __launch_bounds__(256) k_refit(HostNode<T>* nodes, const uint32_t* parent, uint32_t* arrived, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || i == 0 || !is_leaf(nodes[i])) return;
    // (refit_device fills parent[] with 0xFFFFFFFF first: a node no inner node references is the top of its own subtree — the
    //  validation tolerates such arrays like the reference does — and the climb ends there; the reference's traverse_bottom_up,
    //  bvh.h:187-204, walks such a subtree bottom-up too and stops at its top)
    uint32_t cur = parent[i];
    if (cur == 0xFFFFFFFFu) return;
    for (;;) {
        ticket_release();                                     // this lane's box before the ticket (build_common.h: no cache maintenance needed)
        if (atomicAdd(&arrived[cur], 1u) == 0) return;        // first child: the sibling's lane finishes this node
        ticket_acquire();
        HostNode<T>& nd = nodes[cur];
        const uint32_t f = first_of(nd);
        const T* l = nodes[f].bounds;
        const T* r = nodes[f + 1].bounds;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const T lo_l = __hip_atomic_load(&l[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), lo_r = __hip_atomic_load(&r[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const T hi_l = __hip_atomic_load(&l[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi_r = __hip_atomic_load(&r[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nd.bounds[2 * q], pick_min(lo_l, lo_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nd.bounds[2 * q + 1], pick_max(hi_l, hi_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cur == 0) return;
        cur = parent[cur];
        if (cur == 0xFFFFFFFFu) return;
    }
}

} // namespace

template <typename T>
int refit_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream) {   // (dimension-independent: z stays (+0, +0) in 2D)
    StreamScope scratch_on(stream);
    const uint32_t n = static_cast<uint32_t>(node_count);
    if (n < 3) return BVH_AMD_OK;
    DevBuf<uint32_t> parent, arrived;
    DevBuf<T> cost;
    hipError_t e = parent.alloc(n);
    if (e == hipSuccess) e = arrived.alloc(n);
    if (e == hipSuccess) e = cost.alloc(n);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("refit: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(arrived.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemsetAsync(parent.p, 0xFF, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_parents_costs<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p, cost.p, 1, 3);
    hipLaunchKernelGGL(k_refit<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, parent.p, arrived.p, n);
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}
template int refit_device<float>(HostNode<float>*, size_t, hipStream_t);
template int refit_device<double>(HostNode<double>*, size_t, hipStream_t);

static std::atomic<unsigned> g_fast_iterations{0}, g_exact_iterations{0};

void reinsertion_stats(unsigned out[2]) { out[0] = g_fast_iterations.load(); out[1] = g_exact_iterations.load(); }

// The calling thread's latest optimize / High build (bvh_amd_last_optimize_profile): how many iterations ran, how many of them
// had to replay the candidate heap exactly, how many pop + push replacements those replays made and the GPU time of the
// heap kernels (make_heap levels + replacement loop) by events on the stream.
static thread_local bvh_amd_optimize_profile t_profile = {0, 0, 0, 0.0f};
void last_optimize_profile(bvh_amd_optimize_profile* out) { *out = t_profile; }

// ReinsertionOptimizer::optimize on device-resident nodes (reference layout), in place.
template <typename T>
int reinsertion_optimize_config_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim, double batch_size_ratio,
                                        size_t iterations) {
    using U = typename Ord<T>::U;
    StreamScope scratch_on(stream);
    const uint32_t n = static_cast<uint32_t>(node_count);
    if (n < 2 || iterations == 0) return BVH_AMD_OK;
    if (!(batch_size_ratio >= 0.0)) return fail(BVH_AMD_ERR_ARG, "optimize: batch_size_ratio must be >= 0");
    const T ratio = static_cast<T>(batch_size_ratio);         // Config (:19-25)
    if (static_cast<double>(static_cast<T>(node_count) * ratio) >= 1.8e19)        // the reference casts this product to size_t (:238-239)
        return fail(BVH_AMD_ERR_ARG, "optimize: batch_size_ratio * node count does not fit size_t");
    // :238-239; anything beyond the node count selects every node but the root (find_candidates, :91)
    const uint32_t batch = static_cast<uint32_t>(std::min<size_t>(n, std::max<size_t>(1, static_cast<size_t>(static_cast<T>(node_count) * ratio))));
    const uint32_t head = std::min<uint32_t>(n, batch + 1), k = head - 1;
    if (k == 0) return BVH_AMD_OK;
    const char* mode = std::getenv("BVH_AMD_REINSERT");
    const bool always_exact = mode && std::strcmp(mode, "exact") == 0;

    DevBuf<uint32_t> parent, cand, keep, off, order, group_mark, ids, ids_tmp, hist, dirty_state, dirty, dirty_count, not_deferrable;
    DevBuf<Ent<T>> heap_g;
    DevBuf<T> cost, gains, neg_gain;
    DevBuf<U> keys, keys_tmp;
    DevBuf<Move> moves, kept;
    DevBuf<unsigned char> touched;
    DevBuf<ReScalars> scalars;
    DevBuf<HostNode<T>> backup;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(parent.alloc(n)); A(heap_g.alloc(k)); A(cand.alloc(k)); A(keep.alloc(k)); A(off.alloc(k)); A(order.alloc(k));
    A(cost.alloc(n)); A(gains.alloc(k)); A(neg_gain.alloc(k)); A(moves.alloc(k)); A(kept.alloc(k));
    A(touched.alloc(n)); A(scalars.alloc(1)); A(dirty_state.alloc(n)); A(dirty.alloc(size_t{2} * k)); A(dirty_count.alloc(1)); A(not_deferrable.alloc(1));
    if (!always_exact) {
        A(group_mark.alloc(n)); A(ids.alloc(n)); A(ids_tmp.alloc(n)); A(keys.alloc(n)); A(keys_tmp.alloc(n));
        A(hist.alloc(radix_sort_hist_words(n, 1))); A(backup.alloc(n));
    }
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("optimize: hipMalloc: ") + hipGetErrorString(e));
    BVH_HIP_TRY(hipMemsetAsync(scalars.p, 0, sizeof(ReScalars), stream), BVH_AMD_ERR_HIP);
    if (!always_exact) BVH_HIP_TRY(hipMemsetAsync(group_mark.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
    // may the moves leave their refits to one pass per iteration? (k_deferrable; developer knob BVH_AMD_APPLY_DEFER=0: refit move by move)
    bool defer_refit = false;
    if (BVH_DEV_INT("BVH_AMD_APPLY_DEFER", 1) != 0) {
        uint32_t h_not = 1;
        BVH_HIP_TRY(hipMemsetAsync(not_deferrable.p, 0, 4, stream), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_deferrable<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, not_deferrable.p);
        BVH_HIP_TRY(hipMemcpyAsync(&h_not, not_deferrable.p, 4, hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        defer_refit = h_not == 0;
        if (BVH_DEV_STR("BVH_AMD_REINSERT_DEBUG")) std::fprintf(stderr, "[bvh_amd] reinsertion: refits %s\n", defer_refit ? "deferred to one pass per iteration" : "move by move (loose boxes, -0.0 or NaN in the tree)");
    }

    const size_t heap_lds = size_t{HeapCap<T>::v} * sizeof(Ent<T>) + kStreamChunk * sizeof(T);
    const bool below_lds = k >= 1 && k - 1 >= HeapCap<T>::v;
    // heaps that reach below LDS: the register-resident head (heap_head.inc). Developer knob BVH_AMD_HEAP_PIPE: 0 = the one-wave
    // replacement loop, 1 = round 5's two-wave loop, anything else = the head kernel.
    const char* pipe_knob = BVH_DEV_STR("BVH_AMD_HEAP_PIPE");
    const int pipe_mode = pipe_knob ? std::atoi(pipe_knob) : 2;
    const bool use_head = pipe_mode != 0 && pipe_mode != 1 && n < kHeadMaxNodes;     // (its words tag node ids with two bits)
    const bool use_pipe = pipe_mode == 1 || (pipe_mode != 0 && !use_head);
    int heap_depth = 0;                                       // level of the last heap position k - 1
    while ((uint64_t{2} << heap_depth) <= k) ++heap_depth;
    const char* wide_knob = BVH_DEV_STR("BVH_AMD_HEAP_WIDE");                       // developer: the 64-lane ancestor masks on a heap that does not need them
    const bool wide_masks = heap_depth + 12 > 32 || (wide_knob && std::atoi(wide_knob) != 0);
    auto head_kernel = wide_masks ? k_heap_select_head<T, true> : k_heap_select_head<T, false>;               // ancestor masks of 64 / 32 lanes
    const size_t pipe_lds = heap_lds + kQueueCap * sizeof(PipeToken<T>) + sizeof(PipeCtrl);
    if (below_lds && use_pipe)
        BVH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_heap_select_pipe<T>), hipFuncAttributeMaxDynamicSharedMemorySize, int(pipe_lds)),
                    BVH_AMD_ERR_HIP);
    if (below_lds && use_head)
        BVH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(HeadLayout<T>::bytes)),
                    BVH_AMD_ERR_HIP);
    auto heap_kernel = below_lds ? k_heap_select<T, true> : k_heap_select<T, false>;
    BVH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(heap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(heap_lds)),
                BVH_AMD_ERR_HIP);
    auto read_scalars = [&](ReScalars& hs) -> int {
        BVH_HIP_TRY(hipMemcpyAsync(&hs, scalars.p, sizeof(hs), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        return BVH_AMD_OK;
    };
    auto clear_ambiguous = [&]() -> int {
        BVH_HIP_TRY(hipMemsetAsync(&scalars.p->ambiguous, 0, 4, stream), BVH_AMD_ERR_HIP);
        return BVH_AMD_OK;
    };
    t_profile = {static_cast<unsigned>(iterations), 0u, 0ull, 0.0f};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> heap_events;      // around the heap kernels of each exact replay
    struct EventBin { std::vector<std::pair<hipEvent_t, hipEvent_t>>& v; ~EventBin() { for (auto& e : v) { if (e.first) (void)hipEventDestroy(e.first); if (e.second) (void)hipEventDestroy(e.second); } } } event_bin{heap_events};
    bool parents_valid = false;
    uint64_t gid_base = 0;                                    // k_apply numbers its tie groups gid_base + 1 .. gid_base + k at most
    for (size_t it = 0; it < iterations; ++it, gid_base += uint64_t{k} + 1) {
        if (!always_exact && gid_base + k + 1 > 0xFFFFFFFFull) {      // many iterations: start the group numbers over
            BVH_HIP_TRY(hipMemsetAsync(group_mark.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
            gid_base = 0;
        }
        bool exact = always_exact;
        for (;;) {                                            // at most two rounds: fast, then (if the layout matters) exact
            hipLaunchKernelGGL(k_parents_costs<T>, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, n, parent.p, cost.p, parents_valid ? 0 : 1, dim);
            parents_valid = true;
            ReScalars hs;
            int rc;
            if (exact) {
                heap_events.emplace_back(nullptr, nullptr);
                if (hipEventCreate(&heap_events.back().first) == hipSuccess && hipEventCreate(&heap_events.back().second) == hipSuccess)
                    (void)hipEventRecord(heap_events.back().first, stream);
                hipLaunchKernelGGL(k_heap_fill<T>, dim3((k + 255) / 256), dim3(256), 0, stream, cost.p, k, heap_g.p);
                if (k >= 2) {
                    const uint32_t last_parent = (k - 2) / 2;
                    int level = 0;
                    while ((uint64_t{2} << level) - 2 < last_parent) ++level;          // level of last_parent
                    for (; level >= 0; --level) {
                        const uint32_t first = (1u << level) - 1;
                        const uint32_t count = std::min<uint32_t>((2u << level) - 2, last_parent) - first + 1;
                        hipLaunchKernelGGL(k_make_heap_level<T>, dim3((count + 63) / 64), dim3(64), 0, stream, heap_g.p, k, first, count);
                    }
                }
                if (below_lds && use_head) {
                    hipLaunchKernelGGL(head_kernel, dim3(1), dim3(64 * kHeadWaves), HeadLayout<T>::bytes, stream, cost.p, n, batch, heap_g.p, scalars.p);
                    hipLaunchKernelGGL(k_heap_ids<T>, dim3((k + 255) / 256), dim3(256), 0, stream, heap_g.p, k, cand.p);
                } else if (below_lds && use_pipe)
                    hipLaunchKernelGGL(k_heap_select_pipe<T>, dim3(1), dim3(128), pipe_lds, stream, cost.p, n, batch, heap_g.p, cand.p, scalars.p);
                else
                    hipLaunchKernelGGL(heap_kernel, dim3(1), dim3(64), heap_lds, stream, cost.p, n, batch, heap_g.p, cand.p, scalars.p);
                if (heap_events.back().second) (void)hipEventRecord(heap_events.back().second, stream);
            } else {
                BVH_HIP_TRY(hipMemcpyAsync(backup.p, d_nodes, size_t{n} * sizeof(HostNode<T>), hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
                hipLaunchKernelGGL(k_cost_keys<T>, dim3((n + 255) / 256), dim3(256), 0, stream, cost.p, n, keys.p, ids.p);
                rc = radix_sort_pairs<U>(keys.p, ids.p, keys_tmp.p, ids_tmp.p, n - 1, 1, int(sizeof(U) * 8), stream, hist.p);
                if (rc) return rc;
                hipLaunchKernelGGL(k_check_threshold<U>, dim3(1), dim3(1), 0, stream, keys.p, k, n - 1, scalars.p);
                BVH_HIP_TRY(hipMemcpyAsync(cand.p, ids.p, size_t{k} * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
            }
            BVH_HIP_TRY(hipMemsetAsync(touched.p, 0, n, stream), BVH_AMD_ERR_HIP);
            hipLaunchKernelGGL(k_search<T>, dim3((k + 63) / 64), dim3(64), 0, stream, d_nodes, parent.p, cand.p, k, moves.p, gains.p, keep.p, scalars.p, dim);
            BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
            uint32_t m = 0;
            rc = exclusive_scan_u32(keep.p, off.p, k, &m, stream);
            if (rc) return rc;
            if (!exact) {
                if ((rc = read_scalars(hs))) return rc;
                if (hs.ambiguous) {                           // nothing was modified yet
                    if (BVH_DEV_STR("BVH_AMD_REINSERT_DEBUG")) std::fprintf(stderr, "[bvh_amd] reinsertion iteration %zu: tie at the top-k threshold -> exact replay\n", it);
                    if ((rc = clear_ambiguous())) return rc;
                    exact = true;
                    continue;
                }
            }
            if (m != 0) {
                hipLaunchKernelGGL(k_compact<T>, dim3((k + 255) / 256), dim3(256), 0, stream, moves.p, gains.p, keep.p, off.p, k, kept.p, neg_gain.p);
                if (exact) {
                    rc = std_sort_ids<T>(order.p, neg_gain.p, m, 1, 0, 1, stream);
                } else {
                    hipLaunchKernelGGL(k_gain_keys<T>, dim3((m + 255) / 256), dim3(256), 0, stream, neg_gain.p, m, keys.p, order.p);
                    rc = radix_sort_pairs<U>(keys.p, order.p, keys_tmp.p, ids_tmp.p, m, 1, int(sizeof(U) * 8), stream, hist.p);
                }
                if (rc) return rc;
                hipLaunchKernelGGL(k_apply<T>, dim3(1), dim3(1), 0, stream, d_nodes, parent.p, touched.p, kept.p, order.p, neg_gain.p, m,
                                   exact ? 0 : 1, group_mark.p, static_cast<uint32_t>(gid_base), scalars.p,
                                   defer_refit ? dirty.p : static_cast<uint32_t*>(nullptr), dirty_count.p);
                if (defer_refit) {                            // the ancestors of the applied moves' ends, children before parents
                    BVH_HIP_TRY(hipMemsetAsync(dirty_state.p, 0, size_t{n} * 4, stream), BVH_AMD_ERR_HIP);
                    hipLaunchKernelGGL(k_dirty_mark, dim3(static_cast<uint32_t>((size_t{2} * m + 255) / 256)), dim3(256), 0, stream, parent.p, dirty.p, dirty_count.p, dirty_state.p);
                    hipLaunchKernelGGL(k_dirty_refit<T>, dim3(static_cast<uint32_t>((size_t{2} * m + 255) / 256)), dim3(256), 0, stream, d_nodes, parent.p, dirty.p, dirty_count.p, dirty_state.p);
                }
                BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
                if (!exact) {
                    if ((rc = read_scalars(hs))) return rc;
                    if (hs.ambiguous) {                       // roll the iteration back and replay it exactly
                        if (BVH_DEV_STR("BVH_AMD_REINSERT_DEBUG")) std::fprintf(stderr, "[bvh_amd] reinsertion iteration %zu: equal gains with shared nodes -> exact replay\n", it);
                        BVH_HIP_TRY(hipMemcpyAsync(d_nodes, backup.p, size_t{n} * sizeof(HostNode<T>), hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
                        if ((rc = clear_ambiguous())) return rc;
                        parents_valid = false;
                        exact = true;
                        continue;
                    }
                }
            }
            (exact ? g_exact_iterations : g_fast_iterations).fetch_add(1);
            if (exact) ++t_profile.replayed;
            break;
        }
    }
    ReScalars hs;
    BVH_HIP_TRY(hipMemcpyAsync(&hs, scalars.p, sizeof(hs), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
    t_profile.replacements = hs.replacements;
    for (auto& ev : heap_events) {
        float ms = 0.0f;
        if (ev.first && ev.second && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) t_profile.heap_ms += ms;
    }
    (void)hipGetLastError();
    if (hs.error & 2u) return fail(BVH_AMD_ERR_HIP, "optimize: the two-wave candidate-heap replay timed out (internal error)");
    if (hs.error) return fail(BVH_AMD_ERR_OVERFLOW, "optimize: reinsertion search stack exceeded 96 entries");
    return BVH_AMD_OK;
}

// ReinsertionOptimizer::optimize with its default Config {0.05, 3}
template <typename T>
int reinsertion_optimize_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim) {
    return reinsertion_optimize_config_device<T>(d_nodes, node_count, stream, dim, 0.05, 3);
}

template int reinsertion_optimize_device<float>(HostNode<float>*, size_t, hipStream_t, int);
template int reinsertion_optimize_device<double>(HostNode<double>*, size_t, hipStream_t, int);
template int reinsertion_optimize_config_device<float>(HostNode<float>*, size_t, hipStream_t, int, double, size_t);
template int reinsertion_optimize_config_device<double>(HostNode<double>*, size_t, hipStream_t, int, double, size_t);

} // namespace bvh_amd
