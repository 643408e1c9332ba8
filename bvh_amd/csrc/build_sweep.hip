// K8 — sweep-SAH top-down construction on gfx950, bit-exact with the reference's SweepSahBuilder
// (sweep_sah_builder.h:50-138) under TopDownSahBuilder::build (top_down_sah_builder.h:74-139).
//
//  * the three centroid-sorted id arrays come from the exact std::sort emulation (sort_emul.hip); their
//    tie arrangement decides which primitives a candidate plane separates;
//  * per node and axis, the reference's right-to-left / left-to-right sweeps with chunked early-outs are a
//    full evaluation: cost(i) = half_area(box[begin, i]) * (i + 1 - begin) + half_area(box[i + 1, end)) * (end - i - 1),
//    argmin with strict `<`, lowest (axis, position) on ties (probe-verified): here a suffix and a
//    prefix min/max scan (exact: min/max are associative) and an argmin reduction;
//  * the other two axes are stable-partitioned by membership (std::stable_partition, :128-136);
//  * big segments (> 64 primitives) run level-synchronously, one block per (segment, axis) for the scans and
//    2048-primitive chunks for the per-position passes; segments <= 64 are finished by ONE wavefront that keeps
//    the three orders as lane permutations in LDS; node numbering and emission are shared with the binned
//    builder (build_common.h, Phase C).

#include "build_common.h"

#include <cstdlib>
#include <cstring>

namespace bvh_amd {

using namespace bld;

namespace {

constexpr int kScanThreads = 512;

template <typename T> struct AxisBest { T cost; uint32_t pos; uint32_t pad; };

template <typename T>
struct SweepCtx {
    BuildCtx<T> b;                       // b.ids == ord[0]
    uint32_t* ord[3];
    uint32_t* ord_tmp;                   // 2 * n
    uint32_t* marks;                     // per primitive id
    T* cost_r;                           // 3 * n
    AxisBest<T>* axis_best;              // slot_cap * 3
    uint32_t* chunk_true2;               // 2 * task_cap
    // multi-block scans of segments beyond kSweepBig primitives (round 4): per (axis, 2048-primitive chunk) the chunk's box, the boxes
    // of everything before / after it in the segment, its best candidate
    T* chunk_box;                        // 3 * task_cap * 6
    T* carry_l;                          // 3 * task_cap * 6
    T* carry_r;                          // 3 * task_cap * 6
    AxisBest<T>* chunk_best;             // 3 * task_cap
    uint32_t multi;                      // this level launches the multi-block kernels: k_sweep_axis leaves the big segments to them
};
constexpr uint32_t kSweepBig = 4 * kChunk;   // segments of more primitives are scanned by one block per chunk instead of one block per segment

template <typename T> struct Box6 { T lo[3], hi[3]; };

template <typename T> __device__ inline Box6<T> empty_box() {
    Box6<T> b;
#pragma unroll
    for (int k = 0; k < 3; ++k) { b.lo[k] = Ord<T>::kMax; b.hi[k] = -Ord<T>::kMax; }
    return b;
}
template <typename T> __device__ inline Box6<T> join(const Box6<T>& a, const Box6<T>& o) {
    Box6<T> r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = pick_min(a.lo[k], o.lo[k]); r.hi[k] = pick_max(a.hi[k], o.hi[k]); }
    return r;
}
template <typename T> __device__ inline Box6<T> shfl_up_box(const Box6<T>& a, int off) {
    Box6<T> r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = __shfl_up(a.lo[k], off); r.hi[k] = __shfl_up(a.hi[k], off); }
    return r;
}
template <typename T> __device__ inline Box6<T> load_box(const T* bboxes, uint32_t id) {
    Box6<T> r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = bboxes[6ull * id + k]; r.hi[k] = bboxes[6ull * id + 3 + k]; }
    return r;
}

// Inclusive scan of one tile of kScanThreads boxes with a carry from earlier tiles. `valid` lanes beyond the
// data contribute the empty box. Returns the inclusive value for this thread; updates carry (all threads).
template <typename T>
__device__ inline Box6<T> block_scan_boxes(Box6<T> v, Box6<T>& carry, T (*wtot)[6]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Box6<T> o = shfl_up_box(v, off);
        if (lane >= off) v = join(o, v);
    }
    __syncthreads();                                        // wtot reuse across tiles
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { wtot[wave][k] = v.lo[k]; wtot[wave][3 + k] = v.hi[k]; }
    }
    __syncthreads();
    Box6<T> pre = carry, tile = carry;
    for (int w = 0; w < kScanThreads / 64; ++w) {
        Box6<T> t;
#pragma unroll
        for (int k = 0; k < 3; ++k) { t.lo[k] = wtot[w][k]; t.hi[k] = wtot[w][3 + k]; }
        if (w < wave) pre = join(pre, t);
        tile = join(tile, t);
    }
    carry = tile;
    return join(pre, v);
}

// ---- find_best_split of BIG segments, one block per 2048-primitive chunk (round 4) ---------------------------------------------------
// k_sweep_axis below gives every (segment, axis) ONE block that walks the segment tile by tile with a running carry: fine for the
// thousands of small segments of a deep level, but the top of the tree is a single segment — the 10M-triangle soup's pruned top
// level has 6.2M roots, a serial 1M-primitive Medium / High build has 1M — and three blocks then scan millions of boxes on their own
// (VERDICT r3 Weak 7: 3 ms blocks; 6.4 of the 25.7 ms of a 10M Medium build). Segments beyond kSweepBig primitives are therefore
// scanned in three steps over the chunks (= the level's tasks): (1) the box of every chunk, (2) per segment and axis a scan over its
// chunk boxes: what lies before / after each chunk, (3) the right-to-left and the left-to-right scans inside every chunk, started from
// those carries, and an arg-min over the chunks' best candidates with the reference's tie rule (lowest position). min / max are
// associative and exact, so every position sees bit for bit the box the single walk builds.
template <typename T> __device__ inline Box6<T> wave_join_all(Box6<T> v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) { v.lo[k] = pick_min(v.lo[k], __shfl_xor(v.lo[k], off)); v.hi[k] = pick_max(v.hi[k], __shfl_xor(v.hi[k], off)); }
    return v;
}
template <typename T> __device__ inline void store_box(T* p, const Box6<T>& v) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { p[k] = v.lo[k]; p[3 + k] = v.hi[k]; }
}
template <typename T> __device__ inline Box6<T> read_box(const T* p) {
    Box6<T> v;
#pragma unroll
    for (int k = 0; k < 3; ++k) { v.lo[k] = p[k]; v.hi[k] = p[3 + k]; }
    return v;
}
template <typename T> __device__ inline bool big_segment(const SweepCtx<T>& c, uint32_t slot, uint32_t& b, uint32_t& e) {
    const ANode<T>& nd = c.b.nodes[c.b.state[slot].node];
    b = nd.begin; e = nd.end;
    return e - b > kSweepBig;
}

template <typename T>
__global__ void __launch_bounds__(256) k_sweep_chunk_box(SweepCtx<T> c, uint32_t n_tasks) {
    __shared__ T wtot[4][6];
    const uint32_t task = blockIdx.x % n_tasks, axis = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    uint32_t b, e;
    if (!big_segment(c, tk.slot, b, e)) return;
    const uint32_t* ord = c.ord[axis];
    Box6<T> v = empty_box<T>();
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) v = join(v, load_box(c.b.bboxes, ord[pos]));
    v = wave_join_all(v);
    if ((threadIdx.x & 63) == 0) store_box(wtot[threadIdx.x >> 6], v);
    __syncthreads();
    if (threadIdx.x == 0) {
        Box6<T> t = read_box(wtot[0]);
        for (int w = 1; w < 4; ++w) t = join(t, read_box(wtot[w]));
        store_box(c.chunk_box + (size_t{axis} * c.b.task_cap + task) * 6, t);
    }
}

// per (segment, axis), one wave: carry_l[chunk] = box of the chunks before it, carry_r[chunk] = box of the chunks after it
template <typename T>
__global__ void __launch_bounds__(64) k_sweep_chunk_carry(SweepCtx<T> c) {
    const uint32_t slot = blockIdx.x / 3, axis = blockIdx.x % 3;
    uint32_t b, e;
    if (!big_segment(c, slot, b, e)) return;
    const SlotState<T>& st = c.b.state[slot];
    const int lane = threadIdx.x;
    const size_t base = size_t{axis} * c.b.task_cap + st.task0;
    for (int dir = 0; dir < 2; ++dir) {
        Box6<T> carry = empty_box<T>();
        T* out = dir == 0 ? c.carry_l : c.carry_r;
        for (uint32_t t0 = 0; t0 < st.ntasks; t0 += 64) {
            const uint32_t j = t0 + lane;                      // index along the direction of the scan
            const bool in = j < st.ntasks;
            const uint32_t t = dir == 0 ? j : st.ntasks - 1 - (in ? j : 0);
            Box6<T> v = in ? read_box(c.chunk_box + (base + t) * 6) : empty_box<T>();
            Box6<T> incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { Box6<T> o = shfl_up_box(incl, off); if (lane >= off) incl = join(o, incl); }
            Box6<T> excl = shfl_up_box(incl, 1);
            if (lane == 0) excl = empty_box<T>();
            if (in) store_box(out + (base + t) * 6, join(carry, excl));
            Box6<T> total;
#pragma unroll
            for (int k = 0; k < 3; ++k) { total.lo[k] = __shfl(incl.lo[k], 63); total.hi[k] = __shfl(incl.hi[k], 63); }
            carry = join(carry, total);
        }
    }
}

// right-to-left inside one chunk, from the box of everything right of it: cost_r[i] = half_area(box[i, e)) * (e - i), i in (b, e)
template <typename T>
__global__ void __launch_bounds__(kScanThreads) k_sweep_chunk_right(SweepCtx<T> c, uint32_t n_tasks) {
    __shared__ T wtot[kScanThreads / 64][6];
    const uint32_t task = blockIdx.x % n_tasks, axis = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    uint32_t b, e;
    if (!big_segment(c, tk.slot, b, e)) return;
    const uint32_t* ord = c.ord[axis];
    T* cost_r = c.cost_r + size_t{axis} * c.b.n;
    Box6<T> carry = read_box(c.carry_r + (size_t{axis} * c.b.task_cap + task) * 6);
    const uint32_t len = tk.end - tk.begin;
    for (uint32_t base = 0; base < len; base += kScanThreads) {
        const uint32_t r = base + threadIdx.x;
        const bool in = r < len;
        const uint32_t pos = tk.end - 1 - (in ? r : 0);
        Box6<T> v = in ? load_box(c.b.bboxes, ord[pos]) : empty_box<T>();
        v = block_scan_boxes(v, carry, wtot);
        if (in && pos > b) cost_r[pos] = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(e - pos, c.b.sah_log);
    }
}

// left-to-right inside one chunk, from the box of everything left of it; the chunk's best candidate (lowest position on equal cost)
template <typename T>
__global__ void __launch_bounds__(kScanThreads) k_sweep_chunk_left(SweepCtx<T> c, uint32_t n_tasks) {
    __shared__ T wtot[kScanThreads / 64][6];
    __shared__ T wcost[kScanThreads / 64];
    __shared__ uint32_t wpos[kScanThreads / 64];
    const uint32_t task = blockIdx.x % n_tasks, axis = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    uint32_t b, e;
    if (!big_segment(c, tk.slot, b, e)) return;
    const uint32_t* ord = c.ord[axis];
    const T* cost_r = c.cost_r + size_t{axis} * c.b.n;
    Box6<T> carry = read_box(c.carry_l + (size_t{axis} * c.b.task_cap + task) * 6);
    T best = __builtin_inff();
    uint32_t best_pos = 0xFFFFFFFFu;
    const uint32_t len = tk.end - tk.begin;
    for (uint32_t base = 0; base < len; base += kScanThreads) {
        const uint32_t r = base + threadIdx.x;
        const bool in = r < len;
        const uint32_t pos = tk.begin + (in ? r : 0);
        Box6<T> v = in ? load_box(c.b.bboxes, ord[pos]) : empty_box<T>();
        v = block_scan_boxes(v, carry, wtot);
        if (in && pos + 1 < e) {
            const T cost = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(pos + 1 - b, c.b.sah_log) + cost_r[pos + 1];
            if (cost < best) { best = cost; best_pos = pos + 1; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T oc = __shfl_xor(best, off);
        const uint32_t op = __shfl_xor(best_pos, off);
        if (oc < best || (oc == best && op < best_pos)) { best = oc; best_pos = op; }
    }
    if (lane == 0) { wcost[wave] = best; wpos[wave] = best_pos; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kScanThreads / 64; ++w)
            if (wcost[w] < best || (wcost[w] == best && wpos[w] < best_pos)) { best = wcost[w]; best_pos = wpos[w]; }
        AxisBest<T> ab;
        ab.cost = best; ab.pos = best_pos; ab.pad = 0;
        c.chunk_best[size_t{axis} * c.b.task_cap + task] = ab;
    }
}

// per (segment, axis), one wave: the best of the chunks' candidates
template <typename T>
__global__ void __launch_bounds__(64) k_sweep_chunk_best(SweepCtx<T> c) {
    const uint32_t slot = blockIdx.x / 3, axis = blockIdx.x % 3;
    uint32_t b, e;
    if (!big_segment(c, slot, b, e)) return;
    const SlotState<T>& st = c.b.state[slot];
    T best = __builtin_inff();
    uint32_t best_pos = 0xFFFFFFFFu;
    for (uint32_t t = threadIdx.x; t < st.ntasks; t += 64) {
        const AxisBest<T> ab = c.chunk_best[size_t{axis} * c.b.task_cap + st.task0 + t];
        if (ab.cost < best || (ab.cost == best && ab.pos < best_pos)) { best = ab.cost; best_pos = ab.pos; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T oc = __shfl_xor(best, off);
        const uint32_t op = __shfl_xor(best_pos, off);
        if (oc < best || (oc == best && op < best_pos)) { best = oc; best_pos = op; }
    }
    if (threadIdx.x == 0) {
        AxisBest<T> ab;
        ab.cost = best; ab.pos = best_pos; ab.pad = 0;
        c.axis_best[3 * slot + axis] = ab;
    }
}

// find_best_split for one (segment, axis): sweep_sah_builder.h:68-101 as full scans.
template <typename T>
__global__ void __launch_bounds__(kScanThreads) k_sweep_axis(SweepCtx<T> c) {
    __shared__ T wtot[kScanThreads / 64][6];
    __shared__ T wcost[kScanThreads / 64];
    __shared__ uint32_t wpos[kScanThreads / 64];
    const uint32_t slot = blockIdx.x / 3, axis = blockIdx.x % 3;
    const SlotState<T>& st = c.b.state[slot];
    const ANode<T>& nd = c.b.nodes[st.node];
    const uint32_t b = nd.begin, e = nd.end;
    if (c.multi && e - b > kSweepBig) return;                 // scanned chunk by chunk (k_sweep_chunk_*)
    const uint32_t* ord = c.ord[axis];
    T* cost_r = c.cost_r + size_t{axis} * c.b.n;
    // right-to-left: cost_r[i] = half_area(box[i, e)) * (e - i), i in (b, e)
    Box6<T> carry = empty_box<T>();
    for (uint32_t base = 0; base < e - b; base += kScanThreads) {
        const uint32_t r = base + threadIdx.x;              // reversed index
        const bool in = r < e - b;
        const uint32_t pos = e - 1 - (in ? r : 0);
        Box6<T> v = in ? load_box(c.b.bboxes, ord[pos]) : empty_box<T>();
        v = block_scan_boxes(v, carry, wtot);
        if (in && pos > b) cost_r[pos] = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(e - pos, c.b.sah_log);
    }
    __syncthreads();
    // left-to-right: cost(i) = half_area(box[b, i]) * (i + 1 - b) + cost_r[i + 1], split position i + 1
    carry = empty_box<T>();
    T best = __builtin_inff();
    uint32_t best_pos = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < e - b; base += kScanThreads) {
        const uint32_t r = base + threadIdx.x;
        const bool in = r < e - b;
        const uint32_t pos = b + (in ? r : 0);
        Box6<T> v = in ? load_box(c.b.bboxes, ord[pos]) : empty_box<T>();
        v = block_scan_boxes(v, carry, wtot);
        if (in && pos + 1 < e) {
            const T cost = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(pos + 1 - b, c.b.sah_log) + cost_r[pos + 1];
            if (cost < best) { best = cost; best_pos = pos + 1; }          // earlier positions win ties (strict <)
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T oc = __shfl_xor(best, off);
        const uint32_t op = __shfl_xor(best_pos, off);
        if (oc < best || (oc == best && op < best_pos)) { best = oc; best_pos = op; }
    }
    if (lane == 0) { wcost[wave] = best; wpos[wave] = best_pos; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kScanThreads / 64; ++w)
            if (wcost[w] < best || (wcost[w] == best && wpos[w] < best_pos)) { best = wcost[w]; best_pos = wpos[w]; }
        AxisBest<T> ab;
        ab.cost = best; ab.pos = best_pos; ab.pad = 0;
        c.axis_best[3 * slot + axis] = ab;
    }
}

// try_split's decision (sweep_sah_builder.h:108-124)
template <typename T>
__global__ void __launch_bounds__(64) k_sweep_decide(SweepCtx<T> c, uint32_t n_active) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    SlotState<T>& st = c.b.state[slot];
    const ANode<T>& nd = c.b.nodes[st.node];
    const uint32_t size = nd.end - nd.begin;
    const T stay = half_area(nd.lo, nd.hi, c.b.dim) * (sah_prims<T>(size, c.b.sah_log) - c.b.sah_ratio);
    uint32_t pos = (nd.begin + nd.end + 1) / 2; T cost = stay; uint32_t axis = 0;     // :111
    for (int k = 0; k < c.b.dim; ++k) {
        const AxisBest<T> ab = c.axis_best[3 * slot + k];
        if (ab.cost < cost) { cost = ab.cost; pos = ab.pos; axis = k; }
    }
    if (cost >= stay) {                                       // size > 64 > max_leaf_size: median on the widest axis (:122-123)
        pos = (nd.begin + nd.end + 1) / 2;
        axis = widest_axis(nd.lo, nd.hi, c.b.dim);
    }
    st.split = pos;
    st.axis = axis;
    st.mode = MODE_PARTITION;
}

// mark_primitives (:103-106)
template <typename T>
__global__ void __launch_bounds__(256) k_sweep_mark(SweepCtx<T> c) {
    const Task tk = c.b.tasks[blockIdx.x];
    const SlotState<T>& st = c.b.state[tk.slot];
    const uint32_t* ord = c.ord[st.axis];
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) c.marks[ord[pos]] = pos < st.split ? 1u : 0u;
}

__device__ inline uint32_t other_axis(uint32_t axis, uint32_t j) { return j == 0 ? (axis == 0 ? 1 : 0) : (axis == 2 ? 1 : 2); }

template <typename T>
__global__ void __launch_bounds__(256) k_sweep_count(SweepCtx<T> c, uint32_t n_tasks) {
    __shared__ uint32_t total;
    const uint32_t task = blockIdx.x % n_tasks, j = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    const SlotState<T>& st = c.b.state[tk.slot];
    const uint32_t* ord = c.ord[other_axis(st.axis, j)];
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) mine += c.marks[ord[pos]];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) c.chunk_true2[size_t{j} * c.b.task_cap + task] = total;
}

// std::stable_partition of the other axes (:129-136), out of place
template <typename T>
__global__ void __launch_bounds__(256) k_sweep_scatter(SweepCtx<T> c, uint32_t n_tasks) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_sh;
    const uint32_t task = blockIdx.x % n_tasks, j = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    const SlotState<T>& st = c.b.state[tk.slot];
    const ANode<T>& nd = c.b.nodes[st.node];
    const uint32_t* ord = c.ord[other_axis(st.axis, j)];
    uint32_t* out = c.ord_tmp + size_t{j} * c.b.n;
    if (threadIdx.x == 0) base_sh = 0;
    __syncthreads();
    uint32_t before = 0;
    const uint32_t* ct = c.chunk_true2 + size_t{j} * c.b.task_cap;
    for (uint32_t t = st.task0 + threadIdx.x; t < task; t += 256) before += ct[t];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if ((threadIdx.x & 63) == 0 && before) atomicAdd(&base_sh, before);
    __syncthreads();
    uint32_t running = base_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = tk.begin; tile < tk.end; tile += 256) {
        const uint32_t pos = tile + threadIdx.x;
        const bool in = pos < tk.end;
        const uint32_t id = in ? ord[pos] : 0;
        const bool flag = in && c.marks[id] != 0;
        const uint64_t bal = __ballot(flag);
        if (lane == 0) wave_sum[wave] = __popcll(bal);
        __syncthreads();
        uint32_t excl = __popcll(bal & ((uint64_t{1} << lane) - 1)), tile_total = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) excl += wave_sum[w]; tile_total += wave_sum[w]; }
        const uint32_t t_before = running + excl;             // marked elements in [begin, pos)
        if (in) out[flag ? nd.begin + t_before : st.split + (pos - nd.begin - t_before)] = id;
        running += tile_total;
        __syncthreads();
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_sweep_copyback(SweepCtx<T> c, uint32_t n_tasks) {
    const uint32_t task = blockIdx.x % n_tasks, j = blockIdx.x / n_tasks;
    const Task tk = c.b.tasks[task];
    const SlotState<T>& st = c.b.state[tk.slot];
    uint32_t* ord = c.ord[other_axis(st.axis, j)];
    const uint32_t* in = c.ord_tmp + size_t{j} * c.b.n;
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) ord[pos] = in[pos];
}

// =====================================================================================================
// Round 4: one BLOCK builds the sweep-SAH levels of a segment of 65 .. 2048 primitives (cf. build_binned.hip: k_medium)
// =====================================================================================================
// The level loop above costs nine launches and a read-back per level whatever the work; the top-level BVH over a few thousand
// mini-tree roots (mini_tree_builder.h:249-310) is all launch latency (1M triangles, Quality::Low: ~60 launches of 3-18 us for 4096
// roots = 0.45 of the 2.25 ms). A segment of up to 2048 primitives therefore leaves the loop: its boxes and its three orders are
// loaded into LDS once and one block runs the same steps — find_best_split as a right-to-left and a left-to-right scan per (node,
// axis), one wave each (sweep_sah_builder.h:68-101), try_split's decision (:108-124), mark_primitives (:103-106),
// std::stable_partition of the other two orders (:129-136), compute_bbox of both sides over the axis-0 order with the last-zero rule,
// child creation in SATO order — for all the segment's nodes of a level, with barriers instead of kernel boundaries, until every
// piece holds <= 64 primitives (k_small_sweep_levels takes those). Nodes are appended to the node array and numbered through
// MedInfo like k_medium's.
constexpr int kSwThreads = 1024;
constexpr int kSwNodes = 256;
template <typename T> constexpr uint32_t sweep_medium_cap() { return sizeof(T) == 4 ? 2048u : 1024u; }

template <typename T>
struct SweepMediumLds {
    static constexpr uint32_t KS = sweep_medium_cap<T>();
    static constexpr int kSegs = KS / 64;
    T box[6][KS];                            // by SLOT = position in the axis-0 order at load time
    uint32_t id_of[KS];                      // slot -> primitive id
    uint16_t ord[3][KS], tmp[3][KS];         // ord[axis][position] = slot
    T cost_r[3][KS];
    uint8_t mark[KS];                        // by slot: on the left side of its node's split
    uint8_t seg_of[KS];                      // by position: index of the position's node among the level's active nodes, 255: settled
    T best_cost[kSegs][3];
    uint32_t best_pos[kSegs][3];
    uint32_t s_node[kSegs], s_begin[kSegs], s_end[kSegs], s_axis[kSegs], s_cut[kSegs];
    uint8_t s_next[kSegs][2];
    uint32_t act_next[kSegs];
    T cbox[2 * kSegs][6];
    T nbox[kSwNodes][6];
    uint16_t nb[kSwNodes], ne[kSwNodes], nparent[kSwNodes], nchild[kSwNodes];
    uint8_t nwhich[kSwNodes];
    uint16_t small_nodes[kSwNodes];
    uint32_t n_nodes, n_act, n_next, n_small, error, base, small_base, n_levels;
    uint16_t level_start[kMedLevels + 1];
};

template <typename T> __device__ inline Box6<T> lds_box(const SweepMediumLds<T>& L, uint32_t slot) {
    Box6<T> v;
#pragma unroll
    for (int k = 0; k < 3; ++k) { v.lo[k] = L.box[k][slot]; v.hi[k] = L.box[3 + k][slot]; }
    return v;
}

template <typename T>
__global__ void __launch_bounds__(kSwThreads) k_sweep_medium(SweepCtx<T> c) {
    __shared__ SweepMediumLds<T> L;
    constexpr uint32_t KS = SweepMediumLds<T>::KS;
    constexpr uint32_t kWaves = kSwThreads / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t root_id = c.b.medium_list[blockIdx.x];
    const ANode<T>& R = c.b.nodes[root_id];
    const uint32_t B = R.begin, s = R.end - R.begin, tree = R.tree;
    MedInfo* info = c.b.med_info + blockIdx.x;

    // ---- load: boxes by the axis-0 order; the other two orders as slots (through marks[], which holds the slot of every id here)
    for (uint32_t p = tid; p < KS; p += kSwThreads) {
        L.seg_of[p] = p < s ? 0 : 255;
        L.ord[0][p] = static_cast<uint16_t>(p);
        if (p < s) {
            const uint32_t id = c.ord[0][B + p];
            L.id_of[p] = id;
            c.marks[id] = p;
#pragma unroll
            for (int k = 0; k < 3; ++k) { L.box[k][p] = c.b.bboxes[6ull * id + k]; L.box[3 + k][p] = c.b.bboxes[6ull * id + 3 + k]; }
        }
    }
    if (tid < 3) { L.nbox[0][tid] = R.lo[tid]; L.nbox[0][3 + tid] = R.hi[tid]; }
    if (tid == 0) {
        L.nb[0] = 0; L.ne[0] = static_cast<uint16_t>(s); L.nparent[0] = 0; L.nwhich[0] = 0; L.nchild[0] = 0;
        L.n_nodes = 1; L.n_act = 1; L.s_node[0] = 0; L.n_small = 0; L.error = 0; L.n_levels = 1; L.level_start[0] = 0; L.level_start[1] = 1;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t p = tid; p < s; p += kSwThreads) {
        L.ord[1][p] = static_cast<uint16_t>(c.marks[c.ord[1][B + p]]);
        L.ord[2][p] = static_cast<uint16_t>(c.marks[c.ord[2][B + p]]);
    }
    __syncthreads();

    for (;;) {
        const uint32_t n_act = L.n_act;
        if (n_act == 0 || L.error) break;
        if (tid < n_act) { const uint32_t nd = L.s_node[tid]; L.s_begin[tid] = L.nb[nd]; L.s_end[tid] = L.ne[nd]; }
        __syncthreads();
        // ---- find_best_split: one wave per (node, axis): right-to-left costs, then left-to-right costs and the arg-min (:68-101)
        for (uint32_t task = wave; task < 3 * n_act; task += kWaves) {
            const uint32_t sg = task / 3, axis = task % 3;
            if (static_cast<int>(axis) >= c.b.dim) continue;
            const uint32_t b = L.s_begin[sg], e = L.s_end[sg];
            const uint16_t* ord = L.ord[axis];
            T* cost_r = L.cost_r[axis];
            Box6<T> carry = empty_box<T>();
            for (uint32_t base = 0; base < e - b; base += 64) {
                const uint32_t r = base + lane;
                const bool in = r < e - b;
                const uint32_t pos = e - 1 - (in ? r : 0);
                Box6<T> v = in ? lds_box(L, ord[pos]) : empty_box<T>();
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { Box6<T> o = shfl_up_box(v, off); if (static_cast<int>(lane) >= off) v = join(o, v); }
                v = join(carry, v);
                if (in && pos > b) cost_r[pos] = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(e - pos, c.b.sah_log);
#pragma unroll
                for (int k = 0; k < 3; ++k) { carry.lo[k] = __shfl(v.lo[k], 63); carry.hi[k] = __shfl(v.hi[k], 63); }
            }
            wave_sync();
            carry = empty_box<T>();
            T best = __builtin_inff();
            uint32_t best_pos = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < e - b; base += 64) {
                const uint32_t r = base + lane;
                const bool in = r < e - b;
                const uint32_t pos = b + (in ? r : 0);
                Box6<T> v = in ? lds_box(L, ord[pos]) : empty_box<T>();
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { Box6<T> o = shfl_up_box(v, off); if (static_cast<int>(lane) >= off) v = join(o, v); }
                v = join(carry, v);
                if (in && pos + 1 < e) {
                    const T cost = half_area(v.lo, v.hi, c.b.dim) * sah_prims<T>(pos + 1 - b, c.b.sah_log) + cost_r[pos + 1];
                    if (cost < best) { best = cost; best_pos = pos + 1; }          // earlier positions win ties (strict <)
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { carry.lo[k] = __shfl(v.lo[k], 63); carry.hi[k] = __shfl(v.hi[k], 63); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const T oc = __shfl_xor(best, off);
                const uint32_t op = __shfl_xor(best_pos, off);
                if (oc < best || (oc == best && op < best_pos)) { best = oc; best_pos = op; }
            }
            if (lane == 0) { L.best_cost[sg][axis] = best; L.best_pos[sg][axis] = best_pos; }
        }
        __syncthreads();
        // ---- try_split's decision (:108-124; k_sweep_decide)
        if (tid < n_act) {
            const uint32_t nd = L.s_node[tid], b = L.s_begin[tid], e = L.s_end[tid];
            T nlo[3], nhi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[nd][k]; nhi[k] = L.nbox[nd][3 + k]; }
            const T stay = half_area(nlo, nhi, c.b.dim) * (sah_prims<T>(e - b, c.b.sah_log) - c.b.sah_ratio);
            uint32_t pos = (b + e + 1) / 2; T cost = stay; uint32_t axis = 0;      // :111 ((B + b + B + e + 1) / 2 - B: the same number)
            for (int k = 0; k < c.b.dim; ++k)
                if (L.best_cost[tid][k] < cost) { cost = L.best_cost[tid][k]; pos = L.best_pos[tid][k]; axis = static_cast<uint32_t>(k); }
            if (cost >= stay) { pos = (b + e + 1) / 2; axis = static_cast<uint32_t>(widest_axis(nlo, nhi, c.b.dim)); }   // size > 64 > max_leaf_size (:122-123)
            L.s_cut[tid] = pos; L.s_axis[tid] = axis;
        }
        __syncthreads();
        // ---- mark_primitives (:103-106): by slot, from the split axis's order
        for (uint32_t p = tid; p < s; p += kSwThreads) {
            const uint32_t sg = L.seg_of[p];
            if (sg == 255) continue;
            L.mark[L.ord[L.s_axis[sg]][p]] = p < L.s_cut[sg] ? 1 : 0;
        }
        __syncthreads();
        // ---- std::stable_partition of the other two orders (:129-136): one wave per (node, other axis)
        for (uint32_t task = wave; task < 2 * n_act; task += kWaves) {
            const uint32_t sg = task >> 1, axis = other_axis(L.s_axis[sg], task & 1u);
            const uint32_t b = L.s_begin[sg], e = L.s_end[sg], cut = L.s_cut[sg];
            uint16_t* ord = L.ord[axis];
            uint16_t* out = L.tmp[axis];
            uint32_t running = 0;
            for (uint32_t base = b; base < e; base += 64) {
                const uint32_t pos = base + lane;
                const bool in = pos < e;
                const uint32_t slot = in ? ord[pos] : 0u;
                const bool flag = in && L.mark[slot] != 0;
                const uint64_t bal = __ballot(flag);
                const uint32_t t_before = running + __popcll(bal & ((uint64_t{1} << lane) - 1));
                if (in) out[flag ? b + t_before : cut + (pos - b - t_before)] = static_cast<uint16_t>(slot);
                running += __popcll(bal);
            }
            wave_sync();
            for (uint32_t pos = b + lane; pos < e; pos += 64) ord[pos] = out[pos];
        }
        __syncthreads();
        // ---- compute_bbox of both sides over the axis-0 order (top_down_sah_builder.h:96-97, :133-139), one wave per (node, side)
        for (uint32_t r = wave; r < 2 * n_act; r += kWaves) {
            const uint32_t sg = r >> 1, side = r & 1u;
            const uint32_t rb = side ? L.s_cut[sg] : L.s_begin[sg], re = side ? L.s_end[sg] : L.s_cut[sg];
            T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
            uint32_t zl[3] = {0, 0, 0}, zh[3] = {0, 0, 0};
            for (uint32_t p = rb + lane; p < re; p += 64) {
                const uint32_t sl = L.ord[0][p];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T a = L.box[k][sl], bq = L.box[3 + k][sl];
                    lo[k] = pick_min(lo[k], a); hi[k] = pick_max(hi[k], bq);
                    if (a == T(0)) zl[k] = (p << 1) | Ord<T>::sign(a);            // (p ascends within a lane: the last one stays)
                    if (bq == T(0)) zh[k] = (p << 1) | Ord<T>::sign(bq);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
                const uint32_t wzl = wave_max_key(zl[k]), wzh = wave_max_key(zh[k]);
                if (lane == 0) { L.cbox[r][k] = decode_bound<T>(klo, wzl); L.cbox[r][3 + k] = decode_bound<T>(khi, wzh); }
            }
        }
        __syncthreads();
        // ---- child creation with SATO order (:91-113; k_finalize), by the first wave
        const uint32_t n_before = L.n_nodes;
        if (n_before + 2 * n_act > static_cast<uint32_t>(kSwNodes) || L.n_levels + 1 > static_cast<uint32_t>(kMedLevels)) {
            if (tid == 0) L.error = 1;
            __syncthreads();
            break;
        }
        if (wave == 0) {
            const bool on = lane < n_act;
            uint32_t child = 0, big[2] = {0, 0}, rbv[2] = {0, 0}, rev[2] = {0, 0};
            int first = 0;
            if (on) {
                const uint32_t nd = L.s_node[lane];
                first = half_area(&L.cbox[2 * lane][0], &L.cbox[2 * lane][3], c.b.dim) < half_area(&L.cbox[2 * lane + 1][0], &L.cbox[2 * lane + 1][3], c.b.dim) ? 1 : 0;
                child = n_before + 2 * lane;
                rbv[0] = L.s_begin[lane]; rev[0] = L.s_cut[lane]; rbv[1] = L.s_cut[lane]; rev[1] = L.s_end[lane];
                L.nchild[nd] = static_cast<uint16_t>(child);
                for (int w = 0; w < 2; ++w) {
                    const int sd = w == 0 ? first : 1 - first;
#pragma unroll
                    for (int q = 0; q < 6; ++q) L.nbox[child + w][q] = L.cbox[2 * lane + sd][q];
                    L.nb[child + w] = static_cast<uint16_t>(rbv[sd]); L.ne[child + w] = static_cast<uint16_t>(rev[sd]);
                    L.nparent[child + w] = static_cast<uint16_t>(nd); L.nwhich[child + w] = static_cast<uint8_t>(w); L.nchild[child + w] = 0;
                }
                big[0] = rev[0] - rbv[0] > static_cast<uint32_t>(kSmall) ? 1u : 0u;
                big[1] = rev[1] - rbv[1] > static_cast<uint32_t>(kSmall) ? 1u : 0u;
            }
            const uint64_t b0 = __ballot(on && big[0]), b1 = __ballot(on && big[1]), below = (uint64_t{1} << lane) - 1;
            const uint64_t s0 = __ballot(on && !big[0]), s1 = __ballot(on && !big[1]);
            if (on) {
                uint32_t a = __popcll(b0 & below) + __popcll(b1 & below);
                uint32_t q = L.n_small + __popcll(s0 & below) + __popcll(s1 & below);
                for (int sd = 0; sd < 2; ++sd) {
                    const uint32_t node = child + (sd == first ? 0u : 1u);
                    if (big[sd]) { L.act_next[a] = node; L.s_next[lane][sd] = static_cast<uint8_t>(a); ++a; }
                    else { L.small_nodes[q] = static_cast<uint16_t>(node); L.s_next[lane][sd] = 255; ++q; }
                }
            }
            if (lane == 0) {
                L.n_next = __popcll(b0) + __popcll(b1);
                L.n_small += __popcll(s0) + __popcll(s1);
                L.n_nodes = n_before + 2 * n_act;
                L.level_start[L.n_levels + 1] = static_cast<uint16_t>(n_before + 2 * n_act);
                L.n_levels += 1;
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < s; p += kSwThreads) {
            const uint32_t sg = L.seg_of[p];
            if (sg != 255) L.seg_of[p] = L.s_next[sg][p >= L.s_cut[sg] ? 1 : 0];
        }
        __syncthreads();
        if (tid < L.n_next) L.s_node[tid] = L.act_next[tid];
        if (tid == 0) L.n_act = L.n_next;
        __syncthreads();
    }

    if (L.error) {                                             // give up: the host retries the whole build without k_sweep_medium
        if (tid == 0) { atomicOr(&c.b.counters->error, 4u); info->count = 0; info->base = 0; info->n_levels = 0; }
        return;
    }
    if (tid == 0) {
        const uint32_t extra = L.n_nodes - 1;
        L.base = atomicAdd(&c.b.counters->n_nodes, extra);
        L.small_base = atomicAdd(&c.b.counters->n_small, L.n_small);
        if (L.base + extra > c.b.node_cap || L.small_base + L.n_small > c.b.node_cap) { atomicOr(&c.b.counters->error, 2u); L.error = 1; }
    }
    __syncthreads();
    if (L.error) { if (tid == 0) { info->count = 0; info->base = 0; info->n_levels = 0; } return; }
    // ---- results: the three orders, the nodes (appended), the <= 64-primitive pieces
    for (uint32_t p = tid; p < s; p += kSwThreads) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c.ord[k][B + p] = L.id_of[L.ord[k][p]];
    }
    const uint32_t base = L.base, n_nodes = L.n_nodes;
    auto global_id = [&](uint32_t t) { return t == 0 ? root_id : base + t - 1; };
    for (uint32_t t = 1 + tid; t < n_nodes; t += kSwThreads) {
        ANode<T> nd;
#pragma unroll
        for (int k = 0; k < 3; ++k) { nd.lo[k] = L.nbox[t][k]; nd.hi[k] = L.nbox[t][3 + k]; }
        nd.begin = B + L.nb[t]; nd.end = B + L.ne[t];
        nd.child = L.nchild[t] ? global_id(L.nchild[t]) : kNone;
        nd.parent = global_id(L.nparent[t]) | (static_cast<uint32_t>(L.nwhich[t]) << 31);
        nd.kind = L.nchild[t] ? KIND_INNER : KIND_SMALL;
        nd.ic = 0; nd.rank = 0; nd.tree = tree;
        c.b.nodes[global_id(t)] = nd;
    }
    for (uint32_t q = tid; q < L.n_small; q += kSwThreads) c.b.small_list[L.small_base + q] = global_id(L.small_nodes[q]);
    if (tid == 0) {
        ANode<T>& root = c.b.nodes[root_id];
        root.child = global_id(L.nchild[0]);
        root.kind = KIND_INNER;
        info->base = base; info->count = n_nodes; info->n_levels = L.n_levels;
    }
    for (uint32_t q = tid; q <= L.n_levels; q += kSwThreads) info->level_start[q] = L.level_start[q];
}

// =====================================================================================================
// Segments <= 64 primitives: one wavefront, slot = lane holds a primitive, perm[k][position] = slot
// =====================================================================================================

template <typename T>
struct WaveSweepLds {
    uint32_t perm[3][kSmall];
    uint32_t next[kSmall];
    uint32_t markslot[kSmall];
    T nbox[2 * kSmall][6];
    uint32_t stack[kSmall + 4];
};

template <typename T>
__global__ void __launch_bounds__(256) k_small_sweep(SweepCtx<T> c, uint32_t n_small) {
    __shared__ WaveSweepLds<T> lds_all[4];
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    WaveSweepLds<T>& L = lds_all[threadIdx.x >> 6];
    ANode<T>& A = c.b.nodes[c.b.small_list[w]];
    const uint32_t B = A.begin, s = A.end - A.begin;
    HostNode<T>* stage = c.b.stage + 2ull * B;
    const uint64_t lanes_below = (uint64_t{1} << lane) - 1;

    // slot `lane` holds the primitive at position B + lane of the axis-0 order
    uint32_t id = 0xFFFFFFFFu;
    T blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
    if (lane < int(s)) {
        id = c.ord[0][B + lane];
#pragma unroll
        for (int k = 0; k < 3; ++k) { blo[k] = c.b.bboxes[6ull * id + k]; bhi[k] = c.b.bboxes[6ull * id + 3 + k]; }
    }
    L.perm[0][lane] = lane;
    for (int k = 1; k < 3; ++k) {
        const uint32_t want = lane < int(s) ? c.ord[k][B + lane] : 0xFFFFFFFEu;
        uint32_t slot = lane;
        for (uint32_t t = 0; t < s; ++t) if (__shfl(id, int(t)) == want) slot = t;
        L.perm[k][lane] = slot;
    }
    if (lane < 3) { L.nbox[0][lane] = A.lo[lane]; L.nbox[0][3 + lane] = A.hi[lane]; }
    if (lane == 0) L.stack[0] = pack_item(0, 0, s);
    uint32_t sp = 1, ncount = 1;

    while (sp > 0) {
        wave_sync();
        const uint32_t item = L.stack[--sp];
        const uint32_t ln = item & 0xFF, lb = (item >> 8) & 0xFF, le = (item >> 16) & 0xFF;
        const uint32_t cnt = le - lb;
        T nlo[3], nhi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[ln][k]; nhi[k] = L.nbox[ln][3 + k]; }
        const bool in = uint32_t(lane) >= lb && uint32_t(lane) < le;      // lane as a POSITION
        bool split = false;
        uint32_t cut = 0;

        if (cnt > c.b.min_leaf) {
            const T stay = half_area(nlo, nhi, c.b.dim) * (sah_prims<T>(cnt, c.b.sah_log) - c.b.sah_ratio);
            uint32_t best_pos = (B + lb + B + le + 1) / 2 - B; T best_cost = stay; uint32_t best_axis = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= c.b.dim) continue;                   // (wave-uniform)
                const int src = L.perm[k][lane];
                Box6<T> v;
#pragma unroll
                for (int q = 0; q < 3; ++q) { v.lo[q] = __shfl(blo[q], src); v.hi[q] = __shfl(bhi[q], src); }
                if (!in) v = empty_box<T>();
                // suffix scan (inclusive, towards higher positions) and prefix scan
                Box6<T> suf = v, pre = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    Box6<T> o;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { o.lo[q] = __shfl_down(suf.lo[q], off); o.hi[q] = __shfl_down(suf.hi[q], off); }
                    if (lane + off < 64) suf = join(o, suf);
#pragma unroll
                    for (int q = 0; q < 3; ++q) { o.lo[q] = __shfl_up(pre.lo[q], off); o.hi[q] = __shfl_up(pre.hi[q], off); }
                    if (lane >= off) pre = join(o, pre);
                }
                const T cr = half_area(suf.lo, suf.hi, c.b.dim) * sah_prims<T>(le - uint32_t(lane), c.b.sah_log);    // cost of [lane, le)
                const T cr_next = __shfl_down(cr, 1);
                T cost = __builtin_inff();
                uint32_t pos = 0xFFFFFFFFu;
                if (in && uint32_t(lane) + 1 < le) {
                    const T cl = half_area(pre.lo, pre.hi, c.b.dim) * sah_prims<T>(uint32_t(lane) + 1 - lb, c.b.sah_log);
                    const T tot = cl + cr_next;
                    if (tot < cost) { cost = tot; pos = lane + 1; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const T oc = __shfl_xor(cost, off);
                    const uint32_t op = __shfl_xor(pos, off);
                    if (oc < cost || (oc == cost && op < pos)) { cost = oc; pos = op; }
                }
                if (cost < best_cost) { best_cost = cost; best_pos = pos; best_axis = k; }
            }
            bool do_split = true;
            if (best_cost >= stay) {
                if (cnt <= c.b.max_leaf) do_split = false;
                else { best_pos = (B + lb + B + le + 1) / 2 - B; best_axis = widest_axis(nlo, nhi, c.b.dim); }
            }
            if (do_split) {
                // mark_primitives + stable_partition of the other two axes
                L.markslot[L.perm[best_axis][lane]] = (in && uint32_t(lane) < best_pos) ? 1u : 0u;
                wave_sync();
                for (uint32_t k = 0; k < 3; ++k) {
                    if (k == best_axis) continue;
                    const uint32_t slot = L.perm[k][lane];
                    const bool flag = in && L.markslot[slot] != 0;
                    const uint64_t tmask = __ballot(flag), fmask = __ballot(in && !flag);
                    if (in) L.next[flag ? lb + __popcll(tmask & lanes_below) : best_pos + __popcll(fmask & lanes_below)] = slot;
                    wave_sync();
                    if (in) L.perm[k][lane] = L.next[lane];
                    wave_sync();
                }
                split = true;
                cut = best_pos;
            }
        }

        HostNode<T> rec;
#pragma unroll
        for (int k = 0; k < 3; ++k) { rec.bounds[2 * k] = nlo[k]; rec.bounds[2 * k + 1] = nhi[k]; }
        if (split) {
            // compute_bbox of both children in axis-0 position order
            const int src = L.perm[0][lane];
            T plo[3], phi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { plo[q] = __shfl(blo[q], src); phi[q] = __shfl(bhi[q], src); }
            T lo[2][3], hi[2][3];
            const bool left = in && uint32_t(lane) < cut, right = in && uint32_t(lane) >= cut;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[0][k] = left ? plo[k] : Ord<T>::kMax;  hi[0][k] = left ? phi[k] : -Ord<T>::kMax;
                lo[1][k] = right ? plo[k] : Ord<T>::kMax; hi[1][k] = right ? phi[k] : -Ord<T>::kMax;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        lo[q][k] = pick_min(lo[q][k], __shfl_xor(lo[q][k], off));
                        hi[q][k] = pick_max(hi[q][k], __shfl_xor(hi[q][k], off));
                    }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) {                  // sign of a zero bound = last zero in position order
                    const bool side = q == 0 ? left : right;
                    if (lo[q][k] == T(0)) {
                        const uint64_t zm = __ballot(side && plo[k] == T(0));
                        lo[q][k] = Ord<T>::zero(__shfl(Ord<T>::sign(plo[k]), 63 - __clzll(zm)));
                    }
                    if (hi[q][k] == T(0)) {
                        const uint64_t zm = __ballot(side && phi[k] == T(0));
                        hi[q][k] = Ord<T>::zero(__shfl(Ord<T>::sign(phi[k]), 63 - __clzll(zm)));
                    }
                }
            const int first = half_area(lo[0], hi[0], c.b.dim) < half_area(lo[1], hi[1], c.b.dim) ? 1 : 0;
            const uint32_t child = ncount;
            ncount += 2;
            const uint32_t rb[2] = { lb, cut }, re[2] = { cut, le };
            if (lane < 3) {
                L.nbox[child][lane] = lo[first][lane];         L.nbox[child][3 + lane] = hi[first][lane];
                L.nbox[child + 1][lane] = lo[1 - first][lane]; L.nbox[child + 1][3 + lane] = hi[1 - first][lane];
            }
            uint32_t ia = pack_item(child, rb[first], re[first]), ib = pack_item(child + 1, rb[1 - first], re[1 - first]);
            if (re[first] - rb[first] < re[1 - first] - rb[1 - first]) { const uint32_t t = ia; ia = ib; ib = t; }
            if (lane == 0) { L.stack[sp] = ia; L.stack[sp + 1] = ib; }
            sp += 2;
            rec.index = static_cast<typename IndexOf<T>::Type>(child) << kCountBits;
        } else {
            rec.index = (static_cast<typename IndexOf<T>::Type>(B + lb) << kCountBits) | cnt;
        }
        if (lane == 0) stage[ln] = rec;
    }
    wave_sync();
    // prim_ids = the axis-0 order (sweep_sah_builder.h:66)
    const uint32_t out_id = __shfl(id, int(L.perm[0][lane]));
    if (lane < int(s)) c.ord[0][B + lane] = out_id;
    if (lane == 0) A.ic = (ncount - 1) / 2;
}


// ---- the same, level-synchronous (round 2; cf. build_binned.hip: k_small_levels) ------------------------------------------------
// All nodes of one level of the subtree are split at once. The sweeps become SEGMENTED scans over the nodes' contiguous position
// ranges (a scan step only joins a neighbour that lies inside the lane's own node), the arg-min over a node is a segmented min-scan
// read at the node's last position, the stable partitions of the other two orders use one ballot masked with the node's range, the
// child boxes are LDS atomics per child with the last-zero rule, and the reference's numbering is recovered at the end from inner
// counts (bottom-up over the levels) and ranks (top-down): fewer primitives first, ties: the second child.
template <typename T>
struct LevelSweepLds {
    uint32_t perm[3][kSmall];
    uint32_t next[kSmall];
    uint32_t markslot[kSmall];
    T nbox[2 * kSmall][6];
    typename Ord<T>::U cbox_lo[kSmall][3], cbox_hi[kSmall][3];
    uint32_t czlo[kSmall][3], czhi[kSmall][3];
    uint8_t nb[2 * kSmall], ne[2 * kSmall], nparent[2 * kSmall], nwhich[2 * kSmall], nchild[2 * kSmall], nfirst[2 * kSmall], nic[2 * kSmall], nrank[2 * kSmall];
    uint8_t level_start[kSmall + 2];
};

template <typename T>
__global__ void __launch_bounds__(128) k_small_sweep_levels(SweepCtx<T> c, uint32_t n_small) {
    __shared__ LevelSweepLds<T> lds_all[2];
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 2 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    LevelSweepLds<T>& L = lds_all[threadIdx.x >> 6];
    ANode<T>& A = c.b.nodes[c.b.small_list[w]];
    const uint32_t B = A.begin, s = A.end - A.begin;
    HostNode<T>* stage = c.b.stage + 2ull * B;
    const uint64_t lanes_below = (uint64_t{1} << lane) - 1;
    using I = typename IndexOf<T>::Type;

    // slot `lane` holds the primitive at position B + lane of the axis-0 order
    uint32_t id = 0xFFFFFFFFu;
    T blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
    if (lane < int(s)) {
        id = c.ord[0][B + lane];
#pragma unroll
        for (int k = 0; k < 3; ++k) { blo[k] = c.b.bboxes[6ull * id + k]; bhi[k] = c.b.bboxes[6ull * id + 3 + k]; }
    }
    L.perm[0][lane] = lane;
    for (int k = 1; k < 3; ++k) {
        const uint32_t want = lane < int(s) ? c.ord[k][B + lane] : 0xFFFFFFFEu;
        uint32_t slot = lane;
        for (uint32_t t = 0; t < s; ++t) if (__shfl(id, int(t)) == want) slot = t;
        L.perm[k][lane] = slot;
    }
    if (lane < 3) { L.nbox[0][lane] = A.lo[lane]; L.nbox[0][3 + lane] = A.hi[lane]; }
    if (lane == 0) { L.nb[0] = 0; L.ne[0] = static_cast<uint8_t>(s); L.nparent[0] = 0; L.nwhich[0] = 0; L.nchild[0] = 0; L.nfirst[0] = 0; }
    const bool in_tree = lane < int(s);
    uint32_t my_node = 0;                                     // node that holds POSITION `lane` (positions, not slots, belong to nodes)
    uint32_t n_nodes = 1, n_levels = 0, lvl_begin = 0, lvl_end = 1;
    wave_sync();

    while (lvl_begin < lvl_end) {
        if (lane == 0) L.level_start[n_levels] = static_cast<uint8_t>(lvl_begin);
        ++n_levels;
        const bool mine = in_tree && my_node >= lvl_begin;                 // my node belongs to this level (older nodes are leaves)
        const uint32_t lb = mine ? L.nb[my_node] : 0u, le = mine ? L.ne[my_node] : 0u;
        const uint32_t cnt = le - lb;
        T nlo[3] = {0, 0, 0}, nhi[3] = {0, 0, 0};
        if (mine) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[my_node][k]; nhi[k] = L.nbox[my_node][3 + k]; }
        }
        const bool tries = mine && cnt > c.b.min_leaf;
        const uint64_t my_range = mine ? (((le >= 64 ? ~uint64_t{0} : ((uint64_t{1} << le) - 1))) & ~((uint64_t{1} << lb) - 1)) : 0;
        bool split = false;
        uint32_t cut = 0;
        {
            const T stay = half_area(nlo, nhi, c.b.dim) * (sah_prims<T>(cnt, c.b.sah_log) - c.b.sah_ratio);
            const uint32_t mid = (B + lb + B + le + 1) / 2 - B;
            uint32_t best_pos = mid; T best_cost = stay; uint32_t best_axis = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= c.b.dim) continue;                   // (wave-uniform)
                const int src = L.perm[k][lane];
                Box6<T> v;
#pragma unroll
                for (int q = 0; q < 3; ++q) { v.lo[q] = __shfl(blo[q], src); v.hi[q] = __shfl(bhi[q], src); }
                if (!tries) v = empty_box<T>();
                // segmented suffix scan (towards higher positions) and prefix scan inside [lb, le)
                Box6<T> suf = v, pre = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    Box6<T> o;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { o.lo[q] = __shfl_down(suf.lo[q], off); o.hi[q] = __shfl_down(suf.hi[q], off); }
                    if (uint32_t(lane + off) < le) suf = join(o, suf);
#pragma unroll
                    for (int q = 0; q < 3; ++q) { o.lo[q] = __shfl_up(pre.lo[q], off); o.hi[q] = __shfl_up(pre.hi[q], off); }
                    if (lane >= off && uint32_t(lane - off) >= lb && tries) pre = join(o, pre);
                }
                const T cr = half_area(suf.lo, suf.hi, c.b.dim) * sah_prims<T>(le - uint32_t(lane), c.b.sah_log);    // cost of [lane, le)
                const T cr_next = __shfl_down(cr, 1);
                T cost = __builtin_inff();
                uint32_t pos = 0xFFFFFFFFu;
                if (tries && uint32_t(lane) + 1 < le) {
                    const T cl = half_area(pre.lo, pre.hi, c.b.dim) * sah_prims<T>(uint32_t(lane) + 1 - lb, c.b.sah_log);
                    const T tot = cl + cr_next;
                    if (tot < cost) { cost = tot; pos = lane + 1; }
                }
                // segmented arg-min: inclusive min-scan over [lb, lane], the node's result sits at position le - 1
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const T oc = __shfl_up(cost, off);
                    const uint32_t op = __shfl_up(pos, off);
                    if (lane >= off && uint32_t(lane - off) >= lb && tries && (oc < cost || (oc == cost && op < pos))) { cost = oc; pos = op; }
                }
                const int last = tries ? int(le) - 1 : lane;
                cost = __shfl(cost, last);
                pos = __shfl(pos, last);
                if (cost < best_cost) { best_cost = cost; best_pos = pos; best_axis = k; }
            }
            bool do_split = tries;
            if (tries && best_cost >= stay) {
                if (cnt <= c.b.max_leaf) do_split = false;
                else { best_pos = mid; best_axis = static_cast<uint32_t>(widest_axis(nlo, nhi, c.b.dim)); }
            }
            // mark_primitives + stable_partition of the other two orders, every splitting node at once
            L.markslot[lane] = 0;
            wave_sync();
            if (do_split) L.markslot[L.perm[best_axis][lane]] = uint32_t(lane) < best_pos ? 1u : 0u;
            wave_sync();
            for (uint32_t k = 0; k < 3; ++k) {
                const bool act = do_split && k != best_axis;
                const uint32_t slot = L.perm[k][lane];
                const bool flag = act && L.markslot[slot] != 0;
                const uint64_t tmask = __ballot(flag) & my_range, fmask = __ballot(act && !flag) & my_range;
                if (act) L.next[flag ? lb + __popcll(tmask & lanes_below) : best_pos + __popcll(fmask & lanes_below)] = slot;
                wave_sync();
                if (act) L.perm[k][lane] = L.next[lane];
                wave_sync();
            }
            split = do_split;
            cut = best_pos;
        }
        // ---- the children: ids in position order of the splitting nodes, boxes in axis-0 position order (compute_bbox)
        const bool creates = split && uint32_t(lane) == lb;
        const uint64_t cm = __ballot(creates);
        const uint32_t new_first = n_nodes;
        if (creates) L.nchild[my_node] = static_cast<uint8_t>(new_first + 2 * __popcll(cm & lanes_below));
        for (uint32_t q = lane; q < 2 * static_cast<uint32_t>(__popcll(cm)) * 3; q += 64) {
            (&L.cbox_lo[0][0])[q] = Ord<T>::enc(Ord<T>::kMax); (&L.cbox_hi[0][0])[q] = Ord<T>::enc(-Ord<T>::kMax);
            (&L.czlo[0][0])[q] = 0; (&L.czhi[0][0])[q] = 0;
        }
        wave_sync();
        const uint32_t side = split && uint32_t(lane) >= cut ? 1u : 0u;
        {
            const int src = L.perm[0][lane];
            T plo[3], phi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { plo[q] = __shfl(blo[q], src); phi[q] = __shfl(bhi[q], src); }
            if (split) {
                const uint32_t cb = L.nchild[my_node] - new_first + side;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    atomicMin(&L.cbox_lo[cb][k], Ord<T>::enc(plo[k])); atomicMax(&L.cbox_hi[cb][k], Ord<T>::enc(phi[k]));
                    if (plo[k] == T(0)) atomicMax(&L.czlo[cb][k], (static_cast<uint32_t>(lane) << 1) | Ord<T>::sign(plo[k]));
                    if (phi[k] == T(0)) atomicMax(&L.czhi[cb][k], (static_cast<uint32_t>(lane) << 1) | Ord<T>::sign(phi[k]));
                }
            }
        }
        wave_sync();
        if (creates) {
            const uint32_t child = L.nchild[my_node], cb = child - new_first;
            T clo[2][3], chi[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    clo[q][k] = decode_bound<T>(L.cbox_lo[cb + q][k], L.czlo[cb + q][k]);
                    chi[q][k] = decode_bound<T>(L.cbox_hi[cb + q][k], L.czhi[cb + q][k]);
                }
            const int first = half_area(clo[0], chi[0], c.b.dim) < half_area(clo[1], chi[1], c.b.dim) ? 1 : 0;     // SATO
            const uint32_t rb[2] = { lb, cut }, re[2] = { cut, le };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                L.nbox[child][k] = clo[first][k];         L.nbox[child][3 + k] = chi[first][k];
                L.nbox[child + 1][k] = clo[1 - first][k]; L.nbox[child + 1][3 + k] = chi[1 - first][k];
            }
            L.nb[child] = static_cast<uint8_t>(rb[first]);         L.ne[child] = static_cast<uint8_t>(re[first]);
            L.nb[child + 1] = static_cast<uint8_t>(rb[1 - first]); L.ne[child + 1] = static_cast<uint8_t>(re[1 - first]);
            L.nparent[child] = static_cast<uint8_t>(my_node); L.nparent[child + 1] = static_cast<uint8_t>(my_node);
            L.nwhich[child] = 0; L.nwhich[child + 1] = 1;
            L.nchild[child] = 0; L.nchild[child + 1] = 0;
            L.nfirst[my_node] = static_cast<uint8_t>(first);
        }
        n_nodes += 2 * __popcll(cm);
        wave_sync();
        if (split) my_node = L.nchild[my_node] + (side != L.nfirst[my_node] ? 1u : 0u);
        wave_sync();
        lvl_begin = lvl_end;
        lvl_end = n_nodes;
    }
    if (lane == 0) L.level_start[n_levels] = static_cast<uint8_t>(n_nodes);
    wave_sync();
    // ---- the reference's numbering (cf. k_small_levels)
    for (uint32_t lv = n_levels; lv-- > 0;) {
        const uint32_t f = L.level_start[lv], e = L.level_start[lv + 1];
        for (uint32_t t = f + lane; t < e; t += 64) {
            const uint32_t ch = L.nchild[t];
            L.nic[t] = ch ? static_cast<uint8_t>(1 + L.nic[ch] + L.nic[ch + 1]) : 0;
        }
        wave_sync();
    }
    if (lane == 0) L.nrank[0] = 0;
    wave_sync();
    for (uint32_t lv = 0; lv < n_levels; ++lv) {
        const uint32_t f = L.level_start[lv], e = L.level_start[lv + 1];
        for (uint32_t t = f + lane; t < e; t += 64) {
            const uint32_t ch = L.nchild[t];
            if (ch) {
                const uint32_t c0 = L.ne[ch] - L.nb[ch], c1 = L.ne[ch + 1] - L.nb[ch + 1];
                const uint32_t r = L.nrank[t];
                if (c0 < c1) { L.nrank[ch] = static_cast<uint8_t>(r + 1); L.nrank[ch + 1] = static_cast<uint8_t>(r + 1 + L.nic[ch]); }
                else         { L.nrank[ch + 1] = static_cast<uint8_t>(r + 1); L.nrank[ch] = static_cast<uint8_t>(r + 1 + L.nic[ch + 1]); }
            }
        }
        wave_sync();
    }
    for (uint32_t t = lane; t < n_nodes; t += 64) {
        HostNode<T> rec;
#pragma unroll
        for (int k = 0; k < 3; ++k) { rec.bounds[2 * k] = L.nbox[t][k]; rec.bounds[2 * k + 1] = L.nbox[t][3 + k]; }
        const uint32_t ch = L.nchild[t];
        if (ch) rec.index = static_cast<I>(1 + 2 * L.nrank[t]) << kCountBits;
        else rec.index = (static_cast<I>(B + L.nb[t]) << kCountBits) | static_cast<I>(L.ne[t] - L.nb[t]);
        const uint32_t fid = t == 0 ? 0u : 1u + 2u * L.nrank[L.nparent[t]] + L.nwhich[t];
        stage[fid] = rec;
    }
    // prim_ids = the axis-0 order (sweep_sah_builder.h:66)
    const uint32_t out_id = __shfl(id, int(L.perm[0][lane]));
    if (lane < int(s)) c.ord[0][B + lane] = out_id;
    if (lane == 0) A.ic = L.nic[0];
}

} // namespace

// SweepSahBuilder core: nodes in the reference layout into `final_nodes`, prim ids (= the axis-0 order) in ord[0..n).
template <typename T>
int sweep_core(const T* d_bboxes, const T* d_centers, size_t n, uint32_t min_leaf, uint32_t max_leaf,
               DevBuf<HostNode<T>>& final_nodes, DevBuf<uint32_t>& ord, size_t& total_nodes, hipStream_t stream, int dim)
{
    if (n >= (size_t{1} << 28)) return fail(BVH_AMD_ERR_UNSUPPORTED, "build: more than 2^28 primitives");
    const uint32_t n32 = static_cast<uint32_t>(n);
    BVH_HIP_TRY(ord.alloc(3 * n), BVH_AMD_ERR_HIP);
    int rc = std_sort_ids<T>(ord.p, d_centers, n32, 3, 1, 3, stream);          // :57-63
    if (rc) return rc;
    DevBuf<uint32_t> sorted;                                   // a capacity retry restarts from the sorted orders
    BVH_HIP_TRY(sorted.alloc(3 * n), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemcpyAsync(sorted.p, ord.p, 3 * n * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);

    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt) BVH_HIP_TRY(hipMemcpyAsync(ord.p, sorted.p, 3 * n * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
        const uint32_t node_cap = attempt == 0 ? n32 / 8 + 1024 : 2 * n32 + 2;
        const uint32_t slot_cap = n32 / (kSmall + 1) + 2;
        const uint32_t task_cap = n32 / kChunk + slot_cap + 2;
        DevBuf<uint32_t> ord_tmp, marks, chunk_true2, small_list;
        DevBuf<T> cost_r, chunk_box, carry_l, carry_r;
        DevBuf<AxisBest<T>> axis_best, chunk_best;
        DevBuf<ANode<T>> nodes;
        DevBuf<SlotState<T>> st_a, st_b;
        DevBuf<Task> tk_a, tk_b;
        DevBuf<HostNode<T>> stage;
        DevBuf<Counters> counters;
        hipError_t e = hipSuccess;
        auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        A(ord_tmp.alloc(2 * n)); A(marks.alloc(n)); A(chunk_true2.alloc(2 * size_t{task_cap})); A(small_list.alloc(node_cap));
        A(cost_r.alloc(3 * n)); A(axis_best.alloc(3 * size_t{slot_cap})); A(nodes.alloc(node_cap));
        A(st_a.alloc(slot_cap)); A(st_b.alloc(slot_cap)); A(tk_a.alloc(task_cap)); A(tk_b.alloc(task_cap));
        A(stage.alloc(2 * n)); A(counters.alloc(1));
        A(chunk_box.alloc(18 * size_t{task_cap})); A(carry_l.alloc(18 * size_t{task_cap})); A(carry_r.alloc(18 * size_t{task_cap})); A(chunk_best.alloc(3 * size_t{task_cap}));
        // segments of 65 .. 2048 primitives are finished by k_sweep_medium (first attempt only: a retry takes the plain path)
        static const bool sw_medium_off = BVH_DEV_INT("BVH_AMD_SWEEP_MEDIUM", 1) == 0;   // A/B runs
        const uint32_t medium_slots = n32 / (kSmall + 1) + 3;
        DevBuf<uint32_t> medium_list;
        DevBuf<MedInfo> med_info;
        const bool use_medium = attempt == 0 && !sw_medium_off;
        if (use_medium) { A(medium_list.alloc(4 * size_t{medium_slots})); A(med_info.alloc(4 * size_t{medium_slots})); }
        if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));

        SweepCtx<T> sc;
        BuildCtx<T>& c = sc.b;
        c.bboxes = d_bboxes; c.centers = d_centers; c.ids = ord.p; c.n = n32;
        c.min_leaf = min_leaf; c.max_leaf = max_leaf;
        c.dim = dim;
        c.sah_log = ambient_sah().log_cluster; c.sah_ratio = static_cast<T>(ambient_sah().cost_ratio);
        c.nodes = nodes.p; c.node_cap = node_cap; c.bins = nullptr; c.state = st_a.p; c.state_next = st_b.p; c.slot_cap = slot_cap;
        c.tasks = tk_a.p; c.tasks_next = tk_b.p; c.task_cap = task_cap; c.chunk_true = nullptr;
        c.ltab = nullptr; c.rtab = nullptr; c.small_list = small_list.p; c.stage = stage.p; c.counters = counters.p;
        for (int k = 0; k < 3; ++k) sc.ord[k] = ord.p + size_t{n} * k;
        sc.ord_tmp = ord_tmp.p; sc.marks = marks.p; sc.cost_r = cost_r.p; sc.axis_best = axis_best.p; sc.chunk_true2 = chunk_true2.p;
        sc.chunk_box = chunk_box.p; sc.carry_l = carry_l.p; sc.carry_r = carry_r.p; sc.chunk_best = chunk_best.p; sc.multi = 0;
        static const bool multi_off = BVH_DEV_INT("BVH_AMD_SWEEP_MULTI", 1) == 0;   // A/B runs
        c.big_threshold = multi_off ? 0u : kSweepBig;
        if (use_medium) {
            c.medium_cap = sweep_medium_cap<T>(); c.medium_slots = medium_slots; c.medium_min_class = 3;      // classes 0 (<= 256 primitives) and 3: one kernel serves both
            c.medium_list = medium_list.p; c.med_info = med_info.p;
        }

        hipLaunchKernelGGL(k_prepare_root<T>, dim3(1), dim3(1), 0, stream, c);
        const unsigned root_grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 2048));
        hipLaunchKernelGGL(k_init_root<T>, dim3(root_grid), dim3(256), 0, stream, c, false);
        hipLaunchKernelGGL(k_make_root<T>, dim3(1), dim3(1), 0, stream, c);
        Counters h;
        { int rb_ = readback(&h, counters.p, sizeof(h), stream); if (rb_) return rb_; }

        std::vector<uint32_t> level_start{0, 1};
        uint32_t n_active = h.n_active_next, n_tasks = h.n_tasks_next, n_big = h.n_big_next;
        bool overflow = h.error != 0;
        while (n_active > 0 && !overflow) {
            std::swap(c.state, c.state_next);
            std::swap(c.tasks, c.tasks_next);
            BVH_HIP_TRY(hipMemsetAsync(&counters.p->n_active_next, 0, 3 * sizeof(uint32_t), stream), BVH_AMD_ERR_HIP);
            const unsigned slot_grid = (n_active + 63) / 64;
            hipLaunchKernelGGL(k_init_slots<T>, dim3(n_active), dim3(64), 0, stream, c);
            sc.multi = n_big ? 1u : 0u;
            if (n_big) {                                      // segments beyond kSweepBig primitives: one block per chunk instead of one per segment
                hipLaunchKernelGGL(k_sweep_chunk_box<T>, dim3(3 * n_tasks), dim3(256), 0, stream, sc, n_tasks);
                hipLaunchKernelGGL(k_sweep_chunk_carry<T>, dim3(3 * n_active), dim3(64), 0, stream, sc);
                hipLaunchKernelGGL(k_sweep_chunk_right<T>, dim3(3 * n_tasks), dim3(kScanThreads), 0, stream, sc, n_tasks);
                hipLaunchKernelGGL(k_sweep_chunk_left<T>, dim3(3 * n_tasks), dim3(kScanThreads), 0, stream, sc, n_tasks);
                hipLaunchKernelGGL(k_sweep_chunk_best<T>, dim3(3 * n_active), dim3(64), 0, stream, sc);
            }
            hipLaunchKernelGGL(k_sweep_axis<T>, dim3(3 * n_active), dim3(kScanThreads), 0, stream, sc);
            hipLaunchKernelGGL(k_sweep_decide<T>, dim3(slot_grid), dim3(64), 0, stream, sc, n_active);
            hipLaunchKernelGGL(k_sweep_mark<T>, dim3(n_tasks), dim3(256), 0, stream, sc);
            hipLaunchKernelGGL(k_sweep_count<T>, dim3(2 * n_tasks), dim3(256), 0, stream, sc, n_tasks);
            hipLaunchKernelGGL(k_sweep_scatter<T>, dim3(2 * n_tasks), dim3(256), 0, stream, sc, n_tasks);
            hipLaunchKernelGGL(k_sweep_copyback<T>, dim3(2 * n_tasks), dim3(256), 0, stream, sc, n_tasks);
            hipLaunchKernelGGL(k_child_bounds<T>, dim3(n_tasks), dim3(256), 0, stream, c);
            hipLaunchKernelGGL(k_finalize<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
            BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
            { int rb_ = readback(&h, counters.p, sizeof(h), stream); if (rb_) return rb_; }
            overflow = h.error != 0;
            level_start.push_back(h.n_nodes);
            n_active = h.n_active_next;
            n_tasks = h.n_tasks_next;
            n_big = h.n_big_next;
        }
        if (overflow) {
            if (attempt == 0) continue;
            return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
        }
        uint32_t n_medium[4] = { h.n_medium[0], h.n_medium[1], h.n_medium[2], h.n_medium[3] };
        if (n_medium[0] + n_medium[3]) {
            for (uint32_t cls : {0u, 3u}) {
                if (!n_medium[cls]) continue;
                SweepCtx<T> scc = sc;
                scc.b.medium_list = c.medium_list + size_t{cls} * c.medium_slots;
                scc.b.med_info = c.med_info + size_t{cls} * c.medium_slots;
                hipLaunchKernelGGL(k_sweep_medium<T>, dim3(n_medium[cls]), dim3(kSwThreads), 0, stream, scc);
            }
            BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
            { int rb_ = readback(&h, counters.p, sizeof(h), stream); if (rb_) return rb_; }
            if (h.error != 0) {
                if (attempt == 0) continue;
                return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
            }
        }
        const uint32_t n_nodes_a = h.n_nodes, n_small = h.n_small;
        if (n_small) {
            static const bool dfs = BVH_DEV_IS("BVH_AMD_SMALL", "dfs");   // the node-by-node walk (A/B runs)
            if (dfs) hipLaunchKernelGGL(k_small_sweep<T>, dim3((n_small + 3) / 4), dim3(256), 0, stream, sc, n_small);
            else hipLaunchKernelGGL(k_small_sweep_levels<T>, dim3((n_small + 1) / 2), dim3(128), 0, stream, sc, n_small);
        }
        BvhImpl<T> sizes;                                      // only its node vector length is used
        rc = number_and_emit<T>(sizes, c, level_start, n_nodes_a, n_small, final_nodes, stream, n_medium);
        if (rc) return rc;
        total_nodes = sizes.node_count;
        if (!scratch_pool_enabled()) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // plain hipFree of the workspace on return (the pool frees in stream order)
        return BVH_AMD_OK;
    }
    return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
}

// SweepSahBuilder::build on the device.
template <typename T> int reinsertion_optimize_device(HostNode<T>* d_nodes, size_t node_count, hipStream_t stream, int dim);

template <typename T>
int build_sweep_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                       bool optimize, hipStream_t stream)
{
    BVH_HIP_TRY(hipGetDevice(&out.device), BVH_AMD_ERR_HIP);
    DevBuf<HostNode<T>> final_nodes;
    DevBuf<uint32_t> ord;
    size_t total_nodes = 0;
    int rc = sweep_core<T>(d_bboxes, d_centers, n, static_cast<uint32_t>(cfg.min_leaf_size), static_cast<uint32_t>(cfg.max_leaf_size),
                           final_nodes, ord, total_nodes, stream, out.dim);
    if (rc) return rc;
    if (optimize) {                                           // DefaultBuilder serial High (default_builder.h:58-60)
        rc = reinsertion_optimize_device<T>(final_nodes.p, total_nodes, stream, out.dim);
        if (rc) return rc;
    }
    out.node_count = total_nodes;
    return finish_build<T>(out, final_nodes, ord.p, n, stream, /*take_ids=*/false);
}

template int sweep_core<float>(const float*, const float*, size_t, uint32_t, uint32_t, DevBuf<HostNode<float>>&, DevBuf<uint32_t>&, size_t&, hipStream_t, int);
template int sweep_core<double>(const double*, const double*, size_t, uint32_t, uint32_t, DevBuf<HostNode<double>>&, DevBuf<uint32_t>&, size_t&, hipStream_t, int);
template int build_sweep_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, bool, hipStream_t);
template int build_sweep_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, bool, hipStream_t);

} // namespace bvh_amd
