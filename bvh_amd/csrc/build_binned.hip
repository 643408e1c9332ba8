// K5/K6 — binned-SAH top-down construction on gfx950, bit-exact with the reference's
// BinnedSahBuilder (binned_sah_builder.h:82-156) driven by TopDownSahBuilder::build
// (top_down_sah_builder.h:74-139): same nodes, same node numbering, same prim_ids order.
//
// The reference is a serial explicit-stack DFS; its output is reproduced by three phases:
//
//  Phase A  (level-synchronous, segments > 64 primitives): per level, for all such segments at once,
//           bin (LDS-staged bins, one block per 2048-primitive chunk, order-preserving-integer atomics),
//           decide (21 SAH candidates per segment), partition (the exact permutation libstdc++'s Hoare
//           std::partition produces, SURVEY A.3: ballot/prefix ranks, violator tables, pairwise swap),
//           child bounds (LDS-staged min/max), and child creation with SATO order.
//  Phase B  (segments <= 64 primitives): ONE WAVEFRONT finishes the whole subtree, one primitive per
//           lane in registers, replaying the reference's DFS (same stack discipline) with ballot/popcount
//           partitions and LDS bins; nodes are staged in the subtree's own DFS numbering.
//  Phase C  numbering: an inner node's children live at 1 + 2 * (its pre-order rank among inner
//           nodes, in the order the reference's stack pops them: fewer primitives first, ties -> second
//           child); ranks come from bottom-up inner counts + a top-down pass, then nodes are scattered.
//
// Arithmetic is the reference's (-ffp-contract=off; fma only at fast_mul_add sites: bin position,
// binned_sah_builder.h:92, and split plane, :145-148). min/max accumulations are order-independent for
// non-NaN inputs, which is what makes the parallel binning exact; the one order-dependent case, the sign of
// a zero bound when +0 and -0 both occur (the reference keeps the LAST one), is reproduced by tracking the
// last zero-valued position per bound. Known divergence: NaN coordinates.

#include "build_common.h"

namespace bvh_amd {

using namespace bld;

namespace {

// =====================================================================================================
// Phase A kernels
// =====================================================================================================

// fill_bins (binned_sah_builder.h:82-99) for one chunk of one segment.
template <typename T>
__global__ void __launch_bounds__(256) k_bin(BuildCtx<T> c) {
    __shared__ SlotBins<T> sb;
    const Task tk = c.tasks[blockIdx.x];
    const ANode<T>& nd = c.nodes[c.state[tk.slot].node];
    const auto lo0 = Ord<T>::enc(Ord<T>::kMax), hi0 = Ord<T>::enc(-Ord<T>::kMax);
    for (int w = threadIdx.x; w < 3 * kBins * 3; w += 256) { (&sb.lo[0][0][0])[w] = lo0; (&sb.hi[0][0][0])[w] = hi0; }
    for (int w = threadIdx.x; w < 3 * kBins; w += 256) (&sb.cnt[0][0])[w] = 0;
    T scale[3], shift[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {                            // :88-89
        scale[k] = T(kBins) / (nd.hi[k] - nd.lo[k]);
        shift[k] = (-nd.lo[k]) * scale[k];
    }
    __syncthreads();
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256) {
        const uint32_t id = c.ids[pos];
        T ctr[3], blo[3], bhi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ctr[k] = c.centers[3ull * id + k]; blo[k] = c.bboxes[6ull * id + k]; bhi[k] = c.bboxes[6ull * id + 3 + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t s = bin_of(Ord<T>::fma_(ctr[k], scale[k], shift[k]));   // :92-95
#pragma unroll
            for (int j = 0; j < 3; ++j) { atomicMin(&sb.lo[k][s][j], Ord<T>::enc(blo[j])); atomicMax(&sb.hi[k][s][j], Ord<T>::enc(bhi[j])); }
            atomicAdd(&sb.cnt[k][s], 1u);
        }
    }
    __syncthreads();
    SlotBins<T>& gb = c.bins[tk.slot];
    for (int w = threadIdx.x; w < 3 * kBins; w += 256) {
        const uint32_t n = (&sb.cnt[0][0])[w];
        if (!n) continue;
        atomicAdd(&(&gb.cnt[0][0])[w], n);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            atomicMin(&(&gb.lo[0][0][0])[3 * w + j], (&sb.lo[0][0][0])[3 * w + j]);
            atomicMax(&(&gb.hi[0][0][0])[3 * w + j], (&sb.hi[0][0][0])[3 * w + j]);
        }
    }
}

// try_split's decision (binned_sah_builder.h:128-148), one thread per segment.
template <typename T>
__global__ void __launch_bounds__(64) k_decide(BuildCtx<T> c, uint32_t n_active) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    SlotState<T>& st = c.state[slot];
    const ANode<T>& nd = c.nodes[st.node];
    const SlotBins<T>& b = c.bins[slot];
    const int wide = widest_axis(nd.lo, nd.hi, c.dim);
    uint32_t best_bin = kBins / 2; T best_cost = Ord<T>::kMax; int best_axis = wide;     // :132-133
    for (int k = 0; k < c.dim; ++k) {
        T cost; uint32_t bin;
        sweep_axis<T>([&](int i, T* lo, T* hi, uint32_t& n) {
            for (int j = 0; j < 3; ++j) { lo[j] = Ord<T>::dec(b.lo[k][i][j]); hi[j] = Ord<T>::dec(b.hi[k][i][j]); }
            n = b.cnt[k][i];
        }, cost, bin, c.dim, c.sah_log);
        if (cost < best_cost) { best_cost = cost; best_bin = bin; best_axis = k; }
    }
    const uint32_t size = nd.end - nd.begin;
    const T stay = half_area(nd.lo, nd.hi, c.dim) * (sah_prims<T>(size, c.sah_log) - c.sah_ratio);   // split_heuristic.h:35-37
    st.wide = wide;
    st.axis = best_axis;
    if (best_cost >= stay) {
        st.mode = MODE_FALLBACK;                             // size > 64 > max_leaf_size: never a leaf here (:140-142)
    } else {
        st.mode = MODE_PARTITION;
        st.split_pos = Ord<T>::fma_((nd.hi[best_axis] - nd.lo[best_axis]) / T(kBins), static_cast<T>(best_bin), nd.lo[best_axis]);   // :145-148
    }
}

// #primitives of the chunk with center[axis] < split_pos (the std::partition predicate, :151).
template <typename T>
__global__ void __launch_bounds__(256) k_count(BuildCtx<T> c) {
    __shared__ uint32_t total;
    const Task tk = c.tasks[blockIdx.x];
    SlotState<T>& st = c.state[tk.slot];
    if (st.mode != MODE_PARTITION) return;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t pos = tk.begin + threadIdx.x; pos < tk.end; pos += 256)
        mine += c.centers[3ull * c.ids[pos] + st.axis] < st.split_pos ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) { c.chunk_true[blockIdx.x] = total; atomicAdd(&st.m, total); }
}

// Hoare-partition characterisation: with m = #true, the misplaced elements left of begin+m
// (ascending) are swapped pairwise with the misplaced elements right of it (descending).
template <typename T>
__global__ void __launch_bounds__(256) k_scatter(BuildCtx<T> c) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_sh, viol_sh;
    const Task tk = c.tasks[blockIdx.x];
    SlotState<T>& st = c.state[tk.slot];
    if (st.mode != MODE_PARTITION) return;
    const ANode<T>& nd = c.nodes[st.node];
    const uint32_t size = nd.end - nd.begin, m = st.m;
    if (m == 0 || m == size) return;                         // :152-153 -> fallback
    if (threadIdx.x == 0) { base_sh = 0; viol_sh = 0; }
    __syncthreads();
    uint32_t before = 0;                                     // #true in earlier chunks of this segment
    for (uint32_t t = st.task0 + threadIdx.x; t < blockIdx.x; t += 256) before += c.chunk_true[t];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if ((threadIdx.x & 63) == 0 && before) atomicAdd(&base_sh, before);
    __syncthreads();
    uint32_t running = base_sh, viol = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t boundary = nd.begin + m;
    for (uint32_t tile = tk.begin; tile < tk.end; tile += 256) {
        const uint32_t pos = tile + threadIdx.x;
        const bool in = pos < tk.end;
        const bool pred = in && (c.centers[3ull * c.ids[in ? pos : tk.begin] + st.axis] < st.split_pos);
        const uint64_t bal = __ballot(pred);
        if (lane == 0) wave_sum[wave] = __popcll(bal);
        __syncthreads();
        uint32_t excl = __popcll(bal & ((uint64_t{1} << lane) - 1));
        uint32_t tile_total = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) excl += wave_sum[w]; tile_total += wave_sum[w]; }
        const uint32_t t_before = running + excl;            // #true in [begin, pos)
        if (in) {
            if (pos < boundary) {
                if (!pred) { c.ltab[nd.begin + (pos - nd.begin) - t_before] = pos; ++viol; }
            } else if (pred) {
                c.rtab[nd.begin + (m - t_before - 1)] = pos;
            }
        }
        running += tile_total;
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) viol += __shfl_down(viol, off);
    if (lane == 0 && viol) atomicAdd(&viol_sh, viol);
    __syncthreads();
    if (threadIdx.x == 0 && viol_sh) atomicAdd(&st.nviol, viol_sh);
}

// fallback_split (:118-126): std::partial_sort replayed by one lane; also settles the split index.
template <typename T>
__global__ void __launch_bounds__(64) k_fallback(BuildCtx<T> c, uint32_t n_active) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    SlotState<T>& st = c.state[slot];
    const ANode<T>& nd = c.nodes[st.node];
    const uint32_t size = nd.end - nd.begin;
    if (st.mode == MODE_PARTITION && st.m != 0 && st.m != size) { st.split = nd.begin + st.m; return; }
    st.mode = MODE_FALLBACK;
    const uint32_t mid = (nd.begin + nd.end + 1) / 2;        // absolute indices, :119
    const int axis = st.wide;
    const T* ctr = c.centers;
    partial_sort_replay(c.ids + nd.begin, long(mid - nd.begin), long(size), [=](uint32_t id) { return ctr[3ull * id + axis]; });
    st.split = mid;
}

template <typename T>
__global__ void __launch_bounds__(256) k_swap(BuildCtx<T> c) {
    const Task tk = c.tasks[blockIdx.x];
    const SlotState<T>& st = c.state[tk.slot];
    if (st.mode != MODE_PARTITION) return;
    const ANode<T>& nd = c.nodes[st.node];
    const uint32_t chunk = blockIdx.x - st.task0;
    const uint32_t kend = min(st.nviol, (chunk + 1) * kChunk);
    for (uint32_t k = chunk * kChunk + threadIdx.x; k < kend; k += 256) {
        const uint32_t p = c.ltab[nd.begin + k], q = c.rtab[nd.begin + k];
        const uint32_t a = c.ids[p], b = c.ids[q];
        c.ids[p] = b;
        c.ids[q] = a;
    }
}

// =====================================================================================================
// Phase B: one wavefront builds a whole subtree of <= 64 primitives
// =====================================================================================================

template <typename T>
struct WaveLds {
    typename Ord<T>::U lo[3][kBins][3], hi[3][kBins][3];
    uint32_t cnt[3][kBins];
    T nbox[2 * kSmall][6];               // local node boxes {lo xyz, hi xyz}
    uint32_t stack[kSmall + 4];
    uint32_t ltab[kSmall], rtab[kSmall];
    uint32_t perm[kSmall];
    T keys[kSmall];
    T axis_cost[3];
    uint32_t axis_bin[3];
};

template <typename T>
__global__ void __launch_bounds__(256) k_small(BuildCtx<T> c, uint32_t n_small) {
    __shared__ WaveLds<T> lds_all[4];
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    WaveLds<T>& L = lds_all[threadIdx.x >> 6];
    const uint32_t node_id = c.small_list[w];
    ANode<T>& A = c.nodes[node_id];
    const uint32_t B = A.begin, s = A.end - A.begin;
    HostNode<T>* stage = c.stage + 2ull * B;

    // one primitive per lane
    uint32_t id = 0;
    T ctr[3] = {0, 0, 0}, blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
    if (lane < int(s)) {
        id = c.ids[B + lane];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ctr[k] = c.centers[3ull * id + k]; blo[k] = c.bboxes[6ull * id + k]; bhi[k] = c.bboxes[6ull * id + 3 + k]; }
    }
    if (lane < 3) { L.nbox[0][lane] = A.lo[lane]; L.nbox[0][3 + lane] = A.hi[lane]; }
    if (lane == 0) L.stack[0] = pack_item(0, 0, s);
    uint32_t sp = 1, ncount = 1;
    const uint64_t lanes_below = (uint64_t{1} << lane) - 1;

    while (sp > 0) {
        wave_sync();
        const uint32_t item = L.stack[--sp];
        const uint32_t ln = item & 0xFF, lb = (item >> 8) & 0xFF, le = (item >> 16) & 0xFF;
        const uint32_t cnt = le - lb;
        T nlo[3], nhi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[ln][k]; nhi[k] = L.nbox[ln][3 + k]; }
        const bool in = uint32_t(lane) >= lb && uint32_t(lane) < le;
        bool split = false;
        uint32_t cut = 0;                                     // local split index

        if (cnt > c.min_leaf) {                               // top_down_sah_builder.h:89
            // ---- fill_bins
            const auto lo0 = Ord<T>::enc(Ord<T>::kMax), hi0 = Ord<T>::enc(-Ord<T>::kMax);
            for (int q = lane; q < 3 * kBins * 3; q += 64) { (&L.lo[0][0][0])[q] = lo0; (&L.hi[0][0][0])[q] = hi0; }
            if (lane < 3 * kBins) (&L.cnt[0][0])[lane] = 0;
            wave_sync();
            if (in) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T scale = T(kBins) / (nhi[k] - nlo[k]);
                    const T shift = (-nlo[k]) * scale;
                    const uint32_t b = bin_of(Ord<T>::fma_(ctr[k], scale, shift));
#pragma unroll
                    for (int j = 0; j < 3; ++j) { atomicMin(&L.lo[k][b][j], Ord<T>::enc(blo[j])); atomicMax(&L.hi[k][b][j], Ord<T>::enc(bhi[j])); }
                    atomicAdd(&L.cnt[k][b], 1u);
                }
            }
            // ---- find_best_split: lane k sweeps axis k
            wave_sync();
            if (lane < c.dim) {
                T cost; uint32_t bin;
                const int k = lane;
                sweep_axis<T>([&](int i, T* lo, T* hi, uint32_t& n) {
                    for (int j = 0; j < 3; ++j) { lo[j] = Ord<T>::dec(L.lo[k][i][j]); hi[j] = Ord<T>::dec(L.hi[k][i][j]); }
                    n = L.cnt[k][i];
                }, cost, bin, c.dim, c.sah_log);
                L.axis_cost[k] = cost;
                L.axis_bin[k] = bin;
            }
            wave_sync();
            const int wide = widest_axis(nlo, nhi, c.dim);
            uint32_t best_bin = kBins / 2; T best_cost = Ord<T>::kMax; int best_axis = wide;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= c.dim) continue;
                const T cst = L.axis_cost[k];
                if (cst < best_cost) { best_cost = cst; best_bin = L.axis_bin[k]; best_axis = k; }
            }
            const T stay = half_area(nlo, nhi, c.dim) * (sah_prims<T>(cnt, c.sah_log) - c.sah_ratio);
            bool fallback = false;
            if (best_cost >= stay) {
                if (cnt > c.max_leaf) fallback = true;        // else: leaf
            } else {
                const T plane = Ord<T>::fma_((nhi[best_axis] - nlo[best_axis]) / T(kBins), static_cast<T>(best_bin), nlo[best_axis]);
                const T key = best_axis == 0 ? ctr[0] : (best_axis == 1 ? ctr[1] : ctr[2]);
                const bool pred = in && key < plane;
                const uint64_t tmask = __ballot(pred);
                const uint32_t m = __popcll(tmask);
                if (m == 0 || m == cnt) fallback = true;
                else {
                    // Hoare permutation inside the wave
                    const uint32_t boundary = lb + m;
                    const bool lv = in && uint32_t(lane) < boundary && !pred;
                    const bool rv = in && uint32_t(lane) >= boundary && pred;
                    const uint64_t lmask = __ballot(lv), rmask = __ballot(rv);
                    const uint32_t lrank = __popcll(lmask & lanes_below);
                    const uint32_t rrank = __popcll(rmask & ~(lanes_below | (uint64_t{1} << lane)));   // descending rank
                    if (lv) L.ltab[lrank] = lane;
                    if (rv) L.rtab[rrank] = lane;
                    wave_sync();
                    int src = lane;
                    if (lv) src = L.rtab[lrank];
                    if (rv) src = L.ltab[rrank];
                    id = __shfl(id, src);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { ctr[k] = __shfl(ctr[k], src); blo[k] = __shfl(blo[k], src); bhi[k] = __shfl(bhi[k], src); }
                    split = true;
                    cut = boundary;
                }
            }
            if (fallback) {                                   // fallback_split on the widest axis
                const uint32_t mid_abs = (B + lb + B + le + 1) / 2;
                const uint32_t mid = mid_abs - B;
                const T key = wide == 0 ? ctr[0] : (wide == 1 ? ctr[1] : ctr[2]);
                L.keys[lane] = key;
                L.perm[lane] = lane;
                wave_sync();
                if (lane == 0) {
                    const T* keys = L.keys;
                    partial_sort_replay(L.perm + lb, long(mid - lb), long(cnt), [=](uint32_t q) { return keys[q]; });
                }
                wave_sync();
                const int src = L.perm[lane];
                id = __shfl(id, src);
#pragma unroll
                for (int k = 0; k < 3; ++k) { ctr[k] = __shfl(ctr[k], src); blo[k] = __shfl(blo[k], src); bhi[k] = __shfl(bhi[k], src); }
                split = true;
                cut = mid;
            }
        }

        HostNode<T> rec;
#pragma unroll
        for (int k = 0; k < 3; ++k) { rec.bounds[2 * k] = nlo[k]; rec.bounds[2 * k + 1] = nhi[k]; }
        if (split) {
            // child boxes by butterfly reductions over the two lane ranges
            T lo[2][3], hi[2][3];
            const bool left = in && uint32_t(lane) < cut, right = in && uint32_t(lane) >= cut;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[0][k] = left ? blo[k] : Ord<T>::kMax;  hi[0][k] = left ? bhi[k] : -Ord<T>::kMax;
                lo[1][k] = right ? blo[k] : Ord<T>::kMax; hi[1][k] = right ? bhi[k] : -Ord<T>::kMax;
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
            // a zero bound takes the sign of the last zero in position (= lane) order, see SlotState::zlo
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool side = q == 0 ? left : right;
                    if (lo[q][k] == T(0)) {
                        const uint64_t zm = __ballot(side && blo[k] == T(0));
                        lo[q][k] = Ord<T>::zero(__shfl(Ord<T>::sign(blo[k]), 63 - __clzll(zm)));
                    }
                    if (hi[q][k] == T(0)) {
                        const uint64_t zm = __ballot(side && bhi[k] == T(0));
                        hi[q][k] = Ord<T>::zero(__shfl(Ord<T>::sign(bhi[k]), 63 - __clzll(zm)));
                    }
                }
            const int first = half_area(lo[0], hi[0], c.dim) < half_area(lo[1], hi[1], c.dim) ? 1 : 0;   // SATO
            const uint32_t child = ncount;
            ncount += 2;
            const uint32_t rb[2] = { lb, cut }, re[2] = { cut, le };
            if (lane < 3) {
                L.nbox[child][lane] = lo[first][lane];         L.nbox[child][3 + lane] = hi[first][lane];
                L.nbox[child + 1][lane] = lo[1 - first][lane]; L.nbox[child + 1][3 + lane] = hi[1 - first][lane];
            }
            uint32_t ia = pack_item(child, rb[first], re[first]), ib = pack_item(child + 1, rb[1 - first], re[1 - first]);
            if (re[first] - rb[first] < re[1 - first] - rb[1 - first]) { const uint32_t t = ia; ia = ib; ib = t; }   // :116-120
            if (lane == 0) { L.stack[sp] = ia; L.stack[sp + 1] = ib; }
            sp += 2;
            rec.index = static_cast<typename IndexOf<T>::Type>(child) << kCountBits;
        } else {
            rec.index = (static_cast<typename IndexOf<T>::Type>(B + lb) << kCountBits) | cnt;
        }
        if (lane == 0) stage[ln] = rec;
    }
    if (lane < int(s)) c.ids[B + lane] = id;
    if (lane == 0) A.ic = (ncount - 1) / 2;
}

} // namespace

// ---- host side ---------------------------------------------------------------------------------------------

namespace {

template <typename T>
struct BinnedWs {
    DevBuf<uint32_t> ids, chunk_true, ltab, rtab, small_list;
    DevBuf<ANode<T>> nodes;
    DevBuf<SlotBins<T>> bins;
    DevBuf<SlotState<T>> st_a, st_b;
    DevBuf<Task> tk_a, tk_b;
    DevBuf<HostNode<T>> stage;
    DevBuf<Counters> counters;

    // capacities: typical on the first attempt, worst case (degenerate chains of big nodes) on retry
    int alloc(BuildCtx<T>& c, uint32_t n, uint32_t roots, int attempt, bool own_ids) {
        const uint32_t node_cap = (attempt == 0 ? n / 8 + 1024 : 2 * n + 2) + roots;
        const uint32_t slot_cap = n / (kSmall + 1) + 2;
        const uint32_t task_cap = n / kChunk + slot_cap + 2;
        hipError_t e = hipSuccess;
        auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        if (own_ids) A(ids.alloc(n));
        A(chunk_true.alloc(task_cap)); A(ltab.alloc(n)); A(rtab.alloc(n)); A(small_list.alloc(node_cap));
        A(nodes.alloc(node_cap)); A(bins.alloc(slot_cap)); A(st_a.alloc(slot_cap)); A(st_b.alloc(slot_cap));
        A(tk_a.alloc(task_cap)); A(tk_b.alloc(task_cap)); A(stage.alloc(2 * size_t{n})); A(counters.alloc(1));
        if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));
        if (own_ids) c.ids = ids.p;
        c.n = n; c.nodes = nodes.p; c.node_cap = node_cap; c.bins = bins.p; c.state = st_a.p; c.state_next = st_b.p;
        c.slot_cap = slot_cap; c.tasks = tk_a.p; c.tasks_next = tk_b.p; c.task_cap = task_cap; c.chunk_true = chunk_true.p;
        c.ltab = ltab.p; c.rtab = rtab.p; c.small_list = small_list.p; c.stage = stage.p; c.counters = counters.p;
        return BVH_AMD_OK;
    }
};

// Phase A levels + Phase B. Expects the roots already registered (state_next / tasks_next / counters) and `h` read back.
template <typename T>
int run_binned_phases(BuildCtx<T>& c, Counters& h, std::vector<uint32_t>& level_start, bool& overflow, hipStream_t stream) {
    uint32_t n_active = h.n_active_next, n_tasks = h.n_tasks_next;
    overflow = h.error != 0;
    while (n_active > 0 && !overflow) {
        std::swap(c.state, c.state_next);
        std::swap(c.tasks, c.tasks_next);
        BVH_HIP_TRY(hipMemsetAsync(&c.counters->n_active_next, 0, 2 * sizeof(uint32_t), stream), BVH_AMD_ERR_HIP);
        const unsigned slot_grid = (n_active + 63) / 64;
        hipLaunchKernelGGL(k_init_slots<T>, dim3(n_active), dim3(64), 0, stream, c);
        hipLaunchKernelGGL(k_bin<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_decide<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
        hipLaunchKernelGGL(k_count<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_scatter<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_fallback<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
        hipLaunchKernelGGL(k_swap<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_child_bounds<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_finalize<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipMemcpyAsync(&h, c.counters, sizeof(h), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        overflow = h.error != 0;
        level_start.push_back(h.n_nodes);
        n_active = h.n_active_next;
        n_tasks = h.n_tasks_next;
    }
    if (!overflow && h.n_small) hipLaunchKernelGGL(k_small<T>, dim3((h.n_small + 3) / 4), dim3(256), 0, stream, c, h.n_small);
    return BVH_AMD_OK;
}

// Roots of a forest: tree g covers positions [group_begin[g], group_begin[g + 1]) of c.ids. One block per tree
// computes compute_bbox over its range in position order (top_down_sah_builder.h:80, :133-139).
template <typename T>
__global__ void __launch_bounds__(256) k_forest_roots(BuildCtx<T> c, const uint32_t* group_begin) {
    __shared__ typename Ord<T>::U slo[3], shi[3];
    __shared__ uint32_t zlo[3], zhi[3];
    const uint32_t g = blockIdx.x, b = group_begin[g], e = group_begin[g + 1];
    if (threadIdx.x < 3) { slo[threadIdx.x] = Ord<T>::enc(Ord<T>::kMax); shi[threadIdx.x] = Ord<T>::enc(-Ord<T>::kMax); zlo[threadIdx.x] = 0; zhi[threadIdx.x] = 0; }
    __syncthreads();
    T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
    for (uint32_t pos = b + threadIdx.x; pos < e; pos += 256) {
        const uint32_t id = c.ids[pos];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T lo_v = c.bboxes[6ull * id + k], hi_v = c.bboxes[6ull * id + 3 + k];
            lo[k] = pick_min(lo[k], lo_v); hi[k] = pick_max(hi[k], hi_v);
            track_zero(&zlo[k], lo_v, pos); track_zero(&zhi[k], hi_v, pos);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { atomicMin(&slo[k], Ord<T>::enc(lo[k])); atomicMax(&shi[k], Ord<T>::enc(hi[k])); }
    __syncthreads();
    if (threadIdx.x == 0) {
        ANode<T>& r = c.nodes[g];
        for (int k = 0; k < 3; ++k) { r.lo[k] = decode_bound<T>(slo[k], zlo[k]); r.hi[k] = decode_bound<T>(shi[k], zhi[k]); }
        r.begin = b; r.end = e; r.child = kNone; r.parent = kNone; r.ic = 0; r.rank = 0; r.tree = g;
        emit_child(c, g);
    }
}

template <typename T>
__global__ void k_forest_prepare(BuildCtx<T> c, uint32_t n_groups) {
    Counters z = {};
    z.n_nodes = n_groups;
    *c.counters = z;
}

// tree_node_off[g] = sum of (2 * ic + 1) over earlier trees: one block, tiles of 1024 trees with a running carry
template <typename T>
__global__ void __launch_bounds__(1024) k_forest_offsets(BuildCtx<T> c, uint32_t n_groups, uint32_t* tree_node_off) {
    __shared__ uint32_t warp_sums[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_groups; base += 1024) {
        const uint32_t g = base + threadIdx.x;
        const uint32_t v = g < n_groups ? 2 * c.nodes[g].ic + 1 : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= static_cast<uint32_t>(d)) incl += o; }
        if (lane == 63) warp_sums[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = warp_sums[w]; if (w < wave) before += x; all += x; }
        __syncthreads();
        if (g < n_groups) tree_node_off[g] = carry + before + incl - v;
        carry += all;
    }
    if (threadIdx.x == 0) tree_node_off[n_groups] = carry;
}

} // namespace

// BinnedSahBuilder::build on the device. On success `out` holds the host mirror and the device copy.
template <typename T>
int build_binned_device(BvhImpl<T>& out, const T* d_bboxes, const T* d_centers, size_t n, const bvh_build_config& cfg,
                        hipStream_t stream)
{
    if (n >= (size_t{1} << 28)) return fail(BVH_AMD_ERR_UNSUPPORTED, "build: more than 2^28 primitives");
    BVH_HIP_TRY(hipGetDevice(&out.device), BVH_AMD_ERR_HIP);
    for (int attempt = 0; attempt < 2; ++attempt) {
        BinnedWs<T> ws;
        DevBuf<HostNode<T>> final_nodes;
        BuildCtx<T> c;
        c.bboxes = d_bboxes; c.centers = d_centers;
        c.min_leaf = static_cast<uint32_t>(cfg.min_leaf_size); c.max_leaf = static_cast<uint32_t>(cfg.max_leaf_size);
        c.dim = out.dim;
        c.sah_log = ambient_sah().log_cluster; c.sah_ratio = static_cast<T>(ambient_sah().cost_ratio);
        int rc = ws.alloc(c, static_cast<uint32_t>(n), 1, attempt, true);
        if (rc) return rc;
        hipLaunchKernelGGL(k_prepare_root<T>, dim3(1), dim3(1), 0, stream, c);
        const unsigned root_grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 2048));
        hipLaunchKernelGGL(k_init_root<T>, dim3(root_grid), dim3(256), 0, stream, c, true);
        hipLaunchKernelGGL(k_make_root<T>, dim3(1), dim3(1), 0, stream, c);
        Counters h;
        BVH_HIP_TRY(hipMemcpyAsync(&h, c.counters, sizeof(h), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        std::vector<uint32_t> level_start{0, 1};              // A-node id ranges per level
        bool overflow = false;
        rc = run_binned_phases(c, h, level_start, overflow, stream);
        if (rc) return rc;
        if (overflow) {
            if (attempt == 0) continue;
            return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
        }
        rc = number_and_emit<T>(out, c, level_start, h.n_nodes, h.n_small, final_nodes, stream);
        if (rc) return rc;
        rc = finish_build<T>(out, final_nodes, ws.ids.p, n, stream, /*take_ids=*/true);
        if (rc) return rc;
        ws.ids.p = nullptr;                                   // handed over to `out`
        return BVH_AMD_OK;
    }
    return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
}

// One BinnedSahBuilder tree per group (MiniTreeBuilder::BuildTask::run, mini_tree_builder.h:122-139): d_ids holds the
// groups' ids (ascending within a group) and is permuted in place; every tree is emitted standalone into `trees`
// at tree_node_off[g] with leaf first_id relative to the group.
template <typename T>
int build_binned_forest_device(const T* d_bboxes, const T* d_centers, uint32_t* d_ids, uint32_t n, const uint32_t* d_group_begin,
                               uint32_t n_groups, const bvh_build_config& cfg, DevBuf<HostNode<T>>& trees,
                               DevBuf<uint32_t>& tree_node_off, uint32_t& total_nodes, hipStream_t stream)
{
    DevBuf<uint32_t> sorted_ids;                              // a retry must start from the ascending order again
    BVH_HIP_TRY(sorted_ids.alloc(n), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemcpyAsync(sorted_ids.p, d_ids, size_t{n} * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt) BVH_HIP_TRY(hipMemcpyAsync(d_ids, sorted_ids.p, size_t{n} * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
        BinnedWs<T> ws;
        BuildCtx<T> c;
        c.bboxes = d_bboxes; c.centers = d_centers; c.ids = d_ids;
        c.min_leaf = static_cast<uint32_t>(cfg.min_leaf_size); c.max_leaf = static_cast<uint32_t>(cfg.max_leaf_size);
        c.sah_log = ambient_sah().log_cluster; c.sah_ratio = static_cast<T>(ambient_sah().cost_ratio);
        int rc = ws.alloc(c, n, n_groups, attempt, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_forest_prepare<T>, dim3(1), dim3(1), 0, stream, c, n_groups);
        hipLaunchKernelGGL(k_forest_roots<T>, dim3(n_groups), dim3(256), 0, stream, c, d_group_begin);
        Counters h;
        BVH_HIP_TRY(hipMemcpyAsync(&h, c.counters, sizeof(h), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        std::vector<uint32_t> level_start{0, n_groups};
        bool overflow = false;
        rc = run_binned_phases(c, h, level_start, overflow, stream);
        if (rc) return rc;
        if (overflow) {
            if (attempt == 0) continue;
            return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
        }
        rc = number_nodes<T>(c, level_start, stream);
        if (rc) return rc;
        BVH_HIP_TRY(tree_node_off.alloc(n_groups + 1), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_forest_offsets<T>, dim3(1), dim3(1024), 0, stream, c, n_groups, tree_node_off.p);
        BVH_HIP_TRY(hipMemcpyAsync(&total_nodes, tree_node_off.p + n_groups, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(trees.alloc(total_nodes), BVH_AMD_ERR_HIP);
        c.tree_node_off = tree_node_off.p;
        c.tree_begin = d_group_begin;
        hipLaunchKernelGGL(k_emit_tree<T>, dim3((h.n_nodes + 255) / 256), dim3(256), 0, stream, c, h.n_nodes, trees.p);
        if (h.n_small) hipLaunchKernelGGL(k_emit_small<T>, dim3((h.n_small + 3) / 4), dim3(256), 0, stream, c, h.n_small, trees.p);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // workspace dies here
        return BVH_AMD_OK;
    }
    return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
}

template int build_binned_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, hipStream_t);
template int build_binned_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, hipStream_t);
template int build_binned_forest_device<float>(const float*, const float*, uint32_t*, uint32_t, const uint32_t*, uint32_t,
    const bvh_build_config&, DevBuf<HostNode<float>>&, DevBuf<uint32_t>&, uint32_t&, hipStream_t);
template int build_binned_forest_device<double>(const double*, const double*, uint32_t*, uint32_t, const uint32_t*, uint32_t,
    const bvh_build_config&, DevBuf<HostNode<double>>&, DevBuf<uint32_t>&, uint32_t&, hipStream_t);

} // namespace bvh_amd
