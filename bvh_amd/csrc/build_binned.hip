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

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bvh_amd {

using namespace bld;

namespace {

// =====================================================================================================
// Phase A kernels
// =====================================================================================================

// fill_bins (binned_sah_builder.h:82-99) for one chunk of one segment.
template <typename T, int NB = kBins>
__global__ void __launch_bounds__(256) k_bin(BuildCtx<T> c) {
    constexpr int kBins = NB;            // BinCount of this instantiation (binned_sah_builder.h:18)
    __shared__ SlotBins<T, NB> sb;
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
            const uint32_t s = bin_of_n<NB>(Ord<T>::fma_(ctr[k], scale[k], shift[k]));   // :92-95
#pragma unroll
            for (int j = 0; j < 3; ++j) { atomicMin(&sb.lo[k][s][j], Ord<T>::enc(blo[j])); atomicMax(&sb.hi[k][s][j], Ord<T>::enc(bhi[j])); }
            atomicAdd(&sb.cnt[k][s], 1u);
        }
    }
    __syncthreads();
    SlotBins<T, NB>& gb = reinterpret_cast<SlotBins<T, NB>*>(c.bins)[tk.slot];
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
template <typename T, int NB = kBins>
__global__ void __launch_bounds__(64) k_decide(BuildCtx<T> c, uint32_t n_active) {
    constexpr int kBins = NB;
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    SlotState<T>& st = c.state[slot];
    const ANode<T>& nd = c.nodes[st.node];
    const SlotBins<T, NB>& b = reinterpret_cast<const SlotBins<T, NB>*>(c.bins)[slot];
    const int wide = widest_axis(nd.lo, nd.hi, c.dim);
    uint32_t best_bin = kBins / 2; T best_cost = Ord<T>::kMax; int best_axis = wide;     // :132-133
    for (int k = 0; k < c.dim; ++k) {
        T cost; uint32_t bin;
        sweep_axis_n<T, NB>([&](int i, T* lo, T* hi, uint32_t& n) {
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

template <typename T, int NB = kBins>
struct WaveLds {
    typename Ord<T>::U lo[3][NB][3], hi[3][NB][3];
    uint32_t cnt[3][NB];
    T nbox[2 * kSmall][6];               // local node boxes {lo xyz, hi xyz}
    uint32_t stack[kSmall + 4];
    uint32_t ltab[kSmall], rtab[kSmall];
    uint32_t perm[kSmall];
    T keys[kSmall];
    T axis_cost[3];
    uint32_t axis_bin[3];
};

template <typename T, int NB = kBins>
__global__ void __launch_bounds__(256) k_small(BuildCtx<T> c, uint32_t n_small) {
    constexpr int kBins = NB;
    __shared__ WaveLds<T, NB> lds_all[4];
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    WaveLds<T, NB>& L = lds_all[threadIdx.x >> 6];
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
            for (int q = lane; q < 3 * kBins; q += 64) (&L.cnt[0][0])[q] = 0;
            wave_sync();
            if (in) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T scale = T(kBins) / (nhi[k] - nlo[k]);
                    const T shift = (-nlo[k]) * scale;
                    const uint32_t b = bin_of_n<NB>(Ord<T>::fma_(ctr[k], scale, shift));
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
                sweep_axis_n<T, NB>([&](int i, T* lo, T* hi, uint32_t& n) {
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


// ---- Phase B, level-synchronous (round 2) ---------------------------------------------------------------------------------------
// k_small above walks the subtree node by node like the reference's stack does: ~60 strictly sequential node steps for a
// 64-primitive subtree, each with most of the wave idle (a node of 8 primitives uses 8 lanes). The RESULT of that walk does not
// depend on its order: a node's split only looks at its own primitives (kept in its own contiguous lane range), and the reference's
// numbering follows from the shape of the finished subtree (children of the inner node with pre-order rank r — fewer primitives
// first, ties: second child, top_down_sah_builder.h:116-121 — live at 1 + 2r, 2 + 2r; the same rule Phase C applies to the big nodes).
// So here all nodes of one LEVEL are split at once, up to kSlots nodes per pass: bins per slot in LDS (the same atomics, the same
// sweep_axis per (slot, axis) lane), one ballot for all partitions (each lane masks it with its node's lane range), child boxes by
// LDS atomics per child with the last-zero tracking of Phase A; nodes are numbered level by level while the subtree grows and
// renumbered at the end by inner counts (bottom-up over the levels) and ranks (top-down). ~10 passes instead of ~60 node steps.
constexpr int kSlots = 8;

template <typename T>
struct LevelLds {
    // The bins of a pass are dead once its sweeps have run; what the rest of the pass needs — the partition tables, the fallback's keys and
    // the child boxes — lives in the same bytes (round 3: 11.7 -> 9.7 KB per wave = 8 blocks of two waves per CU instead of 6).
    union {
        struct {
            typename Ord<T>::U lo[kSlots][3][kBins][3], hi[kSlots][3][kBins][3];
            uint32_t cnt[kSlots][3][kBins];
        };
        struct {
            typename Ord<T>::U cbox_lo[2 * kSlots][3], cbox_hi[2 * kSlots][3];     // child boxes of this pass (side 0 = left range, 1 = right range)
            uint32_t czlo[2 * kSlots][3], czhi[2 * kSlots][3];                     // last lane with a zero bound << 1 | its sign
            uint32_t ltab[kSmall], rtab[kSmall], perm[kSmall];
            T keys[kSmall];
        };
    };
    T axis_cost[kSlots][3];
    uint32_t axis_bin[kSlots][3];
    // per slot decisions
    uint32_t s_node[kSlots], s_mode[kSlots], s_axis[kSlots], s_wide[kSlots], s_cut[kSlots], s_child[kSlots], s_first[kSlots];
    T s_plane[kSlots];
    // nodes of the subtree in the order they are created (level by level)
    T nbox[2 * kSmall][6];
    uint8_t nb[2 * kSmall], ne[2 * kSmall], nparent[2 * kSmall], nwhich[2 * kSmall], nchild[2 * kSmall], nic[2 * kSmall], nrank[2 * kSmall];
    uint8_t slot_of[2 * kSmall];
};
enum : uint32_t { SM_LEAF = 0, SM_PARTITION = 1, SM_FALLBACK = 2 };

template <typename T>
__global__ void __launch_bounds__(128, sizeof(T) == 4 ? 4 : 1) k_small_levels(BuildCtx<T> c, uint32_t n_small) {
    __shared__ LevelLds<T> lds_all[2];
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 2 + (threadIdx.x >> 6);
    if (w >= n_small) return;
    LevelLds<T>& L = lds_all[threadIdx.x >> 6];
    const uint32_t node_id = c.small_list[w];
    ANode<T>& A = c.nodes[node_id];
    const uint32_t B = A.begin, s = A.end - A.begin;
    HostNode<T>* stage = c.stage + 2ull * B;
    using I = typename IndexOf<T>::Type;

    uint32_t id = 0;
    T ctr[3] = {0, 0, 0}, blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
    if (lane < int(s)) {
        id = c.ids[B + lane];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ctr[k] = c.centers[3ull * id + k]; blo[k] = c.bboxes[6ull * id + k]; bhi[k] = c.bboxes[6ull * id + 3 + k]; }
    }
    if (lane < 3) { L.nbox[0][lane] = A.lo[lane]; L.nbox[0][3 + lane] = A.hi[lane]; }
    if (lane == 0) { L.nb[0] = 0; L.ne[0] = static_cast<uint8_t>(s); L.nparent[0] = 0; L.nwhich[0] = 0; L.nchild[0] = 0; }
    const uint64_t lanes_below = (uint64_t{1} << lane) - 1;
    const bool in_tree = lane < int(s);
    uint32_t my_node = 0;                                     // the deepest node created so far that holds this lane's primitive
    uint32_t n_nodes = 1;                                     // nodes created so far (wave-uniform)
    uint32_t level_first[kSmall + 2];                         // (registers would be too many: kept small by the loop bound below)
    uint32_t n_levels = 0;
    uint32_t lvl_begin = 0, lvl_end = 1;                      // node ids of the current level
    // level boundaries are needed again for the numbering: at most 64 levels (every level has at least one node that splits)
    __shared__ uint8_t level_start_all[2][kSmall + 2];
    uint8_t* level_start = level_start_all[threadIdx.x >> 6];
    (void)level_first;
    wave_sync();

    while (lvl_begin < lvl_end) {
        if (lane == 0) level_start[n_levels] = static_cast<uint8_t>(lvl_begin);
        ++n_levels;
        for (uint32_t pass_first = lvl_begin; pass_first < lvl_end; pass_first += kSlots) {
            const uint32_t n_slots = min(static_cast<uint32_t>(kSlots), lvl_end - pass_first);
            // ---- my slot, my node's range and box
            const bool mine = in_tree && my_node >= pass_first && my_node < pass_first + n_slots;
            const uint32_t slot = mine ? my_node - pass_first : 0u;
            const uint32_t lb = mine ? L.nb[my_node] : 0u, le = mine ? L.ne[my_node] : 0u;
            const uint32_t cnt = le - lb;
            T nlo[3] = {0, 0, 0}, nhi[3] = {0, 0, 0};
            if (mine) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[my_node][k]; nhi[k] = L.nbox[my_node][3 + k]; }
            }
            const bool splits = mine && cnt > c.min_leaf;                 // top_down_sah_builder.h:89
            // ---- fill_bins for every slot of the pass
            {
                const auto lo0 = Ord<T>::enc(Ord<T>::kMax), hi0 = Ord<T>::enc(-Ord<T>::kMax);
                for (uint32_t q = lane; q < n_slots * 3 * kBins * 3; q += 64) { (&L.lo[0][0][0][0])[q] = lo0; (&L.hi[0][0][0][0])[q] = hi0; }
                for (uint32_t q = lane; q < n_slots * 3 * kBins; q += 64) (&L.cnt[0][0][0])[q] = 0;
            }
            wave_sync();
            if (splits) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T scale = T(kBins) / (nhi[k] - nlo[k]);
                    const T shift = (-nlo[k]) * scale;
                    const uint32_t b = bin_of(Ord<T>::fma_(ctr[k], scale, shift));
#pragma unroll
                    for (int j = 0; j < 3; ++j) { atomicMin(&L.lo[slot][k][b][j], Ord<T>::enc(blo[j])); atomicMax(&L.hi[slot][k][b][j], Ord<T>::enc(bhi[j])); }
                    atomicAdd(&L.cnt[slot][k][b], 1u);
                }
            }
            wave_sync();
            // ---- find_best_split: lane (slot, axis) sweeps one axis of one slot
            if (static_cast<uint32_t>(lane) < 3 * n_slots) {
                const int sl = lane / 3, k = lane % 3;
                if (k < c.dim) {
                    T cost; uint32_t bin;
                    sweep_axis<T>([&](int i, T* lo, T* hi, uint32_t& n) {
                        for (int j = 0; j < 3; ++j) { lo[j] = Ord<T>::dec(L.lo[sl][k][i][j]); hi[j] = Ord<T>::dec(L.hi[sl][k][i][j]); }
                        n = L.cnt[sl][k][i];
                    }, cost, bin, c.dim, c.sah_log);
                    L.axis_cost[sl][k] = cost;
                    L.axis_bin[sl][k] = bin;
                }
            }
            wave_sync();
            // (the bins are dead from here on: their bytes now hold this pass's child boxes and partition tables)
            for (uint32_t q = lane; q < 2 * n_slots * 3; q += 64) {
                (&L.cbox_lo[0][0])[q] = Ord<T>::enc(Ord<T>::kMax); (&L.cbox_hi[0][0])[q] = Ord<T>::enc(-Ord<T>::kMax); (&L.czlo[0][0])[q] = 0; (&L.czhi[0][0])[q] = 0;
            }
            // ---- try_split's decision, one lane per slot (binned_sah_builder.h:128-148)
            if (static_cast<uint32_t>(lane) < n_slots) {
                const uint32_t nd = pass_first + lane;
                const uint32_t nbg = L.nb[nd], nen = L.ne[nd], ncnt = nen - nbg;
                T blo6[3], bhi6[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { blo6[k] = L.nbox[nd][k]; bhi6[k] = L.nbox[nd][3 + k]; }
                uint32_t mode = SM_LEAF, axis = 0;
                T plane = T(0);
                const int wide = widest_axis(blo6, bhi6, c.dim);
                if (ncnt > c.min_leaf) {
                    uint32_t best_bin = kBins / 2; T best_cost = Ord<T>::kMax; int best_axis = wide;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (k >= c.dim) continue;
                        const T cst = L.axis_cost[lane][k];
                        if (cst < best_cost) { best_cost = cst; best_bin = L.axis_bin[lane][k]; best_axis = k; }
                    }
                    const T stay = half_area(blo6, bhi6, c.dim) * (sah_prims<T>(ncnt, c.sah_log) - c.sah_ratio);
                    if (best_cost >= stay) {
                        if (ncnt > c.max_leaf) mode = SM_FALLBACK;            // else: leaf
                    } else {
                        mode = SM_PARTITION;
                        axis = static_cast<uint32_t>(best_axis);
                        plane = Ord<T>::fma_((bhi6[best_axis] - blo6[best_axis]) / T(kBins), static_cast<T>(best_bin), blo6[best_axis]);
                    }
                }
                L.s_node[lane] = nd; L.s_mode[lane] = mode; L.s_axis[lane] = axis; L.s_wide[lane] = static_cast<uint32_t>(wide);
                L.s_plane[lane] = plane; L.s_cut[lane] = 0; L.s_child[lane] = 0; L.s_first[lane] = 0;
            }
            wave_sync();
            // ---- std::partition of every slot at once (Hoare permutation inside the node's lane range)
            uint32_t mode = mine ? L.s_mode[slot] : SM_LEAF;
            const uint64_t my_range = mine ? (((le >= 64 ? ~uint64_t{0} : ((uint64_t{1} << le) - 1))) & ~((uint64_t{1} << lb) - 1)) : 0;
            bool fallback = mode == SM_FALLBACK;
            int src = lane;
            uint32_t cut = 0;
            {
                const uint32_t ax = mine ? L.s_axis[slot] : 0u;
                const T key = ax == 0 ? ctr[0] : (ax == 1 ? ctr[1] : ctr[2]);
                const bool pred = mode == SM_PARTITION && key < L.s_plane[slot];
                const uint64_t tmask = __ballot(pred) & my_range;
                const uint32_t m = __popcll(tmask);
                if (mode == SM_PARTITION && (m == 0 || m == cnt)) { fallback = true; mode = SM_FALLBACK; }   // :152-153
                const bool part = mode == SM_PARTITION;
                const uint32_t boundary = lb + m;
                const bool lv = part && uint32_t(lane) < boundary && !pred;
                const bool rv = part && uint32_t(lane) >= boundary && pred;
                const uint64_t lmask = __ballot(lv) & my_range, rmask = __ballot(rv) & my_range;
                const uint32_t lrank = __popcll(lmask & lanes_below);
                const uint32_t rrank = __popcll(rmask & ~(lanes_below | (uint64_t{1} << lane)));   // descending rank
                if (lv) L.ltab[lb + lrank] = lane;
                if (rv) L.rtab[lb + rrank] = lane;
                wave_sync();
                if (lv) src = L.rtab[lb + lrank];
                if (rv) src = L.ltab[lb + rrank];
                if (part) cut = boundary;
            }
            id = __shfl(id, src);
#pragma unroll
            for (int k = 0; k < 3; ++k) { ctr[k] = __shfl(ctr[k], src); blo[k] = __shfl(blo[k], src); bhi[k] = __shfl(bhi[k], src); }
            // ---- fallback_split (:118-126) of the slots that need one: std::partial_sort replayed by one lane each
            uint64_t fb_lanes = __ballot(fallback);
            if (fb_lanes) {
                const uint32_t wide = mine ? L.s_wide[slot] : 0u;
                const T key = wide == 0 ? ctr[0] : (wide == 1 ? ctr[1] : ctr[2]);
                L.keys[lane] = key;
                L.perm[lane] = lane;
                wave_sync();
                const uint32_t mid = (B + lb + B + le + 1) / 2 - B;          // absolute indices, :119
                if (fallback && uint32_t(lane) == lb) {                      // one lane per falling-back node
                    const T* keys = L.keys;
                    partial_sort_replay(L.perm + lb, long(mid - lb), long(cnt), [=](uint32_t q) { return keys[q]; });
                }
                wave_sync();
                const int fsrc = fallback ? static_cast<int>(L.perm[lane]) : lane;
                id = __shfl(id, fsrc);
#pragma unroll
                for (int k = 0; k < 3; ++k) { ctr[k] = __shfl(ctr[k], fsrc); blo[k] = __shfl(blo[k], fsrc); bhi[k] = __shfl(bhi[k], fsrc); }
                if (fallback) cut = mid;
            }
            const bool split = mine && mode != SM_LEAF;
            // the first lane of every splitting node publishes the cut
            if (split && uint32_t(lane) == lb) { L.s_cut[slot] = cut; L.s_mode[slot] = mode; }
            // ---- child boxes: every primitive joins its side's box (order-independent min / max + the last-zero rule)
            const uint32_t side = split && uint32_t(lane) >= cut ? 1u : 0u;
            if (split) {
                const uint32_t cb = 2 * slot + side;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    atomicMin(&L.cbox_lo[cb][k], Ord<T>::enc(blo[k])); atomicMax(&L.cbox_hi[cb][k], Ord<T>::enc(bhi[k]));
                    if (blo[k] == T(0)) atomicMax(&L.czlo[cb][k], (static_cast<uint32_t>(lane) << 1) | Ord<T>::sign(blo[k]));
                    if (bhi[k] == T(0)) atomicMax(&L.czhi[cb][k], (static_cast<uint32_t>(lane) << 1) | Ord<T>::sign(bhi[k]));
                }
            }
            wave_sync();
            // ---- one lane per slot creates the two children (SATO order) in slot order
            {
                const bool creates = static_cast<uint32_t>(lane) < n_slots && L.s_mode[lane] != SM_LEAF;
                const uint64_t cm = __ballot(creates);
                if (creates) {
                    const uint32_t nd = L.s_node[lane];
                    const uint32_t child = n_nodes + 2 * __popcll(cm & lanes_below);
                    T clo[2][3], chi[2][3];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            clo[q][k] = decode_bound<T>(L.cbox_lo[2 * lane + q][k], L.czlo[2 * lane + q][k]);
                            chi[q][k] = decode_bound<T>(L.cbox_hi[2 * lane + q][k], L.czhi[2 * lane + q][k]);
                        }
                    const int first = half_area(clo[0], chi[0], c.dim) < half_area(clo[1], chi[1], c.dim) ? 1 : 0;   // SATO (:99-107)
                    const uint32_t nbg = L.nb[nd], nen = L.ne[nd], ct = L.s_cut[lane];
                    const uint32_t rb[2] = { nbg, ct }, re[2] = { ct, nen };
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        L.nbox[child][k] = clo[first][k];         L.nbox[child][3 + k] = chi[first][k];
                        L.nbox[child + 1][k] = clo[1 - first][k]; L.nbox[child + 1][3 + k] = chi[1 - first][k];
                    }
                    L.nb[child] = static_cast<uint8_t>(rb[first]);         L.ne[child] = static_cast<uint8_t>(re[first]);
                    L.nb[child + 1] = static_cast<uint8_t>(rb[1 - first]); L.ne[child + 1] = static_cast<uint8_t>(re[1 - first]);
                    L.nparent[child] = static_cast<uint8_t>(nd); L.nparent[child + 1] = static_cast<uint8_t>(nd);
                    L.nwhich[child] = 0; L.nwhich[child + 1] = 1;
                    L.nchild[child] = 0; L.nchild[child + 1] = 0;
                    L.nchild[nd] = static_cast<uint8_t>(child);
                    L.s_child[lane] = child; L.s_first[lane] = static_cast<uint32_t>(first);
                }
                n_nodes += 2 * __popcll(cm);
            }
            wave_sync();
            if (split) my_node = L.s_child[slot] + (side != L.s_first[slot] ? 1u : 0u);
            wave_sync();
        }
        lvl_begin = lvl_end;
        lvl_end = n_nodes;
    }
    if (lane == 0) level_start[n_levels] = static_cast<uint8_t>(n_nodes);
    wave_sync();
    // ---- the reference's numbering: inner counts bottom-up, pre-order ranks top-down (Phase C's rule, within the subtree)
    for (uint32_t lv = n_levels; lv-- > 0;) {
        const uint32_t f = level_start[lv], e = level_start[lv + 1];
        for (uint32_t t = f + lane; t < e; t += 64) {
            const uint32_t ch = L.nchild[t];
            L.nic[t] = ch ? static_cast<uint8_t>(1 + L.nic[ch] + L.nic[ch + 1]) : 0;
        }
        wave_sync();
    }
    if (lane == 0) L.nrank[0] = 0;
    wave_sync();
    for (uint32_t lv = 0; lv < n_levels; ++lv) {
        const uint32_t f = level_start[lv], e = level_start[lv + 1];
        for (uint32_t t = f + lane; t < e; t += 64) {
            const uint32_t ch = L.nchild[t];
            if (ch) {
                const uint32_t c0 = L.ne[ch] - L.nb[ch], c1 = L.ne[ch + 1] - L.nb[ch + 1];
                const uint32_t r = L.nrank[t];
                if (c0 < c1) { L.nrank[ch] = static_cast<uint8_t>(r + 1); L.nrank[ch + 1] = static_cast<uint8_t>(r + 1 + L.nic[ch]); }   // fewer primitives first,
                else         { L.nrank[ch + 1] = static_cast<uint8_t>(r + 1); L.nrank[ch] = static_cast<uint8_t>(r + 1 + L.nic[ch + 1]); } // ties: the second child
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
    if (lane < int(s)) c.ids[B + lane] = id;
    if (lane == 0) A.ic = L.nic[0];
}

// =====================================================================================================
// Phase A', round 4: one BLOCK splits a segment of 65 .. medium_cap primitives down to subtrees of <= 64
// =====================================================================================================
// Phase A pays a dozen launches per level, and on segments of a few hundred to a few thousand primitives (every mini-tree of the
// thread-pool builder, mini_tree_builder.h:160-205, starts there) each of them is a short-lived block walking a chain of dependent
// global round trips (task -> slot -> node -> ids -> primitive data): 63 % of k_bin's wave time was waiting. Here the
// segment's primitives (centre + box, 36 bytes each) are loaded ONCE into LDS and the block runs the same level-synchronous steps —
// fill_bins, find_best_split, try_split's decision, std::partition as its Hoare permutation, fallback_split, compute_bbox of both
// sides, child creation in SATO order (binned_sah_builder.h:82-156, top_down_sah_builder.h:89-121) — for ALL the segment's nodes of
// a level at once, with block barriers instead of kernel boundaries, until every remaining piece holds <= 64 primitives (Phase B's
// k_small_levels takes those). Only the permutation `order` moves; the primitive data stays where it was loaded. Arithmetic and
// tie rules are Phase A's, statement by statement. The nodes it creates are appended to c.nodes (one atomicAdd per block), level by
// level; Phase C walks them through MedInfo (build_common.h: k_medium_count / k_medium_rank).
// Four size classes, KM = 256 / 512 / 1024 / 2048 primitives (double: up to 1024), KM / 2 threads each: what a block keeps in LDS
// grows with KM (20 / 38 / 72 / 142 KB for float), so the small classes run 8 / 4 / 2 blocks per CU and hide each other's barriers —
// Quality::Low with a pool makes 4096 mini-trees of ~n / 4096 primitives (no bin merging without pruning), the other qualities ~n / 1400.
template <typename T> constexpr uint32_t medium_cap() { return sizeof(T) == 4 ? 2048u : 1024u; }

template <typename T, uint32_t KM_>
struct MediumLds {
    static constexpr uint32_t KM = KM_;
    static constexpr int kMedThreads = KM / 2;                   // two consecutive positions per thread
    static constexpr int kMedSegs = KM / 64;                     // nodes of > 64 primitives on one level
    static constexpr int kMedNodes = KM >= 1024 ? 256 : KM / 4;  // local nodes (a segment whose lopsided splits need more gives up: the build
                                                                 //  is then retried on the plain Phase A path)
    T ctr[3][KM], blo[3][KM], bhi[3][KM];   // by SLOT = position at load time
    uint32_t order[KM];                     // order[position] = slot
    uint32_t ltab[KM], rtab[KM];            // Hoare violator tables (positions), per segment at [begin ..)
    uint16_t pre[KM + 2];                   // pre[p] = #primitives before position p that satisfy their node's partition predicate
    uint8_t seg_of[KM];                     // index of the position's node among this level's active nodes, 255: settled
    typename Ord<T>::U bin_lo[kMedSegs][3][kBins][3], bin_hi[kMedSegs][3][kBins][3];
    uint32_t bin_cnt[kMedSegs][3][kBins];
    T s_scale[kMedSegs][3], s_shift[kMedSegs][3], s_plane[kMedSegs];
    T axis_cost[kMedSegs][3];
    uint32_t axis_bin[kMedSegs][3];
    uint32_t s_node[kMedSegs], s_begin[kMedSegs], s_end[kMedSegs], s_mode[kMedSegs], s_axis[kMedSegs], s_wide[kMedSegs], s_m[kMedSegs], s_cut[kMedSegs];
    uint8_t s_next[kMedSegs][2];            // active index, in the next level, of the node's left / right range (255: <= 64 primitives)
    uint32_t act_next[kMedSegs];
    typename Ord<T>::U cb_lo[2 * kMedSegs][3], cb_hi[2 * kMedSegs][3];   // key boxes of the left / right range of every active node ...
    uint32_t cz_lo[2 * kMedSegs][3], cz_hi[2 * kMedSegs][3];             // ... and (position << 1 | sign) of their last zero-valued bound
    T cbox[2 * kMedSegs][6];                // decoded: {lo xyz, hi xyz}
    T nbox[kMedNodes][6];
    uint16_t nb[kMedNodes], ne[kMedNodes], nparent[kMedNodes], nchild[kMedNodes];
    uint8_t nwhich[kMedNodes];
    uint16_t small_nodes[kMedNodes];
    uint32_t wave_sums[kMedThreads / 64];
    uint32_t n_nodes, n_act, n_next, n_small, error, base, small_base, n_levels;
    uint16_t level_start[kMedLevels + 1];
};

template <typename T, uint32_t KM_>
__global__ void __launch_bounds__(KM_ / 2) k_medium(BuildCtx<T> c) {
    using Lds = MediumLds<T, KM_>;
    __shared__ Lds L;
    constexpr uint32_t KM = Lds::KM;
    constexpr int kMedThreads = Lds::kMedThreads, kMedSegs = Lds::kMedSegs, kMedNodes = Lds::kMedNodes;
    (void)kMedSegs;
    constexpr uint32_t PPT = KM / kMedThreads;                 // consecutive positions per thread
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t slot = blockIdx.x;
    if (c.med_cum[3]) {                                       // several class lists in one launch: which list, which entry
        uint32_t cls = 0;
        while (cls < 3 && slot >= c.med_cum[cls]) ++cls;
        slot = slot - (cls ? c.med_cum[cls - 1] : 0u) + cls * c.medium_slots;
    }
    const uint32_t root_id = c.medium_list[slot];
    const ANode<T>& R = c.nodes[root_id];
    const uint32_t B = R.begin, s = R.end - R.begin, tree = R.tree;
    MedInfo* info = c.med_info + slot;
    unsigned long long t_mark = c.med_prof ? __builtin_readcyclecounter() : 0ull;
    auto mark = [&](int i) { if (c.med_prof && tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); atomicAdd(&c.med_prof[i], now - t_mark); t_mark = now; } };

    // ---- load: one read of ids, centres and boxes per primitive for all the levels that follow
    for (uint32_t p = tid; p < KM; p += kMedThreads) {
        L.order[p] = p;
        L.seg_of[p] = p < s ? 0 : 255;
        if (p < s) {
            const uint32_t id = c.ids[B + p];
#pragma unroll
            for (int k = 0; k < 3; ++k) { L.ctr[k][p] = c.centers[3ull * id + k]; L.blo[k][p] = c.bboxes[6ull * id + k]; L.bhi[k][p] = c.bboxes[6ull * id + 3 + k]; }
        }
    }
    const bool root_box_pending = R.rank == kNone;             // a forest root whose box k_forest_roots left to this block
    if (tid < 3 && !root_box_pending) { L.nbox[0][tid] = R.lo[tid]; L.nbox[0][3 + tid] = R.hi[tid]; }
    if (tid < 3) { L.cb_lo[0][tid] = Ord<T>::enc(Ord<T>::kMax); L.cb_hi[0][tid] = Ord<T>::enc(-Ord<T>::kMax); L.cz_lo[0][tid] = 0; L.cz_hi[0][tid] = 0; }
    if (tid == 0) {
        L.nb[0] = 0; L.ne[0] = static_cast<uint16_t>(s); L.nparent[0] = 0; L.nwhich[0] = 0; L.nchild[0] = 0;
        L.n_nodes = 1; L.n_act = 1; L.s_node[0] = 0; L.n_small = 0; L.error = 0; L.n_levels = 1; L.level_start[0] = 0; L.level_start[1] = 1;
    }
    __syncthreads();
    if (root_box_pending) {
        // compute_bbox over the tree's range in position order (top_down_sah_builder.h:80, :133-139) from the data just loaded, instead of a
        // second gather of the same boxes by k_forest_roots; a zero bound takes the sign of the last zero in position order
        T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
        uint32_t zl[3] = {0, 0, 0}, zh[3] = {0, 0, 0};
        for (uint32_t p = tid; p < s; p += kMedThreads) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T a = L.blo[k][p], b = L.bhi[k][p];
                lo[k] = pick_min(lo[k], a); hi[k] = pick_max(hi[k], b);
                if (a == T(0)) zl[k] = (p << 1) | Ord<T>::sign(a);
                if (b == T(0)) zh[k] = (p << 1) | Ord<T>::sign(b);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
            const uint32_t wzl = wave_max_key(zl[k]), wzh = wave_max_key(zh[k]);
            if (lane == 0) { atomicMin(&L.cb_lo[0][k], klo); atomicMax(&L.cb_hi[0][k], khi); if (wzl) atomicMax(&L.cz_lo[0][k], wzl); if (wzh) atomicMax(&L.cz_hi[0][k], wzh); }
        }
        __syncthreads();
        if (tid < 3) {
            const T a = decode_bound<T>(L.cb_lo[0][tid], L.cz_lo[0][tid]), b = decode_bound<T>(L.cb_hi[0][tid], L.cz_hi[0][tid]);
            L.nbox[0][tid] = a; L.nbox[0][3 + tid] = b;
            ANode<T>& root = c.nodes[root_id];
            root.lo[tid] = a; root.hi[tid] = b;
            if (tid == 0) root.rank = 0;
        }
        __syncthreads();
    }
    mark(0);                                                   // load

    for (;;) {
        const uint32_t n_act = L.n_act;
        if (n_act == 0 || L.error) break;
        // ---- the level's active nodes: range, bin scale / offset (binned_sah_builder.h:88-89), empty bins
        if (tid < n_act) {
            const uint32_t nd = L.s_node[tid];
            L.s_begin[tid] = L.nb[nd]; L.s_end[tid] = L.ne[nd];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T scale = T(kBins) / (L.nbox[nd][3 + k] - L.nbox[nd][k]);
                L.s_scale[tid][k] = scale;
                L.s_shift[tid][k] = (-L.nbox[nd][k]) * scale;
            }
        }
        {
            const auto lo0 = Ord<T>::enc(Ord<T>::kMax), hi0 = Ord<T>::enc(-Ord<T>::kMax);
            for (uint32_t q = tid; q < n_act * 3 * kBins * 3; q += kMedThreads) { (&L.bin_lo[0][0][0][0])[q] = lo0; (&L.bin_hi[0][0][0][0])[q] = hi0; }
            for (uint32_t q = tid; q < n_act * 3 * kBins; q += kMedThreads) (&L.bin_cnt[0][0][0])[q] = 0;
        }
        __syncthreads();
        // ---- fill_bins (:82-99)
        uint32_t my_seg[PPT], my_slot[PPT];
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) {
            const uint32_t p = tid * PPT + i;
            my_seg[i] = L.seg_of[p];
            my_slot[i] = L.order[p];
            if (my_seg[i] == 255) continue;
            const uint32_t sg = my_seg[i], sl = my_slot[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t b = bin_of(Ord<T>::fma_(L.ctr[k][sl], L.s_scale[sg][k], L.s_shift[sg][k]));   // :92-95
#pragma unroll
                for (int j = 0; j < 3; ++j) { atomicMin(&L.bin_lo[sg][k][b][j], Ord<T>::enc(L.blo[j][sl])); atomicMax(&L.bin_hi[sg][k][b][j], Ord<T>::enc(L.bhi[j][sl])); }
                atomicAdd(&L.bin_cnt[sg][k][b], 1u);
            }
        }
        __syncthreads();
        mark(1);                                               // setup + fill_bins
        // ---- find_best_split: thread (node, axis) sweeps one axis of one node (:101-116)
        if (tid < 3 * n_act) {
            const uint32_t sg = tid / 3, k = tid % 3;
            if (static_cast<int>(k) < c.dim) {
                T cost; uint32_t bin;
                sweep_axis<T>([&](int i, T* lo, T* hi, uint32_t& n) {
                    for (int j = 0; j < 3; ++j) { lo[j] = Ord<T>::dec(L.bin_lo[sg][k][i][j]); hi[j] = Ord<T>::dec(L.bin_hi[sg][k][i][j]); }
                    n = L.bin_cnt[sg][k][i];
                }, cost, bin, c.dim, c.sah_log);
                L.axis_cost[sg][k] = cost;
                L.axis_bin[sg][k] = bin;
            }
        }
        __syncthreads();
        // ---- try_split's decision, one thread per node (:128-148; k_decide)
        if (tid < n_act) {
            const uint32_t nd = L.s_node[tid];
            T nlo[3], nhi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { nlo[k] = L.nbox[nd][k]; nhi[k] = L.nbox[nd][3 + k]; }
            const int wide = widest_axis(nlo, nhi, c.dim);
            uint32_t best_bin = kBins / 2; T best_cost = Ord<T>::kMax; int best_axis = wide;     // :132-133
            for (int k = 0; k < c.dim; ++k) {
                const T cst = L.axis_cost[tid][k];
                if (cst < best_cost) { best_cost = cst; best_bin = L.axis_bin[tid][k]; best_axis = k; }
            }
            const uint32_t size = L.s_end[tid] - L.s_begin[tid];
            const T stay = half_area(nlo, nhi, c.dim) * (sah_prims<T>(size, c.sah_log) - c.sah_ratio);   // split_heuristic.h:35-37
            L.s_wide[tid] = static_cast<uint32_t>(wide);
            L.s_axis[tid] = static_cast<uint32_t>(best_axis);
            if (best_cost >= stay) L.s_mode[tid] = MODE_FALLBACK;    // size > 64 > max_leaf_size: never a leaf here (:140-142)
            else {
                L.s_mode[tid] = MODE_PARTITION;
                L.s_plane[tid] = Ord<T>::fma_((nhi[best_axis] - nlo[best_axis]) / T(kBins), static_cast<T>(best_bin), nlo[best_axis]);   // :145-148
            }
        }
        __syncthreads();
        mark(2);                                               // sweeps + decision
        // ---- the partition predicate of every primitive (:151) and its running count in position order
        bool pred[PPT];
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) {
            const uint32_t sg = my_seg[i];
            pred[i] = sg != 255 && L.s_mode[sg] == MODE_PARTITION && L.ctr[L.s_axis[sg] == 0 ? 0 : (L.s_axis[sg] == 1 ? 1 : 2)][my_slot[i]] < L.s_plane[sg];
            mine += pred[i] ? 1u : 0u;
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= static_cast<uint32_t>(d)) incl += o; }
        if (lane == 63) L.wave_sums[wave] = incl;
        __syncthreads();
        {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += L.wave_sums[w];
            uint32_t run = before + incl - mine;
#pragma unroll
            for (uint32_t i = 0; i < PPT; ++i) { L.pre[tid * PPT + i] = static_cast<uint16_t>(run); run += pred[i] ? 1u : 0u; }
            if (tid == kMedThreads - 1) L.pre[KM] = static_cast<uint16_t>(run);
        }
        __syncthreads();
        if (tid < n_act) {
            const uint32_t m = L.pre[L.s_end[tid]] - L.pre[L.s_begin[tid]], size = L.s_end[tid] - L.s_begin[tid];
            if (L.s_mode[tid] == MODE_PARTITION && (m == 0 || m == size)) L.s_mode[tid] = MODE_FALLBACK;   // :152-153
            L.s_m[tid] = m;
        }
        __syncthreads();
        mark(3);                                               // predicate, scan, m
        // ---- std::partition as its Hoare permutation (k_scatter + k_swap): with m = #true, the misplaced elements left of
        //      begin + m (ascending) are swapped pairwise with the misplaced elements right of it (descending)
        uint32_t vkind[PPT], vrank[PPT];                       // 0: stays, 1: left violator, 2: right violator
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) {
            vkind[i] = 0; vrank[i] = 0;
            const uint32_t sg = my_seg[i];
            if (sg == 255 || L.s_mode[sg] != MODE_PARTITION) continue;
            const uint32_t p = tid * PPT + i, b = L.s_begin[sg], m = L.s_m[sg];
            const uint32_t t_before = L.pre[p] - L.pre[b];     // #true in [begin, p)
            if (p < b + m) {
                if (!pred[i]) { vkind[i] = 1; vrank[i] = (p - b) - t_before; L.ltab[b + vrank[i]] = p; }
            } else if (pred[i]) { vkind[i] = 2; vrank[i] = m - t_before - 1; L.rtab[b + vrank[i]] = p; }
        }
        __syncthreads();
        uint32_t moved[PPT];
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) {
            moved[i] = my_slot[i];
            if (vkind[i] == 1) moved[i] = L.order[L.rtab[L.s_begin[my_seg[i]] + vrank[i]]];
            else if (vkind[i] == 2) moved[i] = L.order[L.ltab[L.s_begin[my_seg[i]] + vrank[i]]];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) if (vkind[i]) L.order[tid * PPT + i] = moved[i];
        __syncthreads();
        mark(4);                                               // Hoare permutation
        // ---- fallback_split (:118-126): std::partial_sort replayed by one thread per node that needs it; the cut of every node
        if (tid < n_act) {
            const uint32_t b = L.s_begin[tid], e = L.s_end[tid];
            if (L.s_mode[tid] == MODE_FALLBACK) {
                const uint32_t mid = (B + b + B + e + 1) / 2 - B;                 // absolute indices, :119
                const T* keys = L.ctr[L.s_wide[tid] == 0 ? 0 : (L.s_wide[tid] == 1 ? 1 : 2)];
                partial_sort_replay(L.order + b, long(mid - b), long(e - b), [=](uint32_t slot) { return keys[slot]; });
                L.s_cut[tid] = mid;
            } else L.s_cut[tid] = b + L.s_m[tid];
        }
        __syncthreads();
        mark(5);                                               // fallback / cut
        // ---- compute_bbox of both sides (top_down_sah_builder.h:96-97, :133-139). Every position joins the key box of its (node, side)
        //      range; a wave whose positions all lie in ONE range (the common case: a wave covers 64 x PPT consecutive positions) folds
        //      them with shuffles first and touches the LDS words once. A zero bound takes the sign of the LAST zero in position order
        //      (build_common.h: SlotState), tracked as max(position << 1 | sign).
        for (uint32_t q = tid; q < 2 * n_act * 3; q += kMedThreads) {
            (&L.cb_lo[0][0])[q] = Ord<T>::enc(Ord<T>::kMax); (&L.cb_hi[0][0])[q] = Ord<T>::enc(-Ord<T>::kMax); (&L.cz_lo[0][0])[q] = 0; (&L.cz_hi[0][0])[q] = 0;
        }
        __syncthreads();
        {
            uint32_t rng[PPT];
            bool uniform = true;
#pragma unroll
            for (uint32_t i = 0; i < PPT; ++i) {
                const uint32_t sg = my_seg[i], p = tid * PPT + i;
                rng[i] = sg == 255 ? 0xFFFFu : 2 * sg + (p >= L.s_cut[sg] ? 1u : 0u);
                uniform = uniform && rng[i] == rng[0];
            }
            const uint32_t r0 = __shfl(rng[0], 0);
            const bool wave_uniform = __ballot(!(uniform && rng[0] == r0)) == 0;
            if (wave_uniform) {
                if (r0 != 0xFFFFu) {
                    T lo[3] = { Ord<T>::kMax, Ord<T>::kMax, Ord<T>::kMax }, hi[3] = { -Ord<T>::kMax, -Ord<T>::kMax, -Ord<T>::kMax };
                    uint32_t zl[3] = {0, 0, 0}, zh[3] = {0, 0, 0};
#pragma unroll
                    for (uint32_t i = 0; i < PPT; ++i) {
                        const uint32_t p = tid * PPT + i, sl = L.order[p];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const T a = L.blo[k][sl], b = L.bhi[k][sl];
                            lo[k] = pick_min(lo[k], a); hi[k] = pick_max(hi[k], b);
                            if (a == T(0)) zl[k] = (p << 1) | Ord<T>::sign(a);
                            if (b == T(0)) zh[k] = (p << 1) | Ord<T>::sign(b);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
                        const uint32_t wzl = wave_max_key(zl[k]), wzh = wave_max_key(zh[k]);
                        if (lane == 0) {
                            atomicMin(&L.cb_lo[r0][k], klo); atomicMax(&L.cb_hi[r0][k], khi);
                            if (wzl) atomicMax(&L.cz_lo[r0][k], wzl);
                            if (wzh) atomicMax(&L.cz_hi[r0][k], wzh);
                        }
                    }
                }
            } else {
#pragma unroll
                for (uint32_t i = 0; i < PPT; ++i) {
                    if (rng[i] == 0xFFFFu) continue;
                    const uint32_t p = tid * PPT + i, sl = L.order[p], r = rng[i];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const T a = L.blo[k][sl], b = L.bhi[k][sl];
                        atomicMin(&L.cb_lo[r][k], Ord<T>::enc(a)); atomicMax(&L.cb_hi[r][k], Ord<T>::enc(b));
                        if (a == T(0)) atomicMax(&L.cz_lo[r][k], (p << 1) | Ord<T>::sign(a));
                        if (b == T(0)) atomicMax(&L.cz_hi[r][k], (p << 1) | Ord<T>::sign(b));
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t q = tid; q < 2 * n_act * 3; q += kMedThreads) {
            const uint32_t r = q / 3, k = q % 3;
            L.cbox[r][k] = decode_bound<T>(L.cb_lo[r][k], L.cz_lo[r][k]);
            L.cbox[r][3 + k] = decode_bound<T>(L.cb_hi[r][k], L.cz_hi[r][k]);
        }
        __syncthreads();
        mark(6);                                               // child boxes
        // ---- child creation with SATO order (:91-113; k_finalize), by the first wave: node t's children are local nodes n + 2t, n + 2t + 1
        const uint32_t n_before = L.n_nodes;
        if (n_before + 2 * n_act > static_cast<uint32_t>(kMedNodes) || L.n_levels + 1 > static_cast<uint32_t>(kMedLevels)) {
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
                first = half_area(&L.cbox[2 * lane][0], &L.cbox[2 * lane][3], c.dim) < half_area(&L.cbox[2 * lane + 1][0], &L.cbox[2 * lane + 1][3], c.dim) ? 1 : 0;   // :105-108
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
            // next level's active nodes, in (node, side) order; the others (<= 64 primitives) go to Phase B
            const uint64_t b0 = __ballot(on && big[0]), b1 = __ballot(on && big[1]), below = (uint64_t{1} << lane) - 1;
            const uint64_t s0 = __ballot(on && !big[0]), s1 = __ballot(on && !big[1]);
            if (on) {
                const uint32_t at = __popcll(b0 & below) + __popcll(b1 & below);
                const uint32_t sm = L.n_small + __popcll(s0 & below) + __popcll(s1 & below);
                uint32_t a = at, q = sm;
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
        // ---- every position learns its node of the next level
#pragma unroll
        for (uint32_t i = 0; i < PPT; ++i) {
            const uint32_t sg = my_seg[i];
            if (sg == 255) continue;
            const uint32_t p = tid * PPT + i;
            L.seg_of[p] = L.s_next[sg][p >= L.s_cut[sg] ? 1 : 0];
        }
        if (tid < L.n_next) L.s_node[tid] = L.act_next[tid];
        if (tid == 0) L.n_act = L.n_next;
        __syncthreads();
        mark(7);                                               // children + next level
    }

    if (L.error) {                                             // give up: the host retries the whole build without k_medium
        if (tid == 0) { atomicOr(&c.counters->error, 4u); info->count = 0; info->base = 0; info->n_levels = 0; }
        return;
    }
    // ---- results: the permuted ids, the nodes (appended to c.nodes), the <= 64-primitive pieces for Phase B
    uint32_t out_id[PPT];
#pragma unroll
    for (uint32_t i = 0; i < PPT; ++i) { const uint32_t p = tid * PPT + i; out_id[i] = p < s ? c.ids[B + L.order[p]] : 0u; }
    if (tid == 0) {
        const uint32_t extra = L.n_nodes - 1;
        L.base = atomicAdd(&c.counters->n_nodes, extra);
        L.small_base = atomicAdd(&c.counters->n_small, L.n_small);
        if (L.base + extra > c.node_cap || L.small_base + L.n_small > c.node_cap) { atomicOr(&c.counters->error, 2u); L.error = 1; }
    }
    __syncthreads();                                           // (also: every read of c.ids above precedes every write below)
    if (L.error) { if (tid == 0) { info->count = 0; info->base = 0; info->n_levels = 0; } return; }
#pragma unroll
    for (uint32_t i = 0; i < PPT; ++i) { const uint32_t p = tid * PPT + i; if (p < s) c.ids[B + p] = out_id[i]; }
    const uint32_t base = L.base, n_nodes = L.n_nodes;
    auto global_id = [&](uint32_t t) { return t == 0 ? root_id : base + t - 1; };
    for (uint32_t t = 1 + tid; t < n_nodes; t += kMedThreads) {
        ANode<T> nd;
#pragma unroll
        for (int k = 0; k < 3; ++k) { nd.lo[k] = L.nbox[t][k]; nd.hi[k] = L.nbox[t][3 + k]; }
        nd.begin = B + L.nb[t]; nd.end = B + L.ne[t];
        nd.child = L.nchild[t] ? global_id(L.nchild[t]) : kNone;
        nd.parent = global_id(L.nparent[t]) | (static_cast<uint32_t>(L.nwhich[t]) << 31);
        nd.kind = L.nchild[t] ? KIND_INNER : KIND_SMALL;
        nd.ic = 0; nd.rank = 0; nd.tree = tree;
        c.nodes[global_id(t)] = nd;
    }
    for (uint32_t q = tid; q < L.n_small; q += kMedThreads) c.small_list[L.small_base + q] = global_id(L.small_nodes[q]);
    if (tid == 0) {
        ANode<T>& root = c.nodes[root_id];
        root.child = global_id(L.nchild[0]);
        root.kind = KIND_INNER;
        info->base = base; info->count = n_nodes; info->n_levels = L.n_levels;
    }
    for (uint32_t q = tid; q <= L.n_levels; q += kMedThreads) info->level_start[q] = L.level_start[q];
    mark(8);                                                   // results
    if (c.med_prof && tid == 0) atomicAdd(&c.med_prof[9], static_cast<unsigned long long>(L.n_levels));
}

} // namespace

// ---- host side ---------------------------------------------------------------------------------------------

namespace {

template <typename T>
struct BinnedWs {
    DevBuf<uint32_t> ids, chunk_true, ltab, rtab, small_list, medium_list;
    DevBuf<MedInfo> med_info;
    DevBuf<unsigned long long> med_prof;
    DevBuf<ANode<T>> nodes;
    DevBuf<SlotBins<T>> bins;
    DevBuf<SlotState<T>> st_a, st_b;
    DevBuf<Task> tk_a, tk_b;
    DevBuf<HostNode<T>> stage;
    DevBuf<Counters> counters;

    // capacities: typical on the first attempt, worst case (degenerate chains of big nodes) on retry
    int alloc(BuildCtx<T>& c, uint32_t n, uint32_t roots, int attempt, bool own_ids, int nb = kBins) {
        const uint32_t node_cap = (attempt == 0 ? n / 8 + 1024 : 2 * n + 2) + roots;
        const uint32_t slot_cap = n / (kSmall + 1) + 2;
        const uint32_t task_cap = n / kChunk + slot_cap + 2;
        hipError_t e = hipSuccess;
        auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        if (own_ids) A(ids.alloc(n));
        A(chunk_true.alloc(task_cap)); A(ltab.alloc(n)); A(rtab.alloc(n)); A(small_list.alloc(node_cap));
        // (a slot's bins hold 3 * nb boxes + counts: a BinCount other than the default takes proportionally more of the default-sized records)
        A(nodes.alloc(node_cap)); A(bins.alloc((size_t{slot_cap} * static_cast<size_t>(nb) + kBins - 1) / kBins)); A(st_a.alloc(slot_cap)); A(st_b.alloc(slot_cap));
        A(tk_a.alloc(task_cap)); A(tk_b.alloc(task_cap)); A(stage.alloc(2 * size_t{n})); A(counters.alloc(1));
        // segments of 65 .. medium_cap primitives go to k_medium — on the first attempt only: a retry (capacity exceeded, or a segment
        // whose lopsided splits outgrow k_medium's local tables) takes the plain Phase A path
        static const bool medium_off = BVH_DEV_INT("BVH_AMD_MEDIUM", 1) == 0;   // A/B runs
        const uint32_t medium_slots = n / (kSmall + 1) + roots + 2;
        c.medium_cap = attempt == 0 && !medium_off && nb == kBins ? medium_cap<T>() : 0u;   // (k_medium is written for the default BinCount)
        c.medium_slots = medium_slots;
        // segments are listed by size class; run_binned_phases decides which kernel serves which list
        c.medium_min_class = 0;
        if (c.medium_cap) { A(medium_list.alloc(4 * size_t{medium_slots})); A(med_info.alloc(4 * size_t{medium_slots})); }
        if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("build: hipMalloc: ") + hipGetErrorString(e));
        c.medium_list = medium_list.p; c.med_info = med_info.p;
        static const bool prof = BVH_DEV_INT("BVH_AMD_MED_PROF", 0) != 0;   // developer knob
        if (prof && c.medium_cap && med_prof.alloc(16) == hipSuccess && hipMemset(med_prof.p, 0, 16 * sizeof(unsigned long long)) == hipSuccess) c.med_prof = med_prof.p;
        if (own_ids) c.ids = ids.p;
        c.n = n; c.nodes = nodes.p; c.node_cap = node_cap; c.bins = bins.p; c.state = st_a.p; c.state_next = st_b.p;
        c.slot_cap = slot_cap; c.tasks = tk_a.p; c.tasks_next = tk_b.p; c.task_cap = task_cap; c.chunk_true = chunk_true.p;
        c.ltab = ltab.p; c.rtab = rtab.p; c.small_list = small_list.p; c.stage = stage.p; c.counters = counters.p;
        return BVH_AMD_OK;
    }
};

// Phase A levels + Phase B. Expects the roots already registered (state_next / tasks_next / counters) and `h` read back.
// nb: BinnedSahBuilder's BinCount (binned_sah_builder.h:18). 8 = the reference's default and the tuned path (k_medium, level-synchronous
// Phase B); 4 / 16 / 32 run the same Phase A levels with their own fill_bins / decision kernels and the node-by-node Phase B.
template <typename T>
int run_binned_phases(BuildCtx<T>& c, Counters& h, std::vector<uint32_t>& level_start, bool& overflow, hipStream_t stream,
                      const std::function<int(PhaseB*)>* roots_final = nullptr, int nb = kBins) {
    uint32_t n_active = h.n_active_next, n_tasks = h.n_tasks_next;
    overflow = h.error != 0;
    while (n_active > 0 && !overflow) {
        std::swap(c.state, c.state_next);
        std::swap(c.tasks, c.tasks_next);
        BVH_HIP_TRY(hipMemsetAsync(&c.counters->n_active_next, 0, 3 * sizeof(uint32_t), stream), BVH_AMD_ERR_HIP);
        const unsigned slot_grid = (n_active + 63) / 64;
        switch (nb) {
#define BVH_BINNED_LEVEL(NB)                                                                                 \
        case NB:                                                                                             \
            hipLaunchKernelGGL((k_init_slots<T, NB>), dim3(n_active), dim3(64), 0, stream, c);               \
            hipLaunchKernelGGL((k_bin<T, NB>), dim3(n_tasks), dim3(256), 0, stream, c);                      \
            hipLaunchKernelGGL((k_decide<T, NB>), dim3(slot_grid), dim3(64), 0, stream, c, n_active);        \
            break;
        BVH_BINNED_LEVEL(4) BVH_BINNED_LEVEL(8) BVH_BINNED_LEVEL(16) BVH_BINNED_LEVEL(32)
#undef BVH_BINNED_LEVEL
        default: return fail(BVH_AMD_ERR_UNSUPPORTED, "build: BinCount must be 4, 8, 16 or 32");
        }
        hipLaunchKernelGGL(k_count<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_scatter<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_fallback<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
        hipLaunchKernelGGL(k_swap<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_child_bounds<T>, dim3(n_tasks), dim3(256), 0, stream, c);
        hipLaunchKernelGGL(k_finalize<T>, dim3(slot_grid), dim3(64), 0, stream, c, n_active);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        { int rb_ = readback(&h, c.counters, sizeof(h), stream); if (rb_) return rb_; }
        overflow = h.error != 0;
        level_start.push_back(h.n_nodes);
        n_active = h.n_active_next;
        n_tasks = h.n_tasks_next;
    }
    const uint32_t n_medium_all = h.n_medium[0] + h.n_medium[1] + h.n_medium[2] + h.n_medium[3];
    if (!overflow && n_medium_all) {
        // Segments are LISTED by size class (<= 256 << cls primitives); which kernel serves a list is decided here, from the lists'
        // lengths. A block's levels are a chain of short LDS phases, so the largest kernel (1024 threads alone on a CU) finishes a
        // segment fastest and a smaller one only pays when there are enough segments to fill the slots it opens: the 256 class always
        // (128 threads, eight blocks per CU), the 512 class from three blocks per CU on (Quality::Low with a pool on 1M uniformly
        // spread triangles: 1036 mini-trees of 257..300 primitives, 0.21 -> 0.12 ms), the 1024 class never (1054 segments: no gain).
        // Lists served by the largest kernel go into ONE launch (their tails would add up). profiles/r04_build_medium_classes_ab.txt
        static const int policy = BVH_DEV_INT("BVH_AMD_MEDIUM_CLASSES", 0);   // A/B runs: 1 always own, 2 never
        constexpr uint32_t largest = sizeof(T) == 4 ? 3u : 2u;
        int cus = 256;
        { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; }
        uint32_t served_by[4];
        for (uint32_t cls = 0; cls < 4; ++cls) {
            const bool own = cls == 0 || cls >= largest || policy == 1 || (policy == 0 && cls == 1 && h.n_medium[1] >= 3u * static_cast<uint32_t>(cus));
            served_by[cls] = own ? std::min(cls, largest) : largest;
        }
        for (uint32_t cls = 0; cls < largest; ++cls) {
            if (!h.n_medium[cls] || served_by[cls] != cls) continue;
            BuildCtx<T> cc = c;
            cc.medium_list = c.medium_list + size_t{cls} * c.medium_slots;
            cc.med_info = c.med_info + size_t{cls} * c.medium_slots;
            if (cls == 0) hipLaunchKernelGGL((k_medium<T, 256>), dim3(h.n_medium[cls]), dim3(128), 0, stream, cc);
            else if (cls == 1) hipLaunchKernelGGL((k_medium<T, 512>), dim3(h.n_medium[cls]), dim3(256), 0, stream, cc);
            else hipLaunchKernelGGL((k_medium<T, 1024>), dim3(h.n_medium[cls]), dim3(512), 0, stream, cc);
        }
        {
            BuildCtx<T> cc = c;
            uint32_t total = 0;
            for (uint32_t cls = 0; cls < 4; ++cls) { if (served_by[cls] == largest && cls <= largest) total += h.n_medium[cls]; cc.med_cum[cls] = total; }
            if (total) {
                if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((k_medium<T, 2048>), dim3(total), dim3(1024), 0, stream, cc);
                else hipLaunchKernelGGL((k_medium<T, 1024>), dim3(total), dim3(512), 0, stream, cc);
            }
        }
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        const uint32_t n_before[4] = { h.n_medium[0], h.n_medium[1], h.n_medium[2], h.n_medium[3] };
        { int rb_ = readback(&h, c.counters, sizeof(h), stream); if (rb_) return rb_; }     // nodes and small pieces it added
        overflow = h.error != 0;
        if (c.med_prof) {                                     // developer knob: where a block's time goes (clock ticks summed over blocks)
            unsigned long long t[16] = {};
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(t, c.med_prof, sizeof(t), hipMemcpyDeviceToHost);
            static const char* names[9] = {"load", "setup+fill_bins", "sweeps+decision", "predicate+scan", "hoare", "fallback/cut", "child boxes", "children+next", "results"};
            unsigned long long sum = 0;
            for (int i = 0; i < 9; ++i) sum += t[i];
            fprintf(stderr, "[k_medium] blocks by class %u / %u / %u / %u, %.1f levels per block, ticks per block:", n_before[0], n_before[1], n_before[2], n_before[3],
                    double(t[9]) / n_medium_all);
            for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.0f (%.0f%%)", names[i], double(t[i]) / n_medium_all, 100.0 * double(t[i]) / double(sum ? sum : 1));
            fprintf(stderr, "\n");
            (void)hipMemset(c.med_prof, 0, sizeof(t));
        }
    }
    // every root's box is final in stream order here (k_forest_roots, or k_medium for the roots it left pending): a forest caller that
    // only needs those boxes — the mini-tree builder's top level without pruning — starts its own work beside Phase B
    PhaseB lane;
    if (!overflow && roots_final) { const int rc = (*roots_final)(&lane); if (rc) return rc; }
    if (!overflow && h.n_small) {
        static const bool dfs = BVH_DEV_IS("BVH_AMD_SMALL", "dfs");   // the node-by-node walk (A/B runs)
        hipStream_t on = lane.stream ? lane.stream : stream;
        if (lane.stream) BVH_HIP_TRY(hipStreamWaitEvent(lane.stream, lane.start, 0), BVH_AMD_ERR_HIP);
        if (nb == 4) hipLaunchKernelGGL((k_small<T, 4>), dim3((h.n_small + 3) / 4), dim3(256), 0, on, c, h.n_small);
        else if (nb == 16) hipLaunchKernelGGL((k_small<T, 16>), dim3((h.n_small + 3) / 4), dim3(256), 0, on, c, h.n_small);
        else if (nb == 32) hipLaunchKernelGGL((k_small<T, 32>), dim3((h.n_small + 3) / 4), dim3(256), 0, on, c, h.n_small);
        else if (dfs) hipLaunchKernelGGL(k_small<T>, dim3((h.n_small + 3) / 4), dim3(256), 0, on, c, h.n_small);
        else hipLaunchKernelGGL(k_small_levels<T>, dim3((h.n_small + 1) / 2), dim3(128), 0, on, c, h.n_small);
        if (lane.stream) {
            BVH_HIP_TRY(hipEventRecord(lane.done, lane.stream), BVH_AMD_ERR_HIP);
            BVH_HIP_TRY(hipStreamWaitEvent(stream, lane.done, 0), BVH_AMD_ERR_HIP);
        }
    }
    return BVH_AMD_OK;
}

// Roots of a forest: tree g covers positions [group_begin[g], group_begin[g + 1]) of c.ids. One block per tree
// computes compute_bbox over its range in position order (top_down_sah_builder.h:80, :133-139).
template <typename T>
__global__ void __launch_bounds__(256) k_forest_roots(BuildCtx<T> c, const uint32_t* group_begin) {
    __shared__ typename Ord<T>::U slo[3], shi[3];
    __shared__ uint32_t zlo[3], zhi[3];
    const uint32_t g = blockIdx.x, b = group_begin[g], e = group_begin[g + 1];
    if (e - b > static_cast<uint32_t>(kSmall) && e - b <= c.medium_cap) {
        // k_medium loads this tree's boxes anyway and computes the root box from them (rank = kNone marks it as pending)
        if (threadIdx.x == 0) {
            ANode<T>& r = c.nodes[g];
            for (int k = 0; k < 3; ++k) { r.lo[k] = T(0); r.hi[k] = T(0); }
            r.begin = b; r.end = e; r.child = kNone; r.parent = kNone; r.ic = 0; r.rank = kNone; r.tree = g;
            emit_child(c, g);
        }
        return;
    }
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
    for (int k = 0; k < 3; ++k) {
        const auto klo = wave_min_key(Ord<T>::enc(lo[k])), khi = wave_max_key(Ord<T>::enc(hi[k]));
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], klo); atomicMax(&shi[k], khi); }
    }
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
    const int nb = static_cast<int>(ambient_sah().bin_count);  // BinnedSahBuilder<Node, BinCount>: 8 unless bvhXX_build_device_binned says otherwise
    if (nb != 4 && nb != 8 && nb != 16 && nb != 32) return fail(BVH_AMD_ERR_UNSUPPORTED, "build: BinCount must be 4, 8, 16 or 32");
    BVH_HIP_TRY(hipGetDevice(&out.device), BVH_AMD_ERR_HIP);
    for (int attempt = 0; attempt < 2; ++attempt) {
        BinnedWs<T> ws;
        DevBuf<HostNode<T>> final_nodes;
        BuildCtx<T> c;
        c.bboxes = d_bboxes; c.centers = d_centers;
        c.min_leaf = static_cast<uint32_t>(cfg.min_leaf_size); c.max_leaf = static_cast<uint32_t>(cfg.max_leaf_size);
        c.dim = out.dim;
        c.sah_log = ambient_sah().log_cluster; c.sah_ratio = static_cast<T>(ambient_sah().cost_ratio);
        int rc = ws.alloc(c, static_cast<uint32_t>(n), 1, attempt, true, nb);
        if (rc) return rc;
        hipLaunchKernelGGL(k_prepare_root<T>, dim3(1), dim3(1), 0, stream, c);
        const unsigned root_grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 2048));
        hipLaunchKernelGGL(k_init_root<T>, dim3(root_grid), dim3(256), 0, stream, c, true);
        hipLaunchKernelGGL(k_make_root<T>, dim3(1), dim3(1), 0, stream, c);
        Counters h;
        { int rb_ = readback(&h, c.counters, sizeof(h), stream); if (rb_) return rb_; }
        std::vector<uint32_t> level_start{0, 1};              // A-node id ranges per level
        bool overflow = false;
        rc = run_binned_phases(c, h, level_start, overflow, stream, nullptr, nb);
        if (rc) return rc;
        if (overflow) {
            if (attempt == 0) continue;
            return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
        }
        rc = number_and_emit<T>(out, c, level_start, h.n_nodes, h.n_small, final_nodes, stream, h.n_medium);
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
                               DevBuf<uint32_t>& tree_node_off, uint32_t& total_nodes, hipStream_t stream,
                               const std::function<int(const ANode<T>*, PhaseB*)>& roots_ready)
{
    bool roots_announced = false;                             // roots_ready(first n_groups working nodes) runs once, also across a retry
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
        { int rb_ = readback(&h, c.counters, sizeof(h), stream); if (rb_) return rb_; }
        std::vector<uint32_t> level_start{0, n_groups};
        bool overflow = false;
        const std::function<int(PhaseB*)> announce = [&](PhaseB* lane) -> int {
            if (roots_announced || !roots_ready) return BVH_AMD_OK;
            roots_announced = true;
            return roots_ready(c.nodes, lane);
        };
        rc = run_binned_phases(c, h, level_start, overflow, stream, &announce);
        if (rc) return rc;
        if (overflow) {
            if (attempt == 0) continue;
            return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
        }
        rc = number_nodes<T>(c, level_start, stream, h.n_medium);
        if (rc) return rc;
        BVH_HIP_TRY(tree_node_off.alloc(n_groups + 1), BVH_AMD_ERR_HIP);
        hipLaunchKernelGGL(k_forest_offsets<T>, dim3(1), dim3(1024), 0, stream, c, n_groups, tree_node_off.p);
        { int rb_ = readback(&total_nodes, tree_node_off.p + n_groups, sizeof(uint32_t), stream); if (rb_) return rb_; }
        BVH_HIP_TRY(trees.alloc(total_nodes), BVH_AMD_ERR_HIP);
        c.tree_node_off = tree_node_off.p;
        c.tree_begin = d_group_begin;
        hipLaunchKernelGGL(k_emit_tree<T>, dim3((h.n_nodes + 255) / 256), dim3(256), 0, stream, c, h.n_nodes, trees.p);
        if (h.n_small) hipLaunchKernelGGL(k_emit_small<T>, dim3((h.n_small + 3) / 4), dim3(256), 0, stream, c, h.n_small, trees.p);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        if (!scratch_pool_enabled()) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // plain hipFree of the workspace on return (the pool frees in stream order)
        return BVH_AMD_OK;
    }
    return fail(BVH_AMD_ERR_OVERFLOW, "build: internal capacity exceeded");
}

template int build_binned_device<float>(BvhImpl<float>&, const float*, const float*, size_t, const bvh_build_config&, hipStream_t);
template int build_binned_device<double>(BvhImpl<double>&, const double*, const double*, size_t, const bvh_build_config&, hipStream_t);
template int build_binned_forest_device<float>(const float*, const float*, uint32_t*, uint32_t, const uint32_t*, uint32_t,
    const bvh_build_config&, DevBuf<HostNode<float>>&, DevBuf<uint32_t>&, uint32_t&, hipStream_t, const std::function<int(const ANode<float>*, PhaseB*)>&);
template int build_binned_forest_device<double>(const double*, const double*, uint32_t*, uint32_t, const uint32_t*, uint32_t,
    const bvh_build_config&, DevBuf<HostNode<double>>&, DevBuf<uint32_t>&, uint32_t&, hipStream_t, const std::function<int(const ANode<double>*, PhaseB*)>&);

} // namespace bvh_amd
