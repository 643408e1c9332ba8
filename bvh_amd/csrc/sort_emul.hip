// Device sorts whose ORDER is part of the reference's observable output.
//
//  * radix_sort_pairs: stable LSD radix sort (8 bits per pass) of (key, value) pairs. Reproduces any
//    stable sort by an integer key, e.g. "ids ascending within a grid bin" (mini_tree_builder.h:124).
//  * std_sort_ids: libstdc++ 11 `std::sort(ids, ids + n, [&](i, j) { return key[i] < key[j]; })`,
//    including the arrangement of EQUAL keys, which is what SweepSahBuilder (sweep_sah_builder.h:57-63) and
//    ReinsertionOptimizer (reinsertion_optimizer.h:256) leak into their results:
//      introsort = repeat { median-of-3 to front (stl_algo.h:79-103); __unguarded_partition (:1878-1895) }
//      on every segment longer than 16 with a depth budget of 2*floor(log2 n), then one insertion sort,
//      which is a STABLE sort of whatever arrangement the partition phase left.
//    Segments are disjoint, so the partition phase runs level-synchronously, one block per segment, using
//    the exact characterisation of the Hoare-style loop: with L = ascending positions whose key is >= pivot and
//    R = descending positions whose key is <= pivot, the loop swaps L_j <-> R_j for j < k = #{j : L_j < R_j} and
//    returns cut = (k == 0) ? L_0 : (L_k exists and L_k < R_{k-1} ? L_k : R_{k-1}).
//    The stable finish is a radix sort on the order-preserving integer image of the key.
//    The depth-exhausted heap-sort branch (stl_algo.h:1933) is replayed by one lane (never observed).

#include "build_common.h"

namespace bvh_amd {

using namespace bld;

namespace {

// ---------------------------------------------------------------------------------------------------
// stable LSD radix sort
// ---------------------------------------------------------------------------------------------------
// Elements per block: 16 per lane. Round 5: 4-byte keys take the 8192-element tile too (75 KB of LDS, two blocks per CU: runs per digit
// twice as long = fuller lines written; scatter of 2^24 ray keys 97 -> 86 us per pass, the histogram 24 -> 19; builds unchanged).
template <typename K> struct RadixShape {
    static constexpr int kTile = sizeof(K) == 2 || sizeof(K) == 4 ? 8192 : 4096;
    static constexpr int kThreads = kTile / 16;
    static constexpr int kWaves = kThreads / 64;
};
constexpr int kRadixMinTile = 4096;     // sizes the histogram scratch for any key type

template <typename K>
__global__ void __launch_bounds__(RadixShape<K>::kThreads) k_radix_hist(const K* keys, uint32_t n, uint32_t blocks_per_array, int shift,
                                                                       uint32_t* hist) {
    constexpr int kTile = RadixShape<K>::kTile, kThreads = RadixShape<K>::kThreads;
    __shared__ uint32_t h[256];
    const uint32_t arr = blockIdx.x / blocks_per_array, blk = blockIdx.x % blocks_per_array;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = size_t{arr} * n;
    const uint32_t b = blk * kTile, e = min(n, b + kTile);
    for (uint32_t i = b + threadIdx.x; i < e; i += uint32_t(kThreads))
        atomicAdd(&h[(keys[base + i] >> shift) & 0xFF], 1u);
    __syncthreads();
    if (threadIdx.x < 256) hist[(size_t{arr} * 256 + threadIdx.x) * blocks_per_array + blk] = h[threadIdx.x];    // digit-major, block-minor
}

// exclusive scan of one array's histogram (256 * blocks entries), one block per array
__global__ void __launch_bounds__(1024) k_radix_scan(uint32_t* hist, uint32_t entries) {
    __shared__ uint32_t part[1024];
    uint32_t* h = hist + size_t{blockIdx.x} * entries;
    const uint32_t per = (entries + 1023) / 1024;
    const uint32_t b = min(entries, threadIdx.x * per), e = min(entries, b + per);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += h[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = b; i < e; ++i) { uint32_t v = h[i]; h[i] = run; run += v; }
}

// Stable scatter of one tile. Wave w owns the w-th slice of the tile (16 elements per lane, kept in registers) and counts its
// digits into a private LDS histogram; the block turns the histograms into tile-local cursors (digit start + the lower waves'
// counts); each wave then ranks its elements 64 at a time (ballot match, no block barrier) and drops them at their tile-local
// sorted position in LDS; finally the tile is written out in sorted order, so that consecutive lanes store consecutive
// addresses of one digit's run (scattering straight from the ranking loop wrote 8-16 byte pieces and ran 2-3x slower).
// Equal digits keep their input order (lower wave first, then step, then lane). vals == nullptr: the values are the input
// positions (first pass over fresh keys); keys_out == nullptr: the keys are not needed any more (last pass).
template <typename K>
__global__ void __launch_bounds__(RadixShape<K>::kThreads) k_radix_scatter(const K* keys, const uint32_t* vals, K* keys_out, uint32_t* vals_out,
                                                                          uint32_t n, uint32_t blocks_per_array, int shift, const uint32_t* hist) {
    constexpr int kTile = RadixShape<K>::kTile, kThreads = RadixShape<K>::kThreads, kWaves = RadixShape<K>::kWaves, kSteps = 16;
    __shared__ uint32_t cursor[kWaves][256];
    __shared__ uint32_t lstart[256], goff[256], part[256];
    __shared__ K skey[kTile];
    __shared__ uint32_t sval[kTile];
    const uint32_t arr = blockIdx.x / blocks_per_array, blk = blockIdx.x % blocks_per_array;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kWaves * 256; i += kThreads) (&cursor[0][0])[i] = 0;
    __syncthreads();
    const size_t base = size_t{arr} * n;
    const uint32_t b = blk * uint32_t(kTile), e = min(n, b + uint32_t(kTile));
    const uint32_t first = b + wave * uint32_t(kSteps * 64) + lane;
    K key[kSteps]; uint32_t val[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const uint32_t i = first + s * 64;
        key[s] = 0; val[s] = 0;
        if (i < e) {
            key[s] = keys[base + i]; val[s] = vals ? vals[base + i] : i;
            atomicAdd(&cursor[wave][(key[s] >> shift) & 0xFF], 1u);
        }
    }
    __syncthreads();
    uint32_t total = 0;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) total += cursor[w][threadIdx.x];
        part[threadIdx.x] = total;
    }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                         // inclusive scan of the 256 digit totals
        uint32_t v = 0;
        if (threadIdx.x < 256 && int(threadIdx.x) >= off) v = part[threadIdx.x - off];
        __syncthreads();
        if (threadIdx.x < 256) part[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x < 256) {
        uint32_t run = part[threadIdx.x] - total;
        lstart[threadIdx.x] = run;
        goff[threadIdx.x] = hist[(size_t{arr} * 256 + threadIdx.x) * blocks_per_array + blk];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const uint32_t c = cursor[w][threadIdx.x]; cursor[w][threadIdx.x] = run; run += c; }
    }
    __syncthreads();
    uint32_t* mine = cursor[wave];
    const uint64_t below = (uint64_t{1} << lane) - 1;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const bool in = first + s * 64 < e;
        const uint32_t d = (key[s] >> shift) & 0xFF;
        uint64_t same = __ballot(in);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const uint64_t bal = __ballot((d >> bit) & 1);
            same &= ((d >> bit) & 1) ? bal : ~bal;
        }
        if (in) {
            const uint32_t rank = __popcll(same & below);
            const uint32_t at = mine[d];                              // every lane of the group reads before its leader advances the cursor
            __builtin_amdgcn_wave_barrier();
            if (rank == 0) mine[d] = at + __popcll(same);
            skey[at + rank] = key[s];
            sval[at + rank] = val[s];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < e - b; j += uint32_t(kThreads)) {
        const K k = skey[j];
        const uint32_t d = (k >> shift) & 0xFF;
        const size_t at = base + goff[d] + (j - lstart[d]);
        if (keys_out) keys_out[at] = k;
        vals_out[at] = sval[j];
    }
}

// ---------------------------------------------------------------------------------------------------
// std::sort partition phase
// ---------------------------------------------------------------------------------------------------
struct SortSeg { uint32_t first, last, depth; };      // indices into the batched id array
// Rounds run back to back without the host looking at anything: round r reads how many segments round r - 1 produced from
// next[r] (next[0] is unused: the first round's count comes with the launch) and appends to next[r + 1]. A segment starts with
// depth 2 lg n and loses one per round (stl_algo.h:1945-1957), so 2 lg n + 1 rounds always suffice; rounds that find no
// segment cost one small launch.
constexpr uint32_t kSortMaxRounds = 66;
struct SortCounters { uint32_t next[kSortMaxRounds + 2]; uint32_t error, n_small, nan; };
// Segments of at most this many ids leave the rounds: a block finishes each of them in LDS (k_sort_finish below)
constexpr uint32_t kSortSmallMax = 4096;

template <typename T>
struct SortCtx {
    uint32_t* ids;              // batch * n
    const T* keys;              // key(a, id) = keys[a * astride + id * istride]
    uint32_t n, astride, istride;
    SortSeg* segs; SortSeg* segs_next;                  // round r reads segs (r even) / segs_next (r odd) and appends to the other
    uint32_t* ltab; uint32_t* rtab;                     // batch * n scratch
    SortCounters* counters;
    uint32_t seg_cap;
    SortSeg* small_segs;                                // segments of 17 .. kSortSmallMax ids go here instead of the next round (nullptr: off)
    uint2* chunk_cnt;                                   // huge segments: (L, R) counts, then offsets, per kSortChunk ABSOLUTE positions
    struct HugeState* huge;                             // per segment of the round's list
    uint32_t skip_huge;                                 // k_sort_partition leaves segments of more than kSortHuge ids to the k_sort_huge_* kernels
};
struct HugeState { uint32_t nl, nr, k, pad; };
constexpr uint32_t kSortHuge = 32768, kSortChunk = 4096;

constexpr int kSortThreads = 512;

template <typename T>
__global__ void __launch_bounds__(kSortThreads) k_sort_partition(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    __shared__ uint32_t wsum_l[8], wsum_r[8];
    __shared__ uint32_t sh_k;
    const uint32_t n_active = round == 0 ? first_count : min(c.counters->next[round], c.seg_cap);
    const SortSeg* segs_in = (round & 1u) ? c.segs_next : c.segs;
    SortSeg* segs_out = (round & 1u) ? c.segs : c.segs_next;
  for (uint32_t seg_id = blockIdx.x; seg_id < n_active; seg_id += gridDim.x) {
    __syncthreads();                                     // (the previous segment's shared scalars are done with)
    const SortSeg sg = segs_in[seg_id];
    const uint32_t arr = sg.first / c.n;
    const T* kb = c.keys + size_t{arr} * c.astride;
    const uint32_t istride = c.istride;
    auto key = [=](uint32_t id) { return kb[size_t{id} * istride]; };
    uint32_t* ids = c.ids;
    const uint32_t first = sg.first, last = sg.last, len = last - first;
    if (sg.depth == 0) {                                 // __partial_sort(first, last, last): heap sort
        if (threadIdx.x == 0) partial_sort_replay(ids + first, long(len), long(len), key);
        continue;
    }
    if (c.skip_huge && len > kSortHuge) continue;        // partitioned by many blocks (k_sort_huge_*)
    if (threadIdx.x == 0) {                              // __move_median_to_first(first, first + 1, mid, last - 1)
        const uint32_t a = first + 1, b = first + len / 2, cc = last - 1;
        const T ka = key(ids[a]), kbv = key(ids[b]), kc = key(ids[cc]);
        uint32_t pick;
        if (ka < kbv) { if (kbv < kc) pick = b; else if (ka < kc) pick = cc; else pick = a; }
        else if (ka < kc) pick = a;
        else if (kbv < kc) pick = cc;
        else pick = b;
        const uint32_t t = ids[first]; ids[first] = ids[pick]; ids[pick] = t;
        sh_k = 0;
    }
    __syncthreads();
    const T pivot = key(ids[first]);
    if (round == 0 && threadIdx.x == 0 && pivot != pivot) c.counters->nan = 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: L (key >= pivot) and R (key <= pivot) position tables over [first + 1, last), both ascending. Every thread takes kItems
    // CONSECUTIVE positions per tile, so that a tile of 4096 positions costs one block scan (round 4: with one position per thread a
    // multi-million-element segment of the first rounds paid two barriers per 512 positions)
    constexpr int kItems = 8;
    uint32_t run_l = 0, run_r = 0;
    for (uint32_t tile = first + 1; tile < last; tile += kSortThreads * kItems) {
        uint32_t ml = 0, mr = 0;
        const uint32_t p0 = tile + threadIdx.x * kItems;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const uint32_t pos = p0 + i;
            if (pos < last) {
                const T kv = key(ids[pos]);
                if (!(kv < pivot)) ml |= 1u << i;
                if (!(pivot < kv)) mr |= 1u << i;
                if (round == 0 && kv != kv) c.counters->nan = 1u;     // (round 0 sees every key; pivots are checked by the host path's fallback too)
            }
        }
        const uint32_t cl = __popc(ml), cr = __popc(mr);
        uint32_t il = cl, ir = cr;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ol = __shfl_up(il, d), orr = __shfl_up(ir, d);
            if (lane >= d) { il += ol; ir += orr; }
        }
        if (lane == 63) { wsum_l[wave] = il; wsum_r[wave] = ir; }
        __syncthreads();
        uint32_t el = il - cl, er = ir - cr, tl = 0, tr = 0;
        for (int w = 0; w < kSortThreads / 64; ++w) { if (w < wave) { el += wsum_l[w]; er += wsum_r[w]; } tl += wsum_l[w]; tr += wsum_r[w]; }
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            if (ml & (1u << i)) c.ltab[first + run_l + el + __popc(ml & ((1u << i) - 1u))] = p0 + i;
            if (mr & (1u << i)) c.rtab[first + run_r + er + __popc(mr & ((1u << i) - 1u))] = p0 + i;
        }
        run_l += tl; run_r += tr;
        __syncthreads();
    }
    const uint32_t nl = run_l, nr = run_r;               // both >= 1 (median-of-3 sentinels)
    __threadfence_block();
    __syncthreads();
    // pass 2: k = #{j : L_j < R_j}, R_j (descending) = rtab[nr - 1 - j]
    const uint32_t lim = min(nl, nr);
    uint32_t mine = 0;
    for (uint32_t j = threadIdx.x; j < lim; j += kSortThreads)
        mine += c.ltab[first + j] < c.rtab[first + nr - 1 - j] ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane == 0 && mine) atomicAdd(&sh_k, mine);
    __syncthreads();
    const uint32_t k = sh_k;
    // pass 3: the swaps
    for (uint32_t j = threadIdx.x; j < k; j += kSortThreads) {
        const uint32_t p = c.ltab[first + j], q = c.rtab[first + nr - 1 - j];
        const uint32_t a = ids[p], b = ids[q];
        ids[p] = b; ids[q] = a;
    }
    if (threadIdx.x == 0) {
        uint32_t cut;
        if (k == 0) cut = c.ltab[first];
        else {
            const uint32_t rk1 = c.rtab[first + nr - k];                 // R_{k-1}
            cut = (k < nl && c.ltab[first + k] < rk1) ? c.ltab[first + k] : rk1;
        }
        const uint32_t cb[2] = { first, cut }, ce[2] = { cut, last };
        for (int s = 0; s < 2; ++s) {
            if (ce[s] - cb[s] > 16) {                                    // _S_threshold
                const bool small = c.small_segs && ce[s] - cb[s] <= kSortSmallMax;
                const uint32_t slot = atomicAdd(small ? &c.counters->n_small : &c.counters->next[round + 1], 1u);
                if (slot < c.seg_cap) (small ? c.small_segs : segs_out)[slot] = SortSeg{ cb[s], ce[s], sg.depth - 1 };
                else atomicOr(&c.counters->error, 1u);
            }
        }
    }
  }
}

// ---- one partition step of a HUGE segment by many blocks (round 4) ---------------------------------------------------------------------
// k_sort_partition gives a segment to one block: the first rounds of a long array are a few blocks walking hundreds of thousands of ids
// each (1M ids x 3 axes: 1.7 + 1.5 + 1.0 + 0.6 + 0.4 ms for rounds 0-4, half of a serial Medium build). The same step for segments
// of more than kSortHuge ids, cut into chunks of kSortChunk ABSOLUTE positions (so a chunk's slot in chunk_cnt needs no allocation):
// (two slots per window of positions: a window is shared by at most two huge segments, one ending and one starting in it)
//   median  (a lane per segment)  __move_median_to_first
//   count   (a block per chunk)   how many L (key >= pivot) and R (key <= pivot) positions the chunk holds
//   scan    (a block per segment) chunk counts -> chunk offsets, totals nl / nr
//   write   (a block per chunk)   the chunk's L / R positions into ltab / rtab behind its offset, ascending
//   k       (blocks over j)       k = #{j : L_j < R_j}
//   swap    (blocks over j)       the k swaps; one lane computes the cut and appends the two sides to the next lists
// — the same tables, the same k, the same swaps and cut as k_sort_partition's three passes. Blocks stride over chunks / segments, so
// the grid only has to be roughly right; segments that are not huge (or out of depth) stay with k_sort_partition in the same round.
template <typename T> struct SortView {
    const SortSeg* segs_in; SortSeg* segs_out; uint32_t n_active;
    __device__ SortView(const SortCtx<T>& c, uint32_t round, uint32_t first_count) {
        n_active = round == 0 ? first_count : min(c.counters->next[round], c.seg_cap);
        segs_in = (round & 1u) ? c.segs_next : c.segs;
        segs_out = (round & 1u) ? c.segs : c.segs_next;
    }
};
__device__ inline bool sort_is_huge(const SortSeg& sg) { return sg.depth != 0 && sg.last - sg.first > kSortHuge; }

template <typename T>
__global__ void __launch_bounds__(256) k_sort_huge_median(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    const SortView<T> v(c, round, first_count);
    const uint32_t seg_id = blockIdx.x * 256 + threadIdx.x;
    if (seg_id >= v.n_active) return;
    const SortSeg sg = v.segs_in[seg_id];
    if (!sort_is_huge(sg)) return;
    const T* kb = c.keys + size_t{sg.first / c.n} * c.astride;
    auto key = [&](uint32_t id) { return kb[size_t{id} * c.istride]; };
    uint32_t* ids = c.ids;
    const uint32_t first = sg.first, a = first + 1, b = first + (sg.last - first) / 2, cc = sg.last - 1;
    const T ka = key(ids[a]), kbv = key(ids[b]), kc = key(ids[cc]);
    uint32_t pick;
    if (ka < kbv) { if (kbv < kc) pick = b; else if (ka < kc) pick = cc; else pick = a; }
    else if (ka < kc) pick = a;
    else if (kbv < kc) pick = cc;
    else pick = b;
    const uint32_t t = ids[first]; ids[first] = ids[pick]; ids[pick] = t;
    if (round == 0 && key(ids[first]) != key(ids[first])) c.counters->nan = 1u;
}

// flags of one chunk: thread t owns 8 consecutive positions (kSortThreads x 8 = kSortChunk)
template <typename T, typename KeyFn>
__device__ inline void sort_chunk_flags(const uint32_t* ids, KeyFn key, T pivot, uint32_t lo, uint32_t hi, uint32_t chunk_base, uint32_t& ml, uint32_t& mr, bool& nan) {
    ml = 0; mr = 0;
    const uint32_t p0 = chunk_base + threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t pos = p0 + i;
        if (pos >= lo && pos < hi) {
            const T kv = key(ids[pos]);
            if (!(kv < pivot)) ml |= 1u << i;
            if (!(pivot < kv)) mr |= 1u << i;
            nan = nan || kv != kv;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kSortThreads) k_sort_huge_count(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    static_assert(kSortThreads * 8 == kSortChunk);
    __shared__ uint32_t wl[8], wr[8];
    const SortView<T> v(c, round, first_count);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t seg_id = blockIdx.y; seg_id < v.n_active; seg_id += gridDim.y) {
        const SortSeg sg = v.segs_in[seg_id];
        if (!sort_is_huge(sg)) continue;
        const T* kb = c.keys + size_t{sg.first / c.n} * c.astride;
        const uint32_t istride = c.istride;
        auto key = [=](uint32_t id) { return kb[size_t{id} * istride]; };
        const T pivot = key(c.ids[sg.first]);
        const uint32_t lo = sg.first + 1, hi = sg.last, g0 = lo / kSortChunk, g1 = (hi - 1) / kSortChunk;
        for (uint32_t g = g0 + blockIdx.x; g <= g1; g += gridDim.x) {
            uint32_t ml, mr; bool nan = false;
            sort_chunk_flags<T>(c.ids, key, pivot, lo, hi, g * kSortChunk, ml, mr, nan);
            if (round == 0 && nan) c.counters->nan = 1u;
            uint32_t cl = __popc(ml), cr = __popc(mr);
            for (int off = 32; off > 0; off >>= 1) { cl += __shfl_down(cl, off); cr += __shfl_down(cr, off); }
            __syncthreads();
            if (lane == 0) { wl[wave] = cl; wr[wave] = cr; }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tl = 0, tr = 0;
                for (int w = 0; w < kSortThreads / 64; ++w) { tl += wl[w]; tr += wr[w]; }
                c.chunk_cnt[2 * g + (g != g0)] = make_uint2(tl, tr);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_sort_huge_scan(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    __shared__ uint32_t wl[4], wr[4];
    __shared__ uint32_t run_l, run_r;
    const SortView<T> v(c, round, first_count);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t seg_id = blockIdx.x; seg_id < v.n_active; seg_id += gridDim.x) {
        const SortSeg sg = v.segs_in[seg_id];
        if (!sort_is_huge(sg)) continue;
        const uint32_t g0 = (sg.first + 1) / kSortChunk, g1 = (sg.last - 1) / kSortChunk;
        __syncthreads();
        if (threadIdx.x == 0) { run_l = 0; run_r = 0; }
        __syncthreads();
        for (uint32_t base = g0; base <= g1; base += 256) {
            const uint32_t g = base + threadIdx.x;
            const uint2 cnt = g <= g1 ? c.chunk_cnt[2 * g + (g != g0)] : make_uint2(0, 0);
            uint32_t il = cnt.x, ir = cnt.y;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t ol = __shfl_up(il, d), orr = __shfl_up(ir, d);
                if (lane >= d) { il += ol; ir += orr; }
            }
            if (lane == 63) { wl[wave] = il; wr[wave] = ir; }
            __syncthreads();
            uint32_t el = run_l + il - cnt.x, er = run_r + ir - cnt.y, tl = 0, tr = 0;
            for (int w = 0; w < 4; ++w) { if (w < wave) { el += wl[w]; er += wr[w]; } tl += wl[w]; tr += wr[w]; }
            if (g <= g1) c.chunk_cnt[2 * g + (g != g0)] = make_uint2(el, er);
            __syncthreads();
            if (threadIdx.x == 0) { run_l += tl; run_r += tr; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { HugeState hs; hs.nl = run_l; hs.nr = run_r; hs.k = 0; hs.pad = 0; c.huge[seg_id] = hs; }
    }
}

template <typename T>
__global__ void __launch_bounds__(kSortThreads) k_sort_huge_write(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    __shared__ uint32_t wl[8], wr[8];
    const SortView<T> v(c, round, first_count);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t seg_id = blockIdx.y; seg_id < v.n_active; seg_id += gridDim.y) {
        const SortSeg sg = v.segs_in[seg_id];
        if (!sort_is_huge(sg)) continue;
        const T* kb = c.keys + size_t{sg.first / c.n} * c.astride;
        const uint32_t istride = c.istride;
        auto key = [=](uint32_t id) { return kb[size_t{id} * istride]; };
        const T pivot = key(c.ids[sg.first]);
        const uint32_t first = sg.first, lo = first + 1, hi = sg.last, g0 = lo / kSortChunk, g1 = (hi - 1) / kSortChunk;
        for (uint32_t g = g0 + blockIdx.x; g <= g1; g += gridDim.x) {
            uint32_t ml, mr; bool nan = false;
            sort_chunk_flags<T>(c.ids, key, pivot, lo, hi, g * kSortChunk, ml, mr, nan);
            const uint32_t cl = __popc(ml), cr = __popc(mr);
            uint32_t il = cl, ir = cr;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t ol = __shfl_up(il, d), orr = __shfl_up(ir, d);
                if (lane >= d) { il += ol; ir += orr; }
            }
            __syncthreads();
            if (lane == 63) { wl[wave] = il; wr[wave] = ir; }
            __syncthreads();
            const uint2 off = c.chunk_cnt[2 * g + (g != g0)];
            uint32_t el = off.x + il - cl, er = off.y + ir - cr;
            for (int w = 0; w < wave; ++w) { el += wl[w]; er += wr[w]; }
            const uint32_t p0 = g * kSortChunk + threadIdx.x * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (ml & (1u << i)) c.ltab[first + el + __popc(ml & ((1u << i) - 1u))] = p0 + i;
                if (mr & (1u << i)) c.rtab[first + er + __popc(mr & ((1u << i) - 1u))] = p0 + i;
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_sort_huge_k(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    const SortView<T> v(c, round, first_count);
    for (uint32_t seg_id = blockIdx.y; seg_id < v.n_active; seg_id += gridDim.y) {
        const SortSeg sg = v.segs_in[seg_id];
        if (!sort_is_huge(sg)) continue;
        const HugeState hs = c.huge[seg_id];
        const uint32_t lim = min(hs.nl, hs.nr), first = sg.first;
        uint32_t mine = 0;
        for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < lim; j += gridDim.x * 256)
            mine += c.ltab[first + j] < c.rtab[first + hs.nr - 1 - j] ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&c.huge[seg_id].k, mine);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_sort_huge_swap(SortCtx<T> c, uint32_t round, uint32_t first_count) {
    const SortView<T> v(c, round, first_count);
    for (uint32_t seg_id = blockIdx.y; seg_id < v.n_active; seg_id += gridDim.y) {
        const SortSeg sg = v.segs_in[seg_id];
        if (!sort_is_huge(sg)) continue;
        const HugeState hs = c.huge[seg_id];
        const uint32_t first = sg.first, last = sg.last, k = hs.k, nl = hs.nl, nr = hs.nr;
        for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < k; j += gridDim.x * 256) {
            const uint32_t p = c.ltab[first + j], q = c.rtab[first + nr - 1 - j];
            const uint32_t a = c.ids[p], b = c.ids[q];
            c.ids[p] = b; c.ids[q] = a;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            uint32_t cut;
            if (k == 0) cut = c.ltab[first];
            else {
                const uint32_t rk1 = c.rtab[first + nr - k];                 // R_{k-1}
                cut = (k < nl && c.ltab[first + k] < rk1) ? c.ltab[first + k] : rk1;
            }
            const uint32_t cb[2] = { first, cut }, ce[2] = { cut, last };
            for (int h = 0; h < 2; ++h) {
                if (ce[h] - cb[h] > 16) {                                    // _S_threshold
                    const bool small = c.small_segs && ce[h] - cb[h] <= kSortSmallMax;
                    const uint32_t slot = atomicAdd(small ? &c.counters->n_small : &c.counters->next[round + 1], 1u);
                    if (slot < c.seg_cap) (small ? c.small_segs : v.segs_out)[slot] = SortSeg{ cb[h], ce[h], sg.depth - 1 };
                    else atomicOr(&c.counters->error, 1u);
                }
            }
        }
    }
}

// std::sort of what fits in LDS (round 4): ids and keys of at most kSortSmallMax positions resident in one block.
//   * k_std_sort_small: the WHOLE of std::sort for arrays of at most kSortSmallMax ids in ONE launch, one block per array. The mini-tree
//     builder's top level without pruning sorts the roots of a 16^3 grid — three arrays of <= 4096 ids — and paid 25 partition rounds
//     + a 4-pass radix sort + a host round trip for them: ~40 launches, 0.34 ms of a 2.1 ms build.
//   * k_sort_finish: longer arrays run k_sort_partition's rounds only until a segment is down to kSortSmallMax ids; one block then
//     takes such a segment through ALL its remaining rounds (its depth budget travels with it). The rounds past that point were most of
//     them — 2 lg n + 1 launches, the late ones grid-striding blocks of 512 threads over tens of thousands of 20..60-id segments.
// Both: __introsort_loop (stl_algo.h:1945-1957) with a WAVE per segment (ballot prefix sums instead of block scans, no block barrier
// inside a segment), a block barrier between rounds, segment lists in LDS (disjoint segments of more than 16 ids: at most 240).
// __final_insertion_sort == the stable sort by key of the arrangement the rounds leave. Every id is then within 15 positions of its
// place (what is left unpartitioned is at most 16 long, and everything further left / right compares <= / >=), so an id's rank is
// its window's start + the ids of the window that sort before it: k_std_sort_small ranks in LDS, k_sort_rank over the whole array
// (instead of four radix passes). Keys that are NaN void that argument (std::sort itself is undefined for them): the small kernel
// then ranks every id against all others, the long path falls back to the radix sort — a permutation either way.
constexpr int kSortSmallThreads = 1024;

template <typename T>
struct SortLds {
    T skey[kSortSmallMax];                               // by local id
    uint32_t sid[kSortSmallMax];                         // the positions being sorted: local ids
    uint16_t ltab[kSortSmallMax], rtab[kSortSmallMax];
    uint32_t seg[2][256];                                // first | last << 12 | depth << 25
    uint32_t seg_count[2];
    uint32_t any_nan;
};

// skey / sid are loaded, seg[0][0] / seg_count[0] describe the one starting segment; all threads of the block call this
template <typename T>
__device__ void lds_introsort_rounds(SortLds<T>& L) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto key = [&](uint32_t id) { return L.skey[id]; };
    uint32_t* sid = L.sid;
    for (uint32_t round = 0;; ++round) {
        const uint32_t cur = round & 1u, n_active = L.seg_count[cur];
        if (n_active == 0) break;
        for (uint32_t s = wave; s < n_active; s += kSortSmallThreads / 64) {
            const uint32_t pk = L.seg[cur][s];
            const uint32_t first = pk & 0xFFFu, last = (pk >> 12) & 0x1FFFu, depth = pk >> 25, len = last - first;
            if (depth == 0) {                             // __partial_sort(first, last, last): heap sort
                if (lane == 0) partial_sort_replay(sid + first, long(len), long(len), key);
                continue;
            }
            if (lane == 0) {                              // __move_median_to_first(first, first + 1, mid, last - 1)
                const uint32_t a = first + 1, b = first + len / 2, cc = last - 1;
                const T ka = key(sid[a]), kbv = key(sid[b]), kc = key(sid[cc]);
                uint32_t pick;
                if (ka < kbv) { if (kbv < kc) pick = b; else if (ka < kc) pick = cc; else pick = a; }
                else if (ka < kc) pick = a;
                else if (kbv < kc) pick = cc;
                else pick = b;
                const uint32_t t = sid[first]; sid[first] = sid[pick]; sid[pick] = t;
            }
            wave_sync();
            const T pivot = key(sid[first]);
            // L (key >= pivot) and R (key <= pivot) position tables over [first + 1, last), both ascending (see k_sort_partition)
            uint32_t nl = 0, nr = 0;
            const uint64_t below = (uint64_t{1} << lane) - 1u;
            for (uint32_t tile = first + 1; tile < last; tile += 64) {
                const uint32_t pos = tile + lane;
                bool fl = false, fr = false;
                if (pos < last) { const T kv = key(sid[pos]); fl = !(kv < pivot); fr = !(pivot < kv); }
                const uint64_t bl = __ballot(fl), br = __ballot(fr);
                if (fl) L.ltab[first + nl + __popcll(bl & below)] = static_cast<uint16_t>(pos);
                if (fr) L.rtab[first + nr + __popcll(br & below)] = static_cast<uint16_t>(pos);
                nl += __popcll(bl); nr += __popcll(br);
            }
            wave_sync();
            const uint32_t lim = min(nl, nr);             // k = #{j : L_j < R_j}, R_j (descending) = rtab[nr - 1 - j]
            uint32_t k = 0;
            for (uint32_t j0 = 0; j0 < lim; j0 += 64) {
                const uint32_t j = j0 + lane;
                k += __popcll(__ballot(j < lim && L.ltab[first + j] < L.rtab[first + nr - 1 - j]));
            }
            for (uint32_t j0 = 0; j0 < k; j0 += 64) {    // the swaps
                const uint32_t j = j0 + lane;
                if (j < k) {
                    const uint32_t p = L.ltab[first + j], q = L.rtab[first + nr - 1 - j];
                    const uint32_t a = sid[p], b = sid[q];
                    sid[p] = b; sid[q] = a;
                }
            }
            if (lane == 0) {
                uint32_t cut;
                if (k == 0) cut = L.ltab[first];
                else {
                    const uint32_t rk1 = L.rtab[first + nr - k];             // R_{k-1}
                    cut = (k < nl && L.ltab[first + k] < rk1) ? L.ltab[first + k] : rk1;
                }
                const uint32_t cb[2] = { first, cut }, ce[2] = { cut, last };
                for (int h = 0; h < 2; ++h)
                    if (ce[h] - cb[h] > 16) L.seg[cur ^ 1u][atomicAdd(&L.seg_count[cur ^ 1u], 1u)] = cb[h] | (ce[h] << 12) | ((depth - 1) << 25);
            }
        }
        __syncthreads();
        if (tid == 0) L.seg_count[cur] = 0;
        __syncthreads();
    }
}

template <typename T>
__global__ void __launch_bounds__(kSortSmallThreads) k_std_sort_small(uint32_t* d_ids, const T* d_keys, uint32_t n, uint32_t astride, uint32_t istride,
                                                                      uint32_t depth0) {
    using U = typename Ord<T>::U;
    __shared__ SortLds<T> L;
    const uint32_t tid = threadIdx.x;
    const T* kb = d_keys + size_t{blockIdx.x} * astride;
    if (tid == 0) { L.seg[0][0] = 0u | (n << 12) | (depth0 << 25); L.seg_count[0] = n > 16 ? 1u : 0u; L.seg_count[1] = 0; L.any_nan = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kSortSmallThreads) {
        const T v = kb[size_t{i} * istride];
        L.skey[i] = v; L.sid[i] = i;                      // local id = id
        if (v != v) L.any_nan = 1;
    }
    __syncthreads();
    lds_introsort_rounds(L);
    // __final_insertion_sort
    auto image = [&](uint32_t pos) { T v = L.skey[L.sid[pos]]; if (v == T(0)) v = T(0); return Ord<T>::enc(v); };    // -0 and +0 are EQUAL keys for operator<
    const bool all = L.any_nan != 0;
    uint32_t* out = d_ids + size_t{blockIdx.x} * n;
    for (uint32_t p = tid; p < n; p += kSortSmallThreads) {
        const U kp = image(p);
        const uint32_t lo = all ? 0u : (p >= 15u ? p - 15u : 0u), hi = all ? n : min(n, p + 16u);
        uint32_t rank = lo;
        for (uint32_t j = lo; j < hi; ++j) {
            const U kj = image(j);
            rank += (kj < kp || (kj == kp && j < p)) ? 1u : 0u;
        }
        out[rank] = L.sid[p];
    }
}

// one block per segment that left k_sort_partition's rounds with at most kSortSmallMax ids: its remaining rounds, in place
template <typename T>
__global__ void __launch_bounds__(kSortSmallThreads) k_sort_finish(SortCtx<T> c) {
    __shared__ SortLds<T> L;
    __shared__ uint32_t orig[kSortSmallMax];              // local id -> id
    const uint32_t tid = threadIdx.x;
    const SortSeg sg = c.small_segs[blockIdx.x];
    const uint32_t len = sg.last - sg.first;
    const T* kb = c.keys + size_t{sg.first / c.n} * c.astride;
    if (tid == 0) { L.seg[0][0] = 0u | (len << 12) | (sg.depth << 25); L.seg_count[0] = 1u; L.seg_count[1] = 0; }
    for (uint32_t p = tid; p < len; p += kSortSmallThreads) {
        const uint32_t id = c.ids[sg.first + p];
        orig[p] = id; L.sid[p] = p;
        L.skey[p] = kb[size_t{id} * c.istride];
    }
    __syncthreads();
    lds_introsort_rounds(L);
    for (uint32_t p = tid; p < len; p += kSortSmallThreads) c.ids[sg.first + p] = orig[L.sid[p]];
}

// __final_insertion_sort over whole arrays: rank of every position within its window of 31 (see above), 256 positions per block
template <typename T>
__global__ void __launch_bounds__(256) k_sort_rank(const uint32_t* ids, const T* keys, uint32_t n, uint32_t astride, uint32_t istride, uint32_t* out) {
    using U = typename Ord<T>::U;
    __shared__ U img[256 + 30];
    const uint32_t arr = blockIdx.y, base = blockIdx.x * 256;
    const uint32_t* a = ids + size_t{arr} * n;
    const T* kb = keys + size_t{arr} * astride;
    for (uint32_t q = threadIdx.x; q < 256 + 30; q += 256) {
        const long pos = long(base) - 15 + long(q);
        if (pos >= 0 && pos < long(n)) {
            T v = kb[size_t{a[pos]} * istride];
            if (v == T(0)) v = T(0);                      // -0 and +0 are EQUAL keys for operator<
            img[q] = Ord<T>::enc(v);
        }
    }
    __syncthreads();
    const uint32_t p = base + threadIdx.x;
    if (p >= n) return;
    const uint32_t lo = p >= 15u ? p - 15u : 0u, hi = min(n, p + 16u);
    const U kp = img[threadIdx.x + 15];
    uint32_t rank = lo;
    for (uint32_t j = lo; j < hi; ++j) {
        const U kj = img[j - base + 15];
        rank += (kj < kp || (kj == kp && j < p)) ? 1u : 0u;
    }
    out[size_t{arr} * n + rank] = a[p];
}

template <typename T>
__global__ void __launch_bounds__(256) k_make_sort_keys(const uint32_t* ids, const T* keys, uint32_t n, uint32_t total, uint32_t astride,
                                                        uint32_t istride, typename Ord<T>::U* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t arr = i / n;
    T v = keys[size_t{arr} * astride + size_t{ids[i]} * istride];
    if (v == T(0)) v = T(0);                              // -0 and +0 are EQUAL keys for operator<: one integer image
    out[i] = Ord<T>::enc(v);
}

__global__ void __launch_bounds__(256) k_iota(uint32_t* ids, uint32_t n, uint32_t total) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) ids[i] = i % n;
}

// exclusive scan of a u32 array (two levels are enough for 2^28 elements with 4096-element blocks... three here)
constexpr int kScanBlock = 4096;
__global__ void __launch_bounds__(1024) k_scan_blocks(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* block_sums) {
    __shared__ uint32_t part[1024];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t o = threadIdx.x >= unsigned(off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 1023 && block_sums) block_sums[blockIdx.x] = part[1023];
}
__global__ void __launch_bounds__(256) k_scan_add(uint32_t* out, uint32_t n, const uint32_t* block_offsets) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += block_offsets[i / kScanBlock];
}

} // namespace

// out[i] = sum of in[0..i) (in == out allowed), entirely in stream order: blocks of 4096, their sums scanned recursively, offsets
// added back. d_total (device, optional) receives the grand total.
static int scan_u32_async(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* d_total, hipStream_t stream) {
    const uint32_t blocks = (n + kScanBlock - 1) / kScanBlock;
    DevBuf<uint32_t> sums, sums_scanned;
    BVH_HIP_TRY(sums.alloc(blocks), BVH_AMD_ERR_HIP);
    hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(1024), 0, stream, in, out, n, sums.p);
    if (blocks > 1) {
        BVH_HIP_TRY(sums_scanned.alloc(blocks), BVH_AMD_ERR_HIP);
        int rc = scan_u32_async(sums.p, sums_scanned.p, blocks, d_total, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, sums_scanned.p);
    } else if (d_total) {
        BVH_HIP_TRY(hipMemcpyAsync(d_total, sums.p, 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
    }
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    if (!sums.pooled) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);     // plain hipFree of the scratch on return
    return BVH_AMD_OK;
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_host, hipStream_t stream) {
    StreamScope scratch_on(stream);
    // out[i] = sum of in[0..i); total optionally returned to the host (then, and only then, the stream is synchronised)
    if (n == 0) { if (total_host) *total_host = 0; return BVH_AMD_OK; }
    DevBuf<uint32_t> d_total;
    if (total_host) BVH_HIP_TRY(d_total.alloc(1), BVH_AMD_ERR_HIP);
    int rc = scan_u32_async(in, out, n, total_host ? d_total.p : nullptr, stream);
    if (rc) return rc;
    if (total_host) {
        { int rb_ = readback(total_host, d_total.p, 4, stream); if (rb_) return rb_; }
    }
    return BVH_AMD_OK;
}


// Stable sort of `batch` independent arrays of n (key, value) pairs by the low `bits` bits of the key.
// keys/vals are overwritten with the result; tmp buffers have the same sizes.
size_t radix_sort_hist_words(uint32_t n, uint32_t batch) { return size_t{batch} * 256 * ((n + kRadixMinTile - 1) / kRadixMinTile); }

template <typename K>
int radix_sort_pairs(K* keys, uint32_t* vals, K* keys_tmp, uint32_t* vals_tmp, uint32_t n, uint32_t batch, int bits, hipStream_t stream,
                     uint32_t* hist_buf, bool iota_vals, bool keys_wanted, uint32_t** vals_result, bool first_hist_done) {
    static_assert(RadixShape<uint32_t>::kTile == kRadixTileU32, "ray_keys_kernel writes the first histogram in tiles of kRadixTileU32 keys");
    if (first_hist_done && (sizeof(K) != 4 || batch != 1 || !hist_buf)) return fail(BVH_AMD_ERR_ARG, "radix_sort_pairs: first_hist_done needs 32-bit keys, one array and the caller's histogram buffer");
    if (vals_result) *vals_result = vals;
    if (n == 0 || batch == 0) return BVH_AMD_OK;
    StreamScope scratch_on(stream);
    constexpr int kTile = RadixShape<K>::kTile, kThreads = RadixShape<K>::kThreads;
    const uint32_t bpa = (n + kTile - 1) / kTile;
    DevBuf<uint32_t> own;
    if (!hist_buf) BVH_HIP_TRY(own.alloc(size_t{batch} * 256 * bpa), BVH_AMD_ERR_HIP);
    uint32_t* const hist = hist_buf ? hist_buf : own.p;       // caller-owned scratch: fully asynchronous
    K* kin = keys; K* kout = keys_tmp; uint32_t* vin = vals; uint32_t* vout = vals_tmp;
    int passes = (bits + 7) / 8;
    if ((passes & 1) && !vals_result) ++passes;           // even number of passes: the result lands in keys/vals
    // (a caller that passes vals_result takes the values wherever the last pass leaves them and gives up the keys)
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (!(p == 0 && first_hist_done)) hipLaunchKernelGGL(k_radix_hist<K>, dim3(batch * bpa), dim3(kThreads), 0, stream, kin, n, bpa, shift, hist);
        if (256 * bpa > 16384) {                              // long histograms: the multi-block scan (one block per array crawls: 1 ms at 10M keys)
            for (uint32_t a = 0; a < batch; ++a) {
                int rc = scan_u32_async(hist + size_t{a} * 256 * bpa, hist + size_t{a} * 256 * bpa, 256 * bpa, nullptr, stream);
                if (rc) return rc;
            }
        } else hipLaunchKernelGGL(k_radix_scan, dim3(batch), dim3(1024), 0, stream, hist, 256 * bpa);
        hipLaunchKernelGGL(k_radix_scatter<K>, dim3(batch * bpa), dim3(kThreads), 0, stream, kin, p == 0 && iota_vals ? nullptr : vin,
                           p == passes - 1 && !keys_wanted ? nullptr : kout, vout, n, bpa, shift, hist);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    if (vals_result) *vals_result = vin;                  // (after the last swap: where the final pass wrote)
    if (!hist_buf && !own.pooled) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // hist is freed on return (the pool frees in stream order)
    return BVH_AMD_OK;
}

// d_ids: batch * n, overwritten with iota then sorted like std::sort with comp(i, j) = key(a,i) < key(a,j).
template <typename T>
int std_sort_ids(uint32_t* d_ids, const T* d_keys, uint32_t n, uint32_t batch, uint32_t astride, uint32_t istride, hipStream_t stream) {
    StreamScope scratch_on(stream);
    if (n == 0 || batch == 0) return BVH_AMD_OK;
    static const bool small_off = BVH_DEV_INT("BVH_AMD_SORT_SMALL", 1) == 0;      // A/B runs
    if (n <= kSortSmallMax && !small_off) {
        uint32_t lg = 0;
        while ((uint64_t{2} << lg) <= n) ++lg;            // std::__lg(n) = floor(log2 n)
        hipLaunchKernelGGL(k_std_sort_small<T>, dim3(batch), dim3(kSortSmallThreads), 0, stream, d_ids, d_keys, n, astride, istride, 2 * lg);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        return BVH_AMD_OK;
    }
    const uint32_t total = n * batch;
    hipLaunchKernelGGL(k_iota, dim3((total + 255) / 256), dim3(256), 0, stream, d_ids, n, total);
    using U = typename Ord<T>::U;
    static const bool finish_off = BVH_DEV_INT("BVH_AMD_SORT_FINISH", 1) == 0;   // A/B runs: every round global + radix sort
    DevBuf<uint32_t> ltab, rtab, vals_tmp;
    DevBuf<SortSeg> seg_a, seg_b, seg_small;
    DevBuf<SortCounters> counters;
    DevBuf<uint2> chunk_cnt;
    DevBuf<HugeState> huge;
    const uint32_t seg_cap = total / 16 + batch + 2;
    static const bool huge_off = BVH_DEV_INT("BVH_AMD_SORT_HUGE", 1) == 0;         // A/B runs: one block per segment always
    uint32_t huge_rounds = 0;                             // rounds that get the many-block step: until halving would have ended it, + 3
    if (!huge_off && n > kSortHuge) { huge_rounds = 4; while ((uint64_t{kSortHuge} << (huge_rounds - 4)) < n) ++huge_rounds; }
    const uint32_t huge_cap = static_cast<uint32_t>(std::min<uint64_t>(seg_cap, uint64_t{batch} << std::min(huge_rounds, 24u))) + 1;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(ltab.alloc(total)); A(rtab.alloc(total)); A(vals_tmp.alloc(total)); A(seg_a.alloc(seg_cap)); A(seg_b.alloc(seg_cap));
    if (!finish_off) A(seg_small.alloc(seg_cap));
    if (huge_rounds) { A(chunk_cnt.alloc(2 * (size_t{total} / kSortChunk + 2))); A(huge.alloc(huge_cap)); }
    A(counters.alloc(1));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("std_sort_ids: hipMalloc: ") + hipGetErrorString(e));

    bool ranked_by_window = false;
    if (n > 16) {
        uint32_t lg = 0;
        while ((uint64_t{2} << lg) <= n) ++lg;            // std::__lg(n) = floor(log2 n)
        std::vector<SortSeg> roots(batch);
        for (uint32_t a = 0; a < batch; ++a) roots[a] = SortSeg{ a * n, a * n + n, 2 * lg };
        BVH_HIP_TRY(hipMemcpyAsync(seg_a.p, roots.data(), batch * sizeof(SortSeg), hipMemcpyHostToDevice, stream), BVH_AMD_ERR_HIP);
        SortCtx<T> c;
        c.ids = d_ids; c.keys = d_keys; c.n = n; c.astride = astride; c.istride = istride;
        c.segs = seg_a.p; c.segs_next = seg_b.p; c.ltab = ltab.p; c.rtab = rtab.p; c.counters = counters.p; c.seg_cap = seg_cap;
        c.small_segs = finish_off ? nullptr : seg_small.p;
        c.chunk_cnt = chunk_cnt.p; c.huge = huge.p; c.skip_huge = 0;
        BVH_HIP_TRY(hipMemsetAsync(counters.p, 0, sizeof(SortCounters), stream), BVH_AMD_ERR_HIP);
        const uint32_t rounds = 2 * lg + 1;                   // <= kSortMaxRounds for any 32-bit n
        // one block per segment; disjoint segments of more than 16 ids each bound their number, a grid-stride loop covers the rest
        const uint32_t grid = std::max<uint32_t>(batch, std::min<uint32_t>(total / 17 + 1, 2048u));
        uint32_t r = 0;
        auto launch_rounds = [&](uint32_t upto) {
            for (; r < upto; ++r) {
                c.skip_huge = r < huge_rounds ? 1u : 0u;
                if (c.skip_huge) {
                    // at most batch << r segments in round r's list; chunks per segment as if the pivots had quartered instead of halved
                    const uint32_t segs = static_cast<uint32_t>(std::min<uint64_t>(uint64_t{batch} << r, huge_cap));
                    const uint32_t per_seg = std::max<uint32_t>(1u, (n / kSortChunk + 1) >> (r > 2 ? r - 2 : 0));
                    const dim3 by_chunk(per_seg, std::min<uint32_t>(segs, 4096u));
                    hipLaunchKernelGGL(k_sort_huge_median<T>, dim3((segs + 255) / 256), dim3(256), 0, stream, c, r, batch);
                    hipLaunchKernelGGL(k_sort_huge_count<T>, by_chunk, dim3(kSortThreads), 0, stream, c, r, batch);
                    hipLaunchKernelGGL(k_sort_huge_scan<T>, dim3(std::min<uint32_t>(segs, 4096u)), dim3(256), 0, stream, c, r, batch);
                    hipLaunchKernelGGL(k_sort_huge_write<T>, by_chunk, dim3(kSortThreads), 0, stream, c, r, batch);
                    hipLaunchKernelGGL(k_sort_huge_k<T>, by_chunk, dim3(256), 0, stream, c, r, batch);
                    hipLaunchKernelGGL(k_sort_huge_swap<T>, by_chunk, dim3(256), 0, stream, c, r, batch);
                }
                hipLaunchKernelGGL(k_sort_partition<T>, dim3(r == 0 ? batch : grid), dim3(kSortThreads), 0, stream, c, r, batch);
            }
        };
        if (finish_off) launch_rounds(rounds);
        else {
            // Rounds go on only while a segment longer than kSortSmallMax exists: lg(n / kSortSmallMax) rounds if every pivot halved its
            // segment; medians of three do nearly that, so a few more are launched before the host looks at the list's length, then four
            // at a time (the bound stays 2 lg n + 1: past its depth budget a segment is heap-sorted in place).
            uint32_t expect = 3;
            while ((uint64_t{kSortSmallMax} << (expect - 3)) < n) ++expect;
            uint32_t big = 1;
            while (big && r < rounds) {
                launch_rounds(std::min(rounds, r + (r == 0 ? expect : 4u)));
                BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
                if (r < rounds) { int rb_ = readback(&big, &counters.p->next[r], sizeof(big), stream); if (rb_) return rb_; }
            }
        }
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        uint32_t tail[3] = {0, 0, 0};                         // error, n_small, nan
        { int rb_ = readback(tail, &counters.p->error, sizeof(tail), stream); if (rb_) return rb_; }
        if (tail[0]) return fail(BVH_AMD_ERR_OVERFLOW, "std_sort_ids: segment capacity exceeded");
        if (tail[1]) hipLaunchKernelGGL(k_sort_finish<T>, dim3(tail[1]), dim3(kSortSmallThreads), 0, stream, c);
        ranked_by_window = !finish_off && tail[2] == 0;
    }
    if (ranked_by_window) {
        // __final_insertion_sort by window ranks (see k_sort_rank); into the scratch copy, then back
        hipLaunchKernelGGL(k_sort_rank<T>, dim3((n + 255) / 256, batch), dim3(256), 0, stream, d_ids, d_keys, n, astride, istride, vals_tmp.p);
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipMemcpyAsync(d_ids, vals_tmp.p, size_t{total} * 4, hipMemcpyDeviceToDevice, stream), BVH_AMD_ERR_HIP);
        if (!scratch_pool_enabled()) BVH_HIP_TRY(hipStreamSynchronize(stream), BVH_AMD_ERR_HIP);   // plain hipFree of the workspace on return
        return BVH_AMD_OK;
    }
    // __final_insertion_sort == stable sort by key of the current arrangement
    DevBuf<U> skeys, skeys_tmp;
    A(skeys.alloc(total)); A(skeys_tmp.alloc(total));
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("std_sort_ids: hipMalloc: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(k_make_sort_keys<T>, dim3((total + 255) / 256), dim3(256), 0, stream, d_ids, d_keys, n, total, astride, istride, skeys.p);
    return radix_sort_pairs<U>(skeys.p, d_ids, skeys_tmp.p, vals_tmp.p, n, batch, int(sizeof(U) * 8), stream);
}

template int radix_sort_pairs<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, int, hipStream_t, uint32_t*, bool, bool, uint32_t**, bool);
template int radix_sort_pairs<uint16_t>(uint16_t*, uint32_t*, uint16_t*, uint32_t*, uint32_t, uint32_t, int, hipStream_t, uint32_t*, bool, bool, uint32_t**, bool);
template int radix_sort_pairs<unsigned long long>(unsigned long long*, uint32_t*, unsigned long long*, uint32_t*, uint32_t, uint32_t, int, hipStream_t, uint32_t*, bool, bool, uint32_t**, bool);
template int std_sort_ids<float>(uint32_t*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
template int std_sort_ids<double>(uint32_t*, const double*, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);

} // namespace bvh_amd
