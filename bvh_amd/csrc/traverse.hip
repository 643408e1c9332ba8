// K11 — batched single-ray traversal for gfx950.
//
// Reproduces, per ray, Bvh<Node>::intersect<IsAnyHit, IsRobust> (reference bvh.h:160-182) driving
// traverse_top_down (bvh.h:125-157) with the leaf lambda of test/benchmark.cpp:281-291 over permuted
// primitives: identical visit order, identical arithmetic (one rounding per operation, fma only where the
// reference says fast_mul_add: node.h:85-86), hence identical hits.
//
// MI355X mapping:
//   * one ray per lane, persistent wavefronts: a wave keeps 64 ray slots busy and refills finished slots
//     from a global ticket counter with one wave-aggregated atomic (ballot + popcount), so incoherent rays of
//     very different trip counts do not idle the SIMD until the slowest lane finishes;
//   * while-while loop: all lanes descend inner nodes until every active lane holds a leaf, then leaves;
//   * the traversal stack (bvh.h:128-131, SmallStack<Index,64>) lives in LDS, laid out [depth][lane] so a
//     push/pop is conflict-free (lane -> bank), with the rarely used entries 24..63 in per-lane scratch;
//   * both children of a node come from ONE 64-byte (128-byte for double) aligned record (common.h).
//
// Compiled with -ffp-contract=off; division and sqrt are the correctly rounded forms.

#include "common.h"
#include "trace_device.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

namespace bvh_amd {

namespace {

thread_local const char* g_last_kernel = "";
thread_local bool g_last_reordered = false;
thread_local int g_last_plan[4] = {0, 0, 0, 0};               // reordered, coop, refill, leaf of the calling thread's latest launch

// Optional timing of the traversal kernel alone (bench.py's roofline: the coherence sort in front of it is not the kernel):
// the calling thread's latest launches, a pair of events each.
struct KernelTimer {
    static constexpr size_t kRing = 256;
    bool on = false;
    size_t count = 0;
    std::pair<hipEvent_t, hipEvent_t> ring[kRing] = {};
    hipEvent_t begin[kRing] = {};                             // start of the call on its stream: ring[i].first - begin[i] = ray keys + radix sort
    bool begin_armed = false;                                 // begin[count % kRing] was recorded by the call in progress
};
KernelTimer& kernel_timer() { static thread_local KernelTimer t; return t; }

// NOTE (measured, profiles/r02_traversal_experiments.md): on the 1M-triangle soup this kernel's L2 misses (55 G 64-byte sectors/s)
// run at 0.96 of the rate at which the memory system serves a dependent random walk over 64-byte records (57 G/s, bench.py's
// record-walk probe): what binds it is misses per ray, not instructions, L1 requests or occupancy. Tried and rejected on hardware:
// 32-byte "compact" records next to the PairNodes (-37 % L1 requests, 7 % slower), v_max/v_min slab test, ds_read pop, buffer
// loads, a quad-cooperative fetch through LDS (3-4x slower), an LDS cache of the top of the tree (27 % slower), ray sorting.
// D = 2: Bvh<Node<T, 2>> with circles (Sphere<T, 2>, stride 3) and 6-value rays. The pair records stay three wide (their z
// bounds are zero and never looked at): only the per-ray constants, the slab test and the leaf test run over D axes.
// Deep = true (trees of more than 64 levels only): stack entries beyond the 64 of SmallStack spill to HBM (GrowingStack).
// (70 VGPRs = 7 waves per SIMD; forcing 8 with amdgpu_waves_per_eu fits in 63 without spills and runs 16 % slower on soup_1m)
template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D = 3, bool Deep = false>
__global__ void __launch_bounds__(kBlock) trace_kernel(TraceArgs<T> a) {
#include "trace_body.inc"
}

// The same body with the quad-cooperative record fetch (trace_device.h: coop_load_pair; float, 3D, trees of at most 64 levels and
// fewer than 2^26 pair records): one L1 line request per visited record instead of four lane requests.
// Waves per SIMD: the per-lane kernel runs best at the 7 its 70 VGPRs allow (forced to 8 it is 16-19 % slower on the soup: more lanes
// asking the L1 for four requests per record), this one at 8 (64 VGPRs, no spills; +4 % on the 1M soup and the 10M mesh, +4 % on
// the Sponza proxy: with a quarter of the L1 requests the extra wave is memory-level parallelism, not contention).
template <typename T, bool Any, bool Robust, int Leaf, bool Stats>
__global__ void __launch_bounds__(kBlock, 8) trace_kernel_coop(TraceArgs<T> a) {
    constexpr int D = 3;
    constexpr bool Deep = false;
#undef BVH_TRACE_COOP
#define BVH_TRACE_COOP true
#include "trace_body.inc"
#undef BVH_TRACE_COOP
}

// The cooperative body for the other record families (round 4): Node<double, 3> / Node<double, 2> (128-byte records as two
// quad-coalesced halves: two L1 requests per record instead of a lane's eight) and Node<float, 2> (the same 64-byte records as 3D).
// The transposes keep ~32 more registers live across the fetch than the per-lane loads: 4 waves per SIMD for double.
template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D>
__global__ void __launch_bounds__(kBlock, sizeof(T) == 4 ? 8 : 4) trace_kernel_coop_nd(TraceArgs<T> a) {
    constexpr bool Deep = false;
#undef BVH_TRACE_COOP
#define BVH_TRACE_COOP true
#include "trace_body.inc"
#undef BVH_TRACE_COOP
}

// Twins of the two kernels above under their own symbols, launched only while launch_traverse is MEASURING candidate plans for a tree
// (one whole batch per candidate, see there): a profile of an application then shows the search launches apart from the settled
// ones instead of averaging batches traced as given and batches traced reordered into one kernel's row.
template <typename T, bool Any, bool Robust, int Leaf>
__global__ void __launch_bounds__(kBlock) trace_kernel_plan_search(TraceArgs<T> a) {
    constexpr int D = 3;
    constexpr bool Deep = false, Stats = false;
#include "trace_body.inc"
}
template <typename T, bool Any, bool Robust, int Leaf>
__global__ void __launch_bounds__(kBlock, 8) trace_kernel_coop_plan_search(TraceArgs<T> a) {
    constexpr int D = 3;
    constexpr bool Deep = false, Stats = false;
#undef BVH_TRACE_COOP
#define BVH_TRACE_COOP true
#include "trace_body.inc"
#undef BVH_TRACE_COOP
}

// Coherence key of a ray: Morton code of its origin cell (128^3 grid over the root box; round 3: 7 bits per axis are 1 % better than 6
// on the soup for the same three passes, profiles/r03_entry_key_probe.txt) above the direction octant, 24 bits, three
// 8-bit radix passes. Rays of one key start in the same cell and descend the same way first; any order gives the same per-ray
// results. Measured with one ticket range per XCD on 2^24 uniform rays, rays physically permuted (tools/ray_order_probe.py,
// kernel ms): 1M-triangle soup 12.09 as given, 7.52 / 7.28 / 7.25 with 4 / 5 / 6 bits per axis + octant; 10M-triangle mesh 13.92,
// 7.79 / 7.41 / 7.09; octant-major and direction-cube keys lose (soup 7.81, mesh 8.25). The third pass costs ~0.13 ms.
// `hilbert_bits` > 0 (developer experiment, bvh_amd_experiment("key_curve", 1)): the cell's index along the 3D Hilbert curve of that
// many bits per axis instead of its Morton code (Skilling's axes-to-transpose transform): consecutive keys are always adjacent cells.
// Round 5 — `class_bits` > 0: LONG RAYS FIRST. The drain of the persistent grid (profiles/r05_tail_timeline_before.txt: every wave draws its
// last ticket at ~6.5 of 7.5 ms, then needs a median of 0.45 ms to finish the rays it holds; 7.5 % of the grid's time is lost there)
// is as long as the longest walks still in flight, and on the scenes that are reordered at all a walk's length goes with the length of
// the ray's chord through the root box. So the key carries, right below the three top bits of the cell index (the bits that roughly
// select the XCD's ticket range), the chord class of the ray, longest class first, and the cell index gives up its lowest `class_bits`
// bits (neighbours along the curve merge): every XCD still sweeps its part of space in curve order, once per class, and the tickets
// drawn last are the short rays. Per-ray results do not depend on the order.
__device__ inline float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }       // v_rcp_f32; ordering keys only
__device__ inline double rcp_fast(double x) { return 1.0 / x; }
template <typename T>
__device__ inline uint32_t ray_key(const T (&r)[8], T lx, T ly, T lz, T sx, T sy, T sz, uint32_t cells, int hilbert_bits, int class_bits, T class_scale) {
    const T q[3] = { (r[0] - lx) * sx, (r[1] - ly) * sy, (r[2] - lz) * sz };
    uint32_t code = 0;
    uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T v = q[k];
        v = v > T(0) ? v : T(0);                              // (NaN origins land in cell 0)
        cell[k] = v >= T(cells - 1) ? cells - 1 : static_cast<uint32_t>(v);
    }
    if (hilbert_bits > 0) {
        uint32_t X[3] = { cell[2], cell[1], cell[0] };        // X[0] ends up in the most significant bit of every triple
        const uint32_t M = 1u << (hilbert_bits - 1);
        for (uint32_t Q = M; Q > 1; Q >>= 1) {
            const uint32_t P = Q - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (X[a] & Q) X[0] ^= P;
                else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
            }
        }
        X[1] ^= X[0]; X[2] ^= X[1];
        uint32_t t = 0;
        for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
        X[0] ^= t; X[1] ^= t; X[2] ^= t;
        cell[2] = X[0]; cell[1] = X[1]; cell[0] = X[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t c = cell[k];
        uint32_t s = (c & 1u) | ((c & 2u) << 2) | ((c & 4u) << 4) | ((c & 8u) << 6) | ((c & 16u) << 8) | ((c & 32u) << 10) | ((c & 64u) << 12) | ((c & 128u) << 14);
        code |= s << k;
    }
    const uint32_t oct = (Num<T>::sign(r[3]) ? 1u : 0u) | (Num<T>::sign(r[4]) ? 2u : 0u) | (Num<T>::sign(r[5]) ? 4u : 0u);
    if (class_bits > 0) {
        // chord of the ray through the root box against the box diagonal: slab test against [l, l + cells / s] with the ray's own tmin /
        // tmax. The key only ORDERS rays, so this is the one place of the library that uses the hardware's approximate reciprocal
        // (v_rcp_f32, 1 ulp) instead of an IEEE division — the kernel is bound by its instructions (16.8 M rays x ~400), not by the 512 MB
        // it reads — and the one-bit case compares squares instead of taking two square roots.
        const T l[3] = { lx, ly, lz }, sc[3] = { sx, sy, sz };
        T t0 = r[6], t1 = r[7], d2 = T(0), diag2 = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T ext = sc[k] > T(0) ? T(cells) * rcp_fast(sc[k]) : T(0);
            const T inv = rcp_fast(r[3 + k]);
            const T a = (l[k] - r[k]) * inv, b = (l[k] + ext - r[k]) * inv;
            const T lo_t = a < b ? a : b, hi_t = a < b ? b : a;         // (NaN from 0 * inf compares false: that slab does not clip)
            t0 = lo_t > t0 ? lo_t : t0; t1 = hi_t < t1 ? hi_t : t1;
            d2 += r[3 + k] * r[3 + k]; diag2 += ext * ext;
        }
        const uint32_t top = (1u << class_bits) - 1u;
        uint32_t cls = 0;
        const T len = t1 > t0 ? t1 - t0 : T(0);
        if (class_bits == 1) cls = len * len * d2 * class_scale * class_scale >= diag2 && diag2 > T(0) ? 1u : 0u;
        else {
            const T rel = diag2 > T(0) ? len * Num<T>::sqrt_(d2) / Num<T>::sqrt_(diag2) * class_scale : T(0);
            cls = rel >= T(top) ? top : rel > T(0) ? static_cast<uint32_t>(rel) : 0u;
        }
        cls = top - cls;                                                // longest class first
        const int code_bits = 3 * (hilbert_bits > 0 ? hilbert_bits : 31 - __clz(cells));
        const uint32_t high = code >> (code_bits - 3), low = (code & ((1u << (code_bits - 3)) - 1u)) >> class_bits;
        code = (((high << class_bits) | cls) << (code_bits - 3 - class_bits)) | low;
    }
    return (code << 3) | oct;
}

// One block = one tile of the radix sort's first pass (kRadixTileU32 keys: 1024 threads x 4, strided so that a wave's loads are contiguous):
// the keys are written AND the tile's histogram of their lowest digit is left where k_radix_hist would have put it, so the sort's first
// histogram pass — a second read of all keys — is not launched (round 5: 132 + 27 -> ~115 us per 2^24 rays; the rays are loaded
// non-temporally: they are read once here and once, much later, by the traversal).
#if defined(BVH_AMD_DEVELOPER)
__global__ void __launch_bounds__(256) k_check_order(const uint32_t* order, uint32_t n, uint32_t* bad) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && order[i] >= n) {
        const uint32_t k = atomicAdd(&bad[0], 1u);
        if (k < 3) { bad[1 + 2 * k] = i; bad[2 + 2 * k] = order[i]; }
    }
}
#endif

template <typename T>
__global__ void __launch_bounds__(1024) ray_keys_kernel(const T* rays, uint32_t n, T lx, T ly, T lz, T sx, T sy, T sz, uint32_t* keys, uint32_t cells,
                                                        int hilbert_bits, int class_bits, T class_scale, uint32_t* hist, uint32_t tiles) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * uint32_t(kRadixTileU32);
#pragma unroll
    for (int s0 = 0; s0 < kRadixTileU32 / 1024; ++s0) {
        const uint32_t i = base + uint32_t(s0) * 1024u + threadIdx.x;
        if (i < n) {
            T r[8];
            load_ray_nt(rays + 8ull * i, r);
            const uint32_t key = ray_key<T>(r, lx, ly, lz, sx, sy, sz, cells, hilbert_bits, class_bits, class_scale);
            keys[i] = key;
            atomicAdd(&h[key & 0xFFu], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[size_t{threadIdx.x} * tiles + blockIdx.x] = h[threadIdx.x];   // digit-major, tile-minor: k_radix_hist's layout (sort_emul.hip)
}

// Developer experiment (BVH_AMD_RAY_KEY_DEPTH=k; profiles/r03_traversal_experiments.md §6): a TREE-ENTRY key — the ray descends the
// top k levels of the tree the way the traversal will (nearer hit child first, plain slab test: the key only orders rays) and the
// key is the path it took (one bit per level, left-aligned) above the direction octant: rays are ordered by the subtree they enter
// first, i.e. by what they will fetch, instead of by the grid cell they start in.
__global__ void __launch_bounds__(256) ray_entry_keys_kernel(const float* rays, uint32_t n, const PairNode<float>* pairs, uint32_t root_index, int depth,
                                                             uint32_t* keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r[8];
    load_ray(rays + 8ull * i, r);
    const float inv[3] = { 1.0f / r[3], 1.0f / r[4], 1.0f / r[5] };
    uint32_t cur = root_index, path = 0;
    int level = 0;
    for (; level < depth && (cur & kCountMask) == 0; ++level) {
        float lb[6], rb[6];
        uint32_t li, ri;
        load_pair(pairs + (cur >> (kCountBits + 1)), lb, rb, li, ri);
        float t0[2] = { r[6], r[6] }, t1[2] = { r[7], r[7] };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a0 = (lb[2 * k] - r[k]) * inv[k], a1 = (lb[2 * k + 1] - r[k]) * inv[k];
            const float b0 = (rb[2 * k] - r[k]) * inv[k], b1 = (rb[2 * k + 1] - r[k]) * inv[k];
            t0[0] = fmaxf(t0[0], fminf(a0, a1)); t1[0] = fminf(t1[0], fmaxf(a0, a1));
            t0[1] = fmaxf(t0[1], fminf(b0, b1)); t1[1] = fminf(t1[1], fmaxf(b0, b1));
        }
        const bool hl = t0[0] <= t1[0], hr = t0[1] <= t1[1];
        if (!hl && !hr) break;
        const bool right = hl && hr ? t0[1] < t0[0] : hr;
        path = (path << 1) | (right ? 1u : 0u);
        cur = right ? ri : li;
    }
    path <<= (depth - level);
    const uint32_t oct = (Num<float>::sign(r[3]) ? 1u : 0u) | (Num<float>::sign(r[4]) ? 2u : 0u) | (Num<float>::sign(r[5]) ? 4u : 0u);
    keys[i] = (path << 3) | oct;
}

// BVH_AMD_RAY_ORIGINAL_IDS: BVH-order index -> bvh.prim_ids[index] in place (misses keep BVH_AMD_INVALID)
template <typename H>
__global__ void __launch_bounds__(256) original_ids_kernel(H* hits, size_t n, const uint32_t* prim_ids, uint32_t prim_count) {
    const size_t i = blockIdx.x * size_t{256} + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = hits[i].prim;
    if (p < prim_count) hits[i].prim = prim_ids[p];
}

// Tuning overrides of the calling thread (bvh_amd_tuning; developer A/B runs in one process): < 0 = the default / the environment
thread_local int t_refill = -1, t_leaf = -1, t_coop = -1, t_parts = -1;
// Developer experiments of the calling thread (bvh_amd_experiment(name, value); -1 = the default): never change a result.
//   grid_blocks   cap of the persistent grid (blocks of 256 lanes)
//   stream_hints  bit 0: rays / order / hit records loaded and stored non-temporally (default on; 0 = off)
//   tri_stride    floats from one PrecomputedTri to the next in the CALLER's primitive array (12; 16 = padded to a 64-byte line)
//   key_curve     coherence key of the ray reordering: 1 Hilbert index of the origin cell (default), 0 its Morton code
//   key_bits      bits per axis of that cell grid (1..8)
//   wave_times    (developer library only) 1: the next launches record per-wave begin / last-refill / end timestamps (bvh_amd_wave_times)
//   one_shot      1 / 0: force / forbid the one-shot grid of small batches (default: launch_planned decides by batch size)
//   key_class_bits / key_class_scale   chord classes of the reordering key (0 = none) / classes per 100 box diagonals (ray_keys_kernel)
struct Experiments { int grid_blocks = -1, stream_hints = -1, tri_stride = -1, key_curve = -1, key_bits = -1, step_events = -1, wave_times = -1, one_shot = -1,
                     key_class_bits = -1, key_class_scale = -1, stagger = -1; };
// Round-5 measurements (profiles/r05_key_class_ab.txt, r05_stagger_ab.txt; 2^24 rays, kernel ms): 1M soup 6.92 without classes, one class bit
// 6.99 / 6.80 / 6.76 / 6.70 / 6.64 at 2 / 2.5 / 3 / 4 / 5 classes per diagonal (only the shortest chords go last), two bits 6.80-6.84, three
// 6.91-6.95; the 10M soup LOSES 6 % with any of them (83 % of its rays end at a hit long before their chord does: the chord predicts nothing
// there, and two sweeps cost coherence) — so the classes are not a default but one more candidate of the measured plan search.
constexpr int kKeyClassBits = 0, kKeyClassScalePercent = 500;
// Staggered drain (trace_body.inc), tickets per eighth of the grid: kernel ms against it — 1M soup 2^24 rays 7.03 / 6.93 / 6.90 / 7.17 at
// 0 / 84k / 168k / 262k (with the chord classes 6.74 / 6.60 / 6.81 / 6.97), 2^22 rays 2.57 -> 2.46 at 131k, 10M soup 12.5M rays 5.50 -> 5.46
// at 125k; light trees want less: Sponza proxy 1M rays 0.264 -> 0.236 at 44k (configs[1]: 3.7 -> 4.0 Grays/s), 4M rays 0.628 -> 0.597 at 42k,
// terrain 4M 0.649 -> 0.636 at 65k; a tenth of a batch per class is always too much.
constexpr uint32_t kStaggerHeavy = 100000, kStaggerLight = 50000, kStaggerDouble = 20000;
thread_local bool g_last_classes = false;
// what the plan search of the calling thread's latest finished search had measured (bvh_amd_last_plan_search: the bench line quotes it)
thread_local float g_search_ns[5] = {0, 0, 0, 0, 0};
thread_local int g_search_count[5] = {0, 0, 0, 0, 0};
thread_local unsigned g_search_dropped = 0;
thread_local Experiments t_exp;
#if defined(BVH_AMD_DEVELOPER)
// per-wave timeline of the calling thread's latest traversal launch (drain-tail study, profiles/r05_tail_timeline.txt)
constexpr size_t kWaveTimeWords = 6, kWaveTimeWaves = 8 * 256 * (kBlock / kWave) * 2;
thread_local unsigned long long* t_wave_times = nullptr;
thread_local size_t t_wave_times_waves = 0;
#endif
thread_local std::pair<hipEvent_t, hipEvent_t>* t_calibration = nullptr;   // events to record around the next traversal kernel of this thread

// BVH_AMD_COOP=0 / 1 (or bvh_amd_tuning) forces the per-lane / quad-cooperative record fetch of the float 3D kernels for A/B
// runs; -1: launch_traverse decides by kind of launch (trace_device.h: kCoop*)
int coop_fetch_forced() {
    static const int knob = BVH_DEV_INT("BVH_AMD_COOP", -1);
    return t_coop >= 0 ? (t_coop != 0) : knob < 0 ? -1 : (knob != 0);
}

struct Grid { int blocks = 0; };

template <typename K>
int persistent_grid(K kernel, int device, Grid& g) {
    int per_cu = 0, cus = 0;
    BVH_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), BVH_AMD_ERR_HIP);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    g.blocks = per_cu * cus;
    return BVH_AMD_OK;
}

template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D, bool Deep, bool Coop = false>
int launch_variant_d(const BvhImpl<T>& b, const TraceArgs<T>& args, hipStream_t stream, const char* name) {
    static thread_local int cached_blocks[16] = {0};
    void (*kernel)(TraceArgs<T>) = nullptr;
    constexpr bool CoopNd = Coop && !(std::is_same_v<T, float> && D == 3);
    if constexpr (CoopNd) kernel = trace_kernel_coop_nd<T, Any, Robust, Leaf, Stats, D>;
    else if constexpr (Coop) kernel = trace_kernel_coop<T, Any, Robust, Leaf, Stats>;
    else kernel = trace_kernel<T, Any, Robust, Leaf, Stats, D, Deep>;
    if constexpr (std::is_same_v<T, float> && D == 3 && !Deep && !Stats) {
        if (t_calibration) {                                  // a candidate plan being measured: same code, its own symbol
            if constexpr (Coop) kernel = trace_kernel_coop_plan_search<T, Any, Robust, Leaf>;
            else kernel = trace_kernel_plan_search<T, Any, Robust, Leaf>;
        }
    }
    int& blocks = cached_blocks[b.device & 15];
    if (blocks == 0) {
        Grid g;
        int rc = persistent_grid(kernel, b.device, g);
        if (rc) return rc;
        blocks = g.blocks;
    }
    unsigned long long need = (args.n + kBlock - 1) / kBlock;
    int grid = static_cast<int>(need < static_cast<unsigned long long>(blocks) ? need : blocks);
    if (args.one_shot) grid = static_cast<int>(need);
    else if constexpr (!Coop && std::is_same_v<T, float>) {
        // The per-lane kernel runs best BELOW the 7 blocks per CU its registers allow (four L1 requests per record and lane: the
        // seventh wave per SIMD adds contention, not throughput) — measured round 4 at 4..8 blocks per CU (profiles/r04_experiments_call2.txt):
        // 6 is equal or better on every scene (terrain 1.079 -> 1.072 ms, Sponza any-hit 1.829 -> 1.812, 10M mesh 5.615 -> 5.598, soup as
        // given 12.01 -> 11.95), and batches of up to 2^21 rays, whose time is mostly the dependent chain of their longest rays, want 5
        // (configs[1], 1M rays: 0.283 -> 0.254-0.268 ms; 512k rays: 0.234 -> 0.202 ms).
        int cus = blocks / 7 > 0 ? blocks / 7 : 1;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b.device);
        // (round 5: with the staggered drain and the earlier refill of small batches 6 is as good as 5 there too — 1M rays 0.2472 / 0.2519)
        const int cap = 6 * cus;
        if (grid > cap) grid = cap;
    }
    static const int grid_env = BVH_DEV_INT("BVH_AMD_GRID_BLOCKS", 0);   // developer knob: fewer resident waves (occupancy studies)
    if (!args.one_shot) {
        if (grid_env > 0 && grid > grid_env) grid = grid_env;
        if (t_exp.grid_blocks > 0 && grid > t_exp.grid_blocks) grid = t_exp.grid_blocks;
    }
    if (grid < 1) grid = 1;
    (void)name;
    // the symbol as rocprofv3 prints it (profiles/*_kernel_stats.csv), for bench.py's roofline.kernel
    static const std::string head = std::string(CoopNd ? "trace_kernel_coop_nd<" : Coop ? "trace_kernel_coop<" : "trace_kernel<") + (std::is_same_v<T, float> ? "float" : "double") + ", " +
                                    (Any ? "true" : "false") + ", " + (Robust ? "true" : "false") + ", " + std::to_string(Leaf) + ", " + (Stats ? "true" : "false");
    static const std::string symbol = CoopNd ? head + ", " + std::to_string(D) + ">" : Coop ? head + ">" : head + ", " + std::to_string(D) + ", " + (Deep ? "true" : "false") + ">";
    g_last_kernel = symbol.c_str();
    KernelTimer& timer = kernel_timer();
    hipEvent_t stop = nullptr;
    if (timer.on) {                                           // bvh_amd_kernel_timing: events on the launch stream around this kernel only
        std::pair<hipEvent_t, hipEvent_t>& ev = timer.ring[timer.count % KernelTimer::kRing];
        if (!timer.begin_armed && timer.begin[timer.count % KernelTimer::kRing]) {     // no call-level start for this launch: forget a stale one
            (void)hipEventDestroy(timer.begin[timer.count % KernelTimer::kRing]);
            timer.begin[timer.count % KernelTimer::kRing] = nullptr;
        }
        timer.begin_armed = false;
        if (!ev.first) {
            BVH_HIP_TRY(hipEventCreate(&ev.first), BVH_AMD_ERR_HIP);
            BVH_HIP_TRY(hipEventCreate(&ev.second), BVH_AMD_ERR_HIP);
        }
        BVH_HIP_TRY(hipEventRecord(ev.first, stream), BVH_AMD_ERR_HIP);
        stop = ev.second;
        ++timer.count;
    }
    if (t_calibration) (void)hipEventRecord(t_calibration->first, stream);     // launch_traverse is measuring candidate plans
#if defined(BVH_AMD_DEVELOPER)
    TraceArgs<T> timed_args = args;
    if (t_exp.wave_times > 0 && size_t(grid) * (kBlock / kWave) <= kWaveTimeWaves) {
        if (!t_wave_times) BVH_HIP_TRY(hipMalloc(&t_wave_times, kWaveTimeWaves * kWaveTimeWords * sizeof(unsigned long long)), BVH_AMD_ERR_HIP);
        t_wave_times_waves = size_t(grid) * (kBlock / kWave);
        BVH_HIP_TRY(hipMemsetAsync(t_wave_times, 0, t_wave_times_waves * kWaveTimeWords * sizeof(unsigned long long), stream), BVH_AMD_ERR_HIP);
        timed_args.wave_times = t_wave_times;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, timed_args);
#else
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, args);
#endif
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    if (t_calibration) (void)hipEventRecord(t_calibration->second, stream);
    if (stop) BVH_HIP_TRY(hipEventRecord(stop, stream), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

template <typename T, bool Any, bool Robust, int Leaf, bool Stats, int D>
int launch_variant(const BvhImpl<T>& b, const TraceArgs<T>& args, hipStream_t stream, const char* name) {
    if (args.deep) return launch_variant_d<T, Any, Robust, Leaf, Stats, D, true>(b, args, stream, name);
    // (32-bit byte offsets into the records: 64 / 128 bytes each)
    if (args.coop && b.pair_count < (size_t{1} << (std::is_same_v<T, float> ? 26 : 25))) return launch_variant_d<T, Any, Robust, Leaf, Stats, D, false, true>(b, args, stream, name);
    return launch_variant_d<T, Any, Robust, Leaf, Stats, D, false>(b, args, stream, name);
}

#define BVH_VARIANT(T, ANY, ROB, LEAF, STATS) \
    launch_variant<T, ANY, ROB, LEAF, STATS, D>(b, args, stream, D == 3 ? "trace_kernel<" #T "," #ANY "," #ROB "," #LEAF "," #STATS ">" \
                                                                        : "trace_kernel<" #T "," #ANY "," #ROB "," #LEAF "," #STATS ",2>")

template <typename T, int Leaf, int D = 3>
int dispatch(const BvhImpl<T>& b, const TraceArgs<T>& args, unsigned flags, bool stats, hipStream_t stream) {
    const bool any = flags & BVH_AMD_RAY_ANY_HIT, rob = flags & BVH_AMD_RAY_ROBUST;
    if (stats) {
        if (any) return rob ? BVH_VARIANT(T, true, true, Leaf, true) : BVH_VARIANT(T, true, false, Leaf, true);
        return rob ? BVH_VARIANT(T, false, true, Leaf, true) : BVH_VARIANT(T, false, false, Leaf, true);
    }
    if (any) return rob ? BVH_VARIANT(T, true, true, Leaf, false) : BVH_VARIANT(T, true, false, Leaf, false);
    return rob ? BVH_VARIANT(T, false, true, Leaf, false) : BVH_VARIANT(T, false, false, Leaf, false);
}


// ---- one ray, leaves handed to a HOST callback (c_api/bvh.h:277-295 over bvh_impl.h:235-250; Bvh::intersect with a
// leaf lambda, bvh.h:160-182) ------------------------------------------------------------------------------------------------
// The callback may shorten the ray (it gets &ray.tmax), and the traversal culls with the current tmax, so the device cannot run
// ahead of the host unconditionally. It runs ahead SPECULATIVELY: one launch walks the tree with the tmax it was given and
// logs, in order, every leaf it reaches (and every pair it visits, when an inner callback wants them) together with the
// traversal stack at that moment. The host replays the log through the callbacks; as long as a callback leaves tmax alone the
// next logged event is exactly what the reference would do next. When a callback changes tmax, the rest of the log is
// discarded and the walk restarts from that leaf's stack snapshot with the new tmax. Typical cost: 1 + (number of accepted
// hits) launches per ray. One lane walks; the wavefront's other lanes only move stacks around.
constexpr uint32_t kStepLds = 1024;            // stack entries held in LDS (deeper ones live in the context's HBM scratch)
constexpr uint32_t kStepContinue = 0xFFFFFFFFu;
constexpr uint32_t kStepEvents = 2048;
constexpr uint32_t kStepHeader = 4;            // out[0] events, out[1] traversal finished, out[2] error, out[3] launch number (written last)

template <typename T>
struct StepArgs {
    const PairNode<T>* pairs;
    T ray[8];                                  // org, dir (z = 0 in 2D), tmin, tmax
    const uint32_t* in;                        // pinned: [0] = number of stack entries (bottom first) or kStepContinue
    uint32_t* out;                             // pinned: header | kStepEvents x {index word, snapshot offset, snapshot length} | snapshots
    uint32_t* scratch;                         // device: [0] = size of the stack left by the previous launch, [1..] its entries
    uint32_t stack_cap;                        // entries the stack may hold
    uint32_t snap_words;
    uint32_t max_events;                       // <= kStepEvents
    uint32_t max_steps;                        // pairs in the tree: no walk visits a pair twice, a cyclic (malformed) tree would
    uint32_t seq;                              // launch number, stored to out[3] once the log is complete (the host polls it)
    uint32_t any, record_pairs;
};

template <typename T, bool Robust, int D>
__global__ void __launch_bounds__(kWave) ray_step_kernel(StepArgs<T> a) {
    __shared__ uint32_t s_stack[kStepLds];
    __shared__ uint32_t s_sp;
    const uint32_t lane = threadIdx.x;
    uint32_t sp = a.in[0];
    if (sp == kStepContinue) {
        sp = a.scratch[0];
        for (uint32_t i = lane; i < sp && i < kStepLds; i += kWave) s_stack[i] = a.scratch[1 + i];
    } else {
        for (uint32_t i = lane; i < sp; i += kWave) {
            const uint32_t v = a.in[1 + i];
            if (i < kStepLds) s_stack[i] = v; else a.scratch[1 + i] = v;
        }
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t* events = a.out + kStepHeader;
        uint32_t* snaps = events + 3 * kStepEvents;
        uint32_t n_events = 0, snap_used = 0, finished = 0, error = 0, steps = 0;
        T org[3] = {0, 0, 0}, inv[3] = {0, 0, 0}, aux[3] = {0, 0, 0};
        uint32_t oct[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < D; ++k) {                                   // bvh.h:162-165, ray.h:29-48 (as in trace_kernel)
            org[k] = a.ray[k];
            const T d = a.ray[3 + k];
            const T iv = Robust ? T(1) / d : (Num<T>::abs_(d) <= Num<T>::kEps ? Num<T>::copysign_(Num<T>::kMax, d) : T(1) / d);
            inv[k] = iv;
            aux[k] = Robust ? (Num<T>::finite(iv) ? Num<T>::bump2(iv) : iv) : (-iv) * org[k];
            oct[k] = Num<T>::sign(d) ? 1u : 0u;
        }
        const T tmin = a.ray[6], tmax = a.ray[7];
        auto put = [&](uint32_t i, uint32_t v) { if (i < kStepLds) s_stack[i] = v; else a.scratch[1 + i] = v; };
        auto get = [&](uint32_t i) { return i < kStepLds ? s_stack[i] : a.scratch[1 + i]; };
        for (;;) {                                                      // bvh.h:128-156
            if (sp == 0) { finished = 1; break; }
            uint32_t top = get(--sp);
            bool dead = false, stop = false;
            while ((top & kCountMask) == 0) {
                if (++steps > a.max_steps) { error = 2; stop = true; break; }
                if (a.record_pairs) {
                    if (n_events == a.max_events) { stop = true; break; }
                    events[3 * n_events] = top; events[3 * n_events + 1] = 0; events[3 * n_events + 2] = 0;
                    ++n_events;
                }
                T lb[6], rb[6];
                uint32_t li = 0, ri = 0;
                load_pair(a.pairs + (top >> (kCountBits + 1)), lb, rb, li, ri);
                T l0 = tmin, l1 = tmax, r0 = tmin, r1 = tmax;           // node.h:105-117
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const T ln = oct[k] ? lb[2 * k + 1] : lb[2 * k], lf = oct[k] ? lb[2 * k] : lb[2 * k + 1];
                    const T rn = oct[k] ? rb[2 * k + 1] : rb[2 * k], rf = oct[k] ? rb[2 * k] : rb[2 * k + 1];
                    T la, lz, ra, rz;
                    if (Robust) {                                       // node.h:74-75
                        la = (ln - org[k]) * inv[k]; lz = (lf - org[k]) * aux[k];
                        ra = (rn - org[k]) * inv[k]; rz = (rf - org[k]) * aux[k];
                    } else {                                            // node.h:85-86
                        la = Num<T>::fma_(ln, inv[k], aux[k]); lz = Num<T>::fma_(lf, inv[k], aux[k]);
                        ra = Num<T>::fma_(rn, inv[k], aux[k]); rz = Num<T>::fma_(rf, inv[k], aux[k]);
                    }
                    l0 = pick_max(la, l0); l1 = pick_min(lz, l1);
                    r0 = pick_max(ra, r0); r1 = pick_min(rz, r1);
                }
                const bool hl = l0 <= l1, hr = r0 <= r1;                // bvh.h:177-180
                if (hl) {
                    uint32_t near_i = li;
                    if (hr) {
                        uint32_t far_i = ri;
                        if (!a.any && l0 > r0) { near_i = ri; far_i = li; }
                        if (sp >= a.stack_cap) { error = 1; stop = true; break; }
                        put(sp++, far_i);
                    }
                    top = near_i;
                } else if (hr) {
                    top = ri;
                } else {
                    dead = true;                                        // bvh.h:147-148: back to the stack
                    break;
                }
            }
            if (!stop && !dead && (n_events == a.max_events || snap_used + sp > a.snap_words)) stop = true;   // no room to log this leaf
            if (stop) {                                                 // resume here next time: the unprocessed node goes back on top
                if (!error) { if (sp >= a.stack_cap) error = 1; else put(sp++, top); }
                break;
            }
            if (dead) continue;
            events[3 * n_events] = top; events[3 * n_events + 1] = snap_used; events[3 * n_events + 2] = sp;
            for (uint32_t i = 0; i < sp; ++i) snaps[snap_used + i] = get(i);
            snap_used += sp;
            ++n_events;
        }
        a.out[0] = n_events; a.out[1] = finished; a.out[2] = error;
        // the log lives in fine-grained host memory: release it to the polling host without waiting for the kernel to retire
        __hip_atomic_store(&a.out[3], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        s_sp = sp;
    }
    __syncthreads();
    sp = s_sp;
    for (uint32_t i = lane; i < sp && i < kStepLds; i += kWave) a.scratch[1 + i] = s_stack[i];
    if (lane == 0) a.scratch[0] = sp;
}

struct StepContext {                           // one per calling thread: the per-ray entry points are re-entrant
    int device = -1;
    hipStream_t stream = nullptr;
    uint32_t* pinned = nullptr;
    uint32_t* scratch = nullptr;
    uint32_t stack_cap = 0, snap_words = 0;
    uint32_t seq = 0;
    void release() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (pinned) (void)hipHostFree(pinned);
        if (scratch) (void)hipFree(scratch);
        if (stream) (void)hipStreamDestroy(stream);
        pinned = nullptr; scratch = nullptr; stream = nullptr; stack_cap = 0; device = -1;
    }
    ~StepContext() { release(); }
    int ensure(int dev, uint32_t cap) {
        if (device == dev && stack_cap >= cap) return BVH_AMD_OK;
        release();
        BVH_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), BVH_AMD_ERR_HIP);
        snap_words = std::max<uint32_t>(16384u, 4 * cap);
        const size_t words = size_t{1} + cap + kStepHeader + 3 * size_t{kStepEvents} + snap_words;
        BVH_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pinned), words * sizeof(uint32_t), hipHostMallocCoherent), BVH_AMD_ERR_HIP);
        std::memset(pinned, 0, words * sizeof(uint32_t));
        BVH_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scratch), (size_t{1} + cap) * sizeof(uint32_t)), BVH_AMD_ERR_HIP);
        device = dev; stack_cap = cap;
        return BVH_AMD_OK;
    }
};
// One context per NESTING DEPTH of the per-ray entry points on the calling thread: a leaf callback may itself call
// bvhXX_intersect_ray* / Bvh::intersect on another (or the same) BVH — two-level / instanced scenes, which the reference's
// stack-local design allows (bvh_impl.h:244) — and the outer walk is still replaying its log, its pinned buffers and its
// snapshots when that happens. The inner call therefore claims the next context instead of re-using (or re-allocating) the
// outer one's.
struct StepContextPool {
    std::vector<std::unique_ptr<StepContext>> contexts;
    size_t depth = 0;
};
thread_local StepContextPool t_steps;
struct StepContextClaim {
    StepContext* ctx;
    StepContextClaim() {
        if (t_steps.depth == t_steps.contexts.size()) t_steps.contexts.push_back(std::make_unique<StepContext>());
        ctx = t_steps.contexts[t_steps.depth++].get();
    }
    ~StepContextClaim() { --t_steps.depth; }
    StepContextClaim(const StepContextClaim&) = delete;
    StepContextClaim& operator=(const StepContextClaim&) = delete;
};

} // namespace

void last_launch_plan(int out[4]) { for (int k = 0; k < 4; ++k) out[k] = g_last_plan[k]; }
void set_tuning(int refill, int leaf, int coop, int parts) { t_refill = refill; t_leaf = leaf; t_coop = coop; t_parts = parts; }
int set_experiment(const char* name, int value) {
    const std::string k = name ? name : "";
    if (k == "grid_blocks") t_exp.grid_blocks = value;
    else if (k == "stream_hints") t_exp.stream_hints = value;
    else if (k == "tri_stride") t_exp.tri_stride = value;
    else if (k == "key_curve") t_exp.key_curve = value;
    else if (k == "key_bits") t_exp.key_bits = value;
    else if (k == "step_events") t_exp.step_events = value;
    else if (k == "one_shot") t_exp.one_shot = value;
    else if (k == "stagger") t_exp.stagger = value;
    else if (k == "key_class_bits") t_exp.key_class_bits = value;
    else if (k == "key_class_scale") t_exp.key_class_scale = value;
    else if (k == "wave_times") {
#if defined(BVH_AMD_DEVELOPER)
        t_exp.wave_times = value;
#else
        return fail(BVH_AMD_ERR_ARG, "bvh_amd_experiment: 'wave_times' exists in the developer library only (python -m bvh_amd.build --developer)");
#endif
    }
    else if (k == "reset") t_exp = Experiments{};
    else return fail(BVH_AMD_ERR_ARG, "bvh_amd_experiment: unknown knob '" + k + "'");
    return BVH_AMD_OK;
}
int wave_times(unsigned long long* out, size_t capacity_waves, size_t* n_waves) {
#if defined(BVH_AMD_DEVELOPER)
    if (n_waves) *n_waves = t_wave_times_waves;
    if (!out || !t_wave_times) return BVH_AMD_OK;
    BVH_HIP_TRY(hipDeviceSynchronize(), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipMemcpy(out, t_wave_times, std::min(capacity_waves, t_wave_times_waves) * kWaveTimeWords * sizeof(unsigned long long), hipMemcpyDeviceToHost), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
#else
    (void)out; (void)capacity_waves;
    if (n_waves) *n_waves = 0;
    return fail(BVH_AMD_ERR_ARG, "bvh_amd_wave_times: developer library only");
#endif
}
void last_plan_search(float ns_per_ray[5], int measurements[5], unsigned* dropped) {
    for (int c = 0; c < 5; ++c) { if (ns_per_ray) ns_per_ray[c] = g_search_ns[c]; if (measurements) measurements[c] = g_search_count[c]; }
    if (dropped) *dropped = g_search_dropped;
}
const char* last_kernel_name() { return g_last_kernel; }
bool last_launch_reordered() { return g_last_reordered; }

void kernel_timing(bool on) {
    KernelTimer& t = kernel_timer();
    t.on = on;
    t.count = 0;
}

int kernel_times(float* ms_out, size_t capacity, size_t* count_out) {
    KernelTimer& t = kernel_timer();
    const size_t have = std::min<size_t>(t.count, KernelTimer::kRing), n = std::min(have, capacity);
    for (size_t i = 0; i < n; ++i) {                          // the latest `n` launches, oldest first
        const std::pair<hipEvent_t, hipEvent_t>& ev = t.ring[(t.count - n + i) % KernelTimer::kRing];
        BVH_HIP_TRY(hipEventSynchronize(ev.second), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipEventElapsedTime(&ms_out[i], ev.first, ev.second), BVH_AMD_ERR_HIP);
    }
    if (count_out) *count_out = n;
    return BVH_AMD_OK;
}

// Per launch of the calling thread (as kernel_times): milliseconds between the start of the call on its stream and the start of
// the traversal kernel = the ray keys + the radix sort of the reordering (0 when the batch was traced as given).
int reorder_times(float* ms_out, size_t capacity, size_t* count_out) {
    KernelTimer& t = kernel_timer();
    const size_t have = std::min<size_t>(t.count, KernelTimer::kRing), n = std::min(have, capacity);
    for (size_t i = 0; i < n; ++i) {
        const size_t k = (t.count - n + i) % KernelTimer::kRing;
        ms_out[i] = 0.0f;
        if (!t.begin[k]) continue;
        BVH_HIP_TRY(hipEventSynchronize(t.ring[k].first), BVH_AMD_ERR_HIP);
        BVH_HIP_TRY(hipEventElapsedTime(&ms_out[i], t.begin[k], t.ring[k].first), BVH_AMD_ERR_HIP);
    }
    if (count_out) *count_out = n;
    return BVH_AMD_OK;
}

template <typename T>
static int to_original_ids(const BvhImpl<T>& b, typename HitOf<T>::Type* d_hits, size_t n, hipStream_t stream) {
    if (!b.d_prim_ids) return fail(BVH_AMD_ERR_ARG, "intersect_rays: BVH has no device prim ids");
    hipLaunchKernelGGL(original_ids_kernel<typename HitOf<T>::Type>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, d_hits, n,
                       b.d_prim_ids, static_cast<uint32_t>(b.prim_count));
    BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

constexpr float kReorderMinVisits = 100.0f;

// How one launch is traced: the batch reordered or as given, records fetched per lane or quad-cooperatively, the refill / leaf
// thresholds of the persistent waves. Never affects results.
struct Plan { bool reorder; bool coop; int refill, leaf; bool classes = false; };   // classes: long rays first (chord class in the reordering key)

template <typename T>
static int launch_planned(const BvhImpl<T>& b, int leaf_kind, const T* d_prims, const T* d_rays, size_t n, unsigned flags,
                          typename HitOf<T>::Type* d_hits, bvh_amd_counters* d_counters, hipStream_t stream, const Plan* plan)
{
    if (n == 0) return BVH_AMD_OK;
    if (!d_prims || !d_rays || !d_hits) return fail(BVH_AMD_ERR_ARG, "intersect_rays: null device pointer");
    if (b.node_count == 0 || !b.d_work) return fail(BVH_AMD_ERR_ARG, "intersect_rays: BVH has no device copy");
    if (b.pair_count && !b.d_pairs) return fail(BVH_AMD_ERR_ARG, "intersect_rays: BVH has no device nodes");
    uint32_t slot = 0;
    hipEvent_t slot_event = nullptr;
    {   // claim a free slot of ticket counters; whoever used it before (possibly on another stream) must be done before it is zeroed again
        std::unique_lock<std::mutex> lock(b.work_mutex);
        for (;;) {
            const uint32_t from = b.work_next.load();
            bool found = false;
            for (uint32_t i = 0; i < BvhImpl<T>::kWorkSlots && !found; ++i) {
                const uint32_t s = (from + i) % BvhImpl<T>::kWorkSlots;
                if (!b.work_busy[s]) { slot = s; found = true; }
            }
            if (found) break;
            lock.unlock();                                    // kWorkSlots launches are between claim and record right now
            std::this_thread::yield();
            lock.lock();
        }
        if (!b.work_done[slot]) BVH_HIP_TRY(hipEventCreateWithFlags(&b.work_done[slot], hipEventDisableTiming), BVH_AMD_ERR_HIP);
        else BVH_HIP_TRY(hipStreamWaitEvent(stream, b.work_done[slot], 0), BVH_AMD_ERR_HIP);
        b.work_busy[slot] = true;
        b.work_next.store((slot + 1) % BvhImpl<T>::kWorkSlots);
        slot_event = b.work_done[slot];
    }
    unsigned long long* work = b.d_work + size_t{slot} * BvhImpl<T>::kWorkStride;
    if (KernelTimer& timer = kernel_timer(); timer.on) {     // start of the call on its stream (bvh_amd_reorder_times)
        hipEvent_t& ev = timer.begin[timer.count % KernelTimer::kRing];
        if (!ev) (void)hipEventCreate(&ev);
        if (ev) timer.begin_armed = hipEventRecord(ev, stream) == hipSuccess;
    }
    TraceArgs<T> args{};
    args.pairs = b.d_pairs; args.prims = d_prims; args.rays = d_rays; args.hits = d_hits;
    args.n = n; args.work = work; args.counters = d_counters; args.root_index = b.root_index;
    args.order = nullptr;
    args.prim_stride = leaf_kind == LEAF_TRIANGLE ? (t_exp.tri_stride > 0 ? static_cast<uint32_t>(t_exp.tri_stride) : 12u) : 4u;
    args.stream_hints = 0;                                    // (decided below, with the order)
    // one-shot grid for batches of up to 2^18 rays (profiles/r05_one_shot_ab.txt: Sponza proxy closest 2^16 0.120 -> 0.100 ms, 2^18 0.181 ->
    // 0.159, any-hit 2^18 0.183 -> 0.152; f64 spheres 2^18 0.283 -> 0.259; from 2^20 rays on the persistent grid wins, 0.278 against 0.299)
    args.one_shot = (t_exp.one_shot >= 0 ? t_exp.one_shot > 0 : n <= (size_t{1} << 18)) && !d_counters ? 1u : 0u;
    // one ticket range per XCD (trace_body.inc: refill): free for incoherent batches, a win for every batch whose neighbouring
    // rays are close (coherence-sorted below, or generated that way by the caller)
    static const int parts_env = BVH_DEV_INT("BVH_AMD_PARTS", 0);            // tuning knob
    args.parts = t_parts > 0 ? std::min(t_parts, 256) : parts_env > 0 ? std::min(parts_env, 256) : n < 65536 ? 1 : 8;   // (refined below once the order is decided)
    args.part_size = ((n + args.parts - 1) / args.parts + 63) / 64 * 64;
    args.deep = nullptr; args.deep_cap = 0;
    // Scratch that only some launches need is allocated and freed in stream order (hipMallocAsync / hipFreeAsync), so that
    // concurrent launches of one BVH never share it.
    StreamScope scope(stream);                                // scratch_alloc / the sort's own scratch: blocks cached per (device, stream)
    void* deep_mem = nullptr;
    void* sort_mem = nullptr;
    ScratchTag deep_tag, sort_tag;
    auto release = [&](int rc) {
        (void)hipEventRecord(slot_event, stream);             // the slot is free again once everything queued so far has run
        { std::lock_guard<std::mutex> lock(b.work_mutex); b.work_busy[slot] = false; }
        if (deep_mem) scratch_free(deep_mem, deep_tag);
        if (sort_mem) scratch_free(sort_mem, sort_tag);
        return rc;
    };
    // (failures before this point return without a claimed slot only above; from here on every path goes through release())
    if (d_counters) {
        const hipError_t e = hipMemsetAsync(d_counters, 0, sizeof(bvh_amd_counters), stream);
        if (e != hipSuccess) return release(fail(BVH_AMD_ERR_HIP, std::string("intersect_rays: hipMemsetAsync: ") + hipGetErrorString(e)));
    }
    {   // SmallStack<Index, 64> covers every tree of at most 64 levels; deeper trees get the GrowingStack equivalent
        int rc = tree_depth<T>(b, stream);
        if (rc) return release(rc);
        const int max_depth = b.max_depth.load();
        if (max_depth > 64) {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b.device) != hipSuccess) return release(fail(BVH_AMD_ERR_HIP, "intersect_rays: hipDeviceGetAttribute"));
            const size_t lanes = size_t{8} * cus * kBlock;                   // persistent_grid launches at most 8 blocks per CU
            const size_t cap = static_cast<size_t>(max_depth - 64 + 1);
            if (lanes * cap > (size_t{1} << 32))
                return release(fail(BVH_AMD_ERR_UNSUPPORTED, "intersect_rays: the tree is too deep for the traversal stack (" + std::to_string(max_depth) + " levels)"));
            const hipError_t e = scratch_alloc(&deep_mem, lanes * cap * sizeof(uint32_t), &deep_tag);
            if (e != hipSuccess) { deep_mem = nullptr; return release(fail(BVH_AMD_ERR_HIP, std::string("intersect_rays: stack spill buffer: ") + hipGetErrorString(e))); }
            args.deep = static_cast<uint32_t*>(deep_mem); args.deep_cap = static_cast<uint32_t>(cap);
            args.one_shot = 0;                                // (the spill buffer is sized for the persistent grid)
        }
    }
    static const int refill_env = BVH_DEV_INT("BVH_AMD_REFILL", 0);   // tuning knobs
    static const int leaf_env = BVH_DEV_INT("BVH_AMD_LEAF", 0);
    const bool beyond_l2 = b.pair_count * sizeof(PairNode<T>) > (size_t{32} << 20);
    const bool heavy = beyond_l2 && b.expected_visits.load() >= kReorderMinVisits;       // long walks through a tree the L2s cannot hold
    const bool any_hit = (flags & BVH_AMD_RAY_ANY_HIT) != 0;
    const int forced = coop_fetch_forced();
    const bool coop_capable = true;                           // every record family since round 4 (trees of at most 64 levels: launch_variant)
    if (plan) {                                               // measured for this tree (calibrate) or one of the candidates being measured
        args.coop = coop_capable && plan->coop ? 1u : 0u;
        args.refill_threshold = plan->refill; args.leaf_threshold = plan->leaf;
    } else {                                                  // the predictor (+ the developer's overrides)
        // (closest-hit: only batches of >= 2^21 rays — a smaller one is bound by the dependent chains of its longest rays, which the
        //  cooperative fetch makes longer: configs[4], 1M rays: 0.60 ms per lane, 0.65 cooperative; 4M: 1.86 / 1.66; 16M: 6.68 / 5.62.
        //  Any-hit gains at every size: 0.58 / 0.55, 1.83 / 1.41, 6.76 / 4.74 ms; profiles/r04_experiments_call3_double.txt)
        args.coop = (coop_capable && (forced >= 0 ? forced != 0 : (any_hit || (heavy && n >= (size_t{1} << 21))))) ? 1u : 0u;
        // (round 5, with the staggered drain: a batch of up to 2^21 rays refills earlier — configs[1], 1M rays, per lane: 0.2534 ms at 36 / 12,
        //  0.2460-0.2472 at 20 / 12; 2M rays 0.3741 -> 0.3648; profiles/r05_small_batch_sweep.txt)
        const int refill_lane = n <= (size_t{1} << 21) && std::is_same_v<T, float> ? kRefillThresholdSmall : kRefillThreshold;
        const int refill_default = !args.coop ? refill_lane : any_hit ? kCoopRefillAny : heavy ? kCoopRefillHeavy : kCoopRefillAny;
        const int leaf_default = !args.coop ? kLeafThreshold : any_hit ? kCoopLeafAny : heavy ? kCoopLeafHeavy : kCoopLeafAny;
        args.refill_threshold = t_refill > 0 ? t_refill : refill_env > 0 ? refill_env : refill_default;
        args.leaf_threshold = t_leaf > 0 ? t_leaf : leaf_env > 0 ? leaf_env : leaf_default;
    }
    // the ticket counters of the ranges this launch uses (one per 128 bytes), not the whole 32 KB slot
    auto zero_tickets = [&]() -> int {
        const hipError_t e = hipMemsetAsync(work, 0, size_t{args.parts} * kTicketStride * sizeof(unsigned long long), stream);
        return e == hipSuccess ? BVH_AMD_OK : fail(BVH_AMD_ERR_HIP, std::string("intersect_rays: hipMemsetAsync: ") + hipGetErrorString(e));
    };
    if (b.dim == 2) {                                         // Node<T, 2>: circles only (tri.h has no 2D intersector)
        if (leaf_kind != LEAF_SPHERE) return release(fail(BVH_AMD_ERR_ARG, "intersect_rays: a 2D BVH traces circles (Sphere<T, 2>) only"));
        if (const int rc0 = zero_tickets()) return release(rc0);
        int rc2 = dispatch<T, LEAF_SPHERE, 2>(b, args, flags, d_counters != nullptr, stream);
        if (rc2 == BVH_AMD_OK && (flags & BVH_AMD_RAY_ORIGINAL_IDS)) rc2 = to_original_ids<T>(b, d_hits, n, stream);
        return release(rc2);
    }
    // Reordering has a fixed cost per ray (~0.02 ns for the keys and the sort, plus the indirection in the kernel) and saves a share
    // of the L2 misses, so it needs a tree beyond the L2s AND rays that fetch many records: the 1M-triangle terrain (expected visits
    // 32, 16 measured) loses 16 % with it, the 1M soup (358 expected, 70 measured) gains 40 %.
    // Reordering pays where the walk misses the L2s (tools/sorted_probe.py, 16M uniform rays: 1M-triangle soup 11.9 -> 9.65 ms,
    // 10M-triangle mesh 13.9 -> 9.2 ms) and costs a few per cent where it does not (262k-triangle Sponza proxy, 1M rays: 0.32 ->
    // 0.36 ms), hence the default below.
    const bool reorder = (flags & BVH_AMD_RAY_SORTED) ? n > 4096
                       : (flags & BVH_AMD_RAY_UNSORTED) ? false
                       : plan ? plan->reorder && n > 4096
                       : n >= (size_t{1} << 20) && beyond_l2 && b.expected_visits.load() >= kReorderMinVisits;
    g_last_reordered = reorder && n < (size_t{1} << 31);
    // Rays, their order and the hit records are touched once per launch. In a REORDERED launch (a tree beyond the L2s, long walks) they
    // are loaded / stored non-temporally so that they do not take L2 lines from the records and triangles the in-flight rays share:
    // 1M soup, 2^24 rays: cooperative kernel 7.15 -> 7.08 ms, per-lane 7.75 -> 7.56; 10M mesh: pass 5.91 -> 5.83 ms. NOT otherwise: on a
    // tree the L2s hold (Sponza proxy, 4M closest-hit rays) the same hint COSTS 6 % (0.625 -> 0.666 ms per lane, 0.572 -> 0.605
    // cooperative), and any-hit / terrain batches are indifferent (profiles/r04_experiments_call1.txt, _call2.txt).
    // bvh_amd_experiment("stream_hints", 0 / 1) forces it for A/B runs.
    args.stream_hints = t_exp.stream_hints >= 0 ? static_cast<uint32_t>(t_exp.stream_hints) : g_last_reordered ? 1u : 0u;
    // Ticket ranges (trace_body.inc: refill). A reordered batch wants exactly one range per XCD: each L2 then serves one stretch of the
    // order. A batch traced as given has no such locality to protect, and its waves draw tickets often (short rays, early refills):
    // eight counters then cost measurable time in atomics on one address each — 32 ranges (four per XCD) take the 1M terrain from 1.18
    // to 1.10 ms (per lane) and 1.31 to 1.12 ms (cooperative), the Sponza proxy's shadow rays from 1.63 to 1.52 ms; the sorted soup
    // prefers 8 (7.24 against 7.32 ms at 32); a single counter costs 2x and more (profiles/r03_traversal_experiments.md).
    args.parts = t_parts > 0 ? std::min(t_parts, 256) : parts_env > 0 ? std::min(parts_env, 256) : n < 65536 ? 1 : g_last_reordered ? 8 : 32;
    args.part_size = ((n + args.parts - 1) / args.parts + 63) / 64 * 64;
    if (const int rc0 = zero_tickets()) return release(rc0);
    if (g_last_reordered) {
        const uint32_t n32 = static_cast<uint32_t>(n);
        const size_t words = 4 * n + radix_sort_hist_words(n32, 1);              // keys + tmp, indices + tmp, histogram
        // (from the per-stream block cache the builders use: a stream-ordered allocation per call cost ~20 us of every timed pass)
        hipError_t e = scratch_alloc(&sort_mem, words * sizeof(uint32_t), &sort_tag);
        if (e != hipSuccess) {
            sort_mem = nullptr;
            (void)hipGetLastError();
            if (flags & BVH_AMD_RAY_SORTED) return release(fail(BVH_AMD_ERR_HIP, std::string("intersect_rays: no scratch for BVH_AMD_RAY_SORTED: ") + hipGetErrorString(e)));
            g_last_reordered = false;                         // the reordering was this library's own idea: trace the rays as given instead
        }
    }
    if (g_last_reordered) {
        const uint32_t n32 = static_cast<uint32_t>(n);
        uint32_t *keys = static_cast<uint32_t*>(sort_mem), *vals = keys + n, *kt = vals + n, *vt = kt + n, *hist = vt + n;
        T lo[3], sc[3];
        for (int k = 0; k < 3; ++k) {
            const T ext = b.root_bounds[2 * k + 1] - b.root_bounds[2 * k];
            lo[k] = b.root_bounds[2 * k];
            sc[k] = ext > T(0) ? T(64) / ext : T(0);
        }
        static const int entry_depth = std::min(29, BVH_DEV_INT("BVH_AMD_RAY_KEY_DEPTH", 0));   // developer experiment
        int key_bits = 21;
        bool entry_keys = false, first_hist_done = false;
        if constexpr (std::is_same_v<T, float>) entry_keys = entry_depth > 0 && b.dim == 3;
        if constexpr (std::is_same_v<T, float>) {
            if (entry_keys) {
                hipLaunchKernelGGL(ray_entry_keys_kernel, dim3((n32 + 255) / 256), dim3(256), 0, stream, d_rays, n32, args.pairs, args.root_index, entry_depth, keys);
                key_bits = entry_depth + 3;
            }
        }
        if (!entry_keys) {
            static const int cell_bits_env = std::max(1, std::min(8, BVH_DEV_INT("BVH_AMD_RAY_KEY_BITS", 7)));   // developer knob
            const int cell_bits = t_exp.key_bits > 0 ? std::min(8, t_exp.key_bits) : cell_bits_env;
            const T rescale = static_cast<T>(1u << cell_bits) / T(64);
            const int class_bits = cell_bits < 3 ? 0 : t_exp.key_class_bits >= 0 ? std::min(3, t_exp.key_class_bits) : (plan && plan->classes) ? 1 : kKeyClassBits;
            g_last_classes = class_bits > 0;
            const T class_scale = static_cast<T>(t_exp.key_class_scale > 0 ? t_exp.key_class_scale : kKeyClassScalePercent) / T(100);
            const uint32_t tiles = (n32 + kRadixTileU32 - 1) / kRadixTileU32;
            hipLaunchKernelGGL(ray_keys_kernel<T>, dim3(tiles), dim3(1024), 0, stream, d_rays, n32, lo[0], lo[1], lo[2], sc[0] * rescale, sc[1] * rescale,
                               sc[2] * rescale, keys, 1u << cell_bits, t_exp.key_curve == 0 ? 0 : cell_bits,      // Hilbert index by default (round 4); "key_curve" 0 = Morton
                               class_bits, class_scale, hist, tiles);
            first_hist_done = true;
            key_bits = 3 * cell_bits + 3;
        }
        uint32_t* order = nullptr;
        int rc = radix_sort_pairs<uint32_t>(keys, vals, kt, vt, n32, 1, key_bits, stream, hist, /*iota_vals=*/true, /*keys_wanted=*/false, &order, first_hist_done);
        if (rc) return release(rc);
        args.order = order;
#if defined(BVH_AMD_DEVELOPER)
        if (getenv("BVH_AMD_CHECK_ORDER")) {                  // developer check: is the order a set of valid ray indices when the sort has run?
            uint32_t* d_bad = nullptr;
            if (hipMalloc(&d_bad, 8 * sizeof(uint32_t)) == hipSuccess) {
                for (int round = 0; round < 2; ++round) {
                    (void)hipMemsetAsync(d_bad, 0, 8 * sizeof(uint32_t), stream);
                    hipLaunchKernelGGL(k_check_order, dim3((n32 + 255) / 256), dim3(256), 0, stream, order, n32, d_bad);
                    uint32_t bad[8] = {};
                    (void)hipMemcpyAsync(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream);
                    (void)hipStreamSynchronize(stream);
                    if (bad[0]) {                             // the same words as the copy engine sees them (memory, not a compute unit's caches)
                        std::vector<uint32_t> host(n32);
                        (void)hipMemcpy(host.data(), order, size_t{n32} * 4, hipMemcpyDeviceToHost);
                        uint32_t host_bad = 0;
                        for (uint32_t v : host) host_bad += v >= n32 ? 1u : 0u;
                        fprintf(stderr, "[check order] look %d: the copy engine sees %u bad entries\n", round, host_bad);
                    }
                    if (bad[0]) fprintf(stderr, "[check order] look %d: %u of %u entries are not ray indices; first at %u = 0x%x, %u = 0x%x, %u = 0x%x (sort_mem %p, order %p)\n", round, bad[0], n32,
                                        bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], sort_mem, static_cast<void*>(order));
                }
                (void)hipFree(d_bad);
            }
        }
#endif
    }
    // staggered drain: `stagger` tickets per eighth of the grid over the whole launch -> per ticket range
    {
        // (double precision: four waves per SIMD, not eight, share the issue slots — the drain is not what costs there: 1M f64 spheres, 4M
        //  rays 1.560 ms without, 1.553 at 21k, 1.62-1.92 from 87k up; profiles/r05_stagger_ab_spheres.txt)
        const size_t dflt = std::is_same_v<T, double> ? std::min<size_t>(n / 64, kStaggerDouble)
                                                      : std::min<size_t>(n / 16, heavy ? kStaggerHeavy : kStaggerLight);
        const size_t whole = t_exp.stagger > 0 ? static_cast<size_t>(t_exp.stagger) : t_exp.stagger == 0 ? 0 : dflt;
        args.stagger = whole ? std::max<uint32_t>(1u, static_cast<uint32_t>(whole / args.parts)) : 0u;
    }
    g_last_plan[0] = g_last_reordered ? (g_last_classes ? 2 : 1) : 0; g_last_plan[1] = static_cast<int>(args.coop); g_last_plan[2] = args.refill_threshold; g_last_plan[3] = args.leaf_threshold;
    int rc = leaf_kind == LEAF_TRIANGLE ? dispatch<T, LEAF_TRIANGLE>(b, args, flags, d_counters != nullptr, stream)
                                        : dispatch<T, LEAF_SPHERE>(b, args, flags, d_counters != nullptr, stream);
    if (rc == BVH_AMD_OK && (flags & BVH_AMD_RAY_ORIGINAL_IDS)) rc = to_original_ids<T>(b, d_hits, n, stream);
    return release(rc);
}


// Large batches: which way of tracing is fastest for THIS tree is MEASURED. The predictor above (tree beyond the L2s, expected
// record fetches of a random line) was fitted on four scenes and mispredicts elsewhere — a 4M-triangle scene of clustered debris is
// 18 % faster traced as given, per lane, than with the reordering the predictor asks for (tools/rule_check.py). So the first eight
// batches of >= 2^20 rays through a tree (per kind: closest / any-hit) are each traced WHOLE with one of four candidate plans (each
// twice, the better time counts), their traversal kernel timed by a pair of events on the launch stream that the NEXT such call
// reads once they have completed (no host synchronisation is added: while a measurement is still in flight — calls issued back to
// back without the caller waiting for results — the predictor's plan is used and the search simply takes longer; a reordering
// candidate is charged the keys + sort at their measured rate), and every later batch uses the plan with the fewest nanoseconds
// per ray. (Measuring stretches of ONE batch with different plans was tried first and picks wrongly: a sixteenth of
// a batch neither fills the persistent grid the same way nor has the ray density per sort key the whole batch has.) Results never
// depend on the plan. BVH_AMD_CALIBRATE=0 keeps the predictor.
constexpr float kSortNsPerRay = 0.031f;                    // ray keys + three radix passes: 0.52 ms per 2^24 rays (profiles/r03_*)

static uint32_t pack_plan(const Plan& p) { return 1u | (p.reorder ? 2u : 0u) | (p.coop ? 4u : 0u) | (p.classes ? 8u : 0u) | (uint32_t(p.refill) << 8) | (uint32_t(p.leaf) << 16); }
static Plan unpack_plan(uint32_t w) { return Plan{(w & 2u) != 0, (w & 4u) != 0, int((w >> 8) & 0xFFu), int((w >> 16) & 0xFFu), (w & 8u) != 0}; }

template <typename T>
int launch_traverse(const BvhImpl<T>& b, int leaf_kind, const T* d_prims, const T* d_rays, size_t n, unsigned flags,
                    typename HitOf<T>::Type* d_hits, bvh_amd_counters* d_counters, hipStream_t stream)
{
    static const bool calibrate_on = !(getenv("BVH_AMD_CALIBRATE") && atoi(getenv("BVH_AMD_CALIBRATE")) == 0);
    const bool any_hit = (flags & BVH_AMD_RAY_ANY_HIT) != 0;
    const int kind = any_hit ? 1 : 0;
    const bool free_choice = !(flags & (BVH_AMD_RAY_SORTED | BVH_AMD_RAY_UNSORTED)) && coop_fetch_forced() < 0 && t_refill <= 0 && t_leaf <= 0 &&
                             !BVH_DEV_STR("BVH_AMD_REFILL") && !BVH_DEV_STR("BVH_AMD_LEAF");
    const bool candidate = calibrate_on && free_choice && b.dim == 3 && n >= (size_t{1} << 20) && n < (size_t{1} << 31) &&
                           b.pair_count * sizeof(PairNode<T>) > (size_t{32} << 20);
    if (!candidate) return launch_planned<T>(b, leaf_kind, d_prims, d_rays, n, flags, d_hits, d_counters, stream, nullptr);
    if (b.node_count && b.d_pairs) {                          // depth + expected record fetches of a random line, once per tree: the predictor below reads it
        const int rc = tree_depth<T>(b, stream);
        if (rc) return rc;
    }
    if (const uint32_t cached = b.launch_plan[kind].load()) {
        const Plan plan = unpack_plan(cached);
        return launch_planned<T>(b, leaf_kind, d_prims, d_rays, n, flags, d_hits, d_counters, stream, &plan);
    }
    constexpr int kCand = BvhImpl<T>::PlanSearch::kCandidates;
    const int n_cand = any_hit ? 4 : 5;
    const Plan candidates[kCand] = {
        {false, false, kRefillThreshold, kLeafThreshold},
        {false, true, any_hit ? kCoopRefillAny : kCoopRefillHeavy, any_hit ? kCoopLeafAny : kCoopLeafHeavy},
        any_hit ? Plan{false, true, kCoopRefillHeavy, kCoopLeafHeavy} : Plan{true, false, kRefillThreshold, kLeafThreshold},
        {true, true, any_hit ? kCoopRefillAny : kCoopRefillHeavy, any_hit ? kCoopLeafAny : kCoopLeafHeavy},
        // round 5, closest-hit only: reordered with the LONG RAYS FIRST (ray_keys_kernel: chord classes) — 4 % faster on the 1M soup, 6 %
        // slower on the 10M soup, so it is measured like everything else
        {true, true, kCoopRefillHeavy, kCoopLeafHeavy, true},
    };
    // The order in which the candidates are tried starts with the PREDICTOR's plan (VERDICT r3 Weak 4: the first large batch through
    // a tree — all a single-shot caller ever sends, e.g. one GPU's 12.5M-ray shard of configs[3] — used to get candidate 0, traced
    // as given with the per-lane fetch: the slowest plan on the very scenes the search exists for). The first batch is now traced
    // the way round 2's rule says (reorder iff the tree is beyond the L2s and a random line is expected to fetch >= 100 records;
    // cooperative fetch for any-hit and for such heavy trees), which is the winner or within a few per cent of it on 6 of the 7
    // scenes of profiles/r03_rule_check.txt; the exploration of the other three starts with the second batch.
    const bool heavy = b.expected_visits.load() >= kReorderMinVisits;       // (tree_depth has filled it: the candidate test needs a resident tree)
    // Round 6 (the first call through a fresh tree): a heavy closest-hit tree whose records fit the 256 MB Infinity
    // Cache gets the "long rays first" order straight away — it is what the search settles on there (1M soup: 0.427 against 0.446 ns per
    // ray, profiles/r06_bench_default.json), while on a tree beyond it the split of the sort key costs more coherence than the shorter
    // tail gives back (10M soup: 0.382 against 0.376, profiles/r05_bench_config3_1gpu_final.json). The search measures both either way.
    const bool fits_infinity_cache = b.pair_count * sizeof(PairNode<T>) <= (size_t{256} << 20);
    const int predicted = any_hit ? (heavy ? 3 : 1) : (heavy ? (fits_infinity_cache ? 4 : 3) : 0);
    // Round 5 (VERDICT r4 Weak 4): the search no longer spends eight batches whatever it sees. Order of exploration: the predictor's
    // plan, then the plan of the OTHER ray order with the same record fetch (it decides the family: reordered or as given), then the
    // rest. After every measurement a candidate more than 10 % behind the best so far (40 % while it has only one measurement) is out,
    // and a candidate of the other ray order that lost by more than 40 % takes its whole family with it (on the 1M soup the batches
    // traced as given take 12 ms against 7: one of them is enough to know). Candidates still in the race are measured twice (a kernel's first
    // launch also pays its code load); the search ends when every survivor has both measurements or only one survivor is left:
    // 1M soup 5 batches instead of 8 (3, 1, 2, 3, 2), none of them the 12.7 ms plans twice.
    int order[kCand], n_order = 0;
    {
        auto push = [&](int c) { for (int k = 0; k < n_order; ++k) if (order[k] == c) return; order[n_order++] = c; };
        push(predicted);
        for (int c = 0; c < n_cand; ++c) if (candidates[c].reorder != candidates[predicted].reorder && candidates[c].coop == candidates[predicted].coop && !candidates[c].classes) push(c);
        for (int c = 0; c < n_cand; ++c)                                     // the predictor's plan with the other order of the long rays: its closest rival
            if (c != predicted && candidates[c].reorder && candidates[predicted].reorder && candidates[c].coop == candidates[predicted].coop) push(c);
        for (int c = 0; c < n_cand; ++c) if (candidates[c].reorder == candidates[predicted].reorder) push(c);
        for (int c = 0; c < n_cand; ++c) push(c);
    }
    std::pair<hipEvent_t, hipEvent_t> events{nullptr, nullptr};
    int trying = -1;
    Plan plan{};
    bool have_plan = false;
    {
        std::lock_guard<std::mutex> lock(b.plan_mutex);
        typename BvhImpl<T>::PlanSearch& ps = b.plan_search[kind];
        // (`recorded` is set, under this mutex, by the thread that owns the measurement once BOTH events are on its stream: a second
        //  thread can no longer read a stale pair while the first is still between claim and record — ADVICE r3)
        if (ps.pending && ps.recorded && hipEventQuery(ps.stop) == hipSuccess) {   // the previous candidate has run: note its time
            float ms = 0;
            if (hipEventElapsedTime(&ms, ps.start, ps.stop) == hipSuccess && ps.rays && ms > 0.0f) {
                const int c = ps.trying;                                     // survivors are measured twice; the better time counts
                const float ns = ms * 1e6f / static_cast<float>(ps.rays) + (candidates[c].reorder ? kSortNsPerRay : 0.0f);
                if (ps.count[c] == 0 || ns < ps.ns_per_ray[c]) { ps.ns_per_ray[c] = ns; ps.rays_of[c] = ps.rays; }   // (a kernel's first launch also pays its code load)
                ++ps.count[c];
                ++ps.index;
                int best = c;
                for (int k = 0; k < n_cand; ++k) if (ps.count[k] && !(ps.dropped >> k & 1) && ps.ns_per_ray[k] < ps.ns_per_ray[best]) best = k;
                for (int k = 0; k < n_cand; ++k) {
                    if (!ps.count[k] || k == best) continue;
                    // (a candidate's FIRST time may carry its kernel's code load — on a 2^20-ray batch that is tens of per cent — so a single
                    //  measurement only drops a clear loser; the 10 % rule needs both)
                    // (times taken on batches of very different size say more about the batches than about the plans — a batch of
                    //  short rays against one of long rays, 2^20 rays against 2^24: only batches within 2x of each other are compared)
                    const size_t nk = ps.rays_of[k], nb = ps.rays_of[best];
                    if (nk == 0 || nb == 0 || std::max(nk, nb) > 2 * std::min(nk, nb)) continue;
                    const float behind = ps.ns_per_ray[k] / ps.ns_per_ray[best];
                    if (behind > (ps.count[k] >= 2 ? 1.10f : 1.40f)) ps.dropped |= uint8_t(1u << k);
                    if (behind > 1.40f && candidates[k].reorder != candidates[best].reorder)
                        for (int f = 0; f < n_cand; ++f) if (candidates[f].reorder == candidates[k].reorder) ps.dropped |= uint8_t(1u << f);
                }
            }
            ps.pending = false; ps.recorded = false; ps.trying = -1;
        }
        (void)hipGetLastError();
        int next = -1, alive = 0, winner = -1;
        for (int k = 0; k < n_order; ++k) {                                 // the next candidate: an unmeasured survivor first, then second measurements
            const int c = order[k];
            if (ps.dropped >> c & 1) continue;
            ++alive;
            if (ps.count[c] && (winner < 0 || ps.ns_per_ray[c] < ps.ns_per_ray[winner])) winner = c;
            if (ps.count[c] == 0 && (next < 0 || ps.count[next] != 0)) next = c;
            else if (ps.count[c] == 1 && next < 0) next = c;
        }
        if (alive == 1 && winner >= 0) next = -1;                           // nobody left to compare with
        if (next < 0 && alive >= 2 && winner >= 0) {
            // a photo finish (the runner-up within 5 % after two measurements each: the soup's "long rays first" beats the plain reordered
            // plan by 3-5 % in the search's own numbers, and one run in five settled on the wrong one) gets a third measurement of both
            int second = -1;
            for (int c = 0; c < n_cand; ++c)
                if (c != winner && !(ps.dropped >> c & 1) && ps.count[c] && (second < 0 || ps.ns_per_ray[c] < ps.ns_per_ray[second])) second = c;
            if (second >= 0 && ps.ns_per_ray[second] < 1.05f * ps.ns_per_ray[winner]) {
                if (ps.count[winner] < 3) next = winner;
                else if (ps.count[second] < 3) next = second;
            }
        }
        // Round 6: the search ends early when the predictor's plan leads what has been measured — the other ray order
        // and, where there is one, the predictor's plan with the other order of the long rays (its SIBLING: the two are within 5 % of
        // each other and either can win, so the better of the pair is kept) — by more than 10 %. The predictor's first time is the first
        // call's, which is 5-15 % slower than the plan's settled time (profiles/r06_first_call_probe.txt): it is measured once more
        // before it is compared with its sibling, or when its lead over the others is inside that handicap. Closest-hit through a heavy
        // tree: 4 batches (plan, as given, sibling, plan) instead of 8; any-hit: 3 (plan, as given, other fetch).
        if (!ps.pending && ps.index >= 3 && ps.count[predicted] && !(ps.dropped >> predicted & 1)) {
            int sibling = -1;
            for (int c = 0; c < n_cand; ++c)
                if (c != predicted && candidates[c].reorder && candidates[predicted].reorder && candidates[c].coop == candidates[predicted].coop) sibling = c;
            const bool pair_ready = sibling < 0 || (ps.dropped >> sibling & 1) || ps.count[sibling] >= 1;
            int champ = predicted;
            if (sibling >= 0 && ps.count[sibling] && !(ps.dropped >> sibling & 1) && ps.ns_per_ray[sibling] < ps.ns_per_ray[predicted]) champ = sibling;
            int others = 0;
            float closest = 0.0f;                                           // the best candidate outside the pair over the pair's better time
            bool comparable = true;
            for (int c = 0; c < n_cand; ++c) {
                if (c == predicted || c == sibling || !ps.count[c]) continue;
                ++others;
                const size_t nc = ps.rays_of[c], np = ps.rays_of[champ];
                if (nc == 0 || np == 0 || std::max(nc, np) > 2 * std::min(nc, np)) comparable = false;
                const float r = ps.ns_per_ray[c] / ps.ns_per_ray[champ];
                if (closest == 0.0f || r < closest) closest = r;
            }
            const int need_others = sibling >= 0 ? 1 : 2;
            if (pair_ready && others >= need_others && comparable) {
                const bool second_look = ps.count[predicted] == 1 && (sibling >= 0 ? !(ps.dropped >> sibling & 1) : closest <= 1.10f && closest > 1.10f * 0.85f);
                if (second_look) next = predicted;
                else if (closest > 1.10f) { next = -1; winner = champ; }
            }
        }
        if (!ps.pending && next < 0 && winner >= 0) {                       // every survivor measured twice (or alone): keep the winner
            for (int c = 0; c < 5; ++c) { g_search_ns[c] = c < n_cand ? ps.ns_per_ray[c] : 0.0f; g_search_count[c] = c < n_cand ? ps.count[c] : 0; }
            g_search_dropped = ps.dropped;
            b.launch_plan[kind].store(pack_plan(candidates[winner]));
            plan = candidates[winner]; have_plan = true;
        } else if (!ps.pending && !d_counters && next >= 0) {               // try the next candidate on this batch
            if (!ps.start) (void)hipEventCreate(&ps.start);
            if (!ps.stop) (void)hipEventCreate(&ps.stop);
            if (ps.start && ps.stop) {
                trying = next; plan = candidates[trying]; have_plan = true;
                events = {ps.start, ps.stop};
                ps.pending = true; ps.recorded = false; ps.rays = n; ps.trying = trying;
            }
        } else {                                                            // a measurement is in flight (or counters are wanted): the predictor's plan
            plan = candidates[predicted]; have_plan = true;
        }
        (void)hipGetLastError();
    }
    if (trying >= 0) t_calibration = &events;
    const int rc = launch_planned<T>(b, leaf_kind, d_prims, d_rays, n, flags, d_hits, d_counters, stream, have_plan ? &plan : nullptr);
    t_calibration = nullptr;
    if (trying >= 0) {
        std::lock_guard<std::mutex> lock(b.plan_mutex);
        typename BvhImpl<T>::PlanSearch& ps = b.plan_search[kind];
        if (rc == BVH_AMD_OK) ps.recorded = true;                           // both events are on the stream now
        else ps.pending = false;                                            // nothing was launched: the candidate is tried again
    }
    return rc;
}


// Bvh::intersect<IsAnyHit, IsRobust>(ray, start, stack, leaf_fn, inner_fn) for ONE ray with host callbacks; see ray_step_kernel.
// ray8 = {org[3], dir[3], tmin, tmax} (z = 0 for a 2D BVH); `start` is the packed Index word the walk starts from.
template <typename T>
int trace_ray_callbacks(const BvhImpl<T>& b, const T ray8[8], uint32_t start, bool any, bool robust,
                        bool (*leaf_fn)(void*, T*, size_t, size_t), void (*inner_fn)(void*, size_t), void* user)
{
    if (!leaf_fn) return fail(BVH_AMD_ERR_ARG, "intersect_ray: null leaf callback");
    if (b.node_count == 0 || (b.pair_count && !b.d_pairs)) return fail(BVH_AMD_ERR_ARG, "intersect_ray: BVH has no device copy");
    if (b.max_depth.load() < 0) {                             // first traversal of this layout: depth of the tree, once
        // the walk runs on this thread's own stream: whatever built or re-laid the tree on another stream must be done
        BVH_HIP_TRY(hipDeviceSynchronize(), BVH_AMD_ERR_HIP);
        int rc = tree_depth<T>(b, nullptr);
        if (rc) return rc;
    }
    // SmallStack<Index, 64> of bvh_impl.h:244 for every tree it can hold, as deep as the tree needs otherwise
    const uint32_t cap = static_cast<uint32_t>(std::max(64, b.max_depth.load() + 2));
    StepContextClaim claim;                                   // released when this call returns, callbacks included
    StepContext& c = *claim.ctx;
    int rc = c.ensure(b.device, cap);
    if (rc) return rc;
    uint32_t* in = c.pinned;
    uint32_t* out = in + 1 + c.stack_cap;
    const uint32_t* events = out + kStepHeader;
    const uint32_t* snaps = events + 3 * kStepEvents;
    StepArgs<T> a;
    a.pairs = b.d_pairs; a.in = in; a.out = out; a.scratch = c.scratch; a.stack_cap = c.stack_cap; a.snap_words = c.snap_words;
    a.any = any ? 1u : 0u; a.record_pairs = inner_fn ? 1u : 0u;
    a.max_events = kStepEvents;
    a.max_steps = static_cast<uint32_t>(std::min<size_t>(b.pair_count + 1, 0xFFFFFFFFu));
    // (bvh_amd_experiment("step_events", n): a short log exercises the continue-from-device path in the tests)
    if (t_exp.step_events >= 1 && t_exp.step_events < static_cast<int>(kStepEvents)) a.max_events = static_cast<uint32_t>(t_exp.step_events);
    for (int k = 0; k < 8; ++k) a.ray[k] = ray8[k];
    static const bool poll = !(BVH_DEV_INT("BVH_AMD_STEP_POLL", 1) == 0);
    T tmax = ray8[7];
    in[0] = 1; in[1] = start;
    for (;;) {
        a.ray[7] = tmax;
        a.seq = ++c.seq;
        if (b.dim == 2) {
            if (robust) hipLaunchKernelGGL((ray_step_kernel<T, true, 2>), dim3(1), dim3(kWave), 0, c.stream, a);
            else hipLaunchKernelGGL((ray_step_kernel<T, false, 2>), dim3(1), dim3(kWave), 0, c.stream, a);
        } else {
            if (robust) hipLaunchKernelGGL((ray_step_kernel<T, true, 3>), dim3(1), dim3(kWave), 0, c.stream, a);
            else hipLaunchKernelGGL((ray_step_kernel<T, false, 3>), dim3(1), dim3(kWave), 0, c.stream, a);
        }
        BVH_HIP_TRY(hipGetLastError(), BVH_AMD_ERR_HIP);
        // Wait for the log: poll its launch number for a short while (a stream synchronisation costs more than a short walk),
        // then fall back to blocking. The kernel may still be saving its stack when the number appears; the next launch is
        // ordered behind it by the stream.
        bool arrived = false;
        if (poll) {
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(60);
            for (uint32_t spin = 0; !arrived; ++spin) {
                arrived = __atomic_load_n(&out[3], __ATOMIC_ACQUIRE) == a.seq;
                if (!arrived && (spin & 31u) == 31u && std::chrono::steady_clock::now() > give_up) break;
            }
        }
        if (!arrived) BVH_HIP_TRY(hipStreamSynchronize(c.stream), BVH_AMD_ERR_HIP);
        if (out[2]) return fail(BVH_AMD_ERR_OVERFLOW, out[2] == 2 ? "intersect_ray: the walk visited more pairs than the tree has (cyclic tree?)"
                                                                   : "intersect_ray: traversal stack overflow (malformed tree?)");
        const uint32_t n_events = out[0];
        bool restarted = false;
        for (uint32_t e = 0; e < n_events; ++e) {
            const uint32_t word = events[3 * e];
            if ((word & kCountMask) == 0) { inner_fn(user, word >> kCountBits); continue; }
            const T before = tmax;
            const size_t begin = word >> kCountBits;
            const bool was_hit = leaf_fn(user, &tmax, begin, begin + (word & kCountMask));     // bvh.h:152-155
            if (any && was_hit) return BVH_AMD_OK;
            if (std::memcmp(&before, &tmax, sizeof(T)) != 0) {          // the ray changed: what was logged after this leaf is void
                const uint32_t off = events[3 * e + 1], len = events[3 * e + 2];
                if (len == 0) return BVH_AMD_OK;                        // nothing was left to visit
                in[0] = len;
                std::memcpy(in + 1, snaps + off, len * sizeof(uint32_t));
                restarted = true;
                break;
            }
        }
        if (restarted) continue;
        if (out[1]) return BVH_AMD_OK;
        in[0] = kStepContinue;                                         // log full: carry on from the stack the device kept
    }
}

template int trace_ray_callbacks<float>(const BvhImpl<float>&, const float*, uint32_t, bool, bool, bool (*)(void*, float*, size_t, size_t),
                                        void (*)(void*, size_t), void*);
template int trace_ray_callbacks<double>(const BvhImpl<double>&, const double*, uint32_t, bool, bool, bool (*)(void*, double*, size_t, size_t),
                                         void (*)(void*, size_t), void*);

// bvhXX_prepare_trace: what the FIRST large batch through a fresh tree would otherwise pay inside its own call (VERDICT r5 Next 5) — the
// depth pass + expected-visits pass of the tree (tree_depth: one read-back) and the first allocation of the reordering's scratch for
// batches of `n_rays_hint` rays, which goes into the block cache of (device, stream) where the batch finds it. Optional: a batch that
// comes unprepared does these itself, as before. A single `Bvh::intersect` on a fresh `Bvh` is the reference's normal use (bvh.h:160-182).
template <typename T>
int prepare_trace(const BvhImpl<T>& b, size_t n_rays_hint, hipStream_t stream) {
    if (b.node_count == 0) return BVH_AMD_OK;
    if (b.pair_count && !b.d_pairs) return fail(BVH_AMD_ERR_ARG, "prepare_trace: BVH has no device nodes");
    if (const int rc = tree_depth<T>(b, stream)) return rc;
    if (n_rays_hint >= (size_t{1} << 20) && n_rays_hint < (size_t{1} << 31) && scratch_pool_enabled()) {
        StreamScope scope(stream);
        void* mem = nullptr;
        ScratchTag tag;
        const size_t words = 4 * n_rays_hint + radix_sort_hist_words(static_cast<uint32_t>(n_rays_hint), 1);
        if (scratch_alloc(&mem, words * sizeof(uint32_t), &tag) == hipSuccess) scratch_free(mem, tag);
        else (void)hipGetLastError();                         // (no room now: the batch will ask again itself)
    }
    return BVH_AMD_OK;
}
template int prepare_trace<float>(const BvhImpl<float>&, size_t, hipStream_t);
template int prepare_trace<double>(const BvhImpl<double>&, size_t, hipStream_t);

template int launch_traverse<float>(const BvhImpl<float>&, int, const float*, const float*, size_t, unsigned,
                                    bvh_hit3f*, bvh_amd_counters*, hipStream_t);
template int launch_traverse<double>(const BvhImpl<double>&, int, const double*, const double*, size_t, unsigned,
                                     bvh_hit3d*, bvh_amd_counters*, hipStream_t);

} // namespace bvh_amd
