// Measurement aids (not on the product path; called by bench.py and tools/): the rates at which this GPU's memory system
// serves the traversal kernel's access pattern with NOTHING else in the way — every participating lane walks a chain of
// 64-byte records through a table (the next record's index comes out of the record just loaded, exactly one record in flight
// per chain, like a tree walk). The table's size selects the level that serves it: a few KB -> the CU's L1 (TCP), a few MB ->
// the XCD's L2 (TCC), beyond 32 MB -> the fabric / Infinity Cache / HBM. bench.py prices the traversal kernel's measured
// L1 accesses, L2 hits and L2 misses against these rates ("roofline": memory-hierarchy bound).
//
// Modes (how a record reaches its lane):
//   0  per lane: four global_load_dwordx4 of the lane's own record (what trace_kernel does)
//   1  quad-cooperative + LDS transpose: in instruction k the four lanes of a quad load the four 16-byte chunks of the record of
//      the quad's lane k (one 64-byte line per quad and instruction), the chunks are parked in LDS in lane order and every lane
//      reads its own record back with four ds_read_b128 (LDS-bound: ~90 clk per wave-step whatever the number of active lanes)
//   2  quad-cooperative loads alone: one chain per QUAD, its four lanes load the four chunks of the chain's record with one
//      instruction (isolates what a quad-coalesced line costs the L1)
//   3  the same 16 chains per wave as mode 2, walked by lane 0 of each quad alone with four loads (the per-lane cost of mode 2's work)
//   5 / 6  128-byte records, one chain per octet of lanes: the octet loads the line with one instruction / lane 0 alone with eight (n_records counts 128-byte records)
//   4  quad-cooperative + in-register transpose (DPP quad_perm butterflies, no LDS): trace_device.h coop_load_pair itself, what
//      trace_kernel<..., Coop = true> does
//   7 / 8  per-lane loads with 2 / 4 INDEPENDENT chains per lane (2 / 4 records in flight per lane); 9: two chains per lane through the
//      cooperative fetch — do the ceilings of modes 0 / 4 belong to the memory system or to one-fetch-in-flight-per-lane?
// `active` (1..64): lanes of a wave that own a chain (scattered over the wave: lane l is active iff (37 l mod 64) < active); the
// others idle in mode 0 and only help loading in mode 1.
#include "common.h"
#include "trace_device.h"

namespace bvh_amd {

namespace {

struct WalkRec { uint32_t w[16]; };

__device__ inline bool lane_active(uint32_t lane, uint32_t active) { return ((lane * 37u) & 63u) < active; }

__global__ void __launch_bounds__(256) k_record_walk(const WalkRec* table, uint32_t n_rec, uint32_t steps, uint32_t active, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (!lane_active(threadIdx.x & 63u, active)) return;
    uint32_t idx = static_cast<uint32_t>((gid * 2654435761ull) % n_rec);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint4* q = reinterpret_cast<const uint4*>(table + idx);
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc += a.y + b.z + c.w + d.y;
        idx = a.x < n_rec ? a.x : 0u;                          // w[0] = the next record of the chain
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);             // keeps the loads alive
}

// modes 7 / 8 (K = 2 / 4): K INDEPENDENT chains per lane, the K records of a step in flight together (VERDICT r3 "Next 5": is the
// records/s ceiling of modes 0 / 4 a property of the memory system, or of a probe that keeps ONE dependent fetch in flight per lane?
// If the ceiling is a count of outstanding requests per CU, more chains per lane change nothing; if it is latency x lanes, the rate
// scales with K). Per-lane loads, 4 x global_load_dwordx4 per record.
template <int K>
__global__ void __launch_bounds__(256) k_record_walk_mlp(const WalkRec* table, uint32_t n_rec, uint32_t steps, uint32_t active, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (!lane_active(threadIdx.x & 63u, active)) return;
    uint32_t idx[K];
#pragma unroll
    for (int k = 0; k < K; ++k) idx[k] = static_cast<uint32_t>(((gid * K + k) * 2654435761ull) % n_rec);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        uint4 a[K], b[K], c[K], d[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {                          // all K x 4 loads are issued before any of them is consumed
            const uint4* q = reinterpret_cast<const uint4*>(table + idx[k]);
            a[k] = q[0]; b[k] = q[1]; c[k] = q[2]; d[k] = q[3];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc += a[k].y + b[k].z + c[k].w + d[k].y;
            idx[k] = a[k].x < n_rec ? a[k].x : 0u;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

// mode 9: two independent chains per lane through the quad-cooperative fetch + in-register transpose (two coop_load_pair in flight)
__global__ void __launch_bounds__(256) k_record_walk_dpp2(const WalkRec* table, uint32_t n_rec, uint32_t steps, uint32_t active, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u;
    const bool mine = lane_active(lane, active);
    uint32_t i0 = static_cast<uint32_t>(((gid * 2u) * 2654435761ull) % n_rec), i1 = static_cast<uint32_t>(((gid * 2u + 1u) * 2654435761ull) % n_rec);
    float acc = 0.0f;
    for (uint32_t it = 0; it < steps; ++it) {
        float lb0[6], rb0[6], lb1[6], rb1[6];
        uint32_t li0 = 0, ri0 = 0, li1 = 0, ri1 = 0;
        coop_load_pair(reinterpret_cast<const PairNode<float>*>(table), mine ? i0 : 0xFFFFFFFFu, static_cast<int>(lane), lb0, rb0, li0, ri0);
        coop_load_pair(reinterpret_cast<const PairNode<float>*>(table), mine ? i1 : 0xFFFFFFFFu, static_cast<int>(lane), lb1, rb1, li1, ri1);
        if (mine) {
            acc += ((lb0[1] + lb0[2]) + (lb0[3] + lb0[4])) + ((lb0[5] + rb0[0]) + (rb0[1] + rb0[2])) + ((rb0[3] + rb0[4]) + rb0[5]);
            acc += ((lb1[1] + lb1[2]) + (lb1[3] + lb1[4])) + ((lb1[5] + rb1[0]) + (rb1[1] + rb1[2])) + ((rb1[3] + rb1[4]) + rb1[5]);
            acc += __uint_as_float((li0 ^ ri0 ^ li1 ^ ri1) & 0xFFu);
            const uint32_t n0 = __float_as_uint(lb0[0]), n1 = __float_as_uint(lb1[0]);
            i0 = n0 < n_rec ? n0 : 0u; i1 = n1 < n_rec ? n1 : 0u;
        }
    }
    if (acc == 1234.5f) atomicAdd(sink, 1ull);
}

// modes 5 / 6: 128-byte records (one full L1 / L2 line): one chain per OCTET of lanes, its 8 lanes load the record's eight 16-byte chunks
// with one instruction (5), or lane 0 of each octet alone with eight loads (6). Does a line-granular miss cost the memory system the
// same whether 64 or 128 of its bytes are wanted? `table` holds n_rec records of 128 bytes, word 0 = the next record.
template <bool Coop>
__global__ void __launch_bounds__(256) k_record_walk_wide(const uint4* table, uint32_t n_rec, uint32_t steps, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, j = lane & 7u;
    uint32_t idx = static_cast<uint32_t>(((gid >> 3) * 2654435761ull) % n_rec);
    uint32_t acc = 0;
    if (!Coop && j != 0) return;
    for (uint32_t it = 0; it < steps; ++it) {
        uint32_t next;
        if (Coop) {
            const uint4 v = table[size_t{idx} * 8 + j];
            acc += v.y + v.w;
            next = static_cast<uint32_t>(__shfl(static_cast<int>(v.x), static_cast<int>(lane & ~7u)));
        } else {
            const uint4* q = table + size_t{idx} * 8;
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
            acc += a.y + b.z + c.w + d.y + e.z + f.w + g.y + h.z;
            next = a.x;
        }
        idx = next < n_rec ? next : 0u;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

template <int K> __device__ inline uint32_t quad_broadcast(uint32_t v) {   // value of lane K of the caller's quad (DPP quad_perm, no LDS)
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), K * 0x55, 0xF, 0xF, false));
}

__global__ void __launch_bounds__(256) k_record_walk_coop(const WalkRec* table, uint32_t n_rec, uint32_t steps, uint32_t active, unsigned long long* sink) {
    __shared__ uint4 stage[4][4][64];                          // [wave][instruction k][loading lane]: 16 KB per block
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, j = lane & 3u, quad = lane >> 2;
    const bool mine = lane_active(lane, active);
    uint32_t idx = static_cast<uint32_t>((gid * 2654435761ull) % n_rec);
    uint32_t acc = 0;
    const uint4* base = reinterpret_cast<const uint4*>(table);
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t want = mine ? idx : 0xFFFFFFFFu;
        const uint32_t o0 = quad_broadcast<0>(want), o1 = quad_broadcast<1>(want), o2 = quad_broadcast<2>(want), o3 = quad_broadcast<3>(want);
        // lane j loads chunk (j - k) & 3 of owner k's record: the quad covers the record's 64 bytes with one instruction, and
        // the rotation makes the owner's four ds_read_b128 below hit four different bank groups within its quad
        uint4 v0 = {}, v1 = {}, v2 = {}, v3 = {};
        if (o0 != 0xFFFFFFFFu) v0 = base[4ull * o0 + ((j + 0u) & 3u)];
        if (o1 != 0xFFFFFFFFu) v1 = base[4ull * o1 + ((j + 3u) & 3u)];
        if (o2 != 0xFFFFFFFFu) v2 = base[4ull * o2 + ((j + 2u) & 3u)];
        if (o3 != 0xFFFFFFFFu) v3 = base[4ull * o3 + ((j + 1u) & 3u)];
        stage[wave][0][lane] = v0; stage[wave][1][lane] = v1; stage[wave][2][lane] = v2; stage[wave][3][lane] = v3;
        __builtin_amdgcn_wave_barrier();                       // (one wave: its LDS operations complete in order)
        if (mine) {
            const uint4* rec = &stage[wave][j][4u * quad];
            const uint4 a = rec[(0u + j) & 3u], b = rec[(1u + j) & 3u], c = rec[(2u + j) & 3u], d = rec[(3u + j) & 3u];
            acc += a.y + b.z + c.w + d.y;
            idx = a.x < n_rec ? a.x : 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

__global__ void __launch_bounds__(256) k_record_walk_dpp(const WalkRec* table, uint32_t n_rec, uint32_t steps, uint32_t active, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u;
    const bool mine = lane_active(lane, active);
    uint32_t idx = static_cast<uint32_t>((gid * 2654435761ull) % n_rec);
    float acc = 0.0f;
    for (uint32_t it = 0; it < steps; ++it) {
        float lb[6], rb[6];
        uint32_t li = 0, ri = 0;
        coop_load_pair(reinterpret_cast<const PairNode<float>*>(table), mine ? idx : 0xFFFFFFFFu, static_cast<int>(lane), lb, rb, li, ri);
        if (mine) {
            acc += ((lb[1] + lb[2]) + (lb[3] + lb[4])) + ((lb[5] + rb[0]) + (rb[1] + rb[2])) + ((rb[3] + rb[4]) + rb[5]);   // every word is used
            const uint32_t next = __float_as_uint(lb[0]);     // w[0]
            idx = next < n_rec ? next : 0u;
            acc += __uint_as_float((li ^ ri) & 0xFFu);
        }
    }
    if (acc == 1234.5f) atomicAdd(sink, 1ull);
}

template <bool Coop>
__global__ void __launch_bounds__(256) k_record_walk_quad(const WalkRec* table, uint32_t n_rec, uint32_t steps, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, j = threadIdx.x & 3u;
    uint32_t idx = static_cast<uint32_t>(((gid >> 2) * 2654435761ull) % n_rec);        // one chain per quad
    uint32_t acc = 0;
    const uint4* base = reinterpret_cast<const uint4*>(table);
    if (Coop) {
        for (uint32_t it = 0; it < steps; ++it) {
            const uint4 v = base[4ull * idx + j];
            acc += v.y;
            const uint32_t next = quad_broadcast<0>(v.x);      // w[0] sits in chunk 0 = lane 0 of the quad
            idx = next < n_rec ? next : 0u;
        }
    } else {
        if (j != 0) return;
        for (uint32_t it = 0; it < steps; ++it) {
            const uint4* q = base + 4ull * idx;
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
            acc += a.y + b.z + c.w + d.y;
            idx = a.x < n_rec ? a.x : 0u;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

// Two dependent walks braided into ONE chain per lane: even steps fetch the next record of a SMALL table (resident in the L2s), odd
// steps the next record of a BIG one (beyond them); each index is made to depend on the record fetched just before (a term that is
// always zero, unknown to the compiler), so a lane has exactly one fetch in flight and alternates L2 hit, L2 miss. If the two levels
// were independent resources the pair of steps would run at the pace of the slower one's THROUGHPUT (the fabric's 57 G misses/s -> 114 G
// steps/s); if they share one — the CU's outstanding lines — the times add (2 / (1 / R_L2 + 1 / R_fabric)).
template <bool Coop>
__global__ void __launch_bounds__(256) k_record_walk_mixed(const WalkRec* small_t, uint32_t n_small, const WalkRec* big_t, uint32_t n_big, uint32_t steps,
                                                           unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t is = static_cast<uint32_t>((gid * 2654435761ull) % n_small), ib = static_cast<uint32_t>((gid * 2246822519ull) % n_big);
    float acc = 0.0f;
    uint32_t carry = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        const bool small_step = (it & 1u) == 0;
        const WalkRec* t = small_step ? small_t : big_t;
        const uint32_t idx = (small_step ? is : ib) + carry;   // carry == 0, but only the previous fetch says so
        float lb[6], rb[6];
        uint32_t li = 0, ri = 0;
        if (Coop) coop_load_pair(reinterpret_cast<const PairNode<float>*>(t), idx, static_cast<int>(lane), lb, rb, li, ri);
        else load_pair(reinterpret_cast<const PairNode<float>*>(t) + idx, lb, rb, li, ri);
        acc += ((lb[1] + lb[2]) + (lb[3] + lb[4])) + ((lb[5] + rb[0]) + (rb[1] + rb[2])) + ((rb[3] + rb[4]) + rb[5]);
        const uint32_t next = __float_as_uint(lb[0]);
        carry = next >> 31;                                    // indices are below 2^31
        if (small_step) is = next < n_small ? next : 0u; else ib = next < n_big ? next : 0u;
        acc += __uint_as_float((li ^ ri) & 0xFFu);
    }
    if (acc == 1234.5f) atomicAdd(sink, 1ull);
}

} // namespace

} // namespace bvh_amd

extern "C" {

// The braided walk above: `steps` fetches per lane, alternating between the two tables (each laid out as one random cycle, word 0 = next).
// coop != 0: quad-cooperative fetch. *records_out = fetches per launch (both tables together).
BVH_AMD_API int bvh_amd_probe_mixed_walk(const void* d_small, uint32_t n_small, const void* d_big, uint32_t n_big, uint32_t steps, int blocks_per_cu, int reps,
                                         int coop, float* ms_out, unsigned long long* records_out, void* stream_)
{
    using namespace bvh_amd;
    if (!d_small || !d_big || !n_small || !n_big || !steps || reps < 1 || !ms_out) return fail(BVH_AMD_ERR_ARG, "probe_mixed_walk: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int device = 0, cus = 0;
    BVH_HIP_TRY(hipGetDevice(&device), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), BVH_AMD_ERR_HIP);
    if (blocks_per_cu < 1) blocks_per_cu = 7;
    const unsigned grid = static_cast<unsigned>(cus * blocks_per_cu);
    unsigned long long* sink = nullptr;
    BVH_HIP_TRY(hipMalloc(&sink, sizeof(*sink)), BVH_AMD_ERR_HIP);
    auto launch = [&]() {
        if (coop) hipLaunchKernelGGL(k_record_walk_mixed<true>, dim3(grid), dim3(256), 0, stream, static_cast<const WalkRec*>(d_small), n_small,
                                     static_cast<const WalkRec*>(d_big), n_big, steps, sink);
        else hipLaunchKernelGGL(k_record_walk_mixed<false>, dim3(grid), dim3(256), 0, stream, static_cast<const WalkRec*>(d_small), n_small,
                                static_cast<const WalkRec*>(d_big), n_big, steps, sink);
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        launch();
        e = hipEventRecord(e0, stream);
        for (int r = 0; r < reps; ++r) launch();
        if (e == hipSuccess) e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess) e = hipGetLastError();
        *ms_out = ms / reps;
        if (records_out) *records_out = static_cast<unsigned long long>(grid) * 256ull * steps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("probe_mixed_walk: ") + hipGetErrorString(e));
    return BVH_AMD_OK;
}

// d_table: n_records x 64 bytes, word 0 of record i = index of the next record of its chain (the caller lays out a random
// permutation cycle). Launches blocks_per_cu x CUs blocks of 256 lanes, `steps` records per chain, `reps` times after one
// warm-up launch; *ms_out = mean launch time, *records_out = records walked per launch (chains x steps).
BVH_AMD_API int bvh_amd_probe_record_walk_ex(const void* d_table, uint32_t n_records, uint32_t steps, int blocks_per_cu, int reps, int mode, int active,
                                             float* ms_out, unsigned long long* records_out, void* stream_)
{
    using namespace bvh_amd;
    if (!d_table || n_records == 0 || steps == 0 || reps < 1 || !ms_out || mode < 0 || mode > 9) return fail(BVH_AMD_ERR_ARG, "probe_record_walk: bad argument");
    if (active < 1 || active > 64) active = 64;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int device = 0, cus = 0;
    BVH_HIP_TRY(hipGetDevice(&device), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), BVH_AMD_ERR_HIP);
    if (blocks_per_cu < 1) blocks_per_cu = 7;
    const unsigned grid = static_cast<unsigned>(cus * blocks_per_cu);
    const auto* table = static_cast<const WalkRec*>(d_table);
    const uint32_t act = static_cast<uint32_t>(active);
    unsigned long long* sink = nullptr;
    BVH_HIP_TRY(hipMalloc(&sink, sizeof(*sink)), BVH_AMD_ERR_HIP);
    auto launch = [&]() {
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_record_walk, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 1: hipLaunchKernelGGL(k_record_walk_coop, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 4: hipLaunchKernelGGL(k_record_walk_dpp, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 5: hipLaunchKernelGGL(k_record_walk_wide<true>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(d_table), n_records, steps, sink); break;
        case 6: hipLaunchKernelGGL(k_record_walk_wide<false>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(d_table), n_records, steps, sink); break;
        case 7: hipLaunchKernelGGL(k_record_walk_mlp<2>, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 8: hipLaunchKernelGGL(k_record_walk_mlp<4>, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 9: hipLaunchKernelGGL(k_record_walk_dpp2, dim3(grid), dim3(256), 0, stream, table, n_records, steps, act, sink); break;
        case 2: hipLaunchKernelGGL(k_record_walk_quad<true>, dim3(grid), dim3(256), 0, stream, table, n_records, steps, sink); break;
        default: hipLaunchKernelGGL(k_record_walk_quad<false>, dim3(grid), dim3(256), 0, stream, table, n_records, steps, sink); break;
        }
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        launch();
        e = hipEventRecord(e0, stream);
        for (int r = 0; r < reps; ++r) launch();
        if (e == hipSuccess) e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess) e = hipGetLastError();
        *ms_out = ms / reps;
        const unsigned long long chains_per_wave = (mode == 2 || mode == 3) ? 16ull : (mode == 5 || mode == 6) ? 8ull : static_cast<unsigned long long>(active);
        const unsigned long long per_lane = mode == 7 || mode == 9 ? 2ull : mode == 8 ? 4ull : 1ull;
        if (records_out) *records_out = static_cast<unsigned long long>(grid) * 4ull * chains_per_wave * per_lane * steps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("probe_record_walk: ") + hipGetErrorString(e));
    return BVH_AMD_OK;
}

// the original entry point: every lane walks, per-lane loads
BVH_AMD_API int bvh_amd_probe_record_walk(const void* d_table, uint32_t n_records, uint32_t steps, int blocks_per_cu, int reps,
                                          float* ms_out, unsigned long long* records_out, void* stream)
{
    return bvh_amd_probe_record_walk_ex(d_table, n_records, steps, blocks_per_cu, reps, 0, 64, ms_out, records_out, stream);
}

} // extern "C"
