// Measurement aid (not on the product path; called by bench.py): the rate at which this GPU's memory system serves the
// traversal kernel's access pattern with NOTHING else in the way — every lane walks a chain of 64-byte records through a
// table (four global_load_dwordx4 per record, the next record's index comes out of the record just loaded, exactly one
// record in flight per lane, like a tree walk). With a table that does not fit the L2s this is the rate of the L2-miss path;
// bench.py prices the traversal kernel's measured L2 misses against it ("roofline_binding").
#include "common.h"

namespace bvh_amd {

namespace {

struct WalkRec { uint32_t w[16]; };

__global__ void __launch_bounds__(256) k_record_walk(const WalkRec* table, uint32_t n_rec, uint32_t steps, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
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

} // namespace

} // namespace bvh_amd

extern "C" {

// d_table: n_records x 64 bytes, word 0 of record i = index of the next record of its chain (the caller lays out a random
// permutation cycle). Launches blocks_per_cu x CUs blocks of 256 lanes, `steps` records per lane, `reps` times after one
// warm-up launch; *ms_out = mean launch time. Records walked per launch = blocks x 256 x steps (returned in *records_out).
BVH_AMD_API int bvh_amd_probe_record_walk(const void* d_table, uint32_t n_records, uint32_t steps, int blocks_per_cu, int reps,
                                          float* ms_out, unsigned long long* records_out, void* stream_)
{
    using namespace bvh_amd;
    if (!d_table || n_records == 0 || steps == 0 || reps < 1 || !ms_out) return fail(BVH_AMD_ERR_ARG, "probe_record_walk: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int device = 0, cus = 0;
    BVH_HIP_TRY(hipGetDevice(&device), BVH_AMD_ERR_HIP);
    BVH_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), BVH_AMD_ERR_HIP);
    if (blocks_per_cu < 1) blocks_per_cu = 7;
    const unsigned grid = static_cast<unsigned>(cus * blocks_per_cu);
    unsigned long long* sink = nullptr;
    BVH_HIP_TRY(hipMalloc(&sink, sizeof(*sink)), BVH_AMD_ERR_HIP);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_record_walk, dim3(grid), dim3(256), 0, stream, static_cast<const WalkRec*>(d_table), n_records, steps, sink);
        e = hipEventRecord(e0, stream);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_record_walk, dim3(grid), dim3(256), 0, stream, static_cast<const WalkRec*>(d_table), n_records, steps, sink);
        if (e == hipSuccess) e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / reps;
        if (records_out) *records_out = static_cast<unsigned long long>(grid) * 256ull * steps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail(BVH_AMD_ERR_HIP, std::string("probe_record_walk: ") + hipGetErrorString(e));
    return BVH_AMD_OK;
}

} // extern "C"
