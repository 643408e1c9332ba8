// Developer microbenchmark (NOT product code): the rate at which the MI355X memory system serves the traversal kernel's access
// pattern — every lane fetches whole 64-byte records (4 x global_load_dwordx4) at random, record-aligned addresses of a table.
//   gather_bench <table MiB> <records per lane> <loads in flight per lane: 1|2|4> <dependent: 0|1> [blocks per CU] [active lanes per wave]
// dependent = 1: the next record index comes out of the record just loaded (one outstanding record per chain, like a tree walk).
// Prints G records/s and TB/s. The table is filled with a random permutation cycle so dependent chains never repeat early.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

struct Rec { uint32_t w[16]; };

template <int InFlight, bool Dependent>
__global__ void __launch_bounds__(256) gather(const Rec* table, uint32_t n_rec, uint32_t per_lane, uint32_t active_lanes, unsigned long long* sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if ((threadIdx.x & 63) >= active_lanes) return;
    uint32_t idx[InFlight];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < InFlight; ++k) idx[k] = (gid * 2654435761u + k * 40503u * 65537u) % n_rec;
    for (uint32_t it = 0; it < per_lane; it += InFlight) {
        uint4 a[InFlight], b[InFlight], c[InFlight], d[InFlight];
#pragma unroll
        for (int k = 0; k < InFlight; ++k) {
            const uint4* q = reinterpret_cast<const uint4*>(table + idx[k]);
            a[k] = q[0]; b[k] = q[1]; c[k] = q[2]; d[k] = q[3];
        }
#pragma unroll
        for (int k = 0; k < InFlight; ++k) {
            acc += a[k].y + b[k].z + c[k].w + d[k].y;
            if (Dependent) idx[k] = a[k].x;                               // w[0] = the next record of this chain
            else idx[k] = (idx[k] * 1664525u + 1013904223u + d[k].x * 0u) % n_rec;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 60;
    const uint32_t per_lane = argc > 2 ? std::atoi(argv[2]) : 256;
    const int in_flight = argc > 3 ? std::atoi(argv[3]) : 1;
    const int dependent = argc > 4 ? std::atoi(argv[4]) : 1;
    const int blocks_per_cu = argc > 5 ? std::atoi(argv[5]) : 7;
    const uint32_t active = argc > 6 ? std::atoi(argv[6]) : 64;
    const uint32_t n_rec = static_cast<uint32_t>(mib * 1024 * 1024 / sizeof(Rec));
    std::vector<Rec> h(n_rec);
    std::vector<uint32_t> perm(n_rec);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (uint32_t i = 0; i < n_rec; ++i) { for (int k = 0; k < 16; ++k) h[perm[i]].w[k] = rng(); h[perm[i]].w[0] = perm[(i + 1) % n_rec]; }
    Rec* d; unsigned long long* sink;
    CHECK(hipMalloc(&d, n_rec * sizeof(Rec))); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemcpy(d, h.data(), n_rec * sizeof(Rec), hipMemcpyHostToDevice));
    int cus = 0; CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = cus * blocks_per_cu;
    auto launch = [&]() {
#define GB(F, D) hipLaunchKernelGGL((gather<F, D>), dim3(grid), dim3(256), 0, 0, d, n_rec, per_lane, active, sink)
        if (dependent) { if (in_flight == 1) GB(1, true); else if (in_flight == 2) GB(2, true); else GB(4, true); }
        else { if (in_flight == 1) GB(1, false); else if (in_flight == 2) GB(2, false); else GB(4, false); }
    };
    launch(); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double recs = double(grid) * 256.0 * (double(active) / 64.0) * (per_lane / in_flight * in_flight);
    std::printf("GATHER table=%zuMiB dependent=%d in_flight=%d blocks/CU=%d active_lanes=%u: %.3f ms  %.1f Grec/s  %.2f TB/s\n", mib, dependent, in_flight,
                blocks_per_cu, active, ms, recs / ms / 1e6, recs * 64 / ms / 1e9);
    return 0;
}
