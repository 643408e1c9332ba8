// Standalone reproducer attempt (no library code) for round 5's finding: scratch that the device's default memory pool UNMAPS
// (hipMemPoolTrimTo after a free) and maps again is read STALE by compute kernels while the copy engine sees the new contents
// (profiles/r05_pool_trim_stale_reads.txt). Per round: hipMallocAsync A -> fill A with pattern 1 -> hipFreeAsync A -> synchronise ->
// hipMemPoolTrimTo(pool, 0) -> hipMallocAsync B (same size: the address range comes back) -> fill B with pattern 2 -> a reader kernel
// counts the words that are NOT pattern 2, with plain / non-temporal / sc1 loads -> hipMemcpy of B to the host counts the same.
// Variants: writer = kernel or hipMemsetD32Async; an agent-scope fence + `buffer_inv sc1` in front of the reader's first load.
//   hipcc --offload-arch=gfx950 -O3 tools/src/pool_remap_repro.hip -o /tmp/pool_remap_repro && /tmp/pool_remap_repro [rounds] [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ void k_fill(uint32_t* p, size_t n, uint32_t tag) {
    for (size_t i = blockIdx.x * size_t{blockDim.x} + threadIdx.x; i < n; i += size_t{gridDim.x} * blockDim.x) p[i] = tag ^ static_cast<uint32_t>(i);
}
template <int Mode>
__global__ void k_count_stale(const uint32_t* p, size_t n, uint32_t tag, int invalidate, unsigned long long* stale) {
    if (invalidate) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("buffer_inv sc1" ::: "memory"); }
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * size_t{blockDim.x} + threadIdx.x; i < n; i += size_t{gridDim.x} * blockDim.x) {
        uint32_t v;
        if (Mode == 0) v = p[i];
        else if (Mode == 1) v = __builtin_nontemporal_load(p + i);
        else asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + i) : "memory");
        bad += v != (tag ^ static_cast<uint32_t>(i));
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    const size_t mib = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 64;
    const size_t n = mib << 18;
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipMemPool_t pool; CHECK(hipDeviceGetDefaultMemPool(&pool, 0));
    unsigned long long* d_stale; CHECK(hipMalloc(&d_stale, 8 * sizeof(unsigned long long)));
    std::vector<uint32_t> host(n);
    unsigned long long totals[2][2][3] = {}, host_bad_total = 0; int remapped = 0, same_address = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int memset_writer = 0; memset_writer < 2; ++memset_writer)
            for (int invalidate = 0; invalidate < 2; ++invalidate) {
                uint64_t threshold = 0; CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold));
                uint32_t *a = nullptr, *b = nullptr;
                CHECK(hipMallocAsync(reinterpret_cast<void**>(&a), n * 4, s));
                hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, a, n, 0x11110000u + r);
                // touch it through the caches of every CU so that translations and lines of THIS mapping exist
                CHECK(hipMemsetAsync(d_stale, 0, 8 * sizeof(unsigned long long), s));
                hipLaunchKernelGGL(k_count_stale<0>, dim3(1024), dim3(256), 0, s, a, n, 0x11110000u + r, 0, d_stale);
                CHECK(hipFreeAsync(a, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipMemPoolTrimTo(pool, 0));
                // something else takes address space in between, so that the range is really mapped anew
                void* other = nullptr; CHECK(hipMalloc(&other, (r % 7 + 1) << 20));
                CHECK(hipMallocAsync(reinterpret_cast<void**>(&b), n * 4, s));
                ++remapped; same_address += a == b;
                const uint32_t tag = 0x22220000u + r * 4 + memset_writer * 2 + invalidate;
                if (memset_writer) {
                    CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b), static_cast<int>(tag), n, s));     // (constant pattern: the check below is adjusted)
                    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, b, n / 2, tag);                        // first half by a kernel on top
                    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, b + n / 2, n - n / 2, tag ^ static_cast<uint32_t>(n / 2));   // (keeps tag ^ i for the whole range)
                } else hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, b, n, tag);
                CHECK(hipMemsetAsync(d_stale, 0, 8 * sizeof(unsigned long long), s));
                hipLaunchKernelGGL(k_count_stale<0>, dim3(1024), dim3(256), 0, s, b, n, tag, invalidate, d_stale + 0);
                hipLaunchKernelGGL(k_count_stale<1>, dim3(1024), dim3(256), 0, s, b, n, tag, invalidate, d_stale + 1);
                hipLaunchKernelGGL(k_count_stale<2>, dim3(1024), dim3(256), 0, s, b, n, tag, invalidate, d_stale + 2);
                unsigned long long st[3];
                CHECK(hipMemcpyAsync(st, d_stale, sizeof(st), hipMemcpyDeviceToHost, s));
                CHECK(hipMemcpyAsync(host.data(), b, n * 4, hipMemcpyDeviceToHost, s));
                CHECK(hipStreamSynchronize(s));
                unsigned long long host_bad = 0;
                for (size_t i = 0; i < n; ++i) host_bad += host[i] != (tag ^ static_cast<uint32_t>(i));
                host_bad_total += host_bad;
                for (int m = 0; m < 3; ++m) totals[memset_writer][invalidate][m] += st[m];
                if (st[0] | st[1] | st[2] | host_bad)
                    std::printf("round %d writer %s invalidate %d: stale words plain %llu nt %llu sc1 %llu, copy engine %llu (a %p b %p)\n", r, memset_writer ? "memset+kernel" : "kernel", invalidate,
                                st[0], st[1], st[2], host_bad, static_cast<void*>(a), static_cast<void*>(b));
                CHECK(hipFreeAsync(b, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipFree(other));
            }
    }
    std::printf("%d free -> trim -> malloc cycles of %zu MiB (%d came back at the same address): copy engine saw %llu wrong words\n", remapped, mib, same_address, host_bad_total);
    for (int w = 0; w < 2; ++w) for (int inv = 0; inv < 2; ++inv)
        std::printf("  writer %-13s invalidate %d: stale words seen by kernels: plain %llu, non-temporal %llu, sc1 %llu\n", w ? "memset+kernel" : "kernel", inv, totals[w][inv][0], totals[w][inv][1], totals[w][inv][2]);
    return 0;
}
