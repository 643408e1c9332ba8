// Developer microbenchmark: host cost of hipMallocAsync / hipFreeAsync (cached pool) and hipMalloc / hipFree per call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t keep = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    hipStream_t s; hipStreamCreate(&s);
    const size_t sizes[] = {4, 4096, 1 << 20, 16 << 20, 64 << 20};
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<void*> p(40);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 40; ++i) hipMallocAsync(&p[i], sizes[i % 5], s);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 40; ++i) hipFreeAsync(p[i], s);
        auto t2 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        std::printf("async: 40 mallocs %.1f us, 40 frees %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<void*> p(40);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 40; ++i) hipMalloc(&p[i], sizes[i % 5]);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 40; ++i) hipFree(p[i]);
        auto t2 = std::chrono::steady_clock::now();
        std::printf("plain: 40 mallocs %.1f us, 40 frees %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    return 0;
}
