// Developer probe (MI355X): does ONE busy workgroup run at the chip's full clock? Times the same dependent-VALU chain (s_memtime ticks and
// wall clock from events) alone and while a "heater" kernel keeps every other CU busy on a second stream.
//   hipcc --offload-arch=gfx950 -O3 tools/src/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k_chain(unsigned long long* out, float seed, int iters) {
    float a = seed + threadIdx.x, b = seed * 2.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
__global__ void k_heater(float* sink, int iters) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 1.0000001f + 1e-9f; }
    if (a + c + d == 123.456f) sink[0] = a;
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64);
    float* sink; (void)hipMalloc(&sink, 64);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200000;      // 3.2 M dependent adds
    for (int heater = 0; heater < 3; ++heater) {
        for (int rep = 0; rep < 3; ++rep) {
            if (heater == 1) hipLaunchKernelGGL(k_heater, dim3(255 * 4), dim3(256), 0, s2, sink, 4000000);
            if (heater == 2) hipLaunchKernelGGL(k_heater, dim3(32), dim3(64), 0, s2, sink, 40000000);
            (void)hipEventRecord(e0, s1);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, d, 1.0f, iters);
            (void)hipEventRecord(e1, s1);
            (void)hipStreamSynchronize(s1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[3]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            std::printf("%s: %.2f ms wall, %.2f memtime ticks per add, %.2f ns per add, wall_clock64 ticks %llu\n",
                        heater == 0 ? "alone            " : heater == 1 ? "beside a full-chip heater" : "beside a 32-wave heater ", ms, double(h[0]) / (16.0 * iters), ms * 1e6 / (16.0 * iters), h[1]);
            (void)hipDeviceSynchronize();
        }
    }
    // the same chain on every SIMD of the chip at once (2 waves per SIMD): does a busy chip clock higher?
    for (int blocks : {256, 1024, 2048}) {
        (void)hipEventRecord(e0, s1);
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, s1, d, 1.0f, iters);
        (void)hipEventRecord(e1, s1);
        (void)hipStreamSynchronize(s1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        std::printf("%d blocks of one wave: %.2f ms wall, %.2f memtime ticks per add, %.2f ns per add (kernel wall / adds per wave)\n", blocks, ms, double(h[0]) / (16.0 * iters), ms * 1e6 / (16.0 * iters));
    }
    return 0;
}
