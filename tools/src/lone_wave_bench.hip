// Developer micro-benchmark (MI355X): what ONE wavefront pays per instruction of the kinds the candidate-heap head is made of,
// alone on its CU and with three neighbour waves of the same workgroup polling LDS. Times are s_memtime ticks (shader clocks).
//   hipcc --offload-arch=gfx950 -O3 tools/src/lone_wave_bench.hip -o tools/bin/lone_wave_bench && tools/bin/lone_wave_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X X X X X X X X X X X X X X X X
constexpr int kIters = 512;

__device__ inline unsigned long long now() { return __builtin_readcyclecounter(); }

__global__ void __launch_bounds__(256) k_bench(unsigned long long* out, int neighbours, float seed, int which) {
    __shared__ uint32_t flag[64];
    __shared__ unsigned long long cells[256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 64) flag[threadIdx.x] = 0;
    cells[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (wave != 0) {
        if (!neighbours) return;
        while (__hip_atomic_load(&flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        return;
    }
    float a = seed + lane, b = seed * 2.0f;
    uint32_t u = lane;
    float c2 = seed;
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    int t = 0;
    unsigned long long t0, t1;
    // 0: empty loop overhead
    if (which < 0 || which == 0) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) asm volatile("" : "+v"(a));
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 1: 16 dependent v_add_f32
    if (which < 0 || which == 1) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 2: 16 x (v_cmp -> v_cndmask) dependent through vcc
    if (which < 0 || which == 2) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 3: 16 x (v_cmp -> s_ff1 -> v_readlane -> v_add) : VALU -> SALU -> VALU(readlane) -> VALU
    if (which < 0 || which == 3) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) {
        REP16({ const uint64_t m = __ballot(a > b) | 1ull; const int j = __ffsll((long long)m) - 1; const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), j)); a += r; asm volatile("" : "+v"(a)); })
    }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 4: 16 x ds_bpermute dependent
    if (which < 0 || which == 4) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ u = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((u & 63u) * 4u), (int)(u + 1u)); asm volatile("" : "+v"(u)); }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 5: 16 x ds_read_b64 dependent address
    if (which < 0 || which == 5) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ u = (uint32_t)cells[u & 255u]; asm volatile("" : "+v"(u)); }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 6: 16 x taken scalar branch (skip over one instruction)
    if (which < 0 || which == 6) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_branch 1f\n v_add_f32 %0, %0, %0\n1:" : "+v"(a));) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 7: 16 x s_memtime
    if (which < 0 || which == 7) {
    t0 = now();
    unsigned long long acc = 0;
    for (int i = 0; i < kIters; ++i) { REP16(acc += now();) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0 + (acc & 0); 
    }
    ++t;
    // 8: 16 x (ds_write ; release store of a flag) : the hand-off pattern
    if (which < 0 || which == 8) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ cells[lane] = u; __hip_atomic_store(&flag[1], u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); u += 1; }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 9: 16 x (s_and_saveexec ; v_mov ; s_or exec) divergence bracket without branch
    if (which < 0 || which == 9) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 10: 16 independent v_add_f32 pairs (two chains)
    if (which < 0 || which == 10) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(c2) : "v"(b));) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 11: 16 x s_add_u32 dependent
    if (which < 0 || which == 11) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 12: 16 x (v_readlane -> s_add -> v_mov from sgpr -> v_add)
    if (which < 0 || which == 12) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ int r = __builtin_amdgcn_readlane((int)u, 3); r += 1; u += (uint32_t)r; asm volatile("" : "+v"(u)); }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 13: 16 x DPP wave_shr
    if (which < 0 || which == 13) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ u = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x138, 0xf, 0xf, false) + 1u; asm volatile("" : "+v"(u)); }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    // 14: 16 x acquire load of an LDS word
    if (which < 0 || which == 14) {
    t0 = now();
    for (int i = 0; i < kIters; ++i) { REP16({ u += __hip_atomic_load(&flag[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" : "+v"(u)); }) }
    t1 = now(); if (lane == 0) out[t] = t1 - t0; 
    }
    ++t;
    if (lane == 0) { out[31] = (unsigned long long)(a + c2) + u + s; }
    __hip_atomic_store(&flag[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
    unsigned long long* d; hipMalloc(&d, 32 * 8);
    const char* names[] = {"empty loop", "v_add_f32 dependent", "v_cmp + v_cndmask (vcc)", "ballot + ffs + readlane + add", "ds_bpermute dependent", "ds_read_b64 dependent",
                           "taken s_branch", "s_memtime", "ds_write + release flag", "saveexec bracket (4 instr)", "2 independent v_add", "s_add_u32 dependent",
                           "readlane + s_add + v_add", "dpp wave_shr + add", "acquire LDS load"};
    for (int nb = 0; nb < 2; ++nb) {
        hipMemset(d, 0, 32 * 8);
        for (int which = 0; which < 15; ++which) {
            if (which == 11) continue;                      // (a scalar-only loop beside polling neighbours never ended on the test box: skipped)
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, d, nb, 1.0f, which);
                hipDeviceSynchronize();
            }
            std::fprintf(stderr, "section %d done\n", which);
        }
        unsigned long long h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        std::fflush(stdout); std::printf("--- %s\n", nb ? "three neighbour waves polling an LDS flag (s_sleep 1)" : "alone on the CU");
        for (int t = 0; t < 15; ++t) if (t != 11) std::printf("%-34s %8.1f clocks per unit (16 units per iteration, %d iterations; loop overhead %.1f per iteration)\n", names[t],
                                                 double(h[t] - (t ? h[0] : 0)) / (kIters * (t ? 16 : 1)), kIters, double(h[0]) / kIters);
    }
    std::fflush(stdout);
    return 0;
}
