// Device-side helpers of the batch traversal kernels (traverse.hip): per-type arithmetic, record / ray / primitive loads, the
// kernel argument block. Kept apart from traverse.hip so that tests/cpp/trace_body_host.cpp can compile the kernel BODY
// (trace_body.inc) for the host with one emulated lane (test infrastructure; the product only ever runs it on the device).
// Expects common.h (or the test's stand-ins for it) to have been included: PairNode, HitOf, kCountBits, kCountMask, kWave, LEAF_*.
#pragma once

// An empty asm the value must pass through: stops the compiler from fusing the loads on either side of it.
#if defined(__HIPCC__)
#define BVH_AMD_KEEP_APART(v) asm volatile("" : "+v"(v))
#else
#define BVH_AMD_KEEP_APART(v) asm volatile("" : "+r"(v))
#endif

namespace bvh_amd {

namespace {

constexpr int kBlock = 256;
constexpr int kLdsDepth = 20;
// Refill when at least this many lanes of the wave are idle / run the leaf code when at least this many lanes wait at a leaf.
// Swept on the device (profiles/README.md): a walk served by the L1 / L2s (small trees, coherent or coherence-sorted batches) is
// bound by issue slots and wants its lanes refilled early (soup_1m sorted: 1744 Mrays/s at 54 / 8, 1970 at 36 / 12; Sponza proxy
// 5.07 -> 5.77 Grays/s, 1M-triangle terrain 5.35 -> 6.59). Only a walk that misses the L2s all the time (unsorted uniform rays on
// the 1M soup) would rather have rare, full refills, and by little: 1404 Mrays/s at 54 / 8, 1382 at 36 / 12.
constexpr int kRefillThreshold = 36;
constexpr int kRefillThresholdSmall = 20;          // per-lane kernel, batches of up to 2^21 rays (traverse.hip: launch_planned)
constexpr int kLeafThreshold = 12;
// Quad-cooperative record fetch (coop_load_pair below) and the thresholds that go with it, by kind of launch. Measured on the
// device (profiles/r03_traversal_experiments.md): the cooperative fetch cuts the L1's lane requests per visited record from 4 to
// 1 but puts ~35 VALU instructions and two DPP stages on the dependent chain of every step, and wants its lanes refilled EARLY
// (idle lanes still help loading). It pays where a step waits on memory anyway — any-hit walks (+16 % on the Sponza proxy's
// shadow rays) and closest-hit walks through big incoherent trees (+5 % on the 1M soup) — and loses where the walk is served by
// the L1 / L2 and the chain's length is what binds (-3 % Sponza proxy, -7 % terrain, closest-hit).
constexpr int kCoopRefillAny = 20, kCoopLeafAny = 20;          // any-hit
constexpr int kCoopRefillHeavy = 12, kCoopLeafHeavy = 12;      // closest-hit, trees beyond the L2s crossed by long walks


template <typename T> struct Num;
template <> struct Num<float> {
    static constexpr float kMax = 3.402823466e+38f, kEps = 1.1920928955078125e-07f;
    __device__ static bool finite(float x) { return (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u; }
    __device__ static float bump2(float x) { return __uint_as_float(__float_as_uint(x) + 2u); }
    __device__ static bool sign(float x) { return (__float_as_uint(x) >> 31) != 0; }
    __device__ static float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    __device__ static float sqrt_(float x) { return __builtin_sqrtf(x); }
    __device__ static float abs_(float x) { return __builtin_fabsf(x); }
    __device__ static float copysign_(float a, float b) { return __builtin_copysignf(a, b); }
    __device__ static float max_num(float a, float b) { return __builtin_fmaxf(a, b); }      // v_max_f32: the non-NaN operand wins
    __device__ static float min_num(float a, float b) { return __builtin_fminf(a, b); }
};
template <> struct Num<double> {
    static constexpr double kMax = 1.7976931348623157e+308, kEps = 2.220446049250313e-16;
    __device__ static bool finite(double x) {
        return (static_cast<uint32_t>(__double_as_longlong(x) >> 32) & 0x7FF00000u) != 0x7FF00000u;
    }
    __device__ static double bump2(double x) { return __longlong_as_double(__double_as_longlong(x) + 2ll); }
    __device__ static bool sign(double x) { return (__double_as_longlong(x) >> 63) != 0; }
    __device__ static double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ static double sqrt_(double x) { return __builtin_sqrt(x); }
    __device__ static double abs_(double x) { return __builtin_fabs(x); }
    __device__ static double copysign_(double a, double b) { return __builtin_copysign(a, b); }
    __device__ static double max_num(double a, double b) { return __builtin_fmax(a, b); }
    __device__ static double min_num(double a, double b) { return __builtin_fmin(a, b); }
};

// utils.h:41-43 — must stay compare+select: the second operand wins on NaN and on equality.
template <typename T> __device__ inline T pick_min(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ inline T pick_max(T a, T b) { return a > b ? a : b; }

template <typename T> __device__ inline T dot3(T a0, T a1, T a2, T b0, T b1, T b2) {   // vec.h:98-100
    return ((T(0) + a0 * b0) + a1 * b1) + a2 * b2;
}
// A ds_read_b32 the optimiser may neither drop nor merge with a load from another address space.
__device__ inline uint32_t lds_word(const uint32_t* shared) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const volatile uint32_t LdsWord;
    return *(LdsWord*)shared;
#else
    return *shared;                                // (tests/cpp/trace_body_host.cpp compiles the body for the host)
#endif
}

template <typename T> __device__ inline T dot2(T a0, T a1, T b0, T b1) { return (T(0) + a0 * b0) + a1 * b1; }   // Vec<T, 2>

constexpr uint32_t kTicketStride = 16;       // ticket counters 128 bytes apart

template <typename T>
struct TraceArgs {
    const PairNode<T>* pairs;
    const T* prims;
    const T* rays;
    typename HitOf<T>::Type* hits;
    unsigned long long n;
    unsigned long long* work;                  // next ray ticket of part p of this launch at work[p * kTicketStride]
    unsigned long long part_size;              // tickets [p * part_size, min(n, (p + 1) * part_size)) form part p (a multiple of 64)
    uint32_t parts;                            // 1, or one part per XCD (see trace_body.inc: refill)
    bvh_amd_counters* counters;
    const uint32_t* order;                     // optional: ticket -> ray index (coherence sort); results are unaffected
    uint32_t* deep;                            // stack entries beyond 64, deep_cap per resident lane (trees deeper than 64 levels only)
    uint32_t deep_cap;
    uint32_t root_index;
    int refill_threshold;                      // refill when at least this many lanes are idle
    int leaf_threshold;                        // leave the inner-node loop when this many lanes wait at a leaf
    uint32_t coop;                             // host side only: launch the quad-cooperative variant (float, 3D, trees below 2^26 pairs)
    uint32_t prim_stride;                      // scalars from one primitive to the next in `prims` (12: PrecomputedTri; developer knob: 16 = padded to a 64-byte line)
    uint32_t stream_hints;                     // bit 0: rays / order / hit records are touched once: load / store them non-temporally (developer knob)
    uint32_t stagger;                          // 0: off; else tickets per eighth of the grid (see trace_body.inc: staggered drain)
    uint32_t one_shot;                         // 1: grid = ceil(n / 256) blocks, wave w traces rays [64 w, 64 w + 64) and leaves (small batches)
    unsigned long long* wave_times;            // developer library only (bvh_amd_experiment("wave_times", 1)): per wave {begin, last refill, end, xcc | rays, ticks inside refills, refills} in
                                               // s_memrealtime ticks (100 MHz) + {XCC id, rays traced}; nullptr otherwise — the release kernels never read it
};

__device__ inline void load_pair(const PairNode<float>* p, float (&lb)[6], float (&rb)[6], uint32_t& li, uint32_t& ri) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1], c = q[2];
    uint2 d = reinterpret_cast<const uint2*>(p)[6];
    lb[0] = a.x; lb[1] = a.y; lb[2] = a.z; lb[3] = a.w; lb[4] = b.x; lb[5] = b.y;
    rb[0] = b.z; rb[1] = b.w; rb[2] = c.x; rb[3] = c.y; rb[4] = c.z; rb[5] = c.w;
    li = d.x; ri = d.y;
}
__device__ inline void load_pair(const PairNode<double>* p, double (&lb)[6], double (&rb)[6], uint32_t& li, uint32_t& ri) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 v = q[i]; lb[2 * i] = v.x; lb[2 * i + 1] = v.y; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 v = q[3 + i]; rb[2 * i] = v.x; rb[2 * i + 1] = v.y; }
    uint2 d = reinterpret_cast<const uint2*>(p)[12];
    li = d.x; ri = d.y;
}

// ---- quad-cooperative record fetch (trace_kernel<..., Coop = true>) -----------------------------------------------------------
// The L1 (TCP) charges a divergent wave-instruction per LANE request: four global_load_dwordx4 of a lane's own 64-byte record
// cost four requests, and measured (csrc/probe.hip, tools/tcp_probe.py) a wave-instruction takes ~8.5 clk + 0.56 clk per distinct
// line request: ~2.8 clk per record at 64 active lanes, 3.5 at 28 — while a QUAD whose four lanes read the four 16-byte chunks of
// ONE record is a single line request (0.95 lines / clk / CU). So the four lanes of a quad fetch each other's records: in
// instruction k every lane loads chunk (lane & 3) of the record wanted by the quad's lane k (skipped when that lane wants none),
// and a 4 x 4 transpose inside the quad — two butterfly stages of DPP quad_perm moves, no LDS — hands every lane the four chunks
// of its own record. Requests per visited record: 1 instead of 4. Every lane of the wave must call this (wave-uniform control flow).
#if defined(__HIPCC__)
template <int CTRL> __device__ inline uint32_t quad_perm(uint32_t v) {       // DPP quad_perm: lane l reads lane (l & ~3) | perm[l & 3]
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xF, 0xF, true));
}
// One butterfly step of the quad transpose on a register pair (a, b) of N dwords: the lower lane of each lane pair (partner =
// lane ^ 1, or lane ^ 2 with ByTwo) keeps a and takes the partner's a into b, the upper lane keeps b and takes the partner's b into
// a. One v_cndmask_b32 with a DPP source per dword and direction (left to the compiler the DPP move and the select stay two
// instructions; inline asm because the select mask has to sit in VCC). The leading s_nop covers the two wait states a DPP read
// needs after a VALU write of its source (the assembler does not see hazards across an asm statement).
template <bool ByTwo> __device__ inline void quad_exchange4(uint32_t (&a)[4], uint32_t (&b)[4]) {
    const uint64_t lower = ByTwo ? 0x3333333333333333ull : 0x5555555555555555ull;
    uint32_t n0, n1, n2, n3;
    if (ByTwo)
        asm("s_nop 1\n\ts_mov_b64 vcc, %12\n\t"
            "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_not_b64 vcc, vcc\n\t"
            "v_cndmask_b32_dpp %4, %8, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %5, %9, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %6, %10, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %7, %11, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
            : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(lower) : "vcc", "scc");
    else
        asm("s_nop 1\n\ts_mov_b64 vcc, %12\n\t"
            "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_not_b64 vcc, vcc\n\t"
            "v_cndmask_b32_dpp %4, %8, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %5, %9, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %6, %10, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %7, %11, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
            : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(lower) : "vcc", "scc");
    a[0] = n0; a[1] = n1; a[2] = n2; a[3] = n3;
}
#elif defined(BVH_HOST_WAVE64)
// tests/cpp/trace_body_host.cpp, 64 fibers: the two quad primitives by their MEANING (every lane exchanges through the harness's
// wave_read), so that coop_load_pair below — which chunk each lane loads for whom, the order of the butterfly stages, where the
// transposed dwords end up — runs on the host as the very text the device compiles. Only the DPP encodings themselves stay untested here.
template <int CTRL> inline uint32_t quad_perm(uint32_t v) {
    const int l = static_cast<int>(threadIdx.x) & 63;
    return wave_read(v, (l & ~3) | ((CTRL >> (2 * (l & 3))) & 3), true);
}
template <bool ByTwo> inline void quad_exchange4(uint32_t (&a)[4], uint32_t (&b)[4]) {
    const int l = static_cast<int>(threadIdx.x) & 63, bit = ByTwo ? 2 : 1, partner = l ^ bit;
    const bool lower = (l & bit) == 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t pa = wave_read(a[i], partner, true), pb = wave_read(b[i], partner, true);
        const uint32_t na = lower ? a[i] : pb, nb = lower ? pa : b[i];
        a[i] = na; b[i] = nb;
    }
}
#endif
#if defined(__HIPCC__) || defined(BVH_HOST_WAVE64)
__device__ inline void coop_load_pair(const PairNode<float>* pairs, uint32_t want, int lane, float (&lb)[6], float (&rb)[6], uint32_t& li, uint32_t& ri) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t j = static_cast<uint32_t>(lane) & 3u;
    const uint32_t o0 = quad_perm<0x00>(want), o1 = quad_perm<0x55>(want), o2 = quad_perm<0xAA>(want), o3 = quad_perm<0xFF>(want);
    // (uniform base + 32-bit byte offset: the loads take the SGPR-base form and need no 64-bit address arithmetic per lane;
    //  launch_variant only selects this kernel for trees of fewer than 2^26 pair records)
    const char* base = reinterpret_cast<const char*>(pairs);
    const uint32_t chunk = j * 16u;
    uint4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    if (o0 != kNone) c0 = *reinterpret_cast<const uint4*>(base + (o0 * 64u + chunk));
    if (o1 != kNone) c1 = *reinterpret_cast<const uint4*>(base + (o1 * 64u + chunk));
    if (o2 != kNone) c2 = *reinterpret_cast<const uint4*>(base + (o2 * 64u + chunk));
    if (o3 != kNone) c3 = *reinterpret_cast<const uint4*>(base + (o3 * 64u + chunk));
    uint32_t v0[4] = {c0.x, c0.y, c0.z, c0.w}, v1[4] = {c1.x, c1.y, c1.z, c1.w}, v2[4] = {c2.x, c2.y, c2.z, c2.w}, v3[4] = {c3.x, c3.y, c3.z, c3.w};
    // v_k[lane j] = chunk j of the record of quad lane k  ->  v_k[lane j] = chunk k of the record of quad lane j
    quad_exchange4<false>(v0, v1); quad_exchange4<false>(v2, v3);
    quad_exchange4<true>(v0, v2); quad_exchange4<true>(v1, v3);
    lb[0] = __uint_as_float(v0[0]); lb[1] = __uint_as_float(v0[1]); lb[2] = __uint_as_float(v0[2]); lb[3] = __uint_as_float(v0[3]);
    lb[4] = __uint_as_float(v1[0]); lb[5] = __uint_as_float(v1[1]); rb[0] = __uint_as_float(v1[2]); rb[1] = __uint_as_float(v1[3]);
    rb[2] = __uint_as_float(v2[0]); rb[3] = __uint_as_float(v2[1]); rb[4] = __uint_as_float(v2[2]); rb[5] = __uint_as_float(v2[3]);
    li = v3[0]; ri = v3[1];
}
// Node<double, N>: 128-byte records, fetched as two 64-byte halves by the same quad scheme (round 4): in instructions k and 4 + k the
// quad's four lanes load the four 16-byte chunks of the first / second half of the record wanted by its lane k — two quad-coalesced
// requests per record instead of a lane's eight — and two 4 x 4 transposes hand every lane its own 128 bytes:
// first half = lb[0..5], rb[0..1]; second half = rb[2..5], {li, ri}, padding (common.h: PairNode<double>).
__device__ inline void coop_load_pair(const PairNode<double>* pairs, uint32_t want, int lane, double (&lb)[6], double (&rb)[6], uint32_t& li, uint32_t& ri) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t j = static_cast<uint32_t>(lane) & 3u;
    const uint32_t o0 = quad_perm<0x00>(want), o1 = quad_perm<0x55>(want), o2 = quad_perm<0xAA>(want), o3 = quad_perm<0xFF>(want);
    const char* base = reinterpret_cast<const char*>(pairs);     // (launch_variant selects this kernel for fewer than 2^25 pair records only)
    const uint32_t chunk = j * 16u;
    uint4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
    if (o0 != kNone) { c0 = *reinterpret_cast<const uint4*>(base + (o0 * 128u + chunk)); d0 = *reinterpret_cast<const uint4*>(base + (o0 * 128u + 64u + chunk)); }
    if (o1 != kNone) { c1 = *reinterpret_cast<const uint4*>(base + (o1 * 128u + chunk)); d1 = *reinterpret_cast<const uint4*>(base + (o1 * 128u + 64u + chunk)); }
    if (o2 != kNone) { c2 = *reinterpret_cast<const uint4*>(base + (o2 * 128u + chunk)); d2 = *reinterpret_cast<const uint4*>(base + (o2 * 128u + 64u + chunk)); }
    if (o3 != kNone) { c3 = *reinterpret_cast<const uint4*>(base + (o3 * 128u + chunk)); d3 = *reinterpret_cast<const uint4*>(base + (o3 * 128u + 64u + chunk)); }
    uint32_t v0[4] = {c0.x, c0.y, c0.z, c0.w}, v1[4] = {c1.x, c1.y, c1.z, c1.w}, v2[4] = {c2.x, c2.y, c2.z, c2.w}, v3[4] = {c3.x, c3.y, c3.z, c3.w};
    uint32_t w0[4] = {d0.x, d0.y, d0.z, d0.w}, w1[4] = {d1.x, d1.y, d1.z, d1.w}, w2[4] = {d2.x, d2.y, d2.z, d2.w}, w3[4] = {d3.x, d3.y, d3.z, d3.w};
    quad_exchange4<false>(v0, v1); quad_exchange4<false>(v2, v3);
    quad_exchange4<true>(v0, v2); quad_exchange4<true>(v1, v3);
    quad_exchange4<false>(w0, w1); quad_exchange4<false>(w2, w3);
    quad_exchange4<true>(w0, w2); quad_exchange4<true>(w1, w3);
    auto dbl = [](uint32_t lo, uint32_t hi) { return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo)); };
    lb[0] = dbl(v0[0], v0[1]); lb[1] = dbl(v0[2], v0[3]); lb[2] = dbl(v1[0], v1[1]); lb[3] = dbl(v1[2], v1[3]);
    lb[4] = dbl(v2[0], v2[1]); lb[5] = dbl(v2[2], v2[3]); rb[0] = dbl(v3[0], v3[1]); rb[1] = dbl(v3[2], v3[3]);
    rb[2] = dbl(w0[0], w0[1]); rb[3] = dbl(w0[2], w0[3]); rb[4] = dbl(w1[0], w1[1]); rb[5] = dbl(w1[2], w1[3]);
    li = w2[0]; ri = w2[1];
    (void)w3;
}
#else
template <typename T> inline void coop_load_pair(const PairNode<T>* p, uint32_t want, int, T (&lb)[6], T (&rb)[6], uint32_t& li, uint32_t& ri) {
    if (want != 0xFFFFFFFFu) load_pair(p + want, lb, rb, li, ri);          // host harness: one emulated lane has no quad
}
#endif

__device__ inline void load_prim12(const float* p, float (&v)[12]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ inline void load_prim12(const double* p, double (&v)[12]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 6; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
__device__ inline void load_prim4(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ inline void load_prim4(const double* p, double (&v)[4]) {
    const double2* q = reinterpret_cast<const double2*>(p);
    double2 a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ inline void load_ray(const float* p, float (&v)[8]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void load_ray(const double* p, double (&v)[8]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
// The same, marked non-temporal: a ray is read once and its hit record written once per launch; without the hint each of them
// takes a line of the XCD's L2 away from the node / triangle working set the in-flight rays share.
__device__ inline void load_ray_nt(const float* p, float (&v)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* q = reinterpret_cast<const f4*>(p);
    const f4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#else
    load_ray(p, v);
#endif
}
__device__ inline void load_ray_nt(const double* p, double (&v)[8]) { load_ray(p, v); }
__device__ inline uint32_t load_word_nt(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
// Ray<T, 2>: {org[2], dir[2], tmin, tmax}
__device__ inline void load_ray2(const float* p, float (&v)[6]) {
    const float2* q = reinterpret_cast<const float2*>(p);
    float2 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}
__device__ inline void load_ray2(const double* p, double (&v)[6]) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) { double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}
__device__ inline void store_hit(bvh_hit3f* out, uint32_t prim, float t, float u, float v) {
    *reinterpret_cast<float4*>(out) = make_float4(__uint_as_float(prim), t, u, v);
}
__device__ inline void store_hit(bvh_hit3d* out, uint32_t prim, double t, double u, double v) {
    double2* q = reinterpret_cast<double2*>(out);
    q[0] = make_double2(__longlong_as_double(static_cast<long long>(prim)), t);
    q[1] = make_double2(u, v);
}
__device__ inline void store_hit_nt(bvh_hit3f* out, uint32_t prim, float t, float u, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 h; h.x = __uint_as_float(prim); h.y = t; h.z = u; h.w = v;
    __builtin_nontemporal_store(h, reinterpret_cast<f4*>(out));
#else
    store_hit(out, prim, t, u, v);
#endif
}
__device__ inline void store_hit_nt(bvh_hit3d* out, uint32_t prim, double t, double u, double v) { store_hit(out, prim, t, u, v); }

} // namespace

} // namespace bvh_amd
