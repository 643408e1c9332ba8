/* The library's scratch must not outlive-depend on a caller's stream: the reference's C API has no
 * lifetime rule beyond _destroy (c_api/bvh.h:129-132), so a program may build and trace on a stream of its own, destroy that stream,
 * and keep using the library — including the BVH built on the dead stream — on other streams. Plain C11 + four HIP runtime calls.
 *
 *   BVH_AMD_CACHE_MB=8 stream_lifetime <n_tris> <n_rays>
 *
 *   1. stream A: tri_bounds, build (thread pool, Low and Medium), PrecomputedTri, a REORDERED ray batch (sort scratch), all on A;
 *      synchronize A; hipStreamDestroy(A). Scratch freed by those calls is now cached by the library.
 *   2. stream B and the null stream: builds that overflow the small cache bound (evictions of the blocks cached in step 1),
 *      bvh_amd_release_cached_memory() (the flush), a trace through the BVH of step 1, its destruction (device memory it took over
 *      from the scratch pool while A was alive), and once more round.
 *   3. every build of the same input gives the same Bvh::serialize stream, every trace the same hits.
 * Prints "stream lifetime ok" and exits 0. */
#include <bvh_amd.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int hipStreamCreateWithFlags(void** stream, unsigned flags);     /* libamdhip64 (no HIP headers needed for these) */
extern int hipStreamDestroy(void* stream);
extern int hipStreamSynchronize(void* stream);
extern int hipDeviceSynchronize(void);

static uint64_t rng_state = 0x1234567ull;
static double uniform01(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 40) / 16777216.0;
}

/* STREAM_LIFETIME_TRACE=1: one line per step on stderr (host-side progress: where the program stood when something went wrong) */
static int trace_on = 0;
#define STEP(...) do { if (trace_on) { fprintf(stderr, "[step] " __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "stream_lifetime: " __VA_ARGS__); fprintf(stderr, " [%s]\n", bvh_amd_last_error()); return 1; } } while (0)

static unsigned char* stream_of(struct bvh3f* bvh, size_t* bytes) {
    *bytes = bvh3f_serialize(bvh, NULL, 0);
    unsigned char* s = malloc(*bytes);
    if (bvh3f_serialize(bvh, s, *bytes) != *bytes) { free(s); return NULL; }
    return s;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 300000, m = argc > 2 ? strtoull(argv[2], NULL, 10) : 1u << 20;
    trace_on = getenv("STREAM_LIFETIME_TRACE") != NULL;
    CHECK(bvh_amd_device_count() >= 1, "no device");
    float* tris = malloc(n * 9 * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        const double c[3] = { uniform01(), uniform01(), uniform01() };
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) tris[9 * i + 3 * v + k] = (float)(c[k] + (2.0 * uniform01() - 1.0) * 0.01);
    }
    struct bvh_ray3f* rays = malloc(m * sizeof *rays);
    for (size_t j = 0; j < m; ++j) {
        rays[j].org.x = (float)uniform01(); rays[j].org.y = (float)uniform01(); rays[j].org.z = (float)uniform01();
        rays[j].dir.x = (float)(2.0 * uniform01() - 1.0); rays[j].dir.y = (float)(2.0 * uniform01() - 1.0); rays[j].dir.z = (float)(2.0 * uniform01() - 1.0);
        rays[j].tmin = 0.0f; rays[j].tmax = 3.4e38f;
    }
    float* d_tris = bvh_amd_device_alloc(n * 9 * sizeof(float));
    float* d_bb = bvh_amd_device_alloc(n * 6 * sizeof(float));
    float* d_cc = bvh_amd_device_alloc(n * 3 * sizeof(float));
    float* d_prims = bvh_amd_device_alloc(n * 12 * sizeof(float));
    struct bvh_ray3f* d_rays = bvh_amd_device_alloc(m * sizeof *rays);
    struct bvh_hit3f* d_hits = bvh_amd_device_alloc(m * sizeof(struct bvh_hit3f));
    CHECK(d_tris && d_bb && d_cc && d_prims && d_rays && d_hits, "device_alloc");
    STEP("d_tris %p d_bb %p d_cc %p d_prims %p d_rays %p .. %p d_hits %p", (void*)d_tris, (void*)d_bb, (void*)d_cc, (void*)d_prims, (void*)d_rays, (void*)(d_rays + m), (void*)d_hits);
    CHECK(bvh_amd_copy_to_device(d_tris, tris, n * 9 * sizeof(float)) == 0 && bvh_amd_copy_to_device(d_rays, rays, m * sizeof *rays) == 0, "copy_to_device");
    struct bvh_build_config low = { .quality = BVH_BUILD_QUALITY_LOW, .min_leaf_size = 1, .max_leaf_size = 8, .parallel_threshold = 1024 };
    struct bvh_build_config med = low;
    med.quality = BVH_BUILD_QUALITY_MEDIUM;

    /* 1. everything on stream A, then A dies */
    void* A = NULL;
    CHECK(hipStreamCreateWithFlags(&A, 1 /* hipStreamNonBlocking */) == 0 && A, "hipStreamCreate");
    CHECK(bvh_amd_tri_bounds3f(d_tris, n, d_bb, d_cc, A) == 0, "tri_bounds on A");
    STEP("tri_bounds queued");
    struct bvh3f* on_a = bvh3f_build_device(d_bb, d_cc, n, &med, BVH_AMD_BUILDER_DEFAULT_PARALLEL, A);
    CHECK(on_a, "Medium build on A");
    STEP("Medium on A built");
    struct bvh3f* low_a = bvh3f_build_device(d_bb, d_cc, n, &low, BVH_AMD_BUILDER_DEFAULT_PARALLEL, A);
    CHECK(low_a, "Low build on A");
    STEP("Low on A built");
    CHECK(bvh_amd_precompute_tris3f(d_tris, bvh3f_device_prim_ids(on_a), n, d_prims, A) == 0, "precompute_tris on A");
    CHECK(bvh3f_intersect_rays_tri(on_a, d_prims, d_rays, m, BVH_AMD_RAY_ROBUST | BVH_AMD_RAY_SORTED, d_hits, NULL, A) == 0, "reordered batch on A");
    STEP("batch on A queued");
    CHECK(hipStreamSynchronize(A) == 0, "synchronize A");
    STEP("A synchronized");
    struct bvh_hit3f* want = malloc(m * sizeof *want);
    CHECK(bvh_amd_copy_to_host(want, d_hits, m * sizeof *want) == 0, "copy hits");
    size_t want_bytes = 0, low_bytes = 0;
    unsigned char* want_stream = stream_of(on_a, &want_bytes);
    unsigned char* low_stream = stream_of(low_a, &low_bytes);
    CHECK(want_stream && low_stream, "serialize");
    bvh3f_destroy(low_a);
    STEP("Low of A destroyed");
    const size_t cached_after_a = bvh_amd_cached_scratch_bytes();
    CHECK(hipStreamDestroy(A) == 0, "hipStreamDestroy(A)");

    /* 2. life goes on without A */
    void* B = NULL;
    CHECK(hipStreamCreateWithFlags(&B, 1) == 0 && B, "hipStreamCreate B");
    size_t hits_found = 0;
    for (int round = 0; round < 3; ++round) {
        void* s = round == 1 ? NULL : B;                               /* the null stream in between */
        struct bvh3f* again = bvh3f_build_device(d_bb, d_cc, n, &med, BVH_AMD_BUILDER_DEFAULT_PARALLEL, s);     /* overflows the bound: evicts A's blocks */
        CHECK(again, "Medium build after A died (round %d)", round);
        STEP("round %d: Medium built", round);
        struct bvh3f* low_again = bvh3f_build_device(d_bb, d_cc, n, &low, BVH_AMD_BUILDER_DEFAULT_PARALLEL, s);
        CHECK(low_again, "Low build after A died (round %d)", round);
        STEP("round %d: Low built", round);
        size_t b1 = 0, b2 = 0;
        unsigned char* s1 = stream_of(again, &b1);
        unsigned char* s2 = stream_of(low_again, &b2);
        CHECK(s1 && b1 == want_bytes && memcmp(s1, want_stream, b1) == 0, "Medium stream differs in round %d", round);
        CHECK(s2 && b2 == low_bytes && memcmp(s2, low_stream, b2) == 0, "Low stream differs in round %d", round);
        free(s1); free(s2);
        /* the BVH that was built on A, traced on another stream */
        CHECK(bvh3f_intersect_rays_tri(round == 2 ? again : on_a, d_prims, d_rays, m, BVH_AMD_RAY_ROBUST | BVH_AMD_RAY_SORTED, d_hits, NULL, s) == 0, "batch in round %d", round);
        STEP("round %d: batch queued", round);
        CHECK(bvh_amd_synchronize(s) == 0, "synchronize");
        STEP("round %d: synchronized", round);
        struct bvh_hit3f* got = malloc(m * sizeof *got);
        CHECK(bvh_amd_copy_to_host(got, d_hits, m * sizeof *got) == 0, "copy hits");
        CHECK(memcmp(got, want, m * sizeof *got) == 0, "hits differ in round %d", round);
        for (size_t j = 0; j < m && round == 0; ++j) hits_found += got[j].prim != BVH_AMD_INVALID;
        free(got);
        bvh3f_destroy(low_again);
        bvh3f_destroy(again);
        STEP("round %d: both destroyed", round);
        if (round == 0) CHECK(bvh_amd_release_cached_memory() == 0, "release_cached_memory (flush of blocks freed under a dead stream)");
        if (round == 1) { bvh3f_destroy(on_a); on_a = NULL; }            /* memory it took from the pool while A was alive */
    }
    CHECK(hits_found > m / 100, "too few hits (%zu)", hits_found);
    CHECK(hipStreamDestroy(B) == 0, "hipStreamDestroy(B)");
    CHECK(bvh_amd_release_cached_memory() == 0, "release_cached_memory");
    CHECK(hipDeviceSynchronize() == 0, "hipDeviceSynchronize");
    printf("stream lifetime ok: %zu triangles, %zu rays, %zu hits, %zu bytes cached when A died, bound %zu\n", n, m, hits_found, cached_after_a,
           bvh_amd_scratch_cache_limit());
    return 0;
}
