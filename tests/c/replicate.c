/* Multi-GPU through the C-ABI alone (plain C11, no HIP headers, no Python): what a C user of the reference API does to shard a
 * ray batch over the GPUs of a node (include/bvh_amd.h "multi-GPU").
 *
 *   replicate <n_tris> <n_rays> [n_devices]
 *
 *   1. build on device 0 (bvh3f_build_device, DefaultBuilder with thread pool, Quality::Medium), PrecomputedTri in BVH order;
 *   2. bvh3f_replicate -> one resident copy per device (RCCL broadcast of the Bvh::serialize stream + primitives);
 *   3. every device traces its contiguous shard of the rays (k * ceil(R / G) ...) with bvh3f_intersect_rays_tri;
 *   4. the concatenated shards must equal, byte for byte, device 0 tracing the whole batch; every copy's bvh3f_serialize stream
 *      must equal the original's;
 *   5. the one-process-per-GPU entry points with a communicator of size 1: bvh_amd_comm_unique_id / _create / bvh3f_broadcast
 *      (the root gets its own objects back), bvh_amd_comm_broadcast of a raw buffer.
 * Prints "replicate ok: ..." and exits 0, or says what differs and exits 1. n_devices defaults to bvh_amd_device_count().
 */
#include <bvh_amd.h>

#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform01(void) {                                /* splitmix64 -> [0, 1) */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 40) / 16777216.0;
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "replicate: " __VA_ARGS__); fprintf(stderr, " [%s]\n", bvh_amd_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 20000, m = argc > 2 ? strtoull(argv[2], NULL, 10) : 100000;
    const int have = bvh_amd_device_count();
    CHECK(have >= 1, "no device");
    int g = argc > 3 ? atoi(argv[3]) : have;
    if (g > have) g = have;
    CHECK(g >= 1 && g <= 64, "bad device count");

    /* scene + rays on the host */
    float* tris = malloc(n * 9 * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        const double c[3] = { uniform01(), uniform01(), uniform01() };
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) tris[9 * i + 3 * v + k] = (float)(c[k] + (2.0 * uniform01() - 1.0) * 0.02);
    }
    struct bvh_ray3f* rays = malloc(m * sizeof *rays);
    for (size_t j = 0; j < m; ++j) {
        rays[j].org.x = (float)uniform01(); rays[j].org.y = (float)uniform01(); rays[j].org.z = (float)uniform01();
        rays[j].dir.x = (float)(2.0 * uniform01() - 1.0); rays[j].dir.y = (float)(2.0 * uniform01() - 1.0); rays[j].dir.z = (float)(2.0 * uniform01() - 1.0);
        rays[j].tmin = 0.0f; rays[j].tmax = 3.4e38f;
    }

    /* 1. build on device 0 */
    CHECK(bvh_amd_device_select(0) == 0 && bvh_amd_device_current() == 0, "device_select(0)");
    float* d_tris = bvh_amd_device_alloc(n * 9 * sizeof(float));
    float* d_bb = bvh_amd_device_alloc(n * 6 * sizeof(float));
    float* d_cc = bvh_amd_device_alloc(n * 3 * sizeof(float));
    float* d_prims = bvh_amd_device_alloc(n * 12 * sizeof(float));
    CHECK(d_tris && d_bb && d_cc && d_prims, "device_alloc");
    CHECK(bvh_amd_copy_to_device(d_tris, tris, n * 9 * sizeof(float)) == 0, "copy_to_device");
    CHECK(bvh_amd_tri_bounds3f(d_tris, n, d_bb, d_cc, NULL) == 0, "tri_bounds");
    struct bvh_build_config cfg = { .quality = BVH_BUILD_QUALITY_MEDIUM, .min_leaf_size = 1, .max_leaf_size = 8, .parallel_threshold = 1024 };
    struct bvh3f* bvh = bvh3f_build_device(d_bb, d_cc, n, &cfg, BVH_AMD_BUILDER_DEFAULT_PARALLEL, NULL);
    CHECK(bvh, "build_device");
    CHECK(bvh_amd_precompute_tris3f(d_tris, bvh3f_device_prim_ids(bvh), n, d_prims, NULL) == 0, "precompute_tris");
    CHECK(bvh_amd_synchronize(NULL) == 0, "synchronize");
    const size_t stream_bytes = bvh3f_serialize(bvh, NULL, 0);
    unsigned char* want_stream = malloc(stream_bytes);
    CHECK(bvh3f_serialize(bvh, want_stream, stream_bytes) == stream_bytes, "serialize");

    /* the whole batch on device 0 = the yardstick */
    struct bvh_ray3f* d_rays0 = bvh_amd_device_alloc(m * sizeof *rays);
    struct bvh_hit3f* d_hits0 = bvh_amd_device_alloc(m * sizeof(struct bvh_hit3f));
    CHECK(d_rays0 && d_hits0, "device_alloc");
    CHECK(bvh_amd_copy_to_device(d_rays0, rays, m * sizeof *rays) == 0, "copy rays");
    CHECK(bvh3f_intersect_rays_tri(bvh, d_prims, d_rays0, m, BVH_AMD_RAY_ROBUST, d_hits0, NULL, NULL) == 0, "intersect (whole batch)");
    struct bvh_hit3f* want = malloc(m * sizeof *want);
    CHECK(bvh_amd_copy_to_host(want, d_hits0, m * sizeof *want) == 0, "copy hits");
    size_t n_hit = 0;
    for (size_t j = 0; j < m; ++j) n_hit += want[j].prim != BVH_AMD_INVALID;
    CHECK(n_hit > m / 100, "suspiciously few hits (%zu of %zu)", n_hit, m);

    /* 2. one copy per device */
    struct bvh3f* copies[64];
    void* prims_of[64];
    CHECK(bvh3f_replicate(bvh, d_prims, n * 12 * sizeof(float), g, NULL, copies, prims_of) == 0, "bvh3f_replicate over %d devices", g);
    CHECK(bvh_amd_device_current() == 0, "replicate changed the current device");
    CHECK(copies[0] == bvh && prims_of[0] == (void*)d_prims, "the root's entry must hold the original objects");

    /* 3 + 4. every device traces its shard; concatenation == whole batch */
    struct bvh_hit3f* got = malloc(m * sizeof *got);
    const size_t per = (m + (size_t)g - 1) / (size_t)g;
    unsigned char* stream = malloc(stream_bytes);
    for (int k = 0; k < g; ++k) {
        CHECK(bvh_amd_device_select(k) == 0, "device_select(%d)", k);
        CHECK(copies[k] && (prims_of[k] || n == 0), "no copy on device %d", k);
        CHECK(bvh3f_serialize(copies[k], stream, stream_bytes) == stream_bytes && memcmp(stream, want_stream, stream_bytes) == 0,
              "the copy on device %d serializes to another stream than the original", k);
        const size_t b = (size_t)k * per < m ? (size_t)k * per : m, e = b + per < m ? b + per : m;
        if (e == b) continue;
        struct bvh_ray3f* d_rays = bvh_amd_device_alloc((e - b) * sizeof *rays);
        struct bvh_hit3f* d_hits = bvh_amd_device_alloc((e - b) * sizeof(struct bvh_hit3f));
        CHECK(d_rays && d_hits, "device_alloc on device %d", k);
        CHECK(bvh_amd_copy_to_device(d_rays, rays + b, (e - b) * sizeof *rays) == 0, "copy rays to device %d", k);
        CHECK(bvh3f_intersect_rays_tri(copies[k], prims_of[k], d_rays, e - b, BVH_AMD_RAY_ROBUST, d_hits, NULL, NULL) == 0, "intersect on device %d", k);
        CHECK(bvh_amd_copy_to_host(got + b, d_hits, (e - b) * sizeof *got) == 0, "copy hits from device %d", k);
        bvh_amd_device_free(d_rays); bvh_amd_device_free(d_hits);
    }
    CHECK(memcmp(got, want, m * sizeof *got) == 0, "sharded hits differ from the single-device hits");
    for (int k = 1; k < g; ++k) {
        CHECK(bvh_amd_device_select(k) == 0, "device_select(%d)", k);
        bvh3f_destroy(copies[k]);
        bvh_amd_device_free(prims_of[k]);
    }
    CHECK(bvh_amd_device_select(0) == 0, "device_select(0)");

    /* 5. the one-process-per-GPU entry points, communicator of size 1 */
    unsigned char id[BVH_AMD_COMM_ID_BYTES];
    CHECK(bvh_amd_comm_unique_id(id) == 0, "comm_unique_id");
    struct bvh_amd_comm* comm = bvh_amd_comm_create(id, 1, 0);
    CHECK(comm && bvh_amd_comm_rank(comm) == 0 && bvh_amd_comm_size(comm) == 1 && bvh_amd_comm_handle(comm), "comm_create");
    void* out_prims = NULL;
    size_t out_bytes = 0;
    struct bvh3f* mine = bvh3f_broadcast(comm, 0, bvh, d_prims, n * 12 * sizeof(float), &out_prims, &out_bytes, NULL);
    CHECK(mine, "bvh3f_broadcast");
    CHECK(out_bytes == n * 12 * sizeof(float), "broadcast reports %zu primitive bytes", out_bytes);
    if (mine != bvh) {                                         /* BVH_AMD_BROADCAST_LOOPBACK=1: the root ran the receiving side too */
        CHECK(out_prims && out_prims != (void*)d_prims, "loopback must hand out a received primitive array");
        CHECK(bvh3f_serialize(mine, stream, stream_bytes) == stream_bytes && memcmp(stream, want_stream, stream_bytes) == 0, "received stream differs");
        CHECK(bvh3f_intersect_rays_tri(mine, out_prims, d_rays0, m, BVH_AMD_RAY_ROBUST, d_hits0, NULL, NULL) == 0, "intersect (received copy)");
        CHECK(bvh_amd_copy_to_host(got, d_hits0, m * sizeof *got) == 0 && memcmp(got, want, m * sizeof *got) == 0, "received copy traces differently");
        bvh3f_destroy(mine);
        bvh_amd_device_free(out_prims);
    } else {
        CHECK(out_prims == (void*)d_prims, "the root must get its own primitive array back");
    }
    CHECK(bvh3f_broadcast(comm, 0, NULL, NULL, 0, &out_prims, NULL, NULL) == NULL && strstr(bvh_amd_last_error(), "root"), "a root without a BVH must be refused");
    CHECK(bvh_amd_comm_broadcast(comm, d_hits0, 64, 0, NULL) == 0 && bvh_amd_synchronize(NULL) == 0, "comm_broadcast");
    bvh_amd_comm_destroy(comm);

    printf("replicate ok: %zu triangles, %zu rays (%zu hits) over %d device(s), %zu-byte stream, shards == whole batch\n", n, m, n_hit, g, stream_bytes);
    bvh3f_destroy(bvh);
    bvh_amd_device_free(d_tris); bvh_amd_device_free(d_bb); bvh_amd_device_free(d_cc); bvh_amd_device_free(d_prims);
    bvh_amd_device_free(d_rays0); bvh_amd_device_free(d_hits0);
    free(tris); free(rays); free(want); free(got); free(stream); free(want_stream);
    return 0;
}
