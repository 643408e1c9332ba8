/* Per-ray traversal through the reference's callback API (c_api/bvh.h:277-295), all four families, in plain C11 against
 * <bvh/v2/c_api/bvh.h>. The shape of test/c_api_example.c's render loop: build, then for every ray one
 * bvhXX_intersect_ray[_any][_robust] call whose leaf callback tests the ORIGINAL primitives through bvhXX_get_prim_id and
 * shortens the ray. The leaf tests restate tri.h:56-74 (3D triangles) and sphere.h:32-49 (2D circles) so that the output
 * can be compared bit for bit with the oracle's.
 *
 *   ray_callback <3f|3d|2f|2d> <closest|any> <robust 0|1> <in.bin> <out.bin> [threads]
 *   in : u64 n_prims, u64 n_rays, prims (3D: 9 scalars each, 2D: cx cy r), rays (3D: 8 scalars, 2D: 6)
 *   out: per ray {i64 original primitive id or -1, f64 t}, then u64 leaf callbacks made in total
 *   threads > 1: the rays are split over that many pthreads tracing concurrently through ONE bvh (the entry points, and the
 *   accessors the callbacks use, are re-entrant like the reference's)
 */
#include <bvh/v2/c_api/bvh.h>

#include <inttypes.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tgmath.h>
#include <time.h>

static double now_s(void) { struct timespec ts; timespec_get(&ts, TIME_UTC); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

struct record { int64_t prim; double t; };

/* Everything that does not depend on the leaf test: the per-thread loop over a slice of the rays and the driver. The family
 * provides leaf##S (the callback), make_ray##S and prim_bounds##S; USE_POOL selects bvhXX_build with or without a thread pool. */
#define DRIVER(T, S, CB, RAY_STRIDE, PRIM_STRIDE, USE_POOL)                                                          \
    struct job##S { struct bvh##S* bvh; const T* prims; const T* rays; size_t first, last; bool any, robust;         \
                    struct record* records; uint64_t calls; };                                                       \
    static void* work##S(void* arg) {                                                                                \
        struct job##S* job = arg;                                                                                    \
        struct user##S u = { .bvh = job->bvh, .prims = job->prims };                                                  \
        const struct CB callback = { .user_data = &u, .user_fn = leaf##S };                                           \
        for (size_t j = job->first; j < job->last; ++j) {                                                            \
            const T* r = job->rays + RAY_STRIDE * j;                                                                 \
            u.ray = make_ray##S(r);                                                                                  \
            u.prim = -1;                                                                                             \
            if (job->any) {                                                                                          \
                if (job->robust) bvh##S##_intersect_ray_any_robust(job->bvh, &u.ray, &callback);                      \
                else bvh##S##_intersect_ray_any(job->bvh, &u.ray, &callback);                                         \
            } else {                                                                                                 \
                if (job->robust) bvh##S##_intersect_ray_robust(job->bvh, &u.ray, &callback);                          \
                else bvh##S##_intersect_ray(job->bvh, &u.ray, &callback);                                             \
            }                                                                                                        \
            job->records[j] = (struct record) { u.prim, (double)u.ray.tmax };                                        \
        }                                                                                                            \
        job->calls = u.calls;                                                                                        \
        return NULL;                                                                                                 \
    }                                                                                                                \
    static int run##S(bool any, bool robust, int n_threads, size_t n, size_t m, const T* prims, const T* rays, FILE* out) { \
        struct bvh_bbox##S* bb = malloc(n * sizeof *bb);                                                              \
        struct bvh_vec##S* cc = malloc(n * sizeof *cc);                                                               \
        for (size_t i = 0; i < n; ++i) prim_bounds##S(prims + PRIM_STRIDE * i, &bb[i], &cc[i]);                       \
        struct bvh_thread_pool* pool = USE_POOL ? bvh_thread_pool_create(0) : NULL;                                                                      \
        struct bvh##S* bvh = bvh##S##_build(pool, bb, cc, n, NULL);                                                   \
        if (pool) bvh_thread_pool_destroy(pool);                                                                     \
        if (!bvh) return 1;                                                                                          \
        struct record* records = malloc((m ? m : 1) * sizeof *records);                                               \
        struct job##S jobs[64];                                                                                      \
        pthread_t tids[64];                                                                                          \
        if (n_threads < 1) n_threads = 1;                                                                            \
        if (n_threads > 64) n_threads = 64;                                                                          \
        const double t_begin = now_s();                                                                              \
        for (int k = 0; k < n_threads; ++k) {                                                                        \
            jobs[k] = (struct job##S) { .bvh = bvh, .prims = prims, .rays = rays, .first = m * (size_t)k / (size_t)n_threads, \
                                        .last = m * (size_t)(k + 1) / (size_t)n_threads, .any = any, .robust = robust, \
                                        .records = records };                                                        \
            if (n_threads == 1) work##S(&jobs[k]);                                                                   \
            else if (pthread_create(&tids[k], NULL, work##S, &jobs[k])) return 1;                                    \
        }                                                                                                            \
        uint64_t calls = 0;                                                                                          \
        for (int k = 0; k < n_threads; ++k) { if (n_threads > 1) pthread_join(tids[k], NULL); calls += jobs[k].calls; } \
        const double seconds = now_s() - t_begin;                                                                    \
        fwrite(records, sizeof *records, m, out);                                                                    \
        fwrite(&calls, sizeof calls, 1, out);                                                                        \
        printf("%zu nodes, %zu rays, %d thread(s), %" PRIu64 " leaf callbacks, %.1f us per ray\n",                    \
               bvh##S##_get_node_count(bvh), m, n_threads, calls, 1e6 * seconds / (double)(m ? m : 1));               \
        bvh##S##_destroy(bvh); free(records); free(bb); free(cc);                                                     \
        return 0;                                                                                                    \
    }

/* ---- 3D: triangles (tri.h:56-74 on e1 = p0 - p1, e2 = p2 - p0, n = e1 x e2) ---------------------------------------------- */
#define FAMILY3(T, S, CB, EPS)                                                                                       \
    struct user##S { struct bvh##S* bvh; const T* prims; struct bvh_ray##S ray; int64_t prim; uint64_t calls; };      \
    static T dot##S(const T* a, const T* b) { return (((T)0 + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]; }            \
    static void cross##S(const T* a, const T* b, T* out) {                                                           \
        out[0] = a[1] * b[2] - a[2] * b[1]; out[1] = a[2] * b[0] - a[0] * b[2]; out[2] = a[0] * b[1] - a[1] * b[0]; } \
    static bool leaf##S(void* data, T* t, size_t begin, size_t end) {                                                \
        struct user##S* u = data;                                                                                    \
        bool was_hit = false;                                                                                        \
        u->calls++;                                                                                                  \
        for (size_t i = begin; i < end; ++i) {                                                                       \
            const size_t id = bvh##S##_get_prim_id(u->bvh, i);                                                        \
            const T* p = u->prims + 9 * id;                                                                          \
            T e1[3], e2[3], n[3], c[3], r[3];                                                                        \
            const T org[3] = { u->ray.org.x, u->ray.org.y, u->ray.org.z }, dir[3] = { u->ray.dir.x, u->ray.dir.y, u->ray.dir.z }; \
            for (int k = 0; k < 3; ++k) { e1[k] = p[k] - p[3 + k]; e2[k] = p[6 + k] - p[k]; c[k] = p[k] - org[k]; }   \
            cross##S(e1, e2, n);                                                                                     \
            cross##S(dir, c, r);                                                                                     \
            const T inv_det = (T)1 / dot##S(n, dir);                                                                 \
            const T uu = dot##S(r, e2) * inv_det, vv = dot##S(r, e1) * inv_det, ww = (T)1 - uu - vv;                  \
            if (uu >= -(EPS) && vv >= -(EPS) && ww >= -(EPS)) {                                                      \
                const T tt = dot##S(n, c) * inv_det;                                                                 \
                if (tt >= u->ray.tmin && tt <= u->ray.tmax) { *t = u->ray.tmax = tt; u->prim = (int64_t)id; was_hit = true; } \
            }                                                                                                        \
        }                                                                                                            \
        return was_hit;                                                                                              \
    }                                                                                                                \
    static struct bvh_ray##S make_ray##S(const T* r) {                                                               \
        return (struct bvh_ray##S) { { r[0], r[1], r[2] }, { r[3], r[4], r[5] }, r[6], r[7] }; }                      \
    static void prim_bounds##S(const T* p, struct bvh_bbox##S* bb, struct bvh_vec##S* cc) {  /* tri.h:24-25 */        \
        T lo[3], hi[3];                                                                                              \
        for (int k = 0; k < 3; ++k) {                                                                                \
            lo[k] = p[k]; hi[k] = p[k];                                                                              \
            for (int v = 1; v < 3; ++v) { lo[k] = lo[k] < p[3 * v + k] ? lo[k] : p[3 * v + k]; hi[k] = hi[k] > p[3 * v + k] ? hi[k] : p[3 * v + k]; } \
        }                                                                                                            \
        *bb = (struct bvh_bbox##S) { { lo[0], lo[1], lo[2] }, { hi[0], hi[1], hi[2] } };                              \
        *cc = (struct bvh_vec##S) { (p[0] + p[3] + p[6]) * (T)(1. / 3.), (p[1] + p[4] + p[7]) * (T)(1. / 3.),          \
                                    (p[2] + p[5] + p[8]) * (T)(1. / 3.) };                                            \
    }                                                                                                                \
    DRIVER(T, S, CB, 8, 9, 1)

/* ---- 2D: circles (sphere.h:32-49 for Sphere<T, 2>); the 2D families build serially ------------------------------------------- */
#define FAMILY2(T, S, CB)                                                                                            \
    struct user##S { struct bvh##S* bvh; const T* prims; struct bvh_ray##S ray; int64_t prim; uint64_t calls; };      \
    static T dot##S(T a0, T a1, T b0, T b1) { return ((T)0 + a0 * b0) + a1 * b1; }                                    \
    static bool leaf##S(void* data, T* t, size_t begin, size_t end) {                                                \
        struct user##S* u = data;                                                                                    \
        bool was_hit = false;                                                                                        \
        u->calls++;                                                                                                  \
        for (size_t i = begin; i < end; ++i) {                                                                       \
            const size_t id = bvh##S##_get_prim_id(u->bvh, i);                                                        \
            const T* s = u->prims + 3 * id;                                                                          \
            const T o0 = u->ray.org.x - s[0], o1 = u->ray.org.y - s[1];                                               \
            const T a = dot##S(u->ray.dir.x, u->ray.dir.y, u->ray.dir.x, u->ray.dir.y);                               \
            const T b = (T)2 * dot##S(u->ray.dir.x, u->ray.dir.y, o0, o1);                                            \
            const T c = dot##S(o0, o1, o0, o1) - s[2] * s[2];                                                         \
            const T delta = b * b - (T)4 * a * c;                                                                    \
            if (delta >= 0) {                                                                                        \
                const T inv = -(T)0.5 / a, root = sqrt(delta);                                                       \
                const T x0 = (b + root) * inv, x1 = (b - root) * inv;                                                \
                const T t0 = x0 > u->ray.tmin ? x0 : u->ray.tmin, t1 = x1 < u->ray.tmax ? x1 : u->ray.tmax;           \
                if (t0 <= t1) { *t = u->ray.tmax = t0; u->prim = (int64_t)id; was_hit = true; }                       \
            }                                                                                                        \
        }                                                                                                            \
        return was_hit;                                                                                              \
    }                                                                                                                \
    static struct bvh_ray##S make_ray##S(const T* r) { return (struct bvh_ray##S) { { r[0], r[1] }, { r[2], r[3] }, r[4], r[5] }; } \
    static void prim_bounds##S(const T* p, struct bvh_bbox##S* bb, struct bvh_vec##S* cc) {  /* sphere.h:24-27 */     \
        *bb = (struct bvh_bbox##S) { { p[0] - p[2], p[1] - p[2] }, { p[0] + p[2], p[1] + p[2] } };                    \
        *cc = (struct bvh_vec##S) { p[0], p[1] };                                                                     \
    }                                                                                                                \
    DRIVER(T, S, CB, 6, 3, 0)

FAMILY3(float, 3f, bvh_intersect_callbackf, 1.1920928955078125e-07f)
FAMILY3(double, 3d, bvh_intersect_callbackd, 2.220446049250313e-16)
FAMILY2(float, 2f, bvh_intersect_callbackf)
FAMILY2(double, 2d, bvh_intersect_callbackd)

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) {
        fprintf(stderr, "usage: %s <3f|3d|2f|2d> <closest|any> <robust 0|1> <in.bin> <out.bin> [threads]\n", argv[0]);
        return 2;
    }
    const bool any = strcmp(argv[2], "any") == 0, robust = atoi(argv[3]) != 0;
    const int n_threads = argc == 7 ? atoi(argv[6]) : 1;
    FILE* in = fopen(argv[4], "rb");
    if (!in) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
    uint64_t head[2];
    if (fread(head, sizeof head, 1, in) != 1) return 2;
    const bool wide = argv[1][1] == 'd', flat = argv[1][0] == '2';
    const size_t scalar = wide ? 8 : 4, per_prim = flat ? 3 : 9, per_ray = flat ? 6 : 8;
    void* prims = malloc(head[0] * per_prim * scalar);
    void* rays = malloc(head[1] * per_ray * scalar);
    if (fread(prims, per_prim * scalar, head[0], in) != head[0] || fread(rays, per_ray * scalar, head[1], in) != head[1]) return 2;
    fclose(in);
    FILE* out = fopen(argv[5], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    int rc = 2;
    if (!strcmp(argv[1], "3f")) rc = run3f(any, robust, n_threads, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "3d")) rc = run3d(any, robust, n_threads, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "2f")) rc = run2f(any, robust, n_threads, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "2d")) rc = run2d(any, robust, n_threads, head[0], head[1], prims, rays, out);
    fclose(out);
    free(prims); free(rays);
    return rc;
}
