/* Per-ray traversal through the reference's callback API (c_api/bvh.h:277-295), all four families, in plain C11 against
 * <bvh/v2/c_api/bvh.h>. The shape of test/c_api_example.c's render loop: build, then for every ray one
 * bvhXX_intersect_ray[_any][_robust] call whose leaf callback tests the ORIGINAL primitives through bvhXX_get_prim_id and
 * shortens the ray. The leaf tests restate tri.h:56-74 (3D triangles) and sphere.h:32-49 (2D circles) so that the output
 * can be compared bit for bit with the oracle's.
 *
 *   ray_callback <3f|3d|2f|2d> <closest|any> <robust 0|1> <in.bin> <out.bin>
 *   in : u64 n_prims, u64 n_rays, prims (3D: 9 scalars each, 2D: cx cy r), rays (3D: 8 scalars, 2D: 6)
 *   out: per ray {i64 original primitive id or -1, f64 t}, then u64 leaf callbacks made in total
 */
#include <bvh/v2/c_api/bvh.h>

#include <inttypes.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tgmath.h>
#include <time.h>

static double now_s(void) { struct timespec ts; timespec_get(&ts, TIME_UTC); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

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
    static int run##S(bool any, bool robust, size_t n, size_t m, const T* prims, const T* rays, FILE* out) {          \
        struct bvh_bbox##S* bb = malloc(n * sizeof *bb);                                                              \
        struct bvh_vec##S* cc = malloc(n * sizeof *cc);                                                               \
        for (size_t i = 0; i < n; ++i) {                                                                             \
            const T* p = prims + 9 * i;                                                                              \
            T lo[3], hi[3];                                                                                          \
            for (int k = 0; k < 3; ++k) {                                                                            \
                lo[k] = p[k]; hi[k] = p[k];                                                                          \
                for (int v = 1; v < 3; ++v) { lo[k] = lo[k] < p[3 * v + k] ? lo[k] : p[3 * v + k]; hi[k] = hi[k] > p[3 * v + k] ? hi[k] : p[3 * v + k]; } \
            }                                                                                                        \
            bb[i] = (struct bvh_bbox##S) { { lo[0], lo[1], lo[2] }, { hi[0], hi[1], hi[2] } };                        \
            cc[i] = (struct bvh_vec##S) { (p[0] + p[3] + p[6]) * (T)(1. / 3.), (p[1] + p[4] + p[7]) * (T)(1. / 3.),    \
                                          (p[2] + p[5] + p[8]) * (T)(1. / 3.) };                                      \
        }                                                                                                            \
        struct bvh_thread_pool* pool = bvh_thread_pool_create(0);                                                    \
        struct bvh##S* bvh = bvh##S##_build(pool, bb, cc, n, NULL);                                                   \
        bvh_thread_pool_destroy(pool);                                                                               \
        if (!bvh) return 1;                                                                                          \
        struct user##S u = { .bvh = bvh, .prims = prims };                                                            \
        const struct CB callback = { .user_data = &u, .user_fn = leaf##S };                                           \
        const double t_begin = now_s();                                                                              \
        for (size_t j = 0; j < m; ++j) {                                                                             \
            const T* r = rays + 8 * j;                                                                               \
            u.ray = (struct bvh_ray##S) { { r[0], r[1], r[2] }, { r[3], r[4], r[5] }, r[6], r[7] };                   \
            u.prim = -1;                                                                                             \
            if (any) { if (robust) bvh##S##_intersect_ray_any_robust(bvh, &u.ray, &callback); else bvh##S##_intersect_ray_any(bvh, &u.ray, &callback); } \
            else { if (robust) bvh##S##_intersect_ray_robust(bvh, &u.ray, &callback); else bvh##S##_intersect_ray(bvh, &u.ray, &callback); } \
            const double t = (double)u.ray.tmax;                                                                     \
            fwrite(&u.prim, sizeof u.prim, 1, out); fwrite(&t, sizeof t, 1, out);                                     \
        }                                                                                                            \
        fwrite(&u.calls, sizeof u.calls, 1, out);                                                                    \
        printf("%zu nodes, %zu rays, %" PRIu64 " leaf callbacks, %.1f us per ray\n", bvh##S##_get_node_count(bvh), m, u.calls,       \
               1e6 * (now_s() - t_begin) / (double)(m ? m : 1));                     \
        bvh##S##_destroy(bvh); free(bb); free(cc);                                                                    \
        return 0;                                                                                                    \
    }

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
    static int run##S(bool any, bool robust, size_t n, size_t m, const T* prims, const T* rays, FILE* out) {          \
        struct bvh_bbox##S* bb = malloc(n * sizeof *bb);                                                              \
        struct bvh_vec##S* cc = malloc(n * sizeof *cc);                                                               \
        for (size_t i = 0; i < n; ++i) {                                                                             \
            const T* s = prims + 3 * i;                                                                              \
            bb[i] = (struct bvh_bbox##S) { { s[0] - s[2], s[1] - s[2] }, { s[0] + s[2], s[1] + s[2] } };              \
            cc[i] = (struct bvh_vec##S) { s[0], s[1] };                                                               \
        }                                                                                                            \
        struct bvh##S* bvh = bvh##S##_build(NULL, bb, cc, n, NULL);   /* the 2D families build serially */             \
        if (!bvh) return 1;                                                                                          \
        struct user##S u = { .bvh = bvh, .prims = prims };                                                            \
        const struct CB callback = { .user_data = &u, .user_fn = leaf##S };                                           \
        const double t_begin = now_s();                                                                              \
        for (size_t j = 0; j < m; ++j) {                                                                             \
            const T* r = rays + 6 * j;                                                                               \
            u.ray = (struct bvh_ray##S) { { r[0], r[1] }, { r[2], r[3] }, r[4], r[5] };                               \
            u.prim = -1;                                                                                             \
            if (any) { if (robust) bvh##S##_intersect_ray_any_robust(bvh, &u.ray, &callback); else bvh##S##_intersect_ray_any(bvh, &u.ray, &callback); } \
            else { if (robust) bvh##S##_intersect_ray_robust(bvh, &u.ray, &callback); else bvh##S##_intersect_ray(bvh, &u.ray, &callback); } \
            const double t = (double)u.ray.tmax;                                                                     \
            fwrite(&u.prim, sizeof u.prim, 1, out); fwrite(&t, sizeof t, 1, out);                                     \
        }                                                                                                            \
        fwrite(&u.calls, sizeof u.calls, 1, out);                                                                    \
        printf("%zu nodes, %zu rays, %" PRIu64 " leaf callbacks, %.1f us per ray\n", bvh##S##_get_node_count(bvh), m, u.calls,       \
               1e6 * (now_s() - t_begin) / (double)(m ? m : 1));                     \
        bvh##S##_destroy(bvh); free(bb); free(cc);                                                                    \
        return 0;                                                                                                    \
    }

FAMILY3(float, 3f, bvh_intersect_callbackf, 1.1920928955078125e-07f)
FAMILY3(double, 3d, bvh_intersect_callbackd, 2.220446049250313e-16)
FAMILY2(float, 2f, bvh_intersect_callbackf)
FAMILY2(double, 2d, bvh_intersect_callbackd)

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s <3f|3d|2f|2d> <closest|any> <robust 0|1> <in.bin> <out.bin>\n", argv[0]); return 2; }
    const bool any = strcmp(argv[2], "any") == 0, robust = atoi(argv[3]) != 0;
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
    if (!strcmp(argv[1], "3f")) rc = run3f(any, robust, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "3d")) rc = run3d(any, robust, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "2f")) rc = run2f(any, robust, head[0], head[1], prims, rays, out);
    else if (!strcmp(argv[1], "2d")) rc = run2d(any, robust, head[0], head[1], prims, rays, out);
    fclose(out);
    free(prims); free(rays);
    return rc;
}
