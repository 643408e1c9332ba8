/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * Plain-C ABI shared by the two CPU checkers under oracle/:
 *   ref_*  : oracle/ref_harness.cpp  -> oracle/_ref/libbvh_ref.so   (the unmodified reference, compiled in place)
 *   orc_*  : oracle/bvh_oracle.cpp   -> oracle/libbvh_oracle.so     (restatement of the reference's algorithm)
 * Both export the same function set (prefix differs) so tests can diff them call by call.
 *
 * Array conventions (all host memory, tightly packed):
 *   bboxes   n x 6 scalars  {min.x,min.y,min.z,max.x,max.y,max.z}        (bvh::v2::BBox, bbox.h:13)
 *   centers  n x 3 scalars                                                (bvh::v2::Vec,  vec.h:15)
 *   tris9    n x 9 scalars  {p0,p1,p2}                                    (bvh::v2::Tri,  tri.h:16)
 *   tris12   n x 12 scalars {p0,e1,e2,n}                                  (PrecomputedTri, tri.h:31)
 *   sph4     n x 4 scalars  {center,radius}                               (Sphere, sphere.h:15)
 *   rays8    n x 8 scalars  {org,dir,tmin,tmax}                           (Ray, ray.h:16)
 *   nodes    28 B (float) / 56 B (double): bounds[6] = {minx,maxx,miny,maxy,minz,maxz}, index (node.h:31-37)
 */
#ifndef BVH_ORACLE_ABI_H
#define BVH_ORACLE_ABI_H

#include <stddef.h>
#include <stdint.h>

#define ORC_EXPORT __attribute__((visibility("default")))

enum orc_builder {
    ORC_BUILDER_DEFAULT_SERIAL   = 0, /* DefaultBuilder::build(bboxes, centers, cfg)        default_builder.h:49 */
    ORC_BUILDER_DEFAULT_PARALLEL = 1, /* DefaultBuilder::build(pool, bboxes, centers, cfg)  default_builder.h:33 */
    ORC_BUILDER_BINNED           = 2, /* BinnedSahBuilder::build                            binned_sah_builder.h:32 */
    ORC_BUILDER_SWEEP            = 3  /* SweepSahBuilder::build                             sweep_sah_builder.h:30 */
};

enum orc_quality { ORC_QUALITY_LOW = 0, ORC_QUALITY_MEDIUM = 1, ORC_QUALITY_HIGH = 2 };

/* Hit record. prim = BVH-order primitive index i (the index the reference's leaf callback receives,
 * bvh.h:152); original id = prim_ids[prim]. Miss: prim = ORC_INVALID and t = the ray's input tmax.
 * Triangles: (t,u,v) of tri.h:56-74. Spheres: t = t0, u = t1 of sphere.h:32-49, v = 0. */
#define ORC_INVALID 0xFFFFFFFFu
typedef struct { uint32_t prim; float t, u, v; } orc_hitf;
typedef struct { uint32_t prim; uint32_t pad; double t, u, v; } orc_hitd;

#endif
