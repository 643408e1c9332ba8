"""bvh_amd — MI355X-native BVH construction and traversal behind madmann91/bvh's API.

Host-side mirror of the reference's interface for the hot path (names follow bvh::v2):

    DefaultBuilder.build(bboxes, centers, Config(quality=Quality.High), thread_pool=None) -> Bvh
    BinnedSahBuilder.build / SweepSahBuilder.build
    Bvh.nodes / Bvh.prim_ids / Bvh.serialize() / Bvh.deserialize()
    intersect(bvh, prims, rays, any_hit=False, robust=False) -> hits            (batched Bvh::intersect)
    tri_bounds / precompute_tris / sphere_bounds                                (Tri::get_bbox, PrecomputedTri)

Everything computes in hand-written HIP kernels through the C-ABI of libbvh_amd.so
(include/bvh_amd.h); torch is only used for device memory and streams. There is no CPU fallback.
"""
from .api import (BinnedSahBuilder, MiniTreeBuilder, SplitHeuristic, Bvh, Config, DefaultBuilder, Quality, RayFlags, SweepSahBuilder, ThreadPool, prepare_trace,  # noqa: F401
                  HITD, HITF, INVALID, NODED, NODEF, NODE2D, NODE2F, hits_to_numpy, intersect, precompute_tris, sphere_bounds,
                  tri_bounds, gather, std_sort_ids, radix_sort_pairs, reinsertion_stats, last_optimize_profile, pinhole_rays, shade_eyelight)
from ._lib import BvhAmdError  # noqa: F401
