"""-m gpu, only under BVH_AMD_PAIRS=compact (EXPERIMENTAL switch, read once per process): run the WHOLE GPU suite with the switch set
(`BVH_AMD_PAIRS=compact python -m pytest tests -m gpu`) — every float / 3D batch traversal then goes through trace_kernel_compact and
is held to the same bit-exact bar by the existing tests. This module only makes sure the switch really took effect, and that a
tree the compact records cannot represent silently keeps the PairNode kernel. Skipped (not failed) when the switch is off."""
import os

import numpy as np
import pytest

from bvh_amd import synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("BVH_AMD_PAIRS") != "compact", reason="BVH_AMD_PAIRS=compact not set (experimental path off)")]


def _last_kernel():
    import bvh_amd
    return bvh_amd._lib.load().bvh_amd_last_kernel_name().decode()


def test_compact_kernel_is_the_one_that_runs(orc):
    import bvh_amd
    tris = synth.soup(50000, seed=3)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(200000, lo, hi, seed=5)
    hits, cnt = bvh_amd.intersect(bvh, prims, rays, False, True, counters=True)
    assert "trace_kernel_compact" in _last_kernel()
    ob = orc.from_arrays(bvh.nodes, bvh.prim_ids)
    ref, ref_cnt = ob.intersect_tri(orc.precompute_tris(tris, bvh.prim_ids), rays, 0, 1, threads=8, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == ref.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == ref_cnt).all()


def test_unrepresentable_tree_keeps_pairnode_kernel(orc):
    import bvh_amd
    tris = synth.soup(5000, seed=4)
    bb, cc = orc.prep_tris(tris)
    ob = orc.build(bb, cc, quality=2)
    nodes = ob.nodes().copy()
    victim = int(np.flatnonzero((nodes["index"] & 15) == 0)[3])
    nodes["bounds"][victim, 0] -= 1.0                        # a hand-edited (grown) box: no child shares that plane
    bvh = bvh_amd.Bvh.from_nodes(nodes, ob.prim_ids())
    prims = bvh_amd.precompute_tris(tris, ob.prim_ids().astype(np.int32))
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(20000, lo, hi, seed=6)
    hits = bvh_amd.intersect(bvh, prims, rays, False, True)
    assert "trace_kernel_compact" not in _last_kernel()
    ref = orc.from_arrays(nodes, ob.prim_ids()).intersect_tri(orc.precompute_tris(tris, ob.prim_ids()), rays, 0, 1, threads=8)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == ref.tobytes()


def test_compact_kernel_double_spheres(orc):
    """The double variant (64-byte records against PairNode<double>'s 128): BASELINE configs[4]'s kind of scene, small."""
    import bvh_amd
    sph = synth.spheres(50000)
    bb, cc = bvh_amd.sphere_bounds(sph)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    ids = bvh.prim_ids
    prims = bvh_amd.gather(sph, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(sph)
    rays = synth.rays_closest(100000, lo, hi, seed=7, dtype=np.float64)
    hits, cnt = bvh_amd.intersect(bvh, prims, rays, False, True, leaf="sphere", counters=True)
    assert "trace_kernel_compact_f64" in _last_kernel()
    ob = orc.from_arrays(bvh.nodes, ids)
    ref, ref_cnt = ob.intersect_sphere(sph[ids.astype(np.int64)], rays, 0, 1, threads=8, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == ref.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == ref_cnt).all()
