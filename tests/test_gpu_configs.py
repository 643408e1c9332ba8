"""-m gpu: BASELINE.json's five configs at their STATED sizes, each held byte for byte to the compiled, unmodified reference
(oracle/_ref; the pinned C++ restatement where that cannot exist). configs[3] (10M triangles + one GPU's 12.5M-ray shard) lives in
tests/test_gpu_10m.py; the other four are here:

  configs[0] test/simple_example.cpp — 2 triangles, 1 ray, float 3D: the reference's known answer
             (primitive 1, distance 1, u = -0, v = 0.5; simple_example.cpp:25-35, :71-93)
  configs[1] Sponza-class mesh (262,144-triangle proxy), binned-SAH build (serial DefaultBuilder, Quality::Low ->
             binned_sah_builder.h:82-156) + exactly 1,000,000 uniform-random closest-hit rays (bvh.h:160-182)
  configs[2] the same tree, 10,000,000 any-hit shadow rays, fast and robust (IsAnyHit: bvh.h:136-150, SATO order from
             top_down_sah_builder.h:119-127)
  configs[4] double-precision 3D BVH over 1,000,000 spheres — DefaultBuilder(thread pool, Quality::High) over Node<double, 3>
             (node.h:18-45: 56-byte nodes, 128-byte pair records on the device) — + 1,000,000 robust closest-hit rays through
             Sphere::intersect (sphere.h:32-49), hits AND traversal counters.

The CPU side runs on the GPU box's host cores (seconds per case)."""
import os

import numpy as np
import pytest

import oracle
from bvh_amd import synth

pytestmark = pytest.mark.gpu

SCALE = float(os.environ.get("BVH_AMD_TEST_CONFIGS_SCALE", "1"))       # < 1 for a quick local run


def _n(x):
    return max(1000, int(x * SCALE))


def _cpu():
    """the compiled reference if it is here, else the restatement (both are pinned to the golden vectors)"""
    return oracle.gpu_checker()


def _threads(cpu):
    return max(1, min(cpu.hardware_threads(), len(os.sched_getaffinity(0))))


def _same_stream(gpu, ref):
    a, b = gpu.serialize(), ref.serialize()
    if a != b:
        assert len(a) == len(b), (len(a), len(b), gpu.node_count, ref.node_count)
        x, y = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
        raise AssertionError(f"streams differ from byte {int(np.flatnonzero(x != y)[0])} of {len(a)}")


def test_config0_simple_example_known_answer():
    import bvh_amd
    tris = np.array([[1, -1, 1, 1, 1, 1, -1, 1, 1], [1, -1, 1, -1, -1, 1, -1, 1, 1]], dtype=np.float32)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    h = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, np.array([[0, 0, 0, 0, 0, 1, 0, 100]], dtype=np.float32)))
    assert int(h["prim"][0]) == 1 and float(h["t"][0]) == 1.0
    assert float(h["u"][0]) == 0.0 and np.signbit(h["u"][0]) and float(h["v"][0]) == 0.5
    cpu = _cpu()
    obb, occ = cpu.prep_tris(tris)
    ob = cpu.build(obb, occ, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=2)
    _same_stream(bvh, ob)


class _Sponza:
    """configs[1] + configs[2] share one tree"""
    def __init__(self):
        import torch
        import bvh_amd
        self.cpu = _cpu()
        self.thr = _threads(self.cpu)
        self.tris = synth.sponza_proxy(262_144)
        self.d_tris = torch.from_numpy(self.tris).cuda()
        d_bb, d_cc = bvh_amd.tri_bounds(self.d_tris)
        self.gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))     # serial: binned SAH
        bb, cc = d_bb.cpu().numpy(), d_cc.cpu().numpy()
        obb, occ = self.cpu.prep_tris(self.tris)
        assert bb.tobytes() == obb.tobytes() and cc.tobytes() == occ.tobytes()
        self.ref = self.cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_LOW)
        self.prims = bvh_amd.precompute_tris(self.d_tris, self.gpu.device_prim_ids())
        self.oprims = self.cpu.precompute_tris(self.tris, self.ref.prim_ids())
        self.lo, self.hi = synth.scene_bounds(self.tris)


@pytest.fixture(scope="module")
def sponza():
    s = _Sponza()
    yield s
    del s


def test_config1_sponza_binned_build_and_1m_closest_rays(sponza):
    import torch
    import bvh_amd
    _same_stream(sponza.gpu, sponza.ref)
    assert sponza.prims.cpu().numpy().tobytes() == sponza.oprims.tobytes()
    n = _n(1_000_000)
    rays = synth.rays_closest(n, sponza.lo, sponza.hi)
    hits, cnt = bvh_amd.intersect(sponza.gpu, sponza.prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True, counters=True)
    rh, rc = sponza.ref.intersect_tri(sponza.oprims, rays, False, True, threads=sponza.thr, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
    assert int((rh["prim"] != oracle.INVALID).sum()) > n // 2
    # the timed variant of the kernel (no counters) returns the same records
    plain = bvh_amd.intersect(sponza.gpu, sponza.prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True)
    assert bvh_amd.hits_to_numpy(plain).tobytes() == rh.tobytes()


@pytest.mark.parametrize("robust", [False, True])
def test_config2_sponza_10m_any_hit_shadow_rays(sponza, robust):
    import torch
    import bvh_amd
    n = _n(10_000_000)
    rays = synth.rays_shadow(n, sponza.lo, sponza.hi)
    d_rays = torch.from_numpy(rays).cuda()
    hits, cnt = bvh_amd.intersect(sponza.gpu, sponza.prims, d_rays, any_hit=True, robust=robust, counters=True)
    rh, rc = sponza.ref.intersect_tri(sponza.oprims, rays, True, robust, threads=sponza.thr, counters=True)
    got = bvh_amd.hits_to_numpy(hits)
    assert got.tobytes() == rh.tobytes()                     # same first-found primitive, not just the same hit / miss flag
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
    occluded = int((rh["prim"] != oracle.INVALID).sum())
    assert n // 20 < occluded < n
    plain = bvh_amd.intersect(sponza.gpu, sponza.prims, d_rays, any_hit=True, robust=robust)
    assert bvh_amd.hits_to_numpy(plain).tobytes() == rh.tobytes()


def test_config4_double_precision_1m_spheres_high_build_and_1m_rays():
    import torch
    import bvh_amd
    cpu = _cpu()
    thr = _threads(cpu)
    n = _n(1_000_000)
    sph = synth.spheres(n)                                   # float64 {center, radius}
    d_sph = torch.from_numpy(sph).cuda()
    d_bb, d_cc = bvh_amd.sphere_bounds(d_sph)
    bb, cc = d_bb.cpu().numpy(), d_cc.cpu().numpy()
    obb, occ = cpu.sphere_bboxes(sph)
    assert bb.tobytes() == obb.tobytes() and cc.tobytes() == occ.tobytes()
    gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    ref = cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=thr)
    assert gpu.dtype == np.float64 and gpu.prim_count == n
    _same_stream(gpu, ref)
    prims = bvh_amd.gather(d_sph, gpu.device_prim_ids())     # spheres in BVH order (simple_example.cpp:57-65 for Sphere)
    oprims = np.ascontiguousarray(sph[ref.prim_ids().astype(np.int64)])
    assert prims.cpu().numpy().tobytes() == oprims.tobytes()
    lo, hi = synth.scene_bounds(sph)
    n_rays = _n(1_000_000)
    rays = synth.rays_closest(n_rays, lo, hi, dtype=np.float64)
    d_rays = torch.from_numpy(rays).cuda()
    hits, cnt = bvh_amd.intersect(gpu, prims, d_rays, any_hit=False, robust=True, leaf="sphere", counters=True)
    rh, rc = ref.intersect_sphere(oprims, rays, False, True, threads=thr, counters=True)
    got = bvh_amd.hits_to_numpy(hits)
    assert got.tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
    assert int((rh["prim"] != oracle.INVALID).sum()) > n_rays // 20
    plain = bvh_amd.intersect(gpu, prims, d_rays, any_hit=False, robust=True, leaf="sphere")
    assert bvh_amd.hits_to_numpy(plain).tobytes() == rh.tobytes()
    # and the alternative pairing of config 5's pieces: any-hit, fast slab test, same spheres
    hits, cnt = bvh_amd.intersect(gpu, prims, d_rays, any_hit=True, robust=False, leaf="sphere", counters=True)
    rh, rc = ref.intersect_sphere(oprims, rays, True, False, threads=thr, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
