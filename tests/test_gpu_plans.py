"""-m gpu: every way the library may TRACE a large batch, held to the compiled, unmodified reference (oracle/_ref) at scale
(VERDICT r3 "Missing 3" / "Next 4"): the headline kernel variant — trace_kernel_coop<float, false, true, 0, false>, closest-hit, no
counters, reordered, thresholds 12 / 12, on the 1M-triangle soup's pool-High tree — and each of the four candidate launch plans
{as given, reordered} x {per-lane fetch, quad-cooperative fetch} that launch_traverse (csrc/traverse.hip) measures and picks from.

Per plan, forced through bvh_amd_tuning + the BVH_AMD_RAY_SORTED / _UNSORTED flags: >= 4M closest-hit rays and >= 2M any-hit rays
through the timed (non-Stats) kernel AND through its Stats twin, hit records byte-equal and (Stats) the three traversal counters equal
to the reference's (bvh.h:160-182 over traverse_top_down :125-157; node.h:68-88; tri.h:56-74). Then the bench's exact settled
configuration at its 2^24 rays. The tree itself is the GPU's build, stream memcmp-equal to the reference's DefaultBuilder(pool, High)."""
import os

import numpy as np
import pytest

import oracle
from bvh_amd import synth

pytestmark = pytest.mark.gpu

SCALE = float(os.environ.get("BVH_AMD_TEST_CONFIGS_SCALE", "1"))       # < 1 for a quick local run


def _n(x):
    return max(70_000, int(x * SCALE))


class _Soup:
    def __init__(self):
        import torch
        import bvh_amd
        self.cpu = oracle.gpu_checker()
        self.thr = max(1, min(self.cpu.hardware_threads(), len(os.sched_getaffinity(0))))
        n = max(50_000, int(1_000_000 * SCALE))
        self.tris = synth.soup(n)
        d_tris = torch.from_numpy(self.tris).cuda()
        d_bb, d_cc = bvh_amd.tri_bounds(d_tris)
        self.gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
        bb, cc = d_bb.cpu().numpy(), d_cc.cpu().numpy()
        self.ref = self.cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=self.thr)
        assert self.gpu.serialize() == self.ref.serialize(), "the GPU's pool-High tree differs from the reference's"
        self.prims = bvh_amd.precompute_tris(d_tris, self.gpu.device_prim_ids())
        self.oprims = self.cpu.precompute_tris(self.tris, self.ref.prim_ids())
        lo, hi = synth.scene_bounds(self.tris)
        self.closest = synth.rays_closest(_n(4_194_304), lo, hi, seed=77)
        self.shadow = synth.rays_shadow(_n(2_097_152), lo, hi, seed=78)
        self.want = {}
        for any_hit, rays in ((False, self.closest), (True, self.shadow)):
            self.want[any_hit] = self.ref.intersect_tri(self.oprims, rays, any_hit, True, threads=self.thr, counters=True)
        self.lo, self.hi = lo, hi


@pytest.fixture(scope="module")
def soup():
    s = _Soup()
    yield s
    del s


PLANS = [("as_given_per_lane", False, 0, 36, 12), ("as_given_coop", False, 1, 12, 12), ("reordered_per_lane", True, 0, 36, 12), ("reordered_coop", True, 1, 12, 12)]


@pytest.mark.parametrize("name,reorder,coop,refill,leaf", PLANS, ids=[p[0] for p in PLANS])
def test_each_launch_plan_equals_the_reference_at_scale(soup, name, reorder, coop, refill, leaf):
    import ctypes as C
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    plan = (C.c_int * 4)()
    try:
        lib.bvh_amd_tuning(refill, leaf, coop, -1)
        for any_hit, rays in ((False, soup.closest), (True, soup.shadow)):
            want_hits, want_cnt = soup.want[any_hit]
            d_rays = torch.from_numpy(rays).cuda()
            # the timed kernel (no counters) ...
            got = bvh_amd.intersect(soup.gpu, soup.prims, d_rays, any_hit=any_hit, robust=True, sort_rays=reorder)
            lib.bvh_amd_last_launch_plan(plan)
            kernel = lib.bvh_amd_last_kernel_name().decode()
            assert [plan[0], plan[1], plan[2], plan[3]] == [int(reorder), coop, refill, leaf], (name, list(plan))
            assert kernel.startswith("trace_kernel_coop<float" if coop else "trace_kernel<float") and kernel.endswith("false>" if coop else "false, 3, false>"), kernel
            assert bvh_amd.hits_to_numpy(got).tobytes() == want_hits.tobytes(), f"{name}, any_hit={any_hit}: hit records differ from the reference's"
            # ... and its Stats twin under the same plan: hits and the reference's own counters (benchmark.cpp:281-296)
            got, cnt = bvh_amd.intersect(soup.gpu, soup.prims, d_rays, any_hit=any_hit, robust=True, sort_rays=reorder, counters=True)
            lib.bvh_amd_last_launch_plan(plan)
            assert [plan[0], plan[1]] == [int(reorder), coop], (name, list(plan))
            assert bvh_amd.hits_to_numpy(got).tobytes() == want_hits.tobytes()
            assert (cnt.cpu().numpy().astype(np.uint64) == want_cnt).all(), (name, any_hit, cnt.cpu().numpy(), want_cnt)
    finally:
        lib.bvh_amd_tuning(-1, -1, -1, -1)


def test_the_bench_configuration_equals_the_reference(soup):
    """bench.py's default line: 2^24 uniform closest-hit rays (seed 1234), robust, the plan the library settles on for this tree
    (reordered, cooperative fetch, 12 / 12), traced by the non-Stats kernel — against the reference on every ray."""
    import ctypes as C
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    n = _n(1 << 24)
    rays = synth.rays_closest(n, soup.lo, soup.hi, seed=1234)
    d_rays = torch.from_numpy(rays).cuda()
    want = soup.ref.intersect_tri(soup.oprims, rays, False, True, threads=soup.thr)
    plan = (C.c_int * 4)()
    try:
        lib.bvh_amd_tuning(12, 12, 1, -1)
        # round 5: the plan search may also settle on "long rays first" (one chord-class bit in the reordering key, plan[0] == 2), and every
        # launch drains staggered by default — both orders of tracing, with and without the stagger, against the reference on every ray
        for classes in (0, 1):
            for stagger in (-1, 0):
                lib.bvh_amd_experiment(b"key_class_bits", classes)
                lib.bvh_amd_experiment(b"stagger", stagger)
                got = bvh_amd.intersect(soup.gpu, soup.prims, d_rays, any_hit=False, robust=True, sort_rays=True)
                lib.bvh_amd_last_launch_plan(plan)
                assert list(plan) == [1 + classes, 1, 12, 12]
                assert lib.bvh_amd_last_kernel_name().decode() == "trace_kernel_coop<float, false, true, 0, false>"
                assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes(), (classes, stagger)
    finally:
        lib.bvh_amd_experiment(b"reset", 0)
        lib.bvh_amd_tuning(-1, -1, -1, -1)
    if SCALE >= 1:
        assert int((want["prim"] != oracle.INVALID).sum()) > n // 2


def test_one_shot_and_persistent_grids_equal_the_reference(soup):
    """Round 5: batches of up to 2^18 rays are traced by a one-shot grid (wave w traces rays 64 w .. 64 w + 63 and leaves: no ticket
    counter, no refill), larger ones by the persistent grid with a staggered drain — both forced here on the same 200k-ray batch, per-lane
    and cooperative fetch, closest and any-hit, against the reference's hit records (bvh.h:125-182)."""
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    n = min(200_000, len(soup.closest), len(soup.shadow))
    try:
        for any_hit, rays in ((False, soup.closest), (True, soup.shadow)):
            want = soup.want[any_hit][0][:n]
            d_rays = torch.from_numpy(rays[:n]).cuda()
            for one_shot in (1, 0):
                for coop in (0, 1):
                    for stagger in ((-1,) if one_shot else (-1, 0, 4096)):
                        lib.bvh_amd_experiment(b"one_shot", one_shot)
                        lib.bvh_amd_experiment(b"stagger", stagger)
                        lib.bvh_amd_tuning(-1, -1, coop, -1)
                        got = bvh_amd.intersect(soup.gpu, soup.prims, d_rays, any_hit=any_hit, robust=True)
                        assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes(), (any_hit, one_shot, coop, stagger)
    finally:
        lib.bvh_amd_experiment(b"reset", 0)
        lib.bvh_amd_tuning(-1, -1, -1, -1)


def test_first_large_batch_uses_the_predicted_plan(soup):
    """VERDICT r3 Weak 4: the first >= 2^20-ray batch through a fresh tree is traced the way the predictor says (this tree: beyond the
    L2s, >= 100 expected record fetches of a random line -> reordered with the long rays first, cooperative fetch), not with the search's old candidate 0
    (as given, per lane); the search explores the other plans from the second batch on, and hits never depend on any of it."""
    import ctypes as C
    import torch
    import bvh_amd
    if SCALE < 1:
        pytest.skip("needs the full 1M-triangle tree (beyond the L2s) to be a candidate for the plan search")
    lib = bvh_amd._lib.load()
    d_tris = torch.from_numpy(soup.tris).cuda()
    d_bb, d_cc = bvh_amd.tri_bounds(d_tris)
    fresh = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    d_rays = torch.from_numpy(soup.closest).cuda()
    plan = (C.c_int * 4)()
    seen = []
    for i in range(12):
        got = bvh_amd.intersect(fresh, soup.prims, d_rays, any_hit=False, robust=True)
        torch.cuda.synchronize()
        lib.bvh_amd_last_launch_plan(plan)
        seen.append((plan[0], plan[1]))
        assert bvh_amd.hits_to_numpy(got).tobytes() == soup.want[False][0].tobytes(), i
    assert seen[0] == (2, 1), seen                            # (round 6: long rays first on a heavy tree that fits the Infinity Cache)
    # round 5: the search prunes — the other ray order with the same fetch is tried second and takes its family with it when it loses by
    # > 40 % (it does on this tree), survivors are measured twice, at most eight batches are spent; round 6: it ends after four batches
    # (plan, as given, sibling, plan again) when the better of plan and sibling leads the rest by more than 10 %
    assert seen[1] == (0, 1), seen
    assert (1, 1) in seen[:4], seen                           # the predicted plan's sibling (the other order of the long rays) was measured ...
    assert seen[10] == seen[11] and seen[10] in seen[:9], seen   # ... and the search has settled (<= 9 batches, photo finish included) on a plan it measured
    assert seen[10][0] >= 1, seen                              # (reordered — 2 = with the long rays first: the as-given family loses by a wide margin on this tree)


# ---- the cooperative fetch of the other record families (round 4: trace_kernel_coop_nd) --------------------------------------------

def _forced(lib, coop, refill, leaf):
    lib.bvh_amd_tuning(refill, leaf, coop, -1)


@pytest.mark.parametrize("leaf", ["sphere", "tri"])
def test_double_precision_cooperative_fetch_equals_the_reference(leaf):
    """Node<double, 3> (node.h:18-45; 128-byte pair records fetched as two quad-coalesced 64-byte halves, trace_device.h): spheres
    (sphere.h:32-49) and triangles (tri.h:56-74), closest + any, robust + fast, per-lane against cooperative fetch — hit records and
    counters byte-equal to the compiled reference."""
    import torch
    import bvh_amd
    cpu = oracle.gpu_checker()
    thr = max(1, min(cpu.hardware_threads(), len(os.sched_getaffinity(0))))
    lib = bvh_amd._lib.load()
    n = max(30_000, int(300_000 * SCALE))
    if leaf == "sphere":
        prim = synth.spheres(n)
        d_bb, d_cc = bvh_amd.sphere_bounds(torch.from_numpy(prim).cuda())
    else:
        prim = synth.soup(n, seed=11, jitter=0.01, dtype=np.float64)
        d_bb, d_cc = bvh_amd.tri_bounds(torch.from_numpy(prim).cuda())
    gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    ref = cpu.build(d_bb.cpu().numpy(), d_cc.cpu().numpy(), builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM, threads=thr)
    assert gpu.serialize() == ref.serialize()
    if leaf == "sphere":
        d_prims = bvh_amd.gather(torch.from_numpy(prim).cuda(), gpu.device_prim_ids())
        oprims = np.ascontiguousarray(prim[ref.prim_ids().astype(np.int64)])
        trace = ref.intersect_sphere
    else:
        d_prims = bvh_amd.precompute_tris(torch.from_numpy(prim).cuda(), gpu.device_prim_ids())
        oprims = cpu.precompute_tris(prim, ref.prim_ids())
        trace = ref.intersect_tri
    lo, hi = synth.scene_bounds(prim)
    nr = _n(1_000_000)
    batches = {False: synth.rays_closest(nr, lo, hi, dtype=np.float64, seed=5), True: synth.rays_shadow(nr, lo, hi, dtype=np.float64, seed=6)}
    try:
        for any_hit, rays in batches.items():
            d_rays = torch.from_numpy(rays).cuda()
            for robust in (True, False):
                want_hits, want_cnt = trace(oprims, rays, any_hit, robust, threads=thr, counters=True)
                for coop, refill, thr_leaf in ((0, 36, 12), (1, 12, 12), (1, 20, 20)):
                    _forced(lib, coop, refill, thr_leaf)
                    got = bvh_amd.intersect(gpu, d_prims, d_rays, any_hit=any_hit, robust=robust, leaf=leaf, sort_rays=False)
                    kernel = lib.bvh_amd_last_kernel_name().decode()
                    assert kernel.startswith("trace_kernel_coop_nd<double" if coop else "trace_kernel<double"), kernel
                    assert bvh_amd.hits_to_numpy(got).tobytes() == want_hits.tobytes(), (leaf, any_hit, robust, coop)
                    got, cnt = bvh_amd.intersect(gpu, d_prims, d_rays, any_hit=any_hit, robust=robust, leaf=leaf, counters=True, sort_rays=False)
                    assert bvh_amd.hits_to_numpy(got).tobytes() == want_hits.tobytes()
                    assert (cnt.cpu().numpy().astype(np.uint64) == want_cnt).all(), (leaf, any_hit, robust, coop)
            # a reordered batch through the cooperative kernel (the order never changes a record)
            _forced(lib, 1, 12, 12)
            want_hits = trace(oprims, rays, any_hit, True, threads=thr)
            got = bvh_amd.intersect(gpu, d_prims, d_rays, any_hit=any_hit, robust=True, leaf=leaf, sort_rays=True)
            assert bvh_amd.hits_to_numpy(got).tobytes() == want_hits.tobytes()
    finally:
        lib.bvh_amd_tuning(-1, -1, -1, -1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_cooperative_fetch_equals_the_reference(dtype):
    """Node<T, 2> (circles; sphere.h:32-49 over two axes): the cooperative kernels of the 2f / 2d families against the per-lane ones
    and the compiled reference, hits and counters."""
    import bvh_amd
    cpu = oracle.gpu_checker()
    lib = bvh_amd._lib.load()
    n = max(20_000, int(200_000 * SCALE))
    rng = np.random.default_rng(9)
    circ = np.ascontiguousarray(np.concatenate([rng.random((n, 2)), 0.0004 + 0.002 * rng.random((n, 1))], axis=1).astype(dtype))
    bb, cc = cpu.sphere_bboxes(circ)
    ref = cpu.build(bb, cc, quality=2)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High))
    assert gpu.dim == 2 and gpu.serialize() == ref.serialize()
    ordered = circ[ref.prim_ids().astype(np.int64)]
    d_ordered = bvh_amd.gather(circ, gpu.device_prim_ids())
    nr = _n(500_000)
    org = rng.random((nr, 2)) * 1.1 - 0.05
    ang = rng.random(nr) * 2 * np.pi
    rays = np.ascontiguousarray(np.concatenate([org, np.cos(ang)[:, None], np.sin(ang)[:, None], np.zeros((nr, 1)), np.full((nr, 1), np.finfo(dtype).max)], axis=1).astype(dtype))
    rays[:50, 2:4] = 0
    try:
        for any_hit in (False, True):
            for robust in (False, True):
                want, cw = ref.intersect_sphere(ordered, rays, any_hit, robust, threads=8, counters=True)
                for coop, refill, thr_leaf in ((0, 36, 12), (1, 12, 12)):
                    _forced(lib, coop, refill, thr_leaf)
                    got, cg = bvh_amd.intersect(gpu, d_ordered, rays, any_hit=any_hit, robust=robust, counters=True)
                    assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes(), (any_hit, robust, coop)
                    assert (cg.cpu().numpy().astype(np.uint64) == cw).all()
                    got = bvh_amd.intersect(gpu, d_ordered, rays, any_hit=any_hit, robust=robust)
                    kernel = lib.bvh_amd_last_kernel_name().decode()
                    assert kernel.startswith("trace_kernel_coop_nd<" if coop else "trace_kernel<") and kernel.rstrip(">").split(", ")[5 if coop else 5] == "2", kernel
                    assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes()
    finally:
        lib.bvh_amd_tuning(-1, -1, -1, -1)
