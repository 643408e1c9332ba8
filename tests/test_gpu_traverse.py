"""-m gpu: the HIP traversal path (through the C-ABI) against the oracle and the golden vectors.
Bar: bit-exact hit records (prim, t, u, v) — stricter than north_star's 1e-5 relative on t, because the
kernel replicates the reference's visit order and arithmetic."""
import numpy as np
import pytest

import oracle
from bvh_amd import synth
from conftest import load_golden, parse_stream

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5   # north_star's tolerance for float traversal; asserted in addition to bit equality


def _check_hits(gpu_hits, ref_hits):
    assert (gpu_hits["prim"] == ref_hits["prim"]).all()
    hit = ref_hits["prim"] != oracle.INVALID
    assert np.all(np.abs(gpu_hits["t"][hit] - ref_hits["t"][hit]) <= REL_TOL * np.abs(ref_hits["t"][hit]))
    assert gpu_hits.tobytes() == ref_hits.tobytes()


@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k", "soup2k_f64", "spheres2k_f64"])
@pytest.mark.parametrize("mode", ["serial_low", "parallel_high"])
@pytest.mark.parametrize("any_hit", [False, True])
@pytest.mark.parametrize("robust", [False, True])
def test_golden_hits(scene, mode, any_hit, robust):
    import bvh_amd
    g = load_golden(scene)
    double = g["prims"].dtype == np.float64
    nodes, ids = parse_stream(g[f"bvh_{mode}"].tobytes(), double)
    bvh = bvh_amd.Bvh.from_nodes(nodes, ids)
    assert bvh.serialize() == g[f"bvh_{mode}"].tobytes()
    rays = g["rays_shadow"] if any_hit else g["rays_closest"]
    key = f"{mode}_{'any' if any_hit else 'closest'}_{'robust' if robust else 'fast'}"
    if "spheres" in scene:
        prims = bvh_amd.gather(g["prims"], ids.astype(np.int32))
        hits, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf="sphere", counters=True)
    else:
        prims = bvh_amd.precompute_tris(g["prims"], ids.astype(np.int32))
        hits, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, counters=True)
    _check_hits(bvh_amd.hits_to_numpy(hits), g[f"hits_{key}"])
    assert (cnt.cpu().numpy().astype(np.uint64) == g[f"counters_{key}"]).all()
    hits2 = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf="sphere" if "spheres" in scene else "tri")
    assert bvh_amd.hits_to_numpy(hits2).tobytes() == g[f"hits_{key}"].tobytes()
    # the batch repeated until it is long enough to be reordered inside the call (every scalar type / leaf / variant): same records
    reps = 4096 // len(rays) + 2
    many = np.tile(rays, (reps, 1))
    hits3 = bvh_amd.intersect(bvh, prims, many, any_hit, robust, leaf="sphere" if "spheres" in scene else "tri", sort_rays=True)
    assert bvh_amd._lib.load().bvh_amd_last_launch_reordered() == 1
    assert bvh_amd.hits_to_numpy(hits3).tobytes() == np.tile(g[f"hits_{key}"], reps).tobytes()


def test_prep_kernels_match_oracle(orc):
    import bvh_amd
    for dt in (np.float32, np.float64):
        tris = synth.soup(5000, seed=2, jitter=0.02, dtype=dt)
        bb, cc = bvh_amd.tri_bounds(tris)
        obb, occ = orc.prep_tris(tris)
        assert bb.cpu().numpy().tobytes() == obb.tobytes() and cc.cpu().numpy().tobytes() == occ.tobytes()
        perm = np.random.default_rng(0).permutation(5000).astype(np.uint64)
        pt = bvh_amd.precompute_tris(tris, perm.astype(np.int32))
        assert pt.cpu().numpy().tobytes() == orc.precompute_tris(tris, perm).tobytes()
        sph = synth.spheres(3000, dtype=dt)
        sb, sc = bvh_amd.sphere_bounds(sph)
        osb, osc = orc.sphere_bboxes(sph)
        assert sb.cpu().numpy().tobytes() == osb.tobytes() and sc.cpu().numpy().tobytes() == osc.tobytes()


def test_reference_known_answers_on_gpu(orc):
    """simple_example: primitive 1, t = 1, (u, v) = (-0, 0.5); cornell render: 1,027,152 hits."""
    import bvh_amd
    ka = load_golden("known_answers")
    tris = ka["simple_tris"]
    bb, cc = orc.prep_tris(tris)
    ob = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
    bvh = bvh_amd.Bvh.from_nodes(ob.nodes(), ob.prim_ids())
    hit = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, bvh_amd.precompute_tris(tris, ob.prim_ids().astype(np.int32)),
                                                  ka["simple_ray"]))
    assert hit.tobytes() == ka["simple_hit"].tobytes()

    g = load_golden("cornell")
    nodes, ids = parse_stream(g["bvh_parallel_high"].tobytes())
    bvh = bvh_amd.Bvh.from_nodes(nodes, ids)
    assert bvh.node_count == 37
    rays = synth.rays_pinhole(1024, 1024, (0, 1, 2), (0, 0, -1), (0, 1, 0))
    prims = bvh_amd.precompute_tris(g["prims"], ids.astype(np.int32))
    hits, cnt = bvh_amd.intersect(bvh, prims, rays, counters=True)
    h = bvh_amd.hits_to_numpy(hits)
    assert int((h["prim"] != oracle.INVALID).sum()) == 1027152
    cnt = cnt.cpu().numpy()
    assert cnt[0] == 7632318 and cnt[2] == 1445436
    assert (cnt.astype(np.uint64) == ka["cornell_render_counters_high"]).all()


@pytest.mark.parametrize("scene,n", [("soup", 200_000), ("terrain", 200_000), ("sponza", 262_144)])
def test_seeded_scene_matches_oracle(orc, scene, n):
    import bvh_amd
    tris = {"soup": lambda: synth.soup(n, jitter=0.01), "terrain": lambda: synth.terrain(n),
            "sponza": lambda: synth.sponza_proxy(n)}[scene]()
    bb, cc = orc.prep_tris(tris)
    ob = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
    ids = ob.prim_ids()
    bvh = bvh_amd.Bvh.from_nodes(ob.nodes(), ids)
    prims = bvh_amd.precompute_tris(tris, ids.astype(np.int32))
    oprims = orc.precompute_tris(tris, ids)
    lo, hi = synth.scene_bounds(tris)
    nr = 200_000
    for any_hit, rays in ((False, synth.rays_closest(nr, lo, hi)), (True, synth.rays_shadow(nr, lo, hi))):
        for robust in (False, True):
            hits, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, counters=True)
            oh, oc = ob.intersect_tri(oprims, rays, any_hit, robust, threads=8, counters=True)
            _check_hits(bvh_amd.hits_to_numpy(hits), oh)
            assert (cnt.cpu().numpy().astype(np.uint64) == oc).all()


def test_edge_cases(orc):
    import bvh_amd
    import torch
    # single-primitive BVH: the root is a leaf and its box is never tested (bvh.h:128-131)
    tri = np.array([[0, 0, 1, 1, 0, 1, 0, 1, 1]], dtype=np.float32)
    bb, cc = orc.prep_tris(tri)
    ob = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    assert ob.node_count == 1
    bvh = bvh_amd.Bvh.from_nodes(ob.nodes(), ob.prim_ids())
    rays = np.array([[0.2, 0.2, 0, 0, 0, 1, 0, 10], [5, 5, 0, 0, 0, 1, 0, 10], [0.2, 0.2, 0, 0, 0, 1, 0, 0.5]], dtype=np.float32)
    pt = bvh_amd.precompute_tris(tri)
    h = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, pt, rays))
    oh = ob.intersect_tri(orc.precompute_tris(tri), rays, 0, 0)
    assert h.tobytes() == oh.tobytes() and h["prim"][0] == 0 and h["prim"][1] == oracle.INVALID
    # empty batch
    out = bvh_amd.intersect(bvh, pt, np.zeros((0, 8), np.float32))
    assert out.shape[0] == 0
    # ragged batch sizes around the wave/block granularity, axis-parallel and degenerate directions
    tris = synth.soup(3000, seed=4, jitter=0.05)
    bb, cc = orc.prep_tris(tris)
    ob = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
    ids = ob.prim_ids()
    bvh = bvh_amd.Bvh.from_nodes(ob.nodes(), ids)
    prims = bvh_amd.precompute_tris(tris, ids.astype(np.int32))
    oprims = orc.precompute_tris(tris, ids)
    lo, hi = synth.scene_bounds(tris)
    base = synth.rays_closest(5000, lo, hi, seed=99)
    base[::7, 3] = 0.0            # zero direction components: inf/NaN slabs (Ize's robust case)
    base[::11, 4] = -0.0
    base[::13, 3:6] = [0, 0, 1]
    base[::17, 3:6] = 0.0         # fully degenerate direction
    base[5::19, 6] = np.nan       # NaN tmin / tmax: every node test fails (robust_max keeps a NaN accumulator, utils.h:41-43)
    base[3::23, 7] = np.nan
    base[2::29, 0] = np.nan       # NaN origin / direction: every slab is NaN, every node is visited
    base[4::31, 4] = np.nan
    base[6::37, 6] = np.inf
    base[8::41, 7] = -np.inf
    for n in (1, 63, 64, 65, 255, 257, 4097):
        for robust in (False, True):
            for any_hit in (False, True):
                got, cg = bvh_amd.intersect(bvh, prims, base[:n], any_hit, robust, counters=True)
                oh, cw = ob.intersect_tri(oprims, base[:n], any_hit, robust, counters=True)
                assert bvh_amd.hits_to_numpy(got).tobytes() == oh.tobytes(), (n, robust, any_hit)
                assert (cg.cpu().numpy().astype(np.uint64) == cw).all(), (n, robust, any_hit)
    torch.cuda.synchronize()


def test_cornell_render_ppm_md5(orc):
    """End to end: the reference's ctest `benchmark cornell_box.obj --eye 0 1 2 --dir 0 0 -1 --up 0 1 0` writes render.ppm with
    md5 96f6bbdc03d7f750fdb833993e9f8538 for every quality. Here: device build (pool, High), device
    traversal (fast, closest), then the eyelight shading of test/benchmark.cpp:363-371 and its PPM writer (:250-255)."""
    import hashlib
    import bvh_amd
    g = load_golden("cornell")
    tris = g["prims"]
    bb, cc = bvh_amd.tri_bounds(tris)
    for q in (bvh_amd.Quality.Low, bvh_amd.Quality.High):
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=q), thread_pool=bvh_amd.ThreadPool())
        ids = bvh.prim_ids
        prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
        W = H = 1024
        rays = synth.rays_pinhole(W, H, (0, 1, 2), (0, 0, -1), (0, 1, 0))
        h = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=False))
        hit = h["prim"] != bvh_amd.INVALID
        assert int(hit.sum()) == 1027152
        # eyelight: |dot(normalize(tri.n), ray.dir)| with the reference's float order of operations
        f = np.float32
        pt = orc.precompute_tris(tris)                                  # original order; n = columns 9..11
        orig = ids[np.minimum(h["prim"], len(ids) - 1)].astype(np.int64)
        n = pt[orig][:, 9:12].astype(f)
        ln = np.sqrt(((f(0) + n[:, 0] * n[:, 0]) + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2], dtype=f)
        nn = n * (f(1) / ln)[:, None]
        d = rays[:, 3:6]
        dot = ((f(0) + nn[:, 0] * d[:, 0]) + nn[:, 1] * d[:, 1]) + nn[:, 2] * d[:, 2]
        inten = np.where(hit, np.abs(dot), f(0)).astype(f)
        pix = np.minimum(np.maximum(0, (inten * f(256)).astype(np.int32)), 255).astype(np.uint8)
        img = np.repeat(pix.reshape(H, W, 1), 3, axis=2)
        ppm = f"P6 {W} {H} 255\n".encode() + img[::-1].tobytes()        # rows written from j = height down to 1
        assert hashlib.md5(ppm).hexdigest() == "96f6bbdc03d7f750fdb833993e9f8538"
        # the same with the camera and the shading on the device: nothing but the PPM rows leaves HBM
        d_rays = bvh_amd.pinhole_rays(W, H, (0, 1, 2), (0, 0, -1), (0, 1, 0))
        assert d_rays.cpu().numpy().tobytes() == rays.tobytes()
        d_hits = bvh_amd.intersect(bvh, prims, d_rays, any_hit=False, robust=False)
        d_img = bvh_amd.shade_eyelight(prims, d_rays, d_hits).cpu().numpy().reshape(H, W, 3)
        assert hashlib.md5(f"P6 {W} {H} 255\n".encode() + d_img[::-1].tobytes()).hexdigest() == "96f6bbdc03d7f750fdb833993e9f8538"


@pytest.mark.parametrize("depth", [60, 64, 65, 300, 3000])
def test_trees_deeper_than_the_small_stack(restatement, depth):
    """The reference's examples and C API trace with SmallStack<Index, 64>; its GrowingStack (stack.h:34-46) has no limit (the
    checker here is the restatement: oracle/_ref's harness uses the SmallStack and cannot walk these trees).
    A chain-shaped tree whose inner child is always nearer than its leaf sibling fills the stack one entry per level: the
    device traversal must give the growing-stack results (LDS + scratch up to 64 entries, HBM spill beyond)."""
    import bvh_amd
    n = depth + 1                                             # leaves; deepest level = depth
    tris = np.zeros((n, 9), dtype=np.float32)
    for k in range(n):
        x = np.float32(4000 - k)
        tris[k] = [x, -1, -1, x, 1, -1, x, 0, 1]
    bb, _ = restatement.prep_tris(tris)
    nodes = np.zeros(2 * n - 1, dtype=oracle.NODEF)
    suffix = bb.copy()                                        # suffix[k] = union of the boxes of leaves k..n-1
    for k in range(n - 2, -1, -1):
        suffix[k, :3] = np.minimum(bb[k, :3], suffix[k + 1, :3])
        suffix[k, 3:] = np.maximum(bb[k, 3:], suffix[k + 1, 3:])
    box = lambda b: [b[0], b[3], b[1], b[4], b[2], b[5]]
    nodes[0]["bounds"], nodes[0]["index"] = box(suffix[0]), 1 << 4
    for k in range(n - 1):
        leaf, rest = 2 * k + 1, 2 * k + 2
        nodes[leaf]["bounds"], nodes[leaf]["index"] = box(bb[k]), (k << 4) | 1
        if k == n - 2:
            nodes[rest]["bounds"], nodes[rest]["index"] = box(bb[n - 1]), ((n - 1) << 4) | 1
        else:
            nodes[rest]["bounds"], nodes[rest]["index"] = box(suffix[k + 1]), (2 * k + 3) << 4
    ids = np.arange(n, dtype=np.uint64)
    ref = restatement.from_arrays(nodes, ids)
    gpu = bvh_amd.Bvh.from_nodes(nodes, ids)
    prims = restatement.precompute_tris(tris)
    rng = np.random.default_rng(depth)
    rays = np.zeros((70_000, 8), dtype=np.float32)
    rays[:, 0] = rng.random(len(rays)) * 100                   # origins in front of the stack of triangles
    rays[:, 1:3] = (rng.random((len(rays), 2)) - 0.5) * 1.5
    rays[:, 3] = 1
    rays[:, 4:6] = (rng.random((len(rays), 2)) - 0.5) * 1e-4
    rays[:, 7] = np.finfo(np.float32).max
    rays[::7, 3] = -1                                          # some point away
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_tri(prims, rays, any_hit, robust, threads=8, counters=True)
            got, cg = bvh_amd.intersect(gpu, prims, rays, any_hit=any_hit, robust=robust, counters=True)
            assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes(), (depth, any_hit, robust)
            assert (cg.cpu().numpy().astype(np.uint64) == cw).all()
    assert int((want["prim"] != oracle.INVALID).sum()) > 10_000


def test_concurrent_batches_on_one_bvh(orc):
    """Bvh::intersect on a const Bvh is re-entrant in the reference: batch launches of ONE device BVH issued from
    several host threads on several streams must not share a ray counter or scratch (work-slot ring, stream-ordered scratch)."""
    import threading
    import torch
    import bvh_amd
    tris = synth.sponza_proxy(30_000)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    obb, occ = orc.prep_tris(tris)
    ob = orc.build(obb, occ, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM)
    oprims = orc.precompute_tris(tris, ob.prim_ids())
    lo, hi = synth.scene_bounds(tris)
    n_threads, rounds = 6, 8
    rays = [synth.rays_closest(200_000, lo, hi, seed=100 + k) for k in range(n_threads)]
    want = [ob.intersect_tri(oprims, r, 0, 1, threads=8) for r in rays]
    torch.cuda.synchronize()
    errors = []

    def work(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d_rays = torch.from_numpy(rays[k]).cuda()
                for i in range(rounds):
                    hits = bvh_amd.intersect(bvh, prims, d_rays, any_hit=False, robust=True, sort_rays=(i % 2 == 1))
                    stream.synchronize()
                    if bvh_amd.hits_to_numpy(hits).tobytes() != want[k].tobytes():
                        errors.append((k, i))
        except Exception as exc:                              # noqa: BLE001
            errors.append((k, repr(exc)))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_original_primitive_ids_flag(orc):
    """BVH_AMD_RAY_ORIGINAL_IDS: hit.prim is bvh.prim_ids[BVH-order index] (SURVEY.md 8b "required extension"), misses stay invalid."""
    import bvh_amd
    for dt, leaf in ((np.float32, "tri"), (np.float64, "sphere")):
        if leaf == "tri":
            geo = synth.soup(20_000, seed=4, jitter=0.02, dtype=dt)
            bb, cc = bvh_amd.tri_bounds(geo)
        else:
            geo = synth.spheres(20_000, dtype=dt)
            bb, cc = bvh_amd.sphere_bounds(geo)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
        prims = bvh_amd.precompute_tris(geo, bvh.device_prim_ids()) if leaf == "tri" else bvh_amd.gather(geo, bvh.device_prim_ids())
        lo, hi = synth.scene_bounds(geo)
        rays = synth.rays_closest(50_000, lo, hi, dtype=dt)
        a = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, rays, robust=True, leaf=leaf))
        b = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, rays, robust=True, leaf=leaf, original_ids=True))
        hit = a["prim"] != oracle.INVALID
        assert hit.sum() > 1000 and (~hit).sum() > 0
        ids = bvh.prim_ids
        assert (b["prim"][hit].astype(np.uint64) == ids[a["prim"][hit].astype(np.int64)]).all()
        assert (b["prim"][~hit] == a["prim"][~hit]).all()
        assert a["t"].tobytes() == b["t"].tobytes() and a["u"].tobytes() == b["u"].tobytes()


def test_launch_plan_search_settles_and_never_changes_results():
    """Large batches through a tree beyond the L2s: the library tries four candidate plans (reordered or as given, per-lane or
    quad-cooperative record fetch, thresholds), one whole batch each, twice, and keeps the fastest (csrc/traverse.hip:
    launch_traverse). Whatever it tries, the hit records are the same bytes; after the search the plan no longer changes; a
    re-laid-out tree (optimize) starts a new search; forced choices bypass it."""
    import ctypes as C
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    tris = synth.soup(1_200_000, seed=5)
    d = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    n = 1 << 21
    for any_hit, rays_h in ((False, synth.rays_closest(n, lo, hi)), (True, synth.rays_shadow(n, lo, hi))):
        rays = torch.from_numpy(rays_h).cuda()
        want = bvh_amd.intersect(bvh, prims, rays, any_hit=any_hit, robust=True, sort_rays=False).cpu().numpy().tobytes()
        plans = []
        for _ in range(14):
            got = bvh_amd.intersect(bvh, prims, rays, any_hit=any_hit, robust=True)
            torch.cuda.synchronize()
            assert got.cpu().numpy().tobytes() == want
            pl = (C.c_int * 4)()
            lib.bvh_amd_last_launch_plan(pl)
            plans.append(tuple(pl))
        assert len(set(plans[:8])) >= 3, plans                  # the candidates really differ ...
        assert len(set(plans[9:])) == 1, plans                  # ... and the search ends
        forced = bvh_amd.intersect(bvh, prims, rays, any_hit=any_hit, robust=True, sort_rays=True)
        assert forced.cpu().numpy().tobytes() == want and lib.bvh_amd_last_launch_reordered() == 1
    settled = plans[-1]
    bvh.optimize()                                              # new layout: the measured plan is forgotten
    rays = torch.from_numpy(synth.rays_shadow(n, lo, hi)).cuda()
    prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
    seen = set()
    for _ in range(4):
        bvh_amd.intersect(bvh, prims, rays, any_hit=True, robust=True)
        torch.cuda.synchronize()
        pl = (C.c_int * 4)()
        lib.bvh_amd_last_launch_plan(pl)
        seen.add(tuple(pl))
    assert len(seen) >= 3, (seen, settled)


def test_ticket_ranges_thresholds_and_fetch_modes_never_change_results():
    """bvh_amd_tuning: any number of ticket ranges (1..256), any refill / leaf thresholds and either record fetch give the same hit
    records (a ragged batch size, so that the last ranges are short or empty)."""
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    tris = synth.sponza_proxy(40_000)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    try:
        for any_hit, rays_h in ((False, synth.rays_closest(200_003, lo, hi)), (True, synth.rays_shadow(70_001, lo, hi))):
            rays = torch.from_numpy(rays_h).cuda()
            lib.bvh_amd_tuning(-1, -1, 0, 1)
            want = bvh_amd.intersect(bvh, prims, rays, any_hit=any_hit, robust=True).cpu().numpy().tobytes()
            for coop in (0, 1):
                for refill, leaf, parts in ((36, 12, 8), (12, 12, 32), (1, 1, 256), (63, 40, 3), (20, 20, 200)):
                    lib.bvh_amd_tuning(refill, leaf, coop, parts)
                    for sort in (False, True):
                        got = bvh_amd.intersect(bvh, prims, rays, any_hit=any_hit, robust=True, sort_rays=sort)
                        assert got.cpu().numpy().tobytes() == want, (any_hit, coop, refill, leaf, parts, sort)
    finally:
        lib.bvh_amd_tuning(-1, -1, -1, -1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_prepare_trace_changes_nothing_but_the_first_call(orc, dtype):
    """bvhXX_prepare_trace (additive): the first batch's per-tree one-offs paid up front — depth pass, first reordering scratch. Hit
    records are the reference's with and without it; the call is idempotent and accepts any hint."""
    import torch
    import bvh_amd
    tris = synth.soup(300_000, dtype=dtype)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM)
    oprims = orc.precompute_tris(tris, ref.prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(1_200_000, lo, hi, seed=11).astype(dtype)
    want = ref.intersect_tri(oprims, rays, False, True, threads=8)
    d_tris = torch.from_numpy(tris).cuda()
    d_bb, d_cc = bvh_amd.tri_bounds(d_tris)
    for prepared in (True, False):
        gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
        assert gpu.serialize() == ref.serialize()
        if prepared:
            for hint in (0, len(rays), len(rays), 1 << 22):
                bvh_amd.prepare_trace(gpu, hint)
        prims = bvh_amd.precompute_tris(d_tris, gpu.device_prim_ids())
        got = bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True)
        assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes()
