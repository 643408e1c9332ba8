"""-m gpu: seeded differential fuzzing of the device path against the oracle on small adversarial scenes: lattice coordinates
(ties in every cost), duplicated and degenerate primitives, signed zeros, mixed magnitudes; every builder mode, leaf limits,
float and double, 3D triangles and 2D circles; then traversal with axis-aligned and degenerate rays."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _scene3(rng, n, kind, dtype):
    if kind == "lattice":
        base = rng.integers(0, 8, size=(n, 1, 3)).astype(dtype)
        tri = rng.integers(0, 3, size=(n, 3, 3)).astype(dtype) * dtype(0.5)
        t = base + tri
    elif kind == "dups":
        m = max(1, n // 4)
        src = rng.random((m, 3, 3)).astype(dtype)
        t = src[rng.integers(0, m, size=n)]
    elif kind == "flat":                                       # everything in the plane z = -0.0 / +0.0
        t = rng.random((n, 3, 3)).astype(dtype)
        t[:, :, 2] = np.where(rng.random((n, 3)) < 0.5, dtype(0.0), -dtype(0.0))
    elif kind == "points":                                     # zero-area triangles (p0 == p1 == p2), some coincident
        p = rng.integers(0, 5, size=(n, 1, 3)).astype(dtype)
        t = np.repeat(p, 3, axis=1)
    elif kind == "scales":
        t = (rng.random((n, 3, 3)) * 10.0 ** rng.integers(-6, 6, size=(n, 1, 1))).astype(dtype)
    else:
        t = (rng.random((n, 1, 3)) + (rng.random((n, 3, 3)) - 0.5) * 0.1).astype(dtype)
    return np.ascontiguousarray(t.reshape(n, 9))


def _rays3(rng, n, lo, hi, dtype):
    r = np.zeros((n, 8), dtype=dtype)
    ext = np.maximum(hi - lo, 1e-3)
    r[:, 0:3] = lo - 0.2 * ext + rng.random((n, 3)) * 1.4 * ext
    d = rng.standard_normal((n, 3))
    axis = rng.integers(0, 4, size=n)                          # a third of the rays are axis-aligned (zeros in dir, both signs)
    for a in range(3):
        sel = axis == a
        d[sel] = 0
        d[sel, a] = np.where(rng.random(sel.sum()) < 0.5, 1.0, -1.0)
    zero = rng.random(n) < 0.05
    d[zero, rng.integers(0, 3)] = -0.0
    r[:, 3:6] = d
    r[:, 6] = np.where(rng.random(n) < 0.1, -1.0, 0.0)
    r[:, 7] = np.where(rng.random(n) < 0.2, rng.random(n) * ext.max(), np.finfo(dtype).max)
    # snap some origins onto box planes
    snap = rng.random(n) < 0.2
    r[snap, 0] = np.floor(r[snap, 0])
    return r


CASES = [(seed, kind) for seed in range(6) for kind in ("lattice", "dups", "flat", "points", "scales", "uniform")]


@pytest.mark.parametrize("seed,kind", CASES)
def test_fuzz_3d(orc, seed, kind):
    import bvh_amd
    rng = np.random.default_rng(1000 * seed + ("lattice", "dups", "flat", "points", "scales", "uniform").index(kind))
    dtype = np.float32 if seed % 3 else np.float64
    n = int(rng.choice([1, 2, 5, 17, 64, 65, 200, 1500, 4000]))
    tris = _scene3(rng, n, kind, dtype)
    bb, cc = orc.prep_tris(tris)
    d_bb, d_cc = bvh_amd.tri_bounds(tris)
    assert d_bb.cpu().numpy().tobytes() == bb.tobytes() and d_cc.cpu().numpy().tobytes() == cc.tobytes()
    lim = [(1, 8), (1, 1), (2, 4), (3, 15)][seed % 4]
    last = None
    for builder, quality in ((2, 0), (3, 0), (0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)):
        cfg = bvh_amd.Config(quality=bvh_amd.Quality(quality), min_leaf_size=lim[0], max_leaf_size=lim[1], parallel_threshold=[1024, 64][seed % 2])
        if builder == 2:
            gpu = bvh_amd.BinnedSahBuilder.build(d_bb, d_cc, cfg)
        elif builder == 3:
            gpu = bvh_amd.SweepSahBuilder.build(d_bb, d_cc, cfg)
        else:
            gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, cfg, thread_pool=bvh_amd.ThreadPool() if builder == 1 else None)
        ref = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=cfg.parallel_threshold)
        assert gpu.serialize() == ref.serialize(), (seed, kind, n, builder, quality, lim)
        last = (gpu, ref)
    # BinnedSahBuilder<Node, BinCount> with another BinCount (binned_sah_builder.h:18): one of 4 / 16 / 32 per case
    bins = (4, 16, 32)[seed % 3]
    try:
        orc.set_bin_count(bins)
        ref_bins = orc.build(bb, cc, builder=2, quality=0, min_leaf=lim[0], max_leaf=lim[1])
    finally:
        orc.set_bin_count(8)
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.Low, min_leaf_size=lim[0], max_leaf_size=lim[1])
    assert bvh_amd.BinnedSahBuilder.build(d_bb, d_cc, cfg, bin_count=bins).serialize() == ref_bins.serialize(), (seed, kind, n, "bins", bins, lim)
    gpu, ref = last
    prims = bvh_amd.precompute_tris(tris, gpu.device_prim_ids())
    lo, hi = tris.reshape(-1, 3).min(axis=0).astype(np.float64), tris.reshape(-1, 3).max(axis=0).astype(np.float64)
    rays = _rays3(rng, 3000, lo, hi, dtype)
    want_prims = orc.precompute_tris(tris, ref.prim_ids())
    assert prims.cpu().numpy().tobytes() == want_prims.tobytes()
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_tri(want_prims, rays, any_hit, robust, counters=True)
            got, cg = bvh_amd.intersect(gpu, prims, rays, any_hit=any_hit, robust=robust, counters=True)
            g = bvh_amd.hits_to_numpy(got)
            same = (g["prim"] == want["prim"]) & ((g["t"] == want["t"]) | (np.isnan(g["t"]) & np.isnan(want["t"])))
            assert same.all(), (seed, kind, any_hit, robust, int((~same).sum()))
            assert (cg.cpu().numpy().astype(np.uint64) == cw).all()


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_2d(orc, seed):
    import bvh_amd
    rng = np.random.default_rng(77 + seed)
    dtype = np.float32 if seed % 2 else np.float64
    n = int(rng.choice([1, 3, 40, 64, 65, 700, 3000]))
    if seed % 4 == 0:
        ctr = rng.integers(0, 6, size=(n, 2)).astype(dtype)    # lattice: ties, coincident circles
        rad = rng.integers(0, 3, size=(n, 1)).astype(dtype) * dtype(0.25)      # (radius 0 included)
    else:
        ctr = rng.random((n, 2)).astype(dtype)
        rad = (rng.random((n, 1)) * 0.05).astype(dtype)
    circ = np.ascontiguousarray(np.concatenate([ctr, rad], axis=1))
    bb, cc = orc.sphere_bboxes(circ)
    for builder, quality in ((2, 0), (3, 0), (0, 0), (0, 1), (0, 2)):
        cfg = bvh_amd.Config(quality=bvh_amd.Quality(quality), min_leaf_size=1 + seed % 3, max_leaf_size=4 + seed % 9)
        ref = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=cfg.min_leaf_size, max_leaf=cfg.max_leaf_size)
        gpu = {2: bvh_amd.BinnedSahBuilder.build, 3: bvh_amd.SweepSahBuilder.build}.get(builder, bvh_amd.DefaultBuilder.build)(bb, cc, cfg)
        assert gpu.serialize() == ref.serialize(), (seed, n, builder, quality)
    ordered = circ[ref.prim_ids().astype(np.int64)]
    rays = np.zeros((3000, 6), dtype=dtype)
    rays[:, 0:2] = rng.random((3000, 2)) * 8 - 1
    d = rng.standard_normal((3000, 2))
    d[::5, 0] = 0
    d[1::5, 1] = -0.0
    rays[:, 2:4] = d
    rays[:, 4] = np.where(rng.random(3000) < 0.1, -2.0, 0.0)
    rays[:, 5] = np.where(rng.random(3000) < 0.3, rng.random(3000) * 3, np.finfo(dtype).max)
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_sphere(ordered, rays, any_hit, robust, counters=True)
            got, cg = bvh_amd.intersect(gpu, ordered, rays, any_hit=any_hit, robust=robust, counters=True)
            g = bvh_amd.hits_to_numpy(got)
            same = (g["prim"] == want["prim"]) & ((g["t"].view(np.uint32 if dtype == np.float32 else np.uint64) == want["t"].view(np.uint32 if dtype == np.float32 else np.uint64)) | (np.isnan(g["t"]) & np.isnan(want["t"])))
            assert same.all(), (seed, any_hit, robust, int((~same).sum()))
            assert (cg.cpu().numpy().astype(np.uint64) == cw).all()


@pytest.mark.parametrize("seed,kind,n", [(11, "lattice", 6000), (12, "dups", 9000), (13, "points", 5000), (14, "lattice", 30000),
                                          (15, "flat", 20000), (16, "scales", 12000), (17, "dups", 40000)])
def test_fuzz_3d_larger_ties(orc, seed, kind, n):
    """enough primitives for the level-synchronous phases (chunked partitions, Hoare violator matching, the std::sort emulation)
    on tie-saturated data, all builder modes incl. mini-trees + reinsertion"""
    import bvh_amd
    rng = np.random.default_rng(seed)
    dtype = np.float32 if seed % 2 else np.float64
    tris = _scene3(rng, n, kind, dtype)
    bb, cc = orc.prep_tris(tris)
    for builder, quality in ((2, 0), (3, 0), (0, 2), (1, 0), (1, 1), (1, 2)):
        cfg = bvh_amd.Config(quality=bvh_amd.Quality(quality), min_leaf_size=1 + seed % 2, max_leaf_size=8 - seed % 5, parallel_threshold=[1024, 300][seed % 2])
        if builder == 2:
            gpu = bvh_amd.BinnedSahBuilder.build(bb, cc, cfg)
        elif builder == 3:
            gpu = bvh_amd.SweepSahBuilder.build(bb, cc, cfg)
        else:
            gpu = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if builder == 1 else None)
        ref = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=cfg.min_leaf_size, max_leaf=cfg.max_leaf_size, parallel_threshold=cfg.parallel_threshold)
        assert gpu.serialize() == ref.serialize(), (seed, kind, n, builder, quality)
    # standalone ops on the last tree: optimize twice, extract a few subtrees, refit after shrinking nothing (identity)
    for _ in range(2):
        ref.optimize(-1)
        gpu.optimize()
        assert gpu.serialize() == ref.serialize()
    nodes = ref.nodes()
    inner = np.flatnonzero((nodes["index"] & 15) == 0)
    for root in [int(x) for x in rng.choice(inner[inner > 0], size=3)] if (inner > 0).any() else []:
        assert gpu.extract_bvh(root).serialize() == ref.extract(root).serialize()
    again = bvh_amd.Bvh.deserialize(gpu.serialize(), dtype=dtype)
    again.refit()
    ref.refit()
    assert again.serialize() == ref.serialize()


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_3d_spheres(orc, seed):
    import bvh_amd
    rng = np.random.default_rng(500 + seed)
    dtype = np.float32 if seed % 2 else np.float64
    n = int(rng.choice([3, 100, 2500]))
    ctr = rng.integers(0, 5, size=(n, 3)).astype(dtype) if seed == 0 else rng.random((n, 3)).astype(dtype)
    rad = (rng.random((n, 1)) * 0.1).astype(dtype)
    rad[rng.random(n) < 0.1] = 0                               # points
    sph = np.ascontiguousarray(np.concatenate([ctr, rad], axis=1))
    bb, cc = orc.sphere_bboxes(sph)
    d_bb, d_cc = bvh_amd.sphere_bounds(sph)
    assert d_bb.cpu().numpy().tobytes() == bb.tobytes()
    ref = orc.build(bb, cc, builder=1, quality=2, parallel_threshold=128)
    gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(parallel_threshold=128), thread_pool=bvh_amd.ThreadPool())
    assert gpu.serialize() == ref.serialize()
    ordered = sph[ref.prim_ids().astype(np.int64)]
    lo, hi = (ctr - rad).min(axis=0).astype(np.float64), (ctr + rad).max(axis=0).astype(np.float64)
    rays = _rays3(rng, 3000, lo, hi, dtype)
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_sphere(ordered, rays, any_hit, robust, counters=True)
            got, cg = bvh_amd.intersect(gpu, ordered, rays, any_hit=any_hit, robust=robust, leaf="sphere", counters=True)
            g = bvh_amd.hits_to_numpy(got)
            it = np.uint32 if dtype == np.float32 else np.uint64
            same = (g["prim"] == want["prim"]) & ((g["t"].view(it) == want["t"].view(it)) | (np.isnan(g["t"]) & np.isnan(want["t"])))
            assert same.all(), (seed, any_hit, robust, int((~same).sum()))
            assert (cg.cpu().numpy().astype(np.uint64) == cw).all()


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_configs(orc, seed):
    """Every Config knob at once on adversarial scenes: SplitHeuristic, leaf limits, MiniTreeBuilder grid / pruning / threshold,
    ReinsertionOptimizer batch ratio / iterations; float and double."""
    import bvh_amd
    rng = np.random.default_rng(9000 + seed)
    dtype = np.float32 if seed % 2 else np.float64
    kind = ("lattice", "dups", "flat", "points", "scales", "uniform")[seed % 6]
    n = int(rng.choice([1, 2, 9, 64, 65, 300, 1500, 6000]))
    tris = _scene3(rng, n, kind, dtype)
    bb, cc = orc.prep_tris(tris)
    log, ratio = int(rng.integers(0, 4)), float(rng.choice([1.0, 0.5, 2.0, 0.0, -0.5, 3.5]))
    lim = [(1, 8), (1, 1), (2, 4), (3, 15), (1, 2)][int(rng.integers(0, 5))]
    sah = bvh_amd.SplitHeuristic(log, ratio)
    orc.set_sah(log, ratio)
    try:
        for builder, quality in ((2, 0), (3, 0), (0, 2), (1, 1), (1, 2)):
            thr = int(rng.choice([1024, 64, 7]))
            cfg = bvh_amd.Config(quality=bvh_amd.Quality(quality), min_leaf_size=lim[0], max_leaf_size=lim[1], parallel_threshold=thr, sah=sah)
            if builder == 2:
                gpu = bvh_amd.BinnedSahBuilder.build(bb, cc, cfg)
            elif builder == 3:
                gpu = bvh_amd.SweepSahBuilder.build(bb, cc, cfg)
            else:
                gpu = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if builder == 1 else None)
            ref = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=thr)
            assert gpu.serialize() == ref.serialize(), (seed, kind, n, builder, quality, lim, log, ratio, thr)
        for _ in range(3):
            mt = dict(min_leaf_size=lim[0], max_leaf_size=lim[1], enable_pruning=bool(rng.integers(0, 2)),
                      pruning_area_ratio=float(rng.choice([0.01, 0.1, 0.5, 1.5, 0.0])), parallel_threshold=int(rng.choice([1024, 100, 8, 1, 0])),
                      log2_grid_dim=int(rng.integers(1, 7)))
            gpu = bvh_amd.MiniTreeBuilder.build(bb, cc, bvh_amd.MiniTreeBuilder.Config(sah=sah, **mt))
            okw = {{"max_leaf_size": "max_leaf", "min_leaf_size": "min_leaf"}.get(k, k): v for k, v in mt.items()}
            ref = orc.build_minitree(bb, cc, **okw)
            assert gpu.serialize() == ref.serialize(), (seed, kind, n, mt, log, ratio)
            br, it = float(rng.choice([0.05, 0.01, 0.3, 1.0, 2.5, 0.0])), int(rng.integers(0, 5))
            gpu.optimize(batch_size_ratio=br, max_iter_count=it)
            ref.optimize(batch_size_ratio=br, max_iter_count=it)
            assert gpu.serialize() == ref.serialize(), (seed, kind, n, mt, "optimize", br, it)
    finally:
        orc.set_sah()
