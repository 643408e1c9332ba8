"""-m gpu: device builders against the oracle / golden streams. Bar: memcmp-equal Bvh::serialize streams
(bit-exact node bounds, packed indices, node numbering and prim_ids order)."""
import numpy as np
import pytest

import oracle
from bvh_amd import synth
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _gpu_binned(bb, cc, **kw):
    import bvh_amd
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.Low, **kw)
    return bvh_amd.BinnedSahBuilder.build(bb, cc, cfg)


@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k", "soup2k_f64", "spheres2k_f64"])
def test_binned_matches_golden_stream(scene):
    import bvh_amd
    g = load_golden(scene)
    bvh = _gpu_binned(g["bboxes"], g["centers"])
    assert bvh.serialize() == g["bvh_binned"].tobytes()
    # DefaultBuilder serial overload at Quality::Low is the same builder (default_builder.h:54-55)
    bvh2 = bvh_amd.DefaultBuilder.build(g["bboxes"], g["centers"], bvh_amd.Config(quality=bvh_amd.Quality.Low))
    assert bvh2.serialize() == g["bvh_serial_low"].tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 63, 64, 65, 66, 127, 129, 1000, 2047, 2048, 2049, 4097, 30000])
def test_binned_sizes(orc, n):
    tris = synth.soup(n, seed=n, jitter=0.05)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    assert _gpu_binned(bb, cc).serialize() == ref.serialize()


@pytest.mark.parametrize("scene,n", [("soup", 300_000), ("terrain", 300_000), ("sponza", 262_144), ("soup", 1_000_000)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_binned_scenes(orc, scene, n, dtype):
    if dtype == np.float64 and n > 300_000:
        pytest.skip("covered by the float case")
    tris = {"soup": lambda: synth.soup(n, dtype=dtype), "terrain": lambda: synth.terrain(n, dtype=dtype),
            "sponza": lambda: synth.sponza_proxy(n, dtype=dtype)}[scene]()
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    gpu = _gpu_binned(bb, cc)
    assert gpu.node_count == ref.node_count
    assert gpu.serialize() == ref.serialize()


# ---- BinnedSahBuilder<Node, BinCount> with BinCount != 8 (binned_sah_builder.h:18; bvhXX_build_device_binned) ----------------------

@pytest.mark.parametrize("scene", ["soup2k", "terrain2k", "soup2k_f64", "circles2k_2f"])
@pytest.mark.parametrize("bins", [4, 16, 32])
def test_binned_bin_counts_match_golden_stream(scene, bins):
    import bvh_amd
    g, gb = load_golden(scene), load_golden("template_knobs")
    bvh = bvh_amd.BinnedSahBuilder.build(g["bboxes"], g["centers"], bin_count=bins)
    assert bvh.serialize() == gb[f"{scene}_bins{bins}"].tobytes()
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.Low, min_leaf_size=2, max_leaf_size=5)
    assert bvh_amd.BinnedSahBuilder.build(g["bboxes"], g["centers"], cfg, bin_count=bins).serialize() == gb[f"{scene}_bins{bins}_leaf2to5"].tobytes()
    # the default is the tuned path and stays what it was
    assert bvh_amd.BinnedSahBuilder.build(g["bboxes"], g["centers"], bin_count=8).serialize() == g["bvh_binned"].tobytes()


@pytest.mark.parametrize("bins", [4, 16, 32])
@pytest.mark.parametrize("n", [1, 2, 9, 63, 64, 65, 129, 1000, 2049, 30000])
def test_binned_bin_counts_sizes(orc, bins, n):
    import bvh_amd
    tris = synth.soup(n, seed=n + bins, jitter=0.05)
    bb, cc = orc.prep_tris(tris)
    try:
        orc.set_bin_count(bins)
        ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    finally:
        orc.set_bin_count(8)
    assert bvh_amd.BinnedSahBuilder.build(bb, cc, bin_count=bins).serialize() == ref.serialize()


@pytest.mark.parametrize("bins", [4, 16, 32])
@pytest.mark.parametrize("scene,n,dtype", [("soup", 300_000, np.float32), ("terrain", 300_000, np.float32), ("sponza", 262_144, np.float32),
                                           ("soup", 200_000, np.float64), ("sponza", 100_000, np.float64), ("soup", 1_000_000, np.float32)])
def test_binned_bin_counts_scenes(orc, bins, scene, n, dtype):
    import bvh_amd
    tris = {"soup": lambda: synth.soup(n, dtype=dtype), "terrain": lambda: synth.terrain(n, dtype=dtype),
            "sponza": lambda: synth.sponza_proxy(n, dtype=dtype)}[scene]()
    bb, cc = orc.prep_tris(tris)
    sah = bvh_amd.SplitHeuristic(1, 0.7) if scene == "sponza" else bvh_amd.SplitHeuristic()
    try:
        orc.set_bin_count(bins)
        if scene == "sponza":
            orc.set_sah(1, 0.7)
        ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    finally:
        orc.set_bin_count(8)
        orc.set_sah()
    gpu = bvh_amd.BinnedSahBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low, sah=sah), bin_count=bins)
    assert gpu.node_count == ref.node_count
    assert gpu.serialize() == ref.serialize()


def test_binned_bin_counts_degenerate_and_refused(orc):
    """Duplicates / flat input force fallback_split under every bin count; a BinCount outside {4, 8, 16, 32} is refused loudly."""
    import bvh_amd
    base = synth.soup(40, seed=1, jitter=0.05)
    t = synth.soup(5000, seed=2, jitter=0.03)
    t[:, 2::3] = 0.5
    t = (np.round(t * 8) / 8).astype(np.float32)
    for tris in (np.repeat(base, 50, axis=0), np.repeat(base[:1], 300, axis=0), t):
        bb, cc = orc.prep_tris(np.ascontiguousarray(tris))
        for bins in (4, 16, 32):
            try:
                orc.set_bin_count(bins)
                ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
            finally:
                orc.set_bin_count(8)
            assert bvh_amd.BinnedSahBuilder.build(bb, cc, bin_count=bins).serialize() == ref.serialize()
    with pytest.raises(Exception):
        bvh_amd.BinnedSahBuilder.build(bb, cc, bin_count=12)


@pytest.mark.parametrize("min_leaf,max_leaf", [(1, 1), (1, 4), (2, 8), (4, 4), (1, 15), (8, 15)])
def test_binned_leaf_limits(orc, min_leaf, max_leaf):
    tris = synth.soup(20000, seed=11, jitter=0.02)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED, min_leaf=min_leaf, max_leaf=max_leaf)
    gpu = _gpu_binned(bb, cc, min_leaf_size=min_leaf, max_leaf_size=max_leaf)
    assert gpu.serialize() == ref.serialize()


def test_binned_degenerate_inputs(orc):
    """Coincident centroids and flat boxes force fallback_split (std::partial_sort order) in both phases."""
    rng = np.random.default_rng(5)
    cases = []
    # many exact duplicates: SAH cannot separate them
    base = synth.soup(40, seed=1, jitter=0.05)
    cases.append(np.repeat(base, 50, axis=0))                      # 2000 prims, 50 copies each
    cases.append(np.repeat(base[:3], 400, axis=0))                 # 1200 prims on 3 sites (big fallback segments)
    # all primitives identical
    cases.append(np.repeat(base[:1], 300, axis=0))
    # flat: everything in the plane z = 0.5, centroids on a coarse lattice (heavy ties)
    t = synth.soup(5000, seed=2, jitter=0.03)
    t[:, 2::3] = 0.5
    t = np.round(t * 8) / 8
    cases.append(t.astype(np.float32))
    # a line of points with equal x
    t = synth.soup(3000, seed=3, jitter=0.0)
    t[:, 0::3] = 0.25
    cases.append(t)
    for tris in cases:
        tris = tris[rng.permutation(len(tris))]
        bb, cc = orc.prep_tris(np.ascontiguousarray(tris))
        ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
        gpu = _gpu_binned(bb, cc)
        assert gpu.serialize() == ref.serialize()


def test_build_then_trace_matches_oracle(orc):
    """End to end on the device: bounds -> build -> permute/precompute -> trace, vs the CPU oracle."""
    import bvh_amd
    tris = synth.sponza_proxy(100_000)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(100_000, lo, hi)
    hits = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, rays, robust=True))
    obb, occ = orc.prep_tris(tris)
    ob = orc.build(obb, occ, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_LOW)
    assert bvh.serialize() == ob.serialize()
    oh = ob.intersect_tri(orc.precompute_tris(tris, ob.prim_ids()), rays, 0, 1, threads=8)
    assert hits.tobytes() == oh.tobytes()


def test_bad_config_fails_loudly():
    import bvh_amd
    tris = synth.soup(500)
    bb, cc = bvh_amd.tri_bounds(tris)
    with pytest.raises(bvh_amd.BvhAmdError):
        bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(max_leaf_size=16))      # 4-bit primitive count


# ---- sweep SAH (SweepSahBuilder, DefaultBuilder serial Medium) ----------------------------------------------

def _gpu_sweep(bb, cc, **kw):
    import bvh_amd
    return bvh_amd.SweepSahBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium, **kw))


@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k", "soup2k_f64", "spheres2k_f64"])
def test_sweep_matches_golden_stream(scene):
    import bvh_amd
    g = load_golden(scene)
    assert _gpu_sweep(g["bboxes"], g["centers"]).serialize() == g["bvh_sweep"].tobytes()
    bvh2 = bvh_amd.DefaultBuilder.build(g["bboxes"], g["centers"], bvh_amd.Config(quality=bvh_amd.Quality.Medium))
    assert bvh2.serialize() == g["bvh_serial_med"].tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 8, 9, 16, 17, 63, 64, 65, 66, 129, 1000, 2049, 4097, 30000])
def test_sweep_sizes(orc, n):
    tris = synth.soup(n, seed=n + 1, jitter=0.05)
    bb, cc = orc.prep_tris(tris)
    assert _gpu_sweep(bb, cc).serialize() == orc.build(bb, cc, builder=oracle.BUILDER_SWEEP).serialize()


@pytest.mark.parametrize("scene,n", [("soup", 200_000), ("terrain", 200_000), ("sponza", 262_144), ("terrain", 1_000_000)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sweep_scenes(orc, scene, n, dtype):
    if dtype == np.float64 and n > 200_000:
        pytest.skip("covered by the float case")
    tris = {"soup": lambda: synth.soup(n, dtype=dtype), "terrain": lambda: synth.terrain(n, dtype=dtype),
            "sponza": lambda: synth.sponza_proxy(n, dtype=dtype)}[scene]()
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_SWEEP)
    gpu = _gpu_sweep(bb, cc)
    assert gpu.node_count == ref.node_count
    assert gpu.serialize() == ref.serialize()


@pytest.mark.parametrize("min_leaf,max_leaf", [(1, 1), (1, 4), (2, 8), (4, 4), (1, 15)])
def test_sweep_leaf_limits(orc, min_leaf, max_leaf):
    tris = synth.terrain(20000)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_SWEEP, min_leaf=min_leaf, max_leaf=max_leaf)
    gpu = _gpu_sweep(bb, cc, min_leaf_size=min_leaf, max_leaf_size=max_leaf)
    assert gpu.serialize() == ref.serialize()


def test_sweep_degenerate_inputs(orc):
    rng = np.random.default_rng(6)
    base = synth.soup(40, seed=1, jitter=0.05)
    cases = [np.repeat(base, 50, axis=0), np.repeat(base[:3], 400, axis=0), np.repeat(base[:1], 300, axis=0)]
    t = synth.soup(5000, seed=2, jitter=0.03)
    t[:, 2::3] = 0.5
    cases.append((np.round(t * 8) / 8).astype(np.float32))
    for tris in cases:
        tris = np.ascontiguousarray(tris[rng.permutation(len(tris))])
        bb, cc = orc.prep_tris(tris)
        assert _gpu_sweep(bb, cc).serialize() == orc.build(bb, cc, builder=oracle.BUILDER_SWEEP).serialize()


# ---- mini-tree builder (DefaultBuilder with a thread pool, Low / Medium) ---------------------------------------

def _gpu_parallel(bb, cc, quality, **kw):
    import bvh_amd
    return bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=quality, **kw), thread_pool=bvh_amd.ThreadPool())


@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k", "soup2k_f64", "spheres2k_f64"])
@pytest.mark.parametrize("q,name", [(0, "parallel_low"), (1, "parallel_med")])
def test_minitree_matches_golden_stream(scene, q, name):
    import bvh_amd
    g = load_golden(scene)
    assert _gpu_parallel(g["bboxes"], g["centers"], bvh_amd.Quality(q)).serialize() == g[f"bvh_{name}"].tobytes()


@pytest.mark.parametrize("scene,n", [("soup", 5000), ("soup", 200_000), ("terrain", 200_000), ("sponza", 262_144), ("soup", 1_000_000)])
@pytest.mark.parametrize("q", [0, 1])
def test_minitree_scenes(orc, scene, n, q):
    import bvh_amd
    tris = {"soup": lambda: synth.soup(n), "terrain": lambda: synth.terrain(n), "sponza": lambda: synth.sponza_proxy(n)}[scene]()
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=q)
    gpu = _gpu_parallel(bb, cc, bvh_amd.Quality(q))
    assert gpu.node_count == ref.node_count
    assert gpu.serialize() == ref.serialize()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_minitree_builder_direct(orc, dtype):
    """MiniTreeBuilder::build(pool, ...) with its own Config (pruning on/off, area ratio, parallel_threshold, leaf limits),
    which DefaultBuilder only reaches in three settings."""
    import bvh_amd
    tris = synth.sponza_proxy(50_000).astype(dtype)
    bb, cc = orc.prep_tris(tris)
    for kw in (dict(), dict(enable_pruning=False), dict(pruning_area_ratio=0.3), dict(pruning_area_ratio=1.5, max_leaf_size=4),
               dict(parallel_threshold=200, pruning_area_ratio=0.05), dict(min_leaf_size=2, max_leaf_size=15, pruning_area_ratio=0.0)):
        okw = {{"max_leaf_size": "max_leaf", "min_leaf_size": "min_leaf"}.get(k, k): v for k, v in kw.items()}
        ref = orc.build_minitree(bb, cc, **okw)
        gpu = bvh_amd.MiniTreeBuilder.build(bb, cc, bvh_amd.MiniTreeBuilder.Config(**kw), thread_pool=bvh_amd.ThreadPool())
        assert gpu.serialize() == ref.serialize(), kw
    for n in (1, 2, 9, 300):                                  # no serial fallback here: tiny inputs go through the grid too
        t = synth.soup(n, jitter=0.05).astype(dtype)
        b2, c2 = orc.prep_tris(t)
        assert bvh_amd.MiniTreeBuilder.build(b2, c2).serialize() == orc.build_minitree(b2, c2).serialize(), n
    for L in (1, 2, 3, 5, 6, 7):                              # other grids than the default 16^3
        for kw in (dict(), dict(enable_pruning=False), dict(parallel_threshold=64, pruning_area_ratio=0.2), dict(parallel_threshold=0)):
            okw = dict(kw, log2_grid_dim=L)
            gpu = bvh_amd.MiniTreeBuilder.build(bb, cc, bvh_amd.MiniTreeBuilder.Config(**okw))
            assert gpu.serialize() == orc.build_minitree(bb, cc, **okw).serialize(), okw
    for L in (0, 11):
        with pytest.raises(bvh_amd.BvhAmdError, match="log2_grid_dim"):
            bvh_amd.MiniTreeBuilder.build(bb, cc, bvh_amd.MiniTreeBuilder.Config(log2_grid_dim=L))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reinsertion_optimizer_config(orc, dtype):
    """ReinsertionOptimizer::Config {batch_size_ratio, max_iter_count} (reinsertion_optimizer.h:18-24): small and large batches
    (a ratio >= 1 selects every node but the root: the candidate heap is never replaced into), zero iterations, many."""
    import bvh_amd
    for scene, tris in (("sponza", synth.sponza_proxy(30_000)), ("terrain", synth.terrain(20_000))):
        bb, cc = orc.prep_tris(tris.astype(dtype))
        for ratio, iters in ((0.05, 3), (0.01, 1), (0.2, 2), (1.0, 1), (3.0, 2), (0.0, 4), (0.5, 0), (1e-9, 5), (0.3, 7)):
            ref = orc.build(bb, cc, quality=1)
            ref.optimize(batch_size_ratio=ratio, max_iter_count=iters)
            gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium))
            gpu.optimize(batch_size_ratio=ratio, max_iter_count=iters)
            assert gpu.serialize() == ref.serialize(), (scene, ratio, iters)
    with pytest.raises(bvh_amd.BvhAmdError, match="batch_size_ratio"):
        gpu.optimize(batch_size_ratio=-0.1)
    with pytest.raises(bvh_amd.BvhAmdError, match="batch_size_ratio"):
        gpu.optimize(batch_size_ratio=float("nan"))
    # 2D
    circ = synth.circles(6000, dtype=dtype)
    b2, c2 = orc.sphere_bboxes(circ)
    ref = orc.build(b2, c2, quality=1)
    ref.optimize(batch_size_ratio=0.5, max_iter_count=2)
    gpu = bvh_amd.DefaultBuilder.build(b2, c2, bvh_amd.Config(quality=bvh_amd.Quality.Medium))
    gpu.optimize(batch_size_ratio=0.5, max_iter_count=2)
    assert gpu.serialize() == ref.serialize()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_split_heuristic_parameters(orc, dtype):
    """TopDownSahBuilder::Config::sah = SplitHeuristic(log_cluster_size, cost_ratio) (split_heuristic.h:17-38) through every
    builder: clusters of 2 / 4 / 8 primitives, cheap and expensive nodes, a negative ratio."""
    import bvh_amd
    from bvh_amd import SplitHeuristic as SH
    tris = synth.sponza_proxy(20_000).astype(dtype)
    bb, cc = orc.prep_tris(tris)
    circ = synth.circles(5000, dtype=dtype)
    b2, c2 = orc.sphere_bboxes(circ)
    default = bvh_amd.SweepSahBuilder.build(bb, cc).serialize()
    for log, ratio in ((1, 1.0), (2, 1.0), (3, 0.5), (0, 2.0), (0, 0.25), (2, 3.0), (0, -1.0)):
        sah = SH(log, ratio)
        orc.set_sah(log, ratio)
        try:
            for name, builder, quality in conftest_modes():
                cfg = bvh_amd.Config(quality=bvh_amd.Quality(quality), sah=sah)
                if builder == 2:
                    gpu = bvh_amd.BinnedSahBuilder.build(bb, cc, cfg)
                elif builder == 3:
                    gpu = bvh_amd.SweepSahBuilder.build(bb, cc, cfg)
                else:
                    gpu = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if builder == 1 else None)
                assert gpu.serialize() == orc.build(bb, cc, builder=builder, quality=quality).serialize(), (name, log, ratio)
                if builder == 3:
                    assert gpu.serialize() != default           # the parameters do change the tree
            gpu = bvh_amd.MiniTreeBuilder.build(bb, cc, bvh_amd.MiniTreeBuilder.Config(pruning_area_ratio=0.2, parallel_threshold=300, sah=sah))
            assert gpu.serialize() == orc.build_minitree(bb, cc, pruning_area_ratio=0.2, parallel_threshold=300).serialize(), (log, ratio)
            gpu = bvh_amd.DefaultBuilder.build(b2, c2, bvh_amd.Config(sah=sah))                         # 2D, serial High
            assert gpu.serialize() == orc.build(b2, c2, quality=2).serialize(), ("2d", log, ratio)
        finally:
            orc.set_sah()
    with pytest.raises(bvh_amd.BvhAmdError, match="log_cluster_size"):
        bvh_amd.SweepSahBuilder.build(bb, cc, bvh_amd.Config(sah=SH(64, 1.0)))


def conftest_modes():
    from conftest import MODES
    return MODES


def test_minitree_threshold_and_clustered_input(orc):
    """A dense cluster puts most primitives into one grid cell (one big mini-tree) and parallel_threshold changes
    the merge; both must follow the reference."""
    import bvh_amd
    a = synth.soup(30000, seed=3, jitter=0.001) * 0.01           # everything within 1% of the box
    b = synth.soup(3000, seed=4, jitter=0.02)
    tris = np.ascontiguousarray(np.concatenate([a, b]).astype(np.float32))
    bb, cc = orc.prep_tris(tris)
    for q in (0, 1):
        for thr in (1024, 100, 5000):
            ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=q, parallel_threshold=thr)
            gpu = _gpu_parallel(bb, cc, bvh_amd.Quality(q), parallel_threshold=thr)
            assert gpu.serialize() == ref.serialize(), (q, thr)


# ---- reinsertion optimizer / Quality::High --------------------------------------------------------------------

@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k", "soup2k_f64", "spheres2k_f64"])
def test_high_quality_matches_golden_stream(scene):
    import bvh_amd
    g = load_golden(scene)
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.High)
    assert bvh_amd.DefaultBuilder.build(g["bboxes"], g["centers"], cfg).serialize() == g["bvh_serial_high"].tobytes()
    assert bvh_amd.DefaultBuilder.build(g["bboxes"], g["centers"], cfg, thread_pool=bvh_amd.ThreadPool()).serialize() \
        == g["bvh_parallel_high"].tobytes()


@pytest.mark.parametrize("scene,n", [("soup", 3), ("soup", 40), ("soup", 5000), ("terrain", 100_000), ("sponza", 262_144), ("soup", 300_000)])
@pytest.mark.parametrize("parallel", [False, True])
def test_high_quality_scenes(orc, scene, n, parallel):
    import bvh_amd
    tris = {"soup": lambda: synth.soup(n, jitter=0.01), "terrain": lambda: synth.terrain(n), "sponza": lambda: synth.sponza_proxy(n)}[scene]()
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL if parallel else oracle.BUILDER_DEFAULT_SERIAL,
                    quality=oracle.QUALITY_HIGH)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High),
                                       thread_pool=bvh_amd.ThreadPool() if parallel else None)
    assert gpu.serialize() == ref.serialize()


@pytest.mark.parametrize("parallel", [False, True])
def test_high_quality_double_large(orc, parallel):
    """double precision with a candidate heap that reaches below LDS (the two-wave replacement loop, 16-byte entries)"""
    import bvh_amd
    tris = synth.soup(250_000, jitter=0.01, dtype=np.float64)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL if parallel else oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_HIGH)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool() if parallel else None)
    assert gpu.serialize() == ref.serialize()


def _lattice(n_side, dtype=np.float32):
    """identical triangles on a power-of-two lattice: node costs and reinsertion gains tie massively"""
    g = np.arange(n_side, dtype=dtype)
    org = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 1, 3)
    tri = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.25], [0.0, 0.5, 0.25]], dtype=dtype)
    return np.ascontiguousarray((org + tri[None]).reshape(-1, 9))


@pytest.mark.parametrize("scene", ["soup", "terrain", "lattice", "lattice_f64", "dup"])
def test_reinsertion_fast_path_and_exact_replay_agree_with_reference(orc, scene, monkeypatch):
    """The heap-free fast path must either prove the candidate-heap layout irrelevant or fall back to the exact replay;
    both, and the forced replay, give the reference's bytes. The soup takes the fast path, the lattice (ties everywhere) the replay."""
    import bvh_amd
    tris = {"soup": lambda: synth.soup(60_000, jitter=0.01), "terrain": lambda: synth.terrain(60_000),
            "lattice": lambda: _lattice(32), "lattice_f64": lambda: _lattice(24, np.float64),
            "dup": lambda: np.concatenate([synth.soup(20_000), synth.soup(20_000)])}[scene]()      # every primitive twice
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_HIGH).serialize()
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.High)
    f0, e0 = bvh_amd.reinsertion_stats()
    assert bvh_amd.DefaultBuilder.build(bb, cc, cfg).serialize() == ref
    f1, e1 = bvh_amd.reinsertion_stats()
    assert (f1 - f0) + (e1 - e0) == 3
    if scene == "soup":
        assert f1 - f0 == 3
    if scene.startswith("lattice"):
        assert e1 - e0 >= 1
    monkeypatch.setenv("BVH_AMD_REINSERT", "exact")
    assert bvh_amd.DefaultBuilder.build(bb, cc, cfg).serialize() == ref
    f2, e2 = bvh_amd.reinsertion_stats()
    assert (f2 - f1, e2 - e1) == (0, 3)


@pytest.mark.parametrize("knobs", ["", "BVH_AMD_HEAP_WIDE=1", "BVH_AMD_HEAP_PIPE=1", "BVH_AMD_HEAP_PIPE=0", "BVH_AMD_APPLY_DEFER=0"])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_every_candidate_heap_kernel_replays_the_reference(orc, knobs, dtype, tmp_path):
    """find_candidates' heap array (reinsertion_optimizer.h:88-105) decides the order of equal gains downstream, so every kernel that
    replays it must leave the reference's bytes: the register-resident head with 32- and 64-lane ancestor masks (heap_head.inc), round 5's
    two-wave loop and the one-wave loop (developer library: the knobs are read once, own process), with the replay forced in all three
    iterations and a candidate heap that reaches below the LDS levels (600k / 300k triangles -> k = 5 % of ~1M / ~0.5M nodes)."""
    import subprocess, sys, os
    n = 600_000 if dtype == "float32" else 300_000
    tris = synth.soup(n, jitter=0.01, dtype=np.dtype(dtype).type)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH).serialize()
    path = tmp_path / "tris.npy"
    np.save(path, tris)
    code = (
        "import numpy as np, torch, bvh_amd, hashlib, sys\n"
        f"tris = torch.from_numpy(np.load(r'{path}')).cuda()\n"
        "bb, cc = bvh_amd.tri_bounds(tris)\n"
        "b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())\n"
        "assert bvh_amd.last_optimize_profile()['replayed'] == 3\n"
        "print('sha1', hashlib.sha1(b.serialize()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BVH_AMD_REINSERT="exact", BVH_AMD_LIB=os.path.join(root, "bvh_amd", "lib", "libbvh_amd_dev.so"))
    if knobs:
        env.update(dict([knobs.split("=")]))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    import hashlib
    assert r.returncode == 0 and ("sha1 " + hashlib.sha1(ref).hexdigest()) in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_standalone_optimize_matches_reference(orc):
    """bvhXX_optimize on an existing BVH (here: a binned tree, which reinsertion improves a lot), twice."""
    import bvh_amd
    tris = synth.sponza_proxy(60_000)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    gpu = bvh_amd.Bvh.from_nodes(ref.nodes(), ref.prim_ids())
    for _ in range(2):
        ref.optimize(-1)
        gpu.optimize()
        assert gpu.serialize() == ref.serialize()
    # the optimised tree still traces identically
    prims = bvh_amd.precompute_tris(tris, gpu.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(50_000, lo, hi)
    hits = bvh_amd.hits_to_numpy(bvh_amd.intersect(gpu, prims, rays, robust=True))
    assert hits.tobytes() == ref.intersect_tri(orc.precompute_tris(tris, ref.prim_ids()), rays, 0, 1, threads=8).tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_optimize_refits_like_the_reference_whatever_the_boxes(orc, dtype):
    """refit_from (reinsertion_optimizer.h:215-225) recomputes the ancestors of a move's two ends and nothing else. The device leaves
    the refits of an iteration to one pass after its last move only when that cannot change a bit (csrc/reinsert.hip: k_deferrable);
    trees where it could — loose inner boxes a caller made by hand, -0.0 in a box (`a < b ? a : b` picks between the zeros by
    position), a loose ROOT (which only a move from under the root recomputes) — keep every byte of the reference's result too."""
    import bvh_amd
    tris = synth.sponza_proxy(40_000).astype(dtype)
    bb, cc = orc.prep_tris(tris)
    base = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    prim_ids = base.prim_ids()
    rng = np.random.default_rng(5)

    def variant(kind):
        nodes = base.nodes().copy()
        bounds = nodes["bounds"]
        inner = np.flatnonzero((nodes["index"] & 15) == 0)
        if kind == "loose":
            pick = rng.choice(inner[inner != 0], size=200, replace=False)
            bounds[pick, 0::2] -= dtype(0.25)
            bounds[pick, 1::2] += dtype(0.25)
        elif kind == "loose_root":
            bounds[0, 0::2] -= dtype(1.0)
            bounds[0, 1::2] += dtype(1.0)
        elif kind == "negative_zero":
            # a scene moved so that boxes end at zero, with both zeros among the leaves
            shift = bounds[0, 0::2].copy()
            bounds[:, 0::2] -= shift
            bounds[:, 1::2] -= shift
            zeros = np.argwhere(bounds == 0)
            assert len(zeros) > 2
            for r, c in zeros[::2]:
                bounds[r, c] = -dtype(0.0)
        return nodes

    for kind in ("tight", "loose", "loose_root", "negative_zero"):
        nodes = variant(kind)
        ref = orc.from_arrays(nodes, prim_ids)
        gpu = bvh_amd.Bvh.from_nodes(nodes, prim_ids)
        for ratio, iters in ((0.05, 3), (0.3, 2)):
            ref.optimize(-1, batch_size_ratio=ratio, max_iter_count=iters)
            gpu.optimize(batch_size_ratio=ratio, max_iter_count=iters)
            assert gpu.serialize() == ref.serialize(), (kind, ratio, iters)
    # small trees with every node a candidate: moves whose `from` hangs under the root put the sibling's node into slot 0
    # (reinsertion_optimizer.h:207) and refit_from(0) recomputes the root — the only moves that do
    for n in (4, 9, 50, 300, 3000):
        for seed in range(6):
            small = synth.soup(n, jitter=0.3, seed=seed).astype(dtype)
            sb, sc_ = orc.prep_tris(small)
            for builder in (oracle.BUILDER_BINNED, oracle.BUILDER_SWEEP):
                ref = orc.build(sb, sc_, builder=builder)
                gpu = bvh_amd.Bvh.from_nodes(ref.nodes(), ref.prim_ids())
                ref.optimize(-1, batch_size_ratio=1.0, max_iter_count=4)
                gpu.optimize(batch_size_ratio=1.0, max_iter_count=4)
                assert gpu.serialize() == ref.serialize(), (n, seed, builder)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_extract_bvh_matches_reference(orc, dtype):
    """Bvh::extract_bvh(root_id) as a device op: node order (right child first, children allocated at visit time) and the
    re-packed prim ids equal the reference's, for subtrees of a device-built tree and of a host-supplied tree."""
    import bvh_amd
    tris = synth.sponza_proxy(40_000).astype(dtype)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    assert gpu.serialize() == ref.serialize()
    nodes = ref.nodes()
    leaf = int(np.flatnonzero(nodes["index"] & 15)[0])
    for root in (1, 2, 5, 6, 1000, 1001, leaf, ref.node_count - 1, 0):
        if root == 0:
            continue                                          # (the reference asserts root_id != 0)
        sub = gpu.extract_bvh(root)
        want = ref.extract(root)
        assert sub.serialize() == want.serialize(), root
        assert sub.node_count == want.node_count and sub.prim_count == want.prim_count
    hosted = bvh_amd.Bvh.from_nodes(ref.nodes(), ref.prim_ids())
    assert hosted.extract_bvh(2).serialize() == ref.extract(2).serialize()
    # an extracted subtree is a complete BVH: it traces like the reference's
    sub, want = gpu.extract_bvh(1), ref.extract(1)
    prims = bvh_amd.precompute_tris(tris, sub.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(20_000, lo, hi).astype(dtype)
    hits = bvh_amd.hits_to_numpy(bvh_amd.intersect(sub, prims, rays, robust=True))
    assert hits.tobytes() == want.intersect_tri(orc.precompute_tris(tris, want.prim_ids()), rays, 0, 1, threads=4).tobytes()
    with pytest.raises(bvh_amd.BvhAmdError):
        gpu.extract_bvh(gpu.node_count)


# ---- refit / node editing -------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_refit_matches_reference(orc, dtype):
    """Bvh::refit after moving leaf boxes: every inner box recomputed bottom-up, bit-identical to the reference."""
    import bvh_amd
    tris = synth.soup(50_000, seed=13, jitter=0.02, dtype=dtype)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM)
    nodes = ref.nodes().copy()
    ids = ref.prim_ids()
    rng = np.random.default_rng(3)
    leaves = np.nonzero((nodes["index"] & 15) != 0)[0]
    moved = rng.choice(leaves, size=len(leaves) // 3, replace=False)
    shift = rng.normal(0, 0.05, size=(len(moved), 3)).astype(dtype)
    nodes["bounds"][moved, 0::2] += shift
    nodes["bounds"][moved, 1::2] += shift
    nodes["bounds"][moved[:50], 0] = 0.0                       # exercise the +0 / -0 tie rule of robust_min
    nodes["bounds"][moved[50:100], 0] = -0.0
    a = orc.from_arrays(nodes, ids)
    a.refit()
    g = bvh_amd.Bvh.from_nodes(nodes, ids)
    g.refit()
    assert g.serialize() == a.serialize()
    # the same through the node setters on the host mirror
    g2 = bvh_amd.Bvh.from_nodes(ref.nodes(), ids)
    for k in moved[:200]:
        b = nodes["bounds"][k]
        g2.set_node_bbox(int(k), b[0::2], b[1::2])
    n2 = ref.nodes().copy()
    n2["bounds"][moved[:200]] = nodes["bounds"][moved[:200]]
    a2 = orc.from_arrays(n2, ids)
    a2.refit()
    g2.refit()
    assert g2.serialize() == a2.serialize()


def test_refit_and_extract_at_a_million_triangles(orc):
    """The bottom-up climbs of refit / extract_bvh hand subtrees from lane to lane through arrival tickets WITHOUT cache maintenance
    (build_common.h: ticket_release / ticket_acquire): on a 1.9M-node tree the lanes of a climb sit on all eight XCDs, and a stale box
    or count anywhere would show in the stream. Repeated, on a tree whose leaves moved."""
    import bvh_amd
    tris = synth.soup(1_000_000)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_LOW)
    nodes, ids = ref.nodes().copy(), ref.prim_ids()
    rng = np.random.default_rng(17)
    leaves = np.flatnonzero(nodes["index"] & 15)
    moved = rng.choice(leaves, size=len(leaves) // 2, replace=False)
    shift = rng.normal(0, 0.01, size=(len(moved), 3)).astype(np.float32)
    nodes["bounds"][moved, 0::2] += shift
    nodes["bounds"][moved, 1::2] += shift
    a = orc.from_arrays(nodes, ids)
    a.refit()
    want = a.serialize()
    for rep in range(3):
        g = bvh_amd.Bvh.from_nodes(nodes, ids)
        g.refit()
        assert g.serialize() == want, rep
    g = bvh_amd.Bvh.from_nodes(ref.nodes(), ids)
    for root in (1, 2):
        assert g.extract_bvh(root).serialize() == ref.extract(root).serialize(), root


def test_arrays_with_unused_sibling_pairs_refit_extract_trace_and_refuse_optimize(orc):
    """from_nodes / deserialize tolerate sibling pairs no inner node references (the reference tolerates them: nodes
    left behind by append_node / remove_last_node edits, hand-built arrays). Everything that walks parent links must then stop at the
    top of such a subtree instead of following a stale word: refit (== the reference's traverse_bottom_up, which refits the unused
    subtree too), extract_bvh, the depth of the traversal stack, tracing; optimize refuses (the reference would hang the unused
    nodes under the root and corrupt the tree)."""
    import bvh_amd
    tris = synth.soup(20_000, seed=21, jitter=0.02)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_MEDIUM)
    nodes, ids = ref.nodes().copy(), ref.prim_ids()
    n = len(nodes)
    # append an unused SUBTREE: an inner node + a leaf as one pair, the inner node's children as a second pair (leaves over prims 0..1)
    extra = np.zeros(4, dtype=nodes.dtype)
    extra["bounds"] = nodes["bounds"][1]
    extra["index"][0] = (n + 2) << 4                          # inner: children at n + 2, n + 3
    extra["index"][1] = (0 << 4) | 1
    extra["index"][2] = (0 << 4) | 1
    extra["index"][3] = (1 << 4) | 1
    extra["bounds"][2] = nodes["bounds"][np.flatnonzero(nodes["index"] & 15)[0]]
    extra["bounds"][3] = nodes["bounds"][np.flatnonzero(nodes["index"] & 15)[1]]
    # ... and a deep unused CHAIN of 80 levels, which must not deepen the traversal stack (tree_depth counts what hangs below the root)
    chain = np.zeros(160, dtype=nodes.dtype)
    base = n + 4
    for lvl in range(80):
        chain["bounds"][2 * lvl] = nodes["bounds"][2]
        chain["bounds"][2 * lvl + 1] = nodes["bounds"][2]
        chain["index"][2 * lvl] = ((base + 2 * (lvl + 1)) << 4) if lvl < 79 else ((0 << 4) | 1)
        chain["index"][2 * lvl + 1] = (0 << 4) | 1
    wide = np.concatenate([nodes, extra, chain])
    g = bvh_amd.Bvh.from_nodes(wide, ids)
    a = orc.from_arrays(wide, ids)
    # tracing ignores what is unreachable, and the stack is sized by the reachable tree (<= 64 levels: not the Deep kernel)
    prims = bvh_amd.precompute_tris(tris, g.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(30_000, lo, hi)
    hits = bvh_amd.hits_to_numpy(bvh_amd.intersect(g, prims, rays, robust=True))
    assert hits.tobytes() == ref.intersect_tri(orc.precompute_tris(tris, ref.prim_ids()), rays, 0, 1, threads=4).tobytes()
    assert "true>" not in bvh_amd._lib.load().bvh_amd_last_kernel_name().decode().split(", ")[-1], "an unreachable chain selected the deep-stack kernel"
    # refit: move some reachable leaves; reachable part AND unused subtree equal the reference's traverse_bottom_up
    rng = np.random.default_rng(5)
    leaves = np.flatnonzero((nodes["index"] & 15) != 0)
    moved = rng.choice(leaves, size=500, replace=False)
    wide2 = wide.copy()
    wide2["bounds"][moved] += np.float32(0.01)
    g = bvh_amd.Bvh.from_nodes(wide2, ids)
    a = orc.from_arrays(wide2, ids)
    g.refit(); a.refit()
    assert g.serialize() == a.serialize()
    # extract of the reachable tree drops the unused nodes; of the unused subtree's top it yields that subtree
    assert g.extract_bvh(1).serialize() == a.extract(1).serialize()
    assert g.extract_bvh(n).serialize() == a.extract(n).serialize()
    with pytest.raises(bvh_amd.BvhAmdError, match="not reachable"):
        g.optimize()
    assert g.serialize() == a.serialize(), "a refused optimize must leave the tree alone"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lopsided_segments_outgrow_the_one_block_builders_and_fall_back(orc, dtype):
    """k_medium / k_sweep_medium (round 4) keep a segment's nodes in a local table of 256 nodes / 40 levels. Centres in geometric progression
    make every binned split peel a handful of primitives off one end (the bins are linear in x): a chain of hundreds of levels in double
    precision — the block gives up, the build is retried on the plain level-synchronous path, and the stream is still the reference's; in
    float the exponent range allows ~37 levels: the same data through the one-block path itself."""
    import bvh_amd
    n = 1000 if dtype == np.float64 else 430
    step = 0.5 if dtype == np.float64 else 0.2
    ctr = np.zeros((n, 3), dtype=dtype)
    ctr[:, 0] = np.exp(step * np.arange(n)).astype(dtype)
    ctr[:, 1] = 1.0
    ctr[:, 2] = -2.0
    half = (ctr[:, :1] * dtype(1e-3)).astype(dtype)
    bb = np.ascontiguousarray(np.concatenate([ctr - half, ctr + half], axis=1).astype(dtype))
    cc = np.ascontiguousarray(ctr)
    for builder, make in ((oracle.BUILDER_BINNED, lambda: bvh_amd.BinnedSahBuilder.build(bb, cc)),
                          (oracle.BUILDER_SWEEP, lambda: bvh_amd.SweepSahBuilder.build(bb, cc)),
                          (oracle.BUILDER_DEFAULT_SERIAL, lambda: bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low)))):
        ref = orc.build(bb, cc, builder=builder, quality=oracle.QUALITY_LOW)
        gpu = make()
        assert gpu.serialize() == ref.serialize(), (dtype, builder)
    if dtype == np.float64:
        nodes = orc.build(bb, cc, builder=oracle.BUILDER_BINNED).nodes()
        assert len(nodes) > 2 * 256, "the scene no longer forces the one-block builder to give up"


def test_refit_on_device_built_tree_is_identity(orc):
    import bvh_amd
    tris = synth.terrain(30_000)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    before = bvh.serialize()
    bvh.refit()                                                # builder output already satisfies parent = union(children)
    assert bvh.serialize() == before


def test_scratch_block_cache_across_streams_and_release(orc):
    """Build scratch comes from a block cache per (device, stream) in front of the stream-ordered pool (common.h: scratch_alloc):
    builds of different sizes interleaved on two streams, a flush of the cache in between, and repeated builds that re-use the
    cached blocks all give the reference's trees."""
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    scenes = [synth.soup(n, seed=s) for n, s in ((30_000, 1), (90_000, 2), (31_000, 3), (200_000, 4))]
    want = []
    for tris in scenes:
        bb, cc = orc.prep_tris(tris)
        want.append([orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=q).serialize() for q in (0, 2)])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        built = []
        for i, tris in enumerate(scenes):
            with torch.cuda.stream(streams[(i + rep) % 2]):
                d = torch.from_numpy(tris).cuda()
                bb, cc = bvh_amd.tri_bounds(d)
                for q in (0, 2):
                    built.append((i, q, bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())))
        torch.cuda.synchronize()
        for i, q, b in built:
            assert b.serialize() == want[i][q // 2], (rep, i, q)
        del built
        if rep == 1:
            assert lib.bvh_amd_release_cached_memory() == 0


def test_concurrent_builds_from_host_threads(orc):
    """Quality::Low builds run their top level on a worker thread + stream of the calling thread (build_minitree.hip) and Phase B on a
    CU-masked stream: four host threads building at once — on the default stream and on streams of their own — get the reference's
    trees every time, as do Medium builds mixed in (they share the scratch cache, the readback words and the pool)."""
    import threading
    import torch
    import bvh_amd
    scenes = [synth.soup(120_000, seed=1), synth.terrain(150_000), synth.sponza_proxy(100_000), synth.soup(40_000, seed=7, jitter=0.03)]
    prepared, want = [], []
    for tris in scenes:
        bb, cc = orc.prep_tris(tris)
        want.append([orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=q).serialize() for q in (0, 1)])
        prepared.append(bvh_amd.tri_bounds(torch.from_numpy(tris).cuda()))
    torch.cuda.synchronize()
    failures = []

    def work(t, own_stream):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                for rep in range(6):
                    i = (t + rep) % len(scenes)
                    q = 0 if rep % 3 else 1
                    bb, cc = prepared[i]
                    b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())
                    if b.serialize() != want[i][q]:
                        failures.append((t, rep, i, q))
        except Exception as e:                                 # noqa: BLE001 - reported by the assertion below
            failures.append((t, repr(e)))

    for own_stream in (False, True):
        threads = [threading.Thread(target=work, args=(t, own_stream)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        assert not failures, (own_stream, failures)


@pytest.mark.parametrize("knob", ["BVH_AMD_CACHE_MB=0", "BVH_AMD_CACHE_MB=1", "BVH_AMD_POOL=0"])
def test_scratch_block_cache_off_and_tiny(knob):
    """BVH_AMD_CACHE_MB=0 (no block cache: every request goes to the pool), =1 (every build overflows the bound and evicts) and
    BVH_AMD_POOL=0 (plain hipMalloc / hipFree with the synchronisations that needs): the golden streams still come out. Own process:
    the knobs are read once."""
    import subprocess, sys, os
    code = (
        "import numpy as np, bvh_amd, sys\n"
        "sys.path.insert(0, 'tests')\n"
        "from conftest import load_golden\n"
        "g = load_golden('soup2k')\n"
        "for rep in range(3):\n"
        "    bb, cc = bvh_amd.tri_bounds(g['prims'])\n"
        "    for q, name in ((0, 'low'), (2, 'high')):\n"
        "        b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())\n"
        "        assert b.serialize() == g['bvh_parallel_' + name].tobytes(), (rep, name)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **dict([knob.split("=")])), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_release_cached_memory_soak(orc):
    """bvh_amd_release_cached_memory() is the one place where the library lets the runtime's pool unmap memory (everything idle before and
    after). 200 x { release; 1M-triangle Medium build; 1M-ray trace }: every stream and every hit record equal to the reference's
    (VERDICT r5 Next 4: the retention policy's evidence was 72 runs with one release each)."""
    import torch
    import bvh_amd
    lib = bvh_amd._lib.load()
    tris = synth.soup(1_000_000)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_MEDIUM)
    want_stream = ref.serialize()
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(1_048_576, lo, hi, seed=5)
    want_hits = ref.intersect_tri(orc.precompute_tris(tris, ref.prim_ids()), rays, False, True, threads=8).tobytes()
    d_tris, d_rays = torch.from_numpy(tris).cuda(), torch.from_numpy(rays).cuda()
    bad = []
    for i in range(200):
        torch.cuda.synchronize()
        assert lib.bvh_amd_release_cached_memory() == 0
        d_bb, d_cc = bvh_amd.tri_bounds(d_tris)
        gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
        prims = bvh_amd.precompute_tris(d_tris, gpu.device_prim_ids())
        hits = bvh_amd.hits_to_numpy(bvh_amd.intersect(gpu, prims, d_rays, any_hit=False, robust=True, sort_rays=True)).tobytes()
        if hits != want_hits or (i % 20 == 0 and gpu.serialize() != want_stream):
            bad.append(i)
        del gpu, prims
    assert not bad, f"rounds with a wrong build or wrong hits after a release: {bad}"


def test_cache_off_does_not_hold_the_sum_of_a_builds_scratch():
    """With the block cache off (BVH_AMD_CACHE_MB=0) every freed block goes back to the runtime's pool. It must go back while the call
    runs (in the order of the freeing stream, csrc/build_device.hip: flush_deferred_locked), not only when the outermost API call
    ends: a 4M-triangle High build's peak pool usage with the cache off stays within reach of the default configuration's peak
    (the deferred blocks used to add up to the SUM of the build's scratch). tests/helpers/pool_peak.py, own processes."""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    peak = {}
    for knob in ("", "0"):
        env = dict(os.environ)
        if knob:
            env["BVH_AMD_CACHE_MB"] = knob
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "pool_peak.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        m = re.search(r"pool_used_high_mb (\d+)", r.stdout)
        assert r.returncode == 0 and m, (r.stdout[-500:], r.stderr[-2000:])
        peak[knob] = int(m.group(1))
    print("pool peak MB: default", peak[""], "cache off", peak["0"])
    assert peak["0"] <= 1.3 * peak[""] + 384, peak


def test_worker_streams_leave_nothing_cached_when_their_thread_ends():
    """A host thread that builds Quality::Low trees owns a worker stream (build_minitree.hip); when the thread ends the stream is destroyed,
    and scratch cached under its handle must be gone by then: the first eviction that met such a block crashed inside the runtime (a hang
    of this suite in round 4). tests/helpers/evict_after_threads.py: three threads build and end, then the main thread's 1M-triangle builds evict under a
    64 MB bound. Own process: the bound is read once."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "evict_after_threads.py")], cwd=root, env=dict(os.environ, BVH_AMD_CACHE_MB="64"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "] done" in r.stdout, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
