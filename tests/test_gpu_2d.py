"""-m gpu: the 2D families `2f` / `2d` (Bvh<Node<T, 2>>, c_api/bvh.cpp:7-10) against the reference: serial DefaultBuilder
(Low = binned, Medium = sweep, High = sweep + reinsertion), BinnedSahBuilder / SweepSahBuilder, optimize, refit, extract_bvh,
save/load, the node accessors, and the batch traversal with circles (Sphere<T, 2>)."""
import ctypes as C
import os
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _circles(n, dtype, seed=7, clustered=False):
    rng = np.random.default_rng(seed)
    ctr = rng.random((n, 2))
    if clustered:
        ctr = (ctr ** 3) * np.array([4.0, 0.25])
    rad = rng.random((n, 1)) * 0.004 + 0.0005
    return np.ascontiguousarray(np.concatenate([ctr, rad], axis=1).astype(dtype))


def _grid_circles(side, dtype):
    """identical circles on a power-of-two lattice: ties in every cost and gain"""
    g = np.arange(side, dtype=dtype)
    xy = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    return np.ascontiguousarray(np.concatenate([xy, np.full((len(xy), 1), 0.25, dtype=dtype)], axis=1))


def _rays2(n, dtype, seed=3, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    org = lo + rng.random((n, 2)) * (hi - lo) * 1.1 - 0.05 * (hi - lo)
    ang = rng.random(n) * 2 * np.pi
    d = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    return np.ascontiguousarray(np.concatenate([org, d, np.zeros((n, 1)), np.full((n, 1), np.finfo(dtype).max)], axis=1).astype(dtype))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene", ["uniform", "clustered", "grid"])
def test_2d_builders_match_reference(orc, dtype, scene):
    import bvh_amd
    circ = {"uniform": lambda: _circles(30_000, dtype), "clustered": lambda: _circles(30_000, dtype, clustered=True),
            "grid": lambda: _grid_circles(128, dtype)}[scene]()
    bb, cc = orc.sphere_bboxes(circ)
    d_bb, d_cc = bvh_amd.sphere_bounds(circ)
    assert d_bb.cpu().numpy().tobytes() == bb.tobytes() and d_cc.cpu().numpy().tobytes() == cc.tobytes()
    for q in (bvh_amd.Quality.Low, bvh_amd.Quality.Medium, bvh_amd.Quality.High):
        ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=int(q))
        gpu = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=q))
        assert gpu.dim == 2 and gpu.nodes.dtype.itemsize == (20 if dtype == np.float32 else 40)
        assert gpu.serialize() == ref.serialize(), (scene, q)
        assert gpu.nodes.tobytes() == ref.nodes().tobytes() and (gpu.prim_ids == ref.prim_ids()).all()
    assert bvh_amd.BinnedSahBuilder.build(bb, cc).serialize() == orc.build(bb, cc, builder=oracle.BUILDER_BINNED).serialize()
    assert bvh_amd.SweepSahBuilder.build(bb, cc).serialize() == orc.build(bb, cc, builder=oracle.BUILDER_SWEEP).serialize()
    for lim in ((1, 1), (2, 6), (4, 15)):
        cfg = bvh_amd.Config(quality=bvh_amd.Quality.Low, min_leaf_size=lim[0], max_leaf_size=lim[1])
        assert bvh_amd.DefaultBuilder.build(bb, cc, cfg).serialize() == orc.build(bb, cc, quality=0, min_leaf=lim[0], max_leaf=lim[1]).serialize()


@pytest.mark.parametrize("n", [1, 2, 3, 9, 64, 65, 1000, 2049])
def test_2d_sizes(orc, n):
    import bvh_amd
    circ = _circles(n, np.float32, seed=n)
    bb, cc = orc.sphere_bboxes(circ)
    for q in (0, 1, 2):
        assert bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q))).serialize() == orc.build(bb, cc, quality=q).serialize()


def test_2d_thread_pool_semantics(orc):
    """With a pool the reference runs the serial builder below parallel_threshold; at or above it the mini-tree builder reads the
    third component of 2D points (undefined behaviour): refused loudly, never guessed."""
    import bvh_amd
    circ = _circles(5000, np.float32)
    bb, cc = orc.sphere_bboxes(circ)
    cfg = bvh_amd.Config(quality=bvh_amd.Quality.Medium, parallel_threshold=10_000)
    assert bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool()).serialize() == \
        orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=1, parallel_threshold=10_000).serialize()
    with pytest.raises(bvh_amd.BvhAmdError, match="2D"):
        bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_traversal_matches_reference(orc, dtype):
    import bvh_amd
    circ = _circles(50_000, dtype)
    bb, cc = orc.sphere_bboxes(circ)
    ref = orc.build(bb, cc, quality=2)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High))
    assert gpu.serialize() == ref.serialize()
    ordered = circ[ref.prim_ids().astype(np.int64)]
    d_ordered = bvh_amd.gather(circ, gpu.device_prim_ids())
    assert d_ordered.cpu().numpy().tobytes() == ordered.tobytes()
    rays = _rays2(100_000, dtype)
    rays[:50, 2:4] = 0                                         # zero directions
    rays[50:100, 4] = -1.0                                     # negative tmin
    rays[100:150, 5] = 0.01                                    # short rays
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_sphere(ordered, rays, any_hit, robust, threads=8, counters=True)
            got, cg = bvh_amd.intersect(gpu, d_ordered, rays, any_hit=any_hit, robust=robust, counters=True)
            assert bvh_amd.hits_to_numpy(got).tobytes() == want.tobytes(), (any_hit, robust)
            assert (cg.cpu().numpy().astype(np.uint64) == cw).all()
    assert not hasattr(bvh_amd._lib.load(), "bvh2f_intersect_rays_tri")       # tri.h has no 2D intersector: not exported


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_optimize_refit_extract_roundtrip(orc, dtype, tmp_path):
    import bvh_amd
    circ = _circles(20_000, dtype, clustered=True)
    bb, cc = orc.sphere_bboxes(circ)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_BINNED)
    gpu = bvh_amd.Bvh.from_nodes(ref.nodes(), ref.prim_ids())
    assert gpu.dim == 2 and gpu.serialize() == ref.serialize()
    for _ in range(2):                                        # standalone ReinsertionOptimizer::optimize
        ref.optimize(-1)
        gpu.optimize()
        assert gpu.serialize() == ref.serialize()
    for root in (1, 2, 77):
        assert gpu.extract_bvh(root).serialize() == ref.extract(root).serialize()
    # refit after moving a leaf box (host-side edit through bvh_node2X_set_bbox)
    nodes = ref.nodes().copy()
    leaf = int(np.flatnonzero(nodes["index"] & 15)[5])
    b = nodes["bounds"][leaf].copy()
    lo, hi = [b[0] - 0.5, b[2] - 0.25], [b[1] + 0.125, b[3] + 1.0]
    gpu.set_node_bbox(leaf, lo, hi)
    gpu.refit()
    nodes["bounds"][leaf] = [lo[0], hi[0], lo[1], hi[1]]
    ref2 = orc.from_arrays(nodes, ref.prim_ids())
    ref2.refit()
    assert gpu.serialize() == ref2.serialize()
    # stream round trip (Bvh::serialize / deserialize with 20/40-byte nodes) and FILE* save/load through the raw C-ABI
    again = bvh_amd.Bvh.deserialize(gpu.serialize(), dtype=dtype, dim=2)
    assert again.serialize() == gpu.serialize()
    lib = bvh_amd._lib.load()
    libc = C.CDLL(None)
    libc.fopen.restype, libc.fopen.argtypes = C.c_void_p, [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    s = "2f" if dtype == np.float32 else "2d"
    path = os.path.join(tmp_path, "bvh2.bin").encode()
    f = libc.fopen(path, b"wb"); getattr(lib, f"bvh{s}_save")(gpu._h, f); libc.fclose(f)
    assert open(path, "rb").read() == gpu.serialize()
    f = libc.fopen(path, b"rb"); h = getattr(lib, f"bvh{s}_load")(f); libc.fclose(f)
    loaded = bvh_amd.Bvh(h, s)
    assert loaded.serialize() == gpu.serialize()
    # accessors
    n0 = getattr(lib, f"bvh{s}_get_node")(loaded._h, 0)
    assert not getattr(lib, f"bvh_node{s}_is_leaf")(n0)
    assert getattr(lib, f"bvh_node{s}_get_first_id")(n0) == int(loaded.nodes[0]["index"]) >> 4
    bbx = getattr(lib, f"bvh_node{s}_get_bbox")(n0)
    root = loaded.nodes[0]["bounds"]
    assert list(bbx.v) == [root[0], root[2], root[1], root[3]]       # {min.x, min.y, max.x, max.y}


@pytest.mark.parametrize("scene", ["circles2k_2f", "circles2k_2d"])
def test_2d_matches_golden(scene):
    """against the committed fixtures generated from the unmodified reference (tests/golden/make_golden.py)"""
    import bvh_amd
    from conftest import MODES, load_golden
    g = load_golden(scene)
    circ = g["prims"]
    d_bb, d_cc = bvh_amd.sphere_bounds(circ)
    assert d_bb.cpu().numpy().tobytes() == g["bboxes"].tobytes() and d_cc.cpu().numpy().tobytes() == g["centers"].tobytes()
    builders = {"binned": lambda: bvh_amd.BinnedSahBuilder.build(d_bb, d_cc), "sweep": lambda: bvh_amd.SweepSahBuilder.build(d_bb, d_cc),
                "serial_low": lambda: bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Low)),
                "serial_med": lambda: bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium)),
                "serial_high": lambda: bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High))}
    for mode, _, _ in MODES[:5]:
        bvh = builders[mode]()
        assert bvh.serialize() == g[f"bvh_{mode}"].tobytes(), mode
        if mode in ("serial_low", "serial_high"):
            pp = bvh_amd.gather(circ, bvh.device_prim_ids())
            for any_hit in (0, 1):
                for robust in (0, 1):
                    key = f"{mode}_{'any' if any_hit else 'closest'}_{'robust' if robust else 'fast'}"
                    hits, cnt = bvh_amd.intersect(bvh, pp, g["rays_shadow"] if any_hit else g["rays_closest"], any_hit=bool(any_hit),
                                                  robust=bool(robust), counters=True)
                    assert bvh_amd.hits_to_numpy(hits).tobytes() == g[f"hits_{key}"].tobytes(), key
                    assert (cnt.cpu().numpy().astype(np.uint64) == g[f"counters_{key}"]).all()


@pytest.mark.parametrize("sfx,scene", [("2f", "circles2k_2f"), ("2d", "circles2k_2d")])
def test_2d_raw_c_abi_host_pointers_and_editing(sfx, scene):
    """bvh2X_build with host pointers exactly as a C program calls it (with and without a thread pool), then the node editing
    API of c_api/bvh.h:170-218 on the 20/40-byte mirror: set_bbox + refit, append_node / remove_last_node + sync_device."""
    from bvh_amd import _lib as L
    from conftest import load_golden
    dll = L.load()
    g = load_golden(scene)
    bb, cc = np.ascontiguousarray(g["bboxes"]), np.ascontiguousarray(g["centers"])
    for quality, key in ((0, "serial_low"), (1, "serial_med"), (2, "serial_high")):
        cfg = L.BuildConfig(quality, 1, 8, 1024)
        h = getattr(dll, f"bvh{sfx}_build")(None, bb.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), len(bb), C.byref(cfg))
        assert h, L.last_error()
        n = getattr(dll, f"bvh{sfx}_serialize")(h, None, 0)
        buf = C.create_string_buffer(n)
        getattr(dll, f"bvh{sfx}_serialize")(h, buf, n)
        assert buf.raw == g[f"bvh_{key}"].tobytes()
        getattr(dll, f"bvh{sfx}_destroy")(h)
    pool = dll.bvh_thread_pool_create(8)
    cfg = L.BuildConfig(1, 1, 8, 1 << 20)                      # below parallel_threshold: the serial builder (default_builder.h:38-39)
    h = getattr(dll, f"bvh{sfx}_build")(pool, bb.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), len(bb), C.byref(cfg))
    assert h, L.last_error()
    cfg = L.BuildConfig(1, 1, 8, 1024)                         # at/above it: undefined in the reference, refused
    assert not getattr(dll, f"bvh{sfx}_build")(pool, bb.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), len(bb), C.byref(cfg))
    assert "2D" in L.last_error()
    dll.bvh_thread_pool_destroy(pool)
    # NULL config = the defaults (Quality::High)
    h2 = getattr(dll, f"bvh{sfx}_build")(None, bb.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), len(bb), None)
    n2 = getattr(dll, f"bvh{sfx}_serialize")(h2, None, 0)
    buf2 = C.create_string_buffer(n2)
    getattr(dll, f"bvh{sfx}_serialize")(h2, buf2, n2)
    assert buf2.raw == g["bvh_serial_high"].tobytes()
    getattr(dll, f"bvh{sfx}_destroy")(h2)
    # append a node, fill it through the setters, remove it again: the stream is unchanged after sync_device
    nn = getattr(dll, f"bvh{sfx}_get_node_count")(h)
    before = C.create_string_buffer(getattr(dll, f"bvh{sfx}_serialize")(h, None, 0))
    getattr(dll, f"bvh{sfx}_serialize")(h, before, len(before))
    getattr(dll, f"bvh{sfx}_append_node")(h)
    assert getattr(dll, f"bvh{sfx}_get_node_count")(h) == nn + 1
    node = getattr(dll, f"bvh{sfx}_get_node")(h, nn)
    ct = C.c_float if sfx == "2f" else C.c_double
    getattr(dll, f"bvh_node{sfx}_set_bbox")(node, (ct * 4)(0.25, 0.5, 0.75, 1.0))
    getattr(dll, f"bvh_node{sfx}_set_first_id")(node, 7)
    getattr(dll, f"bvh_node{sfx}_set_prim_count")(node, 3)
    assert getattr(dll, f"bvh_node{sfx}_is_leaf")(node) and getattr(dll, f"bvh_node{sfx}_get_first_id")(node) == 7
    assert list(getattr(dll, f"bvh_node{sfx}_get_bbox")(node).v) == [0.25, 0.5, 0.75, 1.0]
    getattr(dll, f"bvh{sfx}_remove_last_node")(h)
    assert getattr(dll, f"bvh{sfx}_sync_device")(h) == 0
    after = C.create_string_buffer(len(before))
    getattr(dll, f"bvh{sfx}_serialize")(h, after, len(after))
    assert after.raw == before.raw
    getattr(dll, f"bvh{sfx}_destroy")(h)
