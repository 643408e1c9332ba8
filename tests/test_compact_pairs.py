"""CPU: the compact traversal records (bvh_amd/csrc/compact_pair.h, EXPERIMENTAL / opt-in on the device) are lossless for the walk.

tests/cpp/compact_pair_walk.cpp compiles the very header the device code compiles and walks rays with one scalar lane that
follows trace_body.inc's state machine (32-byte record while the lane holds the parent's box, 64-byte record after a pop).
Hits and visit counters must equal the oracle's traversal of the same tree, for trees of every builder mode (the reinsertion
optimizer rewires and refits nodes: the "every parent plane is a child's plane" property must survive it)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from bvh_amd import synth
from conftest import ROOT, load_golden, parse_stream


@pytest.fixture(scope="module")
def walker(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("compact") / "libcompact_walk.so")
    src = os.path.join(ROOT, "tests", "cpp", "compact_pair_walk.cpp")
    cmd = ["g++", "-std=c++20", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dll = C.CDLL(out)
    dll.compact_encode_tree.restype = C.c_int
    dll.compact_encode_tree.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    dll.compact_walk.restype = None
    dll.compact_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return dll


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _encode(walker, nodes):
    nodes = np.ascontiguousarray(nodes)
    n_pairs = (len(nodes) - 1) // 2
    pairs = np.zeros((max(n_pairs, 1), 16), dtype=np.uint32)
    recs = np.zeros((max(n_pairs, 1), 8), dtype=np.uint32)
    rc = walker.compact_encode_tree(_ptr(nodes), len(nodes), _ptr(pairs), _ptr(recs))
    return rc, pairs, recs


def _walk(walker, nodes, pairs, recs, prims, rays, any_hit, robust):
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    prims = np.ascontiguousarray(prims, dtype=np.float32)
    hits = np.empty(len(rays), dtype=oracle.HITF)
    cnt = np.zeros(4, dtype=np.uint64)
    walker.compact_walk(_ptr(pairs), _ptr(recs), int(nodes["index"][0]), _ptr(prims), _ptr(rays), len(rays), int(any_hit), int(robust),
                        _ptr(hits), _ptr(cnt))
    return hits, cnt


@pytest.mark.parametrize("scene", ["cornell", "soup2k", "terrain2k"])
@pytest.mark.parametrize("mode", ["serial_low", "serial_high", "parallel_med", "parallel_high"])
def test_compact_walk_equals_golden(walker, scene, mode):
    g = load_golden(scene)
    nodes, ids = parse_stream(g[f"bvh_{mode}"].tobytes())
    rc, pairs, recs = _encode(walker, nodes)
    assert rc == 0
    prims = np.zeros((len(ids), 12), dtype=np.float32)
    lib = oracle.load_oracle()
    prims = lib.precompute_tris(g["prims"], ids)
    for any_hit in (False, True):
        for robust in (False, True):
            key = f"{mode}_{'any' if any_hit else 'closest'}_{'robust' if robust else 'fast'}"
            if f"hits_{key}" not in g.files:
                continue
            rays = g["rays_shadow"] if any_hit else g["rays_closest"]
            hits, cnt = _walk(walker, nodes, pairs, recs, prims, rays, any_hit, robust)
            assert hits.tobytes() == g[f"hits_{key}"].tobytes(), key
            assert (cnt[:2] == g[f"counters_{key}"][:2]).all(), key
            assert cnt[2] + cnt[3] == cnt[0]


@pytest.mark.parametrize("gen,n", [("soup", 30000), ("terrain", 20000), ("sponza_proxy", 16384)])
def test_compact_walk_equals_oracle(walker, orc, gen, n):
    tris = getattr(synth, gen)(n)
    bb, cc = orc.prep_tris(tris)
    lo, hi = synth.scene_bounds(tris)
    rays_c = synth.rays_closest(4000, lo, hi, seed=11)
    rays_s = synth.rays_shadow(4000, lo, hi, seed=12)
    for builder, quality in ((oracle.BUILDER_DEFAULT_SERIAL, oracle.QUALITY_HIGH), (oracle.BUILDER_DEFAULT_PARALLEL, oracle.QUALITY_HIGH),
                             (oracle.BUILDER_DEFAULT_PARALLEL, oracle.QUALITY_LOW)):
        bvh = orc.build(bb, cc, builder=builder, quality=quality, threads=4)
        nodes, ids = bvh.nodes(), bvh.prim_ids()
        rc, pairs, recs = _encode(walker, nodes)
        assert rc == 0, "a pair of a builder-made tree is not representable"
        prims = orc.precompute_tris(tris, ids)
        for any_hit, rays in ((False, rays_c), (True, rays_s)):
            for robust in (False, True):
                ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
                hits, cnt = _walk(walker, nodes, pairs, recs, prims, rays, any_hit, robust)
                assert hits.tobytes() == ref_hits.tobytes()
                assert (cnt[:2] == ref_cnt[:2]).all()
        # what the experiment is about: 16-byte requests per visited pair, 4 with PairNode
        _, cnt = _walk(walker, nodes, pairs, recs, prims, rays_c, False, True)
        assert (2 * cnt[2] + 4 * cnt[3]) / (4.0 * cnt[0]) < 0.85


def test_compact_encode_rejects_foreign_boxes(walker):
    g = load_golden("soup2k")
    nodes, _ = parse_stream(g["bvh_serial_high"].tobytes())
    nodes = nodes.copy()
    victim = int(np.flatnonzero((nodes["index"] & 15) == 0)[5])       # an inner node below the root
    assert victim != 0
    b = nodes["bounds"][victim].copy()
    b[0] -= 1.0; b[1] += 1.0                                           # a hand-edited box: no child shares its x planes any more
    nodes["bounds"][victim] = b
    rc, _, _ = _encode(walker, nodes)
    assert rc & 1
    # signed zeros are fine: planes are compared numerically
    nodes2, _ = parse_stream(g["bvh_serial_high"].tobytes())
    nodes2 = nodes2.copy()
    nodes2["bounds"][nodes2["bounds"] == 0.0] = -0.0
    assert _encode(walker, nodes2)[0] == 0


ADVERSARIAL = [(seed, kind) for seed in range(8) for kind in ("lattice", "dups", "flat", "points", "scales", "uniform")]


@pytest.mark.parametrize("seed,kind", ADVERSARIAL)
def test_compact_walk_adversarial(walker, orc, seed, kind):
    """The generators of tests/test_gpu_fuzz.py: ties everywhere, duplicated primitives, planes at +-0 with origins snapped onto them,
    zero direction components of both signs, zero-area triangles, mixed magnitudes. This is where "a decoded plane may differ from
    the child's in the sign of a zero, and that cannot change a comparison" has to hold."""
    import test_gpu_fuzz as F
    rng = np.random.default_rng(7000 + 10 * seed + len(kind))
    n = int(rng.choice([2, 5, 17, 64, 65, 200, 1500, 4000]))
    tris = F._scene3(rng, n, kind, np.float32)
    bb, cc = orc.prep_tris(tris)
    lo = tris.reshape(-1, 3).min(axis=0).astype(np.float64)
    hi = tris.reshape(-1, 3).max(axis=0).astype(np.float64)
    rays = F._rays3(rng, 3000, lo, hi, np.float32)
    lim = [(1, 8), (1, 1), (2, 4), (3, 15)][seed % 4]
    for builder, quality in ((0, 0), (0, 2), (1, 2), (3, 0)):
        bvh = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=[1024, 64][seed % 2])
        nodes, ids = bvh.nodes(), bvh.prim_ids()
        if len(nodes) < 3:
            continue
        rc, pairs, recs = _encode(walker, nodes)
        assert rc == 0
        prims = orc.precompute_tris(tris, ids)
        for any_hit in (False, True):
            for robust in (False, True):
                ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
                hits, cnt = _walk(walker, nodes, pairs, recs, prims, rays, any_hit, robust)
                assert hits.tobytes() == ref_hits.tobytes(), (builder, quality, any_hit, robust)
                assert (cnt[:2] == ref_cnt[:2]).all()


# ---- the kernel BODY itself (bvh_amd/csrc/trace_body.inc), compiled for the host with one emulated lane ---------------------------

@pytest.fixture(scope="module")
def body(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("body") / "libtrace_body_host.so")
    src = os.path.join(ROOT, "tests", "cpp", "trace_body_host.cpp")
    cmd = ["g++", "-std=c++20", "-O1", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-Wno-unknown-pragmas", "-Werror", "-shared", "-fPIC", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dll = C.CDLL(out)
    dll.trace_body_host.restype = C.c_int
    dll.trace_body_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return dll


def _aligned(a, align=128):
    """A copy of `a` whose data pointer is `align`-byte aligned (the records are alignas(64) / alignas(32) on the device)."""
    raw = np.empty(a.nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def _run_body(body, nodes, pairs, recs, prims, rays, any_hit, robust, compact):
    rays = _aligned(np.ascontiguousarray(rays, dtype=np.float32))
    prims = _aligned(np.ascontiguousarray(prims, dtype=np.float32))
    pairs = _aligned(pairs)
    recs = _aligned(recs)
    hits = _aligned(np.zeros(len(rays), dtype=oracle.HITF))
    cnt = np.zeros(3, dtype=np.uint64)
    status = body.trace_body_host(_ptr(pairs), _ptr(recs) if compact else None, int(nodes["index"][0]), _ptr(prims), _ptr(rays), len(rays),
                                  int(any_hit), int(robust), _ptr(hits), _ptr(cnt))
    assert status == 0
    return hits, cnt


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("gen,n", [("soup", 20000), ("terrain", 20000), ("sponza_proxy", 16384)])
def test_kernel_body_on_host_equals_oracle(walker, body, orc, gen, n, compact):
    """trace_body.inc, both variants, one emulated lane: hits and (pairs, tests, leaves) counters equal the oracle's."""
    tris = getattr(synth, gen)(n)
    bb, cc = orc.prep_tris(tris)
    lo, hi = synth.scene_bounds(tris)
    rays_c = synth.rays_closest(3000, lo, hi, seed=21)
    rays_s = synth.rays_shadow(3000, lo, hi, seed=22)
    for builder, quality in ((oracle.BUILDER_DEFAULT_SERIAL, oracle.QUALITY_LOW), (oracle.BUILDER_DEFAULT_PARALLEL, oracle.QUALITY_HIGH)):
        bvh = orc.build(bb, cc, builder=builder, quality=quality, threads=4)
        nodes, ids = bvh.nodes(), bvh.prim_ids()
        rc, pairs, recs = _encode(walker, nodes)
        assert rc == 0
        prims = orc.precompute_tris(tris, ids)
        for any_hit, rays in ((False, rays_c), (True, rays_s)):
            for robust in (False, True):
                ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
                hits, cnt = _run_body(body, nodes, pairs, recs, prims, rays, any_hit, robust, compact)
                assert hits.tobytes() == ref_hits.tobytes(), (builder, quality, any_hit, robust)
                assert (cnt == ref_cnt).all()


@pytest.mark.parametrize("compact", [False, True])
def test_kernel_body_on_host_deep_stack(walker, body, orc, compact):
    """The chain-shaped tree of tests/test_gpu_traverse.py::test_trees_deeper_than_the_small_stack with 60 levels: the inner child is
    always nearer than its leaf sibling, so the stack takes one entry per level and goes past the 20 LDS entries into the scratch
    entries. The compact variant pops with a separate LDS read and a guarded scratch read, the PairNode variant through one select."""
    depth = 60
    n = depth + 1
    tris = np.zeros((n, 9), dtype=np.float32)
    for k in range(n):
        x = np.float32(4000 - k)
        tris[k] = [x, -1, -1, x, 1, -1, x, 0, 1]
    bb, _ = orc.prep_tris(tris)
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
    ref = orc.from_arrays(nodes, ids)
    rc, pairs, recs = _encode(walker, nodes)
    assert rc == 0
    prims = orc.precompute_tris(tris)
    rng = np.random.default_rng(depth)
    rays = np.zeros((3000, 8), dtype=np.float32)
    rays[:, 0] = rng.random(len(rays)) * 100                   # origins in front of the stack of triangles
    rays[:, 1:3] = (rng.random((len(rays), 2)) - 0.5) * 1.5
    rays[:, 3] = 1
    rays[:, 4:6] = (rng.random((len(rays), 2)) - 0.5) * 1e-4
    rays[:, 7] = np.finfo(np.float32).max
    rays[::7, 3] = -1                                          # some point away
    for any_hit in (False, True):
        for robust in (False, True):
            want, cw = ref.intersect_tri(prims, rays, any_hit, robust, counters=True)
            hits, cnt = _run_body(body, nodes, pairs, recs, prims, rays, any_hit, robust, compact)
            assert hits.tobytes() == want.tobytes(), (any_hit, robust)
            assert (cnt == cw).all()
    assert int((want["prim"] != oracle.INVALID).sum()) > 500
    assert int(cw[0]) > 40 * 1000                              # most rays walked (and stacked) the whole chain


@pytest.mark.parametrize("seed,kind", ADVERSARIAL[::2])
def test_kernel_body_on_host_adversarial(walker, body, orc, seed, kind):
    """The compact kernel body (one emulated lane) on the adversarial generators: ties, duplicates, signed-zero planes, axis-aligned
    rays, zero-area triangles, mixed magnitudes."""
    import test_gpu_fuzz as F
    rng = np.random.default_rng(9000 + 10 * seed + len(kind))
    n = int(rng.choice([2, 5, 17, 64, 65, 200, 1500, 4000]))
    tris = F._scene3(rng, n, kind, np.float32)
    bb, cc = orc.prep_tris(tris)
    lo = tris.reshape(-1, 3).min(axis=0).astype(np.float64)
    hi = tris.reshape(-1, 3).max(axis=0).astype(np.float64)
    rays = F._rays3(rng, 2000, lo, hi, np.float32)
    for builder, quality in ((0, 2), (1, 2)):
        bvh = orc.build(bb, cc, builder=builder, quality=quality, parallel_threshold=[1024, 64][seed % 2])
        nodes, ids = bvh.nodes(), bvh.prim_ids()
        if len(nodes) < 3:
            continue
        rc, pairs, recs = _encode(walker, nodes)
        assert rc == 0
        prims = orc.precompute_tris(tris, ids)
        for any_hit in (False, True):
            for robust in (False, True):
                ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
                hits, cnt = _run_body(body, nodes, pairs, recs, prims, rays, any_hit, robust, True)
                assert hits.tobytes() == ref_hits.tobytes(), (builder, quality, any_hit, robust)
                assert (cnt == ref_cnt).all()


@pytest.fixture(scope="module")
def body64(tmp_path_factory):
    """trace_body_host.cpp with -DBVH_HOST_WAVE64: a full wavefront of 64 fibers switching at the wave intrinsics, real thresholds."""
    out = str(tmp_path_factory.mktemp("body64") / "libtrace_body_host64.so")
    src = os.path.join(ROOT, "tests", "cpp", "trace_body_host.cpp")
    cmd = ["g++", "-std=c++20", "-O1", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-strict-aliasing", "-DBVH_HOST_WAVE64", "-Wall", "-Wextra",
           "-Wno-unused-parameter", "-Wno-unknown-pragmas", "-Werror", "-shared", "-fPIC", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    dll = C.CDLL(out)
    dll.trace_body_host.restype = C.c_int
    dll.trace_body_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return dll


@pytest.mark.parametrize("compact", [False, True])
def test_kernel_body_full_wavefront(walker, body64, orc, compact):
    """Both kernel texts as a full 64-lane wavefront (lanes hold different rays, some with a box and some without, some parked at
    leaves, refills of 54+ idle lanes at a time): hits and counters equal the oracle's."""
    tris = synth.soup(20000)
    bb, cc = orc.prep_tris(tris)
    lo, hi = synth.scene_bounds(tris)
    bvh = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=4)
    nodes, ids = bvh.nodes(), bvh.prim_ids()
    rc, pairs, recs = _encode(walker, nodes)
    assert rc == 0
    prims = orc.precompute_tris(tris, ids)
    for any_hit, robust, rays in ((False, True, synth.rays_closest(3000, lo, hi, seed=31)), (True, False, synth.rays_shadow(3000, lo, hi, seed=32))):
        ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
        hits, cnt = _run_body(body64, nodes, pairs, recs, prims, rays, any_hit, robust, compact)
        assert hits.tobytes() == ref_hits.tobytes()
        assert (cnt == ref_cnt).all()
