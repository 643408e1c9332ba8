"""The reference's per-ray entry points with a HOST leaf callback (c_api/bvh.h:277-295, Bvh::intersect of bvh.h:160-182) served by
the device walk of traverse.hip (ray_step_kernel), and the reference's own four example programs compiled unmodified against
this repo's headers ("must compile/link unmodified").

CPU part: everything compiles and links; without a GPU the programs fail loudly. GPU part: results equal the oracle's, and
the number of leaf callbacks equals the oracle's leaf-visit counter (same visit sequence, same culling)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from bvh_amd import synth

PROGS = os.path.join(ROOT, "oracle", "_ref", "progs")
HAVE_REFERENCE = os.path.isdir("/root/reference/test")


def _compile_c(out):
    from bvh_amd import build
    build.build()
    lib = os.path.join(ROOT, "bvh_amd", "lib")
    cmd = ["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "ray_callback.c"), "-L", lib, "-lbvh_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib",
           "-lm", "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _prog(name):
    path = os.path.join(PROGS, name)
    if not os.path.exists(path):
        if not HAVE_REFERENCE:
            pytest.skip("oracle/_ref/progs not built (they are compiled where /root/reference exists and travel with the repo)")
        import oracle
        oracle.build_checkers()
    assert os.path.exists(path)
    return path


def _write_obj(path, tris):
    """OBJ with one `v` per corner; %.9g round-trips a float32 through strtof (load_obj.cpp:69-73)."""
    with open(path, "w") as f:
        for t in tris.reshape(-1, 3):
            f.write("v %.9g %.9g %.9g\n" % tuple(float(x) for x in t))
        for i in range(len(tris)):
            f.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")


def test_callback_programs_compile_and_fail_loudly_without_a_gpu(tmp_path):
    exe = _compile_c(str(tmp_path / "ray_callback"))
    import torch
    if torch.cuda.is_available():
        return
    inp = tmp_path / "in.bin"
    tris = synth.soup(8)
    rays = np.zeros((1, 8), dtype=np.float32)
    inp.write_bytes(np.array([len(tris), 1], dtype=np.uint64).tobytes() + tris.tobytes() + rays.tobytes())
    r = subprocess.run([exe, "3f", "closest", "0", str(inp), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "no ROCm-capable device" in r.stderr          # bvh3f_build says why; no CPU fallback
    if HAVE_REFERENCE:                                        # the reference's own programs, unmodified, against our headers
        for name in ("simple_example", "serialize", "benchmark", "c_api_example"):
            _prog(name)
        r = subprocess.run([_prog("simple_example")], capture_output=True, text=True)
        assert r.returncode != 0 and "no ROCm-capable device" in r.stderr


CASES = [("3f", "closest", 0, 1), ("3f", "closest", 1, 1), ("3f", "any", 0, 1), ("3d", "closest", 1, 1), ("3d", "any", 1, 1),
         ("2f", "closest", 0, 1), ("2f", "any", 1, 1), ("2d", "closest", 1, 1),
         ("3f", "closest", 1, 8), ("2d", "any", 0, 5)]          # several host threads through one bvh


@pytest.mark.gpu
@pytest.mark.parametrize("family,mode,robust,threads", CASES)
def test_c_callback_api_matches_oracle(tmp_path, orc, family, mode, robust, threads):
    import oracle
    exe = _compile_c(str(tmp_path / "ray_callback"))
    dt = np.float32 if family[1] == "f" else np.float64
    any_hit = mode == "any"
    m = 3000
    if family[0] == "3":
        prims = synth.sponza_proxy(20000).astype(dt)
        lo, hi = synth.scene_bounds(prims)
        rays = (synth.rays_shadow(m, lo, hi) if any_hit else synth.rays_closest(m, lo, hi)).astype(dt)
        bb, cc = orc.prep_tris(prims)
        ob = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)        # bvh3X_build(pool, ..., NULL)
        want, cnt = ob.intersect_tri(orc.precompute_tris(prims, ob.prim_ids()), rays, any_hit, robust, counters=True)
    else:
        prims = synth.circles(4000, dtype=dt)
        rays = synth.rays_2d(m, dtype=dt, segment=any_hit)
        bb, cc = orc.sphere_bboxes(prims)
        ob = orc.build(bb, cc, quality=oracle.QUALITY_HIGH)                                                 # bvh2X_build(NULL, ..., NULL)
        want, cnt = ob.intersect_sphere(prims[ob.prim_ids().astype(np.int64)], rays, any_hit, robust, counters=True)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(np.array([len(prims), m], dtype=np.uint64).tobytes() + prims.tobytes() + rays.tobytes())
    r = subprocess.run([exe, family, mode, str(robust), str(inp), str(outp), str(threads)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = outp.read_bytes()
    got = np.frombuffer(raw[:16 * m], dtype=[("prim", "<i8"), ("t", "<f8")])
    calls = int(np.frombuffer(raw[16 * m:], dtype="<u8")[0])
    ids = ob.prim_ids().astype(np.int64)
    hit = want["prim"] != 0xFFFFFFFF
    want_prim = np.where(hit, ids[np.minimum(want["prim"], len(ids) - 1)], -1)
    assert int(hit.sum()) > m // 20
    assert (got["prim"] == want_prim).all()
    assert (got["t"][hit] == want["t"][hit].astype(np.float64)).all()          # bit-exact t (the parity bar is 1e-5)
    assert (got["t"][~hit] == rays[~hit, -1].astype(np.float64)).all()         # a miss leaves tmax alone
    assert calls == int(cnt[2])                               # one callback per leaf the reference visits: same walk, same culling


@pytest.mark.gpu
def test_reference_examples_run_unmodified(tmp_path, orc):
    """test/simple_example.cpp, test/serialize.cpp, test/benchmark.cpp and test/c_api_example.c as the reference ships them
    (binaries built by oracle/Makefile `refprogs` against include/bvh/v2 + libbvh_amd.so): their de-facto golden outputs
    (SURVEY.md 8c / Appendix B)."""
    import bvh_amd
    r = subprocess.run([_prog("simple_example")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "primitive: 1" in r.stdout and "distance: 1" in r.stdout and "barycentric coords.: -0, 0.5" in r.stdout

    r = subprocess.run([_prog("serialize")], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode == 0 and "The deserialized BVH is the same as the original one" in r.stdout, r.stdout + r.stderr
    tris = np.array([[1, -1, 1, 1, 1, 1, -1, 1, 1], [1, -1, 1, -1, -1, 1, -1, 1, 1]], dtype=np.float32)
    bb, cc = orc.prep_tris(tris)
    stream = (tmp_path / "bvh.bin").read_bytes()
    assert len(stream) == 44 and stream == orc.build(bb, cc, quality=2).serialize()      # Appendix B's 44-byte stream

    # benchmark / c_api_example: the Cornell box of the reference's ctest at a reduced resolution (one launch chain per pixel)
    tris = load_golden("cornell")["prims"]
    obj = tmp_path / "cornell.obj"
    _write_obj(str(obj), tris)
    W = H = 96
    cam = ["--eye", "0", "1", "2", "--dir", "0", "0", "-1", "--up", "0", "1", "0", "--width", str(W), "--height", str(H)]
    r = subprocess.run([_prog("benchmark"), str(obj)] + cam, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    # expected image: the batch path, already pinned to the reference's 1024x1024 md5 (test_gpu_traverse.py)
    d_bb, d_cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    d_rays = bvh_amd.pinhole_rays(W, H, (0, 1, 2), (0, 0, -1), (0, 1, 0))
    d_hits, cnt = bvh_amd.intersect(bvh, prims, d_rays, any_hit=False, robust=False, counters=True)
    cnt = cnt.cpu().numpy()
    img = bvh_amd.shade_eyelight(prims, d_rays, d_hits).cpu().numpy().reshape(H, W, 3)
    want = f"P6 {W} {H} 255\n".encode() + img[::-1].tobytes()
    got = (tmp_path / "render.ppm").read_bytes()
    assert hashlib.md5(got).hexdigest() == hashlib.md5(want).hexdigest()
    hits = bvh_amd.hits_to_numpy(d_hits)
    n_hit = int((hits["prim"] != bvh_amd.INVALID).sum())
    assert f"{n_hit} intersection(s) found" in r.stdout
    assert f"{bvh.node_count} node(s)" in r.stdout

    # --render-mode debug counts through the InnerFn of Bvh::intersect (test/benchmark.cpp:292-296): pairs and leaves of the walk
    r = subprocess.run([_prog("benchmark"), str(obj)] + cam + ["--render-mode", "debug"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"Traversal visited {int(cnt[0])} nodes and {int(cnt[2])} leaves" in r.stdout, r.stdout

    # c_api_example: --width/--height too; it starts its camera at the origin looking down +z by default
    r = subprocess.run([_prog("c_api_example"), str(obj)] + cam, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{n_hit} intersection(s) found" in r.stdout and f"{bvh.node_count} node(s)" in r.stdout


def _tri_leaf(prims12, ray, f):
    """The closest-hit leaf loop of test/benchmark.cpp:281-291 over PrecomputedTri records, in numpy scalars of type f
    (one rounding per operation, like the C code)."""
    org, d = ray[0:3], ray[3:6]
    state = {"prim": -1}

    def dot(a, b):
        return ((f(0) + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]

    def cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f)

    def leaf(tmax, begin, end):
        hit = False
        for i in range(begin, end):
            p0, e1, e2, n = prims12[i, 0:3], prims12[i, 3:6], prims12[i, 6:9], prims12[i, 9:12]
            c = p0 - org
            r = cross(d, c)
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                inv_det = f(1) / dot(n, d)
                u, v = dot(r, e2) * inv_det, dot(r, e1) * inv_det
                w = f(1) - u - v
                tol = -np.finfo(f).eps
                if u >= tol and v >= tol and w >= tol:
                    t = dot(n, c) * inv_det
                    if ray[6] <= t <= tmax:
                        tmax, state["prim"], hit = t, i, True
        return hit, tmax
    return leaf, state


@pytest.mark.gpu
def test_python_intersect_ray_inner_callback_start_and_threads(orc):
    """Bvh.intersect_ray (bvhXX_intersect_ray_visit): inner callback counts = the reference's pair visits, a walk started at
    the root's children covers what a walk from the root covers, callbacks may raise, and the entry point is re-entrant."""
    import threading
    import oracle
    import bvh_amd
    f = np.float32
    tris = synth.sponza_proxy(20000)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())
    obb, occ = orc.prep_tris(tris)
    ob = orc.build(obb, occ, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
    assert bvh.serialize() == ob.serialize()
    prims = orc.precompute_tris(tris, ob.prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(240, lo, hi)
    for robust in (False, True):
        want, cnt = ob.intersect_tri(prims, rays, False, robust, counters=True)
        pairs = leaves = 0
        for j, ray in enumerate(rays):
            leaf, state = _tri_leaf(prims, ray, f)
            seen = []
            calls = [0]

            def counted(tmax, begin, end, leaf=leaf, calls=calls):
                calls[0] += 1
                return leaf(tmax, begin, end)
            bvh.intersect_ray(ray, counted, robust=robust, inner_fn=seen.append)
            pairs += len(seen)
            leaves += calls[0]
            assert state["prim"] == (int(want["prim"][j]) if want["prim"][j] != oracle.INVALID else -1)
        assert (pairs, leaves) == (int(cnt[0]), int(cnt[2]))

    # start: the closest hit below the root = the closer of the closest hits below its two children
    nodes = bvh.nodes
    first = int(nodes[0]["index"]) >> 4
    want = ob.intersect_tri(prims, rays, False, True)
    for j, ray in enumerate(rays[:60]):
        best_t, best = ray[7], -1
        for child in (first, first + 1):
            leaf, state = _tri_leaf(prims, ray, f)
            tbox = [ray[7]]

            def keep(tmax, begin, end, leaf=leaf, tbox=tbox):
                hit, tbox[0] = leaf(tmax, begin, end)
                return hit, tbox[0]
            bvh.intersect_ray(ray, keep, robust=True, start=int(nodes[child]["index"]))
            if state["prim"] >= 0 and tbox[0] < best_t:
                best_t, best = tbox[0], state["prim"]
        if want["prim"][j] == oracle.INVALID:
            assert best == -1
        else:
            assert best_t == want["t"][j]

    class Boom(Exception):
        pass

    def explode(tmax, begin, end):
        raise Boom()
    with pytest.raises(Boom):
        bvh.intersect_ray(rays[int(np.argmax(want["prim"] != oracle.INVALID))], explode, robust=True)

    # four threads at once, each with its own stream and buffers
    results = [None] * 4

    def work(k):
        out = []
        for ray in rays[k::4][:40]:
            leaf, state = _tri_leaf(prims, ray, f)
            bvh.intersect_ray(ray, leaf, robust=True)
            out.append(state["prim"])
        results[k] = out
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(4):
        w = want["prim"][k::4][:40]
        assert results[k] == [int(x) if x != oracle.INVALID else -1 for x in w]


@pytest.mark.gpu
def test_callback_walk_on_a_tree_deeper_than_the_small_stack(restatement):
    """A chain of 300 levels: the device keeps as much stack as the tree needs (the reference's GrowingStack), and a log that
    fills up mid-walk continues from the stack the device kept (checker: the restatement; oracle/_ref's harness walks with SmallStack<Index, 64>)."""
    import oracle
    import bvh_amd
    depth = 300
    n = depth + 1
    tris = np.zeros((n, 9), dtype=np.float32)
    for k in range(n):
        x = np.float32(4000 - k)
        tris[k] = [x, -1, -1, x, 1, -1, x, 0, 1]
    bb, _ = restatement.prep_tris(tris)
    nodes = np.zeros(2 * n - 1, dtype=oracle.NODEF)
    suffix = bb.copy()
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
    rng = np.random.default_rng(5)
    rays = np.zeros((40, 8), dtype=np.float32)
    rays[:, 0] = rng.random(len(rays)) * 100
    rays[:, 1:3] = (rng.random((len(rays), 2)) - 0.5) * 1.5
    rays[:, 3] = 1
    rays[:, 4:6] = (rng.random((len(rays), 2)) - 0.5) * 1e-4
    rays[:, 7] = np.finfo(np.float32).max
    for any_hit, log_cap in ((False, None), (True, None), (False, "7"), (False, "1")):
        want, cnt = ref.intersect_tri(prims, rays, any_hit, True, counters=True)
        pairs = leaves = 0
        lib = bvh_amd._lib.load()
        lib.bvh_amd_experiment(b"step_events", int(log_cap) if log_cap else -1)   # a log of 7 (or 1) events per launch
        for j, ray in enumerate(rays):
            leaf, state = _tri_leaf(prims, ray, np.float32)
            seen, calls = [], [0]

            def counted(tmax, begin, end, leaf=leaf, calls=calls):
                calls[0] += 1
                return leaf(tmax, begin, end)
            try:
                gpu.intersect_ray(ray, counted, any_hit=any_hit, robust=True, inner_fn=seen.append)
            finally:
                if j == len(rays) - 1:
                    lib.bvh_amd_experiment(b"step_events", -1)
            pairs += len(seen)
            leaves += calls[0]
            assert state["prim"] == (int(want["prim"][j]) if want["prim"][j] != oracle.INVALID else -1)
        assert (pairs, leaves) == (int(cnt[0]), int(cnt[2]))


@pytest.mark.gpu
def test_nested_per_ray_walks_two_level_scene(orc):
    """A leaf callback that itself traces another BVH (two-level / instanced scenes: what the reference's stack-local
    Bvh::intersect allows, bvh_impl.h:244). The inner walk must not disturb the outer walk's log, snapshots or buffers: the
    outer callback sequence with nesting equals the sequence without, and every inner result equals the standalone one — also
    when the inner BVH is much deeper than the outer one (its context has to be its own, not a re-allocation of the outer's)."""
    import oracle
    import bvh_amd
    f = np.float32
    outer_tris = synth.sponza_proxy(20000)
    inner_tris = synth.soup(30000, seed=17, jitter=0.02)
    lo, hi = synth.scene_bounds(outer_tris)
    inner_tris = (inner_tris.reshape(-1, 3) * (hi - lo).astype(f) + lo.astype(f)).reshape(-1, 9).astype(f)     # same place as the outer scene
    bvhs, prims = [], []
    for tris, q in ((outer_tris, bvh_amd.Quality.Low), (inner_tris, bvh_amd.Quality.High)):
        bb, cc = bvh_amd.tri_bounds(tris)
        b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=q))
        bvhs.append(b)
        prims.append(orc.precompute_tris(tris, b.prim_ids))
    outer, inner = bvhs
    rays = synth.rays_closest(120, lo, hi)
    n_nested = 0
    for ray in rays:
        # standalone answers first
        leaf_i, state_i = _tri_leaf(prims[1], ray, f)
        inner.intersect_ray(ray, leaf_i, robust=True)
        alone_inner = dict(state_i)
        plain_calls = []
        leaf_o, state_o = _tri_leaf(prims[0], ray, f)

        def plain(tmax, begin, end, leaf=leaf_o, log=plain_calls):
            log.append((begin, end, float(tmax)))
            return leaf(tmax, begin, end)
        outer.intersect_ray(ray, plain, robust=True)
        alone_outer = dict(state_o)
        # now the same outer walk with an inner walk inside every leaf callback
        nested_calls = []
        leaf_o2, state_o2 = _tri_leaf(prims[0], ray, f)

        def nested(tmax, begin, end, leaf=leaf_o2, log=nested_calls):
            log.append((begin, end, float(tmax)))
            leaf_n, state_n = _tri_leaf(prims[1], ray, f)
            inner.intersect_ray(ray, leaf_n, robust=True)                 # re-enters the per-ray API on this thread
            assert state_n == alone_inner
            return leaf(tmax, begin, end)
        outer.intersect_ray(ray, nested, robust=True)
        assert nested_calls == plain_calls and state_o2 == alone_outer
        n_nested += len(nested_calls)
    assert n_nested > 100
