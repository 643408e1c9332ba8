"""Generates the committed golden fixtures from the UNMODIFIED reference.

Run in the authoring container only (needs /root/reference and oracle/_ref/libbvh_ref.so):

    python tests/golden/make_golden.py

Every array below is an output of the reference library itself (through oracle/ref_harness.cpp,
which only #includes the reference headers), so these files pin the oracle restatement
(oracle/bvh_oracle.cpp) and, through it, the HIP path. Inputs are regenerated from bvh_amd/synth.py,
except the Cornell box, whose 36 triangles are read from the reference's OBJ with the semantics of
the reference loader (test/load_obj.cpp:57-96: `v` = 3 x strtof, `f` = fan triangulation, negative
indices relative to the current vertex count) and stored as data.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from bvh_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MODES = [("binned", 2, 0), ("sweep", 3, 0),
         ("serial_low", 0, 0), ("serial_med", 0, 1), ("serial_high", 0, 2),
         ("parallel_low", 1, 0), ("parallel_med", 1, 1), ("parallel_high", 1, 2)]


def load_obj(path):
    verts, tris = [], []
    for line in open(path):
        s = line.strip()
        if not s or s[0] == "#":
            continue
        tok = s.split()
        if tok[0] == "v":
            verts.append([np.float32(t) for t in tok[1:4]])
        elif tok[0] == "f":
            idx = []
            for t in tok[1:]:
                k = int(t.split("/")[0])
                idx.append(len(verts) + k if k < 0 else k - 1)
            for i in range(2, len(idx)):
                tris.append(verts[idx[0]] + verts[idx[i - 1]] + verts[idx[i]])
    return np.asarray(tris, dtype=np.float32)


def scene_fixture(ref, name, prims, n_rays, dtype, kind="tri"):
    fx = {"prims": prims}
    if kind == "tri":
        bb, cc = ref.prep_tris(prims)
    else:
        bb, cc = ref.sphere_bboxes(prims)
    fx["bboxes"], fx["centers"] = bb, cc
    lo, hi = synth.scene_bounds(prims)
    rays = synth.rays_closest(n_rays, lo, hi, dtype=dtype)
    srays = synth.rays_shadow(n_rays, lo, hi, dtype=dtype)
    fx["rays_closest"], fx["rays_shadow"] = rays, srays
    for mode, builder, quality in MODES:
        bvh = ref.build(bb, cc, builder=builder, quality=quality, threads=3)
        fx[f"bvh_{mode}"] = np.frombuffer(bvh.serialize(), dtype=np.uint8)
        if mode in ("serial_low", "parallel_high"):
            if kind == "tri":
                pp = ref.precompute_tris(prims, bvh.prim_ids())
                isect = bvh.intersect_tri
            else:
                pp = prims[bvh.prim_ids().astype(np.int64)]
                isect = bvh.intersect_sphere
            for any_hit in (0, 1):
                for robust in (0, 1):
                    hits, cnt = isect(pp, srays if any_hit else rays, any_hit, robust, threads=1, counters=True)
                    key = f"{mode}_{'any' if any_hit else 'closest'}_{'robust' if robust else 'fast'}"
                    fx[f"hits_{key}"] = hits
                    fx[f"counters_{key}"] = cnt
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in fx.items() if k.startswith("bvh_")})


def template_knob_fixture(ref):
    """Template arguments other than the reference's defaults, on the inputs of the scene fixtures above -> template_knobs.npz:
    BinnedSahBuilder<Node, BinCount> (binned_sah_builder.h:18) for BinCount = 4, 16, 32 (the scene files hold the default of 8 under
    `bvh_binned`), and Node<float, 3, IndexBits, PrimCountBits> (node.h:21-22) as Node<float, 3, 32, 2> (SweepSahBuilder, leaves <= 3)
    and Node<float, 3, 64, 6> (DefaultBuilder serial High, and extract_bvh of the root's first child): the reference's own template
    instantiations through oracle/ref_harness.cpp."""
    import ctypes as C
    scenes = {"soup2k": (synth.soup(2048, seed=3, jitter=0.03), "tri"),
              "terrain2k": (synth.terrain(2048), "tri"),
              "soup2k_f64": (synth.soup(2048, seed=3, jitter=0.03, dtype=np.float64), "tri"),
              "circles2k_2f": (synth.circles(2048, dtype=np.float32, rmin=0.002, rmax=0.02), "sphere")}
    fx = {}
    try:
        for name, (prims, kind) in scenes.items():
            bb, cc = ref.prep_tris(prims) if kind == "tri" else ref.sphere_bboxes(prims)
            for bins in (4, 16, 32):
                ref.set_bin_count(bins)
                fx[f"{name}_bins{bins}"] = np.frombuffer(ref.build(bb, cc, builder=oracle.BUILDER_BINNED).serialize(), dtype=np.uint8)
                fx[f"{name}_bins{bins}_leaf2to5"] = np.frombuffer(ref.build(bb, cc, builder=oracle.BUILDER_BINNED, min_leaf=2, max_leaf=5).serialize(), dtype=np.uint8)
    finally:
        ref.set_bin_count(8)
    f = ref.dll.ref_index_variant_stream
    f.restype, f.argtypes = C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    bb, cc = ref.prep_tris(scenes["soup2k"][0])
    for key, variant, max_leaf in (("count2bits_sweep", 0, 3), ("index64_high", 1, 8), ("index64_high_sub", 2, 8)):
        size = f(bb.ctypes.data, cc.ctypes.data, len(bb), variant, max_leaf, None, 0)
        buf = np.zeros(size, dtype=np.uint8)
        f(bb.ctypes.data, cc.ctypes.data, len(bb), variant, max_leaf, buf.ctypes.data, size)
        fx[f"soup2k_{key}"] = buf
    np.savez_compressed(os.path.join(OUT, "template_knobs.npz"), **fx)
    print("template_knobs", {k: v.shape for k, v in fx.items()})


def main():
    ref = oracle.load_ref()
    assert ref is not None, "needs /root/reference"
    if "--knobs" in sys.argv:                                 # only the template-knob fixture (the others are unchanged)
        template_knob_fixture(ref)
        return

    cornell = load_obj("/root/reference/test/scenes/cornell_box.obj")
    assert cornell.shape == (36, 9)
    scene_fixture(ref, "cornell", cornell, 4096, np.float32)
    scene_fixture(ref, "soup2k", synth.soup(2048, seed=3, jitter=0.03), 4096, np.float32)
    scene_fixture(ref, "terrain2k", synth.terrain(2048), 4096, np.float32)
    scene_fixture(ref, "spheres2k_f64", synth.spheres(2048, rmin=0.01, rmax=0.04), 4096, np.float64, kind="sphere")
    scene_fixture(ref, "soup2k_f64", synth.soup(2048, seed=3, jitter=0.03, dtype=np.float64), 2048, np.float64)

    # --- the 2D families (Node<T, 2>, circles): serial builders only (with a pool and n >= parallel_threshold the reference
    # reads p[2] of 2D points, mini_tree_builder.h:183) ---------------------------------------------------------------------
    for name, dt in (("circles2k_2f", np.float32), ("circles2k_2d", np.float64)):
        circ = synth.circles(2048, dtype=dt, rmin=0.002, rmax=0.02)
        fx = {"prims": circ}
        fx["bboxes"], fx["centers"] = ref.sphere_bboxes(circ)
        fx["rays_closest"], fx["rays_shadow"] = synth.rays_2d(4096, dtype=dt), synth.rays_2d(4096, dtype=dt, seed=4321, segment=True)
        for mode, builder, quality in MODES[:5]:
            bvh = ref.build(fx["bboxes"], fx["centers"], builder=builder, quality=quality)
            fx[f"bvh_{mode}"] = np.frombuffer(bvh.serialize(), dtype=np.uint8)
            if mode in ("serial_low", "serial_high"):
                pp = circ[bvh.prim_ids().astype(np.int64)]
                for any_hit in (0, 1):
                    for robust in (0, 1):
                        hits, cnt = bvh.intersect_sphere(pp, fx["rays_shadow"] if any_hit else fx["rays_closest"], any_hit, robust, counters=True)
                        key = f"{mode}_{'any' if any_hit else 'closest'}_{'robust' if robust else 'fast'}"
                        fx[f"hits_{key}"], fx[f"counters_{key}"] = hits, cnt
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
        print(name, {k: v.shape for k, v in fx.items() if k.startswith("bvh_")})

    # --- the reference's own known answers -------------------------------
    ka = {}
    # test/simple_example.cpp:25-35, :70-75: two triangles, one ray, parallel High build, fast traversal
    tris = np.array([[1, -1, 1, 1, 1, 1, -1, 1, 1], [1, -1, 1, -1, -1, 1, -1, 1, 1]], dtype=np.float32)
    bb, cc = ref.prep_tris(tris)
    bvh = ref.build(bb, cc, builder=1, quality=2, threads=2)
    pt = ref.precompute_tris(tris, bvh.prim_ids())
    ray = np.array([[0, 0, 0, 0, 0, 1, 0, 100]], dtype=np.float32)
    ka["simple_tris"], ka["simple_ray"] = tris, ray
    ka["simple_hit"] = bvh.intersect_tri(pt, ray, 0, 0)
    # test/serialize.cpp: same two triangles, serial default (High) build -> 44-byte stream
    ka["serialize_stream"] = np.frombuffer(ref.build(bb, cc, builder=0, quality=2).serialize(), dtype=np.uint8)
    # ctest `benchmark cornell_box.obj --eye 0 1 2 --dir 0 0 -1 --up 0 1 0`: 1024x1024 primary rays
    bb, cc = ref.prep_tris(cornell)
    rays = synth.rays_pinhole(1024, 1024, (0, 1, 2), (0, 0, -1), (0, 1, 0))
    counts, nodes, digests = [], [], []
    for q in (0, 1, 2):
        bvh = ref.build(bb, cc, builder=1, quality=q, threads=2)
        pt = ref.precompute_tris(cornell, bvh.prim_ids())
        hits, cnt = bvh.intersect_tri(pt, rays, 0, 0, threads=4, counters=True)
        orig = np.where(hits["prim"] != oracle.INVALID, bvh.prim_ids()[np.minimum(hits["prim"], 35)], 2**32 - 1)
        counts.append(int((hits["prim"] != oracle.INVALID).sum()))
        nodes.append(bvh.node_count)
        digests.append(hashlib.sha256(orig.astype(np.uint32).tobytes() + hits["t"].tobytes()).hexdigest())
        if q == 2:
            ka["cornell_render_counters_high"] = cnt
    ka["cornell_render_hits"] = np.array(counts)
    ka["cornell_render_nodes"] = np.array(nodes)
    ka["cornell_render_sha256"] = np.array(digests)
    np.savez_compressed(os.path.join(OUT, "known_answers.npz"), **ka)
    print("known answers:", ka["simple_hit"], bytes(ka["serialize_stream"]).hex(), counts, nodes,
          ka["cornell_render_counters_high"])
    template_knob_fixture(ref)


def c_api_symbol_list():
    """The 94 function names the reference's C header declares (src/bvh/v2/c_api/bvh.h) -> tests/golden/c_api_symbols.txt:
    the drop-in library must export every one of them."""
    import re
    header = open("/root/reference/src/bvh/v2/c_api/bvh.h").read()
    names = sorted(set(re.findall(r"BVH_API[^;]*?\b(bvh\w+)\s*\(", header)))
    assert len(names) == 94
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_api_symbols.txt"), "w") as f:
        f.write("\n".join(names) + "\n")


if __name__ == "__main__":
    if "--symbols" in sys.argv:
        c_api_symbol_list()
    else:
        main()
