"""Dumps what tools/src/wide4_model.cpp walks: the reference-built tree (oracle = test infrastructure; here a developer's model), the triangles
and a sample of the bench's rays.    python tools/wide4_model.py [soup|sponza|terrain] [n_tris] [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from bvh_amd import synth

scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
cpu = oracle.gpu_checker()
tris = {"soup": synth.soup, "sponza": synth.sponza_proxy, "terrain": synth.terrain}[scene](n)
bb, cc = cpu.prep_tris(tris)
ref = cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
lo, hi = synth.scene_bounds(tris)
base = f"/tmp/wide4_{scene}"
ref.nodes().tofile(base + ".nodes")
ref.prim_ids().astype(np.uint64).tofile(base + ".prim_ids")
np.ascontiguousarray(tris, dtype=np.float32).tofile(base + ".tris")
synth.rays_closest(nr, lo, hi).astype(np.float32).tofile(base + ".rays")
print(base, ref.node_count, "nodes")
