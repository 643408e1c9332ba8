"""Developer probe: closest-hit robust traversal with and without the library's internal ray reordering (BVH_AMD_RAY_SORTED).
    python tools/sorted_probe.py [soup|terrain|sponza|proc] [n_tris] [n_rays] [quality] [any]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
q = int(sys.argv[4]) if len(sys.argv) > 4 else 2
any_hit = len(sys.argv) > 5 and sys.argv[5] == "any"
tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy, "proc": synth.procedural_10m}[scene](n)
d_tris = torch.from_numpy(tris).cuda()
bb, cc = bvh_amd.tri_bounds(d_tris)
bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())
prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
lo, hi = synth.scene_bounds(tris)
rays = torch.from_numpy((synth.rays_shadow if any_hit else synth.rays_closest)(nr, lo, hi)).cuda()
out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
base = None
for sort_rays in (False, True):
    for _ in range(2):
        bvh_amd.intersect(bvh, prims, rays, any_hit, not any_hit, out=out, sort_rays=sort_rays)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        bvh_amd.intersect(bvh, prims, rays, any_hit, not any_hit, out=out, sort_rays=sort_rays)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    if base is None: base = out.clone()
    else: assert torch.equal(out.view(torch.int32), base.view(torch.int32))
    print(f"SORTED {scene} n={n} rays={nr} any={int(any_hit)} sort_rays={int(sort_rays)} {ms:8.3f} ms {nr / ms / 1e3:8.1f} Mrays/s", flush=True)
