"""Developer check: trace_kernel_wide against the pair kernel under both ray orders, listing differing rays."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bvh_amd
from bvh_amd import synth
lib = bvh_amd._lib.load()
for name, gen, n, seed, nr in (("sponza", synth.sponza_proxy, 262_144, 91, 2_097_152), ("terrain", synth.terrain, 1_000_000, 91, 2_097_152), ("soup", synth.soup, 1_000_000, 77, 4_194_304)):
    tris = gen(n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, gpu.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(nr, lo, hi, seed=seed)
    d_rays = torch.from_numpy(rays).cuda()
    for order in (True, False):
        lib.bvh_amd_tuning(12, 12, 1, -1)
        a = bvh_amd.hits_to_numpy(bvh_amd.intersect(gpu, prims, d_rays, any_hit=False, robust=True, sort_rays=order)).copy()
        lib.bvh_amd_tuning(12, 12, 2, -1)
        b = bvh_amd.hits_to_numpy(bvh_amd.intersect(gpu, prims, d_rays, any_hit=False, robust=True, sort_rays=order)).copy()
        k = lib.bvh_amd_last_kernel_name().decode()
        lib.bvh_amd_tuning(-1, -1, -1, -1)
        bad = np.flatnonzero((a["prim"] != b["prim"]) | (a["t"] != b["t"]) | (a["u"] != b["u"]) | (a["v"] != b["v"]))
        print(name, "reordered" if order else "as given", k, "differing rays:", len(bad), "of", len(rays), flush=True)
        for i in bad[:4]:
            print("   ray", i, rays[i], "pairs:", a[i], "wide:", b[i])
