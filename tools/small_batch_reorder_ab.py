"""Developer probe: does reordering pay BELOW 2^20 rays on a tree the L2s cannot hold? configs[4] (1M f64 spheres, parallel High) and the
1M-triangle soup, batches of 2^18 .. 2^21 and exactly 1,000,000 rays, traced as given / reordered / library default; hits compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth


def timed(fn, reps=8, warm=4):
    for _ in range(warm):
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def scene_spheres():
    sph_h = synth.spheres(1_000_000)
    sph = torch.from_numpy(sph_h).cuda()
    bb, cc = bvh_amd.sphere_bounds(sph)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    return "1M f64 spheres", bvh, bvh_amd.gather(sph, bvh.device_prim_ids()), synth.scene_bounds(sph_h), np.float64, "sphere"


def scene_soup():
    t_h = synth.soup(1_000_000)
    tris = torch.from_numpy(t_h).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
    return "1M-triangle soup f32", bvh, bvh_amd.precompute_tris(tris, bvh.device_prim_ids()), synth.scene_bounds(t_h), np.float32, "tri"


for make in (scene_spheres, scene_soup):
    name, bvh, prims, (lo, hi), dt, leaf = make()
    for n in (1 << 18, 1 << 19, 1_000_000, 1 << 20, 1 << 21):
        rays = torch.from_numpy(synth.rays_closest(n, lo, hi, dtype=dt)).cuda()
        out = {}
        ms = {}
        for label, sort in (("as given", False), ("reordered", True), ("default", None)):
            buf = torch.empty((n, 4), dtype=rays.dtype, device="cuda")
            ms[label] = timed(lambda: bvh_amd.intersect(bvh, prims, rays, False, True, leaf=leaf, out=buf, sort_rays=sort))
            out[label] = buf
        same = bool((out["as given"].view(torch.int32) == out["reordered"].view(torch.int32)).all())
        print(f"{name:22s} {n:8d} rays: as given {ms['as given']:.4f} ms, reordered {ms['reordered']:.4f} ms, default {ms['default']:.4f} ms"
              f"  ({n / ms['as given'] / 1e3:.0f} / {n / ms['reordered'] / 1e3:.0f} / {n / ms['default'] / 1e3:.0f} Mrays/s; hits equal: {same})", flush=True)
