"""Developer check of trace_kernel_wide (two tree levels per fetched record) against the pair-by-pair kernels: hit records byte-equal,
kernel time of both, on the soup, the Sponza proxy and the terrain.    python tools/wide_check.py [log2_rays]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bvh_amd
from bvh_amd import synth
lib = bvh_amd._lib.load()
n_rays = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
for name, gen, n in (("soup_1m", synth.soup, 1_000_000), ("sponza_262k", synth.sponza_proxy, 262_144), ("terrain_1m", synth.terrain, 1_000_000), ("soup_10m", synth.soup, 10_000_000)):
    if os.environ.get("WIDE_CHECK_ONLY") not in (None, name):
        continue
    tris = gen(n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(n_rays, lo, hi, seed=4321)).cuda()
    out = {}
    for label, coop in (("pairs", 1), ("wide", 2)):
        lib.bvh_amd_tuning(12, 12, coop, -1)
        hits = torch.empty((n_rays, 4), dtype=torch.float32, device="cuda")
        for _ in range(3):
            bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=True, out=hits, sort_rays=True)
        torch.cuda.synchronize()
        lib.bvh_amd_kernel_timing(1)
        for _ in range(5):
            bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=True, out=hits, sort_rays=True)
        torch.cuda.synchronize()
        kt = (C.c_float * 16)(); got = C.c_size_t(0)
        lib.bvh_amd_kernel_times(kt, 5, C.byref(got))
        lib.bvh_amd_kernel_timing(0)
        out[label] = (bvh_amd.hits_to_numpy(hits).tobytes(), float(np.mean(kt[:got.value])), lib.bvh_amd_last_kernel_name().decode())
    lib.bvh_amd_tuning(-1, -1, -1, -1)
    a, b = np.frombuffer(out["pairs"][0], dtype=np.uint32).reshape(-1, 4), np.frombuffer(out["wide"][0], dtype=np.uint32).reshape(-1, 4)
    diff = int((a != b).any(axis=1).sum())
    print(f"{name}: {n_rays} rays; pairs {out['pairs'][1]:.3f} ms ({out['pairs'][2]}), wide {out['wide'][1]:.3f} ms ({out['wide'][2]}): "
          f"{out['pairs'][1] / out['wide'][1]:.2f}x; hit records differing: {diff}", flush=True)
    if diff:
        bad = np.flatnonzero((a != b).any(axis=1))[:5]
        for i in bad: print("   ray", i, "pairs", a[i], a[i].view(np.float32)[1:], "wide", b[i], b[i].view(np.float32)[1:])
