import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bvh_amd, oracle
from bvh_amd import synth
lib = bvh_amd._lib.load()
cpu = oracle.gpu_checker()
thr = 16
tris = synth.sponza_proxy(262_144)
d_tris = torch.from_numpy(tris).cuda()
bb, cc = bvh_amd.tri_bounds(d_tris)
gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
ref = cpu.build(bb.cpu().numpy(), cc.cpu().numpy(), builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=thr)
print("tree equal", gpu.serialize() == ref.serialize())
prims = bvh_amd.precompute_tris(d_tris, gpu.device_prim_ids())
oprims = cpu.precompute_tris(tris, ref.prim_ids())
lo, hi = synth.scene_bounds(tris)
rays = synth.rays_closest(2_097_152, lo, hi, seed=91)
want = ref.intersect_tri(oprims, rays, False, True, threads=thr)
for coop in (1, 2, 0):
    lib.bvh_amd_tuning(12, 12, coop, -1)
    got = bvh_amd.hits_to_numpy(bvh_amd.intersect(gpu, prims, torch.from_numpy(rays).cuda(), any_hit=False, robust=True, sort_rays=True))
    lib.bvh_amd_tuning(-1, -1, -1, -1)
    bad = np.flatnonzero((got["prim"] != want["prim"]) | (got["t"] != want["t"]))
    print("coop", coop, lib.bvh_amd_last_kernel_name().decode(), "differs from the reference on", len(bad), "rays")
    for i in bad[:3]: print("   ray", i, rays[i], "gpu", got[i], "ref", want[i])
