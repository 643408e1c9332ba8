"""Developer probe for the kernels of BASELINE configs[2], [3], [4] (the bench profiles configs[1]-class closest-hit on the soup):
builds each scene once and launches its traversal kernel a few times, so that rocprofv3 passes (--kernel-trace --stats, --pmc ...)
of this ONE command yield per-kernel time and counters for
  anyhit    : 262,144-triangle Sponza proxy, serial Low tree, 10M any-hit shadow rays, fast slab test   (trace_kernel<float, true, false, 0, ...>)
  spheres64 : 1M double-precision spheres, (pool, High) tree, 1M robust closest-hit rays               (trace_kernel<double, false, true, 1, ...>)
  shard10m  : 10M-triangle procedural mesh, (pool, <quality>) tree, 12.5M robust closest-hit rays = one GPU's shard of the 100M
    python tools/config_probe.py [anyhit,spheres64,shard10m] [reps] [quality of the 10M tree: 1 | 2]
Prints one JSON line per config (ms per launch by HIP events, P, T, bytes/ray)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth


def timed(fn, reps):
    for _ in range(10):                                       # the library's launch-plan search settles on finished batches
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, bvh, prims, rays, any_hit, robust, leaf, reps):
    n = rays.shape[0]
    out = torch.empty((n, 4), dtype=rays.dtype, device="cuda")
    _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, counters=True)
    c = cnt.cpu().numpy()
    ms = timed(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, out=out), reps)
    lib = bvh_amd._lib.load()
    P, T = c[0] / n, c[1] / n
    f32 = rays.dtype == torch.float32
    b_ray = (32 + 56 * P + (48 if leaf == "tri" else 16) * T + 16) if f32 else (64 + 112 * P + (96 if leaf == "tri" else 32) * T + 32)
    print(json.dumps({"config": name, "kernel": lib.bvh_amd_last_kernel_name().decode(), "reordered": bool(lib.bvh_amd_last_launch_reordered()),
                      "rays": n, "nodes": bvh.node_count, "ms_per_call": round(ms, 4), "mrays_s": round(n / ms / 1e3, 1), "P": round(float(P), 3), "T": round(float(T), 3),
                      "bytes_per_ray": round(float(b_ray), 1), "algorithmic_gbs": round(n * b_ray / (ms * 1e-3) / 1e9, 1)}), flush=True)


def main():
    which = (sys.argv[1] if len(sys.argv) > 1 else "anyhit,spheres64,shard10m").split(",")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    q10 = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    if "anyhit" in which:
        t_h = synth.sponza_proxy(262144)
        tris = torch.from_numpy(t_h).cuda()
        bb, cc = bvh_amd.tri_bounds(tris)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
        prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
        lo, hi = synth.scene_bounds(t_h)
        rays = torch.from_numpy(synth.rays_shadow(10_000_000, lo, hi)).cuda()
        report("configs[2] Sponza proxy 262k, 10M any-hit shadow rays, fast", bvh, prims, rays, True, False, "tri", reps)
        del rays, prims, bvh, tris
    if "spheres64" in which:
        s_h = synth.spheres(1_000_000)
        sph = torch.from_numpy(s_h).cuda()
        bb, cc = bvh_amd.sphere_bounds(sph)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
        prims = bvh_amd.gather(sph, bvh.device_prim_ids())
        lo, hi = synth.scene_bounds(s_h)
        rays = torch.from_numpy(synth.rays_closest(1_000_000, lo, hi, dtype=np.float64)).cuda()
        report("configs[4] 1M f64 spheres (pool, High), 1M robust closest-hit rays", bvh, prims, rays, False, True, "sphere", reps)
        del rays, prims, bvh, sph
    if "shard10m" in which:
        t_h = synth.procedural_10m()
        tris = torch.from_numpy(t_h).cuda()
        bb, cc = bvh_amd.tri_bounds(tris)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q10)), thread_pool=bvh_amd.ThreadPool())
        prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
        lo, hi = synth.scene_bounds(t_h)
        rays = torch.from_numpy(synth.rays_closest(12_500_000, lo, hi, seed=1234 + 3)).cuda()
        report(f"configs[3] 10M-triangle mesh (pool, quality {q10}), 12.5M robust closest-hit rays (one shard of 100M)", bvh, prims, rays, False, True, "tri", reps)


if __name__ == "__main__":
    main()
