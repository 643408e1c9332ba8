"""Runs BASELINE.json's five configs on one MI355X and prints one JSON line per config (profiles/r0N_baseline_configs.jsonl).
Config 4's 100M rays are sharded 8 ways by bench.py --gpus 8; here one GPU traces one shard (12.5M rays)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth


def timed(fn, reps=5, warm=10):
    for _ in range(warm):                                     # (the library settles its launch plan on a tree's first large batches, read
        fn()                                                  #  back one finished batch at a time: csrc/traverse.hip launch_traverse)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def build_timed(bb, cc, cfg, pool, reps=3):
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=pool)
    ts = []
    for _ in range(reps):
        bvh = None                                            # (destroying the previous BVH is not part of a build)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=pool)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return bvh, min(ts) * 1e3


def trace_report(name, bvh, prims, rays, any_hit, robust, leaf="tri", extra=None):
    n = rays.shape[0]
    out = torch.empty((n, 4), dtype=rays.dtype, device="cuda")
    _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, counters=True)
    c = cnt.cpu().numpy()
    ms = timed(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, out=out))
    h = out.view(torch.int32 if rays.dtype == torch.float32 else torch.int64)[:, 0]
    hits = int((h.to(torch.int64) & 0xFFFFFFFF != 0xFFFFFFFF).sum())
    P, T = c[0] / n, c[1] / n
    if rays.dtype == torch.float32:
        b_ray = 32 + 56 * P + (48 if leaf == "tri" else 16) * T + 16
    else:
        b_ray = 64 + 112 * P + (96 if leaf == "tri" else 32) * T + 32
    rec = {"config": name, "rays": n, "ms": round(ms, 3), "mrays_s": round(n / ms / 1e3, 1), "hits": hits,
           "P": round(float(P), 2), "T": round(float(T), 2), "bytes_per_ray": round(float(b_ray), 1),
           "roofline_frac_of_8TBs": round(n / (ms * 1e-3) * b_ray / 8e12, 3)}
    rec.update(extra or {})
    print(json.dumps(rec), flush=True)


def main():
    dev = torch.cuda.get_device_name(0)
    # 1. simple_example
    tris = np.array([[1, -1, 1, 1, 1, 1, -1, 1, 1], [1, -1, 1, -1, -1, 1, -1, 1, 1]], dtype=np.float32)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    h = bvh_amd.hits_to_numpy(bvh_amd.intersect(bvh, prims, np.array([[0, 0, 0, 0, 0, 1, 0, 100]], dtype=np.float32)))
    print(json.dumps({"config": "1 simple_example (2 triangles, 1 ray)", "device": dev, "primitive": int(h["prim"][0]), "distance": float(h["t"][0]),
                      "u": str(h["u"][0]), "v": float(h["v"][0])}), flush=True)
    # 2 + 3. Sponza proxy
    t_h = synth.sponza_proxy(262144)
    tris = torch.from_numpy(t_h).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh, ms = build_timed(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low), None)
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(t_h)
    rays = torch.from_numpy(synth.rays_closest(1_000_000, lo, hi)).cuda()
    trace_report("2 sponza-proxy 262k, binned-SAH (serial Low) + 1M closest-hit rays", bvh, prims, rays, False, True,
                 extra={"build_ms": round(ms, 3), "build_mtris_s": round(262144 / ms / 1e3, 1), "nodes": bvh.node_count})
    srays = torch.from_numpy(synth.rays_shadow(10_000_000, lo, hi)).cuda()
    trace_report("3 sponza-proxy any-hit shadow rays (SATO order), 10M rays, fast", bvh, prims, srays, True, False)
    trace_report("3 sponza-proxy any-hit shadow rays (SATO order), 10M rays, robust", bvh, prims, srays, True, True)
    del rays, srays
    # 5. double + spheres
    sph_h = synth.spheres(1_000_000)
    sph = torch.from_numpy(sph_h).cuda()
    bb, cc = bvh_amd.sphere_bounds(sph)
    bvh, ms = build_timed(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), bvh_amd.ThreadPool(), reps=1)
    sprims = bvh_amd.gather(sph, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(sph_h)
    rays = torch.from_numpy(synth.rays_closest(1_000_000, lo, hi, dtype=np.float64)).cuda()
    trace_report("5 double-precision 3D BVH + 1M spheres (parallel High), 1M rays, robust closest", bvh, sprims, rays, False, True, leaf="sphere",
                 extra={"build_ms": round(ms, 3), "build_mtris_s": round(1e6 / ms / 1e3, 2), "nodes": bvh.node_count})
    del rays, sph, sprims
    # 4. 10M procedural, mini-tree + reinsertion, one GPU's shard of the 100M rays
    t_h = synth.procedural_10m()
    tris = torch.from_numpy(t_h).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    for qname, q, reps in (("Low", bvh_amd.Quality.Low, 2), ("Medium", bvh_amd.Quality.Medium, 2), ("High", bvh_amd.Quality.High, 1)):
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=q), thread_pool=bvh_amd.ThreadPool())      # warm-up build
        ts = []
        for _ in range(reps):
            bvh = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=q), thread_pool=bvh_amd.ThreadPool())
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ms = min(ts)
        print(json.dumps({"config": f"4 10M-triangle procedural mesh, mini-tree build, Quality::{qname}", "build_ms": round(ms, 1),
                          "build_mtris_s": round(1e7 / ms / 1e3, 2), "nodes": bvh.node_count}), flush=True)
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(t_h)
    rays = torch.from_numpy(synth.rays_closest(12_500_000, lo, hi)).cuda()
    # what a single-shot caller sees: the FIRST call through the fresh tree (after a 4096-ray call: code load, depth of the tree)
    out = torch.empty((12_500_000, 4), dtype=torch.float32, device="cuda")
    bvh_amd.intersect(bvh, prims, rays[:4096], False, True, out=out[:4096])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bvh_amd.intersect(bvh, prims, rays, False, True, out=out)
    torch.cuda.synchronize(); first_ms = (time.perf_counter() - t0) * 1e3
    trace_report("4 10M-triangle mesh (parallel High tree), 12.5M closest-hit rays = one GPU's shard of 100M", bvh, prims, rays, False, True,
                 extra={"first_call_ms_fresh_tree": round(first_ms, 3)})


if __name__ == "__main__":
    main()
