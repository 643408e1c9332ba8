"""A/B of the chord-class bits of the ray reordering key (long rays first: ray_keys_kernel, round 5): pass and kernel time of the bench
launch for class_bits x class_scale, per-ray results checked equal.
    python tools/key_class_ab.py [--workload soup_1m] [--rays 16777216] > profiles/r05_key_class_ab.txt"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="soup_1m")
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--quality", default="high")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bvh_amd
    import bench
    from bvh_amd import synth
    lib = bvh_amd._lib.load()
    gen, n_tris, desc, _ = bench.WORKLOADS[args.workload]
    tris = getattr(synth, gen)(n_tris)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality[args.quality.capitalize()]), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(args.rays, lo, hi, seed=1234)).cuda()
    hits = torch.empty((args.rays, 4), dtype=torch.float32, device="cuda")
    for _ in range(12):
        bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)
        torch.cuda.synchronize()
    ref = hits.clone()
    plan = (C.c_int * 4)()
    lib.bvh_amd_last_launch_plan(plan)
    print(f"# {args.workload}: {desc}; {args.rays} rays, plan reordered={plan[0]} coop={plan[1]} refill={plan[2]} leaf={plan[3]}")
    combos = [(0, 0)] + [(1, s) for s in (200, 250, 300, 400, 500)] + [(2, s) for s in (400, 600, 800)] + [(3, s) for s in (800, 1200)] + [(0, 0)]
    for bits, scale in combos:
        lib.bvh_amd_experiment(b"key_class_bits", bits)
        lib.bvh_amd_experiment(b"key_class_scale", scale if scale else -1)
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)
        torch.cuda.synchronize()
        lib.bvh_amd_kernel_timing(1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
        ev[0].record()
        for i in range(args.reps):
            bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)
            ev[i + 1].record()
        torch.cuda.synchronize()
        kt, rt = (C.c_float * 64)(), (C.c_float * 64)()
        got = C.c_size_t(0)
        lib.bvh_amd_kernel_times(kt, args.reps, C.byref(got))
        k_ms = float(np.mean(kt[:got.value]))
        lib.bvh_amd_reorder_times(rt, args.reps, C.byref(got))
        r_ms = float(np.mean(rt[:got.value]))
        lib.bvh_amd_kernel_timing(0)
        p_ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps)]))
        same = bool(torch.equal(ref.view(torch.int32), hits.view(torch.int32)))
        print(f"class_bits {bits} scale {scale:4d}: pass {p_ms:.4f} ms = keys + sort {r_ms:.4f} + kernel {k_ms:.4f}   {args.rays / p_ms / 1e3:.1f} Mrays/s   hits equal: {same}")
    lib.bvh_amd_experiment(b"reset", 0)


if __name__ == "__main__":
    main()
