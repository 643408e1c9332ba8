"""A/B of the staggered drain (trace_body.inc) and of the chord classes of the reordering key: kernel / pass time against `stagger`
(tickets per eighth of the grid) on the bench launch and on small batches.
    python tools/stagger_ab.py > profiles/r05_stagger_ab.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(lib, fn, reps):
    import torch
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.bvh_amd_kernel_timing(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    kt = (C.c_float * 64)()
    got = C.c_size_t(0)
    lib.bvh_amd_kernel_times(kt, reps, C.byref(got))
    lib.bvh_amd_kernel_timing(0)
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])), float(np.median(kt[:got.value]))


def main():
    import torch
    import bvh_amd
    from bvh_amd import synth
    lib = bvh_amd._lib.load()
    cases = [("soup_1m High pool", "soup", 1_000_000, "High", True, [1 << 24, 1 << 22, 1 << 20]),
             ("soup_10m Medium pool", "soup", 10_000_000, "Medium", True, [12_500_000]),
             ("sponza_262k Low serial", "sponza_proxy", 262_144, "Low", False, [1 << 20, 1 << 22]),
             ("terrain_1m Low serial", "terrain", 1_000_000, "Low", False, [1 << 22])]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "spheres":
        cases = [("spheres_1m f64 High pool (configs[4])", "spheres", 1_000_000, "High", True, [1 << 20, 1 << 22])]
    for name, gen, n, q, pool, batches in cases:
        tris = getattr(synth, gen)(n)
        d = torch.from_numpy(tris).cuda()
        leaf = "sphere" if gen == "spheres" else "tri"
        bb, cc = bvh_amd.sphere_bounds(d) if leaf == "sphere" else bvh_amd.tri_bounds(d)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality[q]), thread_pool=bvh_amd.ThreadPool() if pool else None)
        prims = bvh_amd.gather(d, bvh.device_prim_ids()) if leaf == "sphere" else bvh_amd.precompute_tris(d, bvh.device_prim_ids())
        if leaf == "sphere":
            lo, hi = (tris[:, :3] - tris[:, 3:4]).min(axis=0), (tris[:, :3] + tris[:, 3:4]).max(axis=0)
        else:
            lo, hi = synth.scene_bounds(tris)
        for nr in batches:
            rays = torch.from_numpy(synth.rays_closest(nr, lo, hi, seed=1234, dtype=tris.dtype)).cuda()
            hits = torch.empty((nr, 4), dtype=torch.float64 if tris.dtype == np.float64 else torch.float32, device="cuda")
            fn = lambda: bvh_amd.intersect(bvh, prims, rays, robust=True, leaf=leaf, out=hits)   # noqa: E731
            for _ in range(12):
                fn()
                torch.cuda.synchronize()
            ref = hits.clone()
            plan = (C.c_int * 4)()
            lib.bvh_amd_last_launch_plan(plan)
            print(f"## {name}, {nr} closest-hit rays; plan reordered={plan[0]} coop={plan[1]} refill={plan[2]} leaf={plan[3]}")
            base = None
            for cls in ((0, 0), (1, 500), (1, 800)) if plan[0] else ((0, 0),):
                lib.bvh_amd_experiment(b"key_class_bits", cls[0])
                lib.bvh_amd_experiment(b"key_class_scale", cls[1] if cls[1] else -1)
                row = []
                for st in (0, nr // 400, nr // 200, nr // 100, nr // 64, nr // 48, nr // 32, nr // 24, nr // 16):
                    lib.bvh_amd_experiment(b"stagger", st)
                    p, k = measure(lib, fn, 10)
                    assert torch.equal(ref.view(torch.uint8), hits.view(torch.uint8))
                    if base is None:
                        base = p
                    row.append(f"{st}: {p:.4f}/{k:.4f}")
                print(f"   class_bits {cls[0]} scale {cls[1]:4d} | stagger: pass/kernel ms | " + "  ".join(row))
            lib.bvh_amd_experiment(b"reset", 0)


if __name__ == "__main__":
    main()
