"""configs[1] exactly as stated (Sponza proxy 262k, serial Low, 1M closest-hit rays): refill / leaf thresholds, grid cap and stagger swept
for the per-lane kernel. python tools/small_batch_sweep.py > profiles/r05_small_batch_sweep.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med_ms(fn, reps=30):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    import torch
    import bvh_amd
    from bvh_amd import synth
    lib = bvh_amd._lib.load()
    tris = synth.sponza_proxy(262144)
    d = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
    prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    for n in (1_000_000, 2_000_000):
        rays = torch.from_numpy(synth.rays_closest(n, lo, hi)).cuda()
        hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        fn = lambda: bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)   # noqa: E731
        print(f"## {n} rays: library default {med_ms(fn):.4f} ms")
        cus = 256
        for coop in (0, 1):
            for blocks in (4, 5, 6, 7, 8):
                row = []
                for refill, leaf in ((12, 12), (20, 12), (28, 12), (36, 12), (44, 12), (36, 8), (36, 16), (28, 8), (20, 8)):
                    lib.bvh_amd_tuning(refill, leaf, coop, -1)
                    lib.bvh_amd_experiment(b"grid_blocks", blocks * cus)
                    best = min((med_ms(fn), st) for st in (-1, 0, 20000, 35000, 80000) if not lib.bvh_amd_experiment(b"stagger", st))
                    row.append(f"{refill}/{leaf}: {best[0]:.4f}@{best[1]}")
                print(f"   coop {coop} blocks/CU {blocks} | refill/leaf: ms@best stagger | " + "  ".join(row))
        lib.bvh_amd_tuning(-1, -1, -1, -1)
        lib.bvh_amd_experiment(b"reset", 0)


if __name__ == "__main__":
    main()
