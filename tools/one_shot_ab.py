"""A/B of the one-shot grid for small batches (VERDICT r4 Next 4): configs[1] (Sponza proxy 262k, serial Low, closest-hit robust) and
configs[4] (1M f64 spheres, parallel High), batches of 2^16 .. 2^22 rays, per-lane and cooperative fetch, persistent grid vs one-shot.
    python tools/one_shot_ab.py > profiles/r05_one_shot_ab.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, reps=20):
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
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return t[len(t) // 2]


def main():
    import torch
    import bvh_amd
    from bvh_amd import synth
    lib = bvh_amd._lib.load()
    scenes = []
    tris = synth.sponza_proxy(262144)
    d = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
    prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    scenes.append(("configs[1] sponza proxy 262k f32 tri", bvh, prims, "tri", np.float32, lo, hi))
    sph = synth.spheres(1_000_000)
    ds = torch.from_numpy(sph).cuda()
    bb, cc = bvh_amd.sphere_bounds(ds)
    bvh2 = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    sprims = bvh_amd.gather(ds, bvh2.device_prim_ids())
    lo2, hi2 = (sph[:, :3] - sph[:, 3:4]).min(axis=0), (sph[:, :3] + sph[:, 3:4]).max(axis=0)
    scenes.append(("configs[4] 1M spheres f64", bvh2, sprims, "sphere", np.float64, lo2, hi2))
    for name, b, p, leaf, dt, l, h in scenes:
        print(f"## {name}")
        for any_hit in (False, True):
            for logn in (16, 18, 20, 21, 22):
                n = 1 << logn
                rays = torch.from_numpy(synth.rays_closest(n, l, h, dtype=dt) if not any_hit else synth.rays_shadow(n, l, h, dtype=dt)).cuda()
                out = torch.empty((n, 4), dtype=torch.float32 if dt == np.float32 else torch.float64, device="cuda")
                row = []
                ref = None
                for coop in (0, 1):
                    for one in (0, 1):
                        lib.bvh_amd_tuning(-1, -1, coop, -1)
                        lib.bvh_amd_experiment(b"one_shot", one)
                        ms = time_ms(lambda: bvh_amd.intersect(b, p, rays, any_hit=any_hit, robust=True, leaf=leaf, out=out, sort_rays=False))
                        got = out.clone()
                        if ref is None:
                            ref = got
                        assert torch.equal(ref.view(torch.int32), got.view(torch.int32)), "results differ"
                        row.append(ms)
                lib.bvh_amd_tuning(-1, -1, -1, -1)
                lib.bvh_amd_experiment(b"one_shot", -1)
                dflt = time_ms(lambda: bvh_amd.intersect(b, p, rays, any_hit=any_hit, robust=True, leaf=leaf, out=out))
                print(f"  {'any-hit' if any_hit else 'closest'} 2^{logn}: per-lane persistent {row[0]:.4f}  one-shot {row[1]:.4f} | coop persistent {row[2]:.4f}  one-shot {row[3]:.4f} ms "
                      f"| library default {dflt:.4f} ms = {n / dflt / 1e6:.2f} Grays/s; best {n / min(row) / 1e6:.2f} Grays/s")


if __name__ == "__main__":
    main()
