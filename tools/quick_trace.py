"""Developer probe (NOT the bench): Mrays/s of the traversal kernel, unsorted vs coherence-sorted rays."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
    q = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
    hits, cnt = bvh_amd.intersect(bvh, prims, rays, False, True, counters=True)
    c = cnt.cpu().numpy()
    P, T = c[0] / nr, c[1] / nr
    bray = 32 + 56 * P + 48 * T + 16
    ref = None
    for sort in (False, True):
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, rays, False, True, out=out, sort_rays=sort)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        ev0.record()
        for _ in range(reps):
            bvh_amd.intersect(bvh, prims, rays, False, True, out=out, sort_rays=sort)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        mr = nr / ms / 1e3
        same = True if ref is None else bool((out.view(torch.int32) == ref.view(torch.int32)).all())
        ref = out.clone() if ref is None else ref
        print(f"{scene} n={n} q={q} rays={nr} sort={sort}: {ms:.3f} ms  {mr:.1f} Mrays/s  P={P:.2f} T={T:.2f} "
              f"B_ray={bray:.0f} frac={mr*1e6*bray/8e12:.3f} same_as_unsorted={same}", flush=True)

if __name__ == "__main__":
    main()
