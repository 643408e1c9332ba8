"""Developer probe (NOT the bench): ms / Mrays/s of the closest-hit robust traversal kernel on one scene, a few launches.
    python tools/trace_probe.py [soup|terrain|sponza] [n_tris] [n_rays] [quality] [reps]
Used under rocprofv3 --pmc (few launches, one kernel) and for A/B runs of kernel variants selected by environment switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 23
    q = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
    out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
    for any_hit, robust in ((False, True), (True, False)):
        if any_hit and os.environ.get("PROBE_CLOSEST_ONLY"):
            break
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=out)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=out)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        sha = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"PROBE {scene} n={n} q={q} rays={nr} any={int(any_hit)} robust={int(robust)} "
              f"kernel={bvh_amd._lib.load().bvh_amd_last_kernel_name().decode()} {ms:.3f} ms {nr / ms / 1e3:.1f} Mrays/s hits_sha1={sha}", flush=True)

if __name__ == "__main__":
    main()
