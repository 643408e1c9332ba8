"""Developer probe (NOT the bench): Mrays/s of the traversal kernel on an oracle-built BVH."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, bvh_amd
from bvh_amd import synth

def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
    orc = oracle.load_oracle()
    tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)
    bb, cc = orc.prep_tris(tris)
    t0 = time.time()
    ob = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH)
    print(f"oracle build {time.time()-t0:.2f}s nodes={ob.node_count}", flush=True)
    ids = ob.prim_ids()
    bvh = bvh_amd.Bvh.from_nodes(ob.nodes(), ids)
    prims = bvh_amd.precompute_tris(tris, ids.astype(np.int32))
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
    for robust in (True, False):
        hits, cnt = bvh_amd.intersect(bvh, prims, rays, False, robust, counters=True)
        c = cnt.cpu().numpy()
        P, T = c[0] / nr, c[1] / nr
        bray = 32 + 56 * P + 48 * T + 16
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, rays, False, robust, out=out)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        ev0.record()
        for _ in range(reps):
            bvh_amd.intersect(bvh, prims, rays, False, robust, out=out)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        mr = nr / ms / 1e3
        print(f"{scene} n={n} rays={nr} robust={robust}: {ms:.3f} ms  {mr:.1f} Mrays/s  P={P:.2f} T={T:.2f} "
              f"B_ray={bray:.0f}  algGB/s={mr*1e6*bray/1e9:.0f} ({mr*1e6*bray/8e12*100:.1f}% of 8TB/s)", flush=True)

if __name__ == "__main__":
    main()
