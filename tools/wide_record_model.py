"""Paper model for the 128-byte "sibling pair" record layout (VERDICT r3 "Next 2a"), on the CPU: how many of a ray's record fetches
could be merged with an earlier one if the two children's records sat in one 128-byte line and were fetched together whenever BOTH
children are hit and both are inner nodes (the far child's record is then needed later for certain: the reference pops it without
re-testing, bvh.h:125-157). Walks a sample of the bench's rays through the reference-built tree (oracle, test infrastructure) in
plain numpy floats: statistics only, not a parity tool.
    python tools/wide_record_model.py [n_tris] [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from bvh_amd import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nr = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    cpu = oracle.load_ref() or oracle.load_oracle()
    tris = synth.soup(n)
    bb, cc = cpu.prep_tris(tris)
    ref = cpu.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL, quality=oracle.QUALITY_HIGH, threads=8)
    nodes = ref.nodes()
    bounds = nodes["bounds"].astype(np.float64)
    index = nodes["index"].astype(np.int64)
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(nr, lo, hi).astype(np.float64)
    prims = cpu.precompute_tris(tris, ref.prim_ids())
    hits = ref.intersect_tri(prims, rays.astype(np.float32), False, True, threads=8)
    P = both = both_inner = single = depth_sum = 0
    merged_deep = 0
    first0 = int(index[0] >> 4)
    for r in range(nr):
        org, d, tmin = rays[r, 0:3], rays[r, 3:6], rays[r, 6]
        tmax = float(hits["t"][r]) if hits["prim"][r] != 0xFFFFFFFF else rays[r, 7]     # the final tmax: a LOWER bound of the visits (culling is at least this good late in the walk)
        inv = 1.0 / d
        stack = [(first0, 0)]
        while stack:
            f, lvl = stack.pop()
            P += 1
            res = []
            for c in (f, f + 1):
                b = bounds[c]
                t0 = (b[0::2] - org) * inv
                t1 = (b[1::2] - org) * inv
                tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
                a, z = max(tn.max(), tmin), min(tf.min(), tmax)
                res.append((a <= z, a, c))
            h = [x for x in res if x[0]]
            inner = [x for x in h if (index[x[2]] & 15) == 0]
            if len(h) == 2:
                both += 1
                if len(inner) == 2:
                    both_inner += 1
                    if lvl >= 12:
                        merged_deep += 1
            elif len(h) == 1:
                single += 1
            for x in inner:
                stack.append((int(index[x[2]] >> 4), lvl + 1))
    print(f"{n} triangles, {nr} rays (culled with each ray's FINAL tmax: a lower bound of the reference's visits):")
    print(f"  pair records fetched per ray              {P / nr:7.2f}")
    print(f"  ... where both children are hit           {both / nr:7.2f}  ({100.0 * both / P:.0f} %)")
    print(f"  ... and both are inner nodes              {both_inner / nr:7.2f}  ({100.0 * both_inner / P:.0f} %)  = fetches a 128-byte sibling line would merge with an earlier one")
    print(f"  ... of those at tree level >= 12          {merged_deep / nr:7.2f}")


if __name__ == "__main__":
    main()
