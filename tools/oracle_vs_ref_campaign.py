"""Developer tool (CPU only, needs /root/reference): the restatement (oracle/bvh_oracle.cpp) against the compiled reference
(oracle/_ref) on the adversarial generators of tests/test_gpu_fuzz.py: every builder mode, leaf limits, thread counts, then
closest / any-hit x fast / robust traversal with counters. What pins the oracle beyond the committed golden vectors."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle
import test_gpu_fuzz as F

orc, ref = oracle.load_oracle(), oracle.load_ref()
assert ref is not None, "needs oracle/_ref (built where /root/reference exists)"
lo_seed, hi_seed = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
t0 = time.time(); ran = 0; bad = []
KINDS = ("lattice", "dups", "flat", "points", "scales", "uniform")
for seed in range(lo_seed, hi_seed):
    for kind in KINDS:
        if time.time() - t0 > budget: break
        rng = np.random.default_rng(1000 * seed + KINDS.index(kind))
        dtype = np.float32 if seed % 3 else np.float64
        n = int(rng.choice([1, 2, 5, 17, 64, 65, 200, 1500, 4000]))
        tris = F._scene3(rng, n, kind, dtype)
        bb, cc = ref.prep_tris(tris)
        obb, occ = orc.prep_tris(tris)
        ok = bb.tobytes() == obb.tobytes() and cc.tobytes() == occ.tobytes()
        lim = [(1, 8), (1, 1), (2, 4), (3, 15)][seed % 4]
        thr = [1024, 64][seed % 2]
        a = b = None
        for builder, quality in ((2, 0), (3, 0), (0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)):
            a = ref.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=thr, threads=1 + seed % 5)
            b = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=thr)
            ok = ok and a.serialize() == b.serialize()
        prims = ref.precompute_tris(tris, a.prim_ids())
        ok = ok and prims.tobytes() == orc.precompute_tris(tris, b.prim_ids()).tobytes()
        lo, hi = tris.reshape(-1, 3).min(axis=0).astype(np.float64), tris.reshape(-1, 3).max(axis=0).astype(np.float64)
        rays = F._rays3(rng, 3000, lo, hi, dtype)
        for any_hit in (False, True):
            for robust in (False, True):
                wa, ca = a.intersect_tri(prims, rays, any_hit, robust, threads=2, counters=True)
                wb, cb = b.intersect_tri(prims, rays, any_hit, robust, counters=True)
                ok = ok and wa.tobytes() == wb.tobytes() and (ca == cb).all()
        ran += 1
        if not ok:
            bad.append((seed, kind, n)); print("MISMATCH", bad[-1], flush=True)
    if time.time() - t0 > budget: break
print(f"oracle vs reference: {ran} scenes x (8 builds + 4 traversals) in {time.time() - t0:.1f} s, mismatches: {len(bad)}", flush=True)
