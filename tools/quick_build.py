"""Developer probe: device build time per DefaultBuilder mode."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
tris = torch.from_numpy({"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)).cuda()
bb, cc = bvh_amd.tri_bounds(tris)
for name, q, pool in [("serial Low (binned)", 0, False), ("serial Medium (sweep)", 1, False), ("serial High (sweep+reins)", 2, False),
                      ("parallel Low (minitree)", 0, True), ("parallel Medium", 1, True), ("parallel High", 2, True)]:
    cfg = bvh_amd.Config(quality=bvh_amd.Quality(q))
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if pool else None)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{scene} n={n} {name:28s} nodes={b.node_count:8d}  best {min(ts)*1e3:9.2f} ms  ({n/min(ts)/1e6:7.2f} Mtris/s)  all={[round(t*1e3,1) for t in ts]}  reinsertion fast/exact iterations so far={bvh_amd.reinsertion_stats()}", flush=True)
