"""Developer probe: ONE DefaultBuilder mode, warm-up build + `reps` timed builds (host wall clock around a stream sync).
    python tools/build_profile.py <scene> <n_tris> <quality 0|1|2> <pool 0|1> [reps]
Run under `rocprofv3 --kernel-trace --stats` to get the per-kernel table of that one mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

scene, n, q, pool = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
tris = torch.from_numpy({"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)).cuda()
cfg = bvh_amd.Config(quality=bvh_amd.Quality(q))
ts = []
b = None
for r in range(reps + 1):
    b = None                                                  # (destroying the previous BVH is not part of a build)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bb, cc = bvh_amd.tri_bounds(tris)
    b = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if pool else None)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"BUILD {scene} n={n} q={q} pool={pool} nodes={b.node_count} warm={ts[0]:.2f} ms timed={[round(t, 2) for t in ts[1:]]} median={sorted(ts[1:])[len(ts[1:]) // 2]:.2f} ms "
      f"reinsertion fast/exact={bvh_amd.reinsertion_stats()}", flush=True)
