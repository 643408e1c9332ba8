"""Times DefaultBuilder High builds of the 1M soup (serial and thread-pool) with the exact heap replay forced and not forced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bvh_amd
from bvh_amd import synth

tris = torch.from_numpy(synth.soup(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000)).cuda()
bb, cc = bvh_amd.tri_bounds(tris)
for mode in ("exact", ""):
    os.environ["BVH_AMD_REINSERT"] = mode
    for pool in (None, bvh_amd.ThreadPool()):
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=pool)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(f"reinsert={mode or 'auto':5s} {'pool  ' if pool else 'serial'} High build {min(ts):8.1f} ms   fast/exact iterations so far {bvh_amd.reinsertion_stats()}", flush=True)
