import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bvh_amd
from bvh_amd import synth
for name, n in (("soup_1m", 1_000_000), ("soup_4m", 4_000_000)):
    tris = torch.from_numpy(synth.soup(n)).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
        torch.cuda.synchronize(); print(name, "High build ms", (time.perf_counter() - t) * 1e3, bvh_amd.last_optimize_profile(), flush=True)
