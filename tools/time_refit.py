"""Developer probe: device time of Bvh::refit (bvhXX_refit) and Bvh::extract_bvh on a resident tree.   python tools/time_refit.py [n_tris]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bvh_amd
from bvh_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tris = torch.from_numpy(synth.soup(n)).cuda()
bb, cc = bvh_amd.tri_bounds(tris)
bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
for name, fn in (("refit", lambda: bvh.refit()), ("extract_bvh(root's first child)", lambda: bvh.extract_bvh(1))):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: n_tris={n} nodes={bvh.node_count} ms={[round(t, 3) for t in ts]}", flush=True)
