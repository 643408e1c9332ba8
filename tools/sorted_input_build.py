"""Developer probe: how much of a Low build is the random gather of primitive data by id? The same soup built as generated (random order in
memory) and with its triangles physically sorted by the Morton code of their centroids (gathers by id become nearly sequential).
    python tools/sorted_input_build.py <n_tris> [quality] [pool]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

n = int(sys.argv[1]); q = int(sys.argv[2]) if len(sys.argv) > 2 else 0; pool = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tris = synth.soup(n)
c = tris.reshape(n, 3, 3).mean(axis=1)
lo, hi = c.min(0), c.max(0)
g = np.minimum(1023, ((c - lo) / (hi - lo) * 1024).astype(np.int64))


def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v


code = spread(g[:, 0]) | (spread(g[:, 1]) << 1) | (spread(g[:, 2]) << 2)
order = np.argsort(code, kind="stable")
cfg = bvh_amd.Config(quality=bvh_amd.Quality(q))
for name, t in (("as generated", tris), ("Morton-sorted in memory", np.ascontiguousarray(tris.reshape(n, -1)[order]).reshape(tris.shape))):
    d = torch.from_numpy(t).cuda()
    ts = []
    for r in range(4):
        b = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bb, cc = bvh_amd.tri_bounds(d)
        b = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool() if pool else None)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{n} tris q={q} pool={pool} {name:26s}: {sorted(ts[1:])[1]:.2f} ms (nodes {b.node_count})", flush=True)
