"""Developer experiment: does the ORDER of the node pairs in memory matter to the traversal kernel? The High tree of the 1M soup (node order = the
builder's numbering disturbed by the reinsertion optimizer's moves) against the same tree renumbered (a) in depth-first pre-order of the inner
nodes, near... children pairs allocated as the walk meets them, (b) breadth-first. Per-ray results are identical (checked).
    python tools/relayout_probe.py [n_tris] [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth
from collections import deque


def renumber(nodes, order):
    index = nodes["index"]
    n = len(nodes)
    new_of = np.full(n, -1, dtype=np.int64)
    new_of[0] = 0
    nxt = 1
    todo = deque([0])
    first_of = (index >> 4).astype(np.int64)
    count = (index & 15).astype(np.int64)
    while todo:
        i = todo.pop() if order == "dfs" else todo.popleft()
        if count[i]:
            continue
        c = first_of[i]
        new_of[c] = nxt; new_of[c + 1] = nxt + 1
        nxt += 2
        if order == "dfs":
            todo.append(c + 1); todo.append(c)                # left subtree first
        else:
            todo.append(c); todo.append(c + 1)
    assert nxt == n
    out = np.empty_like(nodes)
    out[new_of] = nodes
    inner = (out["index"] & 15) == 0
    out["index"][inner] = (new_of[(out["index"][inner] >> 4).astype(np.int64)].astype(out["index"].dtype) << 4)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
    tris = synth.soup(n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
    nodes, ids = bvh.nodes, bvh.prim_ids
    out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
    base = None
    for name in ("as built", "dfs", "bfs"):
        b = bvh if name == "as built" else bvh_amd.Bvh.from_nodes(renumber(nodes, name), ids)
        for sort_rays in (False, True):
            for _ in range(2):
                bvh_amd.intersect(b, prims, rays, False, True, out=out, sort_rays=sort_rays)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                bvh_amd.intersect(b, prims, rays, False, True, out=out, sort_rays=sort_rays)
            ev1.record(); torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 5
            if base is None: base = out.clone()
            else: assert torch.equal(out.view(torch.int32), base.view(torch.int32)), name
            print(f"LAYOUT soup n={n} rays={nr} nodes {name:9s} sort_rays={int(sort_rays)} {ms:8.3f} ms {nr / ms / 1e3:8.1f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
