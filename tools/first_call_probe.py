"""Wall time of each of the first large batches through a fresh tree (host clock around call + synchronisation), with the plan each
one was traced with: what the first call pays over the settled pass, and where.
    python tools/first_call_probe.py [n_tris] [n_rays_log2]            BVH_AMD_CALIBRATE=0 keeps the predictor's plan throughout;
    PROBE_HEAT=n: n streaming 1 GiB copies right before the first batch of each tree
"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bvh_amd
from bvh_amd import synth

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_rays = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
lib = bvh_amd._lib.load()
tris_h = synth.soup(n_tris, seed=7)
lo, hi = synth.scene_bounds(tris_h)
tris = torch.from_numpy(tris_h).cuda()
batches = [torch.from_numpy(synth.rays_closest(n_rays, lo, hi, seed=100 + i)).cuda() for i in range(3)]
hits = torch.empty((n_rays, 4), dtype=torch.float32, device="cuda")

def warm_process():                                         # the process's one-off costs on a throwaway tree (as bench.py does)
    t = torch.from_numpy(synth.soup(20000, seed=99)).cuda()
    bb, cc = bvh_amd.tri_bounds(t)
    b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
    p = bvh_amd.precompute_tris(t, b.device_prim_ids())
    for any_hit in ((True, False) if os.environ.get("PROBE_WARM_CLOSEST") else (True,)):
        for coop in (0, 1):
            lib.bvh_amd_tuning(-1, -1, coop, -1)
            for order in (True, False):
                bvh_amd.intersect(b, p, batches[0][:70000], any_hit=any_hit, robust=False, out=hits[:70000], sort_rays=order)
    lib.bvh_amd_tuning(-1, -1, -1, -1)
    torch.cuda.synchronize()

warm_process()
for tree in range(2):
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium))
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    t0 = time.perf_counter()
    bvh_amd.prepare_trace(bvh, n_rays)
    torch.cuda.synchronize()
    prep = (time.perf_counter() - t0) * 1e3
    if os.environ.get("PROBE_HEAT"):                        # ~60 ms of streaming copies right before the first batch: is the handicap the device's state?
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b2 = torch.empty_like(a)
        for _ in range(int(os.environ["PROBE_HEAT"])):
            b2.copy_(a)
        torch.cuda.synchronize()
        del a, b2
    rows = []
    for i in range(14):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        bvh_amd.intersect(bvh, prims, batches[i % 3], any_hit=False, robust=False, out=hits)
        t_issue = (time.perf_counter() - t0) * 1e3
        ev1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        plan = (ctypes.c_int * 4)()
        lib.bvh_amd_last_launch_plan(plan)
        rows.append((wall, ev0.elapsed_time(ev1), t_issue, tuple(plan)))
    print(f"tree {tree}: {n_tris} triangles, 2^{n_rays.bit_length() - 1} rays per batch, prepare_trace {prep:.2f} ms; per call: wall ms | GPU ms between events | host ms inside the call | plan")
    for i, r in enumerate(rows):
        print(f"  call {i + 1:2d}: {r[0]:7.3f} | {r[1]:7.3f} | {r[2]:6.3f} | {r[3]}")
    settled = min(r[0] for r in rows[-4:])
    print(f"  first call / settled call (wall): {rows[0][0] / settled:.3f}")
    del bvh, prims
