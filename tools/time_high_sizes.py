"""Times DefaultBuilder High builds (thread-pool flavour, as the bench uses) of the soup at several sizes: wall time, heap kernels, the rest.
    python tools/time_high_sizes.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bvh_amd
from bvh_amd import synth
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    tris = torch.from_numpy(synth.soup(n)).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        prof = bvh_amd.last_optimize_profile()
        del bvh
    r = max(prof["replacements"], 1)
    print(f"soup {n}: High build {min(ts):.1f} ms (runs {[round(t, 1) for t in ts]}), heap kernels {prof['heap_ms']:.1f} ms = {prof['replayed']} replays, {prof['replacements']} replacements, "
          f"{prof['heap_ms'] * 1e3 / r:.3f} us each; everything else {min(ts) - prof['heap_ms']:.1f} ms", flush=True)
