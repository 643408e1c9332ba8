"""Developer probe: DefaultBuilder(pool, High) time and how many reinsertion iterations took the heap-free path, per scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bvh_amd
from bvh_amd import synth

scenes = {"soup_1m": lambda: synth.soup(1_000_000), "terrain_1m": lambda: synth.terrain(1_000_000), "sponza_262k": lambda: synth.sponza_proxy(262_144),
          "soup_262k": lambda: synth.soup(262_144)}
for name, gen in scenes.items():
    tris = torch.from_numpy(gen()).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    for pool in (bvh_amd.ThreadPool(), None):
        f0, e0 = bvh_amd.reinsertion_stats()
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=pool)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        f1, e1 = bvh_amd.reinsertion_stats()
        print(f"{name:12s} {'pool  ' if pool else 'serial'} High {min(ts):8.1f} ms  iterations per build: fast {(f1 - f0) // 2} exact {(e1 - e0) // 2}", flush=True)
