"""Developer tool: differential campaign for the candidate-heap kernels that serve heaps reaching below LDS (csrc/heap_head.inc): High builds with the
exact replay forced in every iteration, random scene kinds / sizes / scalar types, each stream against the compiled reference's.
    python tools/heap_fuzz.py [seconds] [first_seed] [exact|auto]      auto: the library decides per iteration (heap-free attempt, roll-back, replay)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
import oracle, bvh_amd
from bvh_amd import synth

forced = not (len(sys.argv) > 3 and sys.argv[3] == "auto")
if forced:
    os.environ["BVH_AMD_REINSERT"] = "exact"
orc = oracle.gpu_checker()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time(); ran = 0; bad = []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    kind = ("soup", "terrain", "sponza", "dups", "lattice", "clusters")[seed % 6]
    dtype = np.float64 if seed % 5 == 4 else np.float32
    n = int(rng.integers(190_000, 900_000)) if dtype == np.float32 else int(rng.integers(100_000, 400_000))
    if kind == "soup": tris = synth.soup(n, jitter=float(rng.choice([0.002, 0.01, 0.05])), seed=seed, dtype=dtype)
    elif kind == "terrain": tris = synth.terrain(n).astype(dtype)
    elif kind == "sponza": tris = synth.sponza_proxy(n).astype(dtype)
    elif kind == "dups": h = synth.soup(n // 2, seed=seed, dtype=dtype); tris = np.concatenate([h, h])
    elif kind == "lattice":
        side = int(round((n / 1.0) ** (1 / 3)))
        g = np.arange(side, dtype=dtype); org = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 1, 3)
        tri = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.25], [0.0, 0.5, 0.25]], dtype=dtype)
        tris = np.ascontiguousarray((org + tri[None]).reshape(-1, 9))
    else:
        c = rng.random((64, 3)); which = rng.integers(0, 64, n)
        ctr = c[which] + rng.normal(0, 0.01, (n, 3)); v = ctr[:, None, :] + rng.normal(0, 0.002, (n, 3, 3))
        tris = np.ascontiguousarray(v.reshape(n, 9).astype(dtype))
    bb, cc = orc.prep_tris(tris)
    parallel = bool(seed % 2)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL if parallel else oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_HIGH).serialize()
    f0, e0 = bvh_amd.reinsertion_stats()
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool() if parallel else None)
    f1, e1 = bvh_amd.reinsertion_stats()
    ok = gpu.serialize() == ref and (e1 - e0 == 3 or not forced)
    prof = bvh_amd.last_optimize_profile()
    print(f"seed {seed} {kind} {tris.dtype} n={len(tris)} {'pool' if parallel else 'serial'} nodes={gpu.node_count} replayed={e1 - e0} replacements={prof['replacements']} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok: bad.append(seed)
    ran += 1; seed += 1
print(f"ran {ran} High builds with the replay {'forced' if forced else 'as the library decides'} in {time.time() - t0:.0f} s, failures: {bad}", flush=True)
