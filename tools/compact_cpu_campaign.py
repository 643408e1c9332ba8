"""Developer tool (CPU only): the compact-record kernel body compiled for the host (tests/cpp/trace_body_host.cpp, one emulated lane and
a full 64-lane wavefront) against the oracle on many more seeds of the adversarial generators than the committed tests run.

    python tools/compact_cpu_campaign.py <first_seed> <last_seed> [seconds]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle
import test_gpu_fuzz as F
import test_compact_pairs as T

tmp = tempfile.mkdtemp()


def build(name, src, extra):
    out = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unknown-pragmas", "-shared", "-fPIC"]
                          + extra + [os.path.join(root, "tests", "cpp", src), "-o", out])
    return C.CDLL(out)


walker = build("w.so", "compact_pair_walk.cpp", [])
walker.compact_encode_tree.restype = C.c_int
walker.compact_encode_tree.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
bodies = []
for name, extra in (("b1.so", []), ("b64.so", ["-DBVH_HOST_WAVE64"])):
    dll = build(name, "trace_body_host.cpp", extra)
    dll.trace_body_host.restype = C.c_int
    dll.trace_body_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    bodies.append(dll)

orc = oracle.load_oracle()
lo_seed, hi_seed = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
t0 = time.time(); ran = 0; bad = []
KINDS = ("lattice", "dups", "flat", "points", "scales", "uniform")
for seed in range(lo_seed, hi_seed):
    for kind in KINDS:
        if time.time() - t0 > budget:
            break
        rng = np.random.default_rng(50000 + 10 * seed + KINDS.index(kind))
        n = int(rng.choice([2, 5, 17, 64, 65, 200, 1500, 4000]))
        tris = F._scene3(rng, n, kind, np.float32)
        bb, cc = orc.prep_tris(tris)
        lo = tris.reshape(-1, 3).min(axis=0).astype(np.float64)
        hi = tris.reshape(-1, 3).max(axis=0).astype(np.float64)
        rays = F._rays3(rng, 1500, lo, hi, np.float32)
        lim = [(1, 8), (1, 1), (2, 4), (3, 15)][seed % 4]
        builder, quality = [(0, 0), (0, 2), (1, 2), (1, 0), (3, 0), (2, 0)][seed % 6]
        bvh = orc.build(bb, cc, builder=builder, quality=quality, min_leaf=lim[0], max_leaf=lim[1], parallel_threshold=[1024, 64][seed % 2])
        nodes, ids = bvh.nodes(), bvh.prim_ids()
        if len(nodes) < 3:
            continue
        rc, pairs, recs = T._encode(walker, nodes)
        ok = rc == 0
        prims = orc.precompute_tris(tris, ids)
        for any_hit in (False, True):
            for robust in (False, True):
                ref_hits, ref_cnt = bvh.intersect_tri(prims, rays, any_hit, robust, counters=True)
                for body in bodies:
                    hits, cnt = T._run_body(body, nodes, pairs, recs, prims, rays, any_hit, robust, True)
                    ok = ok and hits.tobytes() == ref_hits.tobytes() and bool((cnt == ref_cnt).all())
        ran += 1
        if not ok:
            bad.append((seed, kind)); print("FAIL", seed, kind, flush=True)
    if time.time() - t0 > budget:
        break
print(f"ran {ran} scenes x 4 traversal modes x (1 lane, 64 lanes) in {time.time() - t0:.1f} s, failures: {len(bad)}", flush=True)
