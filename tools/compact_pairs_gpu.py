"""Developer probe for the EXPERIMENTAL compact traversal records (BVH_AMD_PAIRS=compact; bvh_amd/csrc/compact_pair.h).

    python tools/compact_pairs_gpu.py [soup|terrain|sponza|spheres64] [n_prims] [n_rays]

Runs itself twice in child processes (the switch is read once per process): PairNode kernel, then compact kernel. Each child
traces closest-hit (robust + fast) and any-hit rays, prints kernel name, ms and Mrays/s, and writes the hit bytes' checksum; the
parent checks that both children produced identical hit records and counters, and — on a sample — that they equal the oracle's.
NOT the bench, NOT a test of the default path: the compact kernel was written without access to a GPU (round 1 ran out of
GPU minutes) and must pass this before anything else is concluded from it.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scene, n, nr):
    import numpy as np
    import torch
    import bvh_amd
    from bvh_amd import synth
    spheres = scene == "spheres64"                           # BASELINE configs[4]: double precision, sphere primitives
    dt = np.float64 if spheres else np.float32
    leaf = "sphere" if spheres else "tri"
    if spheres:
        tris = synth.spheres(n)
        d_sph = torch.from_numpy(tris).cuda()
        bb, cc = bvh_amd.sphere_bounds(d_sph)
    else:
        tris = {"soup": synth.soup, "terrain": synth.terrain, "sponza": synth.sponza_proxy}[scene](n)
        d_tris = torch.from_numpy(tris).cuda()
        bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.gather(d_sph, bvh.device_prim_ids()) if spheres else bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    res = {"mode": os.environ.get("BVH_AMD_PAIRS", "pairnode"), "runs": []}
    lib = bvh_amd._lib.load()
    for name, rays_h, any_hit, robust in (("closest_robust", synth.rays_closest(nr, lo, hi, dtype=dt), False, True),
                                          ("closest_fast", synth.rays_closest(nr, lo, hi, dtype=dt), False, False),
                                          ("shadow_robust", synth.rays_shadow(nr, lo, hi, dtype=dt), True, True)):
        rays = torch.from_numpy(rays_h).cuda()
        out = torch.empty((nr, 4), dtype=torch.float64 if spheres else torch.float32, device="cuda")
        _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, counters=True)
        for _ in range(2):
            bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, out=out)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        ev0.record()
        for _ in range(reps):
            bvh_amd.intersect(bvh, prims, rays, any_hit, robust, leaf=leaf, out=out)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        hits = bvh_amd.hits_to_numpy(out)
        sample_ok = None
        if name == "closest_robust":                         # oracle on a sample (test infrastructure; the same tree, from its stream)
            import oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import parse_stream
            orc = oracle.load_oracle()
            nodes, ids = parse_stream(bvh.serialize(), spheres)
            cb = orc.from_arrays(nodes, ids)
            ns = min(nr, 200_000)
            if spheres:
                ref = cb.intersect_sphere(tris[ids.astype(np.int64)], rays_h[:ns], any_hit, robust, threads=orc.hardware_threads())
            else:
                ref = cb.intersect_tri(orc.precompute_tris(tris, ids), rays_h[:ns], any_hit, robust, threads=orc.hardware_threads())
            sample_ok = bool(ref.tobytes() == hits[:ns].tobytes())
        res["runs"].append({"name": name, "kernel": lib.bvh_amd_last_kernel_name().decode(), "ms": round(ms, 3),
                            "mrays_s": round(nr / ms / 1e3, 1), "counters": [int(x) for x in cnt.cpu().numpy()],
                            "sha1": hashlib.sha1(hits.tobytes()).hexdigest(), "equals_oracle_on_sample": sample_ok})
    print("RESULT " + json.dumps(res), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    scene = sys.argv[1] if len(sys.argv) > 1 else "soup"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 23
    results = []
    for mode in ("pairnode", "compact"):
        env = dict(os.environ)
        env.pop("BVH_AMD_PAIRS", None)
        if mode == "compact":
            env["BVH_AMD_PAIRS"] = "compact"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scene, str(n), str(nr)], env=env, capture_output=True,
                           text=True, timeout=400)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(f"[{mode}] FAILED rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
            return 1
        results.append(json.loads(line[0][7:]))
    ok = True
    for a, b in zip(results[0]["runs"], results[1]["runs"]):
        same = a["sha1"] == b["sha1"] and a["counters"] == b["counters"]
        ok &= same and "compact" in b["kernel"] and a["equals_oracle_on_sample"] is not False and b["equals_oracle_on_sample"] is not False
        print(f"{scene} n={n} rays={nr} {a['name']}: {a['kernel']} {a['ms']} ms {a['mrays_s']} Mrays/s | {b['kernel']} {b['ms']} ms "
              f"{b['mrays_s']} Mrays/s | speedup {a['ms'] / b['ms']:.3f} | identical hits+counters: {same} | oracle sample: "
              f"{a['equals_oracle_on_sample']}/{b['equals_oracle_on_sample']}")
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
