"""Developer probe: are the library's launch-time decisions (reorder the batch or not; per-lane or quad-cooperative record fetch and
its thresholds) near the best choice on scenes OUTSIDE the set they were fitted on? For every scene: kernel / call ms of the
default ("auto") against every forced alternative, hits compared by sha1.
    python tools/rule_check.py [out.txt]"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

lib = bvh_amd._lib.load()


def times(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
        torch.cuda.synchronize()                              # (the plan search reads a candidate's events once they have completed)
    torch.cuda.synchronize()
    lib.bvh_amd_kernel_timing(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    kt = (C.c_float * 64)(); got = C.c_size_t(0)
    lib.bvh_amd_kernel_times(kt, reps, C.byref(got))
    lib.bvh_amd_kernel_timing(0)
    return float(np.median(kt[:got.value])), e0.elapsed_time(e1) / reps


SCENES = {
    "soup_1m": (lambda: synth.soup(1_000_000), 2, True, 1 << 24),
    "terrain_1m": (lambda: synth.terrain(1_000_000), 2, True, 1 << 23),
    "sponza_262k": (lambda: synth.sponza_proxy(262144), 0, False, 1 << 22),
    "cornell_1m": (lambda: synth.cornell_tessellated(1_000_000), 2, True, 1 << 23),
    "clusters_1m": (lambda: synth.clusters(1_000_000), 2, True, 1 << 23),
    "clusters_4m": (lambda: synth.clusters(4_000_000), 1, True, 1 << 23),
    "soup_100k": (lambda: synth.soup(100_000), 2, True, 1 << 22),
}


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n"); out.flush()
    for name, (gen, q, pool, nr) in SCENES.items():
        t = gen()
        d = torch.from_numpy(t).cuda()
        bb, cc = bvh_amd.tri_bounds(d)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool() if pool else None)
        prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
        lo, hi = synth.scene_bounds(t)
        for any_hit, robust, rays_h in ((False, True, synth.rays_closest(nr, lo, hi)), (True, False, synth.rays_shadow(nr, lo, hi))):
            rays = torch.from_numpy(rays_h).cuda()
            hits = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
            _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit, robust, counters=True)
            P = float(cnt[0]) / nr
            rows = []
            want = None
            for label, sort, tun in (("auto", None, (-1, -1, -1)),
                                     ("as given, per lane 36/12", False, (36, 12, 0)), ("as given, coop 20/20", False, (20, 20, 1)), ("as given, coop 12/12", False, (12, 12, 1)),
                                     ("sorted,   per lane 36/12", True, (36, 12, 0)), ("sorted,   coop 20/20", True, (20, 20, 1)), ("sorted,   coop 12/12", True, (12, 12, 1))):
                lib.bvh_amd_tuning(*tun, -1)
                k_ms, c_ms = times(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=hits, sort_rays=sort), warm=10 if label == "auto" else 2)
                sha = hashlib.sha1(hits.cpu().numpy().tobytes()).hexdigest()[:12]
                want = want or sha
                pl = (C.c_int * 4)(); lib.bvh_amd_last_launch_plan(pl)
                rows.append((label, k_ms, c_ms, bool(pl[0]), f"{pl[1]} {pl[2]}/{pl[3]}", sha == want))
            lib.bvh_amd_tuning(-1, -1, -1, -1)
            best = min(r[2] for r in rows)
            emit(f"## {name} ({len(t)} tris, {bvh.node_count} nodes) {'any-hit fast' if any_hit else 'closest robust'}, {nr} rays, P = {P:.1f}")
            for label, k_ms, c_ms, reordered, coop, same in rows:
                emit(f"   {label:28s} kernel {k_ms:7.3f} ms  call {c_ms:7.3f} ms {nr / c_ms / 1e3:8.1f} Mrays/s  (reordered={int(reordered)} coop={coop}) "
                     f"{'' if same else 'HITS DIFFER '}{'<- best' if c_ms == best else f'{100 * (c_ms / best - 1):+.1f} % vs best' if label == 'auto' else ''}")
            del rays, hits
        del bvh, prims, d


if __name__ == "__main__":
    main()
