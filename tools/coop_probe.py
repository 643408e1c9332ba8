"""Developer probe (one process, bvh_amd_tuning): the float 3D traversal kernels with per-lane against quad-cooperative record
fetch, over refill / leaf thresholds, on the scenes of the bench and of BASELINE configs[1..3]. Kernel ms from the library's own
events (reordering excluded), hits compared by sha1 against the default configuration of each scene.
    python tools/coop_probe.py [out.txt] [quick]"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bvh_amd
from bvh_amd import synth

lib = bvh_amd._lib.load()


def kernel_ms(fn, reps):
    fn(); fn()                                                # (forced plans: nothing to settle)
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


def scene(name):
    if name == "soup":
        t = synth.soup(1_000_000); q, pool, nr, any_hit, robust = 2, True, 1 << 24, False, True
    elif name == "terrain":
        t = synth.terrain(1_000_000); q, pool, nr, any_hit, robust = 2, True, 1 << 23, False, True
    elif name == "sponza":
        t = synth.sponza_proxy(262144); q, pool, nr, any_hit, robust = 0, False, 1 << 22, False, True
    elif name == "sponza_any":
        t = synth.sponza_proxy(262144); q, pool, nr, any_hit, robust = 0, False, 10_000_000, True, False
    elif name == "soup10m":
        t = synth.soup(10_000_000); q, pool, nr, any_hit, robust = 1, True, 12_500_000, False, True
    d = torch.from_numpy(t).cuda()
    bb, cc = bvh_amd.tri_bounds(d)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality(q)), thread_pool=bvh_amd.ThreadPool() if pool else None)
    prims = bvh_amd.precompute_tris(d, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(t)
    rays = torch.from_numpy(synth.rays_shadow(nr, lo, hi) if any_hit else synth.rays_closest(nr, lo, hi)).cuda()
    return bvh, prims, rays, any_hit, robust


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 and sys.argv[1] != "quick" else None
    quick = "quick" in sys.argv

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n"); out.flush()
    combos = [(0, 36, 12), (1, 36, 12), (1, 28, 12), (1, 20, 12), (1, 12, 12), (1, 8, 8), (1, 20, 20), (1, 12, 24), (0, 20, 12)]
    if quick:
        combos = combos[:3]
    for name in (["soup", "sponza"] if quick else ["soup", "sponza", "sponza_any", "terrain", "soup10m"]):
        bvh, prims, rays, any_hit, robust = scene(name)
        n = rays.shape[0]
        hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        want = None
        for coop, refill, leaf in combos:
            lib.bvh_amd_tuning(refill, leaf, coop, -1)
            k_ms, call_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=hits), 5)
            sha = hashlib.sha1(hits.cpu().numpy().tobytes()).hexdigest()[:12]
            want = want or sha
            emit(f"{name:10s} coop={coop} refill={refill:2d} leaf={leaf:2d}: kernel {k_ms:7.3f} ms {n / k_ms / 1e3:8.1f} Mrays/s | call {call_ms:7.3f} ms {n / call_ms / 1e3:8.1f} Mrays/s "
                 f"| {lib.bvh_amd_last_kernel_name().decode()} hits {'==' if sha == want else '!= MISMATCH'}")
        lib.bvh_amd_tuning(-1, -1, -1, -1)
        del bvh, prims, rays, hits


if __name__ == "__main__":
    main()
