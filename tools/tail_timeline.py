"""Drain-tail timeline of the bench launch (VERDICT r4 Next 2(ii)): when does every wavefront of the persistent grid start, draw its
last ticket and leave, per XCD — and how much of the kernel's time is ramp-up and drain rather than steady state.

    BVH_AMD_LIB=bvh_amd/lib/libbvh_amd_dev.so python tools/tail_timeline.py [--workload soup_1m] [--rays 16777216] > profiles/r05_tail_timeline.txt

Needs the developer library (the release kernels carry no timestamp code): bvh_amd_experiment("wave_times", 1) + bvh_amd_wave_times.
Timestamps are s_memrealtime ticks (100 MHz: 10 ns)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BVH_AMD_LIB", os.path.join(ROOT, "bvh_amd", "lib", "libbvh_amd_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="soup_1m")
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--quality", default="high")
    ap.add_argument("--parts", type=int, default=-1, help="ticket ranges (default: the library's)")
    args = ap.parse_args()
    import torch
    import bvh_amd
    import bench
    from bvh_amd import synth
    lib = bvh_amd._lib.load()
    gen, n_tris, desc, _ = bench.WORKLOADS[args.workload]
    tris = getattr(synth, gen)(n_tris)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality[args.quality.capitalize()]), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = torch.from_numpy(synth.rays_closest(args.rays, lo, hi, seed=1234)).cuda()
    hits = torch.empty((args.rays, 4), dtype=torch.float32, device="cuda")
    for _ in range(12):                                       # the plan search settles
        bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)
        torch.cuda.synchronize()
    plan = (C.c_int * 4)()
    lib.bvh_amd_last_launch_plan(plan)
    if args.parts > 0:
        lib.bvh_amd_tuning(-1, -1, -1, args.parts)
    lib.bvh_amd_kernel_timing(1)
    bvh_amd._lib.check(lib.bvh_amd_experiment(b"wave_times", 1), "experiment")
    reps = 5
    rows = []
    for rep in range(reps):
        bvh_amd.intersect(bvh, prims, rays, robust=True, out=hits)
        torch.cuda.synchronize()
        n = C.c_size_t(0)
        bvh_amd._lib.check(lib.bvh_amd_wave_times(None, 0, C.byref(n)), "wave_times")
        buf = np.zeros((n.value, 6), dtype=np.uint64)
        bvh_amd._lib.check(lib.bvh_amd_wave_times(buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "wave_times")
        rows.append(buf)
    kt = (C.c_float * 16)()
    got = C.c_size_t(0)
    lib.bvh_amd_kernel_times(kt, reps, C.byref(got))
    lib.bvh_amd_kernel_timing(0)
    print(f"# drain-tail timeline: {args.workload} ({desc}), {args.rays} closest-hit rays, robust, Quality::{args.quality}; plan reordered={plan[0]} coop={plan[1]} "
          f"refill={plan[2]} leaf={plan[3]}; kernel {lib.bvh_amd_last_kernel_name().decode()}")
    print(f"# kernel_ms of the {reps} instrumented launches (HIP events): {[round(kt[i], 4) for i in range(got.value)]}")
    for rep, buf in enumerate(rows):
        buf = buf[buf[:, 2] != 0]
        begin, last, end = (buf[:, k].astype(np.int64) for k in range(3))
        xcc = (buf[:, 3] >> np.uint64(32)).astype(np.int64)
        nrays = (buf[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        t0 = begin.min()
        us = lambda t: (t - t0) / 100.0                       # noqa: E731  ticks of 10 ns -> microseconds
        total = us(end.max())
        print(f"\n## launch {rep}: {len(buf)} waves, {int(nrays.sum())} rays, first begin -> last end = {total:.1f} us")
        print(f"   begin   : min 0.0  median {np.median(us(begin)):.1f}  max {us(begin).max():.1f} us   (ramp-up: the last wave starts this late)")
        print(f"   last ticket draw: min {us(last).min():.1f}  median {np.median(us(last)):.1f}  max {us(last).max():.1f} us")
        print(f"   end     : min {us(end).min():.1f}  p10 {np.percentile(us(end), 10):.1f}  median {np.median(us(end)):.1f}  p90 {np.percentile(us(end), 90):.1f}  max {us(end).max():.1f} us")
        busy = (end - begin).sum() / 100.0
        print(f"   wave-time between begin and end / (waves x span) = {busy / (len(buf) * total):.4f}   (1 - this = share of the grid's time lost to ramp-up + drain)")
        drain = (end - last) / 100.0
        print(f"   drain of a wave (last ticket draw -> end): median {np.median(drain):.1f}  p90 {np.percentile(drain, 90):.1f}  max {drain.max():.1f} us")
        print("   per XCC: waves, rays, median end, max end, last ticket drawn (us)")
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            print(f"     xcc {x}: {int(m.sum()):5d} waves {int(nrays[m].sum()):9d} rays  end median {np.median(us(end[m])):8.1f}  max {us(end[m]).max():8.1f}   last draw {us(last[m]).max():8.1f}")
        # how many waves are still running as the launch ends
        edges = [0.80, 0.90, 0.95, 0.97, 0.98, 0.99, 0.995, 1.0]
        print("   waves still running at x of the span: " + "  ".join(f"{e:.3f}: {int((us(end) > e * total - 1e-9).sum())}" for e in edges))
        rt, rc = buf[:, 4].astype(np.int64) / 100.0, buf[:, 5].astype(np.int64)
        print(f"   refills per wave: median {int(np.median(rc))}; time inside refills (ticket atomic -> order -> ray loaded) per wave: median {np.median(rt):.1f} us "
              f"= {np.median(rt / np.maximum(1e-9, (end - begin) / 100.0)):.4f} of the wave's life; per refill {np.median(rt / np.maximum(1, rc)):.2f} us")
        print(f"   rays per wave: min {nrays.min()} median {int(np.median(nrays))} max {nrays.max()}")


if __name__ == "__main__":
    main()
