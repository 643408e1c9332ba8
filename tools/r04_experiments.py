"""Round-4 developer experiments on one MI355X, one process (bvh_amd_experiment / bvh_amd_tuning knobs; results never change and every
variant's hit records are compared with the baseline's). Prints one line per measurement; `python tools/r04_experiments.py [section ...]`.

  soup     1M soup, pool-High tree, 2^24 closest-hit rays, forced reordered + cooperative 12/12 (the bench's settled plan):
           stream hints / 64-byte triangles / Hilbert key / key bits, alone and together
  small    configs[1] (262k Sponza proxy, serial Low, 1M closest-hit rays) and configs[4] (1M f64 spheres, 1M rays): persistent grid capped at
           k blocks, batch sizes 1M / 2M / 4M
  first    first large batch through a fresh tree (predictor's plan) against the settled plan
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bvh_amd
from bvh_amd import _lib, synth

lib = _lib.load()


def knob(name, value):
    _lib.check(lib.bvh_amd_experiment(name.encode(), int(value)), "experiment")


def kernel_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.bvh_amd_kernel_timing(1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    kt = (C.c_float * 64)()
    got = C.c_size_t(0)
    _lib.check(lib.bvh_amd_kernel_times(kt, reps, C.byref(got)), "kernel_times")
    lib.bvh_amd_kernel_timing(0)
    ks = sorted(kt[:got.value])
    return ks[len(ks) // 2], ev0.elapsed_time(ev1) / reps


def section_soup():
    tris = synth.soup(1_000_000)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    padded = torch.zeros((prims.shape[0], 16), dtype=torch.float32, device="cuda")
    padded[:, :12] = prims.reshape(-1, 12)
    lo, hi = synth.scene_bounds(tris)
    nr = 1 << 24
    rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
    out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
    ref = None
    variants = [
        ("baseline", {}),
        ("stream_hints=1", {"stream_hints": 1}),
        ("tri_stride=16 (64-byte triangles)", {"tri_stride": 16}),
        ("key_curve=hilbert", {"key_curve": 1}),
        ("hilbert + key_bits=6", {"key_curve": 1, "key_bits": 6}),
        ("hilbert + key_bits=8", {"key_curve": 1, "key_bits": 8}),
        ("hints + stride16", {"stream_hints": 1, "tri_stride": 16}),
        ("hints + stride16 + hilbert", {"stream_hints": 1, "tri_stride": 16, "key_curve": 1}),
        ("baseline again", {}),
    ]
    for coop, refill, leaf in ((1, 12, 12), (0, 36, 12)):
        lib.bvh_amd_tuning(refill, leaf, coop, -1)
        for name, knobs in variants:
            knob("reset", 0)
            for k, v in knobs.items():
                knob(k, v)
            p = padded if knobs.get("tri_stride") == 16 else prims
            k_ms, pass_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, p, rays, False, True, out=out, sort_rays=True))
            same = True
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            print(f"soup_1m {'coop' if coop else 'per-lane'} {refill}/{leaf} | {name:38s} kernel {k_ms:7.3f} ms  pass {pass_ms:7.3f} ms  same_hits={same}", flush=True)
    knob("reset", 0)
    lib.bvh_amd_tuning(-1, -1, -1, -1)


def section_small():
    # configs[1]: Sponza proxy, serial Low build, exactly 1M closest-hit rays (and 2M / 4M for the trend)
    tris = synth.sponza_proxy(262_144)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Low))
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    for nr in (1_000_000, 2_000_000, 4_000_000):
        rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        for coop in (0, 1):
            lib.bvh_amd_tuning(-1, -1, coop, -1)
            for blocks in (-1, 256, 512, 768, 1024, 1280, 1536, 1792):
                knob("grid_blocks", blocks)
                k_ms, pass_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, False, True, out=out), reps=9, warm=3)
                print(f"configs[1] sponza_262k rays={nr} {'coop' if coop else 'per-lane'} grid_blocks={blocks:5d}: kernel {k_ms:7.4f} ms  call {pass_ms:7.4f} ms  "
                      f"{nr / k_ms / 1e3:8.1f} Mrays/s (kernel)  {nr / pass_ms / 1e3:8.1f} (call)", flush=True)
    knob("reset", 0)
    lib.bvh_amd_tuning(-1, -1, -1, -1)
    # configs[4]: 1M double-precision spheres, (pool, High), 1M robust closest-hit rays
    sph = synth.spheres(1_000_000)
    d_sph = torch.from_numpy(sph).cuda()
    d_bb, d_cc = bvh_amd.sphere_bounds(d_sph)
    b64 = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    sp = bvh_amd.gather(d_sph, b64.device_prim_ids())
    lo, hi = synth.scene_bounds(sph)
    for nr in (1_000_000, 4_000_000):
        rays = torch.from_numpy(synth.rays_closest(nr, lo, hi, dtype=np.float64)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float64, device="cuda")
        for blocks in (-1, 256, 512, 768, 1024, 1280):
            knob("grid_blocks", blocks)
            k_ms, pass_ms = kernel_ms(lambda: bvh_amd.intersect(b64, sp, rays, False, True, leaf="sphere", out=out), reps=9, warm=3)
            print(f"configs[4] spheres_f64_1m rays={nr} grid_blocks={blocks:5d}: kernel {k_ms:7.4f} ms  call {pass_ms:7.4f} ms  "
                  f"{nr / k_ms / 1e3:8.1f} Mrays/s (kernel) kernel={lib.bvh_amd_last_kernel_name().decode()}", flush=True)
    knob("reset", 0)


def section_first():
    for scene, n_tris, nr in (("soup", 1_000_000, 1 << 24), ("terrain", 1_000_000, 1 << 23), ("soup", 4_000_000, 12_500_000)):
        tris = getattr(synth, scene)(n_tris)
        d_tris = torch.from_numpy(tris).cuda()
        lo, hi = synth.scene_bounds(tris)
        rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        bb, cc = bvh_amd.tri_bounds(d_tris)
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium), thread_pool=bvh_amd.ThreadPool())
        prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
        plan = (C.c_int * 4)()
        times = []
        for i in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bvh_amd.intersect(bvh, prims, rays, False, True, out=out)
            torch.cuda.synchronize()
            lib.bvh_amd_last_launch_plan(plan)
            times.append(((time.perf_counter() - t0) * 1e3, list(plan), lib.bvh_amd_last_kernel_name().decode().split("<")[0]))
        for i, (ms, pl, kn) in enumerate(times):
            print(f"first-call {scene}_{n_tris} rays={nr} call {i:2d}: {ms:8.3f} ms  plan(reorder, coop, refill, leaf)={pl} {kn}", flush=True)


def _scene(gen, n, quality, pool=True):
    tris = getattr(synth, gen)(n)
    d_tris = torch.from_numpy(tris).cuda()
    bb, cc = bvh_amd.tri_bounds(d_tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=quality), thread_pool=bvh_amd.ThreadPool() if pool else None)
    prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    return bvh, prims, lo, hi


def section_grid():
    """Resident blocks per CU of the persistent grid, per kernel, on every kind of launch; and the round-4 defaults (stream hints, Hilbert key) on / off."""
    cases = [("terrain", 1_000_000, bvh_amd.Quality.High, True, 1 << 23, False, None),
             ("sponza_proxy", 262_144, bvh_amd.Quality.Low, False, 10_000_000, True, None),
             ("sponza_proxy", 262_144, bvh_amd.Quality.Low, False, 4_000_000, False, None),
             ("soup", 1_000_000, bvh_amd.Quality.High, True, 1 << 24, False, False),
             ("soup", 10_000_000, bvh_amd.Quality.Medium, True, 12_500_000, False, True)]
    for gen, n, q, pool, nr, any_hit, sort in cases:
        bvh, prims, lo, hi = _scene(gen, n, q, pool)
        rays = torch.from_numpy(synth.rays_shadow(nr, lo, hi) if any_hit else synth.rays_closest(nr, lo, hi)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        tag = f"{gen}_{n} {'any-hit' if any_hit else 'closest'} rays={nr} {'reordered' if sort else 'as given' if sort is False else 'library order'}"
        for coop, refill, leaf in ((0, 36, 12), (1, 20 if any_hit else 12, 20 if any_hit else 12)):
            lib.bvh_amd_tuning(refill, leaf, coop, -1)
            cells = []
            for per_cu in (4, 5, 6, 7, 8):
                knob("grid_blocks", per_cu * 256)
                k_ms, _ = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, True, out=out, sort_rays=sort), reps=5, warm=2)
                cells.append(f"{per_cu}/CU {k_ms:7.3f}")
            knob("grid_blocks", -1)
            print(f"grid | {tag:62s} {'coop' if coop else 'per-lane'} {refill}/{leaf}: " + "  ".join(cells) + " ms", flush=True)
        if sort:
            lib.bvh_amd_tuning(12, 12, 1, -1)
            for name, knobs in (("defaults (hints on, hilbert)", {}), ("hints off", {"stream_hints": 0}), ("morton", {"key_curve": 0}), ("both off", {"stream_hints": 0, "key_curve": 0})):
                knob("reset", 0)
                for k, v in knobs.items():
                    knob(k, v)
                k_ms, p_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, True, out=out, sort_rays=True), reps=5, warm=2)
                print(f"defaults | {tag:58s} coop 12/12 {name:30s}: kernel {k_ms:7.3f} pass {p_ms:7.3f} ms", flush=True)
            knob("reset", 0)
        else:
            for coop, refill, leaf in ((0, 36, 12), (1, 20 if any_hit else 12, 20 if any_hit else 12)):
                lib.bvh_amd_tuning(refill, leaf, coop, -1)
                for hints in (1, 0):
                    knob("stream_hints", hints)
                    k_ms, p_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, True, out=out, sort_rays=sort), reps=5, warm=2)
                    print(f"defaults | {tag:58s} {'coop' if coop else 'per-lane'} stream_hints={hints}: kernel {k_ms:7.3f} pass {p_ms:7.3f} ms", flush=True)
            knob("reset", 0)
        lib.bvh_amd_tuning(-1, -1, -1, -1)
        del bvh, prims, rays, out


def section_double():
    """configs[4] (1M double-precision spheres, pool-High, robust closest-hit) and its any-hit / fast twin: per-lane against the round-4
    cooperative fetch of the 128-byte records (two quad-coalesced halves), thresholds, batch sizes; then what the library picks itself."""
    sph = synth.spheres(1_000_000)
    d_sph = torch.from_numpy(sph).cuda()
    d_bb, d_cc = bvh_amd.sphere_bounds(d_sph)
    b64 = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    sp = bvh_amd.gather(d_sph, b64.device_prim_ids())
    lo, hi = synth.scene_bounds(sph)
    for any_hit in (False, True):
        for nr in (1_000_000, 4_000_000, 16_000_000):
            rays = torch.from_numpy((synth.rays_shadow if any_hit else synth.rays_closest)(nr, lo, hi, dtype=np.float64)).cuda()
            out = torch.empty((nr, 4), dtype=torch.float64, device="cuda")
            ref = None
            for coop, refill, leaf in ((0, 36, 12), (1, 12, 12), (1, 20, 20), (1, 36, 12)):
                lib.bvh_amd_tuning(refill, leaf, coop, -1)
                for sort in ((False, True) if nr >= 4_000_000 and not any_hit else (False,)):
                    k_ms, c_ms = kernel_ms(lambda: bvh_amd.intersect(b64, sp, rays, any_hit, not any_hit, leaf="sphere", out=out, sort_rays=sort), reps=7, warm=2)
                    same = True if ref is None else bool(torch.equal(out.view(torch.int64), ref.view(torch.int64)))
                    ref = out.clone() if ref is None else ref
                    print(f"double | spheres_f64_1m {'any-hit fast' if any_hit else 'closest robust'} rays={nr:9d} {'coop' if coop else 'per-lane'} {refill}/{leaf} "
                          f"{'reordered' if sort else 'as given '}: kernel {k_ms:7.4f} call {c_ms:7.4f} ms  {nr / c_ms / 1e3:8.1f} Mrays/s (call) same={same} "
                          f"{lib.bvh_amd_last_kernel_name().decode().split('<')[0]}", flush=True)
            lib.bvh_amd_tuning(-1, -1, -1, -1)
            plan = (C.c_int * 4)()
            for i in range(10):
                bvh_amd.intersect(b64, sp, rays, any_hit, not any_hit, leaf="sphere", out=out)
                torch.cuda.synchronize()
            k_ms, c_ms = kernel_ms(lambda: bvh_amd.intersect(b64, sp, rays, any_hit, not any_hit, leaf="sphere", out=out), reps=7, warm=2)
            lib.bvh_amd_last_launch_plan(plan)
            print(f"double | spheres_f64_1m {'any-hit fast' if any_hit else 'closest robust'} rays={nr:9d} LIBRARY'S OWN CHOICE plan={list(plan)}: kernel {k_ms:7.4f} "
                  f"call {c_ms:7.4f} ms  {nr / c_ms / 1e3:8.1f} Mrays/s (call)", flush=True)
    # Node<float, 2> / Node<double, 2>: circles
    rng = np.random.default_rng(9)
    for dtype in (np.float32, np.float64):
        n = 1_000_000
        circ = np.ascontiguousarray(np.concatenate([rng.random((n, 2)), 0.0002 + 0.0008 * rng.random((n, 1))], axis=1).astype(dtype))
        d_bb, d_cc = bvh_amd.sphere_bounds(torch.from_numpy(circ).cuda())
        b2 = bvh_amd.DefaultBuilder.build(d_bb, d_cc, bvh_amd.Config(quality=bvh_amd.Quality.Medium))
        d_ord = bvh_amd.gather(torch.from_numpy(circ).cuda(), b2.device_prim_ids())
        nr = 4_000_000
        org = rng.random((nr, 2)) * 1.1 - 0.05
        ang = rng.random(nr) * 2 * np.pi
        rays = torch.from_numpy(np.ascontiguousarray(np.concatenate([org, np.cos(ang)[:, None], np.sin(ang)[:, None], np.zeros((nr, 1)),
                                                                     np.full((nr, 1), np.finfo(dtype).max)], axis=1).astype(dtype))).cuda()
        for coop, refill, leaf in ((0, 36, 12), (1, 12, 12), (1, 20, 20)):
            lib.bvh_amd_tuning(refill, leaf, coop, -1)
            k_ms, c_ms = kernel_ms(lambda: bvh_amd.intersect(b2, d_ord, rays, False, True), reps=7, warm=2)
            print(f"double | circles_2{'f' if dtype == np.float32 else 'd'}_1m closest rays={nr} {'coop' if coop else 'per-lane'} {refill}/{leaf}: kernel {k_ms:7.4f} "
                  f"call {c_ms:7.4f} ms {lib.bvh_amd_last_kernel_name().decode()}", flush=True)
        lib.bvh_amd_tuning(-1, -1, -1, -1)


def section_hints():
    """Which once-touched data to load non-temporally in a reordered launch: bit 0 = rays / order / hit records (default on); bit 1 = triangles
    existed for the run recorded in profiles/r04_experiments_call4_hints.txt (23-35 % SLOWER: removed from the kernel again)."""
    for gen, n, q, nr in (("soup", 1_000_000, bvh_amd.Quality.High, 1 << 24), ("soup", 10_000_000, bvh_amd.Quality.Medium, 12_500_000)):
        bvh, prims, lo, hi = _scene(gen, n, q, True)
        rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        for coop, refill, leaf in ((1, 12, 12), (0, 36, 12)):
            lib.bvh_amd_tuning(refill, leaf, coop, -1)
            for hints in (1, 3, 0, 2, 1, 3):
                knob("stream_hints", hints)
                k_ms, p_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, False, True, out=out, sort_rays=True), reps=7, warm=2)
                print(f"hints | {gen}_{n} rays={nr} {'coop' if coop else 'per-lane'} stream_hints={hints}: kernel {k_ms:7.3f} pass {p_ms:7.3f} ms", flush=True)
        knob("reset", 0)
        lib.bvh_amd_tuning(-1, -1, -1, -1)
        del bvh, prims, rays, out


def section_ramp():
    """Kernel time against batch size on configs[1]'s tree: t(n) = t0 + n / R. t0 is what a small batch cannot amortise (first touches of
    the tree, the longest ray's dependent chain, the drain of the persistent waves)."""
    bvh, prims, lo, hi = _scene("sponza_proxy", 262_144, bvh_amd.Quality.Low, False)
    for nr in (4096, 16384, 65536, 131072, 262144, 524288, 1_000_000, 2_000_000, 4_000_000, 8_000_000):
        rays = torch.from_numpy(synth.rays_closest(nr, lo, hi)).cuda()
        out = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        cells = []
        for blocks in (-1, 1280, 768):
            knob("grid_blocks", blocks)
            k_ms, c_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, False, True, out=out), reps=9, warm=3)
            cells.append(f"grid {blocks:5d}: kernel {k_ms:7.4f} call {c_ms:7.4f}")
        knob("grid_blocks", -1)
        print(f"ramp | sponza_262k closest rays={nr:8d} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["soup", "small", "first"]
    print(torch.cuda.get_device_name(0), flush=True)
    for w in which:
        {"soup": section_soup, "small": section_small, "first": section_first, "grid": section_grid, "ramp": section_ramp, "double": section_double, "hints": section_hints}[w]()
