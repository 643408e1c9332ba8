#!/usr/bin/env python3
"""bench.py — Mrays/s closest-hit on a ~1M-triangle scene, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload soup_1m|sponza_262k|terrain_1m|soup_10m] [--obj mesh.obj]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic rays: `rays_per_gpu` uniform-random
closest-hit rays (origin uniform in the 1.1x scene box, direction uniform on the sphere, seed 1234 + rank)
traced through the device-resident BVH by ONE launch of the traversal kernel; rays and hit records are
resident in HBM before/after the timed region. The BVH is built on the GPU by the product builder (rank 0),
serialized in the reference's byte format and broadcast to the other ranks over RCCL; nothing else is
exchanged (weak scaling: every rank traces its own `rays_per_gpu`).

Rank 0 prints ONE JSON line (metric/value/unit/... + "roofline" + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (generator, n_tris, description, label of the scene in `metric`)
    "soup_1m": ("soup", 1_000_000, "1,000,000-triangle random soup (M3), worst-case incoherent", "1M-tri scene"),
    "terrain_1m": ("terrain", 1_000_000, "~1M-triangle height field (M2), tie-heavy", "1M-tri terrain"),
    "sponza_262k": ("sponza_proxy", 262_144, "262,144-triangle Sponza proxy (M1) — BASELINE configs[1]", "262k-tri Sponza proxy"),
    "soup_10m": ("soup", 10_000_000, "10,000,000-triangle random soup (M3-10M) — BASELINE configs[3]; use --rays 12500000 for its 100M / 8 shards",
                 "10M-tri scene"),
}
L2_BYTES = 8 * 4 * 1024 * 1024    # MI355X_MICROARCH.md: 4 MiB of L2 per XCD, 8 XCDs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="soup_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--rays", type=int, default=1 << 24, help="rays per GPU per step")
    ap.add_argument("--fast", action="store_true", help="intersect_fast instead of the robust slab test")
    ap.add_argument("--quality", default="high", choices=["low", "medium", "high"], help="DefaultBuilder quality of the traced BVH")
    ap.add_argument("--serial-builder", action="store_true", help="DefaultBuilder without a thread pool (binned/sweep) instead of mini-trees")
    ap.add_argument("--obj", default=None, help="trace this Wavefront OBJ mesh (reference loader semantics, bvh_amd/obj.py) instead of a synthetic workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16_000_000, help="rays of the CPU baseline sample (>= 1 s of work on the host cores)")
    ap.add_argument("--no-reorder", action="store_true", help="trace the rays in the order given (BVH_AMD_RAY_UNSORTED)")
    ap.add_argument("--no-probe", action="store_true", help="skip the record-walk probe behind roofline_binding.peak")
    return ap.parse_args()


def cpu_baseline(tris, bvh, rays_sample, robust, gpu_hits_sample, quality, serial):
    """The reference's CPU path (oracle/_ref when present, else the restatement) on a bounded sample of the same workload.
    Traversal as SURVEY.md 8(d) defines it: the ray array split by the reference's own ParallelExecutor::for_each over a
    persistent ThreadPool of all host threads, each worker running the benchmark.cpp:277-298 loop; one untimed pass, then the
    median of 3. Also a parity spot-check of the GPU result. Test infrastructure only."""
    import ctypes as C
    import oracle
    lib = oracle.load_ref()
    kind = "reference"
    if lib is None:
        lib = oracle.load_oracle()
        kind = "port"
    threads = lib.hardware_threads()
    cb = lib.from_arrays(bvh.nodes, bvh.prim_ids)
    prims = lib.precompute_tris(tris, bvh.prim_ids)
    n = len(rays_sample)
    fn = getattr(lib.dll, "ref_bench_tri3f", None) if kind == "reference" else None
    if fn is not None:
        fn.restype, fn.argtypes = None, [C.c_void_p] * 3 + [C.c_size_t] + [C.c_int] * 4 + [C.c_void_p] * 2
        hits = np.empty(n, dtype=oracle.HITF)
        secs = np.zeros(3)
        fn(cb.h, prims.ctypes.data, rays_sample.ctypes.data, n, 0, int(robust), threads, 3, hits.ctypes.data, secs.ctypes.data)
        how = "ParallelExecutor::for_each over a persistent ThreadPool"
    else:                                                     # the restatement has no executor: std::thread chunks per pass
        secs = []
        cb.intersect_tri(prims, rays_sample[:65536], 0, robust, threads=threads)
        for _ in range(3):
            t0 = time.perf_counter()
            hits = cb.intersect_tri(prims, rays_sample, 0, robust, threads=threads)
            secs.append(time.perf_counter() - t0)
        how = "std::thread ray chunks"
    dt = float(np.median(secs))
    parity = bool(hits.tobytes() == gpu_hits_sample.tobytes())
    _, cnt = cb.intersect_tri(prims, rays_sample[:1_000_000], 0, robust, threads=threads, counters=True)
    nc = min(n, 1_000_000)
    # CPU build of the same tree with the reference's DefaultBuilder (thread pool = all host threads unless --serial-builder)
    bb, cc = lib.prep_tris(tris)
    q = {"low": oracle.QUALITY_LOW, "medium": oracle.QUALITY_MEDIUM, "high": oracle.QUALITY_HIGH}[quality]
    builder = oracle.BUILDER_DEFAULT_SERIAL if serial else oracle.BUILDER_DEFAULT_PARALLEL
    lib.build(bb, cc, builder=builder, quality=q, threads=threads)              # warm-up
    bts = []
    for _ in range(3):
        t0 = time.perf_counter()
        cb2 = lib.build(bb, cc, builder=builder, quality=q, threads=threads)
        bts.append(time.perf_counter() - t0)
    bt = float(np.median(bts))
    same_tree = bool(cb2.serialize() == bvh.serialize())
    return {
        "value": round(n / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": kind,
        "sample": f"first {n} rays of rank 0's batch through the same BVH, {threads} host threads ({how}); "
                  f"median of 3 passes after a warm-up pass ({dt:.2f} s per pass)",
        "passes_s": [round(float(x), 3) for x in secs],
        "build_mtris_s": round(len(tris) / bt / 1e6, 3), "build_threads": 1 if serial else threads, "build_passes_s": [round(x, 3) for x in bts],
        "gpu_matches_cpu_hits": parity, "gpu_tree_equals_cpu_tree": same_tree,
        "P": round(float(cnt[0]) / nc, 3), "T": round(float(cnt[1]) / nc, 3),
    }


def pmc_traffic(args, robust, kernel_name, reordered):
    """HBM-side bytes per launch of the traced kernel from SEPARATE rocprofv3 --pmc passes of this same command (FETCH_SIZE
    and WRITE_SIZE cannot be collected inside a timed run): tools/pmc_traffic.py records them in profiles/pmc_traffic.json
    together with a sha1 of the traced kernel's instructions. Returns (record or None, note): the counts are only quoted for a
    library whose kernel is instruction-identical to the one that was traced."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_traffic.json"
    key = (f"{args.workload}|{args.quality}|{'serial' if args.serial_builder else 'pool'}|{'robust' if robust else 'fast'}|{args.rays}|"
           f"{'reordered' if reordered else 'as_given'}")
    rec = json.load(open(path)).get(key)
    if rec is None:
        return None, f"no --pmc pass recorded for {key}"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_isa import kernel_isa_hash
    from bvh_amd import _lib
    have = kernel_isa_hash(_lib.LIB_PATH, kernel_name)
    if rec.get("kernel") != kernel_name or have is None or have != rec.get("isa_sha1"):
        note = (f"profiles/pmc_traffic.json was traced on another build of {kernel_name} (isa sha1 {rec.get('isa_sha1')} vs loaded "
                f"{have}): traffic withheld, re-run tools/pmc_traffic.py")
        print("[bench] WARNING: " + note, file=sys.stderr)
        return None, note
    return rec, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel isa sha1 {have[:12]} (profiles/pmc_traffic.json)"


def record_walk_probe(working_set_bytes):
    """Rate (G records/s) at which the memory system serves a dependent walk over random 64-byte records of a table as large as
    the traversal's working set (at least 2x the L2s), one record in flight per lane: csrc/probe.hip. Measured live."""
    import ctypes as C
    import torch
    from bvh_amd import _lib
    n = int(max(working_set_bytes, 2 * L2_BYTES) // 64)
    perm = torch.randperm(n, device="cuda", dtype=torch.int64)
    table = torch.randint(0, 2 ** 31 - 1, (n, 16), dtype=torch.int32, device="cuda")
    table[perm, 0] = torch.roll(perm, -1).to(torch.int32)       # one cycle through all records in random order
    ms, recs = C.c_float(0), C.c_ulonglong(0)
    lib = _lib.load()
    _lib.check(lib.bvh_amd_probe_record_walk(table.data_ptr(), n, 256, 7, 3, C.byref(ms), C.byref(recs),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "probe_record_walk")
    return {"table_mib": round(n * 64 / 2 ** 20, 1), "records_per_launch": int(recs.value), "ms": round(ms.value, 4),
            "grec_s": round(recs.value / (ms.value * 1e-3) / 1e9, 2)}


def mean_split_ancestors(nodes, n_prims):
    """L-bar of SURVEY.md 8(d): mean number of inner (split) ancestors per primitive of the built tree."""
    idx = nodes["index"].astype(np.int64)
    cnt, first = idx & 15, idx >> 4
    frontier = np.array([0], dtype=np.int64)
    depth, total = 0, 0
    while len(frontier):
        c = cnt[frontier]
        total += int((c[c > 0] * depth).sum())
        kids = first[frontier[c == 0]]
        frontier = np.concatenate([kids, kids + 1])
        depth += 1
    return total / max(1, n_prims)


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs across processes on this driver
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: bvh_amd has no CPU path")
    # BVH_AMD_BENCH_ONE_DEVICE=1 + BVH_AMD_BENCH_BACKEND=gloo: functional test of the N>1 path on a 1-GPU box
    # (all ranks share cuda:0, collectives over gloo). The driver's runs use one GPU per rank and RCCL.
    one_device = os.environ.get("BVH_AMD_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("BVH_AMD_BENCH_BACKEND", "nccl")
    device_index = 0 if one_device else local_rank
    torch.cuda.set_device(device_index)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import bvh_amd
    from bvh_amd import synth
    from bvh_amd.parallel import broadcast_scene

    gen, n_tris, desc, label = WORKLOADS[args.workload]
    if args.obj:
        args.workload = "obj:" + os.path.basename(args.obj)
        gen, desc = None, f"{os.path.basename(args.obj)} (Wavefront OBJ, reference loader semantics: load_obj.cpp:57-96)"
    robust = not args.fast

    # ---- scene + build on rank 0, broadcast of the serialized BVH + BVH-ordered PrecomputedTri ------------
    tris = None
    build_ms = None
    if rank == 0:
        if args.obj:
            from bvh_amd.obj import load_obj
            tris = load_obj(args.obj)
            if len(tris) == 0:
                raise SystemExit(f"{args.obj}: no faces")
            n_tris = len(tris)
            label = f"{n_tris}-tri OBJ mesh"
        else:
            tris = getattr(synth, gen)(n_tris)
        d_tris = torch.from_numpy(tris).cuda()
        pool = None if args.serial_builder else bvh_amd.ThreadPool()
        builds = {}
        for qname in ("low", "medium", "high"):                               # build Mtris/s of every DefaultBuilder mode
            cfg = bvh_amd.Config(quality=bvh_amd.Quality[qname.capitalize()])
            bb, cc = bvh_amd.tri_bounds(d_tris)
            bvh_q = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=pool)   # warm-up build (allocations, code load)
            times, times_host = [], []
            for _ in range(5 if qname != "high" else 3):                          # median of >= 5 after a warm-up
                bvh_q = None                                                      # (destroying the previous BVH is not part of a build)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bb, cc = bvh_amd.tri_bounds(d_tris)
                bvh_q = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=pool)   # triangles in HBM -> BVH resident in HBM
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                bvh_q.sync_host()                                                 # + device-to-host copy of the reference-layout Bvh
                times.append(t1 - t0)
                times_host.append(time.perf_counter() - t0)
            # B_build = 36 (tri) + 36 (bbox + center) + 76 L-bar + 28 N/n + 4 algorithmic bytes per triangle
            lbar = mean_split_ancestors(bvh_q.nodes, n_tris)
            b_build = 36.0 + 36.0 + 76.0 * lbar + 28.0 * bvh_q.node_count / n_tris + 4.0
            builds[qname] = (sorted(times)[len(times) // 2] * 1e3, bvh_q, sorted(times_host)[len(times_host) // 2] * 1e3, lbar, b_build)
        build_ms, bvh, build_host_ms = builds[args.quality][:3]
        prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    else:
        bvh, prims = None, None
    bcast = {}
    if distributed:
        bvh, prims = broadcast_scene(bvh, prims, src=0, timing=bcast)     # one device-to-device broadcast of the serialized BVH + prims
    lo, hi = synth.scene_bounds(tris) if rank == 0 else (None, None)
    if distributed:
        box = torch.tensor(np.stack([lo, hi]) if rank == 0 else np.zeros((2, 3)), dtype=torch.float64,
                           device="cuda" if backend == "nccl" else "cpu")
        dist.broadcast(box, 0)
        lo, hi = box[0].cpu().numpy(), box[1].cpu().numpy()

    # ---- this rank's ray shard, resident in HBM ---------------------------------------------------------------
    rays_h = synth.rays_closest(args.rays, lo, hi, seed=1234 + rank)
    rays = torch.from_numpy(rays_h).cuda()
    hits = torch.empty((args.rays, 4), dtype=torch.float32, device="cuda")

    sort_rays = False if args.no_reorder else None            # None: the library decides (include/bvh_amd.h: BVH_AMD_RAY_SORTED)

    def step():
        bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=robust, out=hits, sort_rays=sort_rays)

    # traversal statistics of this batch (stats variant of the kernel; equal to the oracle's counters, tests/)
    _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=robust, counters=True, sort_rays=sort_rays)
    cnt = cnt.cpu().numpy()
    reordered = bool(bvh_amd._lib.load().bvh_amd_last_launch_reordered())
    P, T = cnt[0] / args.rays, cnt[1] / args.rays
    b_ray = 32.0 + 56.0 * P + 48.0 * T + 16.0            # ray + node pairs + triangles + hit record

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    lib = bvh_amd._lib.load()
    lib.bvh_amd_kernel_timing(1)                              # a pair of HIP events on the launch stream around the traversal kernel
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pass_ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]))     # whole pass: reordering + kernel
    import ctypes
    kt = (ctypes.c_float * 256)()
    got = ctypes.c_size_t(0)
    bvh_amd._lib.check(lib.bvh_amd_kernel_times(kt, min(args.steps, 256), ctypes.byref(got)), "kernel_times")
    lib.bvh_amd_kernel_timing(0)
    kernel_ms = float(np.mean(kt[:got.value])) if got.value else pass_ms                     # the traversal kernel alone

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_rays = args.rays * world * args.steps
        value = total_rays / elapsed / 1e6
        achieved = b_ray * args.rays / (kernel_ms * 1e-3) / 1e9
        kernel_name = bvh_amd._lib.load().bvh_amd_last_kernel_name().decode()
        rec, traffic_note = pmc_traffic(args, robust, kernel_name, reordered)
        traffic = None if rec is None else round((rec["fetch_kb"] + rec["write_kb"]) * 1024.0 / (kernel_ms * 1e-3) / 1e9, 1)
        # The ceiling that binds this kernel (profiles/r02_traversal_experiments.md): its L2 misses against the rate at which the
        # memory system serves a dependent walk over random 64-byte records, measured live by csrc/probe.hip.
        working_set = bvh.node_count // 2 * 64 + n_tris * 48
        probe = None if args.no_probe else record_walk_probe(working_set)
        miss_rate = None if rec is None else rec["fetch_kb"] * 1024.0 / 64.0 / (kernel_ms * 1e-3) / 1e9     # G 64-byte sectors/s
        binding = {"bound": "l2_miss_path", "unit": "G 64-byte sectors/s",
                   "achieved": None if miss_rate is None else round(miss_rate, 2), "peak": None if probe is None else probe["grec_s"],
                   "frac": None if (miss_rate is None or probe is None) else round(miss_rate / probe["grec_s"], 4),
                   "what": "L2 misses of the kernel (FETCH_SIZE / 64 B per launch / kernel time) over the record rate of a dependent random "
                           "walk through a table of the working set's size, one 64-byte record in flight per lane (bvh_amd_probe_record_walk)",
                   "probe": probe}
        if binding["frac"] is not None:
            binding["reading"] = ("this ceiling binds the kernel: only fewer L2 misses per ray can make it faster" if binding["frac"] >= 0.85 else
                                  "the walk is off the miss path (rays picked up in a coherent order, one stretch of it per XCD): waves now wait for "
                                  "the slowest lane of a step; SQ / TCP / TCC counters in profiles/r02_trace_soup1m_sorted_pmc_sq.csv")
        out = {
            "metric": f"Mrays/s closest-hit ({label})", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; {'robust' if robust else 'fast'} traversal, DefaultBuilder "
                                   f"{'serial' if args.serial_builder else 'with thread pool (mini-trees)'} Quality::{args.quality.capitalize()} "
                                   f"built on the GPU (the reference's default configuration is thread pool + High)"
                                   + ("; rays reordered for coherence inside the timed pass (library default for this tree size)" if reordered else ""),
                       "tris": int(n_tris), "nodes": int(bvh.node_count), "rays_per_gpu_per_step": int(args.rays),
                       "parallelism": f"rays sharded x{world}, BVH broadcast over RCCL" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_frac": None if traffic is None else round(traffic / HBM_PEAK_GBS, 4), "traffic_source": traffic_note,
                         "achieved_is": "algorithmic bytes (SURVEY.md 8d: 32 + 56 P + 48 T + 16 per ray) / kernel time; `traffic` is what the "
                                        "L2's fabric side actually moved. frac > 1 means L1 / L2 serve re-referenced nodes faster than HBM could "
                                        "stream the algorithmic bytes",
                         "kernel": kernel_name,
                         "kernel_ms": round(kernel_ms, 4), "pass_ms": round(pass_ms, 4),
                         "ray_reordering": ("on: 21-bit origin-cell/octant key + three radix passes inside every timed pass, "
                                            f"{round(pass_ms - kernel_ms, 3)} ms of it" if reordered else "off"),
                         "bytes_per_ray": round(b_ray, 1),
                         "P_node_pairs_per_ray": round(float(P), 3), "T_prim_tests_per_ray": round(float(T), 3)},
            "roofline_binding": binding,
            "build": {"mtris_s": round(n_tris / (build_ms * 1e-3) / 1e6, 2), "ms": round(build_ms, 3),
                      "roofline": {q: {"bound": "hbm", "achieved": round(v[4] * n_tris / (v[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(v[4] * n_tris / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "bytes_per_tri": round(v[4], 1),
                                       "mean_split_ancestors": round(v[3], 2)} for q, v in builds.items()},
                      "all_qualities_ms": {k: round(v[0], 3) for k, v in builds.items()},
                      "all_qualities_mtris_s": {k: round(n_tris / (v[0] * 1e-3) / 1e6, 2) for k, v in builds.items()},
                      "ms_with_host_mirror": round(build_host_ms, 3),
                      "mtris_s_with_host_mirror": round(n_tris / (build_host_ms * 1e-3) / 1e6, 2),
                      "what": "tri bounds + DefaultBuilder, triangles resident in HBM -> BVH resident in HBM; *_with_host_mirror adds "
                              "the device-to-host copy of the reference-layout Bvh; median of 5 (High: 3) after a warm-up build"},
        }
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample, args.rays)
            out["cpu_baseline"] = cpu_baseline(tris, bvh, rays_h[:ns], int(robust), bvh_amd.hits_to_numpy(hits[:ns]),
                                               args.quality, args.serial_builder)
        else:
            out["cpu_baseline"] = None
        if world > 1:
            out["broadcast"] = {"ms": round(bcast.get("broadcast_ms", 0.0), 3), "payload_bytes": int(bcast.get("payload_bytes", 0)),
                                "what": "Bvh::serialize stream + BVH-ordered PrecomputedTri, device buffers, one root-to-all broadcast each (outside the timed steps)"}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
