#!/usr/bin/env python3
"""bench.py — Mrays/s closest-hit on a ~1M-triangle scene, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload soup_1m|sponza_262k|terrain_1m|soup_10m]
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
    # name: (generator, n_tris, description)
    "soup_1m": ("soup", 1_000_000, "1,000,000-triangle random soup (M3), worst-case incoherent"),
    "terrain_1m": ("terrain", 1_000_000, "~1M-triangle height field (M2), tie-heavy"),
    "sponza_262k": ("sponza_proxy", 262_144, "262,144-triangle Sponza proxy (M1) — BASELINE configs[1]"),
    "soup_10m": ("soup", 10_000_000, "10,000,000-triangle random soup (M3-10M) — BASELINE configs[3]; use --rays 12500000 for its 100M / 8 shards"),
}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="rays of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(tris, bvh, rays_sample, robust, gpu_hits_sample, quality, serial):
    """The reference's CPU path (oracle/_ref when present, else the restatement) on a bounded sample of the same
    workload, all host threads; also a parity spot-check of the GPU result. Test infrastructure only."""
    import oracle
    lib = oracle.load_ref()
    kind = "reference"
    if lib is None:
        lib = oracle.load_oracle()
        kind = "port"
    threads = lib.hardware_threads()
    cb = lib.from_arrays(bvh.nodes, bvh.prim_ids)
    prims = lib.precompute_tris(tris, bvh.prim_ids)
    cb.intersect_tri(prims, rays_sample[:65536], 0, robust, threads=threads)          # warm-up
    t0 = time.perf_counter()
    hits, cnt = cb.intersect_tri(prims, rays_sample, 0, robust, threads=threads, counters=True)
    dt = time.perf_counter() - t0
    parity = bool(hits.tobytes() == gpu_hits_sample.tobytes())
    # CPU build of the same tree with the reference's DefaultBuilder (thread pool = all host threads unless --serial-builder)
    bb, cc = lib.prep_tris(tris)
    q = {"low": oracle.QUALITY_LOW, "medium": oracle.QUALITY_MEDIUM, "high": oracle.QUALITY_HIGH}[quality]
    builder = oracle.BUILDER_DEFAULT_SERIAL if serial else oracle.BUILDER_DEFAULT_PARALLEL
    lib.build(bb, cc, builder=builder, quality=q, threads=threads)              # warm-up
    t0 = time.perf_counter()
    cb2 = lib.build(bb, cc, builder=builder, quality=q, threads=threads)
    bt = time.perf_counter() - t0
    same_tree = bool(cb2.serialize() == bvh.serialize())
    return {
        "value": round(len(rays_sample) / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": kind,
        "sample": f"first {len(rays_sample)} rays of rank 0's batch, same BVH, {threads} host threads "
                  f"(std::thread ray chunks around Bvh::intersect)",
        "build_mtris_s": round(len(tris) / bt / 1e6, 3), "build_threads": 1 if serial else threads,
        "gpu_matches_cpu_hits": parity, "gpu_tree_equals_cpu_tree": same_tree,
        "P": round(float(cnt[0]) / len(rays_sample), 3), "T": round(float(cnt[1]) / len(rays_sample), 3),
    }


def pmc_traffic_gbs(args, robust, kernel_ms):
    """roofline.traffic: HBM-side bytes per launch of the traced kernel from the SEPARATE rocprofv3 --pmc passes of this
    same command (FETCH_SIZE and WRITE_SIZE cannot be collected inside a timed run), committed as
    profiles/pmc_traffic.json, divided by the live kernel time -> GB/s like `achieved`. None when no pass was recorded
    for this exact workload / quality / ray count."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    key = f"{args.workload}|{args.quality}|{'serial' if args.serial_builder else 'pool'}|{'robust' if robust else 'fast'}|{args.rays}"
    rec = json.load(open(path)).get(key)
    if rec is None:
        return None
    return round((rec["fetch_kb"] + rec["write_kb"]) * 1024.0 / (kernel_ms * 1e-3) / 1e9, 1)


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

    gen, n_tris, desc = WORKLOADS[args.workload]
    robust = not args.fast

    # ---- scene + build on rank 0, broadcast of the serialized BVH + BVH-ordered PrecomputedTri ------------
    tris = None
    build_ms = None
    if rank == 0:
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
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bb, cc = bvh_amd.tri_bounds(d_tris)
                bvh_q = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=pool)   # triangles in HBM -> BVH resident in HBM
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                bvh_q.sync_host()                                                 # + device-to-host copy of the reference-layout Bvh
                times.append(t1 - t0)
                times_host.append(time.perf_counter() - t0)
            builds[qname] = (sorted(times)[len(times) // 2] * 1e3, bvh_q, sorted(times_host)[len(times_host) // 2] * 1e3)
        build_ms, bvh, build_host_ms = builds[args.quality]
        prims = bvh_amd.precompute_tris(d_tris, bvh.device_prim_ids())
    else:
        bvh, prims = None, None
    if distributed:
        bvh, prims = broadcast_scene(bvh, prims, src=0)
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

    def step():
        bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=robust, out=hits)

    # traversal statistics of this batch (stats variant of the kernel; equal to the oracle's counters, tests/)
    _, cnt = bvh_amd.intersect(bvh, prims, rays, any_hit=False, robust=robust, counters=True)
    cnt = cnt.cpu().numpy()
    P, T = cnt[0] / args.rays, cnt[1] / args.rays
    b_ray = 32.0 + 56.0 * P + 48.0 * T + 16.0            # ray + node pairs + triangles + hit record

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
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
    kernel_ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]))   # HIP events, launch stream

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_rays = args.rays * world * args.steps
        value = total_rays / elapsed / 1e6
        achieved = b_ray * args.rays / (kernel_ms * 1e-3) / 1e9
        traffic = pmc_traffic_gbs(args, robust, kernel_ms)
        out = {
            "metric": "Mrays/s closest-hit (1M-tri scene)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; {'robust' if robust else 'fast'} traversal, DefaultBuilder "
                                   f"{'serial' if args.serial_builder else 'with thread pool (mini-trees)'} Quality::{args.quality.capitalize()} "
                                   f"built on the GPU (the reference's default configuration is thread pool + High)",
                       "tris": int(n_tris), "nodes": int(bvh.node_count), "rays_per_gpu_per_step": int(args.rays),
                       "parallelism": f"rays sharded x{world}, BVH broadcast over RCCL" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": bvh_amd._lib.load().bvh_amd_last_kernel_name().decode(),
                         "kernel_ms": round(kernel_ms, 4), "bytes_per_ray": round(b_ray, 1),
                         "P_node_pairs_per_ray": round(float(P), 3), "T_prim_tests_per_ray": round(float(T), 3)},
            "build": {"mtris_s": round(n_tris / (build_ms * 1e-3) / 1e6, 2), "ms": round(build_ms, 3),
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
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
