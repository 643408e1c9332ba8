"""CPU: bench.py's control flow and its one-line JSON contract, run end to end against a FAKE device.

bench.py needs an MI355X; here torch.cuda and the bvh_amd entry points it calls are replaced by stand-ins backed by the oracle
(test infrastructure), on a tiny scene, so that a typo in the timed loop, the build table, the roofline / cpu_baseline objects or
the JSON assembly shows up without a GPU. Nothing about performance is tested, and none of this is a product path."""
import io
import json
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import pytest

import oracle


class _FakeEvent:
    _clock = 0.0

    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self):
        _FakeEvent._clock += 1.0
        self.t = _FakeEvent._clock

    def elapsed_time(self, other):
        return other.t - self.t


class _FakeBvh:
    def __init__(self, cb):
        self.cb = cb
        self.synced = False

    node_count = property(lambda self: self.cb.node_count)
    nodes = property(lambda self: self.cb.nodes())
    prim_ids = property(lambda self: self.cb.prim_ids())

    def serialize(self):
        return self.cb.serialize()

    def sync_host(self):
        self.synced = True

    def device_prim_ids(self):
        return self.cb.prim_ids()


def test_bench_json_contract(monkeypatch, orc):
    import torch
    import bench
    import bvh_amd
    from bvh_amd import synth

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *_: None)
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda *_: types.SimpleNamespace(name="fake MI355X", uuid="GPU-0"))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{x: y for x, y in k.items() if x != "device"}))
    real_tensor = torch.tensor
    monkeypatch.setattr(torch, "tensor", lambda *a, **k: real_tensor(*a, **{x: y for x, y in k.items() if x != "device"}))

    state = {}

    def tri_bounds(t):
        state["tris"] = np.ascontiguousarray(t.numpy() if hasattr(t, "numpy") else t)
        return orc.prep_tris(state["tris"])

    def build(bb, cc, cfg, thread_pool=None):
        return _FakeBvh(orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL if thread_pool is not None else oracle.BUILDER_DEFAULT_SERIAL,
                                  quality=int(cfg.quality), threads=2))

    def precompute_tris(t, ids):
        return orc.precompute_tris(state["tris"], np.asarray(ids))

    def intersect(bvh, prims, rays, any_hit=False, robust=False, counters=False, out=None, **kw):
        r = np.ascontiguousarray(rays.numpy() if hasattr(rays, "numpy") else rays)
        hits, cnt = bvh.cb.intersect_tri(prims, r, any_hit, robust, threads=2, counters=True)
        h = torch.from_numpy(hits.view(np.float32).reshape(-1, 4).copy())
        if out is not None:
            out.copy_(h)
            h = out
        return (h, torch.from_numpy(cnt.astype(np.int64))) if counters else h

    monkeypatch.setattr(bvh_amd, "tri_bounds", tri_bounds)
    monkeypatch.setattr(bvh_amd.DefaultBuilder, "build", staticmethod(build))
    monkeypatch.setattr(bvh_amd, "precompute_tris", precompute_tris)
    monkeypatch.setattr(bvh_amd, "intersect", intersect)
    monkeypatch.setattr(bvh_amd, "prepare_trace", lambda bvh, n_rays_hint=0: None)
    def fake_kernel_times(ms_out, capacity, count_out):
        for i in range(capacity):
            ms_out[i] = 1.0
        count_out._obj.value = capacity
        return 0
    def fake_reorder_times(ms_out, capacity, count_out):
        for i in range(capacity):
            ms_out[i] = 0.25
        count_out._obj.value = capacity
        return 0

    def fake_plan(out):
        out[0], out[1], out[2], out[3] = 1, 1, 12, 12
    fake_lib = types.SimpleNamespace(bvh_amd_last_kernel_name=lambda: b"trace_kernel_coop<float, false, true, 0, false>", bvh_amd_kernel_timing=lambda on: None,
                                     bvh_amd_last_launch_reordered=lambda: 0, bvh_amd_kernel_times=fake_kernel_times, bvh_amd_reorder_times=fake_reorder_times,
                                     bvh_amd_last_launch_plan=fake_plan, bvh_amd_tuning=lambda *a: None, bvh_amd_experiment=lambda *a: 0)
    monkeypatch.setattr(bvh_amd, "last_optimize_profile", lambda: {"iterations": 3, "replayed": 2, "replacements": 1000, "heap_ms": 0.5})
    monkeypatch.setattr(bvh_amd._lib, "load", lambda: fake_lib)
    monkeypatch.setitem(bench.WORKLOADS, "soup_1m", ("soup", 3000, "tiny stand-in scene of the contract test", "3000-tri stand-in"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "1", "--rays", "4096", "--cpu-sample", "2048", "--no-probe", "--no-pmc"])
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)

    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "build"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "Mrays/s" and out["scaling"] == "weak"
    assert out["dtype"] == "f32" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "workload" in out["config"] and "model" not in out["config"]
    assert len(out["config"]["ranks"]) == out["n_gpus"] and out["config"]["ranks"][0]["device"] == 0 and out["config"]["launched_by"] == "single process"
    rf = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "levels", "hbm_algorithmic", "pass_split_ms", "launch_plan"):
        assert key in rf, key
    # the ceiling that binds is the memory hierarchy under dependent record fetches; SURVEY.md 8(d)'s HBM figure is kept beside it
    assert rf["unit"] == "Mrays/s" and rf["achieved"] > 0 and rf["peak"] is None and rf["frac"] is None      # --no-probe: no ceiling is invented
    hb = rf["hbm_algorithmic"]
    assert hb["bound"] == "hbm" and hb["peak"] == 8000.0 and abs(hb["frac"] - hb["achieved"] / hb["peak"]) < 1e-3
    # algorithmic bytes per ray from the batch's own counters
    assert abs(hb["bytes_per_ray"] - (32 + 56 * rf["P_node_pairs_per_ray"] + 48 * rf["T_prim_tests_per_ray"] + 16)) < 0.5
    assert rf["pass_split_ms"]["ray_keys_and_radix_sort"] == 0.25 and rf["pass_split_ms"]["traversal_kernel"] == 1.0
    assert rf["launch_plan"]["quad_cooperative_fetch"] is True and rf["record_fetch"] == "quad-cooperative"
    cb = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "usable_cpus", "threads_used", "mrays_s_per_thread", "hardware_concurrency"):
        assert key in cb, key
    assert cb["threads_used"] <= cb["usable_cpus"] <= cb["affinity_cpus"] and cb["cores"] == cb["threads_used"]
    assert cb["gpu_matches_cpu_hits"] is True and cb["gpu_tree_equals_cpu_tree"] is True
    assert out["metric"] == "Mrays/s closest-hit (3000-tri stand-in)"           # the label follows the workload
    assert "via OBJ" in out["data"]                                             # the mesh went through the OBJ writer and the reference-semantics loader
    assert rf["traffic"] is None and "counters_source" in rf                    # no --pmc pass is recorded for this workload: never a stale number
    b = out["build"]
    for q in ("low", "medium", "high"):
        r = b["roofline"][q]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 1.0 < r["mean_split_ancestors"] < 40.0
        assert abs(r["bytes_per_tri"] - (76.0 + 76.0 * r["mean_split_ancestors"] + 28.0 * out["config"]["nodes"] / 3000)) < 60.0
    assert set(b["all_qualities_ms"]) == {"low", "medium", "high"} and b["ms_with_host_mirror"] >= b["ms"] > 0
    assert b["high"]["iterations"] == 3 and b["high"]["replayed"] == 2 and abs(b["high"]["us_per_replacement"] - 0.5) < 1e-6


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun (the shape of the driver's N = 1 command with another N) starts two ranks itself
    and rank 0 reports both; a --gpus that disagrees with the launcher's WORLD_SIZE is refused instead of being printed."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BVH_AMD_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["self_launched"] is True and sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert len({x["pid"] for x in out["ranks"]}) == 2
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_stdout_carries_the_line_and_nothing_else():
    """bench.py hands file descriptor 1 to stderr for the run (native libraries print there: gloo announces its peers) and writes the
    JSON line to what stdout was: whatever else is printed, through Python or straight to the descriptor, ends up on stderr."""
    import subprocess
    from conftest import ROOT
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.own_stdout(); os.write(1, b'native noise\\n'); "
            "print('python noise'); bench.emit_line({'metric': 'x', 'value': 1})") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": "x", "value": 1}\n'
    assert "native noise" in r.stderr and "python noise" in r.stderr
