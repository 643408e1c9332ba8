"""-m gpu: the reference's `Bvh::serialize` byte stream (bvh.h:221-243, node.h:90-102) as a DEVICE path, and the multi-GPU
exchange built on it: `bvhXX_serialize_device` must write exactly the bytes the oracle's
`Bvh::serialize` gives, `bvhXX_deserialize_device` must rebuild a BVH whose stream and hits are identical, malformed streams
are refused, and a 2-rank run (torch.distributed, both ranks on this one GPU, gloo) must return — shard by shard — the hits
of the single-GPU trace of the same global ray array."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from bvh_amd import synth
from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("parallel", [False, True])
def test_device_stream_equals_oracle_stream_and_round_trips(orc, dtype, parallel):
    import torch
    import bvh_amd
    tris = synth.soup(40_000, seed=5, jitter=0.01, dtype=dtype)
    bb, cc = orc.prep_tris(tris)
    ref = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_PARALLEL if parallel else oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_HIGH)
    gpu = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool() if parallel else None)
    d_stream = gpu.serialize_device()                           # written from the resident nodes; the host mirror is still empty
    assert d_stream.is_cuda and d_stream.cpu().numpy().tobytes() == ref.serialize()
    back = bvh_amd.Bvh.deserialize_device(d_stream, dtype=dtype)
    del d_stream
    assert back.node_count == gpu.node_count and back.prim_count == gpu.prim_count
    assert back.serialize() == ref.serialize()                  # host accessor path of the rebuilt BVH
    assert back.serialize_device().cpu().numpy().tobytes() == ref.serialize()
    # the rebuilt BVH traces like the original
    prims = bvh_amd.precompute_tris(tris, gpu.device_prim_ids())
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(100_000, lo, hi, dtype=dtype)
    a, ca = bvh_amd.intersect(gpu, prims, rays, robust=True, counters=True)
    b, cb = bvh_amd.intersect(back, prims, rays, robust=True, counters=True)
    assert bvh_amd.hits_to_numpy(a).tobytes() == bvh_amd.hits_to_numpy(b).tobytes() and bool((ca == cb).all())
    oprims = orc.precompute_tris(tris, ref.prim_ids())
    assert bvh_amd.hits_to_numpy(b).tobytes() == ref.intersect_tri(oprims, rays, False, True, threads=4).tobytes()
    # after optimize / refit on the rebuilt BVH the device stream follows
    back.refit()
    assert back.serialize_device().cpu().numpy().tobytes() == ref.serialize()


@pytest.mark.parametrize("scene", ["circles2k_2f", "circles2k_2d"])
def test_device_stream_2d(scene):
    import torch
    import bvh_amd
    g = load_golden(scene)
    dt = g["bboxes"].dtype
    key = [k for k in g.files if k.startswith("bvh_serial")][0]
    want = g[key].tobytes()
    buf = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
    bvh = bvh_amd.Bvh.deserialize_device(buf, dtype=dt, dim=2)
    assert bvh.dim == 2 and bvh.serialize() == want
    assert bvh.serialize_device().cpu().numpy().tobytes() == want


def test_malformed_device_streams_are_refused(orc):
    import torch
    import bvh_amd
    tris = synth.soup(500, seed=9)
    bb, cc = orc.prep_tris(tris)
    stream = bytearray(orc.build(bb, cc, quality=oracle.QUALITY_LOW).serialize())
    good = torch.from_numpy(np.frombuffer(bytes(stream), dtype=np.uint8).copy()).cuda()
    assert bvh_amd.Bvh.deserialize_device(good).serialize() == bytes(stream)
    nodes = np.frombuffer(bytes(stream), dtype=oracle.NODEF, count=int(np.frombuffer(bytes(stream), "<u4", 1)[0]), offset=8).copy()
    inner = int(np.flatnonzero((nodes["index"] & 15) == 0)[1])
    leaf = int(np.flatnonzero((nodes["index"] & 15) != 0)[0])

    def with_node(i, index):
        s = bytearray(stream)
        s[8 + 28 * i + 24: 8 + 28 * i + 28] = np.uint32(index).tobytes()
        return torch.from_numpy(np.frombuffer(bytes(s), dtype=np.uint8).copy()).cuda()

    for bad, what in ((with_node(inner, 2 << 4), "odd index"),                          # even first_id: pairs would be misaligned
                      (with_node(inner, (len(nodes) + 1) << 4), "odd index"),            # children beyond the array
                      (with_node(leaf, (600 << 4) | 3), "primitive range"),              # leaf range beyond prim_ids
                      (good[: len(stream) - 4].contiguous(), "truncated"),
                      (good[:4].contiguous(), "truncated")):
        with pytest.raises(bvh_amd.BvhAmdError, match=what):
            bvh_amd.Bvh.deserialize_device(bad)
    # the host entry points apply the same structural check (they upload through the same validation)
    s = bytearray(stream)
    s[8 + 28 * inner + 24: 8 + 28 * inner + 28] = np.uint32(2 << 4).tobytes()
    with pytest.raises(bvh_amd.BvhAmdError, match="odd index"):
        bvh_amd.Bvh.deserialize(bytes(s))
    hdr = bytearray(stream[:8])
    hdr[0:4] = np.uint32(0xFFFFFFFF).tobytes()
    with pytest.raises(bvh_amd.BvhAmdError, match="truncated"):
        bvh_amd.Bvh.deserialize(bytes(hdr) + bytes(stream[8:]))


WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)                                    # both ranks share the box's one GPU: functional test, not a scaling run
import bvh_amd
from bvh_amd import synth
from bvh_amd.parallel import broadcast_scene, intersect_sharded, broadcast_bytes
tris = synth.soup(60_000, seed=3, jitter=0.01)              # every rank can regenerate the scene, only rank 0 builds
lo, hi = synth.scene_bounds(tris)
rays = torch.from_numpy(synth.rays_closest(300_001, lo, hi)).cuda()     # the same GLOBAL ray array on every rank (odd size: ragged last shard)
bvh = prims = None
if rank == 0:
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
timing = {{}}
bvh, prims = broadcast_scene(bvh, prims, src=0, timing=timing)
assert timing["payload_bytes"] > 60_000 * 48
want = broadcast_bytes(bvh.serialize() if rank == 0 else None, 0)      # rank 0's stream through the host path, as the yardstick
assert bvh.serialize() == want, "the broadcast BVH differs from the builder's"
b, e, hits = intersect_sharded(bvh, prims, rays, robust=True)
np.save(os.path.join({tmp!r}, f"hits{{rank}}.npy"), hits.cpu().numpy())
np.save(os.path.join({tmp!r}, f"range{{rank}}.npy"), np.array([b, e]))
dist.barrier()
if rank == 0:
    whole = bvh_amd.intersect(bvh, prims, rays, robust=True).cpu().numpy()
    got = np.concatenate([np.load(os.path.join({tmp!r}, f"hits{{r}}.npy")) for r in range(world)])
    ranges = [np.load(os.path.join({tmp!r}, f"range{{r}}.npy")) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == len(rays) and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert got.shape == whole.shape and got.tobytes() == whole.tobytes(), "sharded hits differ from the single-GPU hits"
    assert (whole.view(np.int32)[:, 0] != -1).sum() > 1000
dist.barrier(); dist.destroy_process_group()
open(os.path.join({tmp!r}, f"rank{{rank}}.ok"), "w").write("ok")
'''


def test_two_ranks_sharded_hits_equal_single_gpu_hits(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, tmp=str(tmp_path)))
    for attempt in range(2):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
