"""Multi-GPU: ray batches shard embarrassingly; the only exchange is one broadcast of the scene.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests).
The BVH travels in the reference's own byte format (Bvh::serialize, reference bvh.h:221-229) followed by the
BVH-ordered primitive array; a broadcast from the building rank is bound by one xGMI link per peer with all
peers served in parallel, so no ring collective is used anywhere. Build: replicas only.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous ray range [begin, end) of `rank`: k * ceil(n / world) ...."""
    per = -(-n // world)
    b = min(n, rank * per)
    return b, min(n, b + per)


def broadcast_bytes(data, src: int = 0, device=None) -> bytes:
    """Broadcasts a byte string from `src` to every rank (two collectives: length, payload)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src)
    return data if rank == src else buf.cpu().numpy().tobytes()


def _coll_device():
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def broadcast_tensor(t, shape_hint=None, src: int = 0, dtype=None, device=None):
    """Broadcasts a tensor whose shape only `src` knows."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    meta = torch.zeros(4, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0] = t.dim()
        for i, s in enumerate(t.shape):
            meta[1 + i] = s
    dist.broadcast(meta, src)
    if rank != src:
        shape = [int(meta[1 + i].item()) for i in range(int(meta[0].item()))]
        t = torch.empty(shape, dtype=dtype, device=dev)
    dist.broadcast(t, src)
    return t


def broadcast_scene(bvh, prims, src: int = 0, timing: dict = None):
    """Rank `src` holds (Bvh, BVH-ordered primitive tensor); every rank returns its own device-resident copy.

    The payload is the reference's `Bvh::serialize` byte stream (bvh.h:221-229) written into HBM by `Bvh.serialize_device`, one
    `torch.distributed.broadcast` of that device buffer (backend nccl = RCCL over xGMI: root-to-all, no ring), and
    `Bvh.deserialize_device` on the receivers; then the primitive array the same way. With RCCL no payload byte touches the
    host on any rank. (gloo, CPU tests only: gloo cannot move device memory, so the same device buffers are staged through
    host tensors around the collective.) `timing`, if given, receives {"broadcast_ms", "payload_bytes"} (wall time of the whole
    exchange on this rank, device-synchronised)."""
    import time
    import torch
    import torch.distributed as dist
    from .api import Bvh
    rank = dist.get_rank()
    on_device = dist.get_backend() == "nccl"
    coll = "cuda" if on_device else "cpu"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    meta = torch.zeros(5, dtype=torch.int64, device=coll)      # stream bytes, is_double, dim, prims rows, prims cols
    buf = None
    if rank == src:
        buf = bvh.serialize_device()
        meta[0], meta[1], meta[2] = buf.numel(), int(bvh.dtype == np.float64), bvh.dim
        meta[3], meta[4] = prims.shape[0], prims.shape[1]
    dist.broadcast(meta, src)
    m = [int(v) for v in meta.tolist()]
    dtype = np.float64 if m[1] else np.float32
    tdt = torch.float64 if m[1] else torch.float32
    if on_device:
        if rank != src:
            buf = torch.empty(m[0], dtype=torch.uint8, device="cuda")
            prims = torch.empty((m[3], m[4]), dtype=tdt, device="cuda")
        dist.broadcast(buf, src)
        dist.broadcast(prims, src)
    else:
        hbuf = buf.cpu() if rank == src else torch.empty(m[0], dtype=torch.uint8)
        hprims = prims.cpu() if rank == src else torch.empty((m[3], m[4]), dtype=tdt)
        dist.broadcast(hbuf, src)
        dist.broadcast(hprims, src)
        if rank != src:
            buf, prims = hbuf.cuda(), hprims.cuda()
    if rank != src:
        bvh = Bvh.deserialize_device(buf, dtype=dtype, dim=m[2])
    torch.cuda.synchronize()
    if timing is not None:
        timing["broadcast_ms"] = (time.perf_counter() - t0) * 1e3
        timing["payload_bytes"] = m[0] + m[3] * m[4] * (8 if m[1] else 4)
    return bvh, prims


def intersect_sharded(bvh, prims, rays, **kw):
    """Traces this rank's contiguous shard of a replicated ray array; returns (begin, end, hits)."""
    import torch.distributed as dist
    from .api import intersect
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    b, e = shard_range(len(rays), rank, world)
    return b, e, intersect(bvh, prims, rays[b:e], **kw)
