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


def broadcast_scene(bvh, prims, src: int = 0):
    """Rank `src` holds (Bvh, BVH-ordered primitive tensor); every rank returns its own device-resident copy."""
    import torch
    import torch.distributed as dist
    from .api import Bvh
    rank = dist.get_rank()
    flag = torch.tensor([0 if (rank != src or bvh.dtype == np.float32) else 1], dtype=torch.int64,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(flag, src)
    dtype = np.float64 if int(flag.item()) else np.float32
    stream = broadcast_bytes(bvh.serialize() if rank == src else None, src)
    if rank != src:
        bvh = Bvh.deserialize(stream, dtype=dtype)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    if dist.get_backend() == "nccl":
        prims = broadcast_tensor(prims, src=src, dtype=tdt)
    else:                                                     # CPU collectives (tests): stage through host memory
        prims = broadcast_tensor(prims.cpu() if rank == src else None, src=src, dtype=tdt, device="cpu").cuda()
    return bvh, prims


def intersect_sharded(bvh, prims, rays, **kw):
    """Traces this rank's contiguous shard of a replicated ray array; returns (begin, end, hits)."""
    import torch.distributed as dist
    from .api import intersect
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    b, e = shard_range(len(rays), rank, world)
    return b, e, intersect(bvh, prims, rays[b:e], **kw)
