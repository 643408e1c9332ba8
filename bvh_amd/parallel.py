"""Multi-GPU: ray batches shard embarrassingly; the only exchange is one broadcast of the scene.

One process per GPU. The exchange itself runs INSIDE libbvh_amd.so (csrc/replicate.hip: `bvh_amd_comm_*`, `bvhXX_broadcast`):
the building rank writes the reference's byte stream (Bvh::serialize, reference bvh.h:221-229) into HBM straight from its
resident nodes, RCCL broadcasts that device buffer and the BVH-ordered primitive array root-to-all over xGMI, every other rank
turns the received buffer into resident nodes + traversal records on its device — no payload byte visits a host.
`torch.distributed` is plumbing only: it carries the 128-byte RCCL unique id to the ranks (any backend) and provides the
barrier of bench.py. Build: replicas only.

Where RCCL cannot be used — the CPU test-suite and the one-GPU functional runs, where several ranks share a device and
ncclCommInitRank refuses — the same device buffers are staged through host tensors around a `torch.distributed` (gloo)
broadcast: `transport="staged"`. The transport actually used is reported in `timing["transport"]`.
"""
from __future__ import annotations

import ctypes as C

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


class Comm:
    """An RCCL communicator owned by libbvh_amd.so (include/bvh_amd.h: bvh_amd_comm_*), bound to this process's current GPU."""

    def __init__(self, handle, owner: bool = True):
        from . import _lib
        if not handle:
            raise _lib.BvhAmdError(_lib.last_error())
        self._h, self._lib, self._owner = handle, _lib.load(), owner

    def __del__(self):
        try:                                                  # (may run during interpreter shutdown, after the library object is gone)
            if getattr(self, "_h", None) and self._owner and self._lib is not None:
                self._lib.bvh_amd_comm_destroy(self._h)
        except Exception:                                     # noqa: BLE001
            pass
        self._h = None

    @property
    def rank(self) -> int:
        return self._lib.bvh_amd_comm_rank(self._h)

    @property
    def size(self) -> int:
        return self._lib.bvh_amd_comm_size(self._h)

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().bvh_amd_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    @staticmethod
    def create(unique_id: bytes, n_ranks: int, rank: int) -> "Comm":
        from . import _lib
        assert len(unique_id) == 128
        return Comm(_lib.load().bvh_amd_comm_create(unique_id, n_ranks, rank))

    @staticmethod
    def from_torch_distributed() -> "Comm":
        """Every rank of the default process group calls this: rank 0 draws the RCCL unique id, torch.distributed carries its 128
        bytes (whatever the backend), every rank joins with ncclCommInitRank on its current GPU."""
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        box = [Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return Comm.create(box[0], world, rank)

    def broadcast_(self, tensor, root: int = 0):
        """ncclBroadcast of a contiguous CUDA tensor, in place, on the current stream."""
        from . import _lib
        from .api import _stream
        assert tensor.is_cuda and tensor.is_contiguous()
        _lib.check(self._lib.bvh_amd_comm_broadcast(self._h, tensor.data_ptr(), tensor.numel() * tensor.element_size(), root, _stream()),
                   "comm_broadcast")
        return tensor


_default_comm = None


def default_comm() -> Comm:
    """The process-wide communicator over the default torch.distributed group (created by all ranks on first use)."""
    global _default_comm
    if _default_comm is None:
        _default_comm = Comm.from_torch_distributed()
    return _default_comm


def release_default_comm():
    """Destroys the process-wide communicator (ncclCommDestroy); every rank calls it, before torch's process group goes away."""
    global _default_comm
    c, _default_comm = _default_comm, None
    if c is not None:
        c.__del__()


def _pick_transport(transport):
    import torch.distributed as dist
    if transport is None:
        transport = "rccl" if (not dist.is_initialized() or dist.get_backend() == "nccl") else "staged"
    if transport not in ("rccl", "staged", "torch"):
        raise ValueError("transport must be 'rccl', 'torch' or 'staged'")
    return transport


def broadcast_scene(bvh, prims, src: int = 0, timing: dict = None, comm: Comm = None, transport: str = None):
    """Rank `src` holds (Bvh, BVH-ordered primitive tensor); every rank returns its own device-resident copy.

    transport "rccl" (default whenever the process group's backend is nccl, or no group exists): three RCCL broadcasts issued by
    libbvh_amd.so on the current stream — a header, the `Bvh::serialize` stream produced and consumed on the device
    (`bvhXX_broadcast`), the primitive array (`bvh_amd_comm_broadcast` into a torch-owned tensor) — nothing staged on a host.
    transport "staged" (gloo groups: CPU tests, several ranks on one GPU): the same device buffers through host tensors and
    `torch.distributed.broadcast`. `timing`, if given, receives {"broadcast_ms", "payload_bytes", "transport"} (wall time of
    the whole exchange on this rank, device-synchronised)."""
    import time
    import torch
    import torch.distributed as dist
    from . import _lib
    from .api import Bvh, _stream, _suffix
    rank = dist.get_rank() if dist.is_initialized() else 0
    transport = _pick_transport(transport)
    # everything that can raise on `src` alone happens before the first collective, so that the ranks cannot diverge
    meta_h = [0] * 8                                           # ndim, shape[0..3], is_double, bvh dim, stream bytes
    if rank == src:
        if bvh is None or prims is None:
            raise ValueError("broadcast_scene: the source rank passes its Bvh and primitive tensor")
        if not (torch.is_tensor(prims) and prims.is_cuda):
            raise TypeError("broadcast_scene: prims must be a CUDA tensor")
        if prims.dim() < 1 or prims.dim() > 4:
            raise ValueError("broadcast_scene: prims must have 1 to 4 dimensions")
        want = torch.float64 if bvh.dtype == np.float64 else torch.float32
        if prims.dtype != want:
            raise TypeError("broadcast_scene: prims dtype must match the BVH scalar type")
        prims = prims.contiguous()
        meta_h[0] = prims.dim()
        for i, s in enumerate(prims.shape):
            meta_h[1 + i] = int(s)
        meta_h[5], meta_h[6] = int(bvh.dtype == np.float64), bvh.dim
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if transport == "rccl" and comm is None:
        # the library's own communicator; should ncclCommInitRank fail on ANY rank, every rank learns it here and all of them take
        # torch.distributed's RCCL broadcast of the same device buffers instead (still no host copy) rather than diverging
        why = None
        try:
            comm = default_comm()
        except Exception as exc:                              # noqa: BLE001
            why = repr(exc)
        if dist.is_initialized() and dist.get_world_size() > 1:
            ok = torch.tensor([0 if why else 1], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                import sys
                print(f"[bvh_amd] rank {rank}: library communicator unavailable ({why or 'on another rank'}); using torch.distributed for the scene broadcast",
                      file=sys.stderr)
                comm, transport = None, "torch"
        elif why:
            raise _lib.BvhAmdError(why)
    if transport == "rccl":
        lib = _lib.load()
        meta = torch.tensor(meta_h, dtype=torch.int64, device="cuda")
        comm.broadcast_(meta, src)
        m = [int(v) for v in meta.tolist()]
        shape = tuple(m[1:1 + m[0]])
        tdt = torch.float64 if m[5] else torch.float32
        s = _suffix(np.dtype(np.float64 if m[5] else np.float32), m[6])
        out_prims, out_bytes = C.c_void_p(0), C.c_size_t(0)
        h = getattr(lib, f"bvh{s}_broadcast")(comm._h, src, bvh._h if rank == src else None, None, 0, C.byref(out_prims), C.byref(out_bytes), _stream())
        if not h:
            raise _lib.BvhAmdError(f"bvh{s}_broadcast: {_lib.last_error()}")
        if rank != src:
            bvh = Bvh(h, s)
            prims = torch.empty(shape, dtype=tdt, device="cuda")
        elif h != bvh._h:                                      # BVH_AMD_BROADCAST_LOOPBACK=1 (test knob): the root got a received copy
            bvh = Bvh(h, s)
        comm.broadcast_(prims, src)
        stream_bytes = 0                                       # (the library does not report it; recomputed below)
    else:
        on_device = transport == "torch" and dist.get_backend() == "nccl"     # torch's own RCCL moves the device buffers; gloo needs host staging
        meta = torch.tensor(meta_h, dtype=torch.int64, device="cuda" if on_device else "cpu")
        buf = None
        if rank == src:
            buf = bvh.serialize_device()
            meta[7] = buf.numel()
        dist.broadcast(meta, src)
        m = [int(v) for v in meta.tolist()]
        shape = tuple(m[1:1 + m[0]])
        tdt = torch.float64 if m[5] else torch.float32
        if on_device:
            if rank != src:
                buf = torch.empty(m[7], dtype=torch.uint8, device="cuda")
                prims = torch.empty(shape, dtype=tdt, device="cuda")
            dist.broadcast(buf, src)
            dist.broadcast(prims, src)
        else:
            hbuf = buf.cpu() if rank == src else torch.empty(m[7], dtype=torch.uint8)
            hprims = prims.cpu() if rank == src else torch.empty(shape, dtype=tdt)
            dist.broadcast(hbuf, src)
            dist.broadcast(hprims, src)
            if rank != src:
                buf, prims = hbuf.cuda(), hprims.cuda()
        if rank != src:
            bvh = Bvh.deserialize_device(buf, dtype=np.float64 if m[5] else np.float32, dim=m[6])
        stream_bytes = m[7]
    torch.cuda.synchronize()
    if timing is not None:
        if not stream_bytes:
            idx = 8 if bvh.dtype == np.float64 else 4
            stream_bytes = 2 * idx + bvh.node_count * (2 * bvh.dim * (8 if bvh.dtype == np.float64 else 4) + idx) + bvh.prim_count * idx
        timing["broadcast_ms"] = (time.perf_counter() - t0) * 1e3
        timing["payload_bytes"] = int(stream_bytes + prims.numel() * prims.element_size())
        timing["transport"] = ("RCCL (ncclBroadcast issued by libbvh_amd.so, device buffers end to end)" if transport == "rccl"
                               else "torch.distributed/nccl = RCCL (device buffers; the library's own communicator was unavailable)"
                               if transport == "torch" and dist.get_backend() == "nccl"
                               else "torch.distributed/" + dist.get_backend() + " with host staging (functional path, not RCCL)")
    return bvh, prims


def intersect_sharded(bvh, prims, rays, **kw):
    """Traces this rank's contiguous shard of a replicated ray array; returns (begin, end, hits)."""
    import torch.distributed as dist
    from .api import intersect
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    b, e = shard_range(len(rays), rank, world)
    return b, e, intersect(bvh, prims, rays[b:e], **kw)


class _DeviceBlock:
    """A device allocation handed out by the library (bvhXX_replicate's primitive copies), exposed to torch without a copy through
    `__cuda_array_interface__`; freed with bvh_amd_device_free on its own device when the last tensor viewing it is gone."""

    def __init__(self, ptr: int, shape, typestr: str, device: int):
        self._ptr, self._device = ptr, device
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    def __del__(self):
        try:
            import torch
            from . import _lib
            if self._ptr:
                with torch.cuda.device(self._device):
                    _lib.load().bvh_amd_device_free(C.c_void_p(self._ptr))
        except Exception:                                     # noqa: BLE001  (interpreter shutdown)
            pass
        self._ptr = 0


def replicate_scene(bvh, prims, devices=None, timing: dict = None):
    """ONE process, several GPUs — what a C caller of the reference API would write (tests/c/replicate.c): `bvhXX_replicate` copies the
    scene from the BVH's own device to every device of `devices` (default: all visible) with the library's ncclCommInitAll + one
    grouped ncclBroadcast of the `Bvh::serialize` stream and the BVH-ordered primitive array. No torch.distributed anywhere.
    Returns [(Bvh, prims tensor on that device)] in the order of `devices`; the entry of the BVH's own device holds the originals."""
    import time
    import torch
    from . import _lib
    from .api import Bvh
    lib = _lib.load()
    if devices is None:
        devices = list(range(torch.cuda.device_count()))
    devices = [int(d) for d in devices]
    n = len(devices)
    prims = prims.contiguous()
    home = prims.device.index
    if home not in devices:
        raise ValueError("replicate_scene: the scene's own device must be among `devices`")
    s = bvh._s
    arr = (C.c_int * n)(*devices)
    bvhs_out, prims_out = (C.c_void_p * n)(), (C.c_void_p * n)()
    torch.cuda.synchronize(home)
    t0 = time.perf_counter()
    with torch.cuda.device(home):
        _lib.check(getattr(lib, f"bvh{s}_replicate")(bvh._h, C.c_void_p(prims.data_ptr()), prims.numel() * prims.element_size(), n, arr,
                                                     C.cast(bvhs_out, C.POINTER(C.c_void_p)), C.cast(prims_out, C.POINTER(C.c_void_p))),
                   f"bvh{s}_replicate")
    for d in devices:
        torch.cuda.synchronize(d)
    if timing is not None:
        timing["replicate_ms"] = (time.perf_counter() - t0) * 1e3
        timing["transport"] = "RCCL inside libbvh_amd.so (bvhXX_replicate: ncclCommInitAll + grouped ncclBroadcast), one process"
    typestr = "<f8" if prims.dtype == torch.float64 else "<f4"
    out = []
    for i, d in enumerate(devices):
        if d == home:
            out.append((bvh, prims))
            continue
        with torch.cuda.device(d):
            block = _DeviceBlock(prims_out[i], prims.shape, typestr, d)
            t = torch.as_tensor(block, device=f"cuda:{d}")
            t._bvh_amd_block = block                              # the tensor keeps the allocation alive
            out.append((_DeviceBvh(bvhs_out[i], s, d), t))
    return out


def _DeviceBvh(handle, suffix, device):
    """A Bvh living on `device`: destroyed with that device current (include/bvh_amd.h: bvhXX_replicate)."""
    from .api import Bvh

    class _Bound(Bvh):
        def __del__(self):
            try:
                import torch
                with torch.cuda.device(device):
                    Bvh.__del__(self)
            except Exception:                                 # noqa: BLE001
                pass

    return _Bound(handle, suffix)
