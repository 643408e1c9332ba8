"""Host-side mirror of the reference interface (bvh::v2) over the C-ABI. torch = device memory + streams."""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field

import numpy as np

from . import _lib

INVALID = 0xFFFFFFFF
HITF = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
HITD = np.dtype([("prim", "<u4"), ("pad", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8")])
NODEF = np.dtype([("bounds", "<f4", (6,)), ("index", "<u4")])     # reference node.h:31-37
NODED = np.dtype([("bounds", "<f8", (6,)), ("index", "<u8")])
NODE2F = np.dtype([("bounds", "<f4", (4,)), ("index", "<u4")])    # Node<float, 2>: {minx,maxx,miny,maxy}, index
NODE2D = np.dtype([("bounds", "<f8", (4,)), ("index", "<u8")])
_NODE_DTYPES = {"3f": NODEF, "3d": NODED, "2f": NODE2F, "2d": NODE2D}


class Quality(enum.IntEnum):          # default_builder.h:21
    Low = 0
    Medium = 1
    High = 2


class RayFlags(enum.IntFlag):
    ANY_HIT = 1
    ROBUST = 2
    SORTED = 4
    UNSORTED = 16


class _Builder(enum.IntEnum):
    DEFAULT_SERIAL = 0
    DEFAULT_PARALLEL = 1
    BINNED = 2
    SWEEP = 3


@dataclass
class SplitHeuristic:                 # split_heuristic.h:17-23: SplitHeuristic(log_cluster_size = 0, cost_ratio = 1)
    log_cluster_size: int = 0
    cost_ratio: float = 1.0

    def _c(self):
        return _lib.SahConfig(int(self.log_cluster_size), float(self.cost_ratio))


@dataclass
class Config:                         # default_builder.h:23-30 + top_down_sah_builder.h:27-40
    quality: Quality = Quality.High
    min_leaf_size: int = 1
    max_leaf_size: int = 8
    parallel_threshold: int = 1024
    sah: SplitHeuristic = field(default_factory=SplitHeuristic)

    def _c(self):
        return _lib.BuildConfig(int(self.quality), self.min_leaf_size, self.max_leaf_size, self.parallel_threshold)


class ThreadPool:
    """bvh::v2::ThreadPool stand-in (thread_pool.h:13-46). The GPU grid replaces the pool; passing one
    selects the reference's parallel (mini-tree) builder semantics exactly like the reference API."""

    def __init__(self, thread_count: int = 0):
        self.thread_count = thread_count


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _lib.BvhAmdError("no HIP device visible to torch: bvh_amd has no CPU path")
    return torch


def _suffix(dtype, dim: int = 3) -> str:
    import torch
    if dim not in (2, 3):
        raise ValueError("BVHs are 2- or 3-dimensional")
    if dtype in (torch.float32, np.float32, np.dtype(np.float32)):
        return f"{dim}f"
    if dtype in (torch.float64, np.float64, np.dtype(np.float64)):
        return f"{dim}d"
    raise TypeError(f"unsupported scalar type {dtype}")


def _dev(x, cols=None):
    """numpy / torch (any device) -> contiguous torch tensor on the current HIP device."""
    torch = _torch()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.cuda()
    x = x.contiguous()
    if cols is not None:
        x = x.reshape(-1, cols)
    return x


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


class Bvh:
    """bvh::v2::Bvh<Node<T,3>> or Bvh<Node<T,2>> (bvh.h:17-89): host mirror in the reference layout + device-resident copy."""

    def __init__(self, handle, suffix: str):
        if not handle:
            raise _lib.BvhAmdError(_lib.last_error())
        self._h, self._s = handle, suffix
        self._lib = _lib.load()

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._lib, f"bvh{self._s}_destroy")(self._h)
            self._h = None

    def _f(self, name):
        return getattr(self._lib, name.format(S=self._s))

    @property
    def dtype(self):
        return np.float32 if self._s[1] == "f" else np.float64

    @property
    def dim(self) -> int:
        return int(self._s[0])

    @property
    def node_count(self) -> int:
        return self._f("bvh{S}_get_node_count")(self._h)

    @property
    def prim_count(self) -> int:
        return self._f("bvh{S}_get_prim_count")(self._h)

    @property
    def nodes(self) -> np.ndarray:
        out = np.empty(self.node_count, dtype=_NODE_DTYPES[self._s])
        self._f("bvh{S}_copy_nodes")(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    @property
    def prim_ids(self) -> np.ndarray:
        out = np.empty(self.prim_count, dtype=np.uint64)
        self._f("bvh{S}_copy_prim_ids")(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def device_prim_ids(self):
        """uint32 prim ids resident in HBM, as a torch tensor view (no copy)."""
        torch = _torch()
        ptr = self._f("bvh{S}_device_prim_ids")(self._h)
        n = self.prim_count

        class _Holder:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        t = torch.as_tensor(_Holder(), device="cuda")
        t._bvh_keepalive = self
        return t

    def optimize(self, thread_pool=None, batch_size_ratio=None, max_iter_count=None):
        """ReinsertionOptimizer::optimize (reinsertion_optimizer.h:27-35), in place on the device; the two keyword arguments are
        its Config (:18-24, defaults 0.05 and 3)."""
        _torch()
        if batch_size_ratio is not None or max_iter_count is not None:
            cfg = _lib.OptimizeConfig(0.05 if batch_size_ratio is None else float(batch_size_ratio),
                                      3 if max_iter_count is None else int(max_iter_count))
            _lib.check(self._f("bvh{S}_optimize_config")(self._h, C.byref(cfg)), "optimize")
            return
        _lib.check(self._f("bvh{S}_optimize_config")(self._h, None), "optimize")      # NULL config = the reference's defaults

    def refit(self):
        """Bvh::refit (bvh.h:211-218) on the device; pushes host-side node edits first."""
        _torch()
        _lib.check(self._f("bvh{S}_refit_status")(self._h), "refit")

    def set_node_bbox(self, node_id: int, lo, hi):
        """bvh_nodeXX_set_bbox on the host mirror (takes effect on the device at the next refit()/sync_device())."""
        node = self._f("bvh{S}_get_node")(self._h, node_id)
        ct = C.c_float if self._s[1] == "f" else C.c_double
        bb = (ct * (2 * self.dim))(*[float(v) for v in list(lo) + list(hi)])
        self._f("bvh_node{S}_set_bbox")(node, bb)

    def sync_device(self):
        _lib.check(self._f("bvh{S}_sync_device")(self._h), "sync_device")

    def sync_host(self):
        """Fills the host mirror (reference-layout nodes + prim ids) from the device now instead of at the first accessor: the
        device-to-host copy that turns a device-resident build into the reference's host `Bvh` (what bvhXX_get_node does first)."""
        if not self._f("bvh{S}_get_node")(self._h, 0):
            raise _lib.BvhAmdError(_lib.last_error())

    def get_root(self):
        return self.nodes[0]

    def serialize(self) -> bytes:                     # bvh.h:221-229
        n = self._f("bvh{S}_serialize")(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._f("bvh{S}_serialize")(self._h, buf, n)
        return buf.raw

    @staticmethod
    def deserialize(data: bytes, dtype=np.float32, dim: int = 3) -> "Bvh":   # bvh.h:231-243
        s = _suffix(np.dtype(dtype), dim)
        lib = _lib.load()
        _torch()
        return Bvh(getattr(lib, f"bvh{s}_deserialize")(data, len(data)), s)

    def serialize_device(self):
        """Bvh::serialize (bvh.h:221-229) into HBM: a uint8 torch tensor holding the reference's byte stream, written from the
        resident nodes on the current stream (no host copy). This is the RCCL broadcast payload."""
        torch = _torch()
        n = self._f("bvh{S}_serialize_device")(self._h, None, 0, _stream())
        buf = torch.empty(n, dtype=torch.uint8, device="cuda")
        if self._f("bvh{S}_serialize_device")(self._h, buf.data_ptr(), n, _stream()) != n:
            raise _lib.BvhAmdError(_lib.last_error())
        return buf

    @staticmethod
    def deserialize_device(buf, dtype=np.float32, dim: int = 3) -> "Bvh":
        """Bvh::deserialize (bvh.h:231-243) of a byte stream resident in HBM (uint8 torch tensor): kernels and device-to-device
        copies only; the stream's structure is validated on the device."""
        torch = _torch()
        s = _suffix(np.dtype(dtype), dim)
        if not (buf.is_cuda and buf.dtype == torch.uint8 and buf.is_contiguous()):
            raise TypeError("deserialize_device takes a contiguous uint8 tensor resident on the GPU")
        return Bvh(getattr(_lib.load(), f"bvh{s}_deserialize_device")(buf.data_ptr(), buf.numel(), _stream()), s)

    def intersect_ray(self, ray, leaf_fn, any_hit: bool = False, robust: bool = False, inner_fn=None, start=None):
        """Bvh::intersect<IsAnyHit, IsRobust>(ray, start, stack, leaf_fn, inner_fn) (bvh.h:72-73) for ONE ray with host
        callbacks, over bvhXX_intersect_ray_visit: leaf_fn(tmax, begin, end) -> (was_hit, new_tmax) receives a BVH-order
        primitive range and the ray's current tmax and returns the possibly shortened tmax; inner_fn(first_child_id), if
        given, is called per visited pair. `ray` = {org, dir, tmin, tmax}; `start` = packed index word (default: the root's).
        The walk runs on the device (several kernel launches per ray): for throughput use bvh_amd.intersect."""
        lib = _lib.load()
        dt = np.float32 if self._s[1] == "f" else np.float64
        r = np.ascontiguousarray(ray, dtype=dt).reshape(2 * self.dim + 2)
        Visitor, leaf_t, inner_t = _lib.ray_visitor_types(self._s)
        failure = []

        def on_leaf(_user, t, begin, end):
            try:
                hit, new_t = leaf_fn(dt(t[0]), int(begin), int(end))
                t[0] = new_t
                return bool(hit)
            except BaseException as exc:                      # an exception cannot cross the C frames: end the walk, re-raise after
                failure.append(exc)
                t[0] = float("-inf")
                return True

        def on_inner(_user, first):
            try:
                if not failure:
                    inner_fn(int(first))
            except BaseException as exc:
                failure.append(exc)

        v = Visitor(None, leaf_t(on_leaf), inner_t(on_inner) if inner_fn is not None else inner_t())
        if start is None:
            start = (1 << 64) - 1                             # BVH_AMD_START_AT_ROOT
        rc = getattr(lib, f"bvh{self._s}_intersect_ray_visit")(self._h, r.ctypes.data, int(start), (RayFlags.ANY_HIT if any_hit else 0) | (RayFlags.ROBUST if robust else 0), C.byref(v))
        if failure:
            raise failure[0]
        _lib.check(rc, "intersect_ray_visit")

    def extract_bvh(self, root_id: int) -> "Bvh":
        """Bvh::extract_bvh (bvh.h:92-122): the subtree under node `root_id` as its own BVH (device op)."""
        h = getattr(_lib.load(), f"bvh{self._s}_extract")(self._h, int(root_id))
        return Bvh(h, self._s)

    @staticmethod
    def from_nodes(nodes: np.ndarray, prim_ids: np.ndarray) -> "Bvh":
        s = {28: "3f", 56: "3d", 20: "2f", 40: "2d"}[nodes.dtype.itemsize]
        lib = _lib.load()
        _torch()
        nodes = np.ascontiguousarray(nodes)
        ids = np.ascontiguousarray(prim_ids, dtype=np.uint64)
        return Bvh(getattr(lib, f"bvh{s}_from_nodes")(nodes.ctypes.data_as(C.c_void_p), len(nodes),
                                                      ids.ctypes.data_as(C.c_void_p), len(ids)), s)


def _build(bboxes, centers, config: Config, builder: _Builder, bin_count: int | None = None) -> Bvh:
    lib = _lib.load()
    dim = int(np.shape(centers)[-1])                          # (n,6) + (n,3), or (n,4) {min.x,min.y,max.x,max.y} + (n,2)
    bb = _dev(bboxes, 2 * dim)
    cc = _dev(centers, dim)
    if bb.dtype != cc.dtype or bb.shape[0] != cc.shape[0]:
        raise ValueError("bboxes (n, 2 dim) and centers (n, dim) must agree in dtype and length")
    s = _suffix(bb.dtype, dim)
    cfg = config._c()
    sah = config.sah._c()
    if bin_count is not None:                                 # BinnedSahBuilder<Node, BinCount> with a BinCount of its own
        h = getattr(lib, f"bvh{s}_build_device_binned")(bb.data_ptr(), cc.data_ptr(), bb.shape[0], C.byref(cfg), C.byref(sah), int(bin_count), _stream())
    else:
        h = getattr(lib, f"bvh{s}_build_device_sah")(bb.data_ptr(), cc.data_ptr(), bb.shape[0], C.byref(cfg), int(builder), C.byref(sah), _stream())
    return Bvh(h, s)


class DefaultBuilder:
    """bvh::v2::DefaultBuilder<Node>::build (default_builder.h:33-62)."""

    Config = Config
    Quality = Quality

    @staticmethod
    def build(bboxes, centers, config: Config | None = None, thread_pool: ThreadPool | None = None) -> Bvh:
        return _build(bboxes, centers, config or Config(),
                      _Builder.DEFAULT_PARALLEL if thread_pool is not None else _Builder.DEFAULT_SERIAL)


class BinnedSahBuilder:
    """bvh::v2::BinnedSahBuilder<Node, BinCount>::build (binned_sah_builder.h:18, :32-38); bin_count = the BinCount template
    argument (4, 8 = the reference's default, 16 or 32)."""

    @staticmethod
    def build(bboxes, centers, config: Config | None = None, bin_count: int = 8) -> Bvh:
        return _build(bboxes, centers, config or Config(), _Builder.BINNED, None if bin_count == 8 else bin_count)


class SweepSahBuilder:
    """bvh::v2::SweepSahBuilder<Node>::build (sweep_sah_builder.h:30-36)."""

    @staticmethod
    def build(bboxes, centers, config: Config | None = None) -> Bvh:
        return _build(bboxes, centers, config or Config(), _Builder.SWEEP)


class MiniTreeBuilder:
    """bvh::v2::MiniTreeBuilder<Node>::build(thread_pool, bboxes, centers, config) (mini_tree_builder.h:29-58), 3D."""

    @dataclass
    class Config:
        min_leaf_size: int = 1
        max_leaf_size: int = 8
        enable_pruning: bool = True
        pruning_area_ratio: float = 0.01
        parallel_threshold: int = 1024
        log2_grid_dim: int = 4
        sah: SplitHeuristic = field(default_factory=SplitHeuristic)

    @staticmethod
    def build(bboxes, centers, config: "MiniTreeBuilder.Config | None" = None, thread_pool: ThreadPool | None = None) -> Bvh:
        c = config or MiniTreeBuilder.Config()
        bb, cc = _dev(bboxes, 6), _dev(centers, 3)
        if bb.dtype != cc.dtype or bb.shape[0] != cc.shape[0]:
            raise ValueError("bboxes (n,6) and centers (n,3) must agree in dtype and length")
        s = _suffix(bb.dtype)
        cfg = _lib.MiniTreeConfig(c.min_leaf_size, c.max_leaf_size, int(c.enable_pruning), float(c.pruning_area_ratio), c.parallel_threshold,
                                  c.log2_grid_dim, int(c.sah.log_cluster_size), float(c.sah.cost_ratio))
        return Bvh(getattr(_lib.load(), f"bvh{s}_build_minitree_device")(bb.data_ptr(), cc.data_ptr(), bb.shape[0], C.byref(cfg), _stream()), s)


def tri_bounds(tris9):
    """Tri::get_bbox / Tri::get_center (tri.h:24-25) for n triangles -> (bboxes (n,6), centers (n,3)) in HBM."""
    torch = _torch()
    t = _dev(tris9, 9)
    s = _suffix(t.dtype)
    bb = torch.empty((t.shape[0], 6), dtype=t.dtype, device=t.device)
    cc = torch.empty((t.shape[0], 3), dtype=t.dtype, device=t.device)
    _lib.check(getattr(_lib.load(), f"bvh_amd_tri_bounds{s}")(t.data_ptr(), t.shape[0], bb.data_ptr(), cc.data_ptr(), _stream()),
               "tri_bounds")
    return bb, cc


def precompute_tris(tris9, perm=None):
    """PrecomputedTri (tri.h:35-37) of tris[perm[i]] -> (n,12) {p0,e1,e2,n} in HBM."""
    torch = _torch()
    t = _dev(tris9, 9)
    s = _suffix(t.dtype)
    if perm is not None:
        perm = _dev(perm) if not isinstance(perm, np.ndarray) else _dev(perm.astype(np.int32))
        perm = perm.to(torch.int32)
        n = perm.shape[0]
    else:
        n = t.shape[0]
    out = torch.empty((n, 12), dtype=t.dtype, device=t.device)
    _lib.check(getattr(_lib.load(), f"bvh_amd_precompute_tris{s}")(t.data_ptr(), perm.data_ptr() if perm is not None else None,
                                                                    n, out.data_ptr(), _stream()), "precompute_tris")
    return out


def sphere_bounds(sph4):
    """Sphere::get_bbox / get_center (sphere.h:24-27): (n,4) spheres {center, radius} -> (n,6), (n,3); (n,3) circles
    {center.x, center.y, radius} -> (n,4) {min.x,min.y,max.x,max.y}, (n,2)."""
    torch = _torch()
    dim = int(np.shape(sph4)[-1]) - 1
    t = _dev(sph4, dim + 1)
    s = _suffix(t.dtype, dim)
    bb = torch.empty((t.shape[0], 2 * dim), dtype=t.dtype, device=t.device)
    cc = torch.empty((t.shape[0], dim), dtype=t.dtype, device=t.device)
    _lib.check(getattr(_lib.load(), f"bvh_amd_sphere_bounds{s}")(t.data_ptr(), t.shape[0], bb.data_ptr(), cc.data_ptr(), _stream()),
               "sphere_bounds")
    return bb, cc


def pinhole_rays(width: int, height: int, eye, direction, up, dtype=np.float32):
    """Primary rays of the reference's benchmark camera (test/benchmark.cpp:343-359), generated on the device: (h*w, 8)."""
    torch = _torch()
    dt = np.dtype(dtype)
    s = _suffix(torch.float32 if dt == np.float32 else torch.float64)
    arr = [np.ascontiguousarray(np.asarray(v, dtype=dt).reshape(3)) for v in (eye, direction, up)]
    out = torch.empty((width * height, 8), dtype=torch.float32 if dt == np.float32 else torch.float64, device="cuda")
    _lib.check(getattr(_lib.load(), f"bvh_amd_pinhole_rays{s}")(arr[0].ctypes.data_as(C.c_void_p), arr[1].ctypes.data_as(C.c_void_p),
                                                              arr[2].ctypes.data_as(C.c_void_p), width, height, out.data_ptr(), _stream()),
               "pinhole_rays")
    return out


def shade_eyelight(prims12, rays, hits):
    """Eyelight shading of test/benchmark.cpp:363-371 on the device -> (n, 3) uint8."""
    torch = _torch()
    p, r = _dev(prims12, 12), _dev(rays, 8)
    s = _suffix(p.dtype)
    out = torch.empty((r.shape[0], 3), dtype=torch.uint8, device=p.device)
    _lib.check(getattr(_lib.load(), f"bvh_amd_shade_eyelight{s}")(p.data_ptr(), r.data_ptr(), hits.data_ptr(), r.shape[0], out.data_ptr(), _stream()),
               "shade_eyelight")
    return out


def gather(records, perm):
    """out[i] = records[perm[i]] on the device (prim permutation, test/benchmark.cpp:221-225)."""
    torch = _torch()
    r = _dev(records)
    p = _dev(perm).to(torch.int32)
    out = torch.empty((p.shape[0],) + tuple(r.shape[1:]), dtype=r.dtype, device=r.device)
    stride = r.element_size() * int(np.prod(r.shape[1:]))
    _lib.check(_lib.load().bvh_amd_gather(r.data_ptr(), p.data_ptr(), p.shape[0], stride, out.data_ptr(), _stream()), "gather")
    return out


def prepare_trace(bvh: Bvh, n_rays_hint: int = 0):
    """bvhXX_prepare_trace: the per-tree one-offs of the first large batch (depth pass, first reordering scratch for batches of
    `n_rays_hint` rays) paid now, on the current stream. Optional; 3D families."""
    if bvh.dim != 3:
        return
    _lib.check(getattr(_lib.load(), f"bvh{bvh._s}_prepare_trace")(bvh._h, int(n_rays_hint), _stream()), "prepare_trace")


def intersect(bvh: Bvh, prims, rays, any_hit: bool = False, robust: bool = False, leaf: str = "tri",
              counters: bool = False, out=None, sort_rays=None, original_ids: bool = False):
    """Batched Bvh::intersect<IsAnyHit, IsRobust> (bvh.h:160-182) with the closest/any-hit leaf loop of
    test/benchmark.cpp:281-291. prims are in BVH order. Returns a torch tensor of hit records
    ((n,4) of the BVH scalar type; view with hits_to_numpy) and, optionally, (pairs, tests, leaves). original_ids: report
    the original primitive id (bvh.prim_ids[i]) instead of the BVH-order index i the reference's leaf callback sees."""
    torch = _torch()
    lib = _lib.load()
    s = bvh._s
    dt = torch.float32 if s[1] == "f" else torch.float64
    r = _dev(rays, 2 * bvh.dim + 2)                           # {org, dir, tmin, tmax}
    p = _dev(prims)
    if bvh.dim == 2:
        leaf = "sphere"                                       # circles {center.x, center.y, radius}: the only 2D leaf (sphere.h)
    if r.dtype != dt or p.dtype != dt:
        raise TypeError("prims/rays dtype must match the BVH scalar type")
    n = r.shape[0]
    if out is None:
        out = torch.empty((n, 4), dtype=dt, device=r.device)
    cnt = torch.zeros(3, dtype=torch.int64, device=r.device) if counters else None
    flags = (RayFlags.ANY_HIT if any_hit else 0) | (RayFlags.ROBUST if robust else 0) | (0 if sort_rays is None else RayFlags.SORTED if sort_rays else RayFlags.UNSORTED) | \
            (8 if original_ids else 0)                        # BVH_AMD_RAY_ORIGINAL_IDS: hit.prim = bvh.prim_ids[BVH-order index]
    fn = getattr(lib, f"bvh{s}_intersect_rays_{'tri' if leaf == 'tri' else 'sphere'}")
    _lib.check(fn(bvh._h, p.data_ptr(), r.data_ptr(), n, int(flags), out.data_ptr(),
                  cnt.data_ptr() if counters else None, _stream()), "intersect_rays")
    if counters:
        return out, cnt
    return out


def hits_to_numpy(hits) -> np.ndarray:
    a = hits.detach().cpu().numpy()
    return a.view(HITF if a.dtype == np.float32 else HITD).reshape(-1)


def reinsertion_stats():
    """(fast, exact) ReinsertionOptimizer iterations run so far in this process (include/bvh_amd.h: bvh_amd_reinsertion_stats)."""
    out = (C.c_uint * 2)()
    _lib.load().bvh_amd_reinsertion_stats(out)
    return int(out[0]), int(out[1])


def last_optimize_profile() -> dict:
    """The calling thread's latest ReinsertionOptimizer run (a High build's optimize step included): iterations, exact heap
    replays among them, pop + push replacements of those replays, GPU milliseconds of the heap kernels."""
    p = _lib.OptimizeProfile()
    _lib.load().bvh_amd_last_optimize_profile(C.byref(p))
    return {"iterations": int(p.iterations), "replayed": int(p.replayed), "replacements": int(p.replacements), "heap_ms": float(p.heap_ms)}


def std_sort_ids(keys):
    """ids sorted exactly like libstdc++'s std::sort(iota, by keys[i] < keys[j]) incl. tie arrangement (int32 tensor)."""
    torch = _torch()
    k = _dev(keys).reshape(-1)
    s = _suffix(k.dtype)
    out = torch.empty(k.shape[0], dtype=torch.int32, device=k.device)
    _lib.check(getattr(_lib.load(), f"bvh_amd_std_sort_ids{s}")(k.data_ptr(), k.shape[0], out.data_ptr(), _stream()), "std_sort_ids")
    return out


def radix_sort_pairs(keys_u32, vals_u32, bits=32):
    """Stable LSD radix sort by the low `bits` bits; returns (keys, vals) as new int32 tensors."""
    torch = _torch()
    k = _dev(keys_u32).to(torch.int32).clone()
    v = _dev(vals_u32).to(torch.int32).clone()
    _lib.check(_lib.load().bvh_amd_radix_sort_pairs_u32(k.data_ptr(), v.data_ptr(), k.shape[0], bits, _stream()), "radix_sort_pairs")
    return k, v
