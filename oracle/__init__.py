"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes bindings for the two CPU checkers (see oracle/oracle_abi.h):

* ``load_oracle()``  -> oracle/libbvh_oracle.so, functions ``orc_*``: restatement of the reference's
  algorithm (oracle/bvh_oracle.cpp, cites the reference file:line it follows).
* ``load_ref()``     -> oracle/_ref/libbvh_ref.so, functions ``ref_*``: the unmodified reference
  compiled where it lies under /root/reference (oracle/ref_harness.cpp); ``None`` when absent.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (bvh_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

BUILDER_DEFAULT_SERIAL, BUILDER_DEFAULT_PARALLEL, BUILDER_BINNED, BUILDER_SWEEP = 0, 1, 2, 3
QUALITY_LOW, QUALITY_MEDIUM, QUALITY_HIGH = 0, 1, 2
INVALID = 0xFFFFFFFF

HITF = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
HITD = np.dtype([("prim", "<u4"), ("pad", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8")])
NODEF = np.dtype([("bounds", "<f4", (6,)), ("index", "<u4")])
NODED = np.dtype([("bounds", "<f8", (6,)), ("index", "<u8")])
NODE2F = np.dtype([("bounds", "<f4", (4,)), ("index", "<u4")])      # Node<float, 2>: {minx,maxx,miny,maxy}, index (20 bytes)
NODE2D = np.dtype([("bounds", "<f8", (4,)), ("index", "<u8")])      # Node<double, 2> (40 bytes)
assert HITF.itemsize == 16 and HITD.itemsize == 32 and NODEF.itemsize == 28 and NODED.itemsize == 56
assert NODE2F.itemsize == 20 and NODE2D.itemsize == 40
NODE_DTYPES = {"3f": NODEF, "3d": NODED, "2f": NODE2F, "2d": NODE2D}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class CpuBvh:
    """A BVH held by one of the CPU checkers."""

    def __init__(self, lib: "CpuLib", handle, suffix: str):
        self.lib, self.h, self.s = lib, handle, suffix

    def __del__(self):
        if getattr(self, "h", None):
            self.lib._fn("destroy", self.s)(self.h)
            self.h = None

    @property
    def node_count(self) -> int:
        return self.lib._fn("node_count", self.s)(self.h)

    @property
    def prim_count(self) -> int:
        return self.lib._fn("prim_count", self.s)(self.h)

    def nodes(self) -> np.ndarray:
        out = np.empty(self.node_count, dtype=NODE_DTYPES[self.s])
        self.lib._fn("get_nodes", self.s)(self.h, _ptr(out))
        return out

    def prim_ids(self) -> np.ndarray:
        out = np.empty(self.prim_count, dtype=np.uint64)
        self.lib._fn("get_prim_ids", self.s)(self.h, _ptr(out))
        return out

    def serialize(self) -> bytes:
        n = self.lib._fn("serialize", self.s)(self.h, None, 0)
        buf = np.empty(n, dtype=np.uint8)
        self.lib._fn("serialize", self.s)(self.h, _ptr(buf), n)
        return buf.tobytes()

    def extract(self, root_id: int) -> "CpuBvh":
        """Bvh::extract_bvh(root_id) (bvh.h:92-122)."""
        return CpuBvh(self.lib, self.lib._fn("extract", self.s)(self.h, root_id), self.s)

    def optimize(self, threads: int = -1, batch_size_ratio=None, max_iter_count=None):
        """ReinsertionOptimizer::optimize; threads < 0 = SequentialExecutor overload; Config (reinsertion_optimizer.h:18-24)."""
        if batch_size_ratio is None and max_iter_count is None:
            self.lib._fn("optimize", self.s)(self.h, threads)
        else:
            self.lib._fn("optimize_config", self.s)(self.h, threads, 0.05 if batch_size_ratio is None else float(batch_size_ratio),
                                                    3 if max_iter_count is None else int(max_iter_count))

    def refit(self):
        self.lib._fn("refit", self.s)(self.h)

    def intersect_tri(self, tris12, rays8, any_hit=False, robust=True, threads=1, counters=False):
        return self._intersect("intersect_tri", tris12, rays8, any_hit, robust, threads, counters)

    def intersect_sphere(self, sph4, rays8, any_hit=False, robust=True, threads=1, counters=False):
        return self._intersect("intersect_sphere", sph4, rays8, any_hit, robust, threads, counters)

    def _intersect(self, name, prims, rays8, any_hit, robust, threads, counters):
        dt = np.float32 if self.s[1] == "f" else np.float64
        dim = int(self.s[0])
        prims = np.ascontiguousarray(prims, dtype=dt)
        rays8 = np.ascontiguousarray(rays8, dtype=dt).reshape(-1, 2 * dim + 2)      # {org[dim], dir[dim], tmin, tmax}
        out = np.empty(len(rays8), dtype=HITF if self.s[1] == "f" else HITD)
        cnt = np.zeros(3, dtype=np.uint64)
        self.lib._fn(name, self.s)(self.h, _ptr(prims), _ptr(rays8), len(rays8), int(any_hit), int(robust),
                                   int(threads), _ptr(out), _ptr(cnt))
        return (out, cnt) if counters else out


class CpuLib:
    def __init__(self, path: str, prefix: str):
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        self._cache = {}

    _SIG = {
        "build": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                               C.c_size_t, C.c_int]),
        "optimize_config": (None, [C.c_void_p, C.c_int, C.c_double, C.c_size_t]),
        "build_minitree": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_double, C.c_size_t, C.c_int, C.c_size_t]),
        "destroy": (None, [C.c_void_p]),
        "node_count": (C.c_size_t, [C.c_void_p]),
        "prim_count": (C.c_size_t, [C.c_void_p]),
        "get_nodes": (None, [C.c_void_p, C.c_void_p]),
        "get_prim_ids": (None, [C.c_void_p, C.c_void_p]),
        "from_arrays": (C.c_void_p, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
        "serialize": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "optimize": (None, [C.c_void_p, C.c_int]),
        "extract": (C.c_void_p, [C.c_void_p, C.c_size_t]),
        "refit": (None, [C.c_void_p]),
        "prep_tris": (None, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
        "precompute_tris": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
        "sphere_bboxes": (None, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
        "intersect_tri": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p]),
        "intersect_sphere": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    }

    def _fn(self, name, suffix):
        key = f"{self.prefix}_{name}{suffix}"
        f = self._cache.get(key)
        if f is None:
            f = getattr(self.dll, key)
            f.restype, f.argtypes = self._SIG[name]
            self._cache[key] = f
        return f

    @staticmethod
    def _sfx(dtype, dim=3):
        return f"{dim}f" if np.dtype(dtype) == np.float32 else f"{dim}d"

    def set_sah(self, log_cluster_size: int = 0, cost_ratio: float = 1.0):
        """SplitHeuristic(log_cluster_size, cost_ratio) (split_heuristic.h:17-23) for the builds that follow; () restores the default."""
        f = getattr(self.dll, f"{self.prefix}_set_sah")
        f.restype, f.argtypes = None, [C.c_size_t, C.c_double]
        f(log_cluster_size, cost_ratio)

    def set_bin_count(self, bin_count: int = 8):
        """BinnedSahBuilder<Node, BinCount> (binned_sah_builder.h:18) for the BUILDER_BINNED builds that follow; () restores the default."""
        f = getattr(self.dll, f"{self.prefix}_set_bin_count")
        f.restype, f.argtypes = C.c_int, [C.c_size_t]
        if f(bin_count) != 0:
            raise ValueError(f"bin_count {bin_count} is not available in this checker")

    def hardware_threads(self) -> int:
        return int(getattr(self.dll, f"{self.prefix}_hardware_threads")())

    def build(self, bboxes, centers, builder=BUILDER_DEFAULT_SERIAL, quality=QUALITY_HIGH, min_leaf=1,
              max_leaf=8, parallel_threshold=1024, threads=0, log2_grid_dim=4) -> CpuBvh:
        dt = bboxes.dtype
        dim = np.asarray(centers).shape[-1]                   # (n, 3) centers + (n, 6) boxes, or (n, 2) + (n, 4) {min, max}
        s = self._sfx(dt, dim)
        bboxes = np.ascontiguousarray(bboxes, dtype=dt).reshape(-1, 2 * dim)
        centers = np.ascontiguousarray(centers, dtype=dt).reshape(-1, dim)
        assert len(bboxes) == len(centers) and len(bboxes) > 0
        h = self._fn("build", s)(_ptr(bboxes), _ptr(centers), len(bboxes), builder, quality, min_leaf, max_leaf,
                                 parallel_threshold, threads)
        if not h:
            raise RuntimeError("oracle build failed")
        return CpuBvh(self, h, s)

    def build_minitree(self, bboxes, centers, min_leaf=1, max_leaf=8, enable_pruning=True, pruning_area_ratio=0.01,
                       parallel_threshold=1024, threads=0, log2_grid_dim=4) -> CpuBvh:
        """MiniTreeBuilder::build(pool, bboxes, centers, config) (mini_tree_builder.h:29-58), 3D."""
        dt = bboxes.dtype
        s = self._sfx(dt)
        bboxes = np.ascontiguousarray(bboxes, dtype=dt).reshape(-1, 6)
        centers = np.ascontiguousarray(centers, dtype=dt).reshape(-1, 3)
        h = self._fn("build_minitree", s)(_ptr(bboxes), _ptr(centers), len(bboxes), min_leaf, max_leaf, int(enable_pruning),
                                          float(pruning_area_ratio), parallel_threshold, threads, log2_grid_dim)
        if not h:
            raise RuntimeError("oracle build failed")
        return CpuBvh(self, h, s)

    def from_arrays(self, nodes: np.ndarray, prim_ids: np.ndarray) -> CpuBvh:
        s = {28: "3f", 56: "3d", 20: "2f", 40: "2d"}[nodes.dtype.itemsize]
        nodes = np.ascontiguousarray(nodes)
        ids = np.ascontiguousarray(prim_ids, dtype=np.uint64)
        h = self._fn("from_arrays", s)(_ptr(nodes), len(nodes), _ptr(ids), len(ids))
        return CpuBvh(self, h, s)

    def prep_tris(self, tris9):
        dt = tris9.dtype
        t = np.ascontiguousarray(tris9, dtype=dt).reshape(-1, 9)
        bb = np.empty((len(t), 6), dtype=dt)
        cc = np.empty((len(t), 3), dtype=dt)
        self._fn("prep_tris", self._sfx(dt))(_ptr(t), len(t), _ptr(bb), _ptr(cc))
        return bb, cc

    def precompute_tris(self, tris9, perm=None):
        dt = tris9.dtype
        t = np.ascontiguousarray(tris9, dtype=dt).reshape(-1, 9)
        n = len(t) if perm is None else len(perm)
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint64)
        out = np.empty((n, 12), dtype=dt)
        self._fn("precompute_tris", self._sfx(dt))(_ptr(t), _ptr(p), n, _ptr(out))
        return out

    def std_sort_ids(self, keys):
        keys = np.ascontiguousarray(keys)
        out = np.empty(len(keys), dtype=np.uint32)
        f = getattr(self.dll, f"{self.prefix}_std_sort_ids{self._sfx(keys.dtype)}")
        f.restype, f.argtypes = None, [C.c_void_p, C.c_size_t, C.c_void_p]
        f(_ptr(keys), len(keys), _ptr(out))
        return out

    def sphere_bboxes(self, sph4):
        """(n, 4) spheres {center, radius} or (n, 3) circles -> bboxes (n, 2 dim) {min, max}, centers (n, dim)."""
        dt = sph4.dtype
        dim = np.asarray(sph4).shape[-1] - 1
        s4 = np.ascontiguousarray(sph4, dtype=dt).reshape(-1, dim + 1)
        bb = np.empty((len(s4), 2 * dim), dtype=dt)
        cc = np.empty((len(s4), dim), dtype=dt)
        self._fn("sphere_bboxes", self._sfx(dt, dim))(_ptr(s4), len(s4), _ptr(bb), _ptr(cc))
        return bb, cc


ORACLE_SO = os.path.join(_HERE, "libbvh_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libbvh_ref.so")


def build_checkers(quiet: bool = True) -> None:
    """Compiles the restatement, and the reference harness when /root/reference is present."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def load_oracle() -> CpuLib:
    if not os.path.exists(ORACLE_SO):
        build_checkers()
    return CpuLib(ORACLE_SO, "orc")


def gpu_checker():
    """What the `-m gpu` parity tests, smoke() and bench.py's cpu_baseline compare against: the compiled, unmodified reference whenever
    oracle/_ref/libbvh_ref.so is in the tree (it travels to the GPU box with the snapshot) — and then it MUST load: a reference library
    that is present but unusable is an error, never a silent fallback to the restatement (VERDICT r4 Weak 1a). Only a tree without the
    file (no /root/reference to build it from) gets the restatement, which is pinned to the reference-generated golden vectors."""
    if os.path.exists(REF_SO) or os.path.isdir("/root/reference/src/bvh/v2"):
        lib = load_ref()
        if lib is None or lib.prefix != "ref":
            raise RuntimeError(f"{REF_SO} exists but did not load as the compiled reference")
        return lib
    return load_oracle()


def load_ref_native():
    """The reference compiled with the GPU box's own `-march=native` expansion (oracle/Makefile: ref_native); None where the file is not in
    the tree. Loading it on a CPU without AVX-512 would fault at the first vector instruction: callers check cpu_has_avx512() first."""
    path = os.path.join(_HERE, "_ref", "libbvh_ref_native.so")
    return CpuLib(path, "ref") if os.path.exists(path) else None


def cpu_has_avx512() -> bool:
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    need = ("avx512f", "avx512vl", "avx512bw", "avx512dq", "avx512cd", "avx512vbmi", "avx512_vbmi2", "avx512_vnni", "avx512_bitalg", "avx512_vpopcntdq")
    if os.environ.get("BVH_AMD_NATIVE_REF_STRICT", "1") != "0":          # (0: run it on a CPU that has the common subsets only — a developer's try)
        need += ("avx512_vp2intersect", "avx512_bf16", "avx512ifma")
    return all(f in flags for f in need)


def load_ref():
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference/src/bvh/v2"):
            build_checkers()
        else:
            return None
    return CpuLib(REF_SO, "ref")
