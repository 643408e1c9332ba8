"""ctypes binding of libbvh_amd.so (include/bvh_amd.h). Fails loudly when the library is missing:
there is no Python/CPU fallback for any entry point."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BVH_AMD_LIB") or os.path.join(HERE, "lib", "libbvh_amd.so")     # (BVH_AMD_LIB: a developer's A/B build)

_lib = None


class BvhAmdError(RuntimeError):
    pass


class BuildConfig(C.Structure):
    _fields_ = [("quality", C.c_int), ("min_leaf_size", C.c_size_t), ("max_leaf_size", C.c_size_t),
                ("parallel_threshold", C.c_size_t)]


class Counters(C.Structure):
    _fields_ = [("node_pairs", C.c_ulonglong), ("prim_tests", C.c_ulonglong), ("leaves", C.c_ulonglong)]


class MiniTreeConfig(C.Structure):                             # struct bvh_amd_minitree_config
    _fields_ = [("min_leaf_size", C.c_size_t), ("max_leaf_size", C.c_size_t), ("enable_pruning", C.c_int),
                ("pruning_area_ratio", C.c_double), ("parallel_threshold", C.c_size_t), ("log2_grid_dim", C.c_size_t),
                ("log_cluster_size", C.c_size_t), ("cost_ratio", C.c_double)]


class BBox3f(C.Structure):
    _fields_ = [("v", C.c_float * 6)]


class BBox3d(C.Structure):
    _fields_ = [("v", C.c_double * 6)]


class BBox2f(C.Structure):
    _fields_ = [("v", C.c_float * 4)]


class BBox2d(C.Structure):
    _fields_ = [("v", C.c_double * 4)]


# name -> (restype, argtypes); `{S}` expands to 3f / 3d
_P, _Z, _I, _U = C.c_void_p, C.c_size_t, C.c_int, C.c_uint
_SIGS = {
    "bvh_amd_last_error": (C.c_char_p, []),
    "bvh_amd_version": (C.c_char_p, []),
    "bvh_amd_last_kernel_name": (C.c_char_p, []),
    "bvh_amd_last_launch_reordered": (C.c_int, []),
    "bvh_amd_kernel_timing": (None, [C.c_int]),
    "bvh_amd_kernel_times": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "bvh_amd_reorder_times": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "bvh_amd_tuning": (None, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bvh_amd_last_plan_search": (None, [C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
    "bvh_amd_experiment": (_I, [C.c_char_p, C.c_int]),
    "bvh_amd_wave_times": (_I, [_P, _Z, C.POINTER(C.c_size_t)]),
    "bvh_amd_last_launch_plan": (None, [C.POINTER(C.c_int)]),
    "bvh_amd_reinsertion_stats": (None, [C.POINTER(C.c_uint)]),
    "bvh_amd_last_optimize_profile": (None, [_P]),
    "bvh_amd_probe_record_walk": (_I, [_P, C.c_uint32, C.c_uint32, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong), _P]),
    "bvh_amd_probe_mixed_walk": (_I, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong), _P]),
    "bvh_amd_probe_record_walk_ex": (_I, [_P, C.c_uint32, C.c_uint32, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong), _P]),
    "bvh_amd_release_cached_memory": (_I, []),
    "bvh_amd_cached_scratch_bytes": (_Z, []),
    "bvh_amd_scratch_cache_limit": (_Z, []),
    "bvh_amd_device_count": (_I, []),
    "bvh_amd_device_name": (_I, [_I, C.c_char_p, _Z]),
    "bvh_amd_device_select": (_I, [_I]),
    "bvh_amd_device_current": (_I, []),
    "bvh_amd_comm_unique_id": (_I, [_P]),
    "bvh_amd_comm_create": (_P, [_P, _I, _I]),
    "bvh_amd_comm_adopt": (_P, [_P]),
    "bvh_amd_comm_destroy": (None, [_P]),
    "bvh_amd_comm_rank": (_I, [_P]),
    "bvh_amd_comm_size": (_I, [_P]),
    "bvh_amd_comm_handle": (_P, [_P]),
    "bvh_amd_comm_broadcast": (_I, [_P, _P, _Z, _I, _P]),
    "bvh_amd_comm_cache_clear": (_I, []),
    "bvh_amd_rccl_library": (C.c_char_p, []),
    "bvh_thread_pool_create": (_P, [_Z]),
    "bvh_thread_pool_destroy": (None, [_P]),
    "bvh_amd_gather": (_I, [_P, _P, _Z, _Z, _P, _P]),
    "bvh_amd_pinhole_rays3f": (_I, [_P, _P, _P, _Z, _Z, _P, _P]),
    "bvh_amd_pinhole_rays3d": (_I, [_P, _P, _P, _Z, _Z, _P, _P]),
    "bvh_amd_shade_eyelight3f": (_I, [_P, _P, _P, _Z, _P, _P]),
    "bvh_amd_shade_eyelight3d": (_I, [_P, _P, _P, _Z, _P, _P]),
    "bvh_amd_device_alloc": (_P, [_Z]),
    "bvh_amd_device_free": (None, [_P]),
    "bvh_amd_copy_to_device": (_I, [_P, _P, _Z]),
    "bvh_amd_copy_to_host": (_I, [_P, _P, _Z]),
    "bvh_amd_synchronize": (_I, [_P]),
    "bvh_amd_std_sort_ids3f": (_I, [_P, _Z, _P, _P]),
    "bvh_amd_std_sort_ids3d": (_I, [_P, _Z, _P, _P]),
    "bvh_amd_radix_sort_pairs_u32": (_I, [_P, _P, _Z, _I, _P]),
}
_SIGS_T = {
    "bvh{S}_build": (_P, [_P, _P, _P, _Z, _P]),
    "bvh{S}_build_device": (_P, [_P, _P, _Z, _P, _I, _P]),
    "bvh{S}_build_sah": (_P, [_P, _P, _P, _Z, _P, _P]),
    "bvh{S}_build_device_sah": (_P, [_P, _P, _Z, _P, _I, _P, _P]),
    "bvh{S}_build_device_binned": (_P, [_P, _P, _Z, _P, _P, _Z, _P]),
    "bvh{S}_build_minitree_device": (_P, [_P, _P, _Z, _P, _P]),
    "bvh{S}_from_nodes": (_P, [_P, _Z, _P, _Z]),
    "bvh{S}_extract": (_P, [_P, _Z]),
    "bvh{S}_destroy": (None, [_P]),
    "bvh{S}_optimize": (None, [_P, _P]),
    "bvh{S}_optimize_config": (_I, [_P, _P]),
    "bvh{S}_refit": (None, [_P]),
    "bvh{S}_refit_status": (_I, [_P]),
    "bvh{S}_sync_device": (_I, [_P]),
    "bvh{S}_append_node": (None, [_P]),
    "bvh{S}_remove_last_node": (None, [_P]),
    "bvh_node{S}_set_prim_count": (None, [_P, _Z]),
    "bvh_node{S}_set_first_id": (None, [_P, _Z]),
    "bvh_node{S}_set_bbox": (None, [_P, _P]),
    "bvh{S}_save": (None, [_P, _P]),
    "bvh{S}_load": (_P, [_P]),
    "bvh{S}_serialize": (_Z, [_P, _P, _Z]),
    "bvh{S}_deserialize": (_P, [_P, _Z]),
    "bvh{S}_serialize_device": (_Z, [_P, _P, _Z, _P]),
    "bvh{S}_deserialize_device": (_P, [_P, _Z, _P]),
    "bvh{S}_broadcast": (_P, [_P, _I, _P, _P, _Z, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _P]),
    "bvh{S}_replicate": (_I, [_P, _P, _Z, _I, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "bvh{S}_get_node": (_P, [_P, _Z]),
    "bvh{S}_get_prim_id": (_Z, [_P, _Z]),
    "bvh{S}_get_prim_count": (_Z, [_P]),
    "bvh{S}_get_node_count": (_Z, [_P]),
    "bvh_node{S}_is_leaf": (C.c_bool, [_P]),
    "bvh_node{S}_get_prim_count": (_Z, [_P]),
    "bvh_node{S}_get_first_id": (_Z, [_P]),
    "bvh{S}_copy_nodes": (None, [_P, _P]),
    "bvh{S}_copy_prim_ids": (None, [_P, _P]),
    "bvh{S}_device_prim_ids": (_P, [_P]),
    "bvh_amd_tri_bounds{S}": (_I, [_P, _Z, _P, _P, _P]),
    "bvh_amd_precompute_tris{S}": (_I, [_P, _P, _Z, _P, _P]),
    "bvh_amd_sphere_bounds{S}": (_I, [_P, _Z, _P, _P, _P]),
    "bvh{S}_intersect_rays_tri": (_I, [_P, _P, _P, _Z, _U, _P, _P, _P]),
    "bvh{S}_intersect_rays_sphere": (_I, [_P, _P, _P, _Z, _U, _P, _P, _P]),
    "bvh{S}_prepare_trace": (_I, [_P, _Z, _P]),
    # one ray, host leaf callback (c_api/bvh.h:277-295): (bvh, ray, callback struct)
    "bvh{S}_intersect_ray": (None, [_P, _P, _P]),
    "bvh{S}_intersect_ray_any": (None, [_P, _P, _P]),
    "bvh{S}_intersect_ray_robust": (None, [_P, _P, _P]),
    "bvh{S}_intersect_ray_any_robust": (None, [_P, _P, _P]),
    "bvh{S}_intersect_ray_visit": (_I, [_P, _P, _Z, _U, _P]),
}


class OptimizeProfile(C.Structure):         # struct bvh_amd_optimize_profile
    _fields_ = [("iterations", C.c_uint), ("replayed", C.c_uint), ("replacements", C.c_ulonglong), ("heap_ms", C.c_float)]


class SahConfig(C.Structure):                 # struct bvh_amd_sah_config
    _fields_ = [("log_cluster_size", C.c_size_t), ("cost_ratio", C.c_double)]


class OptimizeConfig(C.Structure):            # struct bvh_amd_optimize_config
    _fields_ = [("batch_size_ratio", C.c_double), ("max_iter_count", C.c_size_t)]


def ray_visitor_types(suffix: str):
    """(struct bvh_amd_ray_visitor{f,d}, leaf CFUNCTYPE, inner CFUNCTYPE) for a family suffix like '3f'."""
    scalar = C.c_float if suffix[1] == "f" else C.c_double
    leaf_t = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.POINTER(scalar), C.c_size_t, C.c_size_t)
    inner_t = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t)

    class Visitor(C.Structure):
        _fields_ = [("user_data", C.c_void_p), ("leaf_fn", leaf_t), ("inner_fn", inner_t)]
    return Visitor, leaf_t, inner_t


_ONLY_3D = ("bvh_amd_tri_bounds{S}", "bvh_amd_precompute_tris{S}", "bvh{S}_intersect_rays_tri", "bvh{S}_prepare_trace",     # tri.h is 3D only,
            "bvh{S}_build_minitree_device")                                                           # and so is the mini-tree grid


def exported_symbols():
    """Every symbol include/bvh_amd.h declares."""
    names = list(_SIGS)
    for s in ("3f", "3d", "2f", "2d"):
        names += [k.format(S=s) for k in _SIGS_T if s[0] == "3" or k not in _ONLY_3D]
        names += [f"bvh_node{s}_get_bbox"]
    return names


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's). It must be the first one mapped, or the
    # process ends up with two runtimes and the second cannot open the device ("no ROCm-capable device").
    import torch  # noqa: F401  (device memory + streams plumbing)
    if not os.path.exists(LIB_PATH):
        raise BvhAmdError(f"{LIB_PATH} is missing: build it with `python -m bvh_amd.build` "
                          "(no fallback path exists)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    for s in ("3f", "3d", "2f", "2d"):
        for name, (res, args) in _SIGS_T.items():
            if s[0] == "2" and name in _ONLY_3D:
                continue
            f = getattr(lib, name.format(S=s))
            f.restype, f.argtypes = res, args
    lib.bvh_node3f_get_bbox.restype, lib.bvh_node3f_get_bbox.argtypes = BBox3f, [_P]
    lib.bvh_node3d_get_bbox.restype, lib.bvh_node3d_get_bbox.argtypes = BBox3d, [_P]
    lib.bvh_node2f_get_bbox.restype, lib.bvh_node2f_get_bbox.argtypes = BBox2f, [_P]
    lib.bvh_node2d_get_bbox.restype, lib.bvh_node2d_get_bbox.argtypes = BBox2d, [_P]
    _lib = lib
    return lib


def last_error() -> str:
    return load().bvh_amd_last_error().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise BvhAmdError(f"{what} failed ({rc}): {last_error()}")
