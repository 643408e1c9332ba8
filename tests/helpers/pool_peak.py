"""Prints the high-water mark of the device's default memory pool (hipMemPoolAttrUsedMemHigh) over one 4M-triangle High build.
Own process: BVH_AMD_CACHE_MB is read once. Used by tests/test_gpu_build.py::test_cache_off_does_not_hold_the_sum_of_a_builds_scratch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bvh_amd
from bvh_amd import synth

hip = ctypes.CDLL("libamdhip64.so")
pool = ctypes.c_void_p()
assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0) == 0
USED_HIGH = 8                                                 # hipMemPoolAttrUsedMemHigh

def high():
    v = ctypes.c_uint64(0)
    assert hip.hipMemPoolGetAttribute(pool, USED_HIGH, ctypes.byref(v)) == 0
    return v.value

tris = torch.from_numpy(synth.soup(4_000_000)).cuda()
bb, cc = bvh_amd.tri_bounds(tris)
cfg = bvh_amd.Config(quality=bvh_amd.Quality.High)
bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool())      # (first build: code loads, the pool's first growth)
torch.cuda.synchronize()
zero = ctypes.c_uint64(0)
assert hip.hipMemPoolSetAttribute(pool, USED_HIGH, ctypes.byref(zero)) == 0
b = bvh_amd.DefaultBuilder.build(bb, cc, cfg, thread_pool=bvh_amd.ThreadPool())
torch.cuda.synchronize()
print("pool_used_high_mb", high() >> 20, "nodes", b.node_count)
