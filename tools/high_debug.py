"""One High build of the n-triangle soup with the developer library's reinsertion log (BVH_AMD_REINSERT_DEBUG=1): which iterations
replayed the heap, whether the refits were deferred.   python tools/high_debug.py [n_tris]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BVH_AMD_LIB", os.path.join(root, "bvh_amd", "lib", "libbvh_amd_dev.so"))
os.environ.setdefault("BVH_AMD_REINSERT_DEBUG", "1")
sys.path.insert(0, root)
import torch, bvh_amd
from bvh_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = torch.from_numpy(synth.soup(n, seed=7)).cuda()
bb, cc = bvh_amd.tri_bounds(t)
b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High))
torch.cuda.synchronize()
print(bvh_amd.last_optimize_profile())
