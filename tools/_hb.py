import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bvh_amd
from bvh_amd import synth
tris = torch.from_numpy(synth.soup(1_000_000)).cuda()
bb, cc = bvh_amd.tri_bounds(tris)
b = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
torch.cuda.synchronize()
