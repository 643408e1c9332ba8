"""Developer probe: the ray order by grid cell of the origin (shipped) against the tree-entry key (BVH_AMD_RAY_KEY_DEPTH=k, read once
per process: one process per k). Soup 1M (2^24 rays) and the 10M mesh (12.5M rays), reordered, cooperative fetch 12/12; kernel ms +
whole call ms (keys + sort + kernel); sha1 of the hit records.   python tools/entry_key_probe.py <scene>"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bvh_amd
from coop_probe import kernel_ms, scene, lib

name = sys.argv[1]
bvh, prims, rays, any_hit, robust = scene(name)
n = rays.shape[0]
hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
for coop, refill, leaf in ((1, 12, 12), (0, 36, 12)):
    lib.bvh_amd_tuning(refill, leaf, coop, -1)
    k_ms, call_ms = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=hits, sort_rays=True), 5)
    sha = hashlib.sha1(hits.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{name:8s} key_depth={os.environ.get('BVH_AMD_RAY_KEY_DEPTH', 'grid')} cell_bits={os.environ.get('BVH_AMD_RAY_KEY_BITS', '6')} coop={coop}: kernel {k_ms:7.3f} ms | call {call_ms:7.3f} ms | hits {sha}", flush=True)
