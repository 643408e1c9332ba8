"""Developer probe: cost of the per-ray callback entry points (bvhXX_intersect_ray*, device walk + host leaf callback) next to
the batch kernel, and of the reference's own example programs (oracle/_ref/progs, built unmodified) at 256 x 256."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bvh_amd  # noqa: E402
from bvh_amd import synth  # noqa: E402
from test_ray_callback import _compile_c, _write_obj  # noqa: E402

tmp = tempfile.mkdtemp()
exe = _compile_c(os.path.join(tmp, "ray_callback"))
for scene, tris in (("sponza_proxy 262144", synth.sponza_proxy(262144)), ("soup 100000", synth.soup(100000))):
    lo, hi = synth.scene_bounds(tris)
    m = 20000
    rays = synth.rays_closest(m, lo, hi)
    inp = os.path.join(tmp, "in.bin")
    with open(inp, "wb") as f:
        f.write(np.array([len(tris), m], dtype=np.uint64).tobytes() + tris.tobytes() + rays.tobytes())
    for mode, threads in (("closest", 1), ("any", 1), ("closest", 8), ("closest", 32)):
        r = subprocess.run([exe, "3f", mode, "0", inp, os.path.join(tmp, "out.bin"), str(threads)], capture_output=True, text=True)
        print(f"{scene:22s} {mode:8s} per-ray callback API: {r.stdout.strip()} {r.stderr.strip()}", flush=True)
    bb, cc = bvh_amd.tri_bounds(tris)
    bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(), thread_pool=bvh_amd.ThreadPool())
    prims = bvh_amd.precompute_tris(tris, bvh.device_prim_ids())
    d_rays = torch.from_numpy(rays).cuda()
    bvh_amd.intersect(bvh, prims, d_rays)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bvh_amd.intersect(bvh, prims, d_rays)
    torch.cuda.synchronize()
    print(f"{scene:22s} batch kernel, the same {m} rays: {1e6 * (time.perf_counter() - t0) / m:.4f} us per ray", flush=True)

progs = os.path.join(ROOT, "oracle", "_ref", "progs")
if os.path.exists(os.path.join(progs, "benchmark")):
    cornell = np.load(os.path.join(ROOT, "tests", "golden", "cornell.npz"))["prims"]
    obj = os.path.join(tmp, "cornell.obj")
    _write_obj(obj, cornell)
    cam = ["--eye", "0", "1", "2", "--dir", "0", "0", "-1", "--up", "0", "1", "0", "--width", "256", "--height", "256"]
    for name in ("benchmark", "c_api_example"):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(progs, name), obj] + cam, capture_output=True, text=True, cwd=tmp)
        print(f"reference {name} (unmodified), Cornell box 256 x 256: {time.perf_counter() - t0:.2f} s wall; " +
              " | ".join(r.stdout.strip().splitlines()), flush=True)
