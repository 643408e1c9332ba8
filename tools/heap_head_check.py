"""Developer check of the candidate-heap kernels against each other (same bytes) with timing: the register-resident head (default)
vs round 5's two-wave loop (BVH_AMD_HEAP_PIPE=1, developer library) with the exact replay forced in every iteration.
    BVH_AMD_LIB=bvh_amd/lib/libbvh_amd_dev.so python tools/heap_head_check.py [n_triangles ...]"""
import os, sys, time, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(n, mode):
    import torch, bvh_amd
    from bvh_amd import synth
    os.environ["BVH_AMD_REINSERT"] = "exact"
    tris = torch.from_numpy(synth.soup(n, jitter=0.01) if n < 2_000_000 else synth.soup(n)).cuda()
    bb, cc = bvh_amd.tri_bounds(tris)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        bvh = bvh_amd.DefaultBuilder.build(bb, cc, bvh_amd.Config(quality=bvh_amd.Quality.High), thread_pool=bvh_amd.ThreadPool())
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        prof = bvh_amd.last_optimize_profile()
    digest = hashlib.sha1(bvh.serialize()).hexdigest()
    r = prof["replacements"]
    print(f"RESULT n={n} mode={mode} build_ms={min(ts):.1f} heap_ms={prof['heap_ms']:.1f} replacements={r} us_per_replacement={prof['heap_ms'] * 1e3 / max(r, 1):.3f} sha1={digest}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        sizes = [int(a) for a in sys.argv[1:]] or [1_000_000]
        for n in sizes:
            for mode in ("2", "1"):
                env = dict(os.environ, BVH_AMD_HEAP_PIPE=mode)
                subprocess.run([sys.executable, __file__, "--child", str(n), mode], env=env, timeout=900)
