"""Developer tool: runs the differential fuzz tests of tests/test_gpu_fuzz.py over many more seeds than the committed suite."""
import os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle
import test_gpu_fuzz as F

orc = oracle.gpu_checker()        # the compiled reference where it travelled with the tree, else the pinned restatement
lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
t0 = time.time(); ran = 0; bad = []
for seed in range(lo, hi):
    for kind in ("lattice", "dups", "flat", "points", "scales", "uniform"):
        if time.time() - t0 > budget: break
        try:
            F.test_fuzz_3d(orc, seed, kind); ran += 1
        except AssertionError as e:
            bad.append(("3d", seed, kind, str(e)[:200])); print("FAIL", bad[-1], flush=True)
    if time.time() - t0 > budget: break
    for fn, name in ((F.test_fuzz_2d, "2d"), (F.test_fuzz_3d_spheres, "spheres"), (F.test_fuzz_configs, "configs")):
        try:
            fn(orc, seed); ran += 1
        except AssertionError as e:
            bad.append((name, seed, str(e)[:200])); print("FAIL", bad[-1], flush=True)
print(f"ran {ran} cases in {time.time() - t0:.1f} s, failures: {len(bad)}", flush=True)
