"""Developer tool: tests/test_gpu_fuzz.py::test_fuzz_configs over many more seeds (every Config knob randomised)."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle
import test_gpu_fuzz as F

orc = oracle.load_oracle()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
t0 = time.time(); ran = 0; bad = []
for seed in range(lo, hi):
    if time.time() - t0 > budget: break
    try:
        F.test_fuzz_configs(orc, seed); ran += 1
    except Exception as e:                                    # noqa: BLE001
        bad.append((seed, repr(e)[:300])); print("FAIL", bad[-1], flush=True)
        orc.set_sah()
print(f"ran {ran} seeds in {time.time() - t0:.1f} s, failures: {len(bad)}", flush=True)
