"""Developer probe: traversal time vs the refill / leaf-parking thresholds (BVH_AMD_REFILL / BVH_AMD_LEAF) per scene."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
scenes = [("soup", 1000000), ("sponza", 262144), ("terrain", 1000000)]
for scene, n in scenes:
    for refill in [int(x) for x in os.environ.get('SWEEP_REFILL', '16,32,48').split(',')]:
        for leaf in [int(x) for x in os.environ.get('SWEEP_LEAF', '16,32,48').split(',')]:
            env = dict(os.environ, BVH_AMD_REFILL=str(refill), BVH_AMD_LEAF=str(leaf))
            r = subprocess.run([sys.executable, os.path.join(here, "quick_trace.py"), scene, str(n), str(1 << 23), "2"], env=env, capture_output=True, text=True, timeout=200)
            line = [l for l in r.stdout.splitlines() if "sort=False" in l]
            print(scene, "refill", refill, "leaf", leaf, line[0].split(":")[1].split("P=")[0].strip() if line else r.stderr[-200:], flush=True)
