#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'P'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch, bvh_amd
from bvh_amd import synth
from coop_probe import kernel_ms, scene, lib
for name in ("soup", "soup10m", "sponza_any"):
    bvh, prims, rays, any_hit, robust = scene(name)
    n = rays.shape[0]
    hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    for R in (6, 8, 10, 12, 16, 20, 24):
        row = []
        for L in (6, 8, 12, 16, 20):
            lib.bvh_amd_tuning(R, L, 1)
            k, c = kernel_ms(lambda: bvh_amd.intersect(bvh, prims, rays, any_hit, robust, out=hits, sort_rays=(name != "sponza_any")), 4)
            row.append(f"L={L}: {k:6.3f}")
        print(name, f"R={R:2d}", " | ".join(row), flush=True)
    lib.bvh_amd_tuning(-1, -1, -1)
P
