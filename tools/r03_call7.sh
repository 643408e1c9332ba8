#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "soup 1000000 0 1" "soup 10000000 0 1" "soup 1000000 1 1" "soup 10000000 1 1" "terrain 1000000 0 1" "soup 1000000 0 0"; do python tools/build_profile.py $spec 5 | grep BUILD; done
for spec in "soup 1000000 0 1" "soup 10000000 0 1"; do BVH_AMD_GATHER=0 python tools/build_profile.py $spec 5 | grep BUILD | sed 's/^/GATHER=0 /'; done
timeout 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_2d.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_10m.py -x -q -k "minitree_streams" 2>&1 | tail -3
