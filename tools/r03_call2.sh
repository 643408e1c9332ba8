#!/usr/bin/env bash
# round 3, GPU call 2: quad-cooperative record fetch — microbenchmark (mode 4), A/B over thresholds on five scenes, parity of the
# Coop kernels on the traversal / fuzz / property tests
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/tcp_probe.py $o/tcp_probe2.txt > /dev/null 2>$o/tcp_probe2.err; echo "tcp_probe rc=$?"
timeout 900 python tools/coop_probe.py $o/coop_probe.txt > /dev/null 2>$o/coop_probe.err; echo "coop_probe rc=$?"
tail -50 $o/coop_probe.txt
BVH_AMD_COOP=1 timeout 900 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py tests/test_gpu_configs.py -x -q > $o/pytest_coop.log 2>&1; echo "pytest coop rc=$?"
tail -5 $o/pytest_coop.log
