#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
