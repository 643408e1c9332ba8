#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03; timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu.log
