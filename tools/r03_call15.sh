#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $o/pytest_gpu_by_file.log
for f in tests/test_gpu_*.py tests/test_cpp_mirror.py tests/test_ray_callback.py; do
  timeout 900 python -m pytest $f -q -m gpu > $o/one.log 2>&1; rc=$?
  echo "$f rc=$rc $(tail -1 $o/one.log)" | tee -a $o/pytest_gpu_by_file.log
  if [ $rc -ne 0 ]; then grep -E "^FAILED|Segmentation|Error" $o/one.log | head -5 | tee -a $o/pytest_gpu_by_file.log; fi
done
