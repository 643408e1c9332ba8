#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/build_profile.sh $o/build "soup 1000000 0 1" "soup 10000000 0 1" "soup 1000000 1 1" > $o/build_profile.log 2>&1
cat $o/build/timings.log
for t in soup_1000000_0_1 soup_10000000_0_1 soup_1000000_1_1; do
  f=$(find $o/build/$t -name "*kernel_trace.csv" | head -1)
  echo "== $t"; python tools/timeline.py $f 15 | tail -45
  find $o/build/$t -name "*kernel_trace.csv" -size +3M -delete
done
timeout 600 python -m pytest tests/test_gpu_traverse.py tests/test_gpu_properties.py -x -q 2>&1 | tail -3
