#!/usr/bin/env bash
# Developer script (round 4): GPU suite, bench line (with its own PMC passes), BASELINE configs, build timings of every mode, refit / extract timings.
set -u
o=gpurun_out/r04f; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
timeout 900 python bench.py > $o/bench.json 2>$o/bench.err; echo "bench rc=$?"
timeout 900 python tools/run_configs.py > $o/configs.jsonl 2>$o/configs.err; echo "configs rc=$?"
for spec in "soup 1000000 0 1" "soup 1000000 1 1" "soup 1000000 2 1" "terrain 1000000 0 1" "terrain 1000000 1 1" "sponza 262144 0 1" "sponza 262144 1 1" \
            "soup 10000000 0 1" "soup 10000000 1 1" "terrain 10000000 1 1" "soup 1000000 0 0" "soup 1000000 1 0" "sponza 262144 1 0" "soup 10000000 1 0"; do
  timeout 300 python tools/build_profile.py $spec 5 2>&1 | grep BUILD
done | tee $o/builds.txt
timeout 120 python tools/time_refit.py 1000000 2>&1 | grep -v amdgpu.ids | tee $o/refit.txt
