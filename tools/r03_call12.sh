#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/run_configs.py > $o/baseline_configs.jsonl 2>$o/baseline_configs.err; echo "configs rc=$?"; cut -c1-230 $o/baseline_configs.jsonl
rm -rf $o/configs; timeout 1500 bash tools/profile_configs.sh $o/configs anyhit,spheres64,shard10m 2 > $o/profile_configs.log 2>&1; echo "profile rc=$?"
cat $o/configs/unprofiled.jsonl | cut -c1-300; cat $o/configs/kernel_last5.csv | cut -c1-220
