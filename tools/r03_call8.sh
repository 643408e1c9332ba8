#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== 7 waves"; timeout 600 python tools/coop_probe.py $o/coop_w7.txt > /dev/null 2>&1; grep -E "refill=36 leaf=12|refill=12 leaf=12|refill=20 leaf=20" $o/coop_w7.txt
echo "== 8 waves"; BVH_AMD_LIB=$PWD/tools/bin/libbvh_amd_w8.so timeout 600 python tools/coop_probe.py $o/coop_w8.txt > /dev/null 2>&1; grep -E "refill=36 leaf=12|refill=12 leaf=12|refill=20 leaf=20" $o/coop_w8.txt
