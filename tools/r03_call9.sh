#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/rule_check.py $o/rule_check.txt > /dev/null 2>$o/rule_check.err; echo "rule_check rc=$?"
grep -E "^##|auto" $o/rule_check.txt
