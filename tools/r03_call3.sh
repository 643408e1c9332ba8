#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PROBE_MODES="0:64,0:48,0:28,0:16,4:64,4:48,4:28,4:16" timeout 300 python tools/tcp_probe.py $o/tcp_probe3.txt > /dev/null 2>$o/tcp_probe3.err; echo "tcp_probe rc=$?"
grep -E "16KiB|2MiB" $o/tcp_probe3.txt
timeout 900 python tools/coop_probe.py $o/coop_probe3.txt > /dev/null 2>$o/coop_probe3.err; echo "coop_probe rc=$?"
cat $o/coop_probe3.txt
