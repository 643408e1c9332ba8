#!/usr/bin/env bash
# Developer script: rocprofv3 passes of tools/config_probe.py (the kernels of BASELINE configs[2], [3], [4]) — one --kernel-trace
# --stats pass and SEPARATE --pmc passes (counters only, with --kernel-trace; never combined with sys / hip traces).
#   bash tools/profile_configs.sh <outdir under the repo> [configs] [quality of the 10M tree]
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="$here/$1"; cfgs="${2:-anyhit,spheres64,shard10m}"; q10="${3:-1}"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
python "$here/tools/config_probe.py" "$cfgs" 5 "$q10" > "$out/unprofiled.jsonl" 2> "$out/unprofiled.err"; echo "unprofiled rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$here/tools/config_probe.py" "$cfgs" 5 "$q10" > "$out/stats.log" 2>&1; echo "stats rc=$?"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$out/p$i" -- python "$here/tools/config_probe.py" "$cfgs" 2 "$q10" > "$out/p$i.log" 2>&1; echo "pass $i ($pass) rc=$?"
done
python "$here/tools/pmc_summary.py" "$out" trace_kernel 2 > "$out/pmc_summary.csv" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
python "$here/tools/last_dispatch_stats.py" "$(find "$out/stats" -name "*kernel_trace.csv" | head -1)" 5 > "$out/kernel_last5.csv" 2>&1
# keep the merged directory small: the raw per-dispatch CSVs are summarised above
find "$out" -name "*counter_collection.csv" -size +2M -delete; find "$out" -name "*kernel_trace.csv" -size +2M -delete
ls "$out"
