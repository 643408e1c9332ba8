#!/usr/bin/env bash
# Developer script: separate rocprofv3 --pmc passes (counters only, with --kernel-trace; never combined with sys/hip traces) of
# tools/trace_probe.py, summarised per kernel into one CSV.   bash tools/pmc_passes.sh <outdir> "<pass1 counters>" "<pass2 counters>" ...
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="$1"; shift
mkdir -p "$here/$out"
export TMPDIR=/tmp PROBE_CLOSEST_ONLY=1
i=0
for pass in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$here/$out/p$i" -- python "$here/tools/trace_probe.py" ${PROBE_ARGS:-soup 1000000 8388608 2 2} > "$here/$out/p$i.log" 2>&1); echo "pass $i ($pass) rc=$?"
done
python "$here/tools/pmc_summary.py" "$here/$out" trace_kernel > "$here/$out/summary.csv" 2>&1
cat "$here/$out/summary.csv"
