#!/usr/bin/env bash
# Developer script: rocprofv3 --kernel-trace --stats of single DefaultBuilder modes.  bash tools/build_profile.sh <outdir> "<scene n q pool>" ...
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="$here/$1"; shift
mkdir -p "$out"
export TMPDIR=/tmp
for spec in "$@"; do
  tag=$(echo $spec | tr ' ' '_')
  timeout 200 python "$here/tools/build_profile.py" $spec 3 | grep BUILD | tee -a "$out/timings.log"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python "$here/tools/build_profile.py" $spec 1 > "$out/$tag.log" 2>&1)
  f=$(find "$out/$tag" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
done
