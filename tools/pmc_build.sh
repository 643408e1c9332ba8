#!/usr/bin/env bash
# Developer script: one rocprofv3 --pmc pass (counters only + --kernel-trace) of tools/build_profile.py, per-kernel averages for kernels matching $MATCH.
#   MATCH=k_heap bash tools/pmc_build.sh <outdir> "<counters>" <build_profile args...>
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
out="$here/$1"; shift
counters="$1"; shift
mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$out/p" -- python "$here/tools/build_profile.py" "$@" > "$out/run.log" 2>&1)
python "$here/tools/pmc_summary.py" "$out" "${MATCH:-k_heap}"
