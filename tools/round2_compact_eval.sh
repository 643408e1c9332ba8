#!/usr/bin/env bash
# First GPU call of round 2 (developer script, not part of the product): evaluate the EXPERIMENTAL compact traversal records,
# which were written at the end of round 1 without a GPU. Everything is bounded by `timeout`; outputs go to gpurun_out/compact/.
#   gpurun --timeout 1500 -- 'bash tools/round2_compact_eval.sh'
# Order: cheapest and most informative first, so a partial run is still useful.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/compact
mkdir -p "$out"
export TMPDIR=/tmp

# 1. parity of the two kernels against each other and the oracle + ms side by side (small first: a wrong kernel shows here)
timeout 300 python tools/compact_pairs_gpu.py soup 100000 1048576 > "$out/probe_soup100k.log" 2>&1; echo "probe soup100k rc=$?" | tee -a "$out/summary.log"
timeout 500 python tools/compact_pairs_gpu.py soup 1000000 8388608 > "$out/probe_soup1m.log" 2>&1; echo "probe soup1m rc=$?" | tee -a "$out/summary.log"
timeout 400 python tools/compact_pairs_gpu.py sponza 262144 8388608 > "$out/probe_sponza.log" 2>&1; echo "probe sponza rc=$?" | tee -a "$out/summary.log"
timeout 500 python tools/compact_pairs_gpu.py terrain 1000000 8388608 > "$out/probe_terrain.log" 2>&1; echo "probe terrain rc=$?" | tee -a "$out/summary.log"
timeout 500 python tools/compact_pairs_gpu.py spheres64 1000000 4194304 > "$out/probe_spheres64.log" 2>&1; echo "probe spheres64 (double) rc=$?" | tee -a "$out/summary.log"
tail -n 5 "$out"/probe_*.log | tee -a "$out/summary.log"

# 2. the whole GPU suite and a fuzz campaign with the switch on (same bit-exact bar as the default path)
BVH_AMD_PAIRS=compact timeout 600 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu_compact.log" 2>&1; echo "pytest (compact) rc=$?" | tee -a "$out/summary.log"
tail -n 3 "$out/pytest_gpu_compact.log" | tee -a "$out/summary.log"
BVH_AMD_PAIRS=compact timeout 200 python tools/fuzz_campaign.py 1000 3000 120 > "$out/fuzz_compact.log" 2>&1; echo "fuzz (compact) rc=$?" | tee -a "$out/summary.log"
tail -n 2 "$out/fuzz_compact.log" | tee -a "$out/summary.log"

# 3. the bench line both ways, and the kernel trace of the compact run
timeout 400 python bench.py --no-cpu-baseline > "$out/bench_pairnode.json" 2> "$out/bench_pairnode.err"; echo "bench pairnode rc=$?" | tee -a "$out/summary.log"
BVH_AMD_PAIRS=compact timeout 400 python bench.py --no-cpu-baseline > "$out/bench_compact.json" 2> "$out/bench_compact.err"; echo "bench compact rc=$?" | tee -a "$out/summary.log"
cat "$out/bench_pairnode.json" "$out/bench_compact.json" | tee -a "$out/summary.log"
(cd /tmp && BVH_AMD_PAIRS=compact timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- python "$OLDPWD/bench.py" --steps 10 --no-cpu-baseline > "$OLDPWD/$out/prof_bench.log" 2>&1); echo "rocprof rc=$?" | tee -a "$out/summary.log"

# 4. (optional, ~9 GPU-minutes: run separately if 1.-3. look good) the refill / leaf-parking thresholds again with the cheaper node fetch:
#    BVH_AMD_PAIRS=compact SWEEP_REFILL=48,54,58 SWEEP_LEAF=4,8,16 timeout 900 python tools/sweep_thresholds.py > gpurun_out/compact/sweep.log 2>&1
