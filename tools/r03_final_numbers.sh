#!/usr/bin/env bash
# Developer script (round 3): the final numbers: PMC passes + bench line + rocprof kernel stats of the bench command (plan-search launches under their own symbols) + GPU suite
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/pmc_traffic.py --out $o/pmc_traffic.json > $o/pmc_traffic.log 2>&1; echo "pmc_traffic rc=$?"
cp $o/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py > $o/bench.json 2>$o/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$o/bench.json')); r=d['roofline']
print('value', d['value'], 'kernel', r['kernel_ms'], 'pass', r['pass_ms'], 'roofline', r['achieved'], r['peak'], r['frac'], r['binding_level'], r['model_ms'], r['sum_of_levels_ms'], 'traffic', r['traffic'])
print({k: (v['ms'], v.get('grec_s')) for k, v in r['levels'].items()}, r['probe']['l2_and_fabric_times'], r['probe']['active_lanes'], r['kernel'])
c=d['cpu_baseline']; print('cpu', c['value'], c['threads_used'], c['build_mtris_s'], c['gpu_matches_cpu_hits'], c['gpu_tree_equals_cpu_tree']); print('build', d['build']['all_qualities_ms'], d['build']['high']['us_per_replacement'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/bench_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$o/bench_stats.json 2>$GRAFT_REPO_ROOT/$o/bench_stats.err); echo "bench stats rc=$?"
find $o/bench_stats -name "*kernel_stats.csv" -exec cp {} $o/bench_kernel_stats.csv \;
python tools/last_dispatch_stats.py "$(find $o/bench_stats -name '*kernel_trace.csv' | head -1)" 10 trace_kernel > $o/bench_kernel_last10.csv; cat $o/bench_kernel_last10.csv | cut -c1-200
grep -E "trace_kernel|ray_keys|k_radix" $o/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c30-200
find $o/bench_stats -name "*kernel_trace.csv" -delete
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
