#!/usr/bin/env bash
# round 3, GPU call 5: bench line with the new roofline + fair CPU baseline, PMC passes for it, full GPU test-suite
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/pmc_traffic.py --out $o/pmc_traffic.json > $o/pmc_traffic.log 2>&1; echo "pmc_traffic rc=$?"; tail -2 $o/pmc_traffic.log
cp $o/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py > $o/bench.json 2>$o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err
python -c "
import json; d=json.load(open('$o/bench.json')); r=d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'pass', r['pass_ms'], r['pass_split_ms'])
print('roofline', r['achieved'], r['peak'], r['frac'], r['binding_level'], r['model_ms'], r['sum_of_levels_ms']); print(json.dumps(r['levels'])); print(json.dumps(r['probe']))
print('plan', r['launch_plan']); print('build', d['build']['all_qualities_ms'], d['build']['high'])
c=d['cpu_baseline']; print('cpu', c['value'], c['threads_used'], c['usable_cpus'], c['mrays_s_per_thread'], c['thread_sweep_mrays_s'], c['build_mtris_s'], c['gpu_matches_cpu_hits'], c['gpu_tree_equals_cpu_tree'])
"
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest_gpu.log
