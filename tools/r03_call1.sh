#!/usr/bin/env bash
# round 3, GPU call 1: new parity tests, RCCL entry points, L1 / L2 / fabric microbenchmarks, baseline bench line, FETCH_SIZE
# calibration, profiles of the any-hit / f64-sphere / 10M-shard kernels
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showid 2>/dev/null | head -20 > $o/devices.txt
python -c "import torch; print(torch.cuda.device_count(), torch.cuda.get_device_name(0)); import os; print(len(os.sched_getaffinity(0)), os.cpu_count()); print(open('/sys/fs/cgroup/cpu.max').read())" > $o/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rccl.py tests/test_gpu_wire.py -x -q > $o/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -5 $o/pytest_new.log
timeout 300 python tools/tcp_probe.py $o/tcp_probe.txt > /dev/null 2>$o/tcp_probe.err; echo "tcp_probe rc=$?"
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_baseline.json 2>$o/bench_baseline.err; echo "bench rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$o/fetch_cal -- python $GRAFT_REPO_ROOT/tools/fetch_calibration.py > $GRAFT_REPO_ROOT/$o/fetch_cal.json 2>$GRAFT_REPO_ROOT/$o/fetch_cal.err); echo "fetch_cal rc=$?"
python tools/fetch_calibration.py --read $o/fetch_cal $(python -c "import json;print(json.load(open('$o/fetch_cal.json'))['bytes_per_launch'])") > $o/fetch_cal_factor.json 2>&1
cat $o/fetch_cal_factor.json
timeout 1200 bash tools/profile_configs.sh $o/configs anyhit,spheres64,shard10m 2 > $o/profile_configs.log 2>&1; echo "profile rc=$?"
tail -3 $o/profile_configs.log
