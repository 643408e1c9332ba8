#!/usr/bin/env bash
set -u
o=gpurun_out/r03; mkdir -p $o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print('devices', torch.cuda.device_count())" > $o/two_gpu_devices.txt 2>&1
cat $o/two_gpu_devices.txt
rocm-smi --showtopo 2>/dev/null | head -40 >> $o/two_gpu_devices.txt
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -rs > $o/pytest_rccl_2gpu.log 2>&1; echo "pytest rccl rc=$?"
tail -8 $o/pytest_rccl_2gpu.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 > $o/bench_2gpu.json 2>$o/bench_2gpu.err; echo "bench 2gpu rc=$?"
tail -c 600 $o/bench_2gpu.json; tail -5 $o/bench_2gpu.err
