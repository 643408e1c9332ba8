set -x
python -m pytest tests/test_gpu_build.py -x -q -k "high or reinsertion or optimize" 2>&1 | tail -5
python tools/heap_head_check.py 1000000 4000000 2>&1 | grep RESULT
