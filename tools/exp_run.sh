#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_position_index.py tests/test_gpu_comm.py tests/test_gpu_facade.py -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -4 gpurun_out/t.log
