#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_index.py -m gpu -x -q -k "fine_buckets or superkmer" > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -15 gpurun_out/t.log
