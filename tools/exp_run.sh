#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f examples/facade_extras
timeout -k 10 900 python -m pytest tests/test_gpu_fasta.py tests/test_gpu_facade.py -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -15 gpurun_out/t.log
