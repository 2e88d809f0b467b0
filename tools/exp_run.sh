#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_index.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -3 gpurun_out/t.log
for v in head cur; do
L=$PWD/ab/lib$v.so; [ $v = cur ] && L=$PWD/kmerind_amd/libkmerind_hip.so
KMERIND_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/b_$v.log 2>&1 ; tail -1 gpurun_out/b_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
