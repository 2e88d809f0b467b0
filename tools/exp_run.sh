#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_index.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
for v in prev cur prev cur; do
L=$PWD/ab/lib$v.so; [ $v = cur ] && L=$PWD/kmerind_amd/libkmerind_hip.so
KMERIND_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step']['sk_scatter'])"
done
