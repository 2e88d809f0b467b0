#!/bin/bash
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step'])"
done
timeout -k 10 900 python -m pytest tests/test_gpu_index.py -m gpu -x -q 2>&1 | tail -2
