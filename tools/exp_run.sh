#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step'])"
timeout -k 10 500 python tools/sk_dist_emul.py 8 2>&1 | tail -2 | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_index.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
