#!/bin/bash
mkdir -p gpurun_out
( timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --genome 800000000 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lowdup', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" ) 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -4 gpurun_out/t.log
