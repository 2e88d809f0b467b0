#!/bin/bash
# scratch: the command of the current GPU experiment (rewritten per run)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -4 gpurun_out/t.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step'])"
