#!/bin/bash
mkdir -p gpurun_out/r03_z
timeout -k 10 400 python bench.py > gpurun_out/r03_z/bench.log 2>&1
grep '^{' gpurun_out/r03_z/bench.log | tail -1 > gpurun_out/r03_z/bench.json
python3 -c "import json; d=json.load(open('gpurun_out/r03_z/bench.json')); print(d['ms_per_step'], d['extra']['cold'], d['cpu_baseline']['value'])"
timeout -k 10 400 python bench.py 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra']['cold'])"
