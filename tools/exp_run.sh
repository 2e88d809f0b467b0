#!/bin/bash
mkdir -p gpurun_out
rm -rf gpurun_out/r03_z
timeout -k 10 500 bash tools/profile_round.sh r03_z > gpurun_out/prof_z.log 2>&1; echo "profile rc $?"
timeout -k 10 300 bash tools/pmc_kernels.sh r03_z > gpurun_out/pmc_z.log 2>&1; echo "pmc rc $?"
{
( timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --genome 800000000 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('low duplication (10 M reads over 800 Mbp):', round(d['ms_per_step'],2), 'ms', d['roofline']['kernels_ms_per_step'])" ) 2>&1 | tail -1
timeout -k 10 200 python tools/pos_bench.py 10000000 position 2>&1 | tail -2
timeout -k 10 200 python tools/pos_bench.py 10000000 posqual 2>&1 | tail -2
timeout -k 10 300 python tools/config4_bench.py 1000 2>&1 | tail -2
timeout -k 10 200 python tools/dbg_bench.py 10000000 100000000 31 2>&1 | tail -3
timeout -k 10 300 python tools/sk_dist_emul.py 8 2>&1 | tail -2
( timeout -k 10 300 python bench.py --force-dist --dist-mode superkmer --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank rehearsal over RCCL, --transport kmi:', round(d['ms_per_step'],2), 'ms', d['roofline']['kernels_ms_per_step'])" ) 2>&1 | tail -1
} > gpurun_out/r03_z/secondary.txt 2>&1
python3 -c "import json; d=json.load(open('gpurun_out/r03_z/bench.json')); print(d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['extra']['cold'])"
