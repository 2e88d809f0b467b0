#!/bin/bash
mkdir -p gpurun_out
KMI_SPARSE_MIN=4611686018427387904 timeout -k 10 400 python bench.py --force-dist --dist-mode combine --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('combine, dense adopt', round(d['ms_per_step'],3))"
timeout -k 10 400 python bench.py --force-dist --dist-mode combine --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('combine, sparse adopt', round(d['ms_per_step'],3))"
timeout -k 10 400 python bench.py --force-dist --dist-mode superkmer --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('superkmer kmi', round(d['ms_per_step'],3), d['config'].get('transport'))"
