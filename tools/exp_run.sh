#!/bin/bash
for v in base two6 two7 lin7 two7o256 base; do
KMERIND_HIP_LIB=$PWD/ab/lib$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step']['sk_reduce'])"
done
