#!/bin/bash
# scratch: the round-4 A/B runs on the GPU box (output under gpurun_out/<tag>/)
TAG=${1:-r04_x}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
summ() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1].split('/')[-1], "ms/step %.3f"%d["ms_per_step"], "distinct", d["config"].get("distinct_kmers"), {k:round(v,3) for k,v in d["roofline"]["kernels_ms_per_step"].items() if v>0.05})
PY
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout -k 10 240 $B > $OUT/$name.log 2>&1 || echo "$name failed rc=$?"
  grep -h "sk_reduce2\|wave clocks" $OUT/$name.log | tail -3
  summ $OUT/$name.log
done
