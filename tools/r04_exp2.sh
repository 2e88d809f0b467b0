#!/bin/bash
# scratch: like r04_exp.sh with extra bench arguments:  tools/r04_exp2.sh TAG "bench args" name:ENV ...
TAG=$1; shift; ARGS=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra $ARGS"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout -k 10 300 $B > $OUT/$name.log 2>&1 || echo "$name failed rc=$?"
  grep -h "sk_reduce2:" $OUT/$name.log | tail -1
  python - $OUT/$name.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1].split('/')[-1], "ms/step %.3f"%d["ms_per_step"], "distinct", d["config"].get("distinct_kmers"), {k:round(v,3) for k,v in d["roofline"]["kernels_ms_per_step"].items() if v>0.05})
PY
done
