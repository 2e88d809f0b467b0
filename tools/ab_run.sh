#!/bin/bash
# Same-box A/B runs of the bench (on the GPU box, from the repo root): every further argument is name:ENV=... (space separated
# assignments), e.g.  tools/ab_run.sh r04_x "base:X=1" "variant:KMERIND_HIP_LIB=ab/libvariant.so KMI_SK_REDUCE=2"
# (libraries from tools/ab_build.sh). One line per run: ms per step, distinct k-mers, per-kernel ms; logs under gpurun_out/<tag>/.
# BENCH_ARGS="--genome 800000000" adds bench arguments.
TAG=${1:-r04_x}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra $BENCH_ARGS"
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
