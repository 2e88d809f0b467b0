#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root:  bash tools/profile_round.sh r01_g
# 1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time)
# 2./3. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) of one bench step -> HBM bytes per launch
# Outputs land in gpurun_out/<tag>/; tools/summarize_profiles.py turns them into the files kept under profiles/.
set -e
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/write.log 2>&1
cd $ROOT
python3 bench.py > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
grep '^{' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
ls -R $OUT | head -40
