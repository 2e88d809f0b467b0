#!/bin/bash
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/trace/log.txt 2>&1
