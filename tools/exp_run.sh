#!/bin/bash
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1
echo "tests exit $?" >> gpurun_out/t.log
tail -4 gpurun_out/t.log
