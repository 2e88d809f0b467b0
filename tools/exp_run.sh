#!/bin/bash
mkdir -p gpurun_out
for n in 8 4 2; do
KMI_SLACK_DEBUG=1 timeout -k 10 500 python tools/sk_dist_emul.py $n > gpurun_out/emul$n.log 2>&1; grep "fine buckets with room" gpurun_out/emul$n.log | sort | uniq -c | head -4; tail -2 gpurun_out/emul$n.log | cut -c1-300
done
