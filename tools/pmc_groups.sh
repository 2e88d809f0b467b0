#!/bin/bash
# counters of one bench step per kernel, one rocprofv3 --pmc pass per quoted group:
#   bash tools/pmc_groups.sh TAG "CTR_A CTR_B" "CTR_C CTR_D" ...      (run on the GPU box from the repo root)
set -e
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/g$i.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void kmi::", "").replace("kmi::", "")
        out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(out, open("$OUT/counters.json", "w"), indent=1)
for k, v in sorted(out.items()):
    if k.startswith("sk_"):
        print(k[:40], {a: "%.3g" % b for a, b in sorted(v.items())})
PY
