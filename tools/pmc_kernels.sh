#!/bin/bash
# SQ counters of one bench step per kernel (run on the GPU box from the repo root):  bash tools/pmc_kernels.sh r02_x
# separate rocprofv3 --pmc passes, kernel-trace only (the counters do not fit one pass)
set -e
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void kmi::", "").replace("kmi::", "")
        out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(out, open("$OUT/sq_counters.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print(k[:60], {a: "%.3g" % b for a, b in v.items()})
PY
