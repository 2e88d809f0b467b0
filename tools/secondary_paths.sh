#!/bin/bash
# The paths beside the headline build, on the GPU box from the repo root:  bash tools/secondary_paths.sh r04_z
# -> gpurun_out/<tag>/secondary_paths.txt (one line per path + its kernels) and rocprofv3 kernel-stats CSVs of the config-4 and
#    position + quality builds (gpurun_out/<tag>/{config4,posqual}/); copy what is to be kept into profiles/.
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
T=$OUT/secondary_paths.txt
: > $T
run() { echo "## $1" >> $T; shift; timeout -k 10 400 "$@" 2>&1 | grep -v "amdgpu.ids\|^\[Gloo\]\|socket.cpp" >> $T; }
lowdup() { python - "$@" <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra", "--genome", "800000000"], capture_output=True, text=True).stdout
for l in out.splitlines():
    if l.startswith("{"):
        d = json.loads(l); print("low duplication (10 M reads over 800 Mbp): %.2f ms" % d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
}
echo "## low duplication" >> $T; lowdup >> $T 2>&1
run "position index" python tools/pos_bench.py 10000000 position
run "position + quality index" python tools/pos_bench.py 10000000 posqual
run "config 4 (1 Gbp FASTA, k = 63 DNA5, PositionIndex)" python tools/config4_bench.py
run "FASTA count index" python tools/fasta_bench.py
run "de Bruijn nodes" python tools/dbg_bench.py
run "de Bruijn nodes of config 2's input" python tools/dbg_bench.py 10000000 100000000
echo "## de Bruijn nodes of config 2's input, tuple path (KMI_DBG_SUPERKMER=0)" >> $T
KMI_DBG_SUPERKMER=0 timeout -k 10 400 python tools/dbg_bench.py 10000000 100000000 2>&1 | grep -v "amdgpu.ids" | head -2 >> $T
run "config 2 from pinned host memory (kmi_index_build_host)" python tools/host_build_probe.py
run "two builds of config 2 on two streams" python tools/overlap_probe.py
echo "## write runs of a partition pass: aligned lines against runs that start anywhere (tools/write_runs.hip)" >> $T
(/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/write_runs.hip -o /tmp/write_runs 2>/dev/null && timeout -k 10 120 /tmp/write_runs | tail -12) >> $T 2>&1
run "one rank of an 8-rank build (front end + consume)" python tools/sk_dist_emul.py 8
echo "## one-rank rehearsal over RCCL (--force-dist --dist-mode superkmer --transport kmi)" >> $T
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --force-dist --dist-mode superkmer --transport kmi 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.2f ms' % d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" >> $T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config4 -o stats -- python3 $ROOT/tools/config4_bench.py > $OUT/config4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/posqual -o stats -- python3 $ROOT/tools/pos_bench.py 10000000 posqual > $OUT/posqual.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/position -o stats -- python3 $ROOT/tools/pos_bench.py 10000000 position > $OUT/position.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dbg -o stats -- python3 $ROOT/tools/dbg_bench.py 10000000 100000000 > $OUT/dbg.log 2>&1
cd $ROOT
cat $T
