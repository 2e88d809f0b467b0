#!/bin/bash
# Build the current source tree's library a second time as ab/lib<tag>.so, with extra compiler flags (tuning knobs, phase clocks),
# for same-box A/B runs:   tools/ab_build.sh timing -DKMI_R2_TIMING    then    KMERIND_HIP_LIB=ab/libtiming.so python bench.py ...
# Only kmi_index.hip is recompiled (every knob lives there); the other objects come from the regular build (run make first).
set -euo pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/kmerind_amd/csrc
mkdir -p "$ROOT/ab" "$SRC/build"
OBJ=$SRC/build/kmi_index_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -I"$ROOT/include" "$@" -c "$SRC/kmi_index.hip" -o "$OBJ"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/lib$TAG.so" "$SRC/build/kmi_api.hip.o" "$SRC/build/kmi_extract.hip.o" \
  "$SRC/build/kmi_fasta.hip.o" "$OBJ" "$SRC/build/kmi_comm.hip.o" "$SRC/build/kmi_synth.cpp.o" -lpthread -ldl
ls -la "$ROOT/ab/lib$TAG.so"
