#!/bin/bash
# build the current source tree's library as ab/lib<tag>.so (for same-box A/B runs: KMERIND_HIP_LIB=ab/lib<tag>.so)
set -e
TAG=$1; shift
cd /root/repo/kmerind_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c kmi_index.hip -o /tmp/kmi_index_$TAG.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/ab/lib$TAG.so build/kmi_api.hip.o build/kmi_extract.hip.o build/kmi_fasta.hip.o /tmp/kmi_index_$TAG.o build/kmi_comm.hip.o build/kmi_synth.cpp.o -lpthread -ldl
ls -la /root/repo/ab/lib$TAG.so
