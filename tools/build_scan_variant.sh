#!/bin/bash
# build_scan_variant.sh NAME [-DMACRO=..]...: libtamtr_NAME.so under tam-tr_amd/csrc/variants with selscan.hip rebuilt with the given macros
set -e
cd "$(dirname "$0")/../tam-tr_amd/csrc"
mkdir -p variants /tmp/st
name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17"
hipcc $F -fPIC -Wall -Wno-unused-function "$@" -c selscan.hip -o variants/selscan_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtamtr_$name.so $(ls *.o | grep -v '^selscan.o$') variants/selscan_$name.o
hipcc $F "$@" -S --cuda-device-only selscan.hip -o /tmp/st/$name.s 2>/dev/null
echo "$name: $(grep -A8 'selscan_bwd_kernelILb1E' /tmp/st/$name.s | grep -E 'vgpr_count|private_segment_fixed' | tr -d '\n' | sed 's/  */ /g')"
