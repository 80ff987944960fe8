#!/bin/bash
# build_msda_variant.sh NAME [-DMACRO]...: libtamtr_NAME.so under tam-tr_amd/csrc/variants with msdeform.hip rebuilt with the given macros
# (timing ablations: run with TAMTR_HIP_LIB=.../variants/libtamtr_NAME.so)
set -e
cd "$(dirname "$0")/../tam-tr_amd/csrc"
mkdir -p variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c msdeform.hip -o variants/msdeform_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtamtr_$name.so $(ls *.o | grep -v '^msdeform.o$') variants/msdeform_$name.o
echo "built $name"
