#!/bin/bash
# build_variant.sh FILE NAME [-DMACRO]...: libtamtr_NAME.so under tam-tr_amd/csrc/variants with csrc/FILE.hip rebuilt with the given macros
# (timing ablations and parameter A/Bs: run with TAMTR_HIP_LIB=<repo>/tam-tr_amd/csrc/variants/libtamtr_NAME.so; ablated builds compute garbage)
set -e
cd "$(dirname "$0")/../tam-tr_amd/csrc"
mkdir -p variants
file=$1; name=$2; shift 2
extra=""; [ "$file" = imgaug ] && extra="-ffp-contract=off"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c $file.hip -o variants/${file}_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtamtr_$name.so $(ls *.o | grep -v "^$file.o$") variants/${file}_$name.o
echo "built $name"
