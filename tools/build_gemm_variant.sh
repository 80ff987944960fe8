#!/bin/bash
# build_gemm_variant.sh NAME [-DMACRO=..]...: libtamtr_NAME.so under tam-tr_amd/csrc/variants with gemm_bf16.hip rebuilt with the given macros
set -e
cd "$(dirname "$0")/../tam-tr_amd/csrc"
mkdir -p variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c gemm_bf16.hip -o variants/gemm_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtamtr_$name.so $(ls *.o | grep -v '^gemm_bf16.o$') variants/gemm_$name.o
echo "built $name"
