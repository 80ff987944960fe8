#!/bin/bash
# round 4, call BG: wstat GEMM ablations (timing-only builds: results are garbage) - is the kernel bound by memory or by its own synchronisation?
V=$PWD/tam-tr_amd/csrc/variants
for k in base nold nost noldst lsfnoldst base; do echo "== $k"; TAMTR_HIP_LIB=$V/libtamtr_$k.so timeout -k 10 120 python3 tools/gemm_ab.py 2>&1 | tail -1 | cut -c100-400 || exit 1; done
